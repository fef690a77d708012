for r in "" "2048x1280x1280=2;2048x1280x5120=2" "2048x1280x1280=3;2048x1280x5120=3" "2048x1280x1280=5;2048x1280x5120=5" "2048x1280x1280=0;2048x1280x5120=0" "2048x1280x1280=1;2048x1280x5120=1"; do
  echo "RULES=$r"
  IA2P_GEMM_RULES="$r" timeout 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "false>|ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/ms_per_step \1/' | cut -c1-120
done
