# in-situ A/B of GEMM variants: IA2P_GEMM_RULES overrides the tile choice for exact MxNxK shapes. EXPERIMENT knob: build the library with IA2P_EXTRA_FLAGS=-DIA2P_EXPERIMENTS first (the product build ignores it and says so on stderr)
for r in "" "2048x1280x11520=4;2048x1280x17280=4;2048x1280x23040=4"; do
  echo "RULES=$r"
  IA2P_GEMM_RULES="$r" timeout 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "true, 2>|ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/ms_per_step \1/' | cut -c1-120
done
