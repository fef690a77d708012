cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/geglu_tile_diag.py > gpurun_out/r06d_diag.txt 2>&1
grep -c differ gpurun_out/r06d_diag.txt; head -5 gpurun_out/r06d_diag.txt
(timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "geglu or gemm" -x 2>&1 | tail -5) > gpurun_out/r06d_pytest.txt
tail -5 gpurun_out/r06d_pytest.txt
(VARIANTS=18,27 timeout 300 python tools/ffin_ksweep.py) > gpurun_out/r06d_ksweep.txt 2>&1
cat gpurun_out/r06d_ksweep.txt
IA2P_TUNE_LOG=1 timeout 900 python bench.py --tune --save-plans gpurun_out/r06d_plans.txt --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r06d_bench.json 2> gpurun_out/r06d_bench.err
tail -1 gpurun_out/r06d_bench.json | cut -c1-600
grep -E "ff_in|launches/step" gpurun_out/r06d_bench.err | head -12
grep "10240" gpurun_out/r06d_plans.txt | tr ';' '\n' | grep "10240\|5120,640"
IA2P_TUNE_EXCLUDE=27 timeout 900 python bench.py --tune --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/r06d_bench_no27.json 2> gpurun_out/r06d_bench_no27.err
tail -1 gpurun_out/r06d_bench_no27.json | cut -c1-400
timeout 900 python bench.py --plans gpurun_out/r06d_plans.txt --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/r06d_bench_27b.json 2> gpurun_out/r06d_bench_27b.err
tail -1 gpurun_out/r06d_bench_27b.json | cut -c1-400
