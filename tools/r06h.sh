cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/geglu_tile_diag.py > gpurun_out/r06h_diag.txt 2>&1
grep -c "differ" gpurun_out/r06h_diag.txt; grep "differ" gpurun_out/r06h_diag.txt | head -8
(timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "geglu" 2>&1 | tail -3) > gpurun_out/r06h_pytest.txt
tail -3 gpurun_out/r06h_pytest.txt
timeout 300 tools/micro/geglu_clock > gpurun_out/r06h_geglu_clock.txt 2>&1; grep -v "group" gpurun_out/r06h_geglu_clock.txt
(VARIANTS=18,27 timeout 300 python tools/ffin_ksweep.py) > gpurun_out/r06h_ksweep.txt 2>&1
cat gpurun_out/r06h_ksweep.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err
tail -1 gpurun_out/r06h_bench.json | cut -c1-330
grep -E "gemm_geglu" gpurun_out/r06h_bench.err | head -3
