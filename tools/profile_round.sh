#!/bin/bash
# One profiling round on the GPU box: measured plans -> PMC traffic passes -> rocprofv3 kernel stats -> the bench line, all four on the
# SAME measured plan table (saved by the first run) so that the per-kernel figures of the artefacts describe the same launches.
# usage (inside gpurun): bash tools/profile_round.sh r01e        outputs under gpurun_out/<tag>_* (copy what is judged into profiles/)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd $R
# plans: the committed table (bench.py's default) unless PLANS=tune asks for an in-place measurement; either way saved so that every pass below runs the same kernels
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe ${PLANS:+--tune} --save-plans $R/gpurun_out/${TAG}_plans_cfg3.txt > /dev/null 2> $R/gpurun_out/${TAG}_plans.err
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --plans $R/gpurun_out/${TAG}_plans_cfg3.txt > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
# MFMA-pipe utilisation: one SQ/GRBM pass of its own, --kernel-trace only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_${TAG}_MFMA --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --plans $R/gpurun_out/${TAG}_plans_cfg3.txt > $R/gpurun_out/pmc_${TAG}_MFMA.log 2>&1
cd $R
python3 tools/pmc_mfma.py gpurun_out/pmc_${TAG}_MFMA profiles/${TAG}_pmc_mfma.json > gpurun_out/${TAG}_pmc_mfma.txt 2>&1
cp profiles/${TAG}_pmc_mfma.json gpurun_out/
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE profiles/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
cp profiles/${TAG}_pmc_traffic.json gpurun_out/
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --plans $R/gpurun_out/${TAG}_plans_cfg3.txt > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" -exec python3 tools/demangle_stats.py {} gpurun_out/${TAG}_kernel_stats_bench_cfg3.csv \;
python3 bench.py --steps 50 --warmup 3 --plans gpurun_out/${TAG}_plans_cfg3.txt --no-secondary --kernel-table gpurun_out/${TAG}_kernel_table.json > gpurun_out/${TAG}_bench_cfg3.json 2> gpurun_out/${TAG}_bench_cfg3.err
tail -1 gpurun_out/${TAG}_bench_cfg3.json | cut -c1-1800
grep -E "launches/step" gpurun_out/${TAG}_bench_cfg3.err | head -20
# keep the merge small: drop the raw per-dispatch traces
find gpurun_out/prof_${TAG} gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/pmc_${TAG}_MFMA -name "*counter_collection.csv" -delete
find gpurun_out/pmc_${TAG}_MFMA -name "*kernel_trace.csv" -delete
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json
