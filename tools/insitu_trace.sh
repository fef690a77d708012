#!/bin/bash
# raw per-dispatch kernel trace of a short bench run (in-situ kernel durations and the gaps between launches): usage inside gpurun: bash tools/insitu_trace.sh <tag> [bench flags]
TAG=${1:-tr}; shift
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --save-plans $R/gpurun_out/${TAG}_plans.txt "$@" > /dev/null 2> $R/gpurun_out/${TAG}_plans.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_${TAG} --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --plans $R/gpurun_out/${TAG}_plans.txt "$@" > $R/gpurun_out/trace_${TAG}.log 2>&1
cd $R
find gpurun_out/trace_${TAG} -name "*kernel_trace.csv" -exec cp {} gpurun_out/${TAG}_kernel_trace.csv \;
rm -rf gpurun_out/trace_${TAG}
ls -la gpurun_out/${TAG}_kernel_trace.csv; tail -2 gpurun_out/trace_${TAG}.log | cut -c1-300
