#!/bin/bash
# same-box comparison of several values of one environment variable on cfg 3, each with its own autotune: bash tools/ab_vals.sh VAR v1 v2 ...
VAR=$1; shift
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --repeats 1 > /dev/null 2>&1
for i in 1 2; do for v in "$@"; do
env $VAR=$v python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B8 $VAR=$v', d['timing']['runs_ms_per_step'])"
done; done
