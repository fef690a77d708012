#!/bin/bash
# where the chained feed-forward launch loses: same box, same plans, IA2P_CHAIN=1 with the diagnostic switches of IA2P_CHAIN_DIAG (results not valid for d != 0)
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --repeats 1 --save-plans gpurun_out/ab_plans.txt > /dev/null 2>&1
run() { env "$@" python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --plans gpurun_out/ab_plans.txt 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', ['%.3f' % x for x in d['timing']['runs_ms_per_step']])"; }
run IA2P_CHAIN=0
for d in 0 1 2 4 3 7; do run IA2P_CHAIN=1 IA2P_CHAIN_DIAG=$d; done
run IA2P_CHAIN=0
