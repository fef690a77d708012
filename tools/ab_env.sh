#!/bin/bash
# same-box A/B of an environment switch on cfg 3 (and optionally cfg 2): usage inside gpurun: bash tools/ab_env.sh VAR [b1]
VAR=$1
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --repeats 1 --save-plans gpurun_out/ab_plans.txt > /dev/null 2>&1
for i in 1 2; do for v in 0 1; do
env $VAR=$v python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --plans gpurun_out/ab_plans.txt 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B8 $VAR=$v', d['timing']['runs_ms_per_step'])"
done; done
if [ "$2" = "b1" ]; then for v in 0 1; do
env $VAR=$v python3 bench.py --batch 1 --ctx 77 --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B1 $VAR=$v', d['timing']['runs_ms_per_step'])"
done; fi
