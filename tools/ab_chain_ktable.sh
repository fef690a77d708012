#!/bin/bash
# per-kernel-class tables of the chained feed-forward launch against the two launches (same box, same plans); diag runs give wrong results, timing only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --repeats 1 --save-plans gpurun_out/ktab_plans.txt > /dev/null 2>&1
for cfg in "0 0" "1 0" "1 12" "1 4"; do set -- $cfg
  env IA2P_CHAIN=$1 IA2P_CHAIN_DIAG=$2 python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-box-probe --plans gpurun_out/ktab_plans.txt --kernel-table gpurun_out/ktab_chain_$1_$2.json 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('chain=$1 diag=$2 ms/step', ['%.3f' % x for x in d['timing']['runs_ms_per_step']])"
  python3 - $1 $2 <<'PY'
import json, sys
t = json.load(open(f"gpurun_out/ktab_chain_{sys.argv[1]}_{sys.argv[2]}.json")); s = t["steps_profiled"]
for n, k in sorted(t["kernels"].items()):
    if "chain" in n or "128, 160, 2, false" in n or "128, 128, 2, false" in n:
        print(f"   {n[:52]:52s} {k['launches'] / s:5.0f} x {1e3 * k['ms'] / k['launches']:6.1f} us = {k['ms'] / s:6.3f} ms")
PY
done
