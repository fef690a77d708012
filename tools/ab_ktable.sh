#!/bin/bash
# same-box comparison of the per-kernel-class table under several values of one environment variable, ONE plan table for all (measured under the first value):
#   bash tools/ab_ktable.sh VAR v1 v2 ...      (inside gpurun; writes gpurun_out/ktab_VAR_<v>.json and prints the classes that moved)
VAR=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
env $VAR=$1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --repeats 1 --save-plans gpurun_out/ktab_plans.txt > /dev/null 2>&1
for v in "$@"; do
  env $VAR=$v python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --plans gpurun_out/ktab_plans.txt --kernel-table gpurun_out/ktab_${VAR}_$v.json 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$VAR=$v ms/step', ['%.3f' % x for x in d['timing']['runs_ms_per_step']])"
done
python3 - "$VAR" "$@" <<'PY'
import json, sys
var, vals = sys.argv[1], sys.argv[2:]
tabs = {v: json.load(open(f"gpurun_out/ktab_{var}_{v}.json")) for v in vals}
names = sorted({k for t in tabs.values() for k in t["kernels"]})
print(f"{'kernel class':46s}" + "".join(f"{v[:14]:>26s}" for v in vals))
for n in names:
    row = []
    for v in vals:
        t = tabs[v]; k = t["kernels"].get(n); s = t["steps_profiled"]
        row.append(f"{k['launches'] / s:7.0f} x {1e3 * k['ms'] / max(1, k['launches']):6.1f}us = {k['ms'] / s:6.3f}" if k and k["launches"] else " " * 26)
    print(f"{n[:46]:46s}" + "".join(row))
for v in vals:
    t = tabs[v]; print(v, "sum of classes ms/step:", round(sum(k["ms"] for k in t["kernels"].values()) / t["steps_profiled"], 3))
PY
