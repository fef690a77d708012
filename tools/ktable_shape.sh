#!/bin/bash
# per-kernel-class table of one bench shape: bash tools/ktable_shape.sh <tag> <bench.py args...>     (inside gpurun)
TAG=$1; shift
cd $GRAFT_REPO_ROOT
python3 bench.py "$@" --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-secondary --no-box-probe --kernel-table gpurun_out/${TAG}.json > gpurun_out/${TAG}.out 2> gpurun_out/${TAG}.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}.json")); n = d["steps_profiled"]
rows = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"]); tot = sum(v["ms"] for _, v in rows)
for k, v in rows:
    print("  %-52s %6.1f /step  %7.2f us  %6.3f ms/step  %6.1f TF" % (k[:52], v["launches"] / n, 1e3 * v["ms"] / v["launches"], v["ms"] / n, v["flops"] / v["ms"] / 1e9 if v["ms"] else 0))
print("  total %.3f ms/step (events around every launch)" % (tot / n))
PY
python3 -c "import json; d=json.loads(open('gpurun_out/${TAG}.out').read().strip().splitlines()[-1]); print('ms/step', d['timing']['runs_ms_per_step'])"
