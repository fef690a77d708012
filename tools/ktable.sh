#!/bin/bash
# per-kernel-class tables (HIP events around every launch) for B = 8 (cfg 3) and B = 1 (cfg 2): usage inside gpurun: bash tools/ktable.sh <tag>
TAG=${1:-kt}
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-secondary --kernel-table gpurun_out/${TAG}_b8.json > gpurun_out/${TAG}_b8.out 2> gpurun_out/${TAG}_b8.err
python3 bench.py --batch 1 --ctx 77 --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-secondary --kernel-table gpurun_out/${TAG}_b1.json > gpurun_out/${TAG}_b1.out 2> gpurun_out/${TAG}_b1.err
python3 - <<PY
import json
for t in ("b8", "b1"):
    d = json.load(open("gpurun_out/${TAG}_%s.json" % t))
    n = d["steps_profiled"]
    print(t, "steps", n)
    rows = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])
    tot = sum(v["ms"] for _, v in rows)
    for k, v in rows:
        print("  %-52s %6.1f /step  %7.2f us  %6.3f ms/step  %6.1f TF" % (k[:52], v["launches"] / n, 1e3 * v["ms"] / v["launches"], v["ms"] / n, v["flops"] / v["ms"] / 1e9 if v["ms"] else 0))
    print("  total %.3f ms/step" % (tot / n))
PY
