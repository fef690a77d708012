#!/bin/bash
# same-box A/B of the GroupNorm-fused convolutions (IA2P_GN_FUSE=1) against GroupNorm launches (=0): step time + the role / kernel tables of both
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
for m in 0 1; do
  IA2P_GN_FUSE=$m python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --kernel-table gpurun_out/${TAG}_ktable_gn$m.json $2 > gpurun_out/${TAG}_bench_gn$m.json 2> gpurun_out/${TAG}_bench_gn$m.err
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_gn$m.json").read().strip().splitlines()[-1])
print("GN_FUSE=$m", ["%.3f"%x for x in d["timing"]["runs_ms_per_step"]])
r=d["roofline"]
for k,v in r["roles"].items():
    print("   %-70s %6.1f launches %7.3f ms  %7.1f us/launch" % (k[:70], v["launches_per_step"], v["ms_per_step"], v["avg_launch_us"]))
t=json.load(open("gpurun_out/${TAG}_ktable_gn$m.json"))
for k,v in sorted(t["kernels"].items(), key=lambda kv:-kv[1]["ms"]):
    if "conv" in k or "true" in k or "gn_" in k:
        print("   K %-60s %6.1f launches %7.3f ms %7.1f us" % (k[:60], v["launches"]/t["steps_profiled"], v["ms"]/t["steps_profiled"], 1e3*v["ms"]/v["launches"]))
print("   conv region ms", r["conv_blocks"]["ms"])
PY
done
