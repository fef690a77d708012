cd $GRAFT_REPO_ROOT
python3 bench.py --batch ${KB:-2} --latent 128 --ctx ${KC:-81} --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-secondary --kernel-table gpurun_out/kt_1024b2.json > gpurun_out/kt_1024b2.out 2> gpurun_out/kt_1024b2.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/kt_1024b2.json"))
n = d["steps_profiled"]
rows = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
for k, v in rows[:16]:
    print("  %-52s %6.1f /step  %7.2f us  %6.3f ms/step  %6.1f TF" % (k[:52], v["launches"] / n, 1e3 * v["ms"] / v["launches"], v["ms"] / n, v["flops"] / v["ms"] / 1e9 if v["ms"] else 0))
print("  total %.3f ms/step" % (tot / n))
PY
tail -1 gpurun_out/kt_1024b2.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
