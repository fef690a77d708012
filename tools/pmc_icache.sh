#!/bin/bash
# instruction-cache counters per kernel over a few bench steps (one rocprofv3 --pmc pass of its own, --kernel-trace only): bash tools/pmc_icache.sh <tag>      (on the GPU box)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INST_CYCLES_VMEM\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES" | sort -u | tr '\n' ' ' > $R/gpurun_out/${TAG}_icache_counters_avail.txt
cat $R/gpurun_out/${TAG}_icache_counters_avail.txt; echo
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_${TAG}_ICACHE --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe > $R/gpurun_out/pmc_${TAG}_ICACHE.log 2>&1
cd $R
python3 - <<'P' $TAG
import csv, glob, sys, collections
tag = sys.argv[1]
files = glob.glob(f"gpurun_out/pmc_{tag}_ICACHE/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:24]
with open(f"gpurun_out/{tag}_pmc_icache.txt", "w") as o:
    for k, c in rows:
        req, hit, mis = c.get("SQC_ICACHE_REQ", 0), c.get("SQC_ICACHE_HITS", 0), c.get("SQC_ICACHE_MISSES", 0)
        line = f"{k:60s} n={n[k]:5d} icache req/launch {req / max(n[k], 1):10.0f}  miss/launch {mis / max(n[k], 1):9.0f}  miss rate {100 * mis / max(req, 1):5.1f} %  wait-inst share of wave cycles {100 * c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):5.1f} %"
        print(line); o.write(line + "\n")
P
find gpurun_out/pmc_${TAG}_ICACHE -name "*.csv" -delete
