#!/bin/bash
# kernel-trace of a few bench steps with the context projection on the side stream: where do the first ~20 launches of a step start and end?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-box-probe --save-plans $R/gpurun_out/p.txt > /dev/null 2>&1
for v in 0 1; do
export IA2P_KV_OVERLAP=$v
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_ov$v -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-box-probe --plans $R/gpurun_out/p.txt > $R/gpurun_out/trace_ov$v.log 2>&1
V=$v python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]; v=os.environ["V"]
f=glob.glob(R+f"/gpurun_out/trace_ov{v}/**/*kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
rows.sort()
idx=[i for i,r in enumerate(rows) if "ddim_step" in r[2]]
a,b=idx[4]+1, idx[5]+1      # (a step of the first timed run: the last steps of the process are the hoisted-context secondary)
seg=rows[a:b]
t0=seg[0][0]
print(f"== IA2P_KV_OVERLAP={v}: step of {len(seg)} kernels, span {(seg[-1][1]-t0)/1e6:.3f} ms, sum of durations {sum(e-s for s,e,_,_ in seg)/1e6:.3f} ms")
for i,(s,e,n,q) in enumerate(seg):
    if i < 12 or e-s > 150000 or q != seg[0][3]:
        print(f"  #{i:3d} q{q} start {(s-t0)/1e3:8.1f} us  dur {(e-s)/1e3:7.1f} us  {n[:70]}")
print("  queues:", sorted(set(r[3] for r in seg)))
PY
find $R/gpurun_out/trace_ov$v -name "*kernel_trace.csv" -delete
done
