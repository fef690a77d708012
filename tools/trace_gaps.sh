#!/bin/bash
# kernel-trace of a few bench steps: sum of kernel durations vs wall span and the distribution of gaps between consecutive kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_gaps -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --plans $R/profiles/r01i_plans_cfg3.txt > $R/gpurun_out/trace_gaps.log 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/trace_gaps/**/*kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# last 3 steps: take the last 3*1100 kernels roughly: find ddim_step kernels as step delimiters
idx=[i for i,r in enumerate(rows) if "ddim_step" in r[2]]
a,b=idx[-4]+1, idx[-1]+1
seg=rows[a:b]
span=seg[-1][1]-seg[0][0]; busy=sum(e-s for s,e,_ in seg)
gaps=[seg[i+1][0]-seg[i][1] for i in range(len(seg)-1)]
pos=[g for g in gaps if g>0]; neg=[g for g in gaps if g<=0]
print(f"3 steps: {len(seg)} kernels, span {span/3e6:.3f} ms/step, sum of durations {busy/3e6:.3f} ms/step, positive gaps {sum(pos)/3e6:.3f} ms/step ({len(pos)} of {len(gaps)}), overlaps {sum(neg)/3e6:.3f} ms/step")
import statistics
print("gap percentiles (us):", [round(sorted(gaps)[int(len(gaps)*q)]/1e3,2) for q in (0.05,0.25,0.5,0.75,0.95)])
PY
find $R/gpurun_out/trace_gaps -name "*kernel_trace.csv" -delete
