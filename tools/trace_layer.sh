#!/bin/bash
# kernel-trace of a few bench steps: per-kernel durations of the level-2 transformer layers in launch order (median over the 60 layers of the last step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_layer -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/trace_layer.log 2>&1
python3 - <<'PY'
import csv, glob, os, statistics, re
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/trace_layer/**/*kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx=[i for i,r in enumerate(rows) if "ddim_step" in r[2]]
seg=rows[idx[-2]+1:idx[-1]+1]
def short(n):
    n=re.sub(r"\(.*","",n)
    m=re.search(r"gemm_f16_kernel<(\d+), ?(\d+), ?(\d+), ?(\w+)", n)
    if m: return f"gemm{m.group(1)}x{m.group(2)}x{m.group(3)}{'c' if m.group(4) in ('true','1') else ''}"
    for k in ("qproj_xattn","attention_f16","splitk_reduce","gn_stats","gn_apply","concat","conv_in","conv_out","linear_small","embed","ddim","fold_ln"):
        if k in n: return k
    return n[:24]
names=[short(n) for _,_,n in seg]
# find the repeating 9-kernel layer pattern: starts at a gemm followed by attention_f16
starts=[i for i in range(len(names)-9) if names[i+1]=="attention_f16" and names[i+3]=="qproj_xattn"]
print(len(seg),"kernels in the step;",len(starts),"layers found")
from collections import defaultdict
dur=defaultdict(list); gap=defaultdict(list); nm={}
for s in starts:
    if seg[s][0] and names[s].startswith("gemm"):
        pass
    for k in range(9):
        if s+k>=len(seg): break
        d=(seg[s+k][1]-seg[s+k][0])/1e3; g=(seg[s+k+1][0]-seg[s+k][1])/1e3 if s+k+1<len(seg) else 0
        key=(k,names[s+k]); dur[key].append(d); gap[key].append(g)
for key in sorted(dur):
    if len(dur[key])>=20:
        print(f"  pos {key[0]} {key[1]:18s} n={len(dur[key]):3d} median {statistics.median(dur[key]):7.2f} us  (p10 {sorted(dur[key])[len(dur[key])//10]:6.2f}, p90 {sorted(dur[key])[len(dur[key])*9//10]:6.2f})  gap after: {statistics.median(gap[key]):5.2f} us")
tot=sum(e-s for s,e,_ in seg)/1e3; span=(seg[-1][1]-seg[0][0])/1e3
print(f"step: sum of durations {tot/1e3:.3f} ms, span {span/1e3:.3f} ms")
PY
find $R/gpurun_out/trace_layer -name "*kernel_trace.csv" -delete
