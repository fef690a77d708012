"""In-situ kernel durations and inter-kernel gaps of the denoise step from a raw rocprofv3 kernel trace (tools/insitu_trace.sh).
Steps are delimited by ddim_step_kernel; the last STEPS whole steps are averaged position by position, then grouped by (kernel, grid size).
usage: python tools/insitu_trace.py gpurun_out/<tag>_kernel_trace.csv [steps=8] [--sites]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"])))
rows.sort()
ends = [i for i, r in enumerate(rows) if "ddim_step_kernel" in r[2]]
assert len(ends) > nsteps, f"only {len(ends)} steps in the trace"
steps = [rows[ends[k] + 1: ends[k + 1] + 1] for k in range(len(ends) - nsteps - 1, len(ends) - 1)]
n = len(steps[0])
assert all(len(s) == n for s in steps), [len(s) for s in steps]


def short(name):
    m = re.search(r"gemm_f16_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)      # mangled (rocprofv3 raw trace)
    if m:
        bm, bn, st, conv, wgm, bk, pp, wgn = m.groups()
        return f"gemm<{bm},{bn},{st},{'conv' if conv == '1' else 'lin'},{wgm}x{wgn},pp{pp}>"
    m = re.search(r"gemm_f16_kernel<(\d+), (\d+), (\d+), (true|false), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        bm, bn, st, conv, wgm, bk, pp, wgn = m.groups()
        return f"gemm<{bm},{bn},{st},{'conv' if conv == 'true' else 'lin'},{wgm}x{wgn},pp{pp}>"
    m = re.search(r"_Z\d+(\w+?_kernel)(ILi(\d+)E)?", name)
    if m:
        return m.group(1) + (f"<{m.group(3)}>" if m.group(3) else "")
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


pos = []
for i in range(n):
    dur = sum(s[i][1] - s[i][0] for s in steps) / len(steps) / 1e3
    gap = sum((s[i + 1][0] - s[i][1]) if i + 1 < n else 0 for s in steps) / len(steps) / 1e3
    pos.append((short(steps[0][i][2]), steps[0][i][3], dur, gap))
step_us = sum((s[-1][1] - s[0][0]) for s in steps) / len(steps) / 1e3
busy = sum(p[2] for p in pos)
gaps = sum(p[3] for p in pos)
print(f"# {path}: {nsteps} steps, {n} launches per step, step span {step_us / 1e3:.3f} ms = kernels {busy / 1e3:.3f} + gaps {gaps / 1e3:.3f} ms ({gaps / n:.2f} us per boundary)")
agg = defaultdict(lambda: [0, 0.0, 0.0])
for name, grid, dur, gap in pos:
    a = agg[(name, grid)]
    a[0] += 1; a[1] += dur; a[2] += gap
print(f"{'kernel':58s} {'grid':>6s} {'n/step':>6s} {'us each':>8s} {'gap after':>9s} {'ms/step':>8s}")
for (name, grid), (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:58]:58s} {grid:6d} {c:6d} {d / c:8.2f} {g / c:9.2f} {d / 1e3:8.3f}")
if "--sites" in sys.argv:
    print("# position by position")
    for i, (name, grid, dur, gap) in enumerate(pos):
        print(f"{i:4d} {name[:58]:58s} {grid:6d} {dur:8.2f} {gap:6.2f}")
