"""Fold one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE; --kernel-trace only)
into per-kernel MFMA-pipe utilisation (BASELINE north_star: "rocprof ... MFMA utilisation against chip peak").
  cycles            = GRBM_GUI_ACTIVE / 8                     (rocprofv3 sums the 8 XCDs: /opt/skills/guides/MI355X_MICROARCH.md "DVFS give-back")
  mfma_busy_frac    = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 256 CUs * 4 SIMDs)      share of the chip's MFMA-pipe cycles that held an MFMA (the round-1 definition)
  mfma_busy_of_cu   = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)            the same over the cycles in which the CU had a wave at all
  flops             = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512                             (the counter advances by 1 per 512 fp16 MFMA FLOPs: 16x16x32 = 16 384 FLOP = 32 counts)
  clock_ghz         = cycles / rocprofv3 kernel duration (reads high below ~0.3 ms)
usage: python tools/pmc_mfma.py gpurun_out/pmc_r06a_MFMA profiles/r06a_pmc_mfma.json
"""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
NAMES = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE")


def short(name):
    import re
    m = re.match(r"_Z(\d+)", name)
    if m and not name.startswith("_ZN"):
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
        if t:
            args = [("true" if v == "1" else "false") if k == "b" else v for k, v in re.findall(r"L([ib])(\d+)E", t.group(1))]
            return f"{base}<{', '.join(args)}>"
        return base
    m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


d, out_path = sys.argv[1], sys.argv[2]
f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if k.startswith("void at::") or "rocclr" in k:
        continue
    a = agg[short(k)][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
    did = r.get("Dispatch_Id")
    if did not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
        seen.add(did)
        dur[short(k)][0] += 1
        dur[short(k)][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, c in agg.items():
    if not all(n in c for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")):
        continue
    avg = {n: c[n][1] / c[n][0] for n in c}
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    us = dur[k][1] / dur[k][0] / 1e3 if dur[k][0] else None
    e = {"launches_sampled": c["SQ_VALU_MFMA_BUSY_CYCLES"][0], "cycles_per_launch": cyc, "mfma_busy_frac": avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4) if cyc else None,
         "mfma_busy_of_cu_busy": avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * avg["SQ_BUSY_CU_CYCLES"]) if avg.get("SQ_BUSY_CU_CYCLES") else None,
         "mfma_flops_per_launch": avg.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512, "avg_us_profiled": us, "clock_ghz": (cyc / (us * 1e3)) if us else None,
         "wait_any_share_of_wave_cycles": (avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]) if avg.get("SQ_WAVE_CYCLES") else None}
    if us:
        e["tflops_profiled"] = e["mfma_flops_per_launch"] / (us * 1e-6) / 1e12
    out[k] = e
json.dump({"method": "rocprofv3 --kernel-trace --pmc " + " ".join(NAMES) + " (one pass, its own run), bench.py --steps 3 --warmup 1 on the committed plan table; "
                     "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)", "kernels": out}, open(out_path, "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["mfma_flops_per_launch"] * kv[1]["launches_sampled"])):
    if v["mfma_flops_per_launch"] > 0:
        print(f"{k:52s} n={v['launches_sampled']:5d} {v['avg_us_profiled'] or 0:8.1f} us  MFMA busy {100 * (v['mfma_busy_frac'] or 0):5.1f} % of chip, {100 * (v['mfma_busy_of_cu_busy'] or 0):5.1f} % of busy-CU cycles, "
              f"{v.get('tflops_profiled', 0):7.1f} TFLOP/s, clock {v['clock_ghz'] or 0:4.2f} GHz, waves waiting {100 * (v['wait_any_share_of_wave_cycles'] or 0):4.1f} %")
