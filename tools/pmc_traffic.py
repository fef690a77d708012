"""Fold rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, --kernel-trace only) into per-kernel
HBM-side traffic per launch, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md §HBM:
  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     (FETCH_SIZE reports exactly half of wide coalesced reads; unit KiB)
usage: python tools/pmc_traffic.py gpurun_out/pmc_r1_FETCH_SIZE gpurun_out/pmc_r1_WRITE_SIZE profiles/r01_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys


def fold(d, cname):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname and not r["Kernel_Name"].startswith("void at::") and "rocclr" not in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def short(name):
    """kernel name without the argument list; Itanium-mangled template kernels (rocprofv3 does not demangle signatures that contain
    _Float16) are rewritten to the form the executor's own tables use: gemm_f16_kernel<128, 128, 2, false, 2, 64, 0>"""
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


fetch, write = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if k in write:
        fk, wk = fetch[k][1] / fetch[k][0], write[k][1] / write[k][0]
        out[short(k)] = {"launches_sampled": fetch[k][0], "fetch_size_kib_raw": fk, "write_size_kib": wk,
                         "traffic_bytes_per_launch": (2 * fk + wk) * 1024}
json.dump({"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1; "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)", "kernels": out},
          open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"]):
    print(f"{k:45s} {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB/launch")
