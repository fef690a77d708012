"""Rewrite the Name column of a rocprofv3 kernel_stats.csv with readable kernel names: rocprofv3 leaves signatures that contain _Float16
mangled (`_Z15gemm_f16_kernelILi128E...`). usage: python tools/demangle_stats.py in.csv out.csv"""
import csv
import re
import sys


def short(name):
    m = re.match(r"_Z(\d+)", name)
    if m and not name.startswith("_ZN"):
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
        if t:
            args = [("true" if v == "1" else "false") if k == "b" else v for k, v in re.findall(r"L([ib])(\d+)E", t.group(1))]
            return f"{base}<{', '.join(args)}>"
        return base
    return name


rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w", newline=""), quoting=csv.QUOTE_NONNUMERIC)
w.writerow(rows[0])
for r in rows[1:]:
    w.writerow([short(r[0])] + [float(x) if re.fullmatch(r"[-+0-9.eE]+", x) else x for x in r[1:]])
