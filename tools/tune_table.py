"""Format the in-place autotuner log (IA2P_TUNE_LOG=1 python bench.py ... 2> log) as one line per contraction site:
M N K kind GFLOP [TFLOP/s of the winner] candidates best first (tile, K-split, us). usage: python tools/tune_table.py log [top]"""
import collections
import re
import sys

TILES = [(128, 128, 2), (128, 128, 3), (128, 64, 2), (128, 64, 3), (64, 64, 2), (64, 64, 3), (64, 160, 2), (64, 160, 3), (128, 160, 2), (128, 160, 3),
         (160, 128, 2), (160, 160, 2), (256, 128, 3), (64, 64, 4), (64, 64, 6), (128, 64, 4), (128, 80, 2), (128, 80, 4)]
NAMES = ["%dx%ds%d" % t for t in TILES]
NAMES[12] += "pp"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
best = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.search(r"\[ia2p tune\] (\d+) (\d+) (\d+) conv=(\d) geglu=(\d) variant=(\d+) splitk=(\d+) us=([\d.]+)", line)
    if m:
        M, N, K, cv, gg, v, sk, us = m.groups()
        best[(int(M), int(N), int(K), int(cv), int(gg))][(int(v), int(sk))] = float(us)
tot = 0.0
for (M, N, K, cv, gg), c in sorted(best.items()):
    fl = 2.0 * M * N * K
    r = sorted(c.items(), key=lambda kv: kv[1])
    print(f"{M:6d} {N:6d} {K:6d} {'conv' if cv else 'lin '} {'geglu' if gg else '     '} {fl / 1e9:8.1f} GFLOP [{fl / r[0][1] / 1e6:6.0f} TFLOP/s]  " +
          "  ".join(f"{NAMES[v] if v < len(NAMES) else v}{'/k%d' % sk if sk > 1 else ''}:{us:.1f}" for (v, sk), us in r[:top]))
