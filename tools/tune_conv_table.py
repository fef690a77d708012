"""Tuner landscape of the conv sites from an IA2P_TUNE_LOG=1 run: python tools/tune_conv_table.py <stderr file> [conv|lin]"""
import collections
import re
import sys
T = [(128,128,2,0),(128,128,3,0),(128,64,2,0),(128,64,3,0),(64,64,2,0),(64,64,3,0),(64,160,2,0),(64,160,3,0),(128,160,2,0),(128,160,3,0),(160,128,2,0),(160,160,2,0),(256,128,3,1),
     (64,64,4,0),(64,64,6,0),(128,64,4,0),(128,80,2,0),(128,80,4,0),(256,160,3,1),(128,160,3,1),(32,64,3,0),(32,128,3,0),(256,256,2,2),(256,128,2,2)]
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"\[ia2p tune\] (\d+) (\d+) (\d+) conv=(\d) geglu=(\d) variant=(\d+) splitk=(\d+) us=([\d.]+)", l)
    if m:
        M, N, K, c, g, v, sk, us = m.groups()
        rows[(int(M), int(N), int(K), int(c), int(g))].append((float(us), int(v), int(sk)))
want = 1 if (len(sys.argv) < 3 or sys.argv[2] == "conv") else 0
for key in sorted(rows):
    if key[3] != want:
        continue
    r = sorted(rows[key])
    fl = 2.0 * key[0] * key[1] * key[2]
    nm = lambda v, sk: "%dx%ds%d%s%s" % (T[v][0], T[v][1], T[v][2], "p" * T[v][3], ("/k%d" % sk if sk > 1 else ""))
    pp = [x for x in r if T[x[1]][3] == 1][:1]
    print("%6d %6d %6d %s %5.0f TF | " % (key[0], key[1], key[2], "G" if key[4] else " ", fl / r[0][0] / 1e6) + "  ".join("%s:%.1f" % (nm(v, sk), us) for us, v, sk in r[:5]) +
          ("   || best ping-pong %s:%.1f" % (nm(pp[0][1], pp[0][2]), pp[0][0]) if pp else ""))
