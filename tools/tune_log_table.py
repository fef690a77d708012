#!/usr/bin/env python3
"""Summarise an IA2P_TUNE_LOG=1 stderr capture: per contraction site the winner's TFLOP/s and the fastest candidates. usage: tune_log_table.py LOG [TOP]"""
import re, sys, collections
T = [(128,128,2),(128,128,3),(128,64,2),(128,64,3),(64,64,2),(64,64,3),(64,160,2),(64,160,3),(128,160,2),(128,160,3),(160,128,2),(160,160,2),(256,128,3),(64,64,4),(64,64,6),(128,64,4),(128,80,2),(128,80,4),(256,160,3),(128,160,3),(32,64,3),(32,128,3)]
PP = (12, 18, 19)
d = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r'\[ia2p tune\] (\d+) (\d+) (\d+) conv=(\d) geglu=(\d) variant=(\d+) splitk=(\d+) us=([\d.]+)', l)
    if m:
        M, N, K, c, g, v, sk, us = m.groups()
        d.setdefault((int(M), int(N), int(K), int(c), int(g)), []).append((float(us), int(v), int(sk)))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for (M, N, K, c, g), v in d.items():
    v.sort()
    s = '  '.join(f"{T[x[1]][0]}x{T[x[1]][1]}s{T[x[1]][2]}{'pp' if x[1] in PP else ''}{'/k%d' % x[2] if x[2] > 1 else ''}:{x[0]:.1f}" for x in v[:top])
    print(f"{M:6d} {N:6d} {K:6d} {'conv' if c else 'lin'}{' G' if g else '  '} {2.0 * M * N * K / v[0][0] / 1e6:5.0f} TF | {s}")
