#!/bin/bash
# does the tuner's candidate filter (modelled cost within 1.7 x of the modelled best) hide faster plans? Experiment build, slack 4: every shape's measured candidates, best first
cd $GRAFT_REPO_ROOT
IA2P_EXTRA_FLAGS="-DIA2P_EXPERIMENTS" python3 -m instructany2pix_amd.build > /dev/null 2>&1
export IA2P_EXTRA_FLAGS="-DIA2P_EXPERIMENTS"
IA2P_TUNE_SLACK=${1:-4.0} IA2P_TUNE_LOG=1 python3 bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe 2> gpurun_out/tune_slack.err > /dev/null
unset IA2P_EXTRA_FLAGS
python3 - <<'PY'
import re,collections
rows=collections.defaultdict(list)
for l in open("gpurun_out/tune_slack.err"):
    m=re.search(r"tune\] (\d+) (\d+) (\d+) conv=(\d) geglu=(\d) variant=(\d+) splitk=(\d+) us=([\d.]+)", l)
    if m: rows[tuple(map(int,m.groups()[:5]))].append((float(m.group(8)), int(m.group(6)), int(m.group(7))))
for k,v in rows.items():
    v.sort()
    print(k, " | ".join("v%d/sk%d %.1f" % (b,c,a) for a,b,c in v[:6]), "| n=%d" % len(v))
PY
python3 -m instructany2pix_amd.build > /dev/null 2>&1
