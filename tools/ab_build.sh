#!/bin/bash
# same-box A/B of compile-time knobs: rebuilds the library on the GPU box per variant. usage: [BENCH_ARGS="--batch 1 --ctx 77"] bash tools/ab_build.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for i in 1 2; do for fl in "$@"; do
  IA2P_EXTRA_FLAGS="$fl" python3 -m instructany2pix_amd.build > /dev/null 2>&1
  IA2P_EXTRA_FLAGS="$fl" python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe $BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B8 [$fl]', d['timing']['runs_ms_per_step'])"
done; done
