#!/bin/bash
# batch 1 (cfg 2) regressed +0.14 ms from round 3 to round 4 on the same box (profiles/r05_same_box_r03_r04_r05.txt): which of round 4's structures costs it?
# A/B builds / environment knobs of the current tree at batch 1, then the winner's effect at batch 8
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', ['%.3f' % x for x in d['timing']['runs_ms_per_step']])"; }
for i in 1 2; do
  for fl in "" "-DIA2P_LIN_BUF=0" "-DIA2P_REG_EPI_MIN=100000000" "-DIA2P_GN_TWOPASS"; do
    IA2P_EXTRA_FLAGS="$fl" python3 -m instructany2pix_amd.build > /dev/null 2>&1
    export IA2P_EXTRA_FLAGS="$fl"
    run "B1 [$fl]" "--batch 1 --ctx 77"
    if [ "$i" = "1" ] && [ -z "$fl" ]; then IA2P_TUNE_EXCLUDE=24,25,26 run "B1 [no halo tiles]" "--batch 1 --ctx 77"; IA2P_PREFETCH=0 run "B1 [no weight prefetch]" "--batch 1 --ctx 77"; IA2P_WT=0 run "B1 [no write-through]" "--batch 1 --ctx 77"; fi
    unset IA2P_EXTRA_FLAGS
  done
done
python3 -m instructany2pix_amd.build > /dev/null 2>&1
