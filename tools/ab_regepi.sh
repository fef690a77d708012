cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', ['%.3f' % x for x in d['timing']['runs_ms_per_step']])"; }
for i in 1 2; do
  for fl in "" "-DIA2P_REG_EPI_MIN=2049" "-DIA2P_REG_EPI_MIN=4097" "-DIA2P_REG_EPI_MIN=8193"; do
    IA2P_EXTRA_FLAGS="$fl" python3 -m instructany2pix_amd.build > /dev/null 2>&1
    export IA2P_EXTRA_FLAGS="$fl"
    run "B1 [$fl]" "--batch 1 --ctx 77"
    run "B8 [$fl]" ""
    unset IA2P_EXTRA_FLAGS
  done
done
python3 -m instructany2pix_amd.build > /dev/null 2>&1
