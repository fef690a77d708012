cd $GRAFT_REPO_ROOT
IA2P_EXTRA_FLAGS="-DIA2P_EXPERIMENTS" python3 -m instructany2pix_amd.build > /dev/null 2>&1
export IA2P_EXTRA_FLAGS="-DIA2P_EXPERIMENTS"
for i in 1 2 3; do for sl in 1.7 2.5 4.0; do
  IA2P_TUNE_SLACK=$sl python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('slack $sl', ['%.3f' % x for x in d['timing']['runs_ms_per_step']], d['config']['kernel_plans'])"
done; done
unset IA2P_EXTRA_FLAGS
python3 -m instructany2pix_amd.build > /dev/null 2>&1
