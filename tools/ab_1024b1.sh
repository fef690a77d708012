cd $GRAFT_REPO_ROOT
VAR=$1; shift
for i in 1 2; do for v in "$@"; do
env $VAR=$v python3 bench.py --batch 1 --latent 128 --ctx 77 --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1024px B1 $VAR=$v', d['timing']['runs_ms_per_step'])"
done; done
