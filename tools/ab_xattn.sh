cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "qproj or cross_attention or self_attention or tile_choice" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q 2>&1 | tail -5
python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --save-plans gpurun_out/ab_plans.txt > gpurun_out/ab_fuse1.json 2> gpurun_out/ab_fuse1.err
for i in 1 2; do
IA2P_XATTN_FUSE=0 python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --plans gpurun_out/ab_plans.txt 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=0', d['ms_per_step'], d['timing'])"
IA2P_XATTN_FUSE=1 python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline --plans gpurun_out/ab_plans.txt 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=1', d['ms_per_step'], d['timing'])"
done
for f in 0 1 2; do
IA2P_XATTN_MIN_TILES=$((f==2?1:256)) IA2P_XATTN_FUSE=$((f>0)) python3 bench.py --batch 1 --ctx 77 --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B1 fuse=$f', d['ms_per_step'], d['timing'])"
done
