F="--no-roofline --no-cpu-baseline --no-secondary --no-box-probe --steps 50 --repeats 3"
python bench.py $F --save-plans gpurun_out/p.txt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warm', d['ms_per_step'])"
for i in 1 2; do
for v in 0 1; do
IA2P_EMBED_OVERLAP=$v python bench.py $F --plans gpurun_out/p.txt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$v', d['ms_per_step'], d['timing'] if 'timing' in d else '')"
done; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests/test_unet_gpu.py -x -q -m gpu 2>&1 | tail -3
