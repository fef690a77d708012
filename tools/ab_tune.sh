#!/bin/bash
# same-box A/B of two TUNER settings (e.g. tile variants excluded): each setting tunes its own plan table, then the step is timed interleaved on the two tables.
# usage (GPU box): bash tools/ab_tune.sh NAME VALUE_A VALUE_B [extra bench flags]        e.g. bash tools/ab_tune.sh IA2P_TUNE_EXCLUDE 24,25 99
N=$1; A=$2; B=$3; shift 3
F="--no-roofline --no-cpu-baseline --no-secondary --no-box-probe --steps 50 --repeats 3 $*"
env $N=$A python bench.py $F --save-plans gpurun_out/pA.txt > /dev/null 2>&1
env $N=$B python bench.py $F --save-plans gpurun_out/pB.txt > /dev/null 2>&1
for i in 1 2 3; do
for v in A B; do
python bench.py $F --plans gpurun_out/p$v.txt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans $v', round(d['ms_per_step'],3), [round(x,3) for x in d['timing']['runs_ms_per_step']])"
done; done
