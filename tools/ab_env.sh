#!/bin/bash
# A/B of a process environment setting on the headline step, same box, interleaved: bash tools/ab_env.sh NAME VALUE_A VALUE_B [extra bench flags]
N=$1; A=$2; B=$3; shift 3
F="--no-roofline --no-cpu-baseline --no-secondary --no-box-probe --steps 50 --repeats 3 $*"
python bench.py $F --save-plans gpurun_out/p.txt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warm (unset)', round(d['ms_per_step'],3))"
for i in 1 2; do
for v in "$A" "$B"; do
env $N=$v python bench.py $F --plans gpurun_out/p.txt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v', round(d['ms_per_step'],3), [round(x,3) for x in d['timing']['runs_ms_per_step']])"
done; done
