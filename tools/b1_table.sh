#!/bin/bash
# batch-1 (BASELINE configs[1]) kernel table + in-situ tuner table
IA2P_TUNE_LOG=1 python bench.py --batch 1 --ctx 77 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/$1_b1.json 2> gpurun_out/$1_b1.err
grep "launches/step" gpurun_out/$1_b1.err | head -24
python tools/tune_table.py gpurun_out/$1_b1.err 5 | cut -c1-200
python -c "import json;d=json.load(open('gpurun_out/$1_b1.json'));print('B=1 ms/step', d['ms_per_step'], d['timing']['runs_ms_per_step'])"
