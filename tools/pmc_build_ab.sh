#!/bin/bash
# PMC traffic of an A/B BUILD (compile-time knob) next to the shipped one: bash tools/pmc_build_ab.sh <tag> "<flags>"      (inside gpurun)
TAG=$1; FL=$2
R=$GRAFT_REPO_ROOT
cd $R
IA2P_EXTRA_FLAGS="$FL" python3 -m instructany2pix_amd.build > /dev/null 2>&1
IA2P_EXTRA_FLAGS="$FL" python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-box-probe --save-plans $R/gpurun_out/${TAG}_plans.txt --kernel-table $R/gpurun_out/${TAG}_kernel_table.json 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$FL] ms/step', d['timing']['runs_ms_per_step'], 'conv region', d['roofline']['conv_blocks']['ms'])"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  IA2P_EXTRA_FLAGS="$FL" rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --no-box-probe --plans $R/gpurun_out/${TAG}_plans.txt > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/${TAG}_pmc.json | grep -E "true|conv" | head -12
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -delete
