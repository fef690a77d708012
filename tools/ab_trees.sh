#!/bin/bash
# Same-box A/B of whole TREES (boxes of the pool differ by up to 9 % on identical code: only runs interleaved on ONE box compare rounds).
#   here:   mkdir -p .abtrees/r04 && git archive <commit> instructany2pix_amd bench.py oracle include __graft_entry__.py | tar x -C .abtrees/r04
#           (cd .abtrees/r04 && python -m instructany2pix_amd.build)          (.abtrees/ is git-ignored; it travels with the gpurun snapshot)
#   there:  bash tools/ab_trees.sh "<label>=<dir> ..." "<shape>;<shape>..." rounds     e.g.  bash tools/ab_trees.sh "r04=.abtrees/r04 r05=." "cfg3=" 2
# A shape is "name=bench flags"; every (shape, round, tree) is its own process: fresh plans measured on this box by each tree's own tuner.
TREES=${1:-"r04=.abtrees/r04 r05=."}
SHAPES=${2:-"cfg3="}
ROUNDS=${3:-2}
cd $GRAFT_REPO_ROOT
IFS=';' read -ra SH <<< "$SHAPES"
for sh in "${SH[@]}"; do
  name=${sh%%=*}; flags=${sh#*=}
  for i in $(seq 1 $ROUNDS); do
    for t in $TREES; do
      label=${t%%=*}; dir=${t#*=}
      (cd $dir && python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline $flags 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['config']; hoisted=c.get('ms_per_step_with_context_kv_hoisted'); every=c.get('ms_per_step_with_context_kv_every_step'); runs=d['timing']['runs_ms_per_step']; per_req=c.get('context_kv','').startswith('projected once'); print('$name', '$label', 'round $i', 'context K/V every step:', ('%.3f' % every) if every else ['%.3f' % x for x in runs], '| once per request:', ['%.3f' % x for x in runs] if per_req else (('%.3f' % hoisted) if hoisted else '-'), '| probe', '%.0f' % d.get('box_probe', {}).get('gemm_4096_tflops', 0))")
    done
  done
done
