#!/bin/bash
# Same-box A/B of whole TREES (boxes of the pool differ by up to 9 % on identical code: only runs interleaved on ONE box compare rounds).
#   here:   mkdir -p .abtrees/r04 && git archive <commit> instructany2pix_amd bench.py oracle include __graft_entry__.py | tar x -C .abtrees/r04
#           (cd .abtrees/r04 && python -m instructany2pix_amd.build)          (.abtrees/ is git-ignored; it travels with the gpurun snapshot)
#   there:  bash tools/ab_trees.sh "<label>=<dir> ..." "<shape>;<shape>..." rounds     e.g.  bash tools/ab_trees.sh "r04=.abtrees/r04 r05=." "cfg3=" 2
# A shape is "name=bench flags"; every (shape, round, tree) is its own process. Trees up to round 5 measure their plans on the box (their default); the round-6 tree runs
# its committed plan table unless the flags say --tune (pass it for a like-for-like comparison).
TREES=${1:-"r04=.abtrees/r04 r05=."}
SHAPES=${2:-"cfg3="}
ROUNDS=${3:-2}
cd $GRAFT_REPO_ROOT
IFS=';' read -ra SH <<< "$SHAPES"
for sh in "${SH[@]}"; do
  name=${sh%%=*}; flags=${sh#*=}
  for i in $(seq 1 $ROUNDS); do
    for t in $TREES; do
      label=${t%%=*}; dir=${t#*=}; extra=""
      if [[ "$dir" == *:* ]]; then extra=${dir#*:}; dir=${dir%%:*}; fi      # "label=dir:--flag": a flag only this tree's bench.py knows (e.g. r06tune=.:--tune)
      (cd $dir && python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline $flags $extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']; runs = d['timing']['runs_ms_per_step']
# which schedule the tree's headline loop runs, and where it reports the other one (rounds 1-4: every step + _hoisted; round 5: once per request + _every_step; round 6: every step + _once_per_request)
per_req_headline = c.get('context_kv', '').startswith('projected once')
every = runs if not per_req_headline else c.get('ms_per_step_with_context_kv_every_step')
once = runs if per_req_headline else (c.get('ms_per_step_context_kv_once_per_request') or c.get('ms_per_step_with_context_kv_hoisted'))
fmt = lambda v: ['%.3f' % x for x in v] if isinstance(v, list) else ('%.3f' % v if v else '-')
probe = (c.get('box_probe') or d.get('box_probe') or {}).get('gemm_4096_tflops', 0)
print('$name', '$label', 'round $i', 'context K/V every step:', fmt(every), '| once per request:', fmt(once), '| probe %.0f' % probe, '| plans:', c.get('kernel_plans', '')[:40])
")
    done
  done
done
