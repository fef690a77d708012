#!/bin/bash
# same-box A/B of compile-time knobs with the per-region / per-kernel-class split: usage: bash tools/ab_build_regions.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for i in 1 2; do for fl in "$@"; do
  IA2P_EXTRA_FLAGS="$fl" python3 -m instructany2pix_amd.build > /dev/null 2>&1
  IA2P_EXTRA_FLAGS="$fl" python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-box-probe --kernel-table gpurun_out/abr_kt.json 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
kt=json.load(open('gpurun_out/abr_kt.json')); n=kt['steps_profiled']
conv=sum(v['ms'] for k,v in kt['kernels'].items() if 'true' in k)/n; gn=sum(v['ms'] for k,v in kt['kernels'].items() if 'gn_' in k)/n
print('B8 [$fl] step', [round(x,3) for x in d['timing']['runs_ms_per_step']], 'conv region %.3f ms, 3x3 conv kernels %.3f ms, GroupNorm %.3f ms (profiled pass)' % (r['conv_blocks']['ms'], conv, gn))"
done; done
