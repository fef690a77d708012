#!/bin/bash
# same-box A/B of an OLDER COMMIT against the working tree (boxes differ by up to 9 %, so two profile rounds on two boxes say nothing):
#   here:  git worktree add /tmp/old <commit>; (cd /tmp/old && python -m instructany2pix_amd.build)
#          mkdir tools/old_tree; cp -r /tmp/old/{instructany2pix_amd,bench.py,oracle,include} tools/old_tree/     (not committed; travels with the snapshot)
#   then:  gpurun -- 'bash tools/ab_old.sh'
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
(cd tools/old_tree && python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('OLD', d['timing']['runs_ms_per_step'])")
python3 bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('NEW', d['timing']['runs_ms_per_step'])"
done
