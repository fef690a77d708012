#!/bin/bash
# same-box A/B of compile-time knobs with an arbitrary measurement command: bash tools/ab_build_cmd.sh "<cmd>" "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
CMD=$1; shift
for i in 1 2; do for fl in "$@"; do
  IA2P_EXTRA_FLAGS="$fl" python3 -m instructany2pix_amd.build > /dev/null 2>&1
  echo "== [$fl]"; IA2P_EXTRA_FLAGS="$fl" bash -c "$CMD"
done; done
