#!/bin/bash
# same-box A/B of compile-time knobs of the 256 x 320 GEGLU tile on the stand-alone clock tool: bash tools/micro/geglu_ab.sh "" "-DIA2P_G320_SETPRIO"     (on the GPU box)
cd $GRAFT_REPO_ROOT
i=0
for fl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIA2P_CLOCK_STAMP $fl -mllvm -amdgpu-kernarg-preload-count=16 tools/micro/geglu_clock.hip -o /tmp/geglu_clock_$i || exit 1
  i=$((i+1))
done
for r in 1 2 3; do i=0; for fl in "$@"; do echo "[$fl] $(/tmp/geglu_clock_$i | grep -v group | head -2 | cut -c1-150 | tr '\n' '|')"; i=$((i+1)); done; done
