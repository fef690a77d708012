#!/bin/bash
# where the GroupNorm-fused convolution's per-block overhead goes: A/B builds of the in-loop normalisation
#   IA2P_GN_ABL (timing only, wrong results) bit 0: no LDS fetch of the piece / table, bit 1: no VALU steps, bit 2: no LDS write-back
#   IA2P_GN_LEAD / IA2P_GN_EVERY: placement of the steps between the MFMAs (LEAD = 1000: all of them behind the last MFMA)
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  IA2P_EXTRA_FLAGS="$fl" python3 -m instructany2pix_amd.build > /dev/null 2>&1
  echo "== $fl"
  python3 tools/conv_gn_probe.py 24 25 2>/dev/null | grep -E "640\+320->320|8x16x16 1280\+0->1280 \(\+0\) sk2|8x32x32 640\+0|8x64x64 320\+0->320 \(\+0\)" | cut -c1-200
done
python3 -m instructany2pix_amd.build > /dev/null 2>&1
