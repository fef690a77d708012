#!/bin/bash
# per-kernel VGPR / spill / scratch / LDS table of one translation unit: bash tools/resource_usage.sh gemm.hip [extra hipcc flags]
SRC=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16"
case "$SRC" in attention.hip|qxattn.hip) FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form";; esac
/opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -c "$HERE/instructany2pix_amd/csrc/$SRC" -o /tmp/ru_$$.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l) or re.search(r' Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r'VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('spill',r'VGPR Spill: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('sgpr',r'SGPRs: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None and k not in cur: cur[k]=m.group(1)
for r in rows:
    try: name=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',r['name']],capture_output=True,text=True).stdout.strip()
    except Exception: name=r['name']
    name=re.sub(r'\(.*','',name)
    print(f\"{name:70s} vgpr {r.get('vgpr','?'):>3} agpr {r.get('agpr','?'):>3} sgpr {r.get('sgpr','?'):>3} spill {r.get('spill','?'):>3} scratch {r.get('scratch','?'):>4} occ {r.get('occ','?')}\")
"
rm -f /tmp/ru_$$.o
