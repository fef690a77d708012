"""Build libia2p_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m instructany2pix_amd.build` or `__graft_entry__.build()`. hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDES = [os.path.join(os.path.dirname(HERE), "include", f) for f in ("ia2p.h", "ia2p_debug.h")]
OUT = os.path.join(HERE, "libia2p_hip.so")
SOURCES = ["gemm.hip", "qxattn.hip", "attention.hip", "norm.hip", "misc.hip", "engine.hip", "vae_engine.hip", "clip_engine.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"] + os.environ.get("IA2P_EXTRA_FLAGS", "").split()    # (A/B builds of compile-time knobs)
# attention: MFMA results feed VALU softmax directly; the VGPR form avoids ~250 v_accvgpr_read/write per key tile
# kernarg preload: the leading dwords of a kernel's (scalar) arguments arrive in SGPRs instead of through a cold read of the argument block -- FOURTEEN of them
# (16 user SGPRs less the argument block's address: `.amdhsa_user_sgpr_kernarg_preload_length 14` whatever count is asked for; the kernels' leading arguments are packed to fit)
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
FILE_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"] + PRELOAD, "qxattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"] + PRELOAD, "gemm.hip": PRELOAD, "norm.hip": PRELOAD}


def _stamp():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + INCLUDES
    for p in files:
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    h.update((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    stamp_file = os.path.join(CSRC, ".build_stamp")
    stamp = _stamp()
    if not force and os.path.exists(OUT) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return OUT
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    open(stamp_file, "w").write(stamp)
    if verbose:
        print(f"built {OUT} ({os.path.getsize(OUT) // 1024} KiB)")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
