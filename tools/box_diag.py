"""Diagnostics of a GPU box: CPU resources visible to the process and HIP runtime identity."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/memory.max"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print([l for l in open("/proc/cpuinfo") if "model name" in l][:1])
print(open("/proc/meminfo").read().split("\n")[0:3])
from instructany2pix_amd import _ffi
L = _ffi.lib()
print("is_gfx950 before torch init:", L.ia2p_device_is_gfx950())
import torch
print("torch threads", torch.get_num_threads())
torch.cuda.set_device(0)
print("is_gfx950 after set_device:", L.ia2p_device_is_gfx950())
x = torch.zeros(1, device="cuda")
print("is_gfx950 after alloc:", L.ia2p_device_is_gfx950())
print(open(f"/proc/{os.getpid()}/maps").read().count("libamdhip64"), "maps of libamdhip64:",
      sorted({l.split()[-1] for l in open(f"/proc/{os.getpid()}/maps") if "libamdhip64" in l}))
for nt in (None, 8, 16, 32):
    if nt:
        torch.set_num_threads(nt)
    a = torch.randn(2048, 1280); w = torch.randn(10240, 1280)
    t0 = time.time(); (a @ w.t()); t1 = time.time(); (a @ w.t()); t2 = time.time()
    print("threads", torch.get_num_threads(), "matmul 53.7 GFLOP: %.3fs %.3fs" % (t1 - t0, t2 - t1))
t0 = time.time(); torch.randn(100_000_000); print("randn 1e8: %.2fs" % (time.time() - t0))
