"""Yardstick only (NOT used by the product): what the vendor GEMM library reaches on the same shapes, same random fp16 data."""
import torch
shapes = [(2048, 3840, 1280, "L2 qkv"), (2048, 1280, 1280, "L2 proj"), (2048, 10240, 1280, "L2 ff-in"), (2048, 1280, 5120, "L2 ff-out"),
          (8192, 1920, 640, "L1 qkv"), (8192, 640, 640, "L1 proj"), (8192, 5120, 640, "L1 ff-in"), (8192, 640, 2560, "L1 ff-out"),
          (616, 166400, 2048, "ctx kv"), (32768, 320, 2880, "conv320@64 as GEMM"), (2048, 1280, 11520, "conv1280@16 as GEMM"), (4096, 4096, 4096, "4096^3")]
for M, N, K, label in shapes:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    for _ in range(3): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{label:22s} {M}x{N}x{K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.0f} TFLOP/s")
