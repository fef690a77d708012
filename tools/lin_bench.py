import sys, torch
sys.path.insert(0,'.')
from instructany2pix_amd import _ffi
L=_ffi.lib(); s=_ffi.current_stream()
for (M,N,K) in [(8,1280,320),(8,1280,1280),(8,1280,2816),(8,21760,1280),(1,1280,1280),(16,1280,1280)]:
    X=torch.randn(M,K,device='cuda').half(); W=(torch.randn(N,K,device='cuda')*K**-0.5).half(); b=torch.randn(N,device='cuda').half()
    out=torch.empty(M,N,device='cuda',dtype=torch.half)
    f=lambda: _ffi.check(L.ia2p_linear_small(s,_ffi.ptr(X),_ffi.ptr(W),_ffi.ptr(b),_ffi.ptr(out),M,N,K,0,0))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/50*1e3
    print(M,N,K,f"{us:.1f} us  {N*K*2/us/1e3:.0f} GB/s weights")
