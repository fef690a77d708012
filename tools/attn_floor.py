"""Floor of one attention launch: Nq = 256 queries x 20 heads x 8 requests against 4 .. 256 keys in one segment (warm, back to back)."""
import ctypes as C, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
L=_ffi.lib(); s=_ffi.current_stream()
def t(fn,reps=200):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
B,h,N=8,20,256; Cc=h*64
q=torch.randn(B,N,Cc,device='cuda').half(); out=torch.empty_like(q)
for nk in (4,64,81,128,256):
    kv=torch.randn(B*nk,2*Cc,device='cuda').half()
    us=t(lambda: L.ia2p_attention(s,_ffi.ptr(q),Cc,_ffi.ptr(out),Cc,B,h,N,1,_ffi.ptr(kv),C.c_void_p(kv.data_ptr()+2*Cc),2*Cc,nk,1.0,None,None,0,0,0.0))
    print(f"Nq=256 nkeys={nk}: {us:.2f} us")
