"""Dev micro-benchmark: the 16 z-batched Winograd GEMMs at the three channel levels (26 rows)."""
import sys; sys.path.insert(0,'.')
import ctypes as C, torch
from packload import load_pack; load_pack()
from egregora_amd import native
L=native.lib(); p=lambda t: C.c_void_p(t.data_ptr())
for name,(P,Ci,Co) in {"lvl1_256":(26*128*64,256,256),"lvl2_512":(26*64*32,512,512),"lvl3_1024":(26*32*16,1024,1024)}.items():
    V=torch.randn(16,P,Ci,device='cuda'); U=torch.randn(16,Ci//16,Co,16,device='cuda')/Ci**0.5; M=torch.empty(16,P,Co,device='cuda')
    run=lambda: native.check(L.egr_gemm_zbatched(p(V),p(U),p(M),16,P,Ci,Co,P*Ci,U[0].numel(),P*Co,native.stream_ptr()),"g")
    run(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/5; fl=16*2.0*P*Ci*Co
    print(f"{name:10s} {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s")
