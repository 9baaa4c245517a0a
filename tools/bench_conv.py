"""Dev micro-benchmark of egr_conv_nhwc on the dominant FlashSR shapes (for A/B of kernel variants and PMC runs)."""
import sys, time; sys.path.insert(0,'.')
import ctypes as C, torch
from packload import load_pack; load_pack()
from egregora_amd import native
L=native.lib()
SHAPES={ # name: (B,H,W,Cin,Cout,k)
 "s1_lat1024":(26,64,32,1024,1024,3),
 "s2_l0_128":(26,512,256,128,128,3),
 "s3_512":(26,128,64,512,512,3),
 "s4_256":(26,256,128,256,256,3),
 "s5_1x1_1024":(26,64,32,1024,1024,1),
 "s6_unet_small":(26,8,4,640,640,3),
}
names=sys.argv[1].split(",") if len(sys.argv)>1 else list(SHAPES)
reps=int(sys.argv[2]) if len(sys.argv)>2 else 5
p=lambda t: C.c_void_p(t.data_ptr())
for n in names:
    B,H,W,Ci,Co,k=SHAPES[n]
    x=torch.randn(B,H,W,Ci,device='cuda'); w=torch.randn((k*k*Ci+15)//16,Co,16,device='cuda')/ (Ci*k*k)**0.5; b=torch.randn(Co,device='cuda')
    y=torch.empty(B,H,W,Co,device='cuda')
    def run(): native.check(L.egr_conv_nhwc(p(x),p(w),p(b),C.c_void_p(0),C.c_void_p(0),p(y),B,H,W,Ci,H,W,Co,k,k,1,1,k//2,k//2,0,0,0.0,native.stream_ptr()),"conv")
    run(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps; fl=2.0*B*H*W*Co*k*k*Ci
    print(f"{n:14s} {ms:8.3f} ms  {fl/ms/1e9:7.1f} TF/s")
