"""Dev probe: full-size FlashSR engine timing by rows per pass, with per-stage breakdown."""
import sys, time; sys.path.insert(0,'.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg=A.FlashSRConfig(); t0=time.time(); P=A.init_params(cfg,0); print("init params %.1fs"%(time.time()-t0))
e=PyDriverEngine(cfg,P); print("engine ready %.1fs"%(time.time()-t0))
fl=e.flop_count(1); print("flops/row %.3e"%fl)
def sync(): torch.cuda.synchronize()
for R in (1,2,4,8):
    x=0.2*torch.randn(R,cfg.chunk,device='cuda'); nz=e.noise(R,None,0)
    e.forward_rows(x,nz); sync()
    t=time.time(); n=2
    for _ in range(n): e.forward_rows(x,nz)
    sync(); dt=(time.time()-t)/n
    print(f"rows={R} {dt*1e3:.1f} ms  per-row {dt/R*1e3:.1f} ms  xRT/row {5.12*R/dt:.1f}  TF/s {fl*R/dt/1e12:.1f}")
# stage breakdown at R=4
R=4; x=0.2*torch.randn(R,cfg.chunk,device='cuda'); nz=e.noise(R,None,0)
def tm(f,*a):
    sync(); t=time.time(); r=f(*a); sync(); return r,(time.time()-t)*1e3
mel,t1=tm(e.log_mel,x); z,t2=tm(e.vae_encode,mel); v,t3=tm(e.unet,e.concat(nz,z)); z0=e.eltwise(nz,v,1,e.alpha,-e.sigma)
mh,t4=tm(e.vae_decode,z0); y,t5=tm(e.vocoder,mh,x)
print(f"R=4 stages ms: mel {t1:.1f} enc {t2:.1f} unet {t3:.1f} dec {t4:.1f} voc {t5:.1f}")
