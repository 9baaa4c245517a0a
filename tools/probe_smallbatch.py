"""Dev probe: where does a 2-row FlashSR forward spend its time (kernel time sum vs wall)?"""
import sys, time; sys.path.insert(0,'.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg=A.FlashSRConfig(); e=PyDriverEngine(cfg,A.init_params(cfg,0))
R=2
x=0.2*torch.randn(R,cfg.chunk,device='cuda'); nz=e.noise(R,None,0)
e.forward_rows(x,nz); torch.cuda.synchronize()
t=time.perf_counter(); e.forward_rows(x,nz); t_issue=time.perf_counter()-t; torch.cuda.synchronize(); t_all=time.perf_counter()-t
print(f"rows={R}: host issue {t_issue*1e3:.1f} ms, wall {t_all*1e3:.1f} ms")
def tm(f,*a):
    torch.cuda.synchronize(); t=time.perf_counter(); r=f(*a); ti=time.perf_counter()-t; torch.cuda.synchronize(); return r,(time.perf_counter()-t)*1e3, ti*1e3
mel,t1,i1=tm(e.log_mel,x); z,t2,i2=tm(e.vae_encode,mel); v,t3,i3=tm(e.unet,e.concat(nz,z)); z0=e.eltwise(nz,v,1,e.alpha,-e.sigma)
mh,t4,i4=tm(e.vae_decode,z0); y,t5,i5=tm(e.vocoder,mh,x)
print(f"stages wall/issue ms: mel {t1:.1f}/{i1:.1f} enc {t2:.1f}/{i2:.1f} unet {t3:.1f}/{i3:.1f} dec {t4:.1f}/{i4:.1f} voc {t5:.1f}/{i5:.1f}")
