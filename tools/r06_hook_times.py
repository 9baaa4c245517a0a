"""dev (round 6): is the cost of the variant hooks code or data?  Stage times (C3 shape, 800 iterations, node flags) of hook 1 (hard, run-time
level) and hook 3 (soft) at thresholds that gate nearly everything and nearly nothing."""
import os, sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
n=2880000
x=torch.from_numpy((0.25*np.random.default_rng(0).standard_normal((2,n))).astype(np.float32)).cuda()
f=dict(normalize=True,autoscale=False,pcm_in=True,node_post=True)
for var,thr in (("",0.6),("relative",0.6),("relative",0.001),("relative,soft",0.6),("relative,soft",0.001),("soft",50.0),("soft",20000.0),("",20000.0)):
    ts=[]
    for r in range(6):
        torch.cuda.synchronize(); t=time.perf_counter()
        fe.enhance_device(x,1,800,thr,variant=var,**f)
        torch.cuda.synchronize(); ts.append((time.perf_counter()-t)*1e3)
    print(f"{var or 'default':16s} thr {thr:9.3f}  median {sorted(ts[2:])[2]:.2f} ms", flush=True)
