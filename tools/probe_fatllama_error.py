"""Dev probe: float32 round-off of the device loop vs the oracle (float32) vs the float64 run; node-level LSD."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
from oracle import fatllama as ofl, metrics as om
from test_gpu_fatllama import synth
for (C,n,f,it) in [(1,160000,6,5),(2,48000,1,20),(1,480000,1,10)]:
    x=synth(C,n,seed=n+f)
    want=ofl.enhance_channels(x,f,it,0.6,False,False)
    ex=ofl.enhance_channels(x,f,it,0.6,False,False,exact=True)
    got=fe.enhance_device(torch.from_numpy(x).cuda(),f,it,0.6,False,False,False,False).cpu().numpy()
    sc=np.max(np.abs(ex))
    print((C,n,f,it),"maxerr gpu/or: %.3g %.3g"%(np.max(np.abs(got-ex))/sc,np.max(np.abs(want-ex))/sc),
          "rms gpu/or: %.3g %.3g"%(np.sqrt(np.mean((got-ex)**2))/sc,np.sqrt(np.mean((want-ex)**2))/sc),
          "lsd g-o %.4g g-e %.4g o-e %.4g"%(om.lsd_audio(want,got)[0],om.lsd_audio(ex,got)[0],om.lsd_audio(ex,want)[0]))
# node level (PCM16 in/out), C1-like and C3-like
for (C,n,sr,kb,it) in [(1,160000,16000,1411,50),(2,480000,48000,1536,30)]:
    cs=(synth(C,n,seed=n,scale=0.5,integer=False)).astype(np.float32)
    want,sro=ofl.node_run(cs,sr,it,0.6,kb,True,True)
    f=fe.upscale_factor(sr,C,kb)
    got=fe.enhance_device(torch.from_numpy(cs).cuda(),f,it,0.6,True,True,True,True).cpu().numpy()
    lsb=np.abs(got-want)*32768
    print("node",(C,n,sr,kb,it),"f",f,"max lsb %.2f frac>0.5 %.3g"%(lsb.max(),np.mean(lsb>0.5)),"LSD node %.4g p95 %.4g"%om.lsd_audio(want,got))
