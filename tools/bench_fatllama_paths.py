"""Dev: throughput of the three Fat-Llama transform paths (2-level, 3-level, Bluestein)."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
def run(C,n,iters,label):
    x=torch.from_numpy((0.3*np.random.default_rng(0).standard_normal((C,n))).astype(np.float32)).cuda()
    f=dict(normalize=True,autoscale=False,pcm_in=True,node_post=True)
    fe.enhance_device(x,1,2,0.6,**f); torch.cuda.synchronize()
    t=time.perf_counter(); fe.enhance_device(x,1,iters,0.6,**f); torch.cuda.synchronize(); dt=time.perf_counter()-t
    i=fe.plan_info(n,1)
    print(f"{label:34s} C={C} N={n:>10d} iters={iters:4d}: {dt*1e3:9.1f} ms  {dt/iters/C*1e6:8.1f} us/iter/ch  {n/48000/dt:8.1f} xRT@48k  plan M={i['M']} {i['M1']}x{i['M2']}x{i['M3']} bluestein={i['bluestein']}")
    fe.release_plans()
run(2,2880000,800,"C3 60 s stereo (2-level)")
run(2,9600000,200,"200 s stereo (3-level)")
run(2,28800000,200,"10 min stereo (3-level)")
run(2,86400000,200,"30 min @48k stereo (3-level)")
run(2,2880001,100,"60 s + 1 sample (Bluestein)")
run(1,1000003,100,"prime length 1000003 (Bluestein)")
