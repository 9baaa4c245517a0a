"""Dev: time the Fat-Llama C3 stage alone (60 s stereo, 800 iterations) with per-kernel event averages."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
x=torch.from_numpy((0.3*np.random.default_rng(0).standard_normal((2,2880000))).astype(np.float32)).cuda()
f=dict(normalize=True,autoscale=False,pcm_in=True,node_post=True)
fe.enhance_device(x,1,800,0.6,**f); torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(3): fe.enhance_device(x,1,800,0.6,**f)
torch.cuda.synchronize(); ms=(time.perf_counter()-t)/3*1e3
fe.enhance_device(x,1,800,0.6,profile=True,**f)
k=fe.kernel_times(2880000,2,1,0); i=fe.plan_info(2880000,1)
print(f"{ms:7.1f} ms  row {k['row_ms']*1e3:6.1f} us  col {k['col_ms']*1e3:6.1f} us  split {i['M1']}x{i['M2']} tc {i['TC']}")
