"""dev: 800-iteration Fat-Llama stage time for lengths that take the three kinds of plan (scheduled, run-time schedule, chirp-z)."""
import sys, time
sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe, native
native.require_device()
IT = int(__import__('os').environ.get('PROBE_ITERS', '800'))
fl = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True)
import os
for n in [int(v) for v in os.environ.get('PROBE_N', '2880000,2646000,2880002,2646002,960000,5760000').split(',')]:
    x = (0.3 * torch.randn(2, n, device="cuda")).clamp(-1, 1)
    info = fe.plan_info(n, 1)
    for _ in range(2):
        fe.enhance_device(x, 1, IT, 0.6, **fl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fe.enhance_device(x, 1, IT, 0.6, **fl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N = {n}: {dt*1e3:.1f} ms for {IT} iterations x 2 channels ({n/48000/dt:.0f} xRT at 48 kHz); plan {({k: info[k] for k in ('M1','M2','M3','levels','TC') if k in info})} bluestein={info.get('bluestein')}", flush=True)
