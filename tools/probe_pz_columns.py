"""dev: the paired chirp-z loop with and without the thread-per-(row class, column) column kernels (k_pzcol_wl, k_pzpair_wl;
EGR_PZ_COLWL) for one length per instantiated column length L (plans L x 4096): agreement and 200-iteration stage time."""
import os
import sys
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
from test_gpu_fatllama import synth

FL = dict(normalize=False, autoscale=False, pcm_in=False, node_post=False)
IT = int(os.environ.get("PROBE_ITERS", "200"))
for L in (512, 560, 600, 640, 672, 720, 768, 800, 840, 900, 960, 1024):
    n = L * 4096 - 6
    info = fe.plan_info(n, 1)
    x = torch.from_numpy(synth(2, n, seed=L)).cuda()
    res, tim = {}, {}
    for v in ("1", "0"):
        os.environ["EGR_PZ_COLWL"] = v
        fe.release_plans()
        res[v] = fe.enhance_device(x, 1, 3, 0.6, **FL).cpu().numpy()
        fe.enhance_device(x, 1, IT, 0.6, **FL)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fe.enhance_device(x, 1, IT, 0.6, **FL)
        torch.cuda.synchronize(); tim[v] = time.perf_counter() - t0
    fe.release_plans()
    sc = float(np.abs(res["0"]).max())
    print(f"L = {L}: n = {n} plan {info['M1']} x {info['M2']} kind {info['chirpz_kind']}: {1e3 * tim['0']:.1f} -> {1e3 * tim['1']:.1f} ms per {IT} iterations; "
          f"max diff {float(np.abs(res['1'] - res['0']).max()) / sc:.2e} of the peak", flush=True)
