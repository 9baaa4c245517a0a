"""dev: paired chirp-z vs the oracle on a matrix of lengths / kinds (run on the GPU box)."""
import sys, time
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from packload import load_pack
load_pack()
from egregora_amd import fatllama_engine as fe
from oracle import fatllama as ofl
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fatllama import synth

def run(x, f, iters, thr, split=None, **kw):
    y = fe.enhance_device(torch.from_numpy(x).cuda(), f, iters, thr, False, False, False, False, split=split, **kw)
    torch.cuda.synchronize()
    return y.cpu().numpy()

rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
cases = [(1, 101, 1, 3, None), (2, 3, 1, 2, None), (2, 101, 1, 3, None), (3, 1001, 1, 3, None), (1, 7919 * 2, 1, 3, None), (2, 7919 * 2, 1, 3, None),
         (2, 4800, 1, 3, "chirpz1"), (2, 4800, 1, 3, "chirpz2"), (1, 4804, 1, 3, "chirpz1"), (2, 4801, 2, 3, None), (1, 33333, 1, 4, None),
         (2, 48002, 1, 5, None), (2, 48004, 1, 5, "chirpz1"), (2, 48001, 1, 5, None), (1, 1000003, 1, 2, None), (2, 1000002, 1, 2, None),
         (2, 2880002, 1, 2, None), (2, 2880001, 1, 2, None)]
if len(sys.argv) > 1:
    cases = cases[: int(sys.argv[1])]
for C, n, f, iters, split in cases:
    info = fe.plan_info(n, f)
    x = synth(C, n, seed=n)
    want = ofl.enhance_channels(x, f, iters, 0.6, normalize=False, autoscale=False)
    exact = ofl.enhance_channels(x, f, iters, 0.6, normalize=False, autoscale=False, exact=True)
    try:
        got = run(x, f, iters, 0.6, split=split)
    except Exception as e:
        print(C, n, f, split, "FAILED", e); continue
    scale = float(np.max(np.abs(want)))
    print(f"C={C} n={n} f={f} it={iters} split={split} kind={info['chirpz_kind']} P={info['M']}={info['M1']}x{info['M2']}x{info['M3']}: "
          f"max|got-want|/peak {float(np.max(np.abs(got - want))) / scale:.2e}  rms ratio {rms(got - exact) / (rms(want - exact) + 1e-30):.2f}  "
          f"max ratio {float(np.max(np.abs(got - exact))) / (float(np.max(np.abs(want - exact))) + 1e-30):.2f}", flush=True)
# bigger lengths: the 8192 / 16384-point row schedules and every column schedule of the menu (2 iterations, mono)
if len(sys.argv) <= 1:
    from itertools import product
    menu = [512, 560, 600, 640, 672, 720, 768, 800, 840, 900, 960, 1024]
    for L, nc in [(l, 1024) for l in menu] + [(720, 2048), (900, 4096), (720, 8192), (1024, 16384)]:
        D = (L * nc + 1) // 2 - 3
        n = 2 * D if (L // 8) % 2 == 0 else D | 1          # alternate kinds
        info = fe.plan_info(n, 1)
        x = synth(1, n, seed=n)
        want = ofl.enhance_channels(x, 1, 2, 0.6, normalize=False, autoscale=False)
        exact = ofl.enhance_channels(x, 1, 2, 0.6, normalize=False, autoscale=False, exact=True)
        got = run(x, 1, 2, 0.6)
        scale = float(np.max(np.abs(want)))
        print(f"menu {L}x{nc}: n={n} kind={info['chirpz_kind']} plan {info['M1']}x{info['M2']}: max|got-want|/peak {float(np.max(np.abs(got - want))) / scale:.2e} "
              f"rms ratio {rms(got - exact) / rms(want - exact):.2f}", flush=True)
