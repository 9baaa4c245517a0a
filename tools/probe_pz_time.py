"""dev: wall time of the chirp-z stage (stereo, 800 iterations) for a few lengths; per-kernel HIP-event averages."""
import ctypes as C, sys, time
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from packload import load_pack
load_pack()
from egregora_amd import fatllama_engine as fe, native
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fatllama import synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 800
lengths = [int(a) for a in sys.argv[2:]] or [2880002, 2880001, 2880004]
L = native.lib()
for n in lengths:
    info = fe.plan_info(n, 1)
    x = torch.from_numpy(synth(2, n, seed=1)).cuda()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = fe.enhance_device(x, 1, iters, 0.6, True, False, True, True)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    fe.enhance_device(x, 1, 12, 0.6, True, False, True, True, profile=True)
    plan = fe._plan(n, 2, 1, 0)
    ms = (C.c_double * 3)(); cnt = (C.c_int64 * 3)()
    native.check(L.egr_fatllama_kernel_times3(C.c_void_p(plan), ms, cnt), "times3")
    L.egr_fatllama_set_profiling(C.c_void_p(plan), 0)
    print(f"n={n} kind={info['chirpz_kind']} P={info['M']}={info['M1']}x{info['M2']}x{info['M3']} radices {info['radix1']} {info['radix2']}: "
          f"{1e3 * el:.1f} ms for {iters} iterations; events (us): rowconv {1e3 * ms[0]:.1f} pair {1e3 * ms[1]:.1f} crop {1e3 * ms[2]:.1f} (n={list(cnt)})", flush=True)
