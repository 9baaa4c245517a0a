"""dev: the Fat-Llama loop above the memory-side cache (BASELINE configs[4]: N = 172.8 M samples per channel, stereo, three-level plan).
Per-pass HIP-event times and the stage time for a list of explicit factorisations M1 x M2 x M3 of N / 2 and column-tile widths.
  PROBE_N=172800000 PROBE_ITERS=20 PROBE_SPLITS="0;640,72,1875,0;200,180,2400,16" python tools/probe_c5_plans.py
A split is "m1,m2,m3,tc" ("0" = the planner's own choice)."""
import os
import sys
import time

sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe, native

native.require_device()
N = int(os.environ.get('PROBE_N', '172800000'))
IT = int(os.environ.get('PROBE_ITERS', '20'))
C = int(os.environ.get('PROBE_CH', '2'))
fl = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True)
x = (0.3 * torch.randn(C, N, device="cuda")).clamp(-1, 1)
for spec in os.environ.get('PROBE_SPLITS', '0').split(';'):
    v = [int(t) for t in spec.split(',')]
    split = tuple(v[:3]) if len(v) >= 3 else None
    tc = v[3] if len(v) >= 4 else 0
    try:
        fe.release_plans()
        fe.enhance_device(x, 1, 2, 0.6, split=split, tc_hint=tc, **fl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = fe.enhance_device(x, 1, IT, 0.6, split=split, tc_hint=tc, **fl)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fe.enhance_device(x, 1, 2, 0.6, split=split, tc_hint=tc, **fl)
        torch.cuda.synchronize(); d2 = time.perf_counter() - t0
        per_it = (dt - d2) / (IT - 2)
        fe.enhance_device(x, 1, IT, 0.6, profile=True, split=split, tc_hint=tc, **fl)
        L = native.lib()
        import ctypes as Ct
        plan = fe._plan(N, C, 1, 0, 0, tc, split, None)
        ms = (Ct.c_double * 3)(); cnt = (Ct.c_int64 * 3)()
        native.check(L.egr_fatllama_kernel_times3(Ct.c_void_p(plan), ms, cnt), "kernel_times3")
        L.egr_fatllama_set_profiling(Ct.c_void_p(plan), 0)
        info = fe.plan_info(N, 1) if split is None else {}
        state_gb = 8.0 * (N // 2) * C / 1e9
        print(f"split {spec}: {1e3 * per_it:.2f} ms / iteration ({1e3 * dt:.0f} ms for {IT}); passes [row, colA, colB] ms = "
              f"{[round(m, 3) for m in ms]} launches {list(cnt)}; state {state_gb:.2f} GB -> "
              f"{8 * state_gb / per_it / 1e3:.2f} TB/s over 4 passes; finite={bool(torch.isfinite(y).all())} "
              f"{ {k: info[k] for k in ('M1', 'M2', 'M3', 'TC')} if info else ''}", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"split {spec}: FAILED {e}", flush=True)
