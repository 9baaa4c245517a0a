#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic audio already resident in HBM:
  workload "fatllama_c3": Fat-Llama iterative spectral enhance of 60 s stereo 48 kHz, max_iterations=800,
                          threshold 0.6, normalise on, autoscale off (BASELINE.json configs[2]; every
                          iteration is executed).  Fat-Llama does not chunk, so N>1 runs N independent
                          replicas (one file per rank, "replicas only", scaling = weak).
value = audio-seconds processed by all ranks / max-over-ranks wall time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

METRIC = "audio-sec/sec (xRT) FlashSR 48kHz + Fat-Llama 800-iter at 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_c3(seed=303, n=2880000, sr=48000):
    """SURVEY 8(d) C3 input: decorrelated stereo, sum of 8 log-spaced sines + noise, peak 0.5 FS."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64) / sr
    chans = []
    for c in range(2):
        f = np.geomspace(80.0, 6000.0, 8) * (1.0 + 0.013 * c)
        x = sum(np.sin(2 * np.pi * fk * t + rng.uniform(0, 2 * np.pi)) / (k + 1) for k, fk in enumerate(f))
        x = x + rng.standard_normal(n) * 0.01
        chans.append(x)
    x = np.stack(chans)
    x = 0.5 * x / np.max(np.abs(x))
    return x.astype(np.float32)


def cpu_baseline_fatllama(x, sr, budget_s=20.0):
    """Oracle restatement (scipy pocketfft, complex64, one whole-signal FFT + threshold + IFFT per iteration
    per channel) timed on the host cores on a bounded sample: full-length channels, as many iterations as fit
    the budget (at least 2), then scaled to the 800-iteration workload."""
    import scipy.fft as sfft
    from oracle import fatllama as ofl
    cores = os.cpu_count() or 1
    xi = ofl.pcm16_write(x).astype(np.float32)
    thr = np.float32(0.6)
    d = np.where(np.abs(xi[0]) > thr, xi[0], np.float32(0)).astype(np.float32)
    it, t0 = 0, time.perf_counter()
    with sfft.set_workers(cores):
        while True:
            X = sfft.fft(d.astype(np.complex64))
            X = np.where(np.abs(X) > thr, X, 0).astype(np.complex64)
            d = sfft.ifft(X).real.astype(np.float32)
            it += 1
            el = time.perf_counter() - t0
            if (el > budget_s and it >= 2) or it >= 800:
                break
    per_iter_ch = el / it
    total = per_iter_ch * 800 * x.shape[0]
    return {"value": (x.shape[1] / sr) / total, "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "sample": f"{it} iterations of one 2,880,000-sample channel ({el:.1f} s), scaled to 800 iterations x "
                      f"{x.shape[0]} channels; scipy.fft complex64, workers={cores}",
            "sec_per_iteration_channel": per_iter_ch}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=800, help="Fat-Llama max_iterations (headline = 800)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--m1", type=int, default=0)
    ap.add_argument("--tc", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)

    from packload import load_pack
    load_pack()
    from egregora_amd import fatllama_engine as fe, native
    arch = native.require_device()

    sr = 48000
    x = synth_c3(seed=303 + rank)
    xd = torch.from_numpy(x).cuda()
    C, n = x.shape
    flags = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True, m1_hint=args.m1, tc_hint=args.tc)

    def step(profile=False):
        return fe.enhance_device(xd, 1, args.iters, 0.6, profile=profile, **flags)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # per-kernel HIP-event timing of the two loop kernels on the launch stream (separate, untimed pass)
    step(profile=True)
    kt = fe.kernel_times(n, C, 1, local_rank) if (args.m1 == 0 and args.tc == 0) else None
    if kt is None:
        from egregora_amd.fatllama_engine import _plan
        import ctypes as Cc
        L = native.lib()
        plan = _plan(n, C, 1, local_rank, args.m1, args.tc)
        r, c = Cc.c_double(), Cc.c_double()
        nr, nc = Cc.c_int64(), Cc.c_int64()
        L.egr_fatllama_kernel_times(Cc.c_void_p(plan), Cc.byref(r), Cc.byref(c), Cc.byref(nr), Cc.byref(nc))
        kt = {"row_ms": r.value, "col_ms": c.value, "row_launches": nr.value, "col_launches": nc.value}

    if rank == 0:
        info = fe.plan_info(n, 1, args.m1)
        audio_s = n / sr
        value = world * args.steps * audio_s / el
        # algorithmic bytes per launch of a loop kernel: read 4N + write 4N bytes per channel
        bytes_per_launch = 8.0 * n * C
        dom = "k_row" if kt["row_ms"] >= kt["col_ms"] else "k_col<1>"
        dom_ms = max(kt["row_ms"], kt["col_ms"])
        achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        try:   # PMC-measured bytes per launch, recorded by tools/profile_round.sh runs (profiles/traffic.json)
            tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
            if args.iters == 800 and args.m1 == 0 and args.tc == 0:
                traffic = tj.get("fatllama_c3", {}).get(dom)
        except Exception:
            traffic = None
        out = {
            "metric": METRIC, "value": value, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "fatllama_c3: Fat-Llama 60 s stereo 48 kHz, max_iterations=%d, thr=0.6, "
                                   "normalize on, autoscale off, factor 1 (BASELINE configs[2]); one file per rank"
                                   % args.iters,
                       "arch": arch, "split": [info["M1"], info["M2"]], "tile_cols": info["TC"]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": bytes_per_launch, "avg_launch_ms": dom_ms,
                         "k_row_ms": kt["row_ms"], "k_col_ms": kt["col_ms"],
                         "launches": [kt["row_launches"], kt["col_launches"]]},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_fatllama(x, sr, args.cpu_budget)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
