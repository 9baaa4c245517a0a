#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload "chain60" (the two stages BASELINE.json's metric names, in the order of the reference's example workflow
LoadAudio -> EgregoraAudioUpscaler -> EgregoraFatLlamaGPU): per GPU 60 s of stereo 48 kHz audio resident in HBM.
  stage 1  FlashSR: ONE file of N x 60 s is windowed into 5.12 s chunks (hop 4.62 s); rank r runs the contiguous
           block r of the chunk list (rows = chunks x 2 channels batched through student_ldm 1-step UNet + VAE +
           sr_vocoder, declared architecture with seeded synthetic weights), ONE RCCL all-gather of the prediction
           blocks, Hann WOLA on every rank.
  stage 2  Fat-Llama: each rank enhances its own 60 s slice of the stage-1 output, max_iterations = 800,
           threshold 0.6, normalise on, autoscale off, factor 1, PCM_16 hops included, every iteration executed.
A step = stage 1 + stage 2; value = N x 60 audio-seconds / max-over-ranks step time  (weak scaling: per-GPU work
is fixed as N grows).  `parts` reports each stage alone plus BASELINE configs[1] (one stereo 5.12 s chunk).
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
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
MFMA_BF16_PEAK_TFS = 2500.0  # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
SR = 48000
SEG = 60 * SR                # samples per GPU


def synth(seed, n, channels=2):
    """SURVEY 8(d) recipe: decorrelated channels, 8 log-spaced sines 80 Hz..6 kHz (1/k) + noise, peak 0.5 FS."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64) / SR
    chans = []
    for c in range(channels):
        f = np.geomspace(80.0, 6000.0, 8) * (1.0 + 0.013 * c)
        x = sum(np.sin(2 * np.pi * fk * t + rng.uniform(0, 2 * np.pi)) / (k + 1) for k, fk in enumerate(f))
        chans.append(x + rng.standard_normal(n) * 0.01)
    x = np.stack(chans)
    return (0.5 * x / np.max(np.abs(x))).astype(np.float32)


def cpu_baseline_fatllama(x, budget_s=20.0):
    """Oracle restatement of the Fat-Llama loop (scipy pocketfft complex64: one whole-signal FFT + threshold +
    IFFT per iteration per channel) timed on the host cores on a bounded sample, scaled to 800 iterations x C."""
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
    per = el / it
    return {"value": (x.shape[1] / SR) / (per * 800 * x.shape[0]), "unit": "audio-sec/sec", "cores": cores,
            "kind": "port",
            "sample": f"Fat-Llama stage only: {it} iterations of one {x.shape[1]}-sample channel ({el:.1f} s) scaled to 800 "
                      f"iterations x {x.shape[0]} channels; oracle/fatllama.py loop on scipy.fft complex64, workers={cores}",
            "sec_per_iteration_channel": per}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=800, help="Fat-Llama max_iterations (headline = 800)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--rows", type=int, default=0, help="FlashSR rows per pass (default: engine setting)")
    ap.add_argument("--only", default="", help="'flashsr' or 'fatllama': time one stage only (dev)")
    ap.add_argument("--workload", default="chain60", choices=["chain60", "c4"],
                    help="chain60 (default, weak scaling: 60 s per GPU through FlashSR + Fat-Llama) or c4 = BASELINE configs[3]: "
                         "FlashSR long-form, ONE 10-minute stereo file chunk-sharded over the N GPUs with one all-gather + WOLA "
                         "(strong scaling: north_star's '>= 6x chunk-parallel speed-up at 8 GPUs')")
    ap.add_argument("--lean", action="store_true", help="skip the untimed extras (parts); used under rocprofv3 so the "
                                                        "per-kernel averages cover the timed workload only")
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
    from egregora_amd import (audio_glue as ag, fatllama_engine as fe, flashsr_arch as A,
                              flashsr_engine as E, native)
    from egregora_amd.egregora_audio_super_resolution import upscale_48k
    arch = native.require_device()
    if args.rows > 0:
        E.ROWS_PER_PASS = args.rows

    cfg = A.FlashSRConfig()
    eng = E.FlashSREngine(cfg, A.init_params(cfg, seed=0))
    E.set_engine(eng)

    c4 = args.workload == "c4"
    if c4:
        args.only = "flashsr"
    total = 600 * SR if c4 else world * SEG
    x_all = torch.from_numpy(synth(404, total)).cuda()          # the whole file is replicated on every rank
    C = x_all.shape[0]
    n_chunks = len(ag.spans(total))
    fl_flags = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True)

    def stage_flashsr():
        return upscale_48k(x_all, False)                          # [C, total] on every rank

    def stage_fatllama(y48):
        seg = y48[:, rank * SEG:(rank + 1) * SEG].contiguous()
        return fe.enhance_device(seg, 1, args.iters, 0.6, **fl_flags)

    def step():
        if args.only == "fatllama":
            return stage_fatllama(x_all)
        y = stage_flashsr()
        return y if args.only == "flashsr" else stage_fatllama(y)

    def timed(fn, n):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, r

    for _ in range(args.warmup):
        step()
    el, _ = timed(step, args.steps)

    # ---- untimed extras: per-stage times, configs[1], per-kernel HIP-event timing for the rooflines ----
    el_fs, y48 = timed(stage_flashsr, 1)
    el_fl, y_fl = timed(lambda: stage_fatllama(y48), 1)
    # sanity outside the timed region: every sample finite, and all `iters` iterations land where ONE iteration lands (the loop is
    # a projection; a drift between the two would mean the long run is not doing the arithmetic the metric names)
    from egregora_amd import device_ops
    seg48 = y48[:, rank * SEG:(rank + 1) * SEG].contiguous() if not c4 else y48[:, :SEG].contiguous()
    y_one = fe.enhance_device(seg48, 1, 1, 0.6, **fl_flags)
    if c4:
        y_fl = fe.enhance_device(seg48, 1, args.iters, 0.6, **fl_flags)
    assert bool(torch.isfinite(y48).all()) and bool(torch.isfinite(y_fl).all()), "non-finite output"
    lsd_iters = device_ops.lsd(y_one[:, :20 * SR].contiguous(), y_fl[:, :20 * SR].contiguous())
    assert lsd_iters[0] < 0.05, lsd_iters
    el_c2 = None
    if not args.lean:
        x_c2 = x_all[:, :cfg.chunk].contiguous()
        upscale_48k(x_c2, False)
        el_c2, _ = timed(lambda: upscale_48k(x_c2, False), 3)
    prof = eng.c_profile(stage_flashsr)           # HIP events around every MFMA contraction launch of the library's graph walk
    fe.enhance_device(seg48, 1, args.iters, 0.6, profile=True, **fl_flags)
    kt = fe.kernel_times(SEG, C, 1, local_rank)

    if rank == 0:
        audio_s = total / SR
        variants = {k: v for k, v in prof.items() if k.startswith("k_conv")}
        fconv_all = sum(v[1] for v in variants.values())
        dom_conv = max(variants, key=lambda k: variants[k][2])          # the variant with the most GPU time
        nconv, fconv, tconv = variants[dom_conv]
        conv_tfs = fconv / (tconv * 1e-3) / 1e12 if tconv > 0 else 0.0      # fp32-equivalent (algorithmic) rate
        # k_conv_s3 executes SIX bf16 MFMA products per fp32 multiply-add: its roofline is the dense bf16 peak and
        # `achieved` counts the executed bf16 flops (6 x algorithmic)
        s3 = dom_conv.startswith("k_conv_s3") or dom_conv.startswith("k_conv1d_s3")
        mfma_mult, mfma_peak = (6.0, MFMA_BF16_PEAK_TFS) if s3 else (1.0, MFMA_F32_PEAK_TFS)
        # the loop runs the channels as two concurrent pipelines (one launch = C/2 channels) unless EGR_FL_STREAMS=1
        groups = 2 if (C >= 2 and os.environ.get("EGR_FL_STREAMS", "2") != "1") else 1
        row_bytes = 8.0 * SEG * C / groups
        dom_ms = max(kt["row_ms"], kt["col_ms"])
        dom = "k_row" if kt["row_ms"] >= kt["col_ms"] else "k_col<1>"
        hbm_ach = row_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = conv_traffic = None
        try:                # HBM bytes per launch from the committed PMC passes (tools/make_traffic_json.py)
            tk = json.loads((ROOT / "profiles" / "traffic.json").read_text()).get("kernels", {})
            cands = ["k_row<false, 1>", "k_row<false, 0>", "k_row<false>"] if dom == "k_row" else ["k_col<1, 2>", "k_col<1, 0>", "k_col<1>"]
            traffic = next((tk[c]["bytes"] for c in cands if c in tk), None)
            conv_traffic = tk.get(dom_conv, {}).get("bytes")
        except Exception:
            pass
        info = fe.plan_info(SEG, 1)
        out = {
            "metric": METRIC, "value": args.steps * audio_s / el, "unit": "audio-sec/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "strong" if c4 else "weak", "vs_baseline": None, "dtype": "f32" if eng.mfma == "f32" else "f32(bf16x3)",
            "data": "synthetic",
            "config": {"workload": ("c4: BASELINE configs[3], ONE 10 min stereo 48 kHz file, FlashSR (5.12 s chunks, hop 4.62 s, "
                                    "student_ldm 1-step + VAE + sr_vocoder, declared architecture, synthetic weights) chunk-sharded over "
                                    "the GPUs with one all-gather, WOLA on every rank; total work fixed as N grows" if c4 else
                                    "chain60: per GPU 60 s stereo 48 kHz; FlashSR (5.12 s chunks, hop 4.62 s, student_ldm "
                                    "1-step + VAE + sr_vocoder, declared architecture, synthetic weights, chunk-sharded with one "
                                    "all-gather, WOLA) then Fat-Llama max_iterations=%d thr=0.6 normalize on autoscale off, "
                                    "threshold variant '%s' (SPEC.md section 3: absolute level, hard threshold, time-domain "
                                    "pre-threshold, linear up-rating unless listed)" % (args.iters, os.environ.get("EGREGORA_FATLLAMA_SPEC", "") or "default"))
                                   + (f" [only={args.only}]" if args.only and not c4 else ""),
                       "flashsr_executor": "egr_flashsr_infer (C ABI, csrc/egr_flashsr.cpp)" if E.EXECUTOR != "python" else "python driver",
                       "lsd_800_vs_1_iteration_db": lsd_iters[0],
                       "mfma": ("fp32 operands split exactly into three bf16 terms, six partial products on "
                                "v_mfma_f32_32x32x16_bf16 with fp32 accumulation: error vs float64 <= the f32-MFMA kernel's "
                                "(tests/test_gpu_flashsr.py::test_split3_conv_error_vs_float64)") if eng.mfma != "f32"
                       else "v_mfma_f32_32x32x2_f32",
                       "arch": arch, "chunks": n_chunks, "rows_per_pass": E.ROWS_PER_PASS,
                       "fatllama_split": [info["M1"], info["M2"]]},
            "parts": {
                "flashsr_stage_xrt": audio_s / el_fs, "flashsr_stage_ms": 1e3 * el_fs,
                "fatllama_stage_xrt": audio_s / el_fl, "fatllama_stage_ms": 1e3 * el_fl,
                "configs1_flashsr_single_chunk_stereo_xrt": (3 * 5.12 / el_c2) if el_c2 else None,
                "configs1_ms": (1e3 * el_c2 / 3) if el_c2 else None,
                "flashsr_flops_per_row": fconv_all / max(1, (len(ag.spans(total)) + world - 1) // world * C),
                "flashsr_scratch_arena_gb": native.lib().egr_flashsr_scratch_bytes(eng.handle) / 1e9,
                "conv_variants": {k: {"launches": v[0], "tflops": v[1] / 1e12, "ms": v[2],
                                      "avg_launch_ms": v[2] / max(1, v[0])} for k, v in variants.items()},
            },
            # dominant kernel of the step: the implicit-GEMM convolution (all dense contractions of FlashSR)
            "roofline": {"bound": "mfma", "kernel": dom_conv, "achieved": conv_tfs * mfma_mult, "peak": mfma_peak,
                         "unit": "TFLOP/s", "frac": conv_tfs * mfma_mult / mfma_peak, "traffic": conv_traffic,
                         "mfma_dtype": "bf16 (6 executed products per fp32 multiply-add)" if s3 else "f32",
                         "fp32_equivalent_tflops": conv_tfs, "vs_f32_mfma_peak": conv_tfs / MFMA_F32_PEAK_TFS,
                         "launches": nconv, "flops_total": fconv, "ms_total": tconv,
                         "avg_flops_per_launch": fconv / max(1, nconv), "avg_launch_ms": tconv / max(1, nconv)},
            "roofline_fatllama": {"bound": "hbm", "kernel": dom, "achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": hbm_ach / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": row_bytes,
                                  "avg_launch_ms": dom_ms, "k_row_ms": kt["row_ms"], "k_col_ms": kt["col_ms"],
                                  "launches": [kt["row_launches"], kt["col_launches"]], "concurrent_pipelines": groups,
                                  # all loop launches' algorithmic bytes / the stage's wall time (pipelines overlap)
                                  "effective_gbs": (kt["row_launches"] + kt["col_launches"]) * row_bytes / (el_fl * 1e9),
                                  # SURVEY 8(d)'s figure for UNFUSED passes (32 N bytes per iteration per channel: 4 transform
                                  # passes each reading and writing the state) over the stage time; this design moves half of it
                                  "survey_32N": {"bytes_total": 32.0 * SEG * C * args.iters,
                                                 "achieved": 32.0 * SEG * C * args.iters / (el_fl * 1e9),
                                                 "frac": 32.0 * SEG * C * args.iters / (el_fl * 1e9) / HBM_PEAK_GBS}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_fatllama(x_all[:, :SEG].cpu().numpy(), args.cpu_budget)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
