"""Host driver of the device Fat-Llama engine (csrc/egr_fatllama.hip) behind the C ABI.

Stands in for `fat_llama.audio_fattener.feed.upscale` + the temp-file hops of the reference
(egregora_fat_llama_gpu.py:161-224, :34-37, :291-294).  Sample-rate bookkeeping (a14): the reference takes
the output rate from the file upstream wrote, i.e. sr * upscale_factor.
"""
import ctypes as C
import os
import time
from collections import OrderedDict
from typing import Tuple

import torch

from . import native, streams

SAMPLE_WIDTH_BYTES = 2        # temp WAVs are PCM_16 (libsndfile default for float data)

# Which reading of upstream's unverified threshold / interpolation semantics the nodes run (SPEC.md section 3): a comma list out of
# relative, soft, no_init_thr, zero_stuff, linspace, ratio_then_int; empty = absolute level, hard threshold, time-domain
# pre-threshold, linear up-rating by the rounded integer factor.  ratio_then_int (output length int(n * ratio), needs linspace)
# is a host-side rule: it has no device flag of its own.
VARIANT_FLAGS = {"relative": native.FL_THR_RELATIVE, "soft": native.FL_THR_SOFT, "no_init_thr": native.FL_NO_INIT_THR,
                 "zero_stuff": native.FL_ZERO_STUFF, "linspace": native.FL_INTERP_LINSPACE, "ratio_then_int": native.FL_INTERP_LINSPACE,
                 # not a reading of upstream: how the device obtains the relative level (a pass per iteration instead of the carried maximum)
                 "recompute": native.FL_THR_RECOMPUTE}


def variant_names(names=None):
    names = os.environ.get("EGREGORA_FATLLAMA_SPEC", "") if names is None else names
    return [t.strip() for t in str(names).split(",") if t.strip()]


def upscale_ratio(sr: int, channels: int, target_bitrate_kbps: int) -> float:
    """target bits/s over source bits/s, at least 1 (SPEC.md factor_mode "ratio_then_int")."""
    return max(1.0, (target_bitrate_kbps * 1000.0) / (sr * channels * 8 * SAMPLE_WIDTH_BYTES))


def variant_flags(names=None) -> int:
    names = os.environ.get("EGREGORA_FATLLAMA_SPEC", "") if names is None else names
    flags = 0
    for tok in (t.strip() for t in str(names).split(",")):
        if tok:
            if tok not in VARIANT_FLAGS:
                raise RuntimeError(f"EGREGORA_FATLLAMA_SPEC: unknown variant {tok!r} (known: {sorted(VARIANT_FLAGS)})")
            flags |= VARIANT_FLAGS[tok]
    return flags

_PLANS: "OrderedDict[tuple, int]" = OrderedDict()
_MAX_PLANS = 8          # (a plan of 60 s stereo holds 46 MB; EGREGORA_DEVICES adds one per device)


def upscale_factor(sr: int, channels: int, target_bitrate_kbps: int) -> int:
    """round(target bits/s / source bits/s), at least 1 (upstream rule as recalled; see oracle/fatllama.py)."""
    src_bps = sr * channels * 8 * SAMPLE_WIDTH_BYTES
    return max(1, int(round((target_bitrate_kbps * 1000.0) / src_bps)))


def _plan(n_in: int, channels: int, factor: int, device: int, m1_hint: int = 0, tc_hint: int = 0, split=None, n_out=None, slot: int = 0):
    """split = (m1, m2, m3) forces an explicit factorisation of N/2 (m3 = 1: two levels); split = "chirpz" forces the paired chirp-z
    path that lengths without a packed-real plan take automatically ("chirpz1" / "chirpz2": its even/odd-packing / channel-pair
    kind), split = "bluestein" the legacy full-complex chirp-z.  slot: a plan of its own for the same shape (a plan holds the loop's
    state and serves one caller at a time: enhance_devices gives every position of EGREGORA_DEVICES its own, also when an index repeats)."""
    key = (n_in, channels, factor, device, m1_hint, tc_hint, split, n_out, slot)
    h = _PLANS.get(key)
    if h is not None:
        _PLANS.move_to_end(key)
        return h
    L = native.lib()
    out = C.c_void_p()
    if n_out is not None:
        native.check(L.egr_fatllama_plan_create_n(C.byref(out), n_in, int(n_out), channels), "egr_fatllama_plan_create_n")
    elif split == "bluestein":
        native.check(L.egr_fatllama_plan_create_bluestein(C.byref(out), n_in, channels, factor),
                     "egr_fatllama_plan_create_bluestein")
    elif isinstance(split, str) and split.startswith("chirpz"):
        native.check(L.egr_fatllama_plan_create_chirpz(C.byref(out), n_in, channels, factor, int(split[6:] or 0)),
                     "egr_fatllama_plan_create_chirpz")
    elif split is not None:
        native.check(L.egr_fatllama_plan_create_ex(C.byref(out), n_in, channels, factor, int(split[0]), int(split[1]),
                                                   int(split[2]), tc_hint), "egr_fatllama_plan_create_ex")
    else:
        native.check(L.egr_fatllama_plan_create(C.byref(out), n_in, channels, factor, m1_hint, tc_hint),
                     "egr_fatllama_plan_create")
    _PLANS[key] = out.value
    while len(_PLANS) > _MAX_PLANS:
        _, old = _PLANS.popitem(last=False)
        _SIDE_SET.difference_update({k for k in _SIDE_SET if k[0] == old})
        L.egr_fatllama_plan_destroy(C.c_void_p(old))
    return out.value


def release_plans():
    L = native.lib()
    while _PLANS:
        _, old = _PLANS.popitem(last=False)
        _SIDE_SET.difference_update({k for k in _SIDE_SET if k[0] == old})
        L.egr_fatllama_plan_destroy(C.c_void_p(old))


def plan_info(n_in: int, factor: int, m1_hint: int = 0) -> dict:
    """Host-only planning query (works without a GPU)."""
    L = native.lib()
    info = (C.c_int64 * native.FL_INFO_LEN)()
    rc = L.egr_fatllama_plan_query(n_in, factor, m1_hint, info)
    v = list(info)
    d = {"supported": bool(v[0]) and rc == 0, "bluestein": v[0] == 2, "N": v[1], "M": v[2], "M1": v[3], "M2": v[4], "TC": v[5],
         "radix1": [r for r in v[8:8 + v[6]]], "radix2": [r for r in v[22:22 + v[7]]],
         "lds_col": v[36], "lds_row": v[37], "M3": v[38], "levels": v[39], "chirpz_kind": v[40], "D": v[41]}
    if rc != 0:
        d["error"] = native.last_error()
    return d


_TUNED = {}          # (device, caller stream) -> 1 graph replay / 0 plain launches, decided once per process
_SIDE_SET = set()    # (plan, caller stream) pairs that already hold the verified side stream
TUNE_ITERS = 76      # three replays of the captured 25-iteration graph + the plain remainder
TUNE_MIN_JOB = 400   # shorter jobs keep the default (graph replay): tuning would cost as much as the job


def _tune_pipelines(L, plan, x_ct, out, thr, flags, max_iterations):
    """Hand the plan a side stream verified to overlap with the caller's (streams.py) for its second channel pipeline, and pick
    graph replay or plain launches for the loop (identical bits).  How the runtime maps a graph's branches onto hardware queues
    depends on the streams the process has made, not on the signal length, so the timing runs ONCE per (device, caller stream) on
    the first long job -- two short runs per mode into `out`, which the real call overwrites next -- and every later plan reuses it."""
    cur = torch.cuda.current_stream()
    skey = (plan, cur.cuda_stream)
    if skey not in _SIDE_SET:
        side = streams.side_streams(1)
        if side:
            native.check(L.egr_fatllama_set_side_stream(C.c_void_p(plan), C.c_void_p(side[0].cuda_stream)), "egr_fatllama_set_side_stream")
        _SIDE_SET.add(skey)
    if os.environ.get("EGR_FL_GRAPH") is not None:
        return
    key = (x_ct.device.index or 0, cur.cuda_stream)
    best = _TUNED.get(key)
    if best is None:
        if max_iterations < TUNE_MIN_JOB:
            return
        t = {1: float("inf"), 0: float("inf")}
        for rnd in range(2):                                  # the modes alternate: a shader clock still ramping up must not favour the later one
            for mode in (1, 0):
                native.check(L.egr_fatllama_set_graph(C.c_void_p(plan), mode), "egr_fatllama_set_graph")
                for rep in range(3):                          # the first run of a mode captures the graph / warms the path
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    native.check(L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(x_ct), native.ptr(out), TUNE_ITERS, thr, flags,
                                                        native.stream_ptr()), "egr_fatllama_enhance")
                    torch.cuda.synchronize()
                    if rep or rnd:
                        t[mode] = min(t[mode], time.perf_counter() - t0)
        # graph replay is the default; plain launches only where they win clearly (round 6: a two-run timing picked the slower mode in three
        # of six bench processes -- 35.8 instead of 29.9 ms per 800-iteration stereo minute, profiles/r06/same_box_ab_*.txt)
        best = _TUNED[key] = 0 if t[0] < 0.93 * t[1] else 1
    native.check(L.egr_fatllama_set_graph(C.c_void_p(plan), best), "egr_fatllama_set_graph")


def enhance_device(x_ct: torch.Tensor, factor: int, max_iterations: int, threshold_value: float,
                   normalize: bool, autoscale: bool, pcm_in: bool, node_post: bool,
                   m1_hint: int = 0, tc_hint: int = 0, profile: bool = False, split=None, variant=None, n_out=None):
    """x_ct: [C,T] float32 CUDA tensor.  Returns [C,T*factor] float32 CUDA tensor (same stream); with n_out (and the linspace
    variant) [C,n_out] instead.  variant: comma list for variant_flags (None: the EGREGORA_FATLLAMA_SPEC environment variable)."""
    if not (x_ct.is_cuda and x_ct.dtype == torch.float32 and x_ct.dim() == 2):
        raise RuntimeError("enhance_device wants a [C,T] float32 tensor on the GPU")
    x_ct = x_ct.contiguous()
    Cn, T = x_ct.shape
    if T < 1:
        raise RuntimeError("empty audio")
    if n_out is not None and int(n_out) == T * factor:
        n_out = None
    plan = _plan(T, Cn, factor, x_ct.device.index or 0, m1_hint, tc_hint, split, n_out)
    out = torch.empty((Cn, T * factor if n_out is None else int(n_out)), dtype=torch.float32, device=x_ct.device)
    flags = ((native.FL_NORMALIZE if normalize else 0) | (native.FL_AUTOSCALE if autoscale else 0) |
             (native.FL_PCM_IN if pcm_in else 0) | (native.FL_NODE_POST if node_post else 0) | variant_flags(variant))
    L = native.lib()
    # (paired chirp-z plans keep their own side stream and the graph replay: tuning them by 76-iteration runs measured 9 % slower)
    if Cn >= 2 and max_iterations > 100 and not profile and not isinstance(split, str) and n_out is None and not plan_info(T, factor)["bluestein"]:
        _tune_pipelines(L, plan, x_ct, out, float(threshold_value), flags, int(max_iterations))
    if profile:
        L.egr_fatllama_set_profiling(C.c_void_p(plan), 1)
    native.check(L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(x_ct), native.ptr(out), int(max_iterations),
                                        float(threshold_value), flags, native.stream_ptr()), "egr_fatllama_enhance")
    return out


class _ChannelBlockBackend:
    """shard.sharded_channels on the device: egr_fatllama_enhance with EGR_FL_DEFER_FINALIZE on this rank's channels, egr_fatllama_joint_peak,
    egr_fatllama_finalize with the all-reduced peak (include/egregora_amd.h)."""

    def __init__(self, x_ct, factor, max_iterations, threshold_value, flags, n_out=None):
        self.x, self.factor, self.iters, self.thr, self.flags, self.n_out = x_ct, factor, int(max_iterations), float(threshold_value), flags, n_out
        self.t_out = x_ct.shape[1] * factor if n_out is None else int(n_out)
        self.L = native.lib()

    def empty(self, rows):
        return torch.empty((rows, self.t_out), dtype=torch.float32, device=self.x.device)

    def zero_peak(self):
        return torch.zeros(1, dtype=torch.float32, device=self.x.device)

    def run_local(self, lo, hi):
        xb = self.x[lo:hi].contiguous()
        plan = _plan(xb.shape[1], hi - lo, self.factor, xb.device.index or 0, 0, 0, None, self.n_out)
        out = self.empty(hi - lo)
        native.check(self.L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(xb), native.ptr(out), self.iters, self.thr,
                                                 self.flags | native.FL_DEFER_FINALIZE, native.stream_ptr()), "egr_fatllama_enhance")
        return plan, out

    def joint_peak(self, st):
        j = self.zero_peak()
        native.check(self.L.egr_fatllama_joint_peak(C.c_void_p(st[0]), self.flags, native.ptr(j), native.stream_ptr()), "egr_fatllama_joint_peak")
        return j

    def finalize(self, st, joint):
        native.check(self.L.egr_fatllama_finalize(C.c_void_p(st[0]), native.ptr(st[1]), self.flags, native.ptr(joint), native.stream_ptr()),
                     "egr_fatllama_finalize")
        return st[1]


def enhance_channel_parallel(x_ct: torch.Tensor, factor: int, max_iterations: int, threshold_value: float, normalize: bool, autoscale: bool,
                             pcm_in: bool, node_post: bool, group=None, gather: bool = True, variant=None, n_out=None):
    """enhance_device with the channels of x_ct [C, T] (replicated on every rank) spread over the ranks of `group` (SURVEY.md section
    8(e): at most C ranks work; one 4-byte all-reduce(MAX) before the joint normalise).  Bit-identical to enhance_device on one rank."""
    from . import shard
    if not (x_ct.is_cuda and x_ct.dtype == torch.float32 and x_ct.dim() == 2):
        raise RuntimeError("enhance_channel_parallel wants a [C,T] float32 tensor on the GPU")
    flags = ((native.FL_NORMALIZE if normalize else 0) | (native.FL_AUTOSCALE if autoscale else 0) |
             (native.FL_PCM_IN if pcm_in else 0) | (native.FL_NODE_POST if node_post else 0) | variant_flags(variant))
    be = _ChannelBlockBackend(x_ct.contiguous(), factor, max_iterations, threshold_value, flags, n_out)
    return shard.sharded_channels(be, x_ct.shape[0], group=group, gather=gather)


def enhance_devices(x_ct: torch.Tensor, factor: int, max_iterations: int, threshold_value: float, normalize: bool, autoscale: bool,
                    pcm_in: bool, node_post: bool, devs, variant=None, n_out=None) -> torch.Tensor:
    """Channel-parallel Fat-Llama inside ONE process (what a ComfyUI host is): position i of `devs` (the
    EGREGORA_DEVICES list; at most as many as there are channels work) takes a contiguous block of channels on a host thread bound to its
    device -- a peer copy of its channels in, egr_fatllama_enhance with EGR_FL_DEFER_FINALIZE on a plan of its own, its joint peak read
    back (4 bytes), the maximum over the blocks taken on the host, egr_fatllama_finalize with it, a peer copy of its output rows into the
    result on x_ct's device.  Same arithmetic as shard.sharded_channels across processes: bit-identical to enhance_device for even
    lengths.  UNMEASURED on two physical GPUs in this repository's test pool (EGREGORA_DEVICES=0,0 exercises it on one)."""
    import threading
    from . import shard
    if not (x_ct.is_cuda and x_ct.dtype == torch.float32 and x_ct.dim() == 2):
        raise RuntimeError("enhance_devices wants a [C,T] float32 tensor on the GPU")
    x_ct = x_ct.contiguous()
    Cn, T = x_ct.shape
    home = x_ct.device
    t_out = T * factor if n_out is None else int(n_out)
    flags = ((native.FL_NORMALIZE if normalize else 0) | (native.FL_AUTOSCALE if autoscale else 0) |
             (native.FL_PCM_IN if pcm_in else 0) | (native.FL_NODE_POST if node_post else 0) | variant_flags(variant))
    jobs = [(i, d, lo, hi) for i, (d, (lo, hi)) in enumerate(zip(devs, shard.block_bounds(Cn, min(len(devs), Cn)))) if hi > lo]
    out = torch.empty((Cn, t_out), dtype=torch.float32, device=home)
    torch.cuda.current_stream(home).synchronize()            # x_ct is complete before another device's stream reads it
    L = native.lib()
    peaks = [0.0] * len(jobs)
    gate = threading.Barrier(len(jobs))
    errors = []
    # the plans are made (or found) HERE, on the caller's thread, one per position of the list: the workers only run
    plans = []
    for (slot, d, lo, hi) in jobs:
        with torch.cuda.device(d):
            plans.append(_plan(T, hi - lo, factor, d, 0, 0, None, n_out, slot=slot + 1))
    if len(set(plans)) != len(plans) or any(pl not in _PLANS.values() for pl in plans):
        raise RuntimeError("enhance_devices: the plan cache is too small for one plan per device (raise fatllama_engine._MAX_PLANS)")

    def work(j, slot, d, lo, hi):
        try:
            with torch.cuda.device(d):
                dev = torch.device("cuda", d)
                xb = x_ct[lo:hi].to(dev, non_blocking=True).contiguous() if dev != home else x_ct[lo:hi].contiguous()
                plan = plans[j]
                y = torch.empty((hi - lo, t_out), dtype=torch.float32, device=dev)
                native.check(L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(xb), native.ptr(y), int(max_iterations), float(threshold_value),
                                                    flags | native.FL_DEFER_FINALIZE, native.stream_ptr()), "egr_fatllama_enhance")
                jp = torch.zeros(1, dtype=torch.float32, device=dev)
                native.check(L.egr_fatllama_joint_peak(C.c_void_p(plan), flags, native.ptr(jp), native.stream_ptr()), "egr_fatllama_joint_peak")
                peaks[j] = float(jp.item())                  # (synchronises this device's stream)
                gate.wait()                                  # every block's peak is in: the joint peak is their maximum (exact)
                jp.fill_(max(peaks))
                native.check(L.egr_fatllama_finalize(C.c_void_p(plan), native.ptr(y), flags, native.ptr(jp), native.stream_ptr()), "egr_fatllama_finalize")
                out[lo:hi].copy_(y, non_blocking=True)       # a peer copy into the home device's tensor
                torch.cuda.current_stream().synchronize()
        except BaseException as ex:      # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(ex)
            gate.abort()

    threads = [threading.Thread(target=work, args=(j,) + job, name=f"fatllama-dev{job[1]}") for j, job in enumerate(jobs)][1:]
    for t in threads:
        t.start()
    work(0, *jobs[0])
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def kernel_times(n_in: int, channels: int, factor: int, device: int = 0):
    L = native.lib()
    plan = _plan(n_in, channels, factor, device)
    r, c = C.c_double(), C.c_double()
    nr, nc = C.c_int64(), C.c_int64()
    native.check(L.egr_fatllama_kernel_times(C.c_void_p(plan), C.byref(r), C.byref(c), C.byref(nr), C.byref(nc)),
                 "egr_fatllama_kernel_times")
    L.egr_fatllama_set_profiling(C.c_void_p(plan), 0)
    return {"row_ms": r.value, "col_ms": c.value, "row_launches": nr.value, "col_launches": nc.value}


def kernel_times3(n_in: int, channels: int, factor: int, device: int = 0):
    """HIP-event averages of the last profiled run by pass: row pass, outer column pass, second column pass (packed plans: the
    inner pass of a three-level plan; chirp-z plans: k_pz_rowconv, k_pzpair, the crop pass)."""
    L = native.lib()
    plan = _plan(n_in, channels, factor, device)
    ms = (C.c_double * 3)()
    cnt = (C.c_int64 * 3)()
    native.check(L.egr_fatllama_kernel_times3(C.c_void_p(plan), ms, cnt), "egr_fatllama_kernel_times3")
    L.egr_fatllama_set_profiling(C.c_void_p(plan), 0)
    return {"ms": list(ms), "launches": list(cnt)}


def node_run(cs: torch.Tensor, sr: int, max_iterations: int, threshold_value: float, target_bitrate_kbps: int,
             toggle_normalize: bool, toggle_autoscale: bool) -> Tuple[torch.Tensor, int]:
    """Whole node arithmetic for one AUDIO: [C,T] float (any device) -> ([C,T*f] float32 CUDA, sr*f)."""
    native.require_device()
    x = cs.to("cuda", torch.float32, non_blocking=True).contiguous()
    f = upscale_factor(int(sr), x.shape[0], int(target_bitrate_kbps))
    n_out, sr_out = None, int(sr) * f
    if "ratio_then_int" in variant_names():          # SPEC.md factor_mode: the ratio is applied before int()
        r = upscale_ratio(int(sr), x.shape[0], int(target_bitrate_kbps))
        n_out, sr_out, f = int(x.shape[1] * r), int(int(sr) * r), 1
    from . import flashsr_engine
    devs = flashsr_engine.devices()
    if len(devs) > 1 and x.shape[0] >= 2:            # EGREGORA_DEVICES: one channel block per device, one 4-byte exchange
        y = enhance_devices(x, f, int(max_iterations), float(threshold_value), bool(toggle_normalize), bool(toggle_autoscale),
                            pcm_in=True, node_post=True, devs=devs, n_out=n_out)
    else:
        y = enhance_device(x, f, int(max_iterations), float(threshold_value), bool(toggle_normalize),
                           bool(toggle_autoscale), pcm_in=True, node_post=True, n_out=n_out)
    return y, sr_out
