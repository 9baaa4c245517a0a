"""Chunk-parallel sharding of the FlashSR hot loop across the GPUs of one node.

The reference processes chunks sequentially and independently (egregora_audio_super_resolution.py:411-418); they
only meet in WOLA (:420).  Rank r of the default process group takes a contiguous block -- sizes balanced to within one
chunk (130 chunks over 8 ranks: 17, 17, 16, 16, 16, 16, 16, 16) -- and ONE all-gather (RCCL over xGMI on GPUs, gloo in the
CPU tests) of the equal-padded prediction blocks gives every rank all predictions for the WOLA kernel.
No other collective exists on the FlashSR data path.

The Fat-Llama path shards over CHANNELS only (the reference hands upstream the whole file and every channel
is one whole-signal transform per iteration, egregora_fat_llama_gpu.py:272-288; C = 2 => at most 2 GPUs): `sharded_channels` gives
rank r a contiguous block of channels, and the channels meet in upstream's joint normalise alone -- ONE all-reduce(MAX) of a single
float (4 bytes).  Ranks beyond the channel count idle ("replicas only" past C GPUs).
"""
from typing import Callable, List, Tuple

import torch


def block_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    base, extra = divmod(max(n, 0), world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def sharded_chunks(run_block: Callable[[int, int], torch.Tensor], n: int, item_shape: Tuple[int, ...], device,
                   group=None) -> torch.Tensor:
    """run_block(lo, hi) -> [hi-lo, *item_shape]; returns [n, *item_shape] on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return run_block(0, n)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = block_bounds(n, world)
    per = -(-n // world)
    lo, hi = bounds[rank]
    local = torch.zeros((per,) + tuple(item_shape), dtype=torch.float32, device=device)
    if hi > lo:
        local[: hi - lo] = run_block(lo, hi)
    gathered = torch.empty((world * per,) + tuple(item_shape), dtype=torch.float32, device=device)
    try:
        dist.all_gather_into_tensor(gathered, local, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
        gathered = torch.cat(parts, 0)
    if world * per == n:
        return gathered
    return torch.cat([gathered[r * per: r * per + (b - a)] for r, (a, b) in enumerate(bounds) if b > a], 0)


def sharded_channels(backend, channels: int, group=None, gather: bool = True):
    """Channel-parallel Fat-Llama.  backend: .run_local(lo, hi) -> state (the loop on channels [lo, hi): out = y + d and per-channel
    peaks, nothing joint yet), .joint_peak(state) -> float32 tensor [1] (max over ITS channels of the peak after autoscale; zeros for
    an idle rank via .zero_peak()), .finalize(state, joint) -> [hi - lo, T] (autoscale / normalise by the JOINT peak / write patch /
    PCM_16), .empty(rows) -> [rows, T] of the output's dtype and device.  Returns [channels, T] on every rank (gather=True: one
    all-gather of the equal-padded blocks -- the output hand-over, not part of the arithmetic) or this rank's block."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        st = backend.run_local(0, channels)
        return backend.finalize(st, backend.joint_peak(st))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = block_bounds(channels, world)
    lo, hi = bounds[rank]
    st = backend.run_local(lo, hi) if hi > lo else None
    joint = backend.joint_peak(st) if st is not None else backend.zero_peak()
    host_hop = joint.is_cuda and dist.get_backend(group) == "gloo"          # (one-GPU test rigs: gloo collectives on host copies)
    if host_hop:
        jc = joint.cpu()
        dist.all_reduce(jc, op=dist.ReduceOp.MAX, group=group)
        joint.copy_(jc)
    else:
        dist.all_reduce(joint, op=dist.ReduceOp.MAX, group=group)          # the path's only exchange: 4 bytes
    y = backend.finalize(st, joint) if st is not None else backend.empty(0)
    if not gather:
        return y
    per = -(-channels // world)
    local = backend.empty(per)
    if hi > lo:
        local[: hi - lo] = y
    if host_hop:
        lc = local.cpu()
        parts = [torch.empty_like(lc) for _ in range(world)]
        dist.all_gather(parts, lc, group=group)
        parts = [t.to(local.device) for t in parts]
    else:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
    return torch.cat([parts[r][: b - a] for r, (a, b) in enumerate(bounds) if b > a], 0)
