"""Chunk-parallel sharding of the FlashSR hot loop across the GPUs of one node.

The reference processes chunks sequentially and independently (egregora_audio_super_resolution.py:411-418); they
only meet in WOLA (:420).  Rank r of the default process group takes a contiguous block -- sizes balanced to within one
chunk (130 chunks over 8 ranks: 17, 17, 16, 16, 16, 16, 16, 16) -- and ONE all-gather (RCCL over xGMI on GPUs, gloo in the
CPU tests) of the equal-padded prediction blocks gives every rank all predictions for the WOLA kernel.
No other collective exists on the data path.
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
