"""Side streams that really run concurrently with the caller's stream.

HIP multiplexes its streams onto a few hardware queues (four by default) and two streams that land on one queue execute in
order; which pairs collide is not visible through the API and changes with every stream the process creates.  Fat-Llama runs
its two channel pipelines on two streams, so the second one is chosen among candidates that are checked
once with egr_streams_overlap_us -- two 300 us spin kernels take ~300 us together on different queues, ~600 us on one -- and the
verified set is cached per (device, current stream)."""
import ctypes as C

import torch

from . import native

_CACHE = {}


def _overlap(a, b) -> bool:
    us = C.c_double()
    best = 1e9
    for _ in range(2):
        native.check(native.lib().egr_streams_overlap_us(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), 300, C.byref(us)),
                     "egr_streams_overlap_us")
        best = min(best, us.value)
    return best < 450.0


def side_streams(n: int) -> list:
    """Up to n torch streams that overlap with the current stream and with each other (fewer when the runtime has no free queue)."""
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    have = _CACHE.get(key)
    if have is not None and (len(have) >= n or have[-1:] == [None]):
        return [s for s in have if s is not None][:n]
    chosen = [s for s in (have or []) if s is not None]
    for _ in range(12):
        if len(chosen) >= n:
            break
        s = torch.cuda.Stream()
        if _overlap(cur, s) and all(_overlap(o, s) for o in chosen):
            chosen.append(s)
    _CACHE[key] = chosen + ([None] if len(chosen) < n else [])      # None: the search was exhausted, do not repeat it
    return chosen[:n]
