"""Run-to-run determinism of every Fat-Llama plan kind: 30 repetitions of one configuration must be bit-identical.

A workgroup-local race (round 2: a reduction scratch of 8 slots under 16 waves overflowing into the tile) shows up as "a few
wrong samples once in 30 runs" and slips past a single oracle comparison; repeating each kind catches that class."""
import numpy as np
import pytest
import torch

from test_gpu_fatllama import synth

pytestmark = pytest.mark.gpu

REPS = 30

CONFIGS = [
    # (id, C, n, factor, iters, threshold, kwargs)
    ("packed-two-level", 2, 9600, 1, 5, 0.6, {}),
    ("packed-scheduled-c3-length", 2, 2880000, 1, 3, 0.6, {}),
    ("packed-three-level", 2, 48000, 1, 4, 0.6, {"split": (20, 24, 50)}),
    ("packed-three-level-two-barrier-kernels", 2, 5760000, 1, 2, 0.6, {"split": (625, 2, 2304)}),
    ("packed-two-barrier-rows-768", 2, 960000, 1, 3, 0.6, {}),                 # 625 x 768: k_row_wl<12, 8> + k_col_wl
    ("packed-two-barrier-rows-1920", 2, 2400000, 1, 3, 0.6, {}),               # 625 x 1920: k_row_wl<30, 8>
    ("packed-two-barrier-rows-4608", 1, 5760000, 1, 2, 0.6, {}),               # 625 x 4608: k_row_wl<32, 12>, 88 KB of LDS
    ("packed-441-columns-rows-3000", 2, 2646000, 1, 3, 0.6, {}),               # 441 x 3000: k_col_wl<21, 12> + k_row_wl<30, 10> (60 s at 44.1 kHz)
    ("packed-441-columns-odd-cross-radix", 2, 1323000, 1, 3, 0.6, {}),         # 441 x 1500: k_row_wl<15, 10>, the spare LDS block of the self-paired row
    ("packed-odd-cross-radix-even-row-count", 2, 2 * 320 * 960, 1, 3, 0.6, {"split": (320, 960, 1)}),   # k_row_wl<15, 8>, two self-paired rows
    ("packed-three-level-441", 1, 2 * 441 * 2 * 500, 1, 2, 0.6, {"split": (441, 2, 500)}),             # k_col_wl<21, 12>, inner 2, k_row_wl<5, 10>
    ("packed-relative-soft", 2, 9600, 1, 5, 0.02, {"variant": "relative,soft"}),
    ("chirpz-pairs-odd-centre", 2, 2 * 7919, 1, 4, 0.6, {}),
    ("chirpz-pairs-integer-centre", 2, 4 * 1013, 1, 4, 0.6, {}),
    ("chirpz-channel-pairs", 2, 4801, 1, 4, 0.6, {}),
    ("chirpz-three-channels", 3, 1001, 1, 4, 0.6, {}),
    ("chirpz-relative", 2, 4801, 1, 4, 0.02, {"variant": "relative"}),
    ("chirpz-scheduled", 2, 1000002, 1, 3, 0.6, {}),
    ("chirpz-scheduled-8192-rows", 1, 5898234, 1, 2, 0.6, {}),
    ("chirpz-legacy-wide-columns", 1, 2400001, 1, 2, 0.6, {"split": "bluestein"}),
    ("graph-replay-77-iterations", 2, 9600, 1, 77, 0.6, {}),
]


@pytest.mark.parametrize("name,C,n,f,iters,thr,kw", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_thirty_runs_are_bit_identical(pack, name, C, n, f, iters, thr, kw):
    from egregora_amd import fatllama_engine as fe
    x = torch.from_numpy(synth(C, n, seed=n + iters)).cuda()
    first = fe.enhance_device(x, f, iters, thr, True, False, True, True, **kw).clone()
    assert bool(torch.isfinite(first).all())
    for rep in range(1, REPS):
        y = fe.enhance_device(x, f, iters, thr, True, False, True, True, **kw)
        assert torch.equal(y, first), f"{name}: repetition {rep} differs in {int((y != first).sum())} samples"
