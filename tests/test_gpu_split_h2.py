"""The two-term fp16 operand scheme of the split contractions (csrc/egr_nn_gemm_s3.hip scheme 1, egr_conv_h2 / egr_split2h_pack /
egr_absmax_rows; handle logic in csrc/egr_flashsr.cpp: one power-of-two scale per (tensor, batch row), derived ON THE DEVICE from
that row's own maximum -- no measuring call, no read-back, no re-run).

Operator level, against a float64 convolution: the fp16 kernels must be no further from float64 than 1.25x the f32-MFMA kernel
(the same gate the three-term bf16 kernels are held to in tests/test_gpu_flashsr.py::test_split3_conv_error_vs_float64), at input
magnitudes from 1e-4 to 1e4, for every tile shape, the 1-D input-stationary kernel, split-K and the z-streamed GEMMs; rows 60 and
100 dB below their neighbours keep fp32-grade accuracy (their scale is their own); a row's bits do not depend on the other rows.
Handle level: the output of egr_flashsr_infer is a function of (weights, input row, seed, row id): independent of what the handle
processed before, of the other rows of the call and of how rows are spread over handles (a world-2 shard); the call does not
block the host.
"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _drv(*a, **k):
    from tools.flashsr_pydriver import PyDriverEngine as cls
    return cls(*a, **k)


@pytest.fixture(scope="module")
def eng(pack):
    from egregora_amd import flashsr_arch as A
    cfg = A.tiny_config()
    return _drv(cfg, A.init_params(cfg, 0)), cfg


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def scale_for(amax, e):
    """csrc/egr_flashsr.cpp h2_scale_for: the power of two that brings amax into (2^(e-1), 2^e]."""
    return 2.0 ** (e - math.ceil(math.log2(amax)))


RA = 32          # include/egregora_amd.h EGR_ROW_AMAX_STRIDE: the maximum of batch row r sits at float r * 32 (one 128-byte line per row)


def ra_zeros(rows):
    return torch.zeros(rows * RA, device="cuda")


def row_amax(e, x, rows, nz=1, zx=0):
    """egr_absmax_rows: per-batch-row max |x| on the device (what the handle computes when a tensor first feeds a split contraction)."""
    from egregora_amd import native
    ra = ra_zeros(rows)
    per_row = x.numel() // (rows * nz)
    native.check(e.L.egr_absmax_rows(p(x), rows, per_row, nz, zx, p(ra), e._st()), "absmax_rows")
    return ra


def h2_pack(e, wp, Co):
    from egregora_amd import native
    L = e.L
    ns = wp.shape[0]
    slot = torch.zeros(1, device="cuda")
    native.check(L.egr_absmax(p(wp), wp.numel(), p(slot), e._st()), "absmax")
    wmax = float(slot.item())
    assert wmax == float(wp.abs().max())
    ws = scale_for(wmax, 13)
    w2 = torch.empty(ns * 2 * Co * 16, dtype=torch.float16, device="cuda")
    native.check(L.egr_split2h_pack(p(wp), p(w2), ns, Co, ws, e._st()), "split2h")
    return w2, ws


CASES = [(2, 16, 12, 128, 128, 3), (1, 9, 7, 256, 96, 3), (3, 8, 8, 512, 40, 1), (2, 10, 6, 64, 200, 3),
         (9, 128, 128, 32, 128, 3),      # M = 147456: 256-row tiles
         (2, 8, 4, 640, 640, 3),         # small M, long K: split-K
         (1, 5, 3, 16, 20, 3),
         (1, 24, 16, 64, 512, 3),        # 128 x 256 tile
         (2, 8, 8, 48, 256, 1),
         (2, 6, 6, 32, 30, 3),           # Cout % 4 != 0: element stores
         (1, 40, 40, 128, 512, 3)]       # ragged M on the 128 x 256 tile


@pytest.mark.parametrize("mag", [1.0, 1e-4, 1e4])
def test_h2_conv_error_vs_float64(eng, mag):
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(78)
    for (B, H, W, Ci, Co, k) in CASES:
        x = (mag * torch.randn(B, H, W, Ci, generator=g)).cuda()
        w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).cuda()
        b = (mag * torch.randn(Co, generator=g)).cuda()
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
        wp = e.pack_matrix(w.permute(2, 3, 1, 0).reshape(k * k * Ci, Co).contiguous()).cuda()
        ns = wp.shape[0]
        w2, ws = h2_pack(e, wp, Co)
        # the two planes sum back to the scaled weights to 2^-22 of each element (or 2^-25 absolute in scaled units)
        back = w2.view(ns, 2, Co, 16).double().sum(1) / ws
        assert float((back - wp.double()).abs().max()) <= 2.0 ** -21 * float(wp.abs().max())
        y1 = torch.empty(B, H, W, Co, device="cuda")
        y2 = torch.empty_like(y1)
        ra = row_amax(e, x, B)
        assert torch.equal(ra[::RA].cpu(), x.abs().amax(dim=(1, 2, 3)).cpu()), "per-row maxima are exact"
        native.check(L.egr_conv_nhwc(p(x), p(wp), p(b), p(None), p(None), p(y1), B, H, W, Ci, H, W, Co, k, k, 1, 1, k // 2, k // 2,
                                     0, 0, 0.0, e._st()), "conv")
        native.check(L.egr_conv_h2(p(x), p(w2), p(b), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, k // 2, k // 2,
                                   0, 0, 0.0, 1, 1, 0, 0, H, W, 1, 0, 0, 0, ws, p(ra), B, p(None), e._st()), "conv_h2")
        mx = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
        rms = lambda y: float((y.double() - ref).norm() / ref.norm())
        assert mx(y2) <= 1.25 * mx(y1) + 1e-8 and rms(y2) <= 1.25 * rms(y1) + 1e-8, ((B, H, W, Ci, Co, k), mx(y1), mx(y2), rms(y1), rms(y2))
        assert mx(y2) < 2e-6, mx(y2)


def test_h2_quiet_rows_keep_their_precision_and_rows_are_independent(eng):
    """Rows 60 and 100 dB below their neighbour (a quiet passage batched next to a loud one) come out at the SAME relative accuracy
    as the loud row: every batch row is scaled from its own maximum.  And the bits of a row do not depend on the other rows."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(5)
    B, H, W, Ci, Co, k = 2, 8, 32, 128, 128, 3
    x = torch.randn(B, H, W, Ci, generator=g)
    x[1] *= 1e-3
    x = x.cuda()
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).cuda()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, padding=1).permute(0, 2, 3, 1)
    wp = e.pack_matrix(w.permute(2, 3, 1, 0).reshape(k * k * Ci, Co).contiguous()).cuda()
    w2, ws = h2_pack(e, wp, Co)
    y1 = torch.empty(B, H, W, Co, device="cuda")
    y2 = torch.empty_like(y1)
    native.check(L.egr_conv_nhwc(p(x), p(wp), p(None), p(None), p(None), p(y1), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, e._st()), "conv")
    native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, 1, 1, 0, 0, H, W,
                               1, 0, 0, 0, ws, p(row_amax(e, x, B)), B, p(None), e._st()), "conv_h2")
    y2_first = y2.clone()
    for b in range(B):
        r1 = float((y1[b].double() - ref[b]).norm() / ref[b].norm())
        r2 = float((y2[b].double() - ref[b]).norm() / ref[b].norm())
        assert r2 <= 1.25 * r1 + 1e-8, (b, r1, r2)
    # 100 dB below its neighbour (a per-TENSOR scale would leave this row's second terms fp16 subnormals: 1.7e-6 in round 3)
    x[1] *= 1e-2
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, padding=1).permute(0, 2, 3, 1)
    native.check(L.egr_conv_nhwc(p(x), p(wp), p(None), p(None), p(None), p(y1), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, e._st()), "conv")
    native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, 1, 1, 0, 0, H, W,
                               1, 0, 0, 0, ws, p(row_amax(e, x, B)), B, p(None), e._st()), "conv_h2")
    r1 = float((y1[1].double() - ref[1]).norm() / ref[1].norm())
    r2 = float((y2[1].double() - ref[1]).norm() / ref[1].norm())
    print(f"row at 1e-5 of its neighbour: relative error {r2:.2e} (f32 MFMA kernel {r1:.2e})")
    assert r2 <= 1.25 * r1 + 1e-8, (r1, r2)
    assert torch.equal(y2[0], y2_first[0]), "row 0 does not see what row 1 holds"
    # a silent row: exact zeros out (no bias), its neighbour untouched
    x[1] = 0.0
    native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, 1, 1, 0, 0, H, W,
                               1, 0, 0, 0, ws, p(row_amax(e, x, B)), B, p(None), e._st()), "conv_h2")
    assert float(y2[1].abs().max()) == 0.0 and torch.equal(y2[0], y2_first[0])
    # the range end: a row at 1e30 and one at 1e-30 in one launch -- no inf / nan, both at fp32-grade accuracy
    x2 = torch.randn(B, H, W, Ci, generator=g)
    x2[0] *= 1e30
    x2[1] *= 1e-30
    x2 = x2.cuda()
    ref = F.conv2d(x2.double().permute(0, 3, 1, 2), w.double(), None, padding=1).permute(0, 2, 3, 1)
    native.check(L.egr_conv_h2(p(x2), p(w2), p(None), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, 1, 1, 0, 0, H, W,
                               1, 0, 0, 0, ws, p(row_amax(e, x2, B)), B, p(None), e._st()), "conv_h2")
    assert bool(torch.isfinite(y2).all())
    # (1e-30 sits below the 2^-63 floor of the scale's exponent range: such a row keeps fewer bits, see egr_conv.h h2_row_scale)
    assert float((y2[0].double() - ref[0]).norm() / ref[0].norm()) < 1e-6
    assert float((y2[1].double() - ref[1]).norm() / ref[1].norm()) < 1e-3


def test_h2_conv1d_and_zstream(eng):
    """The input-stationary 1-D kernel (vocoder shapes) and the z-streamed GEMMs (Winograd shapes) in the fp16 scheme against the
    same launches on the bf16 terms, both against float64: rms error <= 1.25x, maximum error <= 1.5x the bf16 kernels'."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(11)
    for (B, Wd, Ci, Co, k, dil) in [(3, 512, 64, 64, 7, 3), (2, 1024, 32, 32, 11, 1), (2, 256, 128, 128, 3, 5), (4, 384, 16, 32, 3, 1)]:
        x = torch.randn(B, 1, Wd, Ci, generator=g).cuda()
        w = (torch.randn(Co, Ci, 1, k, generator=g) / math.sqrt(Ci * k)).cuda()
        wp = e.pack_matrix(w.permute(2, 3, 1, 0).reshape(k * Ci, Co).contiguous()).cuda()
        ns = wp.shape[0]
        w3 = torch.empty(ns * 3 * Co * 16, dtype=torch.bfloat16, device="cuda")
        native.check(L.egr_split3_pack(p(wp), p(w3), ns, Co, e._st()), "split3")
        w2, ws = h2_pack(e, wp, Co)
        pad = dil * (k - 1) // 2
        y3 = torch.empty(B, 1, Wd, Co, device="cuda")
        y2 = torch.empty_like(y3)
        native.check(L.egr_conv_s3(p(x), p(w3), p(None), p(None), p(None), p(y3), B, 1, Wd, Ci, 1, Wd, Co, 1, k, 1, dil, 0, pad, 0, 0, 0.0, 1, 1, 0, 0,
                                   1, Wd, 1, 0, 0, 0, e._st()), "conv_s3")
        native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y2), B, 1, Wd, Ci, 1, Wd, Co, 1, k, 1, dil, 0, pad, 0, 0, 0.0, 1, 1, 0, 0,
                                   1, Wd, 1, 0, 0, 0, ws, p(row_amax(e, x, B)), B, p(None), e._st()), "conv_h2")
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, padding=(0, pad), dilation=(1, dil)).permute(0, 2, 3, 1)
        r3 = float((y3.double() - ref).norm() / ref.norm())
        r2 = float((y2.double() - ref).norm() / ref.norm())
        m3 = float((y3.double() - ref).abs().max() / ref.abs().max())
        m2 = float((y2.double() - ref).abs().max() / ref.abs().max())
        assert r2 <= 1.25 * r3 + 1e-8 and m2 <= 1.5 * m3 + 1e-8 and m2 < 2e-6, (Wd, Ci, Co, k, dil, r3, r2, m3, m2)
    # z-stacked GEMMs: nz problems [P][Cin] x [Cin][Cout]
    # (P rows = `rows` batch rows of P / rows tiles each, deliberately not multiples of the 128-row block tile, at different levels)
    for (nz, P, Ci, Co, rows) in [(36, 700, 128, 128, 7), (36, 300, 256, 256, 3), (16, 500, 64, 128, 5)]:
        x = torch.randn(nz, P, Ci, generator=g)
        lv = 10.0 ** torch.linspace(0, -4, rows)
        x = (x.view(nz, rows, P // rows, Ci) * lv.view(1, rows, 1, 1)).reshape(nz, P, Ci).cuda()
        w = (torch.randn(nz, Ci, Co, generator=g) / math.sqrt(Ci)).cuda()
        wp = torch.stack([e.pack_matrix(w[z].cpu()) for z in range(nz)]).cuda().contiguous()       # [nz][ns][Co][16]
        ns = wp.shape[1]
        w3 = torch.empty(nz * ns * 3 * Co * 16, dtype=torch.bfloat16, device="cuda")
        native.check(L.egr_split3_pack(p(wp), p(w3), nz * ns, Co, e._st()), "split3")
        w2, ws = h2_pack(e, wp.view(nz * ns, Co, 16), Co)
        y3 = torch.empty(nz, P, Co, device="cuda")
        y2 = torch.empty_like(y3)
        zf = ns * Co * 16
        native.check(L.egr_conv_s3(p(x), p(w3), p(None), p(None), p(None), p(y3), P, 1, 1, Ci, 1, 1, Co, 1, 1, 1, 1, 0, 0, 0, 0, 0.0, 1, 1, 0, 0, 1, 1,
                                   nz, P * Ci, zf * 3 // 8, P * Co, e._st()), "conv_s3")
        native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y2), P, 1, 1, Ci, 1, 1, Co, 1, 1, 1, 1, 0, 0, 0, 0, 0.0, 1, 1, 0, 0, 1, 1,
                                   nz, P * Ci, zf * 2 // 8, P * Co, ws, p(row_amax(e, x, rows, nz, P * Ci)), rows, p(None), e._st()), "conv_h2")
        ref = torch.einsum("zpc,zcn->zpn", x.double(), w.double())
        for r in range(rows):                        # every batch row at its own level
            sl = slice(r * (P // rows), (r + 1) * (P // rows))
            r3 = float((y3[:, sl].double() - ref[:, sl]).norm() / ref[:, sl].norm())
            r2 = float((y2[:, sl].double() - ref[:, sl]).norm() / ref[:, sl].norm())
            assert r2 <= 1.25 * r3 + 1e-8, (nz, P, Ci, Co, r, r3, r2)


def test_winograd4_input_leaves_the_row_maxima_of_V(eng):
    """egr_winograd4_input_ra: the F(4x4) input transform writes max |V| per image while it writes V (the operand maxima of the
    36 GEMMs that read V) -- equal to a separate egr_absmax_rows pass over V, with and without the fused GroupNorm + SiLU."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(17)
    for (B, H, W, Cc, gn) in [(3, 16, 8, 128, False), (2, 8, 12, 64, True), (5, 4, 4, 32, True), (1, 64, 32, 128, False)]:
        x = torch.randn(B, H, W, Cc, generator=g)
        x = (x * (10.0 ** -torch.arange(B).float()).view(B, 1, 1, 1)).cuda()
        P = B * (H // 4) * (W // 4)
        V1 = torch.empty(36, P, Cc, device="cuda")
        V2 = torch.empty_like(V1)
        sc = (1.0 + 0.1 * torch.randn(B, Cc, generator=g)).cuda() if gn else None
        sh = (0.1 * torch.randn(B, Cc, generator=g)).cuda() if gn else None
        ra = ra_zeros(B)
        native.check(L.egr_winograd4_input(p(x), p(sc), p(sh), 1 if gn else 0, B, H, W, Cc, p(V1), e._st()), "wino4_in")
        native.check(L.egr_winograd4_input_ra(p(x), p(sc), p(sh), 1 if gn else 0, B, H, W, Cc, p(V2), p(ra), e._st()), "wino4_in_ra")
        assert torch.equal(V1, V2)
        want = V1.view(36, B, -1).abs().amax(dim=(0, 2))
        assert torch.equal(ra[::RA], want), (ra, want)
        assert torch.equal(row_amax(e, V1, B, 36, P * Cc)[::RA], want)


def _engines(n, split="f16x2"):
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    old = E.FlashSREngine.SPLIT
    try:
        E.FlashSREngine.SPLIT = split
        return cfg, [E.FlashSREngine(cfg, P) for _ in range(n)]
    finally:
        E.FlashSREngine.SPLIT = old


def test_infer_is_a_pure_function_of_its_rows(pack):
    """egr_flashsr_infer on the fp16 operand terms (the default): the bits of output row r depend on (weights, x[r], seed, id of r)
    only -- not on what the handle processed before (loud input, silence, other seeds), not on a fresh handle vs a used one, and
    (same row count per forward) not on the other rows of the call.  The first call of a handle IS the steady state."""
    cfg, (ea, eb) = _engines(2)
    assert ea.split_info()["enabled"] and ea.split_info()["weights"] > 10
    g = torch.Generator().manual_seed(9)
    x = (0.05 * torch.randn(4, cfg.chunk, generator=g)).cuda()
    ids = torch.tensor([7, 3, 11, 5], dtype=torch.int64, device="cuda")
    y_fresh = ea.c_infer(x, ids, 4)                                   # first call of handle A
    # handle B goes through a history first: 300x louder, silence, 100 dB quieter, another seed, another row count
    eb.c_infer(300.0 * x, ids, 4)
    eb.c_infer(torch.zeros_like(x), ids, 9)
    eb.c_infer(1e-5 * x, None, 1)
    eb.c_infer(x[:3].contiguous(), None, 2)
    y_used = eb.c_infer(x, ids, 4)
    assert torch.equal(y_fresh, y_used), "history-independent"
    assert torch.equal(ea.c_infer(x, ids, 4), y_fresh), "repeatable"
    # the other rows of the call do not matter (same row count, so the same tiles): replace rows 1..3, row 0 keeps its bits
    x2 = x.clone()
    x2[1] *= 1e-4
    x2[2] = 0.0
    x2[3] *= 50.0
    y2 = ea.c_infer(x2, ids, 4)
    assert torch.equal(y2[0], y_fresh[0]) and not torch.equal(y2[1], y_fresh[1])
    # against the bf16-term handle: fp32 round-off apart (different kernels, same function)
    cfg, (ec,) = _engines(1, "bf16x3")
    yb = ec.c_infer(x, ids, 4)
    rel = float((y_fresh - yb).double().norm() / yb.double().norm())
    assert 0.0 < rel < 2e-5, rel
    assert not ec.split_info()["enabled"]
    # switching a handle between the schemes changes nothing but the scheme
    ea.set_split("bf16x3")
    assert torch.equal(ea.c_infer(x, ids, 4), yb)
    ea.set_split("f16x2")
    assert torch.equal(ea.c_infer(x, ids, 4), y_fresh)
    for e_ in (ea, eb, ec):
        e_.close()


def test_pool_of_row_maxima_grows_instead_of_failing_the_call(pack, monkeypatch):
    """the per-forward pool of row-maximum slices is sized by a heuristic over the layer table; a graph that measures
    more tensors than it allowed for used to fail egr_flashsr_infer with EGR_ERR_ALLOC.  With the pool started at TWO slices
    (EGR_FSR_RS_POOL_SLICES, a test knob) the call grows it in mid-forward -- several times -- and returns the same bits."""
    cfg, (ea, eb) = _engines(2)
    x = (0.05 * torch.randn(3, cfg.chunk, generator=torch.Generator().manual_seed(12))).cuda()
    want = ea.c_infer(x, None, 4)
    monkeypatch.setenv("EGR_FSR_RS_POOL_SLICES", "2")
    got = eb.c_infer(x, None, 4)
    assert torch.equal(got, want)
    assert torch.equal(eb.c_infer(x, None, 4), want)        # the next forward starts at the grown size (retired pools are freed there)
    monkeypatch.delenv("EGR_FSR_RS_POOL_SLICES")
    assert torch.equal(eb.c_infer(x, None, 4), want)
    for e_ in (ea, eb):
        e_.close()


def test_warmup_changes_nothing_but_the_first_call_cost(pack):
    """egr_flashsr_warmup (one throw-away pass of silence when the engine is built): the next call's bits are those of a handle that was
    never warmed, the host's call counter does not see it, and the scratch it sized is reported."""
    from egregora_amd import native
    cfg, (ea, eb) = _engines(2)
    x = (0.05 * torch.randn(3, cfg.chunk, generator=torch.Generator().manual_seed(13))).cuda()
    eb.warmup(2)
    assert eb.split_info()["calls"] == 0
    assert native.lib().egr_flashsr_scratch_bytes(eb.handle) >= native.lib().egr_flashsr_scratch_bytes(ea.handle) > 0      # (creation already uses the arena)
    assert torch.equal(eb.c_infer(x, None, 4), ea.c_infer(x, None, 4))
    assert eb.split_info()["calls"] == 1
    for e_ in (ea, eb):
        e_.close()


def test_rows_sharded_over_two_handles_are_bit_identical(pack):
    """A world-2 shard in one process: handle A takes rows 0..6, handle B rows 7..13 (each rank of shard.sharded_chunks owns its
    own handle); together they reproduce the 14-row call of ONE handle -- which runs the same two 7-row forwards as concurrent row
    groups -- bit for bit, with implicit and explicit ids."""
    import ctypes as C
    from egregora_amd import native
    cfg, (e1, ea, eb) = _engines(3)
    native.check(e1.L.egr_flashsr_set_streams(C.c_void_p(e1.handle), 2, 7), "set_streams")
    x = (0.1 * torch.randn(14, cfg.chunk, generator=torch.Generator().manual_seed(21)))
    x[3] *= 1e-3                                                       # a quiet row and a silent one among them
    x[9] = 0.0
    x = x.cuda()
    ids = torch.arange(14, dtype=torch.int64, device="cuda")
    y1 = e1.c_infer(x, None, 5)
    assert torch.equal(y1, e1.c_infer(x, ids, 5))
    ya = ea.c_infer(x[:7].contiguous(), ids[:7].contiguous(), 5)
    yb = eb.c_infer(x[7:].contiguous(), ids[7:].contiguous(), 5)
    assert torch.equal(torch.cat([ya, yb]), y1)
    for e_ in (e1, ea, eb):
        e_.close()


def test_infer_does_not_block_the_host(pack):
    """work is enqueued on the caller's stream.  After a warm-up call (scratch allocation, side-stream check) a call
    returns while its kernels are still queued behind a 200 ms spin kernel on the same stream."""
    import time
    cfg, (e,) = _engines(1)
    x = (0.1 * torch.randn(14, cfg.chunk, generator=torch.Generator().manual_seed(2))).cuda()
    y0 = e.c_infer(x, None, 1)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    torch.cuda._sleep(int(0.2 * 2.1e9))                                # ~200 ms of GPU time ahead of the call on its stream
    t0 = time.perf_counter()
    y1 = e.c_infer(x, None, 1)
    t_call = time.perf_counter() - t0
    done_at_return = st.query()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"egr_flashsr_infer returned after {1e3 * t_call:.1f} ms; stream idle at return: {done_at_return}; everything done after {1e3 * t_all:.1f} ms")
    assert not done_at_return and t_call < 0.5 * t_all, (t_call, t_all)
    assert torch.equal(y0, y1)
    e.close()


def test_fp16_calls_repeat_bit_for_bit_with_concurrent_row_groups(pack):
    """The bug class a flaky test found in round 2 (a rare race in shared scratch): 14 rows = two concurrent 7-row groups on side
    streams, fp16 operand terms with the device-side row maxima (atomics), 20 repetitions -- every repetition must give the same
    bits."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E, native
    cfg = A.tiny_config()
    e = E.FlashSREngine(cfg, A.init_params(cfg, 0))
    native.check(e.L.egr_flashsr_set_streams(C.c_void_p(e.handle), 2, 6), "set_streams")
    x = (0.1 * torch.randn(14, cfg.chunk, generator=torch.Generator().manual_seed(21))).cuda()
    ref = e.c_infer(x, None, 5)
    for _ in range(20):
        assert torch.equal(e.c_infer(x, None, 5), ref)
    assert e.split_info()["calls"] == 21
    e.close()


def test_snake_and_winograd_output_leave_the_row_maxima(eng):
    """egr_snake_aa_ra / egr_winograd4_output_ra: the producers of most split-contraction inputs write max |y| per batch row while
    they write y -- bit-equal outputs and maxima equal to a separate pass."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native, flashsr_arch as A
    g = torch.Generator().manual_seed(23)
    filt = torch.from_numpy(A.kaiser_sinc_filter(12)).float().cuda()
    for (B, Ln, Cc) in [(3, 256, 32), (2, 1000, 16), (5, 48, 64), (1, 17, 8)]:
        x = torch.randn(B, Ln, Cc, generator=g)
        x = (x * (10.0 ** -torch.arange(B).float()).view(B, 1, 1)).cuda()
        al = (0.1 * torch.randn(Cc, generator=g)).cuda()
        be = (0.1 * torch.randn(Cc, generator=g)).cuda()
        y1 = torch.empty_like(x)
        y2 = torch.empty_like(x)
        ra = ra_zeros(B)
        native.check(L.egr_snake_aa(p(x), p(al), p(be), p(filt), p(y1), B, Ln, Cc, 12, e._st()), "snake")
        native.check(L.egr_snake_aa_ra(p(x), p(al), p(be), p(filt), p(y2), B, Ln, Cc, 12, p(ra), e._st()), "snake_ra")
        assert torch.equal(y1, y2) and torch.equal(ra[::RA], y1.abs().amax(dim=(1, 2))), (B, Ln, Cc)
    for (B, H, W, N, res, act, stats) in [(3, 8, 8, 128, True, 0, True), (2, 16, 4, 64, False, 1, False), (5, 4, 4, 32, True, 1, True)]:
        P = B * (H // 4) * (W // 4)
        M = torch.randn(36, P, N, generator=g)
        M = (M.view(36, B, -1) * (10.0 ** -torch.arange(B).float()).view(1, B, 1)).reshape(36, P, N).cuda()
        bias = torch.randn(N, generator=g).cuda() * 1e-6
        r = (1e-6 * torch.randn(B, H, W, N, generator=g)).cuda() if res else None
        y1 = torch.empty(B, H, W, N, device="cuda")
        y2 = torch.empty_like(y1)
        part = torch.empty(P, N // 4, 2, device="cuda") if stats else None
        ra = ra_zeros(B)
        native.check(L.egr_winograd4_output(p(M), p(bias), p(r), p(y1), B, H, W, N, act, e._st()), "wino4_out")
        native.check(L.egr_winograd4_output_ra(p(M), p(bias), p(r), p(y2), B, H, W, N, act, p(part), p(ra), e._st()), "wino4_out_ra")
        assert torch.equal(y1, y2) and torch.equal(ra[::RA], y1.abs().amax(dim=(1, 2, 3))), (B, H, W, N)


def test_elementwise_producers_leave_the_row_maxima(eng):
    """The `_ra` forms of GroupNorm-apply, LayerNorm, the element-wise ops, GEGLU and the channel concat: outputs bit-equal to the plain
    forms, maxima equal to max |y| per batch row."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(31)
    B = 3
    lv = (10.0 ** -torch.arange(B).float())
    # GroupNorm (+SiLU)
    HW, Cc, G = 96, 64, 8
    x = (torch.randn(B, HW, Cc, generator=g) * lv.view(B, 1, 1)).cuda()
    gam, bet = torch.randn(Cc, generator=g).cuda(), torch.randn(Cc, generator=g).cuda()
    ws = torch.empty(L.egr_groupnorm_workspace_bytes(B, Cc, G) // 4 + 64, device="cuda")
    y1, y2, ra = torch.empty_like(x), torch.empty_like(x), ra_zeros(B)
    native.check(L.egr_groupnorm_nhwc(p(x), p(gam), p(bet), p(y1), B, HW, Cc, G, 1e-5, 1, p(ws), e._st()), "gn")
    native.check(L.egr_groupnorm_nhwc_ra(p(x), p(gam), p(bet), p(y2), B, HW, Cc, G, 1e-5, 1, p(ws), p(ra), e._st()), "gn_ra")
    assert torch.equal(y1, y2) and torch.equal(ra[::RA], y1.abs().amax(dim=(1, 2)))
    # LayerNorm over token rows
    T, Cc = 50, 96
    x = (torch.randn(B * T, Cc, generator=g) * lv.repeat_interleave(T).view(-1, 1)).cuda()
    gam, bet = torch.randn(Cc, generator=g).cuda(), (torch.randn(Cc, generator=g) * lv[1]).cuda()
    y1, y2, ra = torch.empty_like(x), torch.empty_like(x), ra_zeros(B)
    native.check(L.egr_layernorm_rows(p(x), p(gam), p(bet), p(y1), B * T, Cc, 1e-5, e._st()), "ln")
    native.check(L.egr_layernorm_rows_ra(p(x), p(gam), p(bet), p(y2), B * T, Cc, 1e-5, B, p(ra), e._st()), "ln_ra")
    assert torch.equal(y1, y2) and torch.equal(ra[::RA], y1.view(B, -1).abs().amax(dim=1))
    ref = F.layer_norm(x, (Cc,), gam, bet, 1e-5)
    assert float((y1 - ref).abs().max()) < 1e-5
    # element-wise ops
    n = 7 * 33
    a = (torch.randn(B, n, generator=g) * lv.view(B, 1)).cuda()
    b = (torch.randn(B, n, generator=g) * lv.view(B, 1)).cuda()
    for op, s0, s1, want in [(0, 0.0, 0.0, a + b), (1, 0.5, -2.0, 0.5 * a + -2.0 * b), (3, 0.25, 0.0, 0.25 * a), (5, 1.0 / 3.0, 0.0, (1.0 / 3.0) * (a + b))]:
        y1, y2, ra = torch.empty_like(a), torch.empty_like(a), ra_zeros(B)
        native.check(L.egr_eltwise(p(a), p(b), p(y1), a.numel(), op, s0, s1, e._st()), "ew")
        native.check(L.egr_eltwise_ra(p(a), p(b), p(y2), a.numel(), op, s0, s1, B, p(ra), e._st()), "ew_ra")
        assert torch.equal(y1, y2) and torch.equal(ra[::RA], y1.abs().amax(dim=1)), op
        assert float((y1 - want).abs().max()) <= 1e-6 * float(want.abs().max()), op
    # GEGLU
    D = 40
    u = (torch.randn(B * T, 2 * D, generator=g) * lv.repeat_interleave(T).view(-1, 1)).cuda()
    y1, y2, ra = torch.empty(B * T, D, device="cuda"), torch.empty(B * T, D, device="cuda"), ra_zeros(B)
    native.check(L.egr_geglu(p(u), p(y1), B * T, D, e._st()), "geglu")
    native.check(L.egr_geglu_ra(p(u), p(y2), B * T, D, B, p(ra), e._st()), "geglu_ra")
    assert torch.equal(y1, y2) and torch.equal(ra[::RA], y1.view(B, -1).abs().amax(dim=1))
    assert float((y1 - u[:, :D] * F.gelu(u[:, D:])).abs().max()) < 1e-5
    # channel concat
    M, C1, C2 = 30, 16, 24
    a = (torch.randn(B * M, C1, generator=g) * lv.repeat_interleave(M).view(-1, 1)).cuda()
    b = (torch.randn(B * M, C2, generator=g) * lv.repeat_interleave(M).view(-1, 1)).cuda()
    y1, y2, ra = torch.empty(B * M, C1 + C2, device="cuda"), torch.empty(B * M, C1 + C2, device="cuda"), ra_zeros(B)
    native.check(L.egr_concat_channels(p(a), p(b), p(y1), B * M, C1, C2, e._st()), "cat")
    native.check(L.egr_concat_channels_ra(p(a), p(b), p(y2), B * M, C1, C2, B, p(ra), e._st()), "cat_ra")
    assert torch.equal(y1, torch.cat([a, b], 1)) and torch.equal(y1, y2) and torch.equal(ra[::RA], y1.view(B, -1).abs().amax(dim=1))


def test_conv_h2_epilogue_leaves_the_row_maxima_of_its_output(eng):
    """egr_conv_h2's out_amax: max |y| per batch row after bias / residual / activation, on every epilogue (transposed 128-row
    tiles, 256-row tiles, the 1-D kernel, split-K through the reduction kernel, strided placement)."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(41)
    for (B, H, W, Ci, Co, k, act) in [(3, 16, 12, 128, 128, 3, 0), (9, 128, 128, 32, 128, 3, 1), (2, 8, 4, 640, 640, 3, 0), (2, 6, 6, 32, 30, 3, 3),
                                      (3, 1, 512, 64, 64, 7, 0), (1, 40, 40, 128, 512, 3, 0)]:
        x = (torch.randn(B, H, W, Ci, generator=g) * (10.0 ** -torch.arange(B).float()).view(B, 1, 1, 1)).cuda()
        kh = 1 if H == 1 else k
        w = (torch.randn(Co, Ci, kh, k, generator=g) / math.sqrt(Ci * kh * k)).cuda()
        b = (1e-3 * torch.randn(Co, generator=g)).cuda()
        r = (1e-3 * torch.randn(B, H, W, Co, generator=g)).cuda()
        wp = e.pack_matrix(w.permute(2, 3, 1, 0).reshape(kh * k * Ci, Co).contiguous()).cuda()
        w2, ws = h2_pack(e, wp, Co)
        y = torch.empty(B, H, W, Co, device="cuda")
        oa = ra_zeros(B)
        native.check(L.egr_conv_h2(p(x), p(w2), p(b), p(None), p(r), p(y), B, H, W, Ci, H, W, Co, kh, k, 1, 1, kh // 2, k // 2, 0, act, 0.0,
                                   1, 1, 0, 0, H, W, 1, 0, 0, 0, ws, p(row_amax(e, x, B)), B, p(oa), e._st()), "conv_h2")
        assert torch.equal(oa[::RA], y.abs().amax(dim=(1, 2, 3))), (B, H, W, Ci, Co, k, oa, y.abs().amax(dim=(1, 2, 3)))


def test_input_stationary_3x3_with_fused_groupnorm_vs_float64(eng):
    """egr_conv_h2_gn (k_conv3x3_is): 3x3 convolution of silu(x * scale[b][c] + shift[b][c]) with zero padding AFTER the
    normalisation, bias, residual, per-row output maxima -- against the float64 convolution of the same normalised tensor; the
    operand scale comes from egr_gn_operand_bound (an upper bound of the normalised row, from the coefficients and the row maxima
    of x), and the error must be no larger than 1.25x the f32-MFMA kernel's on the materialised normalised tensor.  Images at
    different levels and with a large mean (where the bound is loosest) in one launch; a shape that does not qualify is refused."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(51)
    for (B, H, W, Ci, Co, silu, res_on) in [(3, 128, 256, 64, 128, 1, True), (2, 256, 256, 128, 64, 0, False), (5, 64, 256, 32, 128, 1, False)]:
        x = torch.randn(B, H, W, Ci, generator=g)
        x = x * (10.0 ** -torch.arange(B).float()).view(B, 1, 1, 1) + 5.0 * (10.0 ** -torch.arange(B).float()).view(B, 1, 1, 1)    # mean 5 sigma
        sc = (1.0 + 0.2 * torch.randn(B, Ci, generator=g)) * (10.0 ** torch.arange(B).float()).view(B, 1)
        sh = 0.3 * torch.randn(B, Ci, generator=g) - 5.0 * sc * (10.0 ** -torch.arange(B).float()).view(B, 1)
        w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
        b = 0.1 * torch.randn(Co, generator=g)
        r = 0.1 * torch.randn(B, H, W, Co, generator=g) if res_on else None
        xn = x.double() * sc.double().view(B, 1, 1, Ci) + sh.double().view(B, 1, 1, Ci)
        if silu:
            xn = xn * torch.sigmoid(xn)
        ref = F.conv2d(xn.permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        if res_on:
            ref = ref + r.double()
        x, sc, sh, w, b = x.cuda(), sc.cuda().contiguous(), sh.cuda().contiguous(), w.cuda(), b.cuda()
        r = r.cuda() if res_on else None
        wp = e.pack_matrix(w.permute(2, 3, 1, 0).reshape(9 * Ci, Co).contiguous()).cuda()
        w2, ws = h2_pack(e, wp, Co)
        bound, oa = ra_zeros(B), ra_zeros(B)
        native.check(L.egr_gn_operand_bound(p(sc), p(sh), B, Ci, p(row_amax(e, x, B)), p(bound), e._st()), "bound")
        true_max = xn.abs().amax(dim=(1, 2, 3)).float()
        ratio = bound[::RA].cpu() / true_max
        assert bool((ratio >= 1.0).all()) and bool((ratio < 64.0).all()), ratio          # a bound, and within 6 bits here
        y = torch.empty(B, H, W, Co, device="cuda")
        part = torch.zeros(B * H * W // 32, Co // 4, 2, device="cuda")
        native.check(L.egr_conv_h2_gn(p(x), p(sc), p(sh), silu, p(w2), p(b), p(r), p(y), B, H, W, Ci, Co, 0, ws, p(bound), p(oa), p(part), e._st()), "conv_h2_gn")
        # the GroupNorm partials: (sum, sum of squares) per 32-pixel row segment and channel quad
        yq = y.double().view(B * H * W // 32, 32, Co // 4, 4)
        assert float((part[..., 0].double() - yq.sum(dim=(1, 3))).abs().max()) <= 1e-4 * float(yq.abs().sum(dim=(1, 3)).max())
        assert float((part[..., 1].double() - (yq * yq).sum(dim=(1, 3))).abs().max()) <= 1e-4 * float((yq * yq).sum(dim=(1, 3)).max())
        y1 = torch.empty_like(y)
        xn32 = xn.float().cuda().contiguous()
        native.check(L.egr_conv_nhwc(p(xn32), p(wp), p(b), p(None), p(r), p(y1), B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 1, 0, 0, 0.0, e._st()), "conv")
        assert torch.equal(oa[::RA], y.abs().amax(dim=(1, 2, 3)))
        for i in range(B):
            e2 = float((y[i].double().cpu() - ref[i]).norm() / ref[i].norm())
            e1 = float((y1[i].double().cpu() - ref[i]).norm() / ref[i].norm())
            # (the f32 kernel starts from the float32 ROUNDING of the normalised tensor, the fused loader normalises in fp32 itself)
            print(f"input-stationary 3x3 + GroupNorm, row {i} of {B}: rel L2 vs float64 {e2:.2e} (f32-MFMA kernel {e1:.2e})")
            assert e2 <= 1.25 * e1 + 2e-7 and e2 <= 5e-7, (B, H, W, Ci, Co, i, e1, e2)      # relative to the f32-MFMA kernel AND absolute
    # too few tiles / W % 32 != 0: refused, nothing launched
    x = torch.randn(1, 8, 48, 32, generator=g).cuda()
    y = torch.empty(1, 8, 48, 128, device="cuda")
    rc = L.egr_conv_h2_gn(p(x), p(sc), p(sh), 1, p(w2), p(None), p(None), p(y), 1, 8, 48, 32, 128, 0, ws, p(bound), p(None), p(None), e._st())
    assert rc == 3 and "qualify" in native.last_error()


def test_attention_products_on_two_fp16_terms(eng):
    """egr_bgemm_nt_h2 (both operands activations, each scaled per batch row from its own maximum) against the three-term bf16 form
    and float64, in the two shapes attention uses: S = alpha q k^T per head out of [B][T][C] tensors, and o = P v with P in [0, 1]
    (constant maxima) and v transposed; batch rows at different levels; out_amax."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(71)
    B, heads, T, d = 3, 4, 160, 32
    Cc = heads * d
    lv = (10.0 ** -torch.arange(B).float()).view(B, 1, 1)
    q = (torch.randn(B, T, Cc, generator=g) * lv).cuda()
    k = (torch.randn(B, T, Cc, generator=g) * lv * 3.0).cuda()
    S3, S2 = torch.empty(B, heads, T, T, device="cuda"), torch.empty(B, heads, T, T, device="cuda")
    alpha = d ** -0.5
    args = (B, heads, T, T, d, Cc, Cc, T, T * Cc, d, T * Cc, d, heads * T * T, T * T, alpha)
    native.check(L.egr_bgemm_nt_s3(p(q), p(k), p(S3), *args, e._st()), "bgemm_s3")
    native.check(L.egr_bgemm_nt_h2(p(q), p(k), p(S2), *args, p(row_amax(e, q, B)), p(row_amax(e, k, B)), p(None), e._st()), "bgemm_h2")
    ref = alpha * torch.einsum("bthd,bshd->bhts", q.double().view(B, T, heads, d), k.double().view(B, T, heads, d))
    for b in range(B):
        r3 = float((S3[b].double() - ref[b]).norm() / ref[b].norm())
        r2 = float((S2[b].double() - ref[b]).norm() / ref[b].norm())
        assert r2 <= 1.25 * r3 + 1e-8, (b, r3, r2)
    # P v: P = softmax rows in [0, 1] (constant maxima of 1), v transposed per batch row [C][T]
    P = torch.softmax(S2.view(B * heads * T, T), dim=1).view(B, heads, T, T).contiguous()
    v = (torch.randn(B, T, Cc, generator=g) * lv).cuda()
    vt = v.transpose(1, 2).contiguous()                             # [B][C][T]
    ones = torch.ones(B * RA, device="cuda")
    o3, o2 = torch.empty(B, T, Cc, device="cuda"), torch.empty(B, T, Cc, device="cuda")
    oa = ra_zeros(B)
    args = (B, heads, T, d, T, T, T, Cc, heads * T * T, T * T, Cc * T, d * T, T * Cc, d, 1.0)
    native.check(L.egr_bgemm_nt_s3(p(P), p(vt), p(o3), *args, e._st()), "bgemm_s3")
    native.check(L.egr_bgemm_nt_h2(p(P), p(vt), p(o2), *args, p(ones), p(row_amax(e, v, B)), p(oa), e._st()), "bgemm_h2")
    ref = torch.einsum("bhts,bshd->bthd", P.double(), v.double().view(B, T, heads, d)).reshape(B, T, Cc)
    for b in range(B):
        r3 = float((o3[b].double() - ref[b]).norm() / ref[b].norm())
        r2 = float((o2[b].double() - ref[b]).norm() / ref[b].norm())
        assert r2 <= 1.25 * r3 + 1e-8, (b, r3, r2)
    assert torch.equal(oa[::RA], o2.abs().amax(dim=(1, 2)))
