"""The two-term fp16 operand scheme of the split contractions (csrc/egr_nn_gemm_s3.hip scheme 1, egr_conv_h2 / egr_split2h_pack /
egr_absmax; handle logic in csrc/egr_flashsr.cpp: measure -> scale -> verify -> re-run on the bf16 terms).

Operator level, against a float64 convolution: the fp16 kernels must be no further from float64 than 1.25x the f32-MFMA kernel
(the same gate the three-term bf16 kernels are held to in tests/test_gpu_flashsr.py::test_split3_conv_error_vs_float64), at input
magnitudes from 1e-4 to 1e4 with the scale chosen from the measured maximum, for every tile shape, the 1-D input-stationary
kernel, split-K and the z-streamed GEMMs.  Handle level: calls after the first use the fp16 kernels and agree with the bf16 run to
fp32 round-off; identical calls give identical bits; an input 300x louder than the one the scales were measured on is detected and
re-run on the bf16 kernels with bit-identical results to a bf16-only handle.
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
        asc = scale_for(float(x.abs().max()), 12)
        y1 = torch.empty(B, H, W, Co, device="cuda")
        y2 = torch.empty_like(y1)
        amax = torch.zeros(1, device="cuda")
        native.check(L.egr_conv_nhwc(p(x), p(wp), p(b), p(None), p(None), p(y1), B, H, W, Ci, H, W, Co, k, k, 1, 1, k // 2, k // 2,
                                     0, 0, 0.0, e._st()), "conv")
        native.check(L.egr_conv_h2(p(x), p(w2), p(b), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, k // 2, k // 2,
                                   0, 0, 0.0, 1, 1, 0, 0, H, W, 1, 0, 0, 0, asc, ws, p(amax), e._st()), "conv_h2")
        assert float(amax.item()) == float(x.abs().max()), "the loader's maximum is the tensor's (stride 1: every element is read)"
        mx = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
        rms = lambda y: float((y.double() - ref).norm() / ref.norm())
        assert mx(y2) <= 1.25 * mx(y1) + 1e-8 and rms(y2) <= 1.25 * rms(y1) + 1e-8, ((B, H, W, Ci, Co, k), mx(y1), mx(y2), rms(y1), rms(y2))
        assert mx(y2) < 2e-6, mx(y2)


def test_h2_small_elements_keep_their_precision(eng):
    """Rows 60 dB below the tensor's maximum (a quiet passage next to a loud one) still come out at fp32-grade relative accuracy:
    every element within 2^-15 of the maximum keeps 22 significand bits."""
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
                               1, 0, 0, 0, scale_for(float(x.abs().max()), 12), ws, p(None), e._st()), "conv_h2")
    for b in range(B):
        r1 = float((y1[b].double() - ref[b]).norm() / ref[b].norm())
        r2 = float((y2[b].double() - ref[b]).norm() / ref[b].norm())
        assert r2 <= 1.25 * r1 + 1e-8, (b, r1, r2)
    # 100 dB below the maximum the second terms are fp16 SUBNORMALS (the matrix pipe must not flush them: an 11-bit operand would
    # put ~2e-4 on this row): absolute error 2^-37 of the maximum per element, i.e. <= 4e-6 of a row at 1e-5 of the maximum
    x[1] *= 1e-2
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, padding=1).permute(0, 2, 3, 1)
    native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, 1, 1, 0, 0, H, W,
                               1, 0, 0, 0, scale_for(float(x.abs().max()), 12), ws, p(None), e._st()), "conv_h2")
    r2 = float((y2[1].double() - ref[1]).norm() / ref[1].norm())
    print(f"row at 1e-5 of the maximum: relative error {r2:.2e}")
    assert r2 <= 4e-6, r2


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
        amax = torch.zeros(1, device="cuda")
        native.check(L.egr_conv_s3(p(x), p(w3), p(None), p(None), p(None), p(y3), B, 1, Wd, Ci, 1, Wd, Co, 1, k, 1, dil, 0, pad, 0, 0, 0.0, 1, 1, 0, 0,
                                   1, Wd, 1, 0, 0, 0, e._st()), "conv_s3")
        native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y2), B, 1, Wd, Ci, 1, Wd, Co, 1, k, 1, dil, 0, pad, 0, 0, 0.0, 1, 1, 0, 0,
                                   1, Wd, 1, 0, 0, 0, scale_for(float(x.abs().max()), 12), ws, p(amax), e._st()), "conv_h2")
        assert float(amax.item()) == float(x.abs().max())
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, padding=(0, pad), dilation=(1, dil)).permute(0, 2, 3, 1)
        r3 = float((y3.double() - ref).norm() / ref.norm())
        r2 = float((y2.double() - ref).norm() / ref.norm())
        m3 = float((y3.double() - ref).abs().max() / ref.abs().max())
        m2 = float((y2.double() - ref).abs().max() / ref.abs().max())
        assert r2 <= 1.25 * r3 + 1e-8 and m2 <= 1.5 * m3 + 1e-8 and m2 < 2e-6, (Wd, Ci, Co, k, dil, r3, r2, m3, m2)
    # z-stacked GEMMs: nz problems [P][Cin] x [Cin][Cout]
    for (nz, P, Ci, Co) in [(36, 700, 128, 128), (36, 300, 256, 256), (16, 500, 64, 128)]:
        x = torch.randn(nz, P, Ci, generator=g).cuda()
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
                                   nz, P * Ci, zf * 2 // 8, P * Co, scale_for(float(x.abs().max()), 12), ws, p(None), e._st()), "conv_h2")
        ref = torch.einsum("zpc,zcn->zpn", x.double(), w.double())
        r3 = float((y3.double() - ref).norm() / ref.norm())
        r2 = float((y2.double() - ref).norm() / ref.norm())
        assert r2 <= 1.25 * r3 + 1e-8, (nz, P, Ci, Co, r3, r2)


def test_h2_out_of_range_is_visible(eng):
    """A value beyond fp16's range after scaling cannot pass silently: the slot reports the true maximum (the handle's check:
    amax * a_scale >= 60000 -> re-run on the bf16 terms)."""
    e, cfg = eng
    L = e.L
    from egregora_amd import native
    g = torch.Generator().manual_seed(3)
    B, H, W, Ci, Co, k = 1, 8, 8, 32, 32, 3
    x = torch.randn(B, H, W, Ci, generator=g).cuda()
    x[0, 3, 3, 7] = 1.0e3
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).cuda()
    wp = e.pack_matrix(w.permute(2, 3, 1, 0).reshape(k * k * Ci, Co).contiguous()).cuda()
    w2, ws = h2_pack(e, wp, Co)
    y = torch.empty(B, H, W, Co, device="cuda")
    amax = torch.zeros(1, device="cuda")
    asc = 1024.0                                     # measured on an earlier, quieter input
    native.check(L.egr_conv_h2(p(x), p(w2), p(None), p(None), p(None), p(y), B, H, W, Ci, H, W, Co, k, k, 1, 1, 1, 1, 0, 0, 0.0, 1, 1, 0, 0, H, W,
                               1, 0, 0, 0, asc, ws, p(amax), e._st()), "conv_h2")
    assert float(amax.item()) == 1.0e3 and float(amax.item()) * asc >= 60000.0


def test_handle_scheme_measure_scale_verify(pack):
    """egr_flashsr_infer on a handle with the fp16 scheme: call 1 measures on the bf16 kernels (bit-equal to a bf16-only handle),
    calls 2 and 3 run the fp16 kernels (bit-equal to each other, fp32 round-off from call 1), a 300x louder input trips the range
    check and is re-run on the bf16 kernels (bit-equal to the bf16-only handle), and the call after it runs the fp16 kernels again."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    old = E.FlashSREngine.SPLIT
    try:
        E.FlashSREngine.SPLIT = "f16x2"
        eh = E.FlashSREngine(cfg, P)
        E.FlashSREngine.SPLIT = "bf16x3"
        eb = E.FlashSREngine(cfg, P)
    finally:
        E.FlashSREngine.SPLIT = old
    g = torch.Generator().manual_seed(9)
    x = (0.05 * torch.randn(3, cfg.chunk, generator=g)).cuda()
    yb = eb.c_infer(x, None, 4)
    assert not eb.split_info()["enabled"] and eh.split_info()["enabled"] and eh.split_info()["slots"] > 10
    y1 = eh.c_infer(x, None, 4)
    assert torch.equal(y1, yb) and eh.split_info()["calibrated"]
    y2 = eh.c_infer(x, None, 4)
    y3 = eh.c_infer(x, None, 4)
    assert torch.equal(y2, y3)
    rel = float((y2 - yb).double().norm() / yb.double().norm())
    assert 0.0 < rel < 2e-5, rel                     # different kernels (not bit-equal), same function
    assert eh.split_info()["reruns"] == 0
    xl = 300.0 * x
    ybl = eb.c_infer(xl, None, 4)
    yl = eh.c_infer(xl, None, 4)
    info = eh.split_info()
    if info["reruns"] == 1:                          # (an input normalisation inside the model may keep the activations in range)
        assert torch.equal(yl, ybl)
    else:
        assert float((yl - ybl).double().norm() / ybl.double().norm()) < 2e-5
    yl2 = eh.c_infer(xl, None, 4)
    assert eh.split_info()["reruns"] == info["reruns"]
    assert float((yl2 - ybl).double().norm() / ybl.double().norm()) < 2e-5
    assert bool(torch.isfinite(yl2).all())
    # the other end of the range: silence (all maxima zero: scales kept, nothing to verify) and an input 100 dB below the one the
    # scales were measured on (re-run on the bf16 kernels unless the model's own normalisation keeps the activations in range)
    yz = eh.c_infer(torch.zeros_like(x), None, 4)
    ybz = eb.c_infer(torch.zeros_like(x), None, 4)            # (not silence at the output: the diffusion noise drives the model)
    assert bool(torch.isfinite(yz).all()) and float((yz - ybz).double().norm() / ybz.double().norm()) < 2e-5
    xq = 1e-5 * xl
    ybq = eb.c_infer(xq, None, 4)
    before = eh.split_info()["reruns"]
    yq = eh.c_infer(xq, None, 4)
    if eh.split_info()["reruns"] > before:
        assert torch.equal(yq, ybq)
    else:
        assert float((yq - ybq).double().norm() / (ybq.double().norm() + 1e-30)) < 2e-5
    eh.set_split("bf16x3")
    assert torch.equal(eh.c_infer(x, None, 4), yb)
    # A first call with more rows than the measuring part takes (6): rows 0..5 are the bf16 walk bit for bit, the rest of the SAME call
    # already runs the fp16 terms (ids continue: implicit ids 6..9 = the explicit ones).
    eh.set_split("f16x2")
    assert not eh.split_info()["calibrated"]
    x10 = (0.05 * torch.randn(10, cfg.chunk, generator=g)).cuda()
    yb10 = eb.c_infer(x10, None, 4)
    yh10 = eh.c_infer(x10, None, 4)
    assert torch.equal(yh10[:6], eb.c_infer(x10[:6].contiguous(), None, 4)) and eh.split_info()["calibrated"]
    assert not torch.equal(yh10[6:], yb10[6:])
    assert float((yh10 - yb10).double().norm() / yb10.double().norm()) < 2e-4       # (tile choices follow the row count of a forward)
    yh10e = eh.c_infer(x10, torch.arange(10, dtype=torch.int64, device="cuda"), 4)
    assert float((yh10e - yb10).double().norm() / yb10.double().norm()) < 2e-4
    assert torch.equal(yh10e, eh.c_infer(x10, None, 4))
    eh.close(); eb.close()


def test_fp16_calls_repeat_bit_for_bit_with_concurrent_row_groups(pack):
    """The bug class a flaky test found in round 2 (a rare race in shared scratch): 14 rows = two concurrent 7-row groups on side
    streams, fp16 operand terms with the loaders' atomic maxima, 20 repetitions -- every repetition must give the same bits, and
    the maxima the handle keeps must not drift (no re-run, scales kept)."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E, native
    cfg = A.tiny_config()
    e = E.FlashSREngine(cfg, A.init_params(cfg, 0))
    native.check(e.L.egr_flashsr_set_streams(C.c_void_p(e.handle), 2, 6), "set_streams")
    x = (0.1 * torch.randn(14, cfg.chunk, generator=torch.Generator().manual_seed(21))).cuda()
    e.c_infer(x, None, 5)                            # measuring part + the first fp16 rows
    ref = e.c_infer(x, None, 5)
    for _ in range(20):
        assert torch.equal(e.c_infer(x, None, 5), ref)
    info = e.split_info()
    assert info["reruns"] == 0 and info["calls"] == 22
    e.close()
