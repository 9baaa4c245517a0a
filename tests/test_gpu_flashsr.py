"""GPU numerics of the FlashSR operators and of the assembled engine against plain PyTorch fp32 of the same
op / the same layer table (oracle/flashsr_torch.py, run on the CPU in float32).

Tolerances (float32 both sides; only summation order and exp/sin approximations differ):
  single operators : max|diff| <= 2e-5 * max|ref| (+ 1e-6)
  assembled stages : relative L2 error <= 2e-4 per stage at the toy config, <= 1e-3 for the final waveform
  output LSD (reference metric) between engine and torch reference waveforms <= 0.05 dB at the toy config
Upstream FlashSR itself is absent: parity with it is unpinned (see flashsr_arch.py).
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def PyDriverEngine(*a, **k):
    """The operator-by-operator Python walk of the graph (tools/flashsr_pydriver.py: test tooling on top of the product engine)."""
    from tools.flashsr_pydriver import PyDriverEngine as cls
    return cls(*a, **k)

pytestmark = pytest.mark.gpu


def close(got, want, tol=2e-5):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = float((got - want).abs().max())
    ref = float(want.abs().max())
    assert err <= tol * ref + 1e-6, (err, ref)


def rel_l2(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return float((got - want).norm() / (want.norm() + 1e-30))


@pytest.fixture(scope="module")
def eng(pack):
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    return PyDriverEngine(cfg, P), cfg, P


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, up2
    (2, 16, 24, 32, 48, 3, 1, 1, 0),
    (1, 33, 17, 16, 128, 3, 1, 1, 0),      # ragged M, BN=128
    (2, 16, 16, 64, 64, 3, 2, 1, 0),       # UNet downsample
    (2, 16, 16, 32, 32, 3, 2, 0, 0),       # VAE downsample: pad (0,1,0,1)
    (2, 8, 12, 48, 40, 3, 1, 1, 1),        # nearest 2x + conv
    (3, 10, 10, 1, 32, 3, 1, 1, 0),        # Cin = 1 (generic gather path)
    (2, 12, 12, 32, 1, 3, 1, 1, 0),        # Cout = 1 (scalar weight path)
    (2, 9, 7, 130, 200, 1, 1, 0, 0),       # 1x1, Cin not a multiple of 16
    (1, 64, 32, 256, 256, 3, 1, 1, 0),
    (1, 9, 7, 32, 256, 3, 1, 1, 0),        # 128x256 tile, ragged M
    (2, 8, 8, 64, 512, 1, 1, 0, 0),        # 128x256 tile, 1x1
    (1, 6, 10, 48, 256, 3, 1, 1, 1),       # 128x256 tile, nearest-2x input
]


@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,pad,up", CONV_CASES)
def test_conv_nhwc_vs_torch(eng, B, H, W, Ci, Co, k, s, pad, up):
    e, cfg, P = eng
    g = torch.Generator().manual_seed(H * 131 + Ci)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
    b = torch.randn(Co, generator=g)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    if s == 2 and pad == 0:
        want = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w, b, stride=2)
    else:
        want = F.conv2d(xi, w, b, stride=s, padding=pad)
    res = torch.randn(want.shape, generator=g)
    want = F.silu(want + res)
    OH, OW = want.shape[2], want.shape[3]
    e.add_weight("t.weight", w)
    e.w["t.bias"] = b.cuda()
    got = e.conv(nhwc(x).cuda(), "t", B, H, W, Ci, OH, OW, Co, k, k, s, 1, pad, pad, up, 1, res=nhwc(res).cuda())
    close(nchw(got), want)


def test_split3_conv_error_vs_float64(eng):
    """The bf16x3 kernel (three-way exact split, six products on the bf16 MFMA, fp32 accumulate) must be at least as close
    to float64 as the f32-MFMA kernel (an fmaf chain): max and rms error within 1.25x of it on every case, and both far
    inside the operator tolerance.  Covers the 256x128 / 128x{128,64,32} tiles, ragged M and Cout, split-K shapes."""
    e, cfg, P = eng
    L = e.L
    from egregora_amd import native
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    g = torch.Generator().manual_seed(77)
    cases = [(2, 16, 12, 128, 128, 3), (1, 9, 7, 256, 96, 3), (3, 8, 8, 512, 40, 1), (2, 10, 6, 64, 200, 3),
             (9, 128, 128, 32, 128, 3),      # M = 147456: 256-row tiles
             (2, 8, 4, 640, 640, 3),         # small M, long K: split-K
             (1, 5, 3, 16, 20, 3)]
    for (B, H, W, Ci, Co, k) in cases:
        x = torch.randn(B, H, W, Ci, generator=g).cuda()
        w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).cuda()
        b = torch.randn(Co, generator=g).cuda()
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
        wp = e.pack_matrix(w.permute(2, 3, 1, 0).reshape(k * k * Ci, Co).contiguous()).cuda()
        ns = wp.shape[0]
        w3 = torch.empty(ns * 3 * Co * 16, dtype=torch.bfloat16, device="cuda")
        native.check(L.egr_split3_pack(p(wp), p(w3), ns, Co, e._st()), "split3")
        # the three planes sum back to the fp32 weights exactly
        planes = w3.view(ns, 3, Co, 16).float().sum(1)
        assert torch.equal(planes, wp)
        y1 = torch.empty(B, H, W, Co, device="cuda")
        y2 = torch.empty_like(y1)
        native.check(L.egr_conv_nhwc(p(x), p(wp), p(b), p(None), p(None), p(y1), B, H, W, Ci, H, W, Co, k, k, 1, 1, k // 2, k // 2,
                                     0, 0, 0.0, e._st()), "conv")
        native.check(L.egr_conv_s3(p(x), p(w3), p(b), p(None), p(None), p(y2), B, H, W, Ci, H, W, Co, k, k, 1, 1, k // 2, k // 2,
                                   0, 0, 0.0, 1, 1, 0, 0, H, W, 1, 0, 0, 0, e._st()), "conv_s3")
        mx = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
        rms = lambda y: float((y.double() - ref).norm() / ref.norm())
        assert mx(y2) <= 1.25 * mx(y1) + 1e-8 and rms(y2) <= 1.25 * rms(y1) + 1e-8, ((B, H, W, Ci, Co, k), mx(y1), mx(y2), rms(y1), rms(y2))
        assert mx(y2) < 2e-6, mx(y2)


def test_f32_mfma_mode_still_matches(pack):
    """EGREGORA_FLASHSR_MFMA=f32 keeps every contraction on v_mfma_f32_32x32x2_f32; both modes agree to fp32 round-off."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    old = E.FlashSREngine.MFMA_MODE
    try:
        E.FlashSREngine.MFMA_MODE = "f32"
        e1 = PyDriverEngine(cfg, P)
        E.FlashSREngine.MFMA_MODE = "bf16x3"
        e2 = PyDriverEngine(cfg, P)
    finally:
        E.FlashSREngine.MFMA_MODE = old
    assert not e1.w3 and e2.w3
    x = 0.3 * torch.randn(2, cfg.chunk, generator=torch.Generator().manual_seed(5)).cuda()
    nz = e1.noise(2, None, 3)
    y1, y2 = e1.forward_rows(x, nz), e2.forward_rows(x, nz)
    assert rel_l2(y2, y1) < 2e-5, rel_l2(y2, y1)


@pytest.mark.parametrize("f4", [False, True])
def test_winograd_conv_vs_torch(eng, f4):
    """Winograd paths (input transform, z-batched MFMA GEMMs, output transform + epilogue): F(2x2,3x3) with 16
    components, and F(4x4,3x3) with 36 where H and W are multiples of 4 (other shapes fall back to F(2x2))."""
    e, cfg, P = eng
    g = torch.Generator().manual_seed(41)
    oldf4 = e.WINO_F4          # the switches are frozen per engine at construction: set them on the instance
    e.WINO_F4 = f4
    for (B, H, W, Ci, Co, use_res, act) in [(2, 8, 12, 32, 48, True, 1), (1, 16, 16, 64, 128, False, 0), (3, 6, 4, 144, 80, True, 0),
                                            (2, 4, 4, 16, 20, True, 1), (1, 8, 8, 24, 36, False, 1)]:     # last: Cin % 16 != 0 (f32 GEMMs)
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
        b = torch.randn(Co, generator=g)
        want = F.conv2d(x, w, b, padding=1)
        res = torch.randn(want.shape, generator=g)
        if use_res:
            want = want + res
        if act:
            want = F.silu(want)
        e.add_weight("wg.weight", w)
        e.add_winograd("wg.weight", w)
        e.w["wg.bias"] = b.cuda()
        assert ("wg.weight.wino4" in e.w) == f4
        got = e._conv_winograd(nhwc(x).cuda(), "wg", act, nhwc(res).cuda() if use_res else None, None)
        close(nchw(got), want, 3e-5)
        e.w.pop("wg.weight.wino")
        e.w.pop("wg.weight.wino4", None)
    e.WINO_F4 = oldf4


def test_winograd4_error_vs_float64(eng):
    """F(4x4,3x3) in fp32 with the split-bf16 GEMMs: max error <= 1e-5 of the output maximum against a float64 direct
    convolution at FlashSR's channel counts (the fp32 direct form sits at ~3e-7, F(2x2,3x3) at ~1e-6) --
    half the 2e-5 single-operator tolerance of this file (measured 2e-6 .. 5e-6 with the points 0, +-3/4, +-3/2)."""
    e, cfg, P = eng
    g = torch.Generator().manual_seed(45)
    for (B, H, W, Ci, Co) in [(1, 16, 16, 256, 256), (1, 8, 8, 1024, 512)]:
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
        b = torch.randn(Co, generator=g)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        e.add_weight("w4.weight", w)
        e.add_winograd("w4.weight", w)
        e.w["w4.bias"] = b.cuda()
        assert "w4.weight.wino4" in e.w
        got = nchw(e._conv_winograd(nhwc(x).cuda(), "w4", 0, None, None)).cpu().double()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err <= 1e-5, err
        e.w.pop("w4.weight.wino"), e.w.pop("w4.weight.wino4")


def test_groupnorm_statistics_from_winograd_output_partials(eng):
    """The F(4x4) output transform leaves per-thread (sum, sum of squares); GroupNorm coefficients of its output formed from
    those partials (no second pass over y, fixed summation order) equal the ones computed from y itself."""
    e, cfg, P = eng
    g = torch.Generator().manual_seed(47)
    old = e.cfg.gn_groups
    try:
        for (B, H, W, Ci, Co, G, use_res, act) in [(2, 8, 12, 32, 64, 8, True, 1), (3, 16, 8, 64, 128, 32, False, 0), (1, 4, 4, 16, 48, 4, True, 0)]:
            e.cfg.gn_groups = G
            x = torch.randn(B, Ci, H, W, generator=g)
            w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
            b = torch.randn(Co, generator=g)
            res = torch.randn(B, Co, H, W, generator=g)
            ga, be = 1 + 0.1 * torch.randn(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
            e.add_weight("wp.weight", w)
            e.add_winograd("wp.weight", w)
            e.w["wp.bias"] = b.cuda()
            e.w["gp.weight"], e.w["gp.bias"] = ga.cuda(), be.cuda()
            y = e._conv_winograd(nhwc(x).cuda(), "wp", act, nhwc(res).cuda() if use_res else None, None)
            assert hasattr(y, "_egr_gn_partials")
            sc, sh = e.gn_coeff(y, "gp", 1e-6)
            del y._egr_gn_partials
            sc2, sh2 = e.gn_coeff(y, "gp", 1e-6)                  # statistics kernel over y itself
            yd = nchw(y).cpu().double().view(B, G, Co // G, H * W)
            mean, var = yd.mean(dim=(2, 3)), yd.var(dim=(2, 3), unbiased=False)
            want_sc = (ga.double().view(1, G, -1) / (var + 1e-6).sqrt().unsqueeze(-1)).reshape(B, Co)
            want_sh = be.double().view(1, Co) - (mean.unsqueeze(-1) * want_sc.view(B, G, -1)).reshape(B, Co)
            for got_sc, got_sh in ((sc, sh), (sc2, sh2)):
                assert float((got_sc.cpu().double() - want_sc).abs().max() / want_sc.abs().max()) <= 2e-6
                assert float((got_sh.cpu().double() - want_sh).abs().max()) <= 2e-6 * float(want_sh.abs().max() + 1.0)
            e.w.pop("wp.weight.wino"), e.w.pop("wp.weight.wino4")
    finally:
        e.cfg.gn_groups = old


def test_fused_groupnorm_conv_vs_torch(eng):
    """conv3x3(silu(groupnorm(x))) with the normalisation applied in the conv loader / the Winograd input transform."""
    e, cfg, P = eng
    g = torch.Generator().manual_seed(43)
    old, oldf = e.cfg.gn_groups, e.FUSE_GN
    e.cfg.gn_groups = 8
    e.FUSE_GN = "all"
    try:
        for wino in (False, True):
            for (B, H, W, Ci, Co) in [(2, 8, 12, 32, 48), (3, 6, 4, 64, 80)]:
                x = torch.randn(B, Ci, H, W, generator=g) * 2 + 0.5
                ga, be = 1 + 0.1 * torch.randn(Ci, generator=g), 0.1 * torch.randn(Ci, generator=g)
                w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
                b = torch.randn(Co, generator=g)
                res = torch.randn(B, Co, H, W, generator=g)
                want = F.conv2d(F.silu(F.group_norm(x, 8, ga, be, 1e-6)), w, b, padding=1) + res
                e.w["fg.weight"], e.w["fg.bias"] = ga.cuda(), be.cuda()
                e.add_weight("fc.weight", w)
                e.w["fc.bias"] = b.cuda()
                if wino:
                    e.add_winograd("fc.weight", w)
                got = e.gn_conv3(nhwc(x).cuda(), "fg", 1e-6, "fc", res=nhwc(res).cuda())
                close(nchw(got), want, 4e-5)
                e.w.pop("fc.weight.wino", None)
                e.w.pop("fc.weight.wino4", None)
    finally:
        e.cfg.gn_groups = old
        e.FUSE_GN = oldf


def test_upsample_conv_as_four_phase_convs(eng):
    """nearest-2x + 3x3 conv evaluated as four 2x2 phase convs with pre-summed taps (4/9 of the multiplies)."""
    e, cfg, P = eng
    g = torch.Generator().manual_seed(21)
    for (B, H, W, Ci, Co) in [(2, 8, 12, 32, 48), (1, 5, 7, 16, 130), (3, 16, 4, 64, 32)]:
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
        b = torch.randn(Co, generator=g)
        want = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
        e.add_weight("up.weight", w)
        e.add_upsample_phases("up.weight", w)
        e.w["up.bias"] = b.cuda()
        got = e.conv3(nhwc(x).cuda(), "up", up2=1)
        close(nchw(got), want)


def test_thin_end_convs_vs_torch(eng):
    """The VAE's thin ends: 3x3 convs with 1..3 outputs as a 1x1 contraction onto per-tap products + egr_tap_gather, and the
    one-input-channel 3x3 conv on the vector ALU (egr_conv_cin1); borders are the zero padding."""
    e, cfg, P = eng
    g = torch.Generator().manual_seed(77)
    for (B, H, W, Ci, Co) in [(2, 8, 12, 32, 1), (1, 5, 7, 128, 1), (3, 16, 4, 64, 3), (2, 9, 6, 16, 2)]:
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
        b = torch.randn(Co, generator=g)
        e.add_weight("te.weight", w)
        e.add_weight("te.weight.taps", w.permute(2, 3, 0, 1).reshape(9 * Co, Ci))
        e.w["te.bias"] = b.cuda()
        assert e.thin and e._s3("te.weight.taps", Ci, nhwc(x).cuda()) is not None
        close(nchw(e.conv3(nhwc(x).cuda(), "te")), F.conv2d(x, w, b, padding=1))
        e.w.pop("te.weight.taps"), e.w3.pop("te.weight.taps")
    for (B, H, W, Co) in [(2, 8, 12, 32), (1, 5, 7, 128), (3, 16, 4, 4)]:
        x = torch.randn(B, 1, H, W, generator=g)
        w = torch.randn(Co, 1, 3, 3, generator=g) / 3.0
        b = torch.randn(Co, generator=g)
        e.add_weight("ti.weight", w)
        e.w["ti.bias"] = b.cuda()
        close(nchw(e.conv3(nhwc(x).cuda(), "ti")), F.conv2d(x, w, b, padding=1))


@pytest.mark.parametrize("k,dil,stride", [(3, 1, 1), (7, 3, 1), (11, 5, 1), (5, 1, 2), (13, 1, 6)])
def test_conv1d_vs_torch(eng, k, dil, stride):
    e, cfg, P = eng
    g = torch.Generator().manual_seed(k)
    B, L, Ci, Co = 2, 600, 32, 48
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) / math.sqrt(Ci * k)
    b = torch.randn(Co, generator=g)
    pad = dil * (k - 1) // 2 if stride == 1 else (k - 1) // 2
    want = F.conv1d(x, w, b, stride=stride, dilation=dil, padding=pad)
    e.add_weight("t1.weight", w)
    e.w["t1.bias"] = b.cuda()
    got = e.conv1d(x.permute(0, 2, 1).contiguous().cuda(), "t1", k, stride=stride, dil=dil, pad=pad)
    close(got.permute(0, 2, 1), want)


@pytest.mark.parametrize("Ci,Co,L,k,dil", [(16, 16, 256, 11, 5), (32, 32, 384, 7, 3), (64, 64, 128, 3, 1), (128, 128, 256, 11, 1),
                                           (256, 256, 128, 7, 5), (48, 40, 256, 3, 3), (16, 130, 128, 11, 3), (32, 32, 512, 11, 5),
                                           (64, 64, 768, 7, 1), (16, 16, 1024, 3, 3)])
def test_input_stationary_conv1d_vs_torch(eng, Ci, Co, L, k, dil):
    """Stride-1 'same' 1-D convolutions with L % 128 == 0 take k_conv1d_s3 (halo tile split once into LDS, every tap reads
    it at a row offset): all channel-chunk widths and tile variants, with bias + residual + leaky epilogue."""
    e, cfg, P = eng
    g = torch.Generator().manual_seed(Ci + k)
    B = 3
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) / math.sqrt(Ci * k)
    b = torch.randn(Co, generator=g)
    res = torch.randn(B, Co, L, generator=g)
    pad = dil * (k - 1) // 2
    want = F.conv1d(x, w, b, dilation=dil, padding=pad) + res
    e.add_weight("t1s.weight", w)
    e.w["t1s.bias"] = b.cuda()
    got = e.conv1d(x.permute(0, 2, 1).contiguous().cuda(), "t1s", k, dil=dil, pad=pad, res=res.permute(0, 2, 1).contiguous().cuda())
    close(got.permute(0, 2, 1), want)


def test_groupnorm_layernorm_softmax_geglu(eng):
    e, cfg, P = eng
    g = torch.Generator().manual_seed(3)
    for (B, H, W, Cc, G) in [(2, 16, 8, 32, 8), (1, 37, 5, 64, 8), (3, 4, 4, 256, 8)]:
        x = torch.randn(B, Cc, H, W, generator=g) * 3 + 1.5
        ga, be = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
        e.w["gn.weight"], e.w["gn.bias"] = ga.cuda(), be.cuda()
        old = e.cfg.gn_groups
        e.cfg.gn_groups = G
        for silu in (False, True):
            want = F.group_norm(x, G, ga, be, 1e-6)
            want = F.silu(want) if silu else want
            close(nchw(e.groupnorm(nhwc(x).cuda(), "gn", 1e-6, silu)), want, 3e-5)
        e.cfg.gn_groups = old
    x = torch.randn(50, 96, generator=g) * 2 + 0.3
    ga, be = torch.randn(96, generator=g), torch.randn(96, generator=g)
    e.w["ln.weight"], e.w["ln.bias"] = ga.cuda(), be.cuda()
    close(e.layernorm(x.cuda(), "ln"), F.layer_norm(x, (96,), ga, be), 3e-5)
    s = torch.randn(37, 300, generator=g) * 4
    sc = s.clone().cuda()
    from egregora_amd import native
    native.check(e.L.egr_softmax_rows(C.c_void_p(sc.data_ptr()), 37, 300, native.stream_ptr()), "softmax")
    close(sc, torch.softmax(s, -1), 2e-5)
    for cols in (1024, 1500, 2048, 4096, 5000):      # the row in registers up to 4096 columns (k_softmax_reg), the three-pass form beyond
        s = torch.randn(5, cols, generator=g) * 6
        sc = s.clone().cuda()
        native.check(e.L.egr_softmax_rows(C.c_void_p(sc.data_ptr()), 5, cols, native.stream_ptr()), "softmax")
        close(sc, torch.softmax(s.double(), -1).float(), 2e-5)
        assert float(sc.sum(-1).sub(1).abs().max()) < 1e-5
    u = torch.randn(20, 64, generator=g)
    out = torch.empty(20, 32, device="cuda")
    native.check(e.L.egr_geglu(C.c_void_p(u.cuda().data_ptr()), C.c_void_p(out.data_ptr()), 20, 32, native.stream_ptr()), "geglu")
    a, gate = u.chunk(2, -1)
    close(out, a * F.gelu(gate))


@pytest.mark.parametrize("B,T,Cc,heads", [(2, 64, 32, 2), (1, 200, 64, 1), (2, 128, 96, 3), (1, 272, 128, 2), (2, 144, 1024, 1)])
def test_attention_vs_torch(eng, B, T, Cc, heads):
    e, cfg, P = eng
    g = torch.Generator().manual_seed(T)
    q, k, v = (torch.randn(B, T, Cc, generator=g) for _ in range(3))
    d = Cc // heads
    sp = lambda t: t.reshape(B, T, heads, d).permute(0, 2, 1, 3)
    w = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * d ** -0.5, -1)
    want = (w @ sp(v)).permute(0, 2, 1, 3).reshape(B * T, Cc)
    got = e.attention(q.reshape(-1, Cc).cuda(), k.reshape(-1, Cc).cuda(), v.reshape(-1, Cc).cuda(), B, T, Cc, heads)
    close(got, want)


def test_snake_aa_and_convtranspose_vs_torch(eng):
    e, cfg, P = eng
    from oracle import flashsr_torch as R
    g = torch.Generator().manual_seed(9)
    for (B, L, Cc) in [(2, 100, 8), (1, 257, 70), (2, 12, 3), (2, 300, 64), (1, 1000, 32), (2, 777, 16), (1, 20, 128), (1, 4, 16)]:
        x = torch.randn(B, Cc, L, generator=g)
        al, be = 0.3 * torch.randn(Cc, generator=g), 0.3 * torch.randn(Cc, generator=g)
        e.w["sa"], e.w["sb"] = al.cuda(), be.cuda()
        want = R._act_aa(x, al, be, e.filt.cpu())
        got = e.snake(x.permute(0, 2, 1).contiguous().cuda(), "sa", "sb")
        close(got.permute(0, 2, 1), want, 3e-5)
    from egregora_amd import flashsr_arch as A, native
    for r in (2, 3, 5, 6):
        kt = A.up_kernel(r)
        B, Lin, Ci, Co = 2, 40, 32, 24
        x = torch.randn(B, Ci, Lin, generator=g)
        w = torch.randn(Ci, Co, kt, generator=g) / math.sqrt(Ci * 2)
        b = torch.randn(Co, generator=g)
        add = torch.randn(B, Co, Lin * r, generator=g)
        want = F.conv_transpose1d(x, w, b, stride=r, padding=(kt - r) // 2) + add
        wt = e.pack_matrix(w.permute(0, 2, 1).reshape(Ci, kt * Co).contiguous()).cuda()
        Y = e.conv(x.permute(0, 2, 1).contiguous().cuda(), None, B * Lin, 1, 1, Ci, 1, 1, kt * Co, 1, 1, bias=False, w=wt)
        out = torch.empty(B, Lin * r, Co, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        native.check(e.L.egr_col2im_convtr1d(p(Y), p(b.cuda()), p(add.permute(0, 2, 1).contiguous().cuda()), p(out), B, Lin,
                                             Lin * r, Co, kt, r, (kt - r) // 2, native.stream_ptr()), "col2im")
        close(out.permute(0, 2, 1), want)


def test_logmel_vs_torch(eng):
    e, cfg, P = eng
    from oracle import flashsr_torch as R
    from egregora_amd import flashsr_arch as A
    g = torch.Generator().manual_seed(4)
    x = 0.3 * torch.randn(3, cfg.chunk, generator=g)
    want = R.log_mel(x, cfg, torch.from_numpy(A.mel_filterbank(cfg)))
    got = e.log_mel(x.cuda())
    close(got.permute(0, 3, 1, 2), want, 5e-5)


def test_lowpass_input_vs_torch(pack):
    """lowpass_input=True: cutoff detection + zero-phase Chebyshev gain on the device vs the torch definition."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    from oracle import flashsr_torch as R
    import dataclasses
    cfg = A.tiny_config()
    cfg = dataclasses.replace(cfg, chunk=3840, n_fft=128, hop=30)
    e = PyDriverEngine(cfg, A.init_params(cfg, 0))
    g = torch.Generator().manual_seed(11)
    t = torch.arange(cfg.chunk) / cfg.sr
    # band-limited rows (content below ~6 kHz / ~3 kHz) plus a little wide-band noise
    x = torch.stack([sum(torch.sin(2 * math.pi * f0 * t + i) / (i + 1) for i, f0 in enumerate((300.0, 1200.0, 2500.0, 5800.0))),
                     sum(torch.sin(2 * math.pi * f0 * t + i) / (i + 1) for i, f0 in enumerate((200.0, 900.0, 2900.0)))]).float()
    x = 0.3 * x / x.abs().max() + 1e-4 * torch.randn(2, cfg.chunk, generator=g)
    want, cuts = R.lowpass_ref(x, cfg)
    got = e.lowpass(x.cuda())
    assert e.last_cutoff_bins.cpu().tolist() == cuts and 0 < cuts[1] < cuts[0] < cfg.n_fft // 2
    close(got, want, 2e-4)
    # and the full-size chunk length plans too
    cfgF = A.FlashSRConfig()
    from tools.flashsr_pydriver import PyDriverEngine as PD
    eF = PD.__new__(PD)          # only the pieces lowpass() needs
    eF.cfg, eF.dev, eF.L = cfgF, torch.device("cuda"), e.L
    eF.window = torch.hann_window(cfgF.n_fft, periodic=True).cuda()
    eF.ldm = ((cfgF.n_fft // 2 + 1 + 15) // 16) * 16
    xf = 0.2 * torch.randn(2, cfgF.chunk, generator=g)
    wantF, cutsF = R.lowpass_ref(xf, cfgF)
    gotF = eF.lowpass(xf.cuda())
    assert eF.last_cutoff_bins.cpu().tolist() == cutsF
    close(gotF, wantF, 2e-4)


def test_randn_is_rank_independent_and_normal(eng):
    e, cfg, P = eng
    ids = torch.tensor([5, 6, 7, 1000], dtype=torch.int64, device="cuda")
    a = e.noise(4, ids, 123)
    b = e.noise(2, ids[1:3].contiguous(), 123)
    assert torch.equal(a[1:3], b)                         # a row depends only on (seed, global row id)
    c = e.noise(4, ids, 124)
    assert not torch.equal(a, c)
    big = torch.empty(8, 100000, device="cuda")
    from egregora_amd import native
    native.check(e.L.egr_randn(C.c_void_p(big.data_ptr()), 100000, 8, 7, C.c_void_p(0), native.stream_ptr()), "randn")
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1.0) < 5e-3
    assert abs(float((big ** 4).mean()) - 3.0) < 0.05


def test_engine_matches_torch_reference_stage_by_stage(eng):
    e, cfg, P = eng
    from oracle import flashsr_torch as R
    from egregora_amd import flashsr_arch as A
    g = torch.Generator().manual_seed(1)
    B = 3
    x = 0.3 * torch.randn(B, cfg.chunk, generator=g)
    ids = torch.arange(B, dtype=torch.int64, device="cuda")
    noise = e.noise(B, ids, 0)
    got_st, want_st = {}, {}
    y = e.forward_rows(x.cuda(), noise, got_st)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = R.flashsr_forward(x, nchw(noise.cpu()), P, cfg, A.unet_blocks(cfg), torch.from_numpy(A.mel_filterbank(cfg)),
                                 torch.from_numpy(A.kaiser_sinc_filter(cfg.aa_taps)), want_st)
    for k in ("mel", "z_cond", "v", "z0", "mel_hat"):
        gt = got_st[k]
        gt = gt.permute(0, 3, 1, 2) if gt.dim() == 4 else gt
        assert rel_l2(gt, want_st[k]) <= 2e-4, (k, rel_l2(gt, want_st[k]))
    assert rel_l2(y, want) <= 1e-3
    from oracle import metrics as om
    assert om.lsd_audio(want.numpy().reshape(-1), y.cpu().numpy().reshape(-1), 256, 64)[0] <= 0.05


def test_node_end_to_end_with_synthetic_engine(pack, eng):
    """The ComfyUI node drives chunking -> engine -> WOLA on the device; checked against the oracle glue wrapped
    around the torch reference graph (toy config: chunk 3840, hop scaled likewise)."""
    e, cfg, P = eng
    from egregora_amd import audio_glue as ag, device_ops as ops, flashsr_engine as E
    from oracle import flashsr_torch as R, glue as og
    from egregora_amd import flashsr_arch as A
    win, hop = cfg.chunk, cfg.chunk - 375
    total = 2 * hop + 1000
    g = torch.Generator().manual_seed(2)
    x = 0.3 * torch.randn(2, total, generator=g)
    E.set_engine(e)
    try:
        sp = ag.spans(total, win, hop)
        preds = E.infer_spans(x.cuda(), len(sp), win, hop, False)
        got = ops.wola_stitch(preds, total, win, hop).cpu().numpy()
    finally:
        E.set_engine(None)
    fb, filt = torch.from_numpy(A.mel_filterbank(cfg)), torch.from_numpy(A.kaiser_sinc_filter(cfg.aa_taps))
    idx = [0]

    def model(c):
        k = idx[0]; idx[0] += 1
        ids = torch.tensor([k * 2, k * 2 + 1], dtype=torch.int64, device="cuda")
        nz = nchw(e.noise(2, ids, E.SEED).cpu())
        with torch.no_grad():
            return R.flashsr_forward(torch.from_numpy(c), nz, P, cfg, A.unet_blocks(cfg), fb, filt).numpy()
    want = og.flashsr_node_glue(x.numpy(), model, win, hop)
    assert got.shape == want.shape
    assert float(np.abs(got - want).max()) <= 2e-3 * float(np.abs(want).max())


def _to_device(o, dev):
    if isinstance(o, torch.Tensor):
        return o.to(dev)
    if isinstance(o, dict):
        return {k: _to_device(v, dev) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_to_device(v, dev) for v in o)
    return o


def f64_forward_on_gpu(x, nz_nchw, P, cfg, stages=None):
    """The oracle's graph (oracle/flashsr_torch.py, the same code object) in FLOAT64 with its tensors on the GPU: PyTorch's own
    double-precision kernels (im2col + rocBLAS dgemm for the convolutions) take 3 s for two full-size rows where the host CPU takes
    114 s per row.  A yardstick only; pinned to the host run in test_full_size_engine_vs_torch_reference_one_row (3e-9 relative L2
    on the waveform: four orders below anything measured against it)."""
    from egregora_amd import flashsr_arch as A
    from oracle import flashsr_torch as R
    fb, filt = torch.from_numpy(A.mel_filterbank(cfg)).double().cuda(), torch.from_numpy(A.kaiser_sinc_filter(cfg.aa_taps)).double().cuda()
    with torch.no_grad():
        y = R.flashsr_forward(x.double().cuda(), nz_nchw.double().cuda(), _to_device(R.to_float64(P), "cuda"), cfg, A.unet_blocks(cfg), fb, filt, stages)
    torch.cuda.synchronize()
    return y.cpu()


def test_full_size_engine_vs_torch_reference_one_row(pack):
    """The declared full-size architecture (Winograd / phase-conv / split-K paths active, contractions on the bf16 pipe as exact
    three-way splits) against the PyTorch graph on the host CPU, one row, stage by stage, with a FLOAT64 run of the same graph as
    the yardstick -- the pattern of tests/test_gpu_fatllama.py.  Gates:
      * waveform: LSD(HIP, f64) with the reference's metric (2048/512) <= 1e-3 dB mean AND p95 -- the north star's tolerance, against
        the float64 truth rather than against another fp32 run (measured 1.6e-4 / 2.8e-4 dB; torch fp32 itself: 0.8e-4 / 1.3e-4);
      * every stage: rms(HIP - f64) <= 3 x rms(torch fp32 - f64) and <= 1e-5 of the stage's rms.  Measured ratios: mel 1.06,
        z_cond 1.56, v / z0 1.46, mel_hat 2.46, y 2.10 -- the VAE decoder's Winograd F(4x4,3x3) layers amplify the GEMMs' fp32
        accumulation error by max|M| / max|y| ~ 12 where torch's direct convolutions do not; the absolute
        errors stay at 4e-7 .. 5e-6 of the signal;
      * the previous relative-L2 gates vs torch fp32 (5e-4 per stage, 2e-3 waveform) stay as a coarse net."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    from oracle import flashsr_torch as R, metrics as om
    cfg = A.FlashSRConfig()
    P = A.init_params(cfg, 0)
    e = PyDriverEngine(cfg, P)
    x = 0.2 * torch.randn(1, cfg.chunk, generator=torch.Generator().manual_seed(5))
    nz = e.noise(1, torch.zeros(1, dtype=torch.int64, device="cuda"), 0)
    got_st, want_st, ex_st = {}, {}, {}
    y = e.c_forward(x.cuda(), nz, got_st)                     # through the C ABI (egr_flashsr_forward)
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    fb, filt = torch.from_numpy(A.mel_filterbank(cfg)), torch.from_numpy(A.kaiser_sinc_filter(cfg.aa_taps))
    with torch.no_grad():
        want = R.flashsr_forward(x, nchw(nz.cpu()), P, cfg, A.unet_blocks(cfg), fb, filt, want_st)
        exact = R.flashsr_forward(x.double(), nchw(nz.cpu()).double(), R.to_float64(P), cfg, A.unet_blocks(cfg), fb.double(),
                                  filt.double(), ex_st)
    rms = lambda a: float(a.double().pow(2).mean().sqrt())
    # the float64 yardstick of the multi-row test runs on the GPU (f64_forward_on_gpu): pinned here to the host's float64 run
    exact_gpu = f64_forward_on_gpu(x, nchw(nz.cpu()), P, cfg)
    pin = float((exact_gpu - exact).norm() / exact.norm())
    print(f"\nfloat64 graph on the GPU vs on the host: relative L2 {pin:.2e}")
    assert pin <= 1e-7, pin
    report = {}
    for k in ("mel", "z_cond", "v", "z0", "mel_hat", "y"):
        gt = got_st[k].permute(0, 3, 1, 2).cpu() if got_st[k].dim() == 4 else got_st[k].cpu()
        eg, eo = rms(gt.double() - ex_st[k]), rms(want_st[k].double() - ex_st[k])
        report[k] = (eg, eo, eg / max(eo, 1e-300), rms(ex_st[k]))
    lsd_g = om.lsd_audio(exact.numpy(), y.cpu().numpy())
    lsd_o = om.lsd_audio(exact.numpy(), want.numpy())
    print("\nfull-size row vs float64: stage -> (rms err HIP, rms err torch32, ratio, rms of the stage)")
    for k, v in report.items():
        print(f"  {k:8s} {v[0]:.3e} {v[1]:.3e} ratio {v[2]:.2f}  (signal rms {v[3]:.3e})")
    print(f"  LSD(HIP, f64) mean/p95 = {lsd_g[0]:.3e} / {lsd_g[1]:.3e} dB ; LSD(torch32, f64) = {lsd_o[0]:.3e} / {lsd_o[1]:.3e} dB")
    for k, v in report.items():
        assert v[0] <= 3.0 * v[1] + 1e-9 * v[3] and v[0] <= 1e-5 * v[3], (k, v)
    assert lsd_g[0] <= 1e-3 and lsd_g[1] <= 1e-3, (lsd_g, lsd_o)
    for k in ("mel", "z_cond", "v", "z0", "mel_hat"):
        gt = got_st[k].permute(0, 3, 1, 2)
        assert rel_l2(gt, want_st[k]) <= 5e-4, (k, rel_l2(gt, want_st[k]))
    assert rel_l2(y, want) <= 2e-3, rel_l2(y, want)
    # The same gates on the PRODUCT call: egr_flashsr_infer = two fp16 terms per operand, every batch row scaled from its own maximum
    # on the device (csrc/egr_nn_gemm_s3.hip scheme 1); its first call is its steady state.
    ids = torch.zeros(1, dtype=torch.int64, device="cuda")
    y_h = e.c_infer(x.cuda(), ids, 0)
    assert e.split_info()["enabled"] and not torch.equal(y_h, y)
    assert torch.equal(e.c_infer(x.cuda(), ids, 0), y_h)
    e.set_split("f16x2+forward")
    h_st = {}
    y_hf = e.c_forward(x.cuda(), nz, h_st)
    assert torch.equal(y_hf, y_h)                              # the stage taps below belong to the fp16-term run
    lsd_h = om.lsd_audio(exact.numpy(), y_h.cpu().numpy())
    print("fp16 operand terms (steady state of egr_flashsr_infer) vs float64:")
    for k in ("mel", "z_cond", "v", "z0", "mel_hat", "y"):
        gt = h_st[k].permute(0, 3, 1, 2).cpu() if h_st[k].dim() == 4 else h_st[k].cpu()
        eg = rms(gt.double() - ex_st[k])
        print(f"  {k:8s} {eg:.3e} ratio to torch32 {eg / max(report[k][1], 1e-300):.2f}")
        assert eg <= 3.0 * report[k][1] + 1e-9 * report[k][3] and eg <= 1e-5 * report[k][3], (k, eg, report[k])
    print(f"  LSD(HIP fp16 terms, f64) mean/p95 = {lsd_h[0]:.3e} / {lsd_h[1]:.3e} dB")
    assert lsd_h[0] <= 1e-3 and lsd_h[1] <= 1e-3, lsd_h


def test_full_size_rows_of_different_level_in_one_pass_vs_float64(pack):
    """The product call (egr_flashsr_infer, two fp16 terms per operand) on FOUR full-size rows of one pass at 0 dB, -40 dB, -80 dB
    and digital silence -- a quiet passage batched next to a full-scale one, where a per-tensor operand scale would leave the quiet
    rows on fp16 subnormals (round 3: 1.7e-6 relative on a row at 1e-5 of the maximum, operator level).  Every batch row is scaled
    from its own maximum on the device, so each row must meet the north star's tolerance on its own: LSD against the FLOAT64 run of
    the same graph (reference metric, 2048 / 512) <= 1e-3 dB mean AND p95, through the 29-block UNet, the VAE and the vocoder.
    And the rows do not see each other: row 0 alone gives the bits it has in the mixed pass when the pass has the same row count."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    from oracle import flashsr_torch as R, metrics as om
    cfg = A.FlashSRConfig()
    P = A.init_params(cfg, 0)
    e = E.FlashSREngine(cfg, P)
    assert e.split_info()["enabled"]
    g = torch.Generator().manual_seed(15)
    t = torch.arange(cfg.chunk) / cfg.sr
    base = sum(torch.sin(2 * math.pi * f * t + i) / (i + 1) for i, f in enumerate((110.0, 440.0, 1234.0, 5000.0, 9000.0)))
    base = 0.5 * base / base.abs().max() + 0.02 * torch.randn(cfg.chunk, generator=g)
    x = torch.stack([base, 1e-2 * base, 1e-4 * base, torch.zeros_like(base)]).float()
    ids = torch.tensor([3, 4, 5, 6], dtype=torch.int64, device="cuda")
    y = e.c_infer(x.cuda(), ids, 11).cpu()
    nz = e.noise(4, ids, 11)
    # float64 run of the same graph: on the GPU (4 rows in seconds instead of ~4 minutes of host time; pinned to the host run in
    # test_full_size_engine_vs_torch_reference_one_row)
    exact = f64_forward_on_gpu(x, nchw(nz.cpu()), P, cfg)
    assert bool(torch.isfinite(y).all())
    print()
    for r, name in enumerate(("0 dB", "-40 dB", "-80 dB", "silence")):
        l = om.lsd_audio(exact[r:r + 1].numpy(), y[r:r + 1].numpy())
        rel = float((y[r].double() - exact[r]).norm() / exact[r].norm())
        print(f"  row at {name:8s}: LSD(HIP fp16 terms, f64) mean / p95 = {l[0]:.3e} / {l[1]:.3e} dB, relative L2 {rel:.2e}, output rms {float(exact[r].pow(2).mean().sqrt()):.3e}")
        assert l[0] <= 1e-3 and l[1] <= 1e-3, (name, l)
    # same row count, other rows replaced: row 0 keeps its bits
    x2 = x.clone()
    x2[1:] = 0.3 * torch.randn(3, cfg.chunk, generator=g)
    y2 = e.c_infer(x2.cuda(), ids, 11).cpu()
    assert torch.equal(y2[0], y[0])
    e.close()


def test_bench_shape_26_rows_in_two_row_groups_vs_float64(pack):
    """The FlashSR stage of the headline bench at ITS OWN shape: 26 full-size rows (60 s stereo = 13 chunks x 2 channels) through
    egr_flashsr_infer, which runs them as two concurrent 13-row groups on the handle's side streams (own scratch arena each) -- every
    row against the float64 run of the same graph (f64_forward_on_gpu), LSD <= 1e-3 dB mean and p95 per row.  Rows at levels from
    0 to -60 dB, different material per row."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    from oracle import metrics as om
    cfg = A.FlashSRConfig()
    P = A.init_params(cfg, 0)
    e = E.FlashSREngine(cfg, P)
    R_ = 26
    g = torch.Generator().manual_seed(26)
    t = torch.arange(cfg.chunk) / cfg.sr
    rows_ = []
    for r in range(R_):
        f0 = 80.0 * (1.0 + 0.37 * r)
        tone = sum(torch.sin(2 * math.pi * f0 * (k + 1) * t + r + k) / (k + 1) for k in range(5))
        rows_.append((10.0 ** (-(r % 7) / 2.0)) * (0.5 * tone / tone.abs().max() + 0.03 * torch.randn(cfg.chunk, generator=g)))
    x = torch.stack(rows_).float()
    ids = torch.arange(R_, dtype=torch.int64, device="cuda")
    y = e.c_infer(x.cuda(), ids, 3).cpu()
    assert bool(torch.isfinite(y).all())
    nz = e.noise(R_, ids, 3)
    worst = (0.0, 0.0)
    for lo in range(0, R_, 13):                 # the float64 graph in two halves (its im2col buffers are 8-byte)
        exact = f64_forward_on_gpu(x[lo:lo + 13], nchw(nz[lo:lo + 13].cpu()), P, cfg)
        for r in range(exact.shape[0]):
            l = om.lsd_audio(exact[r:r + 1].numpy(), y[lo + r:lo + r + 1].numpy())
            worst = (max(worst[0], l[0]), max(worst[1], l[1]))
            assert l[0] <= 1e-3 and l[1] <= 1e-3, (lo + r, l)
    print(f"\n26 rows, two 13-row groups: worst row LSD vs float64 mean / p95 = {worst[0]:.3e} / {worst[1]:.3e} dB")
    e.close()


def test_full_size_engine_shapes_and_determinism(pack):
    """Declared full-size architecture with synthetic weights: one row, shapes + finite output + same-seed repeatability."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.FlashSRConfig()
    e = PyDriverEngine(cfg, A.init_params(cfg, 0))
    x = 0.2 * torch.randn(1, cfg.chunk, generator=torch.Generator().manual_seed(5))
    ids = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = {}
    y1 = e.forward_rows(x.cuda(), e.noise(1, ids, 0), st)
    y2 = e.forward_rows(x.cuda(), e.noise(1, ids, 0))
    torch.cuda.synchronize()
    assert y1.shape == (1, cfg.chunk) and torch.isfinite(y1).all()
    assert st["mel"].shape == (1, 512, 256, 1) and st["z_cond"].shape == (1, 64, 32, 16) and st["mel_hat"].shape == (1, 512, 256, 1)
    assert torch.equal(y1, y2)
    fl = e.flop_count(1)
    assert 1e12 < fl < 1e13


def test_flashsr_min_cli_runs_the_node(pack, eng, tmp_path, monkeypatch):
    """The flashsr_min-shaped CLI reads a WAV, runs the upscaler node on the device and writes a PCM_16 WAV."""
    e, cfg, P = eng
    from egregora_amd import audio_glue as ag, flashsr_engine as E, flashsr_min, wavio
    x = (0.3 * np.sin(2 * np.pi * 440 * np.arange(9000) / 48000)).astype(np.float32)
    wavio.write_wav_pcm16(str(tmp_path / "in.wav"), np.stack([x, 0.5 * x], 1), 48000)
    monkeypatch.setattr(ag, "CHUNK_SAMPLES", cfg.chunk)
    monkeypatch.setattr(ag, "HOP_SAMPLES", cfg.chunk - 375)
    E.set_engine(e)
    try:
        flashsr_min.main(["--ckpt-dir", str(tmp_path), "--in", str(tmp_path / "in.wav"), "--out", str(tmp_path / "out.wav")])
    finally:
        E.set_engine(None)
    y, sr = wavio.read_wav(str(tmp_path / "out.wav"))
    assert sr == 48000 and y.shape == (9000, 2) and np.isfinite(y).all() and float(np.abs(y).max()) > 0


def test_node_run_with_rate_conversion_both_sides(pack, eng, monkeypatch):
    """EgregoraAudioSuperResolution.run end to end on the device: 44.1 kHz stereo in -> polyphase to 48 kHz -> chunked
    engine + WOLA -> polyphase to 96 kHz, against the oracle glue (scipy resample_poly, torch reference graph, numpy WOLA)."""
    e, cfg, P = eng
    from egregora_amd import audio_glue as ag, flashsr_arch as A, flashsr_engine as E
    from oracle import flashsr_torch as R, glue as og
    win, hop = cfg.chunk, cfg.chunk - 375
    monkeypatch.setattr(ag, "CHUNK_SAMPLES", win)
    monkeypatch.setattr(ag, "HOP_SAMPLES", hop)
    g = torch.Generator().manual_seed(8)
    x = 0.3 * torch.randn(1, 2, 7000, generator=g)
    E.set_engine(e)
    try:
        node = pack.NODE_CLASS_MAPPINGS["EgregoraAudioUpscaler"]()
        (res,) = node.run(audio={"waveform": x, "sample_rate": 44100}, lowpass_input=False, output_sr="96000")
    finally:
        E.set_engine(None)
    assert res["sample_rate"] == 96000 and res["waveform"].dtype == torch.float32 and res["waveform"].dim() == 3
    fb, filt = torch.from_numpy(A.mel_filterbank(cfg)), torch.from_numpy(A.kaiser_sinc_filter(cfg.aa_taps))
    x48 = og.resample_hq(x[0].numpy(), 44100, 48000)
    idx = [0]

    def model(c):
        k = idx[0]; idx[0] += 1
        ids = torch.tensor([k * 2, k * 2 + 1], dtype=torch.int64, device="cuda")
        nz = nchw(e.noise(2, ids, E.SEED).cpu())
        with torch.no_grad():
            return R.flashsr_forward(torch.from_numpy(c), nz, P, cfg, A.unet_blocks(cfg), fb, filt).numpy()
    y48 = og.flashsr_node_glue(x48, model, win, hop)
    want = og.resample_hq(y48, 48000, 96000)
    got = res["waveform"][0].numpy()
    assert got.shape == want.shape
    assert float(np.abs(got - want).max()) <= 3e-3 * float(np.abs(want).max())


def test_default_engine_within_north_star_lsd_of_the_strict_f32_engine(pack):
    """Full-size FlashSR forward: the default engine (contractions as exact three-way bf16 splits on the bf16 matrix pipe,
    Winograd F(4x4,3x3), z-streamed GEMMs, input-stationary 1-D convs, per-tap conv_out, vector-ALU conv_in) against the strict engine of the same layer table
    (v_mfma_f32_32x32x2_f32 everywhere, no Winograd).  Bar = the north star's tolerance: LSD <= 1e-3 dB with the reference's
    own metric (measured 2.6e-4 mean / 4.7e-4 p95), and every stage within 5e-5 relative L2 (measured <= 8.5e-6)."""
    from egregora_amd import device_ops, flashsr_arch as A, flashsr_engine as E
    cfg = A.FlashSRConfig()
    P = A.init_params(cfg, 0)
    old = (E.FlashSREngine.MFMA_MODE, E.FlashSREngine.WINO_MIN_CH, E.FlashSREngine.THIN_ENDS)
    try:
        e_fast = PyDriverEngine(cfg, P)
        E.FlashSREngine.MFMA_MODE, E.FlashSREngine.WINO_MIN_CH, E.FlashSREngine.THIN_ENDS = "f32", 1 << 30, False
        e_ref = PyDriverEngine(cfg, P)
    finally:
        E.FlashSREngine.MFMA_MODE, E.FlashSREngine.WINO_MIN_CH, E.FlashSREngine.THIN_ENDS = old
    assert e_fast.w3 and not e_ref.w3 and not any(k.endswith(".wino4") or k.endswith(".taps") for k in e_ref.w)
    assert any(k.endswith(".taps") for k in e_fast.w)
    rng = np.random.Generator(np.random.PCG64(202))
    t = np.arange(cfg.chunk) / 48000.0
    x = sum(np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) / (k + 1) for k, f in enumerate(np.geomspace(80, 6000, 8)))
    x = x + 0.01 * rng.standard_normal(cfg.chunk)
    x = torch.from_numpy((0.5 * x / np.abs(x).max()).astype(np.float32))[None].repeat(2, 1).cuda()
    nz = e_fast.noise(2, None, 7)
    sa, sb = {}, {}
    ya = e_fast.forward_rows(x, nz, stages=sa)
    yb = e_ref.forward_rows(x, nz, stages=sb)
    for k in ("mel", "z_cond", "v", "z0", "mel_hat", "y"):
        assert rel_l2(sa[k], sb[k]) <= 5e-5, (k, rel_l2(sa[k], sb[k]))
    mean, p95 = device_ops.lsd(ya[:1].contiguous(), yb[:1].contiguous())
    assert mean <= 1e-3 and p95 <= 1e-3, (mean, p95)
    assert device_ops.si_sdr(yb[:1].contiguous(), ya[:1].contiguous()) >= 90.0
