"""The fused AMP unit of the vocoder's 16-channel stage (csrc/egr_nn_amp.hip, egr_amp_unit_h2; round 6 removed the slower 32-channel instantiation):
    y = conv2(snake2(conv1(snake1(x)))) + x
against (a) the float64 composite of oracle.flashsr_torch (_act_aa + F.conv1d: what the upstream vocoder's AMP unit computes as the
reference reaches it through FlashSR.__call__, /root/reference/egregora_audio_super_resolution.py:361-369) and (b) the four launches it
replaces (egr_snake_aa, egr_conv_h2 on the 1-D kernel): the fused kernel must be no further from float64 than 1.5x the unfused chain
(both carry two fp16 terms per operand; the fused one scales per TILE from a bound, the chain per batch row from the measured maximum)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def pack_matrix(w2):
    K, Co = w2.shape
    Kp = ((K + 15) // 16) * 16
    if Kp != K:
        w2 = torch.cat([w2, w2.new_zeros(Kp - K, Co)], 0)
    return w2.reshape(Kp // 16, 16, Co).permute(0, 2, 1).contiguous()


def h2_pack(L, wp, Co, st):
    from egregora_amd import native
    ns = wp.shape[0]
    wmax = float(wp.abs().max())
    ws = 2.0 ** (13 - math.ceil(math.log2(wmax)))
    w2 = torch.empty(ns * 2 * Co * 16, dtype=torch.float16, device="cuda")
    native.check(L.egr_split2h_pack(p(wp), p(w2), ns, Co, ws, st), "split2h")
    return w2, ws


def ref64(x, prm, k, d, filt):
    from oracle import flashsr_torch as R
    xd = x.double().permute(0, 2, 1).contiguous()                      # [B, C, L]
    f = filt.double()
    t = R._act_aa(xd, prm["a1"].double(), prm["b1"].double(), f)
    t = F.conv1d(t, prm["w1"].double(), prm["c1"].double(), dilation=d, padding=d * (k - 1) // 2)
    t = R._act_aa(t, prm["a2"].double(), prm["b2"].double(), f)
    t = F.conv1d(t, prm["w2"].double(), prm["c2"].double(), padding=(k - 1) // 2)
    return (t + xd).permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("Cc,k,d,L,big", [(16, 3, 1, 1000, 0), (16, 7, 3, 4133, 0), (16, 11, 5, 2048, 0), (16, 11, 1, 300, 0), (16, 7, 5, 256, 0), (16, 3, 3, 17, 0),
                                          (16, 11, 5, 500, 0), (16, 3, 5, 16, 0),
                                          # snake arguments far outside v_sin_f32's +-256 revolutions (alpha ~ 4, |x| ~ 50: ~1800 revolutions);
                                          # without the range reduction the fused unit dropped the sin^2 term there and the four launches did not
                                          (16, 7, 3, 1000, 1), (16, 3, 1, 333, 1)])
def test_fused_amp_unit_vs_float64_and_vs_the_four_launches(pack, Cc, k, d, L, big):
    from egregora_amd import flashsr_arch as A, native
    lib = native.lib()
    native.require_device()
    st = native.stream_ptr()
    B = 4
    g = torch.Generator().manual_seed(100 * k + d)
    x = torch.randn(B, L, Cc, generator=g) * (10.0 ** -torch.arange(B).float()).view(B, 1, 1)           # rows at 0 / -20 / -40 / -60 dB
    if big:
        x = x * 50.0
    prm = {"a1": (4.0 if big else 0.0) + 0.3 * torch.randn(Cc, generator=g), "b1": 0.3 * torch.randn(Cc, generator=g), "a2": 0.3 * torch.randn(Cc, generator=g),
           "b2": 0.3 * torch.randn(Cc, generator=g), "w1": torch.randn(Cc, Cc, k, generator=g) / math.sqrt(Cc * k), "c1": 0.1 * torch.randn(Cc, generator=g),
           "w2": torch.randn(Cc, Cc, k, generator=g) / math.sqrt(Cc * k), "c2": 0.1 * torch.randn(Cc, generator=g)}
    filt = torch.from_numpy(A.kaiser_sinc_filter(12))
    want = ref64(x, prm, k, d, filt)
    dev = {n: v.cuda().contiguous() for n, v in prm.items()}
    xg, fg = x.cuda().contiguous(), filt.cuda().contiguous()
    packs = {}
    for n in ("w1", "w2"):
        wp = pack_matrix(prm[n].permute(2, 1, 0).reshape(k * Cc, Cc).contiguous()).cuda()               # K ordered (tap, ci)
        packs[n] = (wp,) + h2_pack(lib, wp, Cc, st)
    y = torch.full((B, L, Cc), float("nan"), device="cuda")
    native.check(lib.egr_amp_unit_h2(p(xg), p(y), B, L, Cc, k, d, p(dev["a1"]), p(dev["b1"]), p(packs["w1"][1]), packs["w1"][2], p(dev["c1"]),
                                     p(dev["a2"]), p(dev["b2"]), p(packs["w2"][1]), packs["w2"][2], p(dev["c2"]), p(fg), 12, st), "egr_amp_unit_h2")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    # the four launches it replaces (when the 1-D kernel takes the shape: L % 128 == 0), else the bf16-term convolution
    RA = 32
    s1 = torch.empty_like(xg); ra1 = torch.zeros(B * RA, device="cuda")
    native.check(lib.egr_snake_aa_ra(p(xg), p(dev["a1"]), p(dev["b1"]), p(fg), p(s1), B, L, Cc, 12, p(ra1), st), "snake1")
    c1 = torch.empty_like(xg)
    native.check(lib.egr_conv_nhwc(p(s1), p(packs["w1"][0]), p(dev["c1"]), p(None), p(None), p(c1), B, 1, L, Cc, 1, L, Cc, 1, k, 1, d, 0, d * (k - 1) // 2, 0, 0, 0.0, st), "conv1")
    s2 = torch.empty_like(xg); ra2 = torch.zeros(B * RA, device="cuda")
    native.check(lib.egr_snake_aa_ra(p(c1), p(dev["a2"]), p(dev["b2"]), p(fg), p(s2), B, L, Cc, 12, p(ra2), st), "snake2")
    y4 = torch.empty_like(xg)
    native.check(lib.egr_conv_nhwc(p(s2), p(packs["w2"][0]), p(dev["c2"]), p(None), p(xg), p(y4), B, 1, L, Cc, 1, L, Cc, 1, k, 1, 1, 0, (k - 1) // 2, 0, 0, 0.0, st), "conv2")
    torch.cuda.synchronize()
    for i in range(B):
        ref = want[i]
        e_f = float((y[i].double().cpu() - ref).norm() / ref.norm())
        e_4 = float((y4[i].double().cpu() - ref).norm() / ref.norm())
        m_f = float((y[i].double().cpu() - ref).abs().max() / ref.abs().max())
        print(f"C {Cc} k {k} d {d} L {L} row {i}: fused rel L2 {e_f:.2e} (max {m_f:.2e} of the peak), four fp32-grade launches {e_4:.2e}")
        # relative to the launches it replaces AND absolute (measured 0.7 - 2.7e-7).  With huge snake arguments the float32 product u * alpha
        # itself is good to 1e-4 revolutions (ANY float32 implementation: the chain's e_4 shows it), so there only the relative gate and
        # "the sin^2 term is there" (a dropped term is an error of order 1 / beta) apply
        assert e_f <= 1.5 * e_4 + 3e-7 and (e_f <= 5e-7 or (big and e_f <= 2e-3)), (Cc, k, d, L, i, e_f, e_4)
    # a row's result depends on that row alone
    y1 = torch.empty(1, L, Cc, device="cuda")
    native.check(lib.egr_amp_unit_h2(p(xg[2:3].contiguous()), p(y1), 1, L, Cc, k, d, p(dev["a1"]), p(dev["b1"]), p(packs["w1"][1]), packs["w1"][2], p(dev["c1"]),
                                     p(dev["a2"]), p(dev["b2"]), p(packs["w2"][1]), packs["w2"][2], p(dev["c2"]), p(fg), 12, st), "egr_amp_unit_h2")
    assert torch.equal(y1[0], y[2])


def test_fused_amp_unit_refuses_other_shapes(pack):
    from egregora_amd import native
    lib = native.lib()
    x = torch.zeros(1, 64, 64, device="cuda"); y = torch.zeros_like(x); v = torch.zeros(64, device="cuda"); w = torch.zeros(1 << 16, dtype=torch.float16, device="cuda")
    rc = lib.egr_amp_unit_h2(p(x), p(y), 1, 64, 64, 3, 1, p(v), p(v), p(w), 1.0, p(v), p(v), p(v), p(w), 1.0, p(v), p(v), 12, native.stream_ptr())
    assert rc == 3 and "qualify" in native.last_error()
