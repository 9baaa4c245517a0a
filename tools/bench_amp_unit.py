"""dev: the fused AMP unit (egr_amp_unit_h2) at the 16-channel stage's size against the four launches it replaces (egr_snake_aa_ra + egr_conv_h2)."""
import math, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ctypes as C
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, native
from test_gpu_amp_unit import pack_matrix, h2_pack, p
L_ = native.lib(); native.require_device(); st = native.stream_ptr()
B, L, Cc = 26, int(os.environ.get("AMP_L", "245760")), int(os.environ.get("AMP_C", "16"))
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, L, Cc, device="cuda", generator=g)
fg = torch.from_numpy(A.kaiser_sinc_filter(12)).cuda()
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for k, d in ((3, 1), (7, 3), (11, 5)):
    v = [0.3 * torch.randn(Cc, device="cuda", generator=g) for _ in range(6)]
    packs = []
    for _ in range(2):
        w = torch.randn(Cc, Cc, k, device="cuda", generator=g) / math.sqrt(Cc * k)
        wp = pack_matrix(w.permute(2, 1, 0).reshape(k * Cc, Cc).contiguous())
        packs.append((wp,) + h2_pack(L_, wp, Cc, st))
    y = torch.empty_like(x)
    def fused():
        native.check(L_.egr_amp_unit_h2(p(x), p(y), B, L, Cc, k, d, p(v[0]), p(v[1]), p(packs[0][1]), packs[0][2], p(v[2]), p(v[3]), p(v[4]), p(packs[1][1]), packs[1][2], p(v[5]),
                                        p(fg), 12, st), "amp")
    s1, c1, s2, y4 = (torch.empty_like(x) for _ in range(4))
    RA = 32
    ra = [torch.zeros(B * RA, device="cuda") for _ in range(2)]
    def four():
        native.check(L_.egr_snake_aa_ra(p(x), p(v[0]), p(v[1]), p(fg), p(s1), B, L, Cc, 12, p(ra[0]), st), "snake1")
        native.check(L_.egr_conv_h2(p(s1), p(packs[0][1]), p(v[2]), p(None), p(None), p(c1), B, 1, L, Cc, 1, L, Cc, 1, k, 1, d, 0, d * (k - 1) // 2, 0, 0, 0.0, 1, 1, 0, 0, 1, L, 1, 0, 0, 0,
                                    packs[0][2], p(ra[0]), B, p(None), st), "conv1")
        native.check(L_.egr_snake_aa_ra(p(c1), p(v[3]), p(v[4]), p(fg), p(s2), B, L, Cc, 12, p(ra[1]), st), "snake2")
        native.check(L_.egr_conv_h2(p(s2), p(packs[1][1]), p(v[5]), p(None), p(x), p(y4), B, 1, L, Cc, 1, L, Cc, 1, k, 1, 1, 0, (k - 1) // 2, 0, 0, 0.0, 1, 1, 0, 0, 1, L, 1, 0, 0, 0,
                                    packs[1][2], p(ra[1]), B, p(None), st), "conv2")
    tf = timed(fused)
    try:
        t4 = timed(four)
        rel = float((y - y4).double().norm() / y4.double().norm())
    except Exception as ex:      # noqa: BLE001
        t4, rel = float("nan"), str(ex)[:80]
    T = B * L * Cc * 4 / 1e9
    print(f"EGR_AMP_TL={os.environ.get('EGR_AMP_TL', '')!r} C {Cc} k {k} d {d}: fused {tf:.3f} ms ({2 * T / tf:.2f} TB/s of x + y), four launches {t4:.3f} ms; fused vs four rel L2 {rel}")
