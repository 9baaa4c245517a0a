"""Dev A/B of egr_snake_aa variants on the vocoder shapes (EGR_SNAKE=tiled selects the LDS-tiled kernel)."""
import sys; sys.path.insert(0, '.')
import ctypes as C, torch
from packload import load_pack; load_pack()
from egregora_amd import native, flashsr_arch as A
L_ = native.lib(); p = lambda t: C.c_void_p(t.data_ptr())
filt = torch.from_numpy(A.kaiser_sinc_filter(12)).cuda()
for (B, L, Cc) in [(26, 61440, 64), (26, 122880, 32), (26, 245760, 16), (26, 15360, 128), (26, 3072, 256), (3, 1000, 48)]:
    x = torch.randn(B, L, Cc, device='cuda'); al = 0.1 * torch.randn(Cc, device='cuda'); be = 0.1 * torch.randn(Cc, device='cuda'); y = torch.empty_like(x)
    ra = torch.zeros(B * 32, device='cuda')
    import os
    if os.environ.get("SNAKE_RA", "1") == "1":
        run = lambda: native.check(L_.egr_snake_aa_ra(p(x), p(al), p(be), p(filt), p(y), B, L, Cc, 12, p(ra), native.stream_ptr()), "snake")
    else:
        run = lambda: native.check(L_.egr_snake_aa(p(x), p(al), p(be), p(filt), p(y), B, L, Cc, 12, native.stream_ptr()), "snake")
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
    print(f"B{B} L{L} C{Cc}: {ms:.3f} ms  {2 * x.numel() * 4 / ms / 1e6:.0f} GB/s  checksum {float(y.double().sum()):.6f} {float(y.abs().max()):.5f}")
