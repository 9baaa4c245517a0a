"""Dev timing of egr_groupnorm_coeff on the FlashSR shapes."""
import sys; sys.path.insert(0, '.')
import ctypes as C, torch
from packload import load_pack; load_pack()
from egregora_amd import native
L = native.lib(); p = lambda t: C.c_void_p(t.data_ptr())
for (B, HW, Cc) in [(26, 512 * 256, 128), (26, 256 * 128, 256), (26, 128 * 64, 512), (26, 64 * 32, 1024), (26, 32 * 16, 384), (26, 8 * 4, 640), (3, 77, 48)]:
    x = torch.randn(B, HW, Cc, device='cuda') * 2 + 0.3; ga = torch.randn(Cc, device='cuda'); be = torch.randn(Cc, device='cuda')
    ws = torch.empty(int(L.egr_groupnorm_workspace_bytes(B, Cc, 32 if Cc % 32 == 0 else 8)) + 1024, dtype=torch.uint8, device='cuda')
    G = 32 if Cc % 32 == 0 else 8
    sc = torch.empty(B, Cc, device='cuda'); sh = torch.empty(B, Cc, device='cuda')
    run = lambda: native.check(L.egr_groupnorm_coeff(p(x), p(ga), p(be), B, HW, Cc, G, 1e-6, p(ws), p(sc), p(sh), native.stream_ptr()), "gn")
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
    xg = x.double().view(B, HW, G, Cc // G); mean = xg.mean(dim=(1, 3)); var = xg.var(dim=(1, 3), unbiased=False)
    want_sc = (ga.double().view(1, G, -1) / (var + 1e-6).sqrt().unsqueeze(-1)).reshape(B, Cc)
    err = float((sc.double() - want_sc).abs().max() / want_sc.abs().max())
    print(f"B{B} HW{HW} C{Cc}: {ms:.3f} ms  {x.numel() * 4 / ms / 1e6:.0f} GB/s  scale err {err:.1e}")
