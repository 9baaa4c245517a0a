"""Dev timing of the F(4x4,3x3) transform kernels on the FlashSR shapes."""
import sys; sys.path.insert(0, '.')
import ctypes as C, torch
from packload import load_pack; load_pack()
from egregora_amd import native
L = native.lib(); p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
for (B, H, W, Cc) in [(26, 512, 256, 128), (26, 256, 128, 256), (26, 128, 64, 512), (26, 64, 32, 1024)]:
    x = torch.randn(B, H, W, Cc, device='cuda'); P = B * (H // 4) * (W // 4)
    V = torch.empty(36, P, Cc, device='cuda'); y = torch.empty_like(x)
    sc = torch.randn(B, Cc, device='cuda'); sh = torch.randn(B, Cc, device='cuda')
    def t(fn):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 10
    ti = t(lambda: native.check(L.egr_winograd4_input(p(x), p(sc), p(sh), 1, B, H, W, Cc, p(V), native.stream_ptr()), "in"))
    to = t(lambda: native.check(L.egr_winograd4_output(p(V), p(None), p(x), p(y), B, H, W, Cc, 0, native.stream_ptr()), "out"))
    gb = x.numel() * 4 / 1e9
    print(f"B{B} {H}x{W} C{Cc}: in {ti:.3f} ms ({gb * 3.25 / ti:.2f} TB/s)  out {to:.3f} ms ({gb * 4.25 / to:.2f} TB/s)")
