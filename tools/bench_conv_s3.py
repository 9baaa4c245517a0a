"""Dev A/B: egr_conv_nhwc (f32 MFMA) vs egr_conv_s3 (3-way bf16 split on the bf16 MFMA): error vs float64 and speed."""
import sys; sys.path.insert(0, '.')
import ctypes as C, torch
from packload import load_pack; load_pack()
from egregora_amd import native
L = native.lib()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = native.stream_ptr

def pack(w2):
    K, Co = w2.shape; Kp = (K + 15) // 16 * 16
    if Kp != K: w2 = torch.cat([w2, w2.new_zeros(Kp - K, Co)], 0)
    return w2.reshape(Kp // 16, 16, Co).permute(0, 2, 1).contiguous()

def split3(wp):
    ns, Co, _ = wp.shape
    w3 = torch.empty(ns * 3 * Co * 16, dtype=torch.bfloat16, device='cuda')
    native.check(L.egr_split3_pack(p(wp), p(w3), ns, Co, st()), "split3")
    return w3

def conv_f32(x, wp, b, y, B, H, W, Ci, Co, k, dil=1):
    native.check(L.egr_conv_nhwc(p(x), p(wp), p(b), p(None), p(None), p(y), B, H, W, Ci, H, W, Co, k, k, 1, dil, k // 2, k // 2, 0, 0, 0.0, st()), "conv")

def conv_s3(x, w3, b, y, B, H, W, Ci, Co, k, dil=1):
    native.check(L.egr_conv_s3(p(x), p(w3), p(b), p(None), p(None), p(y), B, H, W, Ci, H, W, Co, k, k, 1, dil, k // 2, k // 2, 0, 0, 0.0,
                               1, 1, 0, 0, H, W, 1, 0, 0, 0, st()), "conv_s3")

# ---- accuracy vs float64
torch.manual_seed(0)
for (B, H, W, Ci, Co, k) in [(2, 16, 12, 128, 128, 3), (1, 9, 7, 256, 96, 3), (3, 8, 8, 512, 40, 1), (2, 10, 6, 64, 200, 3)]:
    x = torch.randn(B, H, W, Ci, device='cuda'); w = torch.randn(Co, Ci, k, k, device='cuda') / (Ci * k * k) ** 0.5; b = torch.randn(Co, device='cuda')
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
    wp = pack(w.permute(2, 3, 1, 0).reshape(k * k * Ci, Co).contiguous()); w3 = split3(wp)
    y1 = torch.empty(B, H, W, Co, device='cuda'); y2 = torch.empty_like(y1)
    conv_f32(x, wp, b, y1, B, H, W, Ci, Co, k); conv_s3(x, w3, b, y2, B, H, W, Ci, Co, k)
    torch.cuda.synchronize()
    e = lambda y: ((y.double() - ref).abs().max() / ref.abs().max()).item()
    r = lambda y: ((y.double() - ref).norm() / ref.norm()).item()
    print(f"B{B} {H}x{W} Ci{Ci} Co{Co} k{k}: f32-mfma max {e(y1):.2e} rms {r(y1):.2e} | s3 max {e(y2):.2e} rms {r(y2):.2e}")

SHAPES = {  # name: (B,H,W,Cin,Cout,k)
    "s1_lat1024": (26, 64, 32, 1024, 1024, 3), "s2_l0_128": (26, 512, 256, 128, 128, 3), "s3_512": (26, 128, 64, 512, 512, 3),
    "s4_256": (26, 256, 128, 256, 256, 3), "s5_1x1_1024": (26, 64, 32, 1024, 1024, 1), "s6_unet_small": (26, 8, 4, 640, 640, 3),
    "s7_co64": (26, 512, 256, 128, 64, 3), "s8_co32": (26, 2048, 64, 64, 32, 3),
}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for n in names:
    B, H, W, Ci, Co, k = SHAPES[n]
    x = torch.randn(B, H, W, Ci, device='cuda'); wp = torch.randn((k * k * Ci + 15) // 16, Co, 16, device='cuda') / (Ci * k * k) ** 0.5
    b = torch.randn(Co, device='cuda'); y = torch.empty(B, H, W, Co, device='cuda'); w3 = split3(wp)
    fl = 2.0 * B * H * W * Co * k * k * Ci
    out = []
    for f, wt in ((conv_f32, wp), (conv_s3, w3)):
        f(x, wt, b, y, B, H, W, Ci, Co, k); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f(x, wt, b, y, B, H, W, Ci, Co, k)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out.append(f"{ms:8.3f} ms {fl / ms / 1e9:7.1f} TF/s")
    print(f"{n:14s} f32: {out[0]}   s3: {out[1]}")
