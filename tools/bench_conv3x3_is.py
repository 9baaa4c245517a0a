"""dev: the 128-channel level's 3x3 convolution with fused GroupNorm + SiLU (egr_conv_h2_gn) at the 26-row shape of the bench --
A/B of the input-stationary kernels: EGR_S3_CONV3X3=1 (round 4: halo phase between barriers) vs default (k_conv3x3_isp)."""
import math, os, sys
sys.path.insert(0, '.')
import ctypes as C
import torch
from packload import load_pack; load_pack()
from egregora_amd import native
L = native.lib()
native.require_device()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
RA = 32
B, H, W, Ci, Co = [int(v) for v in os.environ.get("SHAPE", "26,512,256,128,128").split(",")]
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(B, H, W, Ci, device="cuda", generator=g)
if os.environ.get("ZERO"):
    x.zero_()
sc = 1.0 + 0.2 * torch.randn(B, Ci, device="cuda", generator=g)
sh = 0.3 * torch.randn(B, Ci, device="cuda", generator=g)
w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) / math.sqrt(Ci * 9)
K = 9 * Ci
wm = w.permute(2, 3, 1, 0).reshape(K, Co).contiguous()
wp = wm.view(K // 16, 16, Co).permute(0, 2, 1).contiguous()          # slab-major [K/16][Cout][16]
w2 = torch.empty((K // 16) * 2 * Co * 16, dtype=torch.float16, device="cuda")
ws = 8192.0
native.check(L.egr_split2h_pack(p(wp), p(w2), K // 16, Co, ws, native.stream_ptr()), "split2h")
xa = torch.zeros(B * RA, device="cuda")
native.check(L.egr_absmax_rows(p(x), B, H * W * Ci, 1, 0, p(xa), native.stream_ptr()), "absmax")
bound = torch.zeros(B * RA, device="cuda")
native.check(L.egr_gn_operand_bound(p(sc), p(sh), B, Ci, p(xa), p(bound), native.stream_ptr()), "bound")
y = torch.empty(B, H, W, Co, device="cuda")
oa = torch.zeros(B * RA, device="cuda")
part = torch.zeros(B * H * W // 32, Co // 4, 2, device="cuda")
def run():
    native.check(L.egr_conv_h2_gn(p(x), p(sc), p(sh), 1, p(w2), p(None), p(None), p(y), B, H, W, Ci, Co, 0, ws, p(bound), p(oa), p(part),
                                  native.stream_ptr()), "conv_h2_gn")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("REPS", "5"))
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * H * W * Co * K
print(f"EGR_C3_BREG={os.environ.get('EGR_C3_BREG', '')!r}: {ms:.3f} ms per launch, {fl / ms / 1e9:.1f} TFLOP/s fp32-equivalent, "
      f"{3 * fl / ms / 1e9:.0f} executed f16; checksum {float(y.double().sum()):.6e} {float(y.double().abs().max()):.6e}")
