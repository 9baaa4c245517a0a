"""Dev probe: full-size FlashSR stage (26 rows = 60 s stereo) through egr_flashsr_infer in both operand schemes: time per call,
agreement, per-kernel HIP-event totals (one row group)."""
import sys, time; sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from oracle import metrics as OM
import numpy as np
cfg = A.FlashSRConfig(); P = A.init_params(cfg, 0)
e = E.FlashSREngine(cfg, P)
R = 26
x = 0.2 * torch.randn(R, cfg.chunk, device='cuda')
def sync(): torch.cuda.synchronize()
def timed(n=3):
    sync(); t = time.time()
    for _ in range(n): y = e.c_infer(x, None, 1)
    sync(); return y, (time.time() - t) / n * 1e3
t0 = time.time(); y0 = e.c_infer(x, None, 1); sync()          # call 1: allocates the scratch arena
print("first call %.1f ms" % ((time.time() - t0) * 1e3), e.split_info())
yh, th = timed()
print("f16x2: %.1f ms per call" % th, e.split_info())
import os
if os.environ.get("PROBE_QUICK"):
    sys.exit(0)
prof_h = e.c_profile(lambda: e.c_infer(x, None, 1))
e.set_split("bf16x3")
yb, tb = timed()
print("bf16x3: %.1f ms per call" % tb)
prof_b = e.c_profile(lambda: e.c_infer(x, None, 1))
print("bit-equal(call 1, later calls):", bool(torch.equal(y0, yh)))
rel = float((yh - yb).double().norm() / yb.double().norm())
print("rel L2 (f16x2 vs bf16x3) %.3e  max %.3e of peak" % (rel, float((yh - yb).abs().max() / yb.abs().max())))
l = [OM.lsd_audio(yh[i].cpu().numpy()[None], yb[i].cpu().numpy()[None]) for i in range(4)]
print("LSD(f16x2, bf16x3) dB:", l)
for name, pr in (("f16x2", prof_h), ("bf16x3", prof_b)):
    tot = sum(v[2] for v in pr.values())
    print(name, "contraction launches total %.1f ms" % tot)
    for k, v in sorted(pr.items(), key=lambda kv: -kv[1][2])[:12]:
        print("   %-44s n=%4d  %.2f ms  %.1f TF/s-eq" % (k, v[0], v[2], v[1] / v[2] / 1e9 if v[2] else 0))
