"""Dev probe: full-size FlashSR forward, default engine (bf16x3 contractions + Winograd F(4x4)) vs the strict engine
(f32 MFMA, no Winograd, EGREGORA_FLASHSR_MFMA=f32 / WINOGRAD_MIN_CH huge): relative L2 per stage and LSD of the waveforms."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E, device_ops
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig(); P = A.init_params(cfg, 0)
e_fast = PyDriverEngine(cfg, P)
E.FlashSREngine.MFMA_MODE = "f32"; E.FlashSREngine.WINO_MIN_CH = 1 << 30
e_ref = PyDriverEngine(cfg, P)
rng = np.random.Generator(np.random.PCG64(202)); t = np.arange(cfg.chunk) / 48000.0
x = sum(np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) / (k + 1) for k, f in enumerate(np.geomspace(80, 6000, 8))) + 0.01 * rng.standard_normal(cfg.chunk)
x = torch.from_numpy((0.5 * x / np.abs(x).max()).astype(np.float32))[None].repeat(2, 1).cuda()
nz = e_fast.noise(2, None, 7)
sa, sb = {}, {}
ya = e_fast.forward_rows(x, nz, stages=sa); yb = e_ref.forward_rows(x, nz, stages=sb)
for k in ("mel", "z_cond", "v", "z0", "mel_hat", "y"):
    a, b = sa[k].double(), sb[k].double()
    print(f"{k:8s} rel L2 {float((a - b).norm() / b.norm()):.2e}  max/|max| {float((a - b).abs().max() / b.abs().max()):.2e}")
print("LSD(mean, p95) dB:", device_ops.lsd(ya[:1].contiguous(), yb[:1].contiguous()))
print("SI-SDR dB:", device_ops.si_sdr(yb[:1].contiguous(), ya[:1].contiguous()))
