"""Dev probe: time of the VAE's thin-end convs (conv_in 1 -> 128, conv_out 128 -> 1) at 26 rows, new paths vs the MFMA tiles."""
import sys; sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig(); P = A.init_params(cfg, 0)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for thin in (True, False):
    E.FlashSREngine.THIN_ENDS = thin
    e = PyDriverEngine(cfg, P)
    x1 = torch.randn(26, 512, 256, 1, device='cuda'); x128 = torch.randn(26, 512, 256, 128, device='cuda')
    print("thin", thin, "conv_in %.3f ms" % t(lambda: e.conv3(x1, "vae.encoder.conv_in")), "conv_out %.3f ms" % t(lambda: e.conv3(x128, "vae.decoder.conv_out")))
