"""dev: one chirp-z Fat-Llama run (60 s + 2 samples, stereo, 100 iterations) for rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe, native
native.require_device()
x = (0.3 * torch.randn(2, 2880002, device="cuda")).clamp(-1, 1)
fl = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True)
for _ in range(2):
    fe.enhance_device(x, 1, 100, 0.6, **fl)
torch.cuda.synchronize()
