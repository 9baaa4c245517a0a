"""Dev: Fat-Llama C3 stage with 1, 2, 4 channels (single vs overlapped pipelines)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
f = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True)
for C in (1, 2, 4):
    x = torch.from_numpy((0.3 * np.random.default_rng(0).standard_normal((C, 2880000))).astype(np.float32)).cuda()
    fe.enhance_device(x, 1, 800, 0.6, **f); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): fe.enhance_device(x, 1, 800, 0.6, **f)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 3 * 1e3
    print(f"C={C}: {ms:7.1f} ms  ({ms / C:.1f} ms per channel)")
