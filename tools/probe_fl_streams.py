"""Fat-Llama: two channel pipelines on two streams vs one pipeline (EGR_FL_STREAMS=1), bit for bit, at the C3 shape; run as two
processes because the setting is read at plan creation."""
import os, subprocess, sys, hashlib
code = r'''
import sys; sys.path.insert(0, '.')
import torch, hashlib, numpy as np
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
rng = np.random.Generator(np.random.PCG64(3))
x = torch.from_numpy((rng.standard_normal((2, 2880000)) * 3000).round().astype(np.float32)).cuda()
for rep in range(3):
    y = fe.enhance_device(x, 1, 300, 0.6, True, False, True, True)
    torch.cuda.synchronize()
    print("HASH", hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16])
'''
for env in ({}, {"EGR_FL_STREAMS": "1"}, {"EGR_FL_GRAPH": "0"}):
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True)
    print(env, [l.split()[1] for l in out.stdout.splitlines() if l.startswith("HASH")], out.stderr[-200:] if out.returncode else "")
