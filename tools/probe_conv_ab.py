"""Dev probe: per-shape MFMA-kernel time of one FlashSR forward under two environments (A: as built, B: argv env assignments),
printing the shapes whose time differs."""
import os, subprocess, sys, json
def run(env):
    code = r'''
import sys; sys.path.insert(0, '.')
import torch, json
from collections import defaultdict
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig(); e = PyDriverEngine(cfg, A.init_params(cfg, 0))
x = 0.2 * torch.randn(26, cfg.chunk, device='cuda'); nz = e.noise(26, None, 0)
for _ in range(2): e.forward_rows(x, nz)
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for rep in range(3):
    e.prof = []; e.forward_rows(x, nz); torch.cuda.synchronize()
    for kind, fl, a, b, shape in e.prof:
        g = agg[str(shape)]; g[0] += 1; g[1] += a.elapsed_time(b) / 3
    e.prof = None
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(3): e.forward_rows(x, nz)
t1.record(); torch.cuda.synchronize()
print("JSON" + json.dumps({"shapes": agg, "forward_ms": t0.elapsed_time(t1) / 3}))
'''
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("JSON")][0][4:])
envb = dict(a.split("=", 1) for a in sys.argv[1:])
A_, B_ = run({}), run(envb)
print("forward ms: A %.2f  B(%s) %.2f" % (A_["forward_ms"], envb, B_["forward_ms"]))
keys = set(A_["shapes"]) | set(B_["shapes"])
rows = sorted(((A_["shapes"].get(k, [0, 0])[1] - B_["shapes"].get(k, [0, 0])[1], k) for k in keys), key=lambda r: -abs(r[0]))
for d, k in rows[:25]:
    print("%-62s A %7.3f ms  B %7.3f ms  (A-B %+.3f)" % (k, A_["shapes"].get(k, [0, 0])[1], B_["shapes"].get(k, [0, 0])[1], d))
print("sum A %.2f  sum B %.2f" % (sum(v[1] for v in A_["shapes"].values()), sum(v[1] for v in B_["shapes"].values())))
