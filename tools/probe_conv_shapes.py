"""Dev probe: per-(kernel, shape) efficiency of the MFMA launches inside one FlashSR forward (rows = argv[1], default 26)."""
import sys; sys.path.insert(0, '.')
import torch
from collections import defaultdict
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig(); e = PyDriverEngine(cfg, A.init_params(cfg, 0))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 26
x = 0.2 * torch.randn(R, cfg.chunk, device='cuda'); nz = e.noise(R, None, 0)
e.forward_rows(x, nz); torch.cuda.synchronize()
e.prof = []; e.forward_rows(x, nz); torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0, 0.0])
for kind, fl, a, b, shape in e.prof:
    t = a.elapsed_time(b); g = agg[(kind, shape)]; g[0] += 1; g[1] += fl; g[2] += t
tot = sum(v[2] for v in agg.values())
print("total MFMA-kernel ms %.1f  flops %.3e  TF/s(fp32-eq) %.1f" % (tot, sum(v[1] for v in agg.values()), sum(v[1] for v in agg.values()) / tot / 1e9))
for (kind, shape), (n, fl, t) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
    print("%-32s %-58s n=%3d ms=%6.2f (%4.1f%%) avg_us=%7.1f TF/s=%6.1f" % (kind, str(shape), n, t, 100 * t / tot, 1e3 * t / n, fl / t / 1e9))
