"""dev: BASELINE configs[1] (one stereo 5.12 s chunk = 2 rows) -- wall time per forward and, under rocprofv3 --stats, where it goes."""
import sys, time
sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig()
e = PyDriverEngine(cfg, A.init_params(cfg, 0))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = 0.2 * torch.randn(rows, cfg.chunk, device="cuda")
for _ in range(3): e.c_infer(x, None, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): e.c_infer(x, None, 0)
torch.cuda.synchronize()
print(f"{rows} rows: {(time.perf_counter() - t0) * 100:.2f} ms per forward")
