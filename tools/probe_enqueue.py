"""Dev probe: host enqueue time vs GPU time of one FlashSR forward at small row counts (is configs[1] launch-bound?)."""
import sys, time; sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig(); e = PyDriverEngine(cfg, A.init_params(cfg, 0))
for R in (1, 2, 4, 8):
    x = 0.2 * torch.randn(R, cfg.chunk, device='cuda'); nz = e.noise(R, None, 0)
    e.forward_rows(x, nz); torch.cuda.synchronize()
    t0 = time.perf_counter(); y = e.forward_rows(x, nz); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rows {R}: enqueue {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms")
    if R <= 2:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            e.forward_rows(x, nz)
        torch.cuda.current_stream().wait_stream(s)
        try:
            with torch.cuda.graph(g):
                yg = e.forward_rows(x, nz)
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): g.replay()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            print(f"   graph replay {1e3 * (t1 - t0) / 5:.1f} ms  max|diff| {float((yg - y).abs().max()):.2e}")
        except Exception as ex:
            print("   graph capture failed:", repr(ex)[:300])
