"""dev: does it pay to run the two 13-row groups of a FlashSR pass HALF A FORWARD APART instead of in lockstep?  In lockstep both
groups are in the same phase (both in the MFMA-bound UNet / VAE, or both in the HBM-bound vocoder); staggered, one group's vocoder
shares the chip with the other's contractions.  Two handles, two verified-concurrent streams, K forwards each, issued from two host
threads (EGR_FSR_NO_STREAM_GUARD=1: the library must not chain them); prints ms per 26 rows for delays of 0 ... 1 forward."""
import os, sys, threading, time
os.environ.setdefault("EGR_FSR_NO_STREAM_GUARD", "1")
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E, streams
from flashsr_pydriver import PyDriverEngine
K = int(os.environ.get("PROBE_K", "8"))
ROWS = int(os.environ.get("PROBE_ROWS", "13"))
cfg = A.FlashSRConfig()
P = A.init_params(cfg, 0)
e = [PyDriverEngine(cfg, P), PyDriverEngine(cfg, P)]
x = 0.2 * torch.randn(2 * ROWS, cfg.chunk, generator=torch.Generator().manual_seed(9)).cuda()
nz = e[0].noise(2 * ROWS, None, 0)
side = streams.side_streams(2)
assert len(side) >= 2, "needs two concurrent side streams"
xs = [x[:ROWS], x[ROWS:]]; ns = [nz[:ROWS], nz[ROWS:]]
for i in range(2):
    with torch.cuda.stream(side[i]):
        e[i].c_forward(xs[i], ns[i])
torch.cuda.synchronize()
# one forward alone
t0 = time.perf_counter()
with torch.cuda.stream(side[0]):
    for _ in range(4):
        e[0].c_forward(xs[0], ns[0])
torch.cuda.synchronize()
alone = (time.perf_counter() - t0) / 4
print(f"one {ROWS}-row forward alone: {1e3 * alone:.1f} ms", flush=True)

def worker(i, delay):
    torch.cuda.set_device(0)
    if delay > 0:
        time.sleep(delay)
    with torch.cuda.stream(side[i]):
        for _ in range(K):
            e[i].c_forward(xs[i], ns[i])

for frac in (0.0, 0.25, 0.5, 0.75, 0.0, 0.5):
    delay = frac * alone * 1.6          # (two concurrent forwards take ~1.6x one alone)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(0, 0.0)), threading.Thread(target=worker, args=(1, delay))]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"delay {1e3 * delay:6.1f} ms: {K} x 2 forwards in {1e3 * wall:7.1f} ms -> {1e3 * wall / K:6.1f} ms per {2 * ROWS} rows "
          f"({1e3 * (wall - delay) / K:6.1f} without the ramp)", flush=True)
