"""dev / test helper: full-size FlashSR forwards of TWO handles issued back to back on two verified-concurrent streams, compared bit
for bit with single-stream results.  With EGR_FSR_NO_STREAM_GUARD=1 the library does NOT chain the forwards, so their kernels really share the
GPU -- the configuration that returned wrong STFT bins before the packed-fp32 erratum was found.
Prints "bad rounds: N of R"; exit code 1 when N > 0."""
import os, sys
sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E, streams
from flashsr_pydriver import PyDriverEngine
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = A.FlashSRConfig()
P = A.init_params(cfg, 0)
e = PyDriverEngine(cfg, P)
e2 = PyDriverEngine(cfg, P)          # a second handle: its own scratch arena and workspaces (a handle serves ONE stream at a time)
x = 0.2 * torch.randn(18, cfg.chunk, generator=torch.Generator().manual_seed(9)).cuda()
nz = e.noise(18, None, 0)
ref = [e.c_forward(x[:9], nz[:9]).clone(), e.c_forward(x[9:], nz[9:]).clone()]
torch.cuda.synchronize()
side = streams.side_streams(1)
if not side:
    print("no concurrent stream available"); sys.exit(0)
cur = torch.cuda.current_stream()
bad = 0
import time
t0 = time.perf_counter()
for _ in range(rounds):
    ready = cur.record_event()
    side[0].wait_event(ready)
    with torch.cuda.stream(side[0]):
        a = e2.c_forward(x[:9], nz[:9])
    b = e.c_forward(x[9:], nz[9:])
    torch.cuda.synchronize()
    bad += int(not (torch.equal(a, ref[0]) and torch.equal(b, ref[1])))
dt = (time.perf_counter() - t0) / rounds
print(f"guard {'OFF' if os.environ.get('EGR_FSR_NO_STREAM_GUARD') == '1' else 'on'}: bad rounds: {bad} of {rounds}; {1e3 * dt:.1f} ms per round of 2 x 9 rows")
sys.exit(1 if bad else 0)
