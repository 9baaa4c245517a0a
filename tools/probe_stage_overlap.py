"""dev: does the Fat-Llama stage of file i overlap with the FlashSR stage of file i + 1 when it runs on a HIGH-PRIORITY stream?
(On equal-priority streams it does not: the contraction kernels fill every CU's LDS and the loop kernels queue behind them.)"""
import sys, time
sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe, flashsr_arch as A, flashsr_engine as E, native, streams
from egregora_amd.egregora_audio_super_resolution import upscale_48k
import bench
native.require_device()
cfg = A.FlashSRConfig(); eng = E.FlashSREngine(cfg, A.init_params(cfg, seed=0)); E.set_engine(eng)
x = torch.from_numpy(bench.synth(404, bench.SEG)).cuda()
fl = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True)
prio = int(sys.argv[1]) if len(sys.argv) > 1 else -1
sA = torch.cuda.Stream(priority=0)
sB = torch.cuda.Stream(priority=prio)
if len(sys.argv) > 2:      # the loop's second pipeline on a high-priority stream as well
    _orig = torch.cuda.Stream
    streams.torch.cuda.Stream = lambda *a, **k: _orig(priority=prio)
def fs():
    with torch.cuda.stream(sA):
        return upscale_48k(x, False)
def fat(y):
    with torch.cuda.stream(sB):
        return fe.enhance_device(y, 1, 800, 0.6, **fl)
y = fs(); torch.cuda.synchronize(); fat(y); torch.cuda.synchronize(); fat(y); torch.cuda.synchronize()
n = 4
t0 = time.perf_counter()
for _ in range(n):
    y = fs(); torch.cuda.synchronize(); z = fat(y); torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
yp = y
for _ in range(n):
    z = fat(yp); y2 = fs(); torch.cuda.synchronize(); yp = y2
pip = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    y = fs()
torch.cuda.synchronize(); f_only = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    z = fat(y)
torch.cuda.synchronize(); l_only = (time.perf_counter() - t0) / n
print(f"priority {prio} side_hi={len(sys.argv) > 2}: sequential {seq*1e3:.1f} ms/file, pipelined {pip*1e3:.1f} ms/file, flashsr alone {f_only*1e3:.1f}, fat-llama alone {l_only*1e3:.1f}", flush=True)
