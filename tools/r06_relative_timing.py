"""Round 6: the Fat-Llama stage at BASELINE configs[2] (60 s stereo 48 kHz, 800 iterations) under every threshold reading, carried
maximum vs a maximum pass per iteration (variant "recompute" = rounds 1-5), packed plan and the two chirp-z lengths.
Median of 5 after 2 warm-ups, inputs resident, node flags (PCM_16 hops, normalise on, autoscale off)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe

def run(n, variant, thr, iters=800, C=2):
    rng = np.random.Generator(np.random.PCG64(303))
    x = torch.from_numpy((0.25 * rng.standard_normal((C, n))).astype(np.float32)).cuda()
    f = dict(normalize=True, autoscale=False, pcm_in=True, node_post=True, variant=variant)
    ts = []
    for r in range(7):
        torch.cuda.synchronize(); t = time.perf_counter()
        y = fe.enhance_device(x, 1, iters, thr, **f)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    assert torch.isfinite(y).all()
    ts = sorted(ts[2:])
    i = fe.plan_info(n, 1)
    print(f"N={n:>8d} {variant or 'default':32s} thr {thr:5.2f}: median {ts[len(ts)//2]:8.2f} ms  min {ts[0]:8.2f}  plan {i['M1']}x{i['M2']}x{i['M3']} chirpz={i['chirpz_kind']}", flush=True)

for n in (2880000, 2880002, 2880001):
    run(n, "", 0.6)
    run(n, "relative", 0.6)
    run(n, "relative,recompute", 0.6)
    run(n, "relative,soft", 0.02)
    run(n, "relative,soft,recompute", 0.02)
    run(n, "soft", 50.0)
    fe.release_plans()
