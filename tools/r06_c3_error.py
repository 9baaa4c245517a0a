"""Round 6: the device loop's float32 round-off at BASELINE configs[2]'s own size (one channel, N = 2 880 000, 800 iterations, default
spec) against the float64 loop (torch.fft on the GPU, tests/test_gpu_fatllama.py::f64_loop_on_gpu), for the library named by
EGREGORA_AMD_LIB (precision-switch builds: tools/r06_build_precision_variants.sh).  The float32 ORACLE's own figures at this
size (profiles/r05/parity_report.txt:71: max 0.4338, rms 0.04904, plain LSD 8.12e-4 dB) are the yardstick; also times the stereo stage."""
import os, sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
from oracle import metrics as om
from test_gpu_fatllama import synth, f64_loop_on_gpu
ORACLE = (0.4338, 0.04904, 8.12e-4)
n = int(os.environ.get("PROBE_N", "2880000"))
x = synth(1, 2880002, seed=2880)[:, :n].copy()
exact = f64_loop_on_gpu(x, 800, 0.6)
got = fe.enhance_device(torch.from_numpy(x).cuda(), 1, 800, 0.6, False, False, False, False).cpu().numpy()
rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
seg = slice(0, 960000)
mg, rg = float(np.max(np.abs(got - exact))), rms(got - exact)
lsd = om.lsd_audio(exact[:, seg], got[:, seg])[0]
# stereo stage time, node flags
rng = np.random.Generator(np.random.PCG64(303))
xs = torch.from_numpy((0.25 * rng.standard_normal((2, n))).astype(np.float32)).cuda()
ts = []
for r in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    fe.enhance_device(xs, 1, 800, 0.6, True, False, True, True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
ts = sorted(ts[2:])
print(f"{os.path.basename(os.environ.get('EGREGORA_AMD_LIB', 'shipped')):28s} N={n}: max {mg:.4f} ({mg / ORACLE[0]:.2f}x oracle32)  rms {rg:.5f} ({rg / ORACLE[1]:.2f}x)  "
      f"plain LSD {lsd:.2e} dB ({lsd / ORACLE[2]:.2f}x)  stereo stage {ts[len(ts) // 2]:.2f} ms", flush=True)
