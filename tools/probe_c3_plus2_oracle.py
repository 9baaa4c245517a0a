"""Dev / provenance: the float32 oracle (pocketfft, Bluestein for N = 2 x a prime) at N = 2 880 002, 800 iterations, against the float64
loop (torch.fft on the GPU) -- the reference errors tests/test_gpu_fatllama.py::test_c3_full_length_800_iterations_against_the_oracle
holds the device's chirp-z path to at that length (about ten minutes of host time, hence not part of the suite)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import fatllama as ofl, metrics as om
import test_gpu_fatllama as T
n = 2880002
x = T.synth(1, 2880002, seed=2880)[:, :n].copy()
t = time.time(); exact = T.f64_loop_on_gpu(x, 800, 0.6); print("f64 on GPU %.1f s" % (time.time() - t), flush=True)
t = time.time(); want = ofl.enhance_channels(x, 1, 800, 0.6, normalize=False, autoscale=False); print("oracle32 %.1f s" % (time.time() - t), flush=True)
scale = float(np.max(np.abs(exact)))
rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
seg = slice(0, 960000)
print("N = %d: oracle32 max err %.4e (%.3e of the peak %.1f), rms %.4e, LSD plain %.3e dB" % (
    n, float(np.max(np.abs(want - exact))), float(np.max(np.abs(want - exact))) / scale, scale, rms(want - exact), om.lsd_audio(exact[:, seg], want[:, seg])[0]))

# provenance record read by tests/test_gpu_fatllama.py (tests/golden/oracle32_c3_plus2.json): copy the file written here into tests/golden/
import datetime, json, os, platform
import torch
rec = {"what": "float32 oracle (oracle/fatllama.py on scipy pocketfft, Bluestein at this length) vs the float64 loop, N = 2 880 002, 800 iterations, default spec, thr 0.6",
       "n": n, "iterations": 800, "input": "tests/test_gpu_fatllama.py synth(1, 2880002, seed=2880)", "seed": 2880,
       "max_err": float(np.max(np.abs(want - exact))), "rms_err": rms(want - exact), "lsd_plain_db": float(om.lsd_audio(exact[:, seg], want[:, seg])[0]),
       "peak": scale, "date": datetime.datetime.utcnow().isoformat() + "Z", "host_cpu": platform.processor() or platform.machine(),
       "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
       "numpy": np.__version__, "scipy": __import__("scipy").__version__, "torch": torch.__version__,
       "gpu_for_the_float64_loop": torch.cuda.get_device_name(0) if torch.cuda.is_available() else None,
       "generator": "tools/probe_c3_plus2_oracle.py"}
out = os.environ.get("PROBE_OUT", "gpurun_out/r05/oracle32_c3_plus2.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rec, open(out, "w"), indent=1)
print("wrote", out, rec)
