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
