"""ORACLE (test infrastructure, not product): CPU restatement of the Fat-Llama path.

PARITY UNPINNED for the inner arithmetic.  The algorithm lives in PyPI `fat-llama>=1.1.0` (CuPy) /
`fat-llama-fftw>=1.0.4.4` (pyFFTW) -- requirements.txt:10-11, lower bounds only, no lock file -- and
neither package (nor cupy/pyfftw/pydub/soundfile) exists in the build image or on the GPU box.
`feed.upscale` below is restated from the package's published algorithm as recalled (SURVEY.md
section 8 row a13, tag UPSTREAM-RECALL) and anchored on the reference's own call sites:
  egregora_fat_llama_gpu.py:213-224 (10 kwargs), egregora_fat_llama_cpu.py:126-134 (7 kwargs),
  the read/write patches at egregora_fat_llama_gpu.py:174-208, the temp-WAV hand-over :34-37,52-53 and
  the read-back :291-294.
What IS pinned (fixture G9, tests/golden/g9_fatllama_adapter.json): the kwargs, the write-patch
scaling rule and the dict/tuple hand-over.  Every uncertain upstream constant is a named field of
`FatLlamaSpec` so it can be corrected without touching the kernels.

The FFT here is scipy.fft on complex64 (pocketfft, float32 arithmetic), matching upstream's
complex64 spectrum of a float32 signal.
"""
from dataclasses import dataclass

import numpy as np

try:  # scipy keeps complex64; numpy>=2 does as well, numpy<2 upcasts
    import scipy.fft as _fft
except Exception:  # pragma: no cover
    _fft = np.fft


@dataclass
class FatLlamaSpec:
    # ---- disk hand-over (libsndfile default subtype for float data to WAV/FLAC is PCM_16) ----
    pcm_write_scale: float = 32767.0     # libsndfile f2s: lrintf(x * 0x7FFF), clipping off => wrap
    pcm_read_scale: float = 32768.0      # libsndfile s2f: x / 0x8000 (sf.read dtype float32)
    sample_width: int = 2                # pydub AudioSegment.sample_width for PCM_16
    # ---- upstream feed.upscale ----
    factor_rounding: str = "round"       # upscale_factor = round(target_bps / source_bps), min 1
    interp: str = "linear"               # y[i*f+j] = (1-j/f) x[i] + (j/f) x[i+1], i < n-1; tail zero
                                         # "zero_stuff": y[i*f] = x[i], zeros between (spectral-sparsity interpolation)
                                         # "linspace": np.interp(np.linspace(0, n-1, n_out), np.arange(n), x) -- endpoint-inclusive
                                         #   grid, no zero tail (the round-2 judge's recollection; as unverified as the survey's)
    factor_mode: str = "integer"         # "integer": n_out = n * upscale_factor (rounded per factor_rounding);
                                         # "ratio_then_int": n_out = int(n * ratio), ratio = target_bps / source_bps >= 1 applied
                                         #   BEFORE int() (needs interp = "linspace": the only up-rating defined for a ratio)
    normalize_scope: str = "joint"       # out / max|out| over all channels
    autoscale: str = "match_peak"        # per channel: out *= max|in| / max|out|
    # ---- threshold semantics of the IST loop (SPEC.md section 3; every combination runs on the device) ----
    threshold_ref: str = "absolute"      # "absolute": t = threshold_value; "relative_to_max": t = threshold_value * max|.| of
                                         #   the array being thresholded (per channel, recomputed every iteration)
    threshold_kind: str = "hard"         # "hard": X [|X| > t]; "soft": X max(0, 1 - t/|X|) (complex soft shrink)
    init_threshold: str = "same"         # "same": d0 = hard-threshold of y with the same reference rule; "none": d0 = y
    use_rfft: bool = False               # oracle default: full complex FFT like upstream


DEFAULT_SPEC = FatLlamaSpec()


def pcm16_write(x, spec=DEFAULT_SPEC):
    """float -> int16 as sf.write(path, x) does for WAV/FLAC (egregora_fat_llama_gpu.py:36): round to
    nearest even of x*32767, then 16-bit two's-complement wrap (no clipping)."""
    q = np.rint(np.asarray(x, np.float32) * np.float32(spec.pcm_write_scale)).astype(np.int64)
    return ((q + 32768) % 65536 - 32768).astype(np.int16)


def pcm16_read(q, spec=DEFAULT_SPEC):
    """int16 -> float32 as sf.read(dtype='float32') does (egregora_fat_llama_gpu.py:291)."""
    return (np.asarray(q, np.int16).astype(np.float32) / np.float32(spec.pcm_read_scale)).astype(np.float32)


def upscale_factor(sr, channels, target_bitrate_kbps, spec=DEFAULT_SPEC):
    src_bps = sr * channels * 8 * spec.sample_width
    r = (target_bitrate_kbps * 1000.0) / src_bps
    f = int(round(r)) if spec.factor_rounding == "round" else int(r)
    return max(1, f)


def upscale_ratio(sr, channels, target_bitrate_kbps, spec=DEFAULT_SPEC):
    """target bits/s over source bits/s, at least 1 (factor_mode "ratio_then_int")."""
    return max(1.0, (target_bitrate_kbps * 1000.0) / (sr * channels * 8 * spec.sample_width))


def interpolate(x, f, spec=DEFAULT_SPEC, n_out=None):
    """Up-rate by integer factor f (or to n_out samples, linspace only).  linear: the last input sample's f slots stay zero;
    zero_stuff: y[::f] = x; linspace: numpy.interp on the endpoint-inclusive grid (float64 arithmetic, rounded to float32)."""
    x = np.asarray(x, np.float32)
    n = x.shape[0]
    if spec.interp == "linspace":
        m = n * f if n_out is None else int(n_out)
        return np.interp(np.linspace(0, n - 1, m), np.arange(n), x.astype(np.float64)).astype(np.float32)
    if n_out is not None and n_out != n * f:
        raise ValueError("an output length that is not n * f needs interp = 'linspace'")
    y = np.zeros(n * f, np.float32)
    if spec.interp == "zero_stuff":
        y[::f] = x
        return y
    if n > 1:
        t = (np.arange(f, dtype=np.float32) / np.float32(f))[None, :]
        a = x[:-1, None]
        b = x[1:, None]
        y[: (n - 1) * f] = ((np.float32(1.0) - t) * a + t * b).reshape(-1)
    return y


def _level(mag, thr, spec):
    """The threshold level for an array of magnitudes under spec.threshold_ref (same dtype as `mag`)."""
    t = mag.dtype.type(thr)
    if spec.threshold_ref == "relative_to_max":
        t = t * (np.max(mag) if mag.size else mag.dtype.type(0))
    return t


def _shrink(X, thr, spec):
    """Threshold a (complex or real) array under spec.threshold_ref / spec.threshold_kind; dtype preserved."""
    mag = np.abs(X)
    t = _level(mag, thr, spec)
    if spec.threshold_kind == "soft":
        one = mag.dtype.type(1)
        g = np.where(mag > t, one - t / np.where(mag > t, mag, one), mag.dtype.type(0))
        return (X * g).astype(X.dtype)
    return np.where(mag > t, X, 0).astype(X.dtype)


def ist_loop(y, max_iter, thr, spec=DEFAULT_SPEC, trace=None, exact=False):
    """d0 = where(|y|>t0, y, 0); repeat: X=fft(d); X=shrink(X, t); d=ifft(X).real, t per SPEC.md section 3.
    exact=True runs the same loop in float64/complex128 (a yardstick for float32 round-off, not upstream)."""
    rt = np.float64 if exact else np.float32
    ct = np.complex128 if exact else np.complex64
    y = np.asarray(y, rt)
    if spec.init_threshold == "none":
        d = y.copy()
    else:
        mag = np.abs(y)
        d = np.where(mag > _level(mag, thr, spec), y, rt(0)).astype(rt)
    for it in range(int(max_iter)):
        if exact:
            d = np.fft.ifft(_shrink(np.fft.fft(d), thr, spec)).real
        elif spec.use_rfft:
            d = _fft.irfft(_shrink(_fft.rfft(d).astype(ct), thr, spec), n=d.shape[0]).astype(rt)
        else:
            d = _fft.ifft(_shrink(_fft.fft(d.astype(ct)).astype(ct), thr, spec)).real.astype(rt)
        if trace is not None:
            trace.append(d.copy())
    return d


def enhance_channels(x_ci, factor, max_iter, thr, normalize=True, autoscale=True, spec=DEFAULT_SPEC,
                     exact=False, n_out=None):
    """x_ci: [C,N] float32 on the *integer* PCM scale (what pydub hands upstream).
    Returns [C, N*factor] float32 (before the write patch); float64 when exact=True (loop only)."""
    x_ci = np.asarray(x_ci, np.float32)
    outs = []
    for c in range(x_ci.shape[0]):
        y = interpolate(x_ci[c], factor, spec, n_out)
        d = ist_loop(y, max_iter, thr, spec, exact=exact)
        outs.append(y.astype(np.float64) + d if exact else (y + d).astype(np.float32))
    out = np.stack(outs, 0)
    if exact:
        return out
    if autoscale and spec.autoscale == "match_peak":
        for c in range(out.shape[0]):
            po = float(np.max(np.abs(out[c]))) if out.shape[1] else 0.0
            pi = float(np.max(np.abs(x_ci[c]))) if x_ci.shape[1] else 0.0
            if po > 0.0:
                out[c] = out[c] * np.float32(pi / po)
    if normalize:
        if spec.normalize_scope == "joint":
            m = float(np.max(np.abs(out))) if out.size else 0.0
            if m > 0.0:
                out = out / np.float32(m)
        else:
            for c in range(out.shape[0]):
                m = float(np.max(np.abs(out[c]))) if out.shape[1] else 0.0
                if m > 0.0:
                    out[c] = out[c] / np.float32(m)
    return out.astype(np.float32)


def write_patch_scale(out, spec=DEFAULT_SPEC):
    """The reference's write_audio patch (egregora_fat_llama_gpu.py:191-205): iff max|out|>1 divide by
    2**(8*sw-1) when the sample width is known, else by max."""
    m = float(np.max(np.abs(out))) if out.size else 0.0
    if m > 1.0:
        if spec.sample_width:
            return (out / np.float32(2 ** (8 * spec.sample_width - 1))).astype(np.float32)
        return (out / np.float32(m)).astype(np.float32)
    return out


def node_run(cs, sr, max_iterations, threshold_value, target_bitrate_kbps,
             toggle_normalize=True, toggle_autoscale=True, spec=DEFAULT_SPEC):
    """Whole node path for an AUDIO-dict input: [C,T] float32 in, ([C,T*f] float32, sr*f) out.
    Follows EgregoraFatLlamaGPU.run (egregora_fat_llama_gpu.py:257-294) with the disk hops replaced
    by their arithmetic: PCM16 write -> upstream on the integer scale -> write patch -> PCM16 write
    -> float read-back."""
    cs = np.asarray(cs, np.float32)
    C = cs.shape[0]
    xi = pcm16_write(cs, spec).astype(np.float32)            # temp WAV in, pydub ints out
    f = upscale_factor(sr, C, target_bitrate_kbps, spec)
    n_out, sr_out = None, sr * f
    if spec.factor_mode == "ratio_then_int":
        r = upscale_ratio(sr, C, target_bitrate_kbps, spec)
        n_out, sr_out, f = int(cs.shape[1] * r), int(sr * r), 1
    out = enhance_channels(xi, f, max_iterations, threshold_value, toggle_normalize, toggle_autoscale, spec, n_out=n_out)
    out = write_patch_scale(out, spec)
    y = pcm16_read(pcm16_write(out, spec), spec)             # upstream sf.write + node sf.read
    # _to_cs on read-back (egregora_fat_llama_gpu.py:292): peak > 1 cannot occur after PCM16
    return y, sr_out
