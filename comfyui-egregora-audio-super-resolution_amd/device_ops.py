"""Thin torch-tensor wrappers over the glue entry points of the C ABI (csrc/egr_glue.hip)."""
import math

import numpy as np
import torch

from . import native


def _chk(t, what):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{what}: want a contiguous float32 CUDA tensor")


_WINDOWS = {}


def hann_window(n: int, device) -> torch.Tensor:
    """np.hanning(n) as float32 on `device` (bit-identical to the reference's _hann, :210-211)."""
    key = (n, str(device))
    w = _WINDOWS.get(key)
    if w is None:
        w = torch.from_numpy(np.hanning(n).astype(np.float32)).to(device)
        _WINDOWS[key] = w
    return w


def pcm16_roundtrip(x: torch.Tensor, write_scale: float = 32767.0, read_div: float = 32768.0) -> torch.Tensor:
    _chk(x, "pcm16_roundtrip")
    y = torch.empty_like(x)
    native.check(native.lib().egr_pcm16_roundtrip(native.ptr(x), native.ptr(y), x.numel(), write_scale, read_div,
                                                   native.stream_ptr()), "egr_pcm16_roundtrip")
    return y


def chunk_gather(x_ct: torch.Tensor, win: int, hop: int, chunk_begin: int, n_chunks: int) -> torch.Tensor:
    """[C,T] -> [n_chunks, C, win] zero-padded windows starting at (chunk_begin+k)*hop."""
    _chk(x_ct, "chunk_gather")
    C, T = x_ct.shape
    out = torch.empty((n_chunks, C, win), dtype=torch.float32, device=x_ct.device)
    native.check(native.lib().egr_chunk_gather(native.ptr(x_ct), C, T, win, hop, chunk_begin, n_chunks,
                                                native.ptr(out), native.stream_ptr()), "egr_chunk_gather")
    return out


def wola_stitch(preds: torch.Tensor, total: int, win: int, hop: int) -> torch.Tensor:
    """preds [n_chunks, C, Lp] -> [C,total]: Hann WOLA with division by the weight sum."""
    _chk(preds, "wola_stitch")
    n, C, lp = preds.shape
    out = torch.empty((C, total), dtype=torch.float32, device=preds.device)
    w = hann_window(win, preds.device)
    native.check(native.lib().egr_wola_stitch(native.ptr(preds), n, C, lp, total, win, hop, native.ptr(w),
                                               native.ptr(out), native.stream_ptr()), "egr_wola_stitch")
    return out


def stft_mag(x: torch.Tensor, n_fft: int = 2048, hop: int = 512) -> torch.Tensor:
    """[T] or [C,T] -> [n_fft/2+1, frames] (a transposed view of the frame-major device buffer), matching
    the reference's _stft_mag layout (egregora_audio_eval_pack.py:389-402)."""
    if x.dim() == 1:
        x = x[None, :]
    x = x.contiguous()
    _chk(x, "stft_mag")
    C, T = x.shape
    frames = 1 + max(0, (T - n_fft) // hop)
    out = torch.empty((frames, n_fft // 2 + 1), dtype=torch.float32, device=x.device)
    w = hann_window(n_fft, x.device)
    native.check(native.lib().egr_stft_mag(native.ptr(x), C, T, n_fft, hop, native.ptr(w), native.ptr(out),
                                            native.stream_ptr()), "egr_stft_mag")
    return out.t()


def _stft_frames(x: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """[C,T] -> frame-major [frames, n_fft/2+1] magnitudes of the mono downmix."""
    return stft_mag(x, n_fft, hop).t()


def lsd(a: torch.Tensor, b: torch.Tensor, n_fft: int = 2048, hop: int = 512):
    """(mean, p95) log-spectral distance in dB between two [C,T] / [T] CUDA signals of equal length, as the reference's
    _stft_mag + _lsd (egregora_audio_eval_pack.py:389-411): everything per sample / per bin runs on the device; the host
    only divides the frame sum and interpolates numpy's linear percentile between two order statistics."""
    import ctypes as C
    L = native.lib()
    SA, SB = _stft_frames(a, n_fft, hop), _stft_frames(b, n_fft, hop)
    frames, nb = SA.shape
    per = torch.empty((frames,), dtype=torch.float32, device=SA.device)
    native.check(L.egr_lsd_frames(native.ptr(SA), native.ptr(SB), frames, nb, native.ptr(per), native.stream_ptr()),
                 "egr_lsd_frames")
    tot = torch.empty((1,), dtype=torch.float64, device=SA.device)
    native.check(L.egr_sum_f64(native.ptr(per), frames, native.ptr(tot), native.stream_ptr()), "egr_sum_f64")
    pos = 0.95 * (frames - 1)                       # numpy percentile, method="linear"
    lo = int(np.floor(pos))
    hi = min(lo + 1, frames - 1)
    o2 = torch.empty((2,), dtype=torch.float32, device=SA.device)
    native.check(L.egr_order_stats2(native.ptr(per), frames, lo, hi, native.ptr(o2), native.stream_ptr()), "egr_order_stats2")
    v_lo, v_hi = (float(v) for v in o2.cpu())
    t = pos - lo
    p95 = v_hi - (v_hi - v_lo) * (1.0 - t) if t >= 0.5 else v_lo + (v_hi - v_lo) * t
    return float(np.float32(float(tot.cpu()) / frames)), float(np.float32(p95))


def si_sdr(s: torch.Tensor, s_hat: torch.Tensor) -> float:
    """Scale-invariant SDR in dB of the mono downmixes (reference _si_sdr, egregora_audio_eval_pack.py:414-429); the four
    sums are formed in double on the device."""
    if s.dim() == 1:
        s = s[None, :]
    if s_hat.dim() == 1:
        s_hat = s_hat[None, :]
    s, s_hat = s.contiguous(), s_hat.contiguous()
    _chk(s, "si_sdr"), _chk(s_hat, "si_sdr")
    n = min(s.shape[1], s_hat.shape[1])
    out = torch.empty((4,), dtype=torch.float64, device=s.device)
    native.check(native.lib().egr_si_sdr_terms(native.ptr(s), s.shape[0], s.shape[1], native.ptr(s_hat), s_hat.shape[0],
                                                s_hat.shape[1], n, native.ptr(out), native.stream_ptr()), "egr_si_sdr_terms")
    _, _, tt, ee = (float(v) for v in out.cpu())
    return float(10.0 * np.log10((tt + 1e-20) / (ee + 1e-20)))


def resample_linear(x_ct: torch.Tensor, n_out: int) -> torch.Tensor:
    x = x_ct.contiguous()
    _chk(x, "resample_linear")
    y = torch.empty((x.shape[0], int(n_out)), dtype=torch.float32, device=x.device)
    native.check(native.lib().egr_resample_linear(native.ptr(x), x.shape[0], x.shape[1], native.ptr(y), int(n_out),
                                                   native.stream_ptr()), "egr_resample_linear")
    return y


def xcorr_delay(a: torch.Tensor, b: torch.Tensor, sr: int, max_shift_smp: int) -> float:
    """Delay of b against a in samples by GCC-PHAT with a parabolic refinement, as the reference's _xcorr_delay
    (egregora_null_test_suite.py:213-237) -- including its convention: the centred correlation puts lag 0 one index left of
    the centre, so the value returned is the true lag minus one (reference quirk Q8, reproduced because callers compensate
    with this very number).  a, b: 1-D float32 CUDA tensors; the whole-signal transforms run on the Fat-Llama passes."""
    import ctypes as C
    from . import fatllama_engine as fe
    a, b = a.contiguous(), b.contiguous()
    _chk(a, "xcorr_delay"), _chk(b, "xcorr_delay")
    n = 1
    while n < a.numel() + b.numel():
        n <<= 1
    plan = fe._plan(2 * n, 1, 1, a.device.index or 0)
    work = torch.empty(4 * n, dtype=torch.float32, device=a.device)
    out4 = torch.empty(4, dtype=torch.float32, device=a.device)
    native.check(native.lib().egr_gcc_phat(C.c_void_p(plan), native.ptr(a), a.numel(), native.ptr(b), b.numel(),
                                           int(max_shift_smp), native.ptr(work), native.ptr(out4), native.stream_ptr()),
                 "egr_gcc_phat")
    o = out4.cpu()
    rel = int(o[:1].view(torch.int32)[0])
    y0, y1, y2 = (float(v) for v in o[1:])
    centre = n // 2
    idx = centre + rel
    frac = 0.0
    if 1 <= idx < n - 1:
        denom = 2 * (y0 - 2 * y1 + y2)
        frac = 0.0 if abs(denom) < 1e-12 else (y0 - y2) / denom
    return float(rel + frac)


# ------------------------------------------------------------------ null-test suite levels and sums (egr_glue.hip, egr_fatllama.hip)
def _rows(x: torch.Tensor, what: str) -> torch.Tensor:
    if x.dim() == 1:
        x = x[None, :]
    x = x.contiguous()
    _chk(x, what)
    return x


def _f64(n: int, device) -> torch.Tensor:
    return torch.empty((n,), dtype=torch.float64, device=device)


def mono_mean(x_ct: torch.Tensor, n: int = None) -> torch.Tensor:
    """float32 mean over channels of the first n samples (numpy .mean(axis=0) order)."""
    x = _rows(x_ct, "mono_mean")
    n = x.shape[1] if n is None else int(n)
    y = torch.empty((n,), dtype=torch.float32, device=x.device)
    native.check(native.lib().egr_mono_mean(native.ptr(x), x.shape[0], x.shape[1], n, native.ptr(y), native.stream_ptr()), "egr_mono_mean")
    return y


def block_mean_squares(x_ct: torch.Tensor, block: int, hop: int) -> np.ndarray:
    """float64 mean squares of the mono downmix over frames = 1 + max(0, (N - block) // hop) blocks (host array)."""
    x = _rows(x_ct, "block_mean_squares")
    frames = 1 + max(0, (x.shape[1] - block) // hop)
    out = _f64(frames, x.device)
    native.check(native.lib().egr_frame_meansq(native.ptr(x), x.shape[0], x.shape[1], int(block), int(hop), frames, native.ptr(out),
                                               native.stream_ptr()), "egr_frame_meansq")
    return out.cpu().numpy()


def rms_db(x_ct: torch.Tensor) -> float:
    """10 log10(mean(mono^2) + 1e-20): the reference's _rms_db on samples.mean(axis=0) (egregora_null_test_suite.py:119-122)."""
    n = x_ct.shape[-1]
    return 10.0 * math.log10(float(block_mean_squares(x_ct, n, n)[0]) + 1e-20)


def k_weight(x_ct: torch.Tensor, sr: int) -> torch.Tensor:
    x = _rows(x_ct, "k_weight")
    k = math.exp(-2 * math.pi * (60.0 / (sr * 0.5)))
    y = torch.empty_like(x)
    native.check(native.lib().egr_kweight(native.ptr(x), x.shape[0], x.shape[1], float(np.float32(1 - k)), float(np.float32(k)),
                                          native.ptr(y), native.stream_ptr()), "egr_kweight")
    return y


def integrated_lufs(x_ct: torch.Tensor, sr: int) -> float:
    """The suite's gated loudness (integrated_lufs, egregora_null_test_suite.py:143-165): K-weighting and the 400 ms block
    energies on the device, the gate over the few block values on the host in the reference's own expressions."""
    ms = block_mean_squares(k_weight(x_ct, sr), max(1, int(round(0.400 * sr))), max(1, int(round(0.100 * sr)))) + 1e-20
    ungated = -0.691 + 10.0 * np.log10(np.mean(ms))
    keep = (-0.691 + 10.0 * np.log10(ms)) >= ungated - 10.0
    if np.any(keep):
        ms = ms[keep]
    return float(-0.691 + 10.0 * np.log10(np.mean(ms)))


def pair_stats(a_ct: torch.Tensor, b_ct: torch.Tensor, n: int, k: float = None):
    """(sum a, sum b, sum ab, sum aa, sum bb) in double over the first n samples of the mono downmixes; b scaled by float32(k) first."""
    a, b = _rows(a_ct, "pair_stats"), _rows(b_ct, "pair_stats")
    out = _f64(5, a.device)
    native.check(native.lib().egr_pair_stats(native.ptr(a), a.shape[0], a.shape[1], native.ptr(b), b.shape[0], b.shape[1], int(n),
                                             float(np.float32(k if k is not None else 1.0)), int(k is not None), native.ptr(out),
                                             native.stream_ptr()), "egr_pair_stats")
    return tuple(float(v) for v in out.cpu())


def null_mix(a_ct: torch.Tensor, b_ct: torch.Tensor, n: int, k: float = None, invert_b: bool = True):
    """null = a +- float32(b * k) on [C, n]; -> (null, sum of squares of its mono downmix, number of |null| > 1)."""
    a, b = _rows(a_ct, "null_mix"), _rows(b_ct, "null_mix")
    if a.shape[0] != b.shape[0]:
        raise ValueError(f"operands could not be broadcast together with shapes {tuple(a[:, :n].shape)} {tuple(b[:, :n].shape)}")
    nul = torch.empty((a.shape[0], int(n)), dtype=torch.float32, device=a.device)
    out = _f64(2, a.device)
    native.check(native.lib().egr_null_mix(native.ptr(a), a.shape[1], native.ptr(b), b.shape[1], a.shape[0], int(n),
                                           float(np.float32(k if k is not None else 1.0)), int(k is not None), int(bool(invert_b)),
                                           native.ptr(nul), native.ptr(out), native.stream_ptr()), "egr_null_mix")
    s, overs = (float(v) for v in out.cpu())
    return nul, s, int(overs)


def scale(x: torch.Tensor, gain: float) -> torch.Tensor:
    x = x.contiguous()
    _chk(x, "scale")
    y = torch.empty_like(x)
    native.check(native.lib().egr_eltwise(native.ptr(x), None, native.ptr(y), x.numel(), 3, float(np.float32(gain)), 0.0,
                                          native.stream_ptr()), "egr_eltwise")
    return y


def band_energy_hi_db(x_ct: torch.Tensor, sr: int, lo_hz: float) -> float:
    """10 log10(E_hi / E_all + 1e-20) with the one-sided energies of the length-N rfft of the mono downmix, bins f >= lo_hz on top
    (reference _band_energy_hi_db, egregora_null_test_suite.py:192-199).  The transform runs at the signal's own length on a
    Fat-Llama plan (packed-real, or chirp-z for lengths without one): egr_band_filter leaves the high band in the time domain and
    Parseval turns time-domain sums into the two energies."""
    import ctypes as C
    from . import fatllama_engine as fe
    mono = mono_mean(x_ct)
    n = mono.numel()
    step = 1.0 / (n * (1.0 / sr))                      # np.fft.rfftfreq: k * (1 / (n d))
    lo = max(0, int(math.ceil(lo_hz / step)))
    while lo > 0 and (lo - 1) * step >= lo_hz:
        lo -= 1
    while lo * step < lo_hz:
        lo += 1
    if lo > n // 2:
        return 10.0 * math.log10(1e-20)
    plan = fe._plan(n, 1, 1, mono.device.index or 0)
    hi = torch.empty_like(mono)
    native.check(native.lib().egr_band_filter(C.c_void_p(plan), native.ptr(mono), lo, native.ptr(hi), native.stream_ptr()),
                 "egr_band_filter")
    out = _f64(6, mono.device)
    native.check(native.lib().egr_band_sums(native.ptr(mono), native.ptr(hi), n, native.ptr(out), native.stream_ptr()), "egr_band_sums")
    xx, x0, xn, yy, y0, yn = (float(v) for v in out.cpu())
    even = 1.0 if n % 2 == 0 else 0.0
    e_all = 0.5 * (n * xx + x0 * x0 + even * xn * xn)
    e_hi = 0.5 * (n * yy + y0 * y0 + even * yn * yn)
    return 10.0 * math.log10(e_hi / (e_all + 1e-20) + 1e-20)
