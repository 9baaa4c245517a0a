"""ORACLE (test infrastructure, not product): numpy restatement of the reference's spectral metrics,
which define the north-star parity yardstick ("within 1e-3 LSD").

Parity status: PINNED against fixture G7 (tests/golden/g7_*), captured from the reference.
Follows egregora_audio_eval_pack.py:389-429 (duplicate at egregora_null_test_suite.py:167-189).
"""
import numpy as np


def stft_mag(x, n_fft=2048, hop=512):
    """|rFFT| of Hann(n_fft, symmetric, float32)-windowed frames, hop, no centring, mono downmix by
    mean over channels, frames = 1 + max(0,(N-n_fft)//hop), short frame zero-padded.
    Output [n_fft/2+1, frames] float32.  egregora_audio_eval_pack.py:389-402."""
    x = np.asarray(x)
    mono = x if x.ndim == 1 else x.mean(axis=0)
    N = mono.shape[0]
    k = np.arange(n_fft, dtype=np.float64)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * k / (n_fft - 1))).astype(np.float32)
    frames = 1 + max(0, (N - n_fft) // hop)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    padded = mono if N >= n_fft else np.pad(mono, (0, n_fft - N))
    fr = padded[idx] * win[None, :]           # float32 product as in the reference
    X = np.fft.rfft(fr, axis=1)               # float64 transform (numpy upcasts), like the reference
    return np.abs(X).astype(np.float32).T.copy()


def lsd(SA, SB):
    """(mean, p95) over frames of sqrt(mean_bins((20log10(A+eps)-20log10(B+eps))^2)+1e-12).  :405-411."""
    eps = 1e-12
    LA = 20 * np.log10(SA + eps)
    LB = 20 * np.log10(SB + eps)
    per = np.sqrt(np.mean((LA - LB) ** 2, axis=0) + 1e-12)
    return float(np.mean(per)), float(np.percentile(per, 95))


def lsd_audio(a, b, n_fft=2048, hop=512):
    n = min(a.shape[-1], b.shape[-1])
    return lsd(stft_mag(a[..., :n], n_fft, hop), stft_mag(b[..., :n], n_fft, hop))


def lsd_masked(a, b, floor_db=None, n_fft=2048, hop=512, f32_run=None, margin_db=80.0):
    """Mean over frames of the log-spectral distance between `a` (the trusted float64 run) and `b`, restricted to the STFT bins
    where a float32 implementation CAN hold the north star's 1e-3 dB.  Test-only companion of `lsd`; not a reference function.

    A float32 transform chain leaves a round-off floor e in every bin; a bin of magnitude S then moves by 8.7 e/S dB, so 1e-3 dB
    needs S >= 8700 e whatever the implementation (upstream's included).  The floor is MEASURED, not assumed: e = rms over all
    bins of |STFT(f32_run)| - |STFT(a)|, where f32_run is the float32 ORACLE's output for the same input (never the device's), and
    the bins kept are those with S >= e * 10^(margin_db/20) (80 dB: the float32 oracle itself sits at 1e-4 relative there).
    Alternatively floor_db keeps the bins within floor_db of the largest magnitude.  Returns (lsd_db, fraction of bins kept)."""
    n = min(a.shape[-1], b.shape[-1])
    SA, SB = stft_mag(a[..., :n], n_fft, hop).astype(np.float64), stft_mag(b[..., :n], n_fft, hop).astype(np.float64)
    if f32_run is not None:
        SO = stft_mag(f32_run[..., :n], n_fft, hop).astype(np.float64)
        keep = SA >= np.sqrt(np.mean((SO - SA) ** 2)) * 10.0 ** (margin_db / 20.0)
    else:
        keep = SA > SA.max() * 10.0 ** (-float(floor_db) / 20.0)
    d = (20 * np.log10(SA + 1e-12) - 20 * np.log10(SB + 1e-12)) ** 2
    per = np.sqrt((d * keep).sum(axis=0) / np.maximum(keep.sum(axis=0), 1) + 1e-12)
    return float(np.mean(per)), float(keep.mean())


def si_sdr(s, s_hat):
    """Scale-invariant SDR in dB on the mono downmix, float64.  :414-429."""
    s = np.asarray(s, np.float64)
    s_hat = np.asarray(s_hat, np.float64)
    if s.ndim > 1:
        s = s.mean(axis=0)
    if s_hat.ndim > 1:
        s_hat = s_hat.mean(axis=0)
    n = min(s.shape[-1], s_hat.shape[-1])
    s, s_hat = s[:n], s_hat[:n]
    alpha = np.dot(s_hat, s) / (np.dot(s, s) + 1e-20)
    tgt = alpha * s
    err = s_hat - tgt
    return 10.0 * np.log10((np.dot(tgt, tgt) + 1e-20) / (np.dot(err, err) + 1e-20))


def xcorr_delay(a, b, sr, max_shift_smp):
    """GCC-PHAT delay of b against a with parabolic peak refinement; follows egregora_null_test_suite.py:213-237.

    Transform length n = smallest power of two >= len(a) + len(b); cross-spectrum normalised by (|.| + 1e-12); the
    correlation is re-centred so that index n/2 - 1 holds lag 0 (np.roll by n/2 - 1) while the search window is centred on
    n/2 -- hence the reference's result is the true lag minus one (quirk Q8, pinned by fixture G12)."""
    n = 1 << max(0, int(np.ceil(np.log2(max(1, a.size + b.size)))))
    cross = np.fft.rfft(b, n=n) * np.conj(np.fft.rfft(a, n=n))
    cross /= np.abs(cross) + 1e-12
    centred = np.roll(np.fft.irfft(cross, n=n), n // 2 - 1)
    mid = n // 2
    lo = mid - max_shift_smp
    peak = lo + int(np.argmax(centred[lo:mid + max_shift_smp + 1]))
    shift = 0.0
    if 0 < peak < n - 1:
        left, top, right = centred[peak - 1], centred[peak], centred[peak + 1]
        curv = 2 * (left - 2 * top + right)
        if abs(curv) >= 1e-12:
            shift = (left - right) / curv
    return float(peak - mid + shift)
