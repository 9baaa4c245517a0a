"""`Audio Align (XCorr)` of the null-test suite against fixture G13, captured from the reference
node (tests/golden/make_golden_align.py).

not gpu : node surface; the FIR design equals the reference's taps (through the oracle of the shift + FIR arithmetic)
gpu     : the node through the C ABI: delay within 5e-3 samples of the reference, aligned audio within 1e-2 of its peak (the taps
          follow the estimated fraction), shapes / rates / meta / return tuple as the reference; plus egr_shift_fir against
          numpy's shift + convolve("same") on its own (1e-6)
"""
import inspect
import json
import math

import numpy as np
import pytest
import torch

from conftest import gjson, gnpz


def cases():
    rng = np.random.Generator(np.random.PCG64(13))
    out = {}
    for name, n, d, sr, sr2, kw in (("plus5.3", 24000, 5.3, 48000, 48000, {}), ("minus12.6", 30000, -12.6, 48000, 48000, dict(fir_len=32)),
                                    ("int_only", 20000, 7.4, 44100, 44100, dict(fractional=False, max_shift_ms=50)),
                                    ("other_rate", 24000, 3.0, 48000, 32000, dict(max_shift_ms=20))):
        a = rng.standard_normal((2, n)).astype(np.float32)
        a[1] = 0.5 * a[0] + 0.5 * a[1]
        k = np.arange(-32, 33)
        h = np.sinc(k - d + np.round(d)) * np.hanning(65)
        b = np.stack([np.convolve(np.roll(a[c], int(np.round(d))), h, mode="same") for c in range(2)]) * 0.8
        b = (b + 0.01 * rng.standard_normal(b.shape)).astype(np.float32)[:, : n - 100]
        if sr2 != sr:
            t_old = np.linspace(0.0, 1.0, b.shape[1], endpoint=False)
            n2 = int(round(b.shape[1] * sr2 / sr))
            b = np.stack([np.interp(np.linspace(0.0, 1.0, n2, endpoint=False), t_old, b[c]) for c in range(2)]).astype(np.float32)
        out[name] = (a, sr, b, sr2, kw)
    return out


def np_shift_fir(x, delay, taps, n_out):
    """numpy restatement of the reference's _apply_frac_delay_CN + _pad_or_crop_CN (egregora_null_test_suite.py:203-266)."""
    C, N = x.shape
    y = x.copy()
    if abs(delay) >= 1e-6:
        int_d = int(math.floor(abs(delay)))
        frac = abs(delay) - int_d
        y = np.zeros_like(x)
        if int_d < N:
            if delay >= 0:
                y[:, int_d:] = x[:, :N - int_d]
            else:
                y[:, :N - int_d] = x[:, int_d:]
        if frac > 1e-6:
            m = max(16, int(taps))
            n = np.arange(m)
            h = (np.sinc(n - (m - 1) / 2.0 - frac) * np.hanning(m)).astype(np.float32)
            h /= np.sum(h)
            y = np.stack([np.convolve(y[c], h, mode="same").astype(np.float32) for c in range(C)])
    out = np.zeros((C, n_out), np.float32)
    out[:, :min(N, n_out)] = y[:, :n_out]
    return out


def test_surface_equals_reference(pack):
    g = gjson("g13_align")["surface"]
    cls = pack.NODE_CLASS_MAPPINGS["Audio Align (XCorr)"]
    it = cls.INPUT_TYPES()
    assert json.loads(json.dumps(it)) == g["INPUT_TYPES"]
    assert {k: list(v.keys()) for k, v in it.items()} == g["widget_order"]
    assert list(cls.RETURN_TYPES) == g["RETURN_TYPES"] and list(cls.RETURN_NAMES) == g["RETURN_NAMES"]
    assert cls.FUNCTION == g["FUNCTION"] and cls.CATEGORY == g["CATEGORY"]
    assert str(inspect.signature(cls.execute)) == g["signature"]
    assert pack.NODE_DISPLAY_NAME_MAPPINGS["Audio Align (XCorr)"] == g["display"]


def test_restatement_reproduces_the_reference_alignment():
    """The numpy restatement above, fed the reference's own delay, reproduces the reference's aligned audio bit for bit."""
    from oracle import metrics as om
    g, z = gjson("g13_align"), gnpz("g13_align")
    for name, (a, sr, b, sr2, kw) in cases().items():
        if sr2 != sr:
            continue                     # the rate match is np.interp: exercised on the device in the gpu test
        n = min(a.shape[1], b.shape[1])
        lag = om.xcorr_delay(a[:, :n].mean(axis=0), b[:, :n].mean(axis=0), sr, int(sr * (kw.get("max_shift_ms", 200) / 1000.0)))
        assert lag == g["cases"][name]["delay_samples"]
        comp = -lag if kw.get("fractional", True) else -round(lag)
        y = np_shift_fir(b, comp, kw.get("fir_len", 64), a.shape[1])
        assert np.array_equal(y[:, ::23], z[name]), name


@pytest.mark.gpu
def test_shift_fir_kernel_vs_numpy(pack):
    from egregora_amd import egregora_null_test_suite as nt
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.standard_normal((3, 5000)).astype(np.float32)
    for delay, taps, n_out in ((0.0, 64, 5000), (3.0, 64, 5000), (-4.0, 64, 4000), (2.37, 64, 6000), (-7.81, 33, 5000), (0.25, 16, 5000),
                               (6000.5, 64, 5000)):
        got = nt.apply_delay(torch.from_numpy(x).cuda(), delay, taps, n_out).cpu().numpy()
        assert np.abs(got - np_shift_fir(x, delay, taps, n_out)).max() <= 2e-6, (delay, taps, n_out)


@pytest.mark.gpu
def test_node_matches_reference(pack):
    g, z = gjson("g13_align"), gnpz("g13_align")
    node = pack.NODE_CLASS_MAPPINGS["Audio Align (XCorr)"]()
    for name, (a, sr, b, sr2, kw) in cases().items():
        out, d_s, d_ms, pk, img = node.execute({"waveform": torch.from_numpy(a)[None], "sample_rate": sr},
                                               {"waveform": torch.from_numpy(b)[None], "sample_rate": sr2, "meta": {"m": 2}}, **kw)
        c = g["cases"][name]
        assert abs(d_s - c["delay_samples"]) <= 5e-3 and abs(d_ms - c["delay_ms"]) <= 1e-3 and pk == c["peak_corr"], (name, d_s)
        assert list(out["waveform"].shape) == c["shape"] and out["sample_rate"] == c["sr"] and sorted(out.keys()) == c["keys"]
        assert out["meta"] == c["meta"] and img.dim() == c["image_ndim"] and img.shape[-1] == c["image_last"]
        want = z[name]
        assert np.abs(out["samples"][:, ::23] - want).max() <= 1e-2 * np.abs(want).max(), name
