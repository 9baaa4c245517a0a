"""Evaluation-pack nodes built on the device kernels against fixture G10, captured
from the reference's egregora_audio_eval_pack.py by tests/golden/make_golden_eval.py.

not gpu : node surface, AUDIO coercion rules, the oracle's metrics on the G10 inputs
gpu     : Metrics (LSD + SI-SDR) and Resample Audio (HQ) executed through the C ABI
Tolerances: LSD 2e-5 relative (float32 log10 / STFT round-off; the reference transforms in float64), SI-SDR 1e-6 dB,
resampled samples 3e-6 absolute (fixture G4's bound), linear mode 1e-6.
"""
import inspect
import json

import numpy as np
import pytest
import torch

from conftest import gjson, gnpz


def signals():
    """Same seeded inputs as tests/golden/make_golden_eval.py."""
    rng = np.random.Generator(np.random.PCG64(10))
    n = 60000
    t = np.arange(n) / 48000.0
    a = np.stack([0.3 * np.sin(2 * np.pi * 330 * t) + 0.1 * np.sin(2 * np.pi * 7000 * t) + 0.02 * rng.standard_normal(n),
                  0.25 * np.sin(2 * np.pi * 500 * t + 0.3) + 0.02 * rng.standard_normal(n)]).astype(np.float32)
    b = (a * np.float32(0.97) + 2e-3 * rng.standard_normal(a.shape)).astype(np.float32)[:, :59000]
    m = (0.4 * np.sin(2 * np.pi * 1000 * np.arange(44100) / 44100.0) + 0.05 * rng.standard_normal(44100)).astype(np.float32)
    return a, b, m


def test_eval_node_surface_equals_reference(pack):
    g = gjson("g10_eval")
    for key, e in g["surface"].items():
        cls = pack.NODE_CLASS_MAPPINGS[key]
        assert pack.NODE_DISPLAY_NAME_MAPPINGS[key] == g["display"][key]
        assert cls.__name__ == g["class_names"][key]
        it = cls.INPUT_TYPES()
        assert json.loads(json.dumps(it)) == e["INPUT_TYPES"]
        assert {k: list(v.keys()) for k, v in it.items()} == e["widget_order"]
        assert list(cls.RETURN_TYPES) == e["RETURN_TYPES"] and list(cls.RETURN_NAMES) == e["RETURN_NAMES"]
        assert cls.FUNCTION == e["FUNCTION"] and cls.CATEGORY == e["CATEGORY"]
        assert str(inspect.signature(getattr(cls, cls.FUNCTION))) == e["signature"]


def test_audio_coercion_rules(pack):
    from egregora_amd import egregora_audio_eval_pack as ev
    g = gjson("g10_eval")
    for s, want in g["normalize_CN"].items():
        shape = tuple(int(v) for v in s.strip("()").split(",") if v.strip())
        assert list(ev.normalize_cn(np.zeros(shape, np.float32)).shape) == want
    d = ev.to_internal_audio({"waveform": torch.zeros(2, 3, 50), "sample_rate": 8000})      # batch element 0 only
    assert sorted(d.keys()) == g["resample"]["keys"] and d["samples"].shape == (3, 50) and d["waveform"].shape == (1, 3, 50)
    d = ev.to_internal_audio({"sr": 16000, "samples": np.zeros((70, 2))})
    assert d["samples"].shape == (2, 70) and d["sample_rate"] == 16000
    with pytest.raises(ValueError, match="Unsupported AUDIO object"):
        ev.to_internal_audio((np.zeros(4), 8000))
    with pytest.raises(ValueError, match="missing samples"):
        ev.to_internal_audio({"sr": 8000})


def test_oracle_metrics_reproduce_the_reference_node():
    from oracle import metrics as om
    g = gjson("g10_eval")
    a, b, _ = signals()
    am, bm = a.mean(axis=0)[:59000], b.mean(axis=0)
    mean, p95 = om.lsd_audio(am, bm)
    assert abs(mean - g["metrics_default"]["lsd_mean_db"]) < 1e-6 and abs(p95 - g["metrics_default"]["lsd_p95_db"]) < 1e-6
    assert abs(om.si_sdr(am, bm) - g["metrics_default"]["si_sdr_db"]) < 1e-9
    mean, p95 = om.lsd_audio(am, bm, 1024, 256)
    assert abs(mean - g["metrics_1024_256"]["lsd_mean_db"]) < 1e-6


@pytest.mark.gpu
def test_metrics_node_matches_reference(pack):
    g = gjson("g10_eval")
    a, b, _ = signals()
    A = {"waveform": torch.from_numpy(a)[None], "sample_rate": 48000}
    B = {"waveform": torch.from_numpy(b)[None], "sample_rate": 48000}
    node = pack.NODE_CLASS_MAPPINGS["Metrics (LSD + SI-SDR)"]()
    for name, kw in (("metrics_default", {}), ("metrics_1024_256", dict(n_fft=1024, hop=256)),
                     ("metrics_lsd_only", dict(compute_si_sdr=False))):
        (got,) = node.execute(A, B, **kw)
        want = g[name]
        assert sorted(got) == sorted(want)
        for k in ("lsd_mean_db", "lsd_p95_db"):
            assert abs(got[k] - want[k]) <= 2e-5 * want[k], (name, k, got[k], want[k])
        if "si_sdr_db" in want:
            assert abs(got["si_sdr_db"] - want["si_sdr_db"]) <= 1e-6, (got["si_sdr_db"], want["si_sdr_db"])
    (self_,) = node.execute(A, A)
    assert abs(self_["lsd_mean_db"] - 1e-6) < 1e-9 and abs(self_["lsd_p95_db"] - 1e-6) < 1e-9      # the sqrt(1e-12) floor
    assert self_["si_sdr_db"] > 150.0           # reference: 230.9 dB, i.e. round-off of alpha in float64


@pytest.mark.gpu
def test_order_statistics_and_sum_kernels(pack):
    import ctypes as C
    from egregora_amd import native
    L = native.lib()
    rng = np.random.Generator(np.random.PCG64(3))
    for n in (1, 2, 17, 5000, 200001):
        v = rng.standard_normal(n).astype(np.float32)
        if n > 10:
            v[::7] = v[3]                      # ties
        vt = torch.from_numpy(v).cuda()
        srt = np.sort(v)
        for k0, k1 in ((0, n - 1), (n // 2, min(n - 1, n // 2 + 1)), (int(0.95 * (n - 1)), min(n - 1, int(0.95 * (n - 1)) + 1))):
            o = torch.empty(2, device="cuda")
            native.check(L.egr_order_stats2(native.ptr(vt), n, k0, k1, native.ptr(o), native.stream_ptr()), "order")
            assert o.cpu().numpy().tolist() == [float(srt[k0]), float(srt[k1])]
        t = torch.empty(1, dtype=torch.float64, device="cuda")
        native.check(L.egr_sum_f64(native.ptr(vt), n, native.ptr(t), native.stream_ptr()), "sum")
        assert abs(float(t.cpu()) - float(v.astype(np.float64).sum())) <= 1e-9 * max(1.0, n)


@pytest.mark.gpu
def test_resample_node_matches_reference(pack):
    g = gjson("g10_eval")
    z = gnpz("g10_eval")
    a, _, m = signals()
    node = pack.NODE_CLASS_MAPPINGS["Resample Audio (HQ)"]()
    (r48,) = node.execute({"waveform": torch.from_numpy(m)[None, None], "sample_rate": 44100}, 48000, "scipy_polyphase")
    assert list(r48["waveform"].shape) == g["resample"]["m_to_48k_shape"] and r48["sample_rate"] == 48000
    assert sorted(r48.keys()) == g["resample"]["keys"] and r48["waveform"].dtype == torch.float32
    assert np.abs(r48["samples"][:, ::37] - z["m_to_48k"]).max() <= 3e-6
    A = {"waveform": torch.from_numpy(a)[None], "sample_rate": 48000}
    (r2,) = node.execute(A, 44100, "auto")
    assert list(r2["waveform"].shape) == g["resample"]["a_to_441_shape"]
    assert np.abs(r2["samples"][:, ::53] - z["a_to_441"]).max() <= 3e-6
    (same,) = node.execute(A, 48000)
    assert list(same["waveform"].shape) == g["resample"]["same_rate_returns_input_shape"]
    # linear branch (also taken for mode="torchaudio" when torchaudio is absent, as in the reference): numpy definition
    for mode in ("linear", "torchaudio"):
        (lin,) = node.execute(A, 32000, mode)
        n_new = int(round(a.shape[1] * (32000 / 48000)))
        t_old = np.linspace(0.0, 1.0, a.shape[1], endpoint=False)
        t_new = np.linspace(0.0, 1.0, n_new, endpoint=False)
        want = np.stack([np.interp(t_new, t_old, a[c]) for c in range(2)]).astype(np.float32)
        assert lin["samples"].shape == want.shape and np.abs(lin["samples"] - want).max() <= 1e-6


@pytest.mark.gpu
def test_stft_magnitude_on_the_whole_widget_grid(pack):
    """The reference's Metrics / Null Test widgets offer n_fft = 512 .. 8192 in steps of 128 (np.fft.rfft takes any of them); 19 of
    the 61 values have a prime factor above 13 in n_fft / 2 (2176 = 2^7 * 17, 2432, 2944, ...) and run one generic radix stage.
    Every grid value against the oracle's float64 STFT: <= 3e-6 of the frame maximum."""
    import torch
    from egregora_amd import device_ops
    from oracle import metrics as om
    rng = np.random.Generator(np.random.PCG64(5))
    x = (0.3 * rng.standard_normal((2, 20000))).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    worst = 0.0
    for n_fft in range(512, 8192 + 1, 128):
        got = device_ops.stft_mag(xt, n_fft, n_fft // 4).cpu().numpy()          # [bins][frames] like the oracle
        want = om.stft_mag(x, n_fft, n_fft // 4)
        assert got.shape == want.shape, n_fft
        err = float(np.abs(got - want).max() / np.abs(want).max())
        worst = max(worst, err)
        assert err <= 3e-6, (n_fft, err)
    print(f"\nSTFT grid 512..8192/128: worst relative error {worst:.2e}")
