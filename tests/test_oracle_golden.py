"""Pins the oracle (oracle/*.py) to fixtures captured from the reference itself (G1-G7, G9)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import gjson, gnpz
from oracle import fatllama as ofl
from oracle import glue, metrics


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_g1_chunk_spans():
    g = gjson("g1_chunks")
    assert (g["win"], g["hop"]) == (glue.WIN, glue.HOP)
    for t, e in g["totals"].items():
        sp = glue.chunk_spans(int(t))
        assert len(sp) == e["n"]
        if "spans" in e:
            assert [list(s) for s in sp] == e["spans"]
        else:
            assert list(sp[0]) == e["first"] and list(sp[1]) == e["second"] and list(sp[-1]) == e["last"]
            assert sum(s for s, _ in sp) == e["sum_start"] and sum(l for _, l in sp) == e["sum_len"]
    for e in g["generic"]:
        assert [list(s) for s in glue.chunk_spans(e["total"], e["win"], e["hop"])] == e["spans"]


def test_g2_hann_bit_exact():
    g = gjson("g2_hann")
    h = glue.hann_sym(glue.WIN)
    assert str(h.dtype) == g["dtype"] and sha(h) == g["sha256"]
    assert h[:8].tolist() == g["first8"] and h[-8:].tolist() == g["last8"]
    for n, v in g["small"].items():
        assert glue.hann_sym(int(n)).tolist() == v


def _g3_inputs():
    total = 576000
    x = np.random.Generator(np.random.PCG64(1)).standard_normal((2, total)).astype(np.float32)
    spans = glue.chunk_spans(total)
    return total, x, spans


def test_g3_wola_identity_and_random_bit_exact():
    g, z = gjson("g3_wola"), gnpz("g3_wola")
    total, x, spans = _g3_inputs()
    out = glue.flashsr_node_glue(x, lambda c: c)
    assert sha(out) == g["ident_sha256"]
    assert out[:, 0].tolist() == g["ident_first"] == [0.0, 0.0]      # quirk Q1
    rng2 = np.random.Generator(np.random.PCG64(2))
    lp = z["lp"].tolist()
    preds = [(rng2.standard_normal((2, lp[i % 3])).astype(np.float32), s, L) for i, (s, L) in enumerate(spans)]
    rnd = glue.wola(preds, total)
    assert sha(rnd) == g["rnd_sha256"]
    np.testing.assert_array_equal(rnd[:, ::int(z["dec"])], z["rnd_dec"])
    assert glue.wola([], 5).tolist() == g["empty"]
    for key, e in g["small"].items():
        t, w, h = map(int, key.split("_"))
        pr = [(np.array(p, np.float32), s, L) for p, (s, L) in zip(e["preds"], e["spans"])]
        np.testing.assert_array_equal(glue.wola(pr, t, w), np.array(e["out"], np.float32))


def test_g4_resample_scipy_branch():
    g, z = gjson("g4_resample"), gnpz("g4_resample")
    for key, e in g.items():
        if key == "same_sr":
            continue
        src, dst = map(int, key.split("_"))
        x = z[f"in_noise_{key}"]
        y = glue.resample_hq(x, src, dst)
        assert list(y.shape) == e["noise_shape"] and str(y.dtype) == e["dtype"]
        np.testing.assert_allclose(y, z[f"out_noise_{key}"], rtol=0, atol=2e-6)
    # the direct polyphase restatement agrees with scipy's routine (definition check)
    x = z["in_noise_44100_48000"][0, :600]
    from scipy.signal import resample_poly
    np.testing.assert_allclose(glue.resample_poly_ref(x, 160, 147), resample_poly(x, 160, 147), atol=2e-6)
    np.testing.assert_allclose(glue.resample_poly_ref(x, 2, 1), resample_poly(x, 2, 1), atol=2e-6)
    np.testing.assert_allclose(glue.resample_poly_ref(x, 147, 160), resample_poly(x, 147, 160), atol=2e-6)


def test_g5_shape_heuristics_and_errors():
    g = gjson("g5_shapes")
    for e in g["from_audio_dict"]:
        if "in_shape" in e:
            shp = tuple(e["in_shape"])
            a = np.arange(int(np.prod(shp)), dtype=np.float32).reshape(shp) / 1000.0
            cs, sr = glue.coerce_audio((a, 44100.0))
        else:
            shp = tuple(e["dict_shape"])
            wf = torch.arange(int(np.prod(shp)), dtype=torch.float64).reshape(shp) / 100.0
            cs, sr = glue.coerce_audio({"waveform": wf, "sample_rate": 48000.0})
        assert list(cs.shape) == e["out_shape"] and sr == e["sr"] and sha(cs) == e["sha256"]
        assert str(cs.dtype) == e["dtype"] and isinstance(sr, int)
    for e in g["to_cs"]:
        shp = tuple(e["in_shape"])
        a = (np.arange(int(np.prod(shp)), dtype=np.float32).reshape(shp) - 3.0) * e["scale"]
        cs = glue.to_cs(a)
        assert list(cs.shape) == e["out_shape"] and sha(cs) == e["sha256"]
    for e in g["errors"]:
        if e.get("mod") == "sr" and "shape" in e:
            with pytest.raises(RuntimeError) as ei:
                glue.coerce_audio({"waveform": torch.zeros(e["shape"]), "sample_rate": 1})
            assert str(ei.value) == e["msg"]
    with pytest.raises(RuntimeError):
        glue.coerce_audio(None)


def test_g6_make_audio():
    g = gjson("g6_make_audio")
    d = glue.make_audio(44100.0, np.arange(12, dtype=np.float64).reshape(2, 6))
    assert list(d["waveform"].shape) == g["shape"] and str(d["waveform"].dtype) == g["dtype"]
    assert d["waveform"].is_contiguous() and d["sample_rate"] == g["sr"] and isinstance(d["sample_rate"], int)
    assert list(glue.make_audio(8000, np.arange(4))["waveform"].shape) == g["shape_1d"]
    assert sorted(d.keys()) == g["keys"]


def test_g7_stft_lsd_sisdr():
    g, z = gjson("g7_metrics"), gnpz("g7_stft")
    S = metrics.stft_mag(z["sig"])
    assert list(S.shape) == g["S_shape"] and str(S.dtype) == g["S_dtype"]
    np.testing.assert_array_equal(S, z["S"])
    np.testing.assert_array_equal(metrics.stft_mag(z["sig2"]), z["S2"])
    np.testing.assert_array_equal(metrics.stft_mag(z["sig"][:1000]), z["Sshort"])
    np.testing.assert_array_equal(metrics.stft_mag(z["sig"][:20000], 1024, 256), z["S_1024_256"])
    assert list(metrics.lsd(S, S)) == g["lsd_self"]
    assert list(metrics.lsd(S, z["Sp"])) == pytest.approx(g["lsd_pert"], rel=1e-12)
    assert list(metrics.lsd(S, metrics.stft_mag((z["sig"] * np.float32(1.001)).astype(np.float32)))) == \
        pytest.approx(g["lsd_gain_1p001"], rel=1e-9)
    assert metrics.si_sdr(z["sig"], z["pert"]) == pytest.approx(g["si_sdr_pert"], rel=1e-12)
    assert metrics.si_sdr(z["sig2"], z["sig2"][:, :20000] * 0.9) == pytest.approx(g["si_sdr_stereo"], rel=1e-9)


def test_g9_write_patch_and_handover():
    g = gjson("g9_fatllama_adapter")
    data = np.array([16384.0, -32768.0, 100.0], np.float32)
    assert ofl.write_patch_scale(data).astype(np.float64).tolist() == g["gpu"]["write_patch"]["gt1_sw2"]
    spec_none = ofl.FatLlamaSpec(sample_width=0)
    assert ofl.write_patch_scale(data, spec_none).astype(np.float64).tolist() == g["gpu"]["write_patch"]["gt1_swNone"]
    le1 = np.array([0.5, -1.0, 0.25], np.float32)
    assert ofl.write_patch_scale(le1).astype(np.float64).tolist() == g["gpu"]["write_patch"]["le1"]
    assert g["gpu"]["write_patch"] == g["cpu"]["write_patch"]
    # the dict path hands frames-first [S,C] float data to the WAV writer un-normalised
    assert g["dict_path"]["written_shape"] == [3, 2] and g["dict_path"]["write_extra_args"] == [[], {}]


def test_fatllama_oracle_selfconsistency():
    """rfft formulation == full complex formulation up to float32 rounding; idempotence after iter 1."""
    rng = np.random.Generator(np.random.PCG64(11))
    x = np.rint(rng.standard_normal(3000) * 3000).astype(np.float32)
    y = ofl.interpolate(x, 2)
    assert y.shape == (6000,) and y[0] == x[0] and y[1] == np.float32(0.5) * (x[0] + x[1])
    assert y[-2] == 0 and y[-1] == 0
    tr = []
    d = ofl.ist_loop(y, 5, 0.6, trace=tr)
    d_r = ofl.ist_loop(y, 5, 0.6, ofl.FatLlamaSpec(use_rfft=True))
    scale = np.max(np.abs(y))
    assert np.max(np.abs(d - d_r)) / scale < 2e-6
    assert np.max(np.abs(tr[-1] - tr[0])) / scale < 2e-6
    assert ofl.upscale_factor(16000, 1, 1411) == 6 and ofl.upscale_factor(48000, 2, 1536) == 1
    assert ofl.upscale_factor(48000, 2, 1411) == 1 and ofl.upscale_factor(48000, 2, 64) == 1
    q = ofl.pcm16_write(np.array([0.0, 1.0, -1.0, 0.5, 1.5], np.float32))
    assert q.tolist() == [0, 32767, -32767, 16384, 49150 - 65536]  # 1.5*32767=49150.5 -> rint-even 49150 -> wraps
