"""The drop-in boundary: node keys, display names, INPUT_TYPES, class attributes and run() signatures must
equal what the reference exposes (fixture G8 captured by importing the reference)."""
import inspect

import numpy as np
import pytest
import torch

from conftest import gjson


def test_mappings_and_surface_equal_reference(pack):
    g = gjson("g8_surface")
    core = {"EgregoraAudioUpscaler", "EgregoraFatLlamaGPU", "EgregoraFatLlamaCPU"}
    assert set(pack.NODE_CLASS_MAPPINGS) == core | {"Metrics (LSD + SI-SDR)", "Resample Audio (HQ)", "Egregora_DeepFilterNet_Denoise",
                                                   "Audio Align (XCorr)", "Audio Gain Match", "Audio Null Test", "Audio Plotter", "Null Test (Full)"}
    for key in sorted(core):
        assert pack.NODE_DISPLAY_NAME_MAPPINGS[key] == g["display"][key]
        cls = pack.NODE_CLASS_MAPPINGS[key]
        e = g[key]
        assert cls.__name__ == g["class_names"][key]
        got = cls.INPUT_TYPES()
        # JSON turns tuples into lists; compare structurally
        import json
        assert json.loads(json.dumps(got)) == e["INPUT_TYPES"]
        assert list(cls.RETURN_TYPES) == e["RETURN_TYPES"] and isinstance(cls.RETURN_TYPES, tuple)
        assert cls.FUNCTION == e["FUNCTION"] and cls.CATEGORY == e["CATEGORY"] and cls.OUTPUT_NODE == e["OUTPUT_NODE"]
        assert str(inspect.signature(getattr(cls, cls.FUNCTION))) == e["run_signature"]
        # widget order matters to saved workflows
        assert {k: list(v.keys()) for k, v in got.items()} == e["widget_order"]


def test_per_module_mappings_exist(pack):
    from egregora_amd import egregora_audio_super_resolution as a, egregora_fat_llama_cpu as c, egregora_fat_llama_gpu as b
    assert list(a.NODE_CLASS_MAPPINGS) == ["EgregoraAudioUpscaler"]
    assert list(b.NODE_CLASS_MAPPINGS) == ["EgregoraFatLlamaGPU"]
    assert list(c.NODE_CLASS_MAPPINGS) == ["EgregoraFatLlamaCPU"]


def test_runner_constants(pack):
    from egregora_amd import audio_glue as ag
    g = gjson("g8_surface")["runner_consts"]
    assert (ag.REQ_SR, ag.CHUNK_S, ag.OVERLAP_S, ag.CHUNK_SAMPLES, ag.HOP_SAMPLES) == \
        (g["REQ_SR"], g["CHUNK_S"], g["OVERLAP_S"], g["CHUNK_SAMPLES"], g["HOP"])


def test_host_glue_matches_reference_fixtures(pack):
    import hashlib
    from egregora_amd import audio_glue as ag
    from egregora_amd.egregora_fat_llama_gpu import resolve_input
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    g1 = gjson("g1_chunks")
    for t, e in g1["totals"].items():
        sp = ag.spans(int(t))
        assert len(sp) == e["n"]
        if "spans" in e:
            assert [list(s) for s in sp] == e["spans"]
        else:
            assert list(sp[-1]) == e["last"] and sum(s for s, _ in sp) == e["sum_start"]
    for e in g1["generic"]:
        if e["hop"] <= e["win"]:
            assert [list(s) for s in ag.spans(e["total"], e["win"], e["hop"])] == e["spans"]
    g5 = gjson("g5_shapes")
    for e in g5["from_audio_dict"]:
        if "in_shape" in e:
            shp = tuple(e["in_shape"])
            a = np.arange(int(np.prod(shp)), dtype=np.float32).reshape(shp) / 1000.0
            t, sr = ag.upscaler_input((a, 44100.0))
        else:
            shp = tuple(e["dict_shape"])
            wf = torch.arange(int(np.prod(shp)), dtype=torch.float64).reshape(shp) / 100.0
            t, sr = ag.upscaler_input({"waveform": wf, "sample_rate": 48000.0})
        assert list(t.shape) == e["out_shape"] and sr == e["sr"] and sha(t.numpy()) == e["sha256"]
    for e in g5["to_cs"]:
        if e["mod"] != "gpu":
            continue
        shp = tuple(e["in_shape"])
        a = (np.arange(int(np.prod(shp)), dtype=np.float32).reshape(shp) - 3.0) * e["scale"]
        cs = ag.channels_first(a)
        assert list(cs.shape) == e["out_shape"] and sha(cs) == e["sha256"]
    # error behaviour: same exception type and text as the reference
    for e in g5["errors"]:
        if e.get("raised") is None:
            continue
        if "shape" in e:
            bad = {"waveform": torch.zeros(e["shape"]), "sample_rate": 1}
            fn = (lambda: ag.upscaler_input(bad)) if e["mod"] == "sr" else (lambda: resolve_input(bad))
        elif e["mod"] == "sr":
            fn = lambda: ag.upscaler_input(None)
        elif e["mod"] == "gpu_path":
            fn = lambda: resolve_input(None, "/nonexistent/x.wav", "")
        else:
            fn = lambda: resolve_input(None, "", "")
        with pytest.raises(RuntimeError) as ei:
            fn()
        assert str(ei.value) == e["msg"]
    # tuple path applies the peak>1 rescale, dict path does not (fixture G9)
    g9 = gjson("g9_fatllama_adapter")
    cs, sr = resolve_input((np.array([[0.5, 2.0], [-4.0, 1.0], [0.25, 0.0]]), 22050))
    np.testing.assert_array_equal(cs.numpy().T.astype(np.float64), np.array(g9["tuple_path"]["written"]))
    wf = torch.tensor([[[0.1, -0.2, 0.3], [1.5, -2.5, 0.0]]], dtype=torch.float32)
    cs, sr = resolve_input({"waveform": wf, "sample_rate": 44100})
    np.testing.assert_array_equal(cs.numpy().T.astype(np.float64), np.array(g9["dict_path"]["written"]))
    d = ag.package(44100.0, torch.arange(12, dtype=torch.float64).reshape(2, 6))
    g6 = gjson("g6_make_audio")
    assert list(d["waveform"].shape) == g6["shape"] and str(d["waveform"].dtype) == g6["dtype"]
    assert d["waveform"].is_contiguous() and isinstance(d["sample_rate"], int) and sorted(d) == g6["keys"]


def test_upscale_factor_matches_oracle(pack):
    from egregora_amd import fatllama_engine as fe
    from oracle import fatllama as ofl
    for sr in (8000, 16000, 22050, 44100, 48000, 96000):
        for ch in (1, 2):
            for kb in (64, 256, 1411, 1536, 3072, 5000):
                assert fe.upscale_factor(sr, ch, kb) == ofl.upscale_factor(sr, ch, kb)


def test_wav_reader_roundtrip(pack, tmp_path):
    import struct
    from egregora_amd import wavio
    pcm = np.array([[0, 100], [-32768, 32767], [1, -1]], "<i2")
    body = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 22050, 22050 * 4, 4, 16)
    p = tmp_path / "t.wav"
    p.write_bytes(hdr + b"data" + struct.pack("<I", len(body)) + body)
    y, sr = wavio.read_wav(str(p))
    assert sr == 22050 and y.shape == (3, 2)
    np.testing.assert_array_equal(y, pcm.astype(np.float32) / 32768.0)
    with pytest.raises(RuntimeError):
        wavio.read_wav_bytes(b"fLaC" + b"\0" * 40)


def test_resampler_filter_design_equals_scipy(pack):
    """The host-side FIR design is scipy.signal.firwin's (which the reference reaches through resample_poly)."""
    from scipy.signal import firwin
    from egregora_amd import resample
    for up, down in [(160, 147), (2, 1), (147, 160), (3, 1), (1, 2)]:
        h, half = resample.design_filter(up, down)
        mx = max(up, down)
        want = firwin(2 * 10 * mx + 1, 1.0 / mx, window=("kaiser", 5.0)).astype(np.float32)
        want *= np.float32(up)
        assert half == 10 * mx and h.dtype == np.float32
        assert np.max(np.abs(h - want)) <= 1e-7 * np.max(np.abs(want))
        assert np.mean(h == want) > 0.99
    assert resample.rates(44100, 48000) == (160, 147) and resample.rates(48000, 96000) == (2, 1)
