"""The stage around the DeepFilterNet model against fixture G11, captured from the reference's
Egregora_DeepFilterNet_Denoise.execute run with a documented stand-in for the absent upstream model
(tests/golden/make_golden_dfn.py).

not gpu : node surface; the oracle restatement reproduces every G11 table and output BIT FOR BIT
gpu     : the node through the C ABI with the same stand-in registered as the enhancer backend: helper tables within one
          float32 ulp (the frame RMS is summed in double; sin / cos <= 2e-7), outputs <= 3e-7 absolute, peaks and limiter
          included
"""
import inspect
import json

import numpy as np
import pytest
import torch

from conftest import gjson, gnpz


def fake_wet(x: torch.Tensor, model_name: str = "") -> torch.Tensor:
    y = torch.zeros_like(x)
    y[:, 1:] = 0.8 * x[:, :-1]
    return y


def signal(seed=11, n=96000, C=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 48000.0
    env = (np.sin(2 * np.pi * 1.5 * t) > 0.2).astype(np.float64) * (0.5 + 0.5 * np.sin(2 * np.pi * 0.3 * t) ** 2)
    x = np.stack([env * (0.7 * np.sin(2 * np.pi * (180 + 40 * c) * t) + 0.3 * np.sin(2 * np.pi * 2100 * t)) +
                  0.02 * rng.standard_normal(n) for c in range(C)])
    return (0.9 * x / np.max(np.abs(x))).astype(np.float32)


def test_surface_equals_reference(pack):
    g = gjson("g11_dfn")["surface"]
    cls = pack.NODE_CLASS_MAPPINGS["Egregora_DeepFilterNet_Denoise"]
    it = cls.INPUT_TYPES()
    assert json.loads(json.dumps(it)) == g["INPUT_TYPES"]
    assert {k: list(v.keys()) for k, v in it.items()} == g["widget_order"]
    assert list(cls.RETURN_TYPES) == g["RETURN_TYPES"] and cls.FUNCTION == g["FUNCTION"] and cls.CATEGORY == g["CATEGORY"]
    assert str(inspect.signature(cls.execute)) == g["signature"]
    assert pack.NODE_DISPLAY_NAME_MAPPINGS["Egregora_DeepFilterNet_Denoise"] == g["display"]


def test_oracle_reproduces_the_reference_bit_for_bit():
    from oracle import dfn_mix as o
    g, z = gjson("g11_dfn"), gnpz("g11_dfn")
    x = signal()
    pr = o.vad_probs_rms_48k(x[0])
    assert np.array_equal(pr, z["vad_rms"])
    sm = o.smooth_probs(pr, 60)
    assert np.array_equal(sm, z["vad_smooth60"])
    for m in ("off", "more_on_noise", "more_on_speech", "gate_on_noise"):
        assert np.array_equal(o.strength_per_frame(0.65, sm, m, 0.45, 0.9), z["strength_" + m])
    gd, gw = o.gains(z["strength_more_on_noise"], "equal_power")
    assert np.array_equal(gd, z["g_dry"]) and np.array_equal(gw, z["g_wet"])
    for name, c in g["cases"].items():
        kw = dict(c["kwargs"])
        mono = kw.pop("stereo_mode", "per_channel") != "per_channel"
        d = torch.from_numpy(x)[None].mean(dim=1, keepdim=True)[0].numpy() if mono else x
        wet = np.stack([fake_wet(torch.from_numpy(d[i:i + 1]))[0].numpy() for i in range(d.shape[0])])
        y = o.mix_stage(d, wet, 48000, **kw)
        assert np.array_equal(y[:, ::17], z[name]), name
        assert abs(float(np.abs(y).max()) - c["peak"]) == 0.0


def test_reference_quirk_q6_is_recorded():
    assert gjson("g11_dfn")["vad_none_raises"] == "ValueError"


@pytest.mark.gpu
def test_node_matches_reference_with_the_stand_in_enhancer(pack):
    from egregora_amd import egregora_audio_enhance_extras as X
    g, z = gjson("g11_dfn"), gnpz("g11_dfn")
    x = signal()
    A = {"waveform": torch.from_numpy(x)[None], "sample_rate": 48000, "meta": {"k": 1}}
    node = pack.NODE_CLASS_MAPPINGS["Egregora_DeepFilterNet_Denoise"]()
    X.set_enhancer(fake_wet)
    try:
        for name, c in g["cases"].items():
            (out,) = node.execute(A, **c["kwargs"])
            y = out["waveform"].numpy()
            assert list(y.shape) == c["shape"] and out["sample_rate"] == c["sr"] and y.dtype == np.float32
            assert np.abs(y[0][:, ::17] - z[name]).max() <= 3e-7, (name, float(np.abs(y[0][:, ::17] - z[name]).max()))
            assert abs(float(np.abs(y).max()) - c["peak"]) <= 3e-7
            assert abs(float(y.astype(np.float64).sum()) - c["sum"]) <= 2e-3 and sorted(out["meta"].keys()) == c["meta_keys"]
            got_meta = {k: v for k, v in out["meta"]["deepfilternet"].items() if k != "device"}
            assert got_meta == c["dfn_meta"]
        (q6,) = node.execute(A, adaptive_vad_source="none", post_gain_db=0.0, limit_ceiling=False)      # reference raises here
        s = np.float32(0.65)
        want = np.clip(np.cos(np.float32(0.5 * np.pi) * s) * x + np.sin(np.float32(0.5 * np.pi) * s) * fake_wet(torch.from_numpy(x)).numpy(), -1, 1)
        assert np.abs(q6["waveform"][0].numpy() - want).max() <= 3e-7
    finally:
        X.set_enhancer(None)
    with pytest.raises(RuntimeError, match="DeepFilterNet"):
        node.execute(A)


@pytest.mark.gpu
def test_vad_gain_tables_on_device(pack):
    from egregora_amd import native
    import ctypes as C
    z = gnpz("g11_dfn")
    L = native.lib()
    x = torch.from_numpy(signal()[:1]).cuda().contiguous()
    n48 = x.shape[1]
    nfr = (n48 + 479) // 480
    for mode, key in ((1, "more_on_noise"), (2, "more_on_speech"), (3, "gate_on_noise"), (0, "off")):
        ws = torch.empty(int(L.egr_dfn_workspace_bytes(1, n48)), dtype=torch.uint8, device="cuda")
        gd = torch.empty((1, nfr), device="cuda")
        gw = torch.empty((1, nfr), device="cuda")
        native.check(L.egr_dfn_vad_gains(native.ptr(x), 1, n48, 60.0, mode, 0.65, 0.45, 0.9, 1, native.ptr(ws), native.ptr(gd),
                                         native.ptr(gw), native.stream_ptr()), "gains")
        # linear curve: g_wet IS the per-frame strength.  The frame RMS is summed in double on the device (numpy: float32
        # pairwise), so a strength may differ by one float32 ulp; everything downstream is rounded like numpy.
        got = gw.cpu().numpy()[0]
        assert np.abs(got - z["strength_" + key]).max() <= 1.2e-7, key
        assert np.mean(got == z["strength_" + key]) > 0.9
    native.check(L.egr_dfn_vad_gains(native.ptr(x), 1, n48, 60.0, 1, 0.65, 0.45, 0.9, 0, native.ptr(ws), native.ptr(gd),
                                     native.ptr(gw), native.stream_ptr()), "gains")
    assert np.abs(gd.cpu().numpy()[0] - z["g_dry"]).max() <= 2e-7 and np.abs(gw.cpu().numpy()[0] - z["g_wet"]).max() <= 2e-7
