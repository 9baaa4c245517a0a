#!/usr/bin/env python3
"""Golden fixture G11: the stage AROUND the DeepFilterNet model in the reference's
`Egregora_DeepFilterNet_Denoise.execute` (egregora_audio_enhance_extras.py:450-724) -- RMS VAD in 10 ms frames, one-pole
smoothing, adaptive strength, equal-power / linear wet-dry gains, clip, post-gain, ceiling limiter.

The upstream model (`df.enhance`) is absent from the build image, so the reference is run with a FAKE `df` package whose
`enhance()` is a fixed, documented stand-in (wet = 0.8 * x delayed by one sample); everything else is the reference's own
code.  `torchaudio` is imported at the reference module's top but not used on this path: an empty stub satisfies the import.
Data only (inputs are regenerated from seeds by the tests).

  python tests/golden/make_golden_dfn.py        # writes tests/golden/g11_dfn.json, g11_dfn.npz
"""
import importlib.util
import inspect
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def fake_wet(x: torch.Tensor) -> torch.Tensor:
    """The stand-in for df.enhance.enhance(model, state, xin): [1,T] -> [1,T]."""
    y = torch.zeros_like(x)
    y[:, 1:] = 0.8 * x[:, :-1]
    return y


def install_fakes():
    sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    df = types.ModuleType("df")
    enh = types.ModuleType("df.enhance")
    io_ = types.ModuleType("df.io")

    class _M:
        def to(self, d): return self
        def eval(self): return self
    enh.init_df = lambda name, config_allow_defaults=True: (_M(), object(), None)
    enh.enhance = lambda model, state, xin: fake_wet(xin)

    def _res(x, a, b):
        raise RuntimeError("fixture runs at 48 kHz only")
    io_.resample = _res
    df.enhance, df.io = enh, io_
    sys.modules.update({"df": df, "df.enhance": enh, "df.io": io_})


def signal(seed=11, n=96000, C=2):
    """Speech-like bursts over a noise floor, peak ~0.9 (so clip / limiter paths are exercised)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 48000.0
    env = (np.sin(2 * np.pi * 1.5 * t) > 0.2).astype(np.float64) * (0.5 + 0.5 * np.sin(2 * np.pi * 0.3 * t) ** 2)
    x = np.stack([env * (0.7 * np.sin(2 * np.pi * (180 + 40 * c) * t) + 0.3 * np.sin(2 * np.pi * 2100 * t)) +
                  0.02 * rng.standard_normal(n) for c in range(C)])
    return (0.9 * x / np.max(np.abs(x))).astype(np.float32)


CASES = {
    "default": {},
    "linear_off": dict(mix_curve="linear", adaptive_mode="off", strength=0.4, post_gain_db=0.0),
    "speech": dict(adaptive_mode="more_on_speech", adaptive_amount=0.8, vad_smooth_ms=0, post_gain_db=6.0, ceiling=0.7),
    "gate": dict(adaptive_mode="gate_on_noise", adaptive_amount=0.6, vad_threshold=0.5, vad_smooth_ms=200, limit_ceiling=False,
                 post_gain_db=3.0),
    "mono_full": dict(stereo_mode="downmix_mono", strength=1.0, adaptive_mode="more_on_noise"),
}


def main():
    install_fakes()
    spec = importlib.util.spec_from_file_location("ref_extras", REF / "egregora_audio_enhance_extras.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_extras"] = mod
    spec.loader.exec_module(mod)
    cls = mod.Egregora_DeepFilterNet_Denoise
    node = cls()
    x = signal()
    A = {"waveform": torch.from_numpy(x)[None], "sample_rate": 48000, "meta": {"k": 1}}
    g, arrs = {"cases": {}}, {}
    for name, kw in CASES.items():
        (out,) = node.execute(A, **kw)
        y = out["waveform"].numpy()
        arrs[name] = y[0][:, ::17].copy()
        g["cases"][name] = {"kwargs": kw, "shape": list(y.shape), "sr": out["sample_rate"], "peak": float(np.abs(y).max()),
                            "sum": float(y.astype(np.float64).sum()), "sumsq": float((y.astype(np.float64) ** 2).sum()),
                            "meta_keys": sorted(out["meta"].keys()), "dfn_meta": {k: v for k, v in out["meta"]["deepfilternet"].items() if k != "device"}}
    # quirk: adaptive_vad_source="none" makes _strength_per_frame return a 1-element array that is repeated to one hop
    # (480 samples) and then fails to broadcast against the signal -- the reference raises for every input longer than 10 ms
    try:
        node.execute(A, adaptive_vad_source="none")
        g["vad_none_raises"] = None
    except Exception as e:      # noqa: BLE001
        g["vad_none_raises"] = type(e).__name__
    # helper tables
    mono = x[0]
    probs = node._vad_probs_rms_48k(mono)
    arrs["vad_rms"] = probs
    arrs["vad_smooth60"] = node._smooth_probs(probs, 60)
    for mode in ("off", "more_on_noise", "more_on_speech", "gate_on_noise"):
        arrs["strength_" + mode] = node._strength_per_frame(0.65, arrs["vad_smooth60"], mode, 0.45, 0.9)
    gd, gw = node._gains_from_strength(arrs["strength_more_on_noise"], "equal_power")
    arrs["g_dry"], arrs["g_wet"] = gd, gw
    it = cls.INPUT_TYPES()
    g["surface"] = {"INPUT_TYPES": it, "widget_order": {k: list(v.keys()) for k, v in it.items()}, "RETURN_TYPES": list(cls.RETURN_TYPES),
                    "FUNCTION": cls.FUNCTION, "CATEGORY": cls.CATEGORY, "signature": str(inspect.signature(cls.execute)),
                    "display": mod.NODE_DISPLAY_NAME_MAPPINGS["Egregora_DeepFilterNet_Denoise"]}
    np.savez_compressed(OUT / "g11_dfn.npz", **arrs)
    (OUT / "g11_dfn.json").write_text(json.dumps(g, indent=1, sort_keys=True, ensure_ascii=False) + "\n", encoding="utf-8")
    print("wrote g11_dfn.json / g11_dfn.npz", {k: v["peak"] for k, v in g["cases"].items()})


if __name__ == "__main__":
    main()
