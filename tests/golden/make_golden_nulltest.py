#!/usr/bin/env python3
"""Golden fixture G14: the reference's `Audio Gain Match`, `Audio Null Test`, `Audio Plotter` and `Null Test (Full)` nodes
(egregora_null_test_suite.py:342-668) on seeded stereo signals: levels (K-weighted gated loudness / RMS), matched audio, null
signal, metric dictionaries, image shapes; plus the four node surfaces.  Data only.

  python tests/golden/make_golden_nulltest.py      # writes tests/golden/g14_nulltest.json, g14_nulltest.npz
"""
import importlib.util
import inspect
import json
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
KEYS = ("Audio Gain Match", "Audio Null Test", "Audio Plotter", "Null Test (Full)")


def signals():
    """name -> (ref [2,N], sr, proc [2,N'], sr'): proc = a filtered, scaled, noisy copy of ref."""
    rng = np.random.Generator(np.random.PCG64(14))
    out = {}
    for name, n, n2, sr, sr2, g in (("even", 24000, 24000, 48000, 48000, 0.5), ("odd", 23999, 23000, 44100, 44100, 1.7),
                                    ("rate", 24000, 16000, 48000, 32000, 0.25), ("short", 9000, 9000, 48000, 48000, 3.0)):
        t = np.arange(n) / sr
        env = 0.3 + 0.7 * (np.sin(2 * np.pi * 1.5 * t) > -0.3)                 # loud / quiet stretches so the loudness gate bites
        a = (0.2 * rng.standard_normal((2, n)) * env + 0.1 * np.sin(2 * np.pi * 440 * t) + 0.01).astype(np.float32)
        a[1] = 0.6 * a[0] + 0.4 * a[1]
        m = min(n, int(round(n2 * sr / sr2)))
        b = np.stack([np.convolve(a[c, :m], [0.1, 0.8, 0.1], mode="same") for c in range(2)]) * g
        b = b + 0.003 * rng.standard_normal(b.shape)
        if sr2 != sr:
            b = np.stack([np.interp(np.linspace(0.0, 1.0, n2, endpoint=False), np.linspace(0.0, 1.0, m, endpoint=False), b[c])
                          for c in range(2)])
        out[name] = (a, sr, b.astype(np.float32)[:, :n2], sr2)
    return out


GAIN_CASES = (("even", {}), ("even", dict(mode="RMS")), ("odd", dict(mode="LUFS-I", max_gain_db=3.0)), ("rate", dict(mode="LUFS-I")),
              ("short", dict(mode="RMS", max_gain_db=6.0)), ("short", dict(mode="LUFS-I", max_gain_db=48.0)))
NULL_CASES = (("even", {}), ("even", dict(least_squares_scale=True, compute_hf_residual=True, hf_band_hz=6000)),
              ("odd", dict(invert_b=False, compute_hf_residual=True, n_fft=1024, hop=256)),
              ("short", dict(least_squares_scale=True, compute_corr=False, compute_null_lufs=False, compute_hf_residual=True, hf_band_hz=1000)))
FULL_CASES = (("even", dict(draw_waveforms=False, draw_spectrograms=False, draw_diffspec=False)),
              ("rate", dict(match_mode="RMS", least_squares_scale=True, compute_hf_residual=True, fir_len=32, draw_waveforms=False,
                            draw_spectrograms=False, draw_diffspec=False)))


def aud(x, sr, meta=None):
    d = {"waveform": torch.from_numpy(np.ascontiguousarray(x))[None], "sample_rate": sr}
    if meta is not None:
        d["meta"] = meta
    return d


def main():
    spec = importlib.util.spec_from_file_location("ref_null", REF / "egregora_null_test_suite.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_null"] = mod
    spec.loader.exec_module(mod)
    sig = signals()
    g, arrs = {"gain": [], "null": [], "full": [], "surface": {}}, {}
    for i, (s, kw) in enumerate(GAIN_CASES):
        a, sr, b, sr2 = sig[s]
        out, gdb, rl, il = mod.Audio_Gain_Match().execute(aud(a, sr), aud(b, sr2, {"tag": i}), **kw)
        arrs[f"gain{i}"] = out["samples"][:, ::29].copy()
        g["gain"].append({"signal": s, "kwargs": kw, "gain_db": gdb, "ref_level": rl, "in_level": il, "shape": list(out["waveform"].shape),
                          "sr": out["sample_rate"], "meta": out["meta"], "keys": sorted(out.keys())})
    for i, (s, kw) in enumerate(NULL_CASES):
        a, sr, b, sr2 = sig[s]
        out, met = mod.Audio_Null_Test().execute(aud(a, sr), aud(b, sr), **kw)
        arrs[f"null{i}"] = out["samples"][:, ::29].copy()
        g["null"].append({"signal": s, "kwargs": kw, "metrics": met, "metric_order": list(met.keys()), "shape": list(out["waveform"].shape),
                          "sr": out["sample_rate"], "meta": out["meta"]})
    for i, (s, kw) in enumerate(FULL_CASES):
        a, sr, b, sr2 = sig[s]
        r = mod.Null_Test_Full().execute(aud(a, sr), aud(b, sr2, {"src": s}), **kw)
        arrs[f"full{i}_matched"], arrs[f"full{i}_null"] = r[0]["samples"][:, ::29].copy(), r[1]["samples"][:, ::29].copy()
        g["full"].append({"signal": s, "kwargs": kw, "delay_ms": r[2], "gain_db": r[3], "metrics": r[4], "shapes": [list(r[0]["waveform"].shape),
                          list(r[1]["waveform"].shape)], "meta": [r[0]["meta"], r[1]["meta"]], "images": [list(im.shape) for im in r[5:]]})
    a, sr, b, sr2 = sig["short"]
    nul, _ = mod.Audio_Null_Test().execute(aud(a, sr), aud(b, sr))
    g["plotter"] = {"all": [list(im.shape) for im in mod.Audio_Plotter().execute(aud(a, sr), aud(b, sr), nul)],
                    "none": [list(im.shape) for im in mod.Audio_Plotter().execute(aud(a, sr), aud(b, sr), nul, False, False, False)]}
    try:
        mod.Audio_Null_Test().execute(aud(a, sr), aud(b, 44100))
    except ValueError as e:
        g["rate_mismatch_error"] = str(e)
    for key in KEYS:
        cls = mod.NODE_CLASS_MAPPINGS[key]
        it = cls.INPUT_TYPES()
        g["surface"][key] = {"INPUT_TYPES": it, "widget_order": {k: list(v.keys()) for k, v in it.items()}, "RETURN_TYPES": list(cls.RETURN_TYPES),
                             "RETURN_NAMES": list(cls.RETURN_NAMES), "FUNCTION": cls.FUNCTION, "CATEGORY": cls.CATEGORY,
                             "signature": str(inspect.signature(cls.execute)), "display": mod.NODE_DISPLAY_NAME_MAPPINGS[key]}
    np.savez_compressed(OUT / "g14_nulltest.npz", **arrs)
    (OUT / "g14_nulltest.json").write_text(json.dumps(g, indent=1, sort_keys=True) + "\n", encoding="utf-8")
    for k in ("gain", "null", "full"):
        for c in g[k]:
            print(k, c["signal"], {x: c[x] for x in ("gain_db", "ref_level", "in_level", "delay_ms", "metrics") if x in c})
    print(g["plotter"])


if __name__ == "__main__":
    main()
