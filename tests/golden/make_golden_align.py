#!/usr/bin/env python3
"""Golden fixture G13: the reference's `Audio Align (XCorr)` node (egregora_null_test_suite.py:272-340) -- GCC-PHAT delay,
integer + fractional (windowed-sinc FIR) compensation, pad / crop -- on seeded stereo signals.  Data only; the debug IMAGE
output (a matplotlib plot) is not captured.

  python tests/golden/make_golden_align.py      # writes tests/golden/g13_align.json, g13_align.npz
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


def cases():
    """name -> (ref [2,N] @sr, proc [2,N'] @sr', kwargs): proc = ref delayed by d samples (+ gain / noise / other rate)."""
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


def main():
    spec = importlib.util.spec_from_file_location("ref_null", REF / "egregora_null_test_suite.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_null"] = mod
    spec.loader.exec_module(mod)
    cls = mod.Audio_Align_XCorr
    node = cls()
    g, arrs = {"cases": {}}, {}
    for name, (a, sr, b, sr2, kw) in cases().items():
        out, d_s, d_ms, pk, img = node.execute({"waveform": torch.from_numpy(a)[None], "sample_rate": sr},
                                               {"waveform": torch.from_numpy(b)[None], "sample_rate": sr2, "meta": {"m": 2}}, **kw)
        arrs[name] = out["samples"][:, ::23].copy()
        g["cases"][name] = {"kwargs": kw, "delay_samples": d_s, "delay_ms": d_ms, "peak_corr": pk, "shape": list(out["waveform"].shape),
                            "sr": out["sample_rate"], "keys": sorted(out.keys()), "meta": out["meta"], "image_ndim": img.dim(),
                            "image_last": int(img.shape[-1]), "sum": float(out["samples"].astype(np.float64).sum())}
    it = cls.INPUT_TYPES()
    g["surface"] = {"INPUT_TYPES": it, "widget_order": {k: list(v.keys()) for k, v in it.items()}, "RETURN_TYPES": list(cls.RETURN_TYPES),
                    "RETURN_NAMES": list(cls.RETURN_NAMES), "FUNCTION": cls.FUNCTION, "CATEGORY": cls.CATEGORY,
                    "signature": str(inspect.signature(cls.execute)), "display": mod.NODE_DISPLAY_NAME_MAPPINGS["Audio Align (XCorr)"]}
    np.savez_compressed(OUT / "g13_align.npz", **arrs)
    (OUT / "g13_align.json").write_text(json.dumps(g, indent=1, sort_keys=True) + "\n", encoding="utf-8")
    print({k: (v["delay_samples"], v["shape"]) for k, v in g["cases"].items()})


if __name__ == "__main__":
    main()
