#!/usr/bin/env python3
"""Golden fixture G10: the reference's evaluation-pack nodes that sit on the
device kernels of this pack -- `Metrics (LSD + SI-SDR)` and `Resample Audio (HQ)` -- captured by importing
/root/reference/egregora_audio_eval_pack.py in the build container.  Data only (inputs are regenerated from
seeds by the tests; outputs and node surfaces are stored).

  python tests/golden/make_golden_eval.py       # writes tests/golden/g10_eval.json, g10_eval.npz

Reference call sites exercised: egregora_audio_eval_pack.py : to_internal_audio (:89), _normalize_CN (:60),
make_audio (:76), Metrics_LSD_SISDR (:432-470), Resample_Audio_HQ (:476-522, scipy polyphase branch),
NODE_CLASS_MAPPINGS / NODE_DISPLAY_NAME_MAPPINGS (:528-544).
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


def _load(name, fname):
    spec = importlib.util.spec_from_file_location(name, REF / fname)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def signals():
    """Seeded inputs shared with tests/test_eval_nodes.py (PCG64 seed 10)."""
    rng = np.random.Generator(np.random.PCG64(10))
    n = 60000
    t = np.arange(n) / 48000.0
    a = np.stack([0.3 * np.sin(2 * np.pi * 330 * t) + 0.1 * np.sin(2 * np.pi * 7000 * t) + 0.02 * rng.standard_normal(n),
                  0.25 * np.sin(2 * np.pi * 500 * t + 0.3) + 0.02 * rng.standard_normal(n)]).astype(np.float32)
    b = (a * np.float32(0.97) + 2e-3 * rng.standard_normal(a.shape)).astype(np.float32)[:, :59000]
    m = (0.4 * np.sin(2 * np.pi * 1000 * np.arange(44100) / 44100.0) + 0.05 * rng.standard_normal(44100)).astype(np.float32)
    return a, b, m


def main():
    ev = _load("ref_eval", "egregora_audio_eval_pack.py")
    a, b, m = signals()
    A = {"waveform": torch.from_numpy(a)[None], "sample_rate": 48000}
    B = {"waveform": torch.from_numpy(b)[None], "sample_rate": 48000}
    node = ev.Metrics_LSD_SISDR()
    g = {"metrics_default": node.execute(A, B)[0],
         "metrics_1024_256": node.execute(A, B, n_fft=1024, hop=256)[0],
         "metrics_lsd_only": node.execute(A, B, compute_si_sdr=False)[0],
         "metrics_self": node.execute(A, A)[0]}
    rs = ev.Resample_Audio_HQ()
    M = {"waveform": torch.from_numpy(m)[None, None], "sample_rate": 44100}
    (r48,) = rs.execute(M, 48000, "scipy_polyphase")
    (r2,) = rs.execute(A, 44100, "auto")
    (same,) = rs.execute(A, 48000)
    g["resample"] = {"m_to_48k_shape": list(r48["waveform"].shape), "m_to_48k_sr": r48["sample_rate"],
                     "a_to_441_shape": list(r2["waveform"].shape), "keys": sorted(r48.keys()),
                     "same_rate_returns_input_shape": list(same["waveform"].shape), "same_rate_keys": sorted(same.keys())}
    np.savez_compressed(OUT / "g10_eval.npz", m_to_48k=r48["samples"][:, ::37], a_to_441=r2["samples"][:, ::53])

    def surf(cls):
        it = cls.INPUT_TYPES()
        return {"INPUT_TYPES": it, "widget_order": {k: list(v.keys()) for k, v in it.items()}, "RETURN_TYPES": list(cls.RETURN_TYPES),
                "RETURN_NAMES": list(cls.RETURN_NAMES), "FUNCTION": cls.FUNCTION, "CATEGORY": cls.CATEGORY,
                "signature": str(inspect.signature(getattr(cls, cls.FUNCTION)))}
    g["surface"] = {k: surf(ev.NODE_CLASS_MAPPINGS[k]) for k in ("Metrics (LSD + SI-SDR)", "Resample Audio (HQ)")}
    g["display"] = {k: ev.NODE_DISPLAY_NAME_MAPPINGS[k] for k in g["surface"]}
    g["class_names"] = {k: ev.NODE_CLASS_MAPPINGS[k].__name__ for k in g["surface"]}
    # _normalize_CN shape rule (time axis = the longer one; >2-D folds everything but the longest axis)
    g["normalize_CN"] = {str(s): list(ev._normalize_CN(np.zeros(s, np.float32)).shape)
                         for s in [(5,), (2, 9), (9, 2), (3, 3), (1, 2, 7), (2, 7, 3), (1, 1, 4)]}
    (OUT / "g10_eval.json").write_text(json.dumps(g, indent=1, sort_keys=True, ensure_ascii=False) + "\n", encoding="utf-8")
    print("wrote g10_eval.json / g10_eval.npz")


if __name__ == "__main__":
    main()
