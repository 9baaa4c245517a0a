#!/usr/bin/env python3
"""Golden fixture G12: the reference's GCC-PHAT delay estimator `_xcorr_delay` (egregora_null_test_suite.py:213-237) on seeded
signals with integer and fractional delays.  Data only.

  python tests/golden/make_golden_xcorr.py      # writes tests/golden/g12_xcorr.json
"""
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def cases():
    """(name, a, b, max_shift): b = a delayed by d samples (fractional: windowed-sinc interpolation), plus noise."""
    rng = np.random.Generator(np.random.PCG64(12))
    out = []
    for name, n, d, ms in (("int+3", 6000, 3.0, 200), ("int-7", 6000, -7.0, 200), ("zero", 5000, 0.0, 64), ("frac+2.4", 9000, 2.4, 300),
                           ("frac-11.75", 20000, -11.75, 480), ("long+40", 150000, 40.0, 4800)):
        a = rng.standard_normal(n).astype(np.float32)
        k = np.arange(-32, 33)
        h = np.sinc(k - d + np.round(d)) * np.hanning(65)
        b = np.convolve(np.roll(a, int(np.round(d))), h, mode="same") + 0.01 * rng.standard_normal(n)
        out.append((name, a, b.astype(np.float32), ms))
    return out


def main():
    spec = importlib.util.spec_from_file_location("ref_null", REF / "egregora_null_test_suite.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_null"] = mod
    spec.loader.exec_module(mod)
    g = {name: {"delay": mod._xcorr_delay(a, b, 48000, ms), "max_shift": ms, "n": int(a.size)} for name, a, b, ms in cases()}
    (OUT / "g12_xcorr.json").write_text(json.dumps(g, indent=1, sort_keys=True) + "\n", encoding="utf-8")
    print(g)


if __name__ == "__main__":
    main()
