"""Adds the low parts (cl / sl) to every Trig<R> of csrc/egr_trig_tables.inc: c[m] + cl[m] = cos(2 pi m / R) to 2^-48, so that a
product with the constant can be formed as fma(x, c, x * cl) (one rounding, no constant-rounding bias).  The high parts stay the
literals the file has (float(literal)); the low part is float(literal - float(literal)).  Idempotent."""
import re, sys
from pathlib import Path
import numpy as np
p = Path(__file__).resolve().parents[1] / "comfyui-egregora-audio-super-resolution_amd" / "csrc" / "egr_trig_tables.inc"
src = p.read_text()
src = re.sub(r"\n    static constexpr float [cs]l\[\d+\] = \{[^}]*\};", "", src)
def lo_line(m):
    name, n, body = m.group(1), m.group(2), m.group(3)
    vals = [float(t.strip().rstrip("f")) for t in body.split(",")]
    lo = [float(np.float32(v - float(np.float32(v)))) for v in vals]
    return m.group(0) + f"\n    static constexpr float {name}l[{n}] = {{" + ", ".join(f"{v!r}f" if v != 0 else "0.0f" for v in lo) + "};"
out = re.sub(r"    static constexpr float ([cs])\[(\d+)\] = \{([^}]*)\};", lo_line, src)
p.write_text(out)
print("wrote", p)
