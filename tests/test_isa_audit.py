"""Build audit: the shipped library holds no VOP3P instruction whose op_sel routes a HIGH source dword into the LOW result lane.

gfx950 returns wrong values in lanes 48-63 for such instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 and even
v_pk_mov_b32 with `op_sel:[..1..]`) while a wave of another kernel issuing v_mfma_f32_32x32x16_bf16 shares the SIMD -- DESIGN.md
4.4a, bisected in tools/ubench/coresidency.hip (micro victims 2, 7, 8).  It is a cross-wave interaction, so it cannot be seen by
any single-stream numerics test: the guard is that the instruction form does not occur in the device code at all.  The compiler
forms it from complex arithmetic under the SLP vectoriser and from (re, im) swaps of register pairs, hence -fno-slp-vectorize on
the VALU translation units (csrc/Makefile).  `op_sel_hi:[..0..]` alone (a 32-bit constant / SGPR broadcast to both halves) is the
harmless direction and is allowed.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "comfyui-egregora-audio-super-resolution_amd", "libegregora_amd.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _device_disassembly(tmp_path):
    lib = tmp_path / "lib.so"
    shutil.copy(LIB, lib)
    subprocess.run([OBJDUMP, "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(p for p in tmp_path.iterdir() if "amdgcn" in p.name and p.stat().st_size > 0)
    assert objs, "no gfx950 code objects found in the library"
    text = []
    for o in objs:
        assert o.name.endswith("gfx950"), o.name        # one architecture, no multi-target bundle
        text.append(subprocess.run([OBJDUMP, "-d", str(o)], check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library or llvm-objdump missing")
def test_no_swizzled_packed_instructions_in_the_device_code(tmp_path):
    dis = _device_disassembly(tmp_path)
    packed = [ln for ln in dis.splitlines() if re.search(r"\bv_pk_\w+", ln)]
    assert len(packed) > 100                             # the contraction epilogues do use (unswizzled) packed adds
    bad = [ln.strip() for ln in packed if re.search(r"\bop_sel:\[", ln)]
    assert not bad, "swizzled packed instructions (gfx950 co-residency erratum):\n" + "\n".join(bad[:20])
    # and the matrix instruction the hot contractions are built on is really there
    assert "v_mfma_f32_32x32x16_bf16" in dis
