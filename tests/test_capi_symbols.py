"""The C-ABI library loads on a machine without a GPU and exports every symbol include/egregora_amd.h declares;
host-only planning entry points work (no compute call is made here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    txt = (ROOT / "include" / "egregora_amd.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(egr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pack):
    from egregora_amd import native
    lib = ctypes.CDLL(str(native.LIB_PATH))
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/egregora_amd.h but not exported"
        assert s in native.SIGNATURES, f"{s} has no ctypes signature in native.py"
    assert sorted(native.SIGNATURES) == syms


def test_abi_version_and_host_only_planning(pack):
    from egregora_amd import fatllama_engine as fe, native
    assert native.lib().egr_abi_version() == native.ABI_VERSION == 5
    i = fe.plan_info(2880000, 1)
    assert i["supported"] and i["M1"] * i["M2"] == 1440000 and i["N"] == 2880000
    prod = 1
    for r in i["radix1"]:
        prod *= r
    assert prod == i["M1"]
    assert fe.plan_info(160000, 6)["M"] == 480000
    odd = fe.plan_info(101, 1)                 # no packed-real plan: paired chirp-z, channel pairs, P >= 2D - 1 complex points
    assert odd["supported"] and odd["bluestein"] and odd["chirpz_kind"] == 2 and odd["D"] == 101 and odd["M"] >= 201
    even = fe.plan_info(2880002, 1)            # 60 s + 2 samples: even/odd packing, D = N / 2, both passes on scheduled lengths
    assert even["chirpz_kind"] == 1 and even["D"] == 1440001 and (even["M1"], even["M2"]) == (720, 4096)
    assert fe.plan_info(2880001, 1)["chirpz_kind"] == 2 and fe.plan_info(2880001, 1)["M2"] == 8192
    assert not fe.plan_info(2880000, 1)["bluestein"]
    bad = fe.plan_info(1, 1)
    assert not bad["supported"] and "unsupported" in bad["error"]
    assert fe.plan_info(2880000, 1, 1000)["M1"] == 1000
    assert i["lds_col"] <= 160 * 1024 and i["lds_row"] <= 160 * 1024 and i["levels"] == 2
    big = fe.plan_info(172800000, 1)            # BASELINE C5: 30 min at 96 kHz per channel
    assert big["supported"] and big["levels"] == 3 and big["M1"] * big["M2"] * big["M3"] == 86400000
    assert big["M1"] <= 2048 and big["M2"] <= 1024 and big["M3"] <= 4096
    assert fe.plan_info(16777216, 1)["levels"] == 2         # two levels reach 2 * 2048 * 4096 samples (outer columns up to 2048)
    long48 = fe.plan_info(9600000, 1)           # 200 s at 48 kHz: three levels around the two-barrier kernels (625 columns, rows of 3840)
    assert (long48["M1"], long48["M2"], long48["M3"]) == (625, 2, 3840)
    assert fe.plan_info(28800000, 1)["levels"] == 3
    #  rows of 4608 are a candidate of the three-level family too (338.7 s at 48 kHz = 441 x 4 x 4608), and a third
    # level -- two more passes per iteration -- never displaces a two-level plan the planner prices as fine (23.4 s: 500 x 1125)
    l46 = fe.plan_info(16257024, 1)
    assert (l46["M1"], l46["M2"], l46["M3"]) == (441, 4, 4608)
    short = fe.plan_info(1125000, 1)
    assert short["levels"] == 2 and short["M1"] <= 640 and short["M2"] <= 2560


def test_planner_matches_python_model_schedule(pack):
    """The C++ radix schedule equals the numpy kernel model's (tests/kernel_model.py)."""
    import kernel_model as km
    from egregora_amd import fatllama_engine as fe
    for n in (100, 2400, 48000, 245760, 960000, 2880000):
        i = fe.plan_info(n, 1)
        assert i["radix1"] == km.radix_schedule(i["M1"]) and i["radix2"] == km.radix_schedule(i["M2"])


def test_compute_nodes_fail_loudly_without_gpu(pack):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    node = pack.NODE_CLASS_MAPPINGS["EgregoraFatLlamaGPU"]()
    with pytest.raises(RuntimeError, match="No AMD GPU|no CPU fallback"):
        node.run("wav", 1, 0.6, 1411, True, True, AUDIO={"waveform": torch.zeros(1, 1, 64), "sample_rate": 48000})


def test_flag_values_in_the_header_equal_the_python_binding(pack):
    """The EGR_FL_* flag bits of egr_fatllama_enhance (include/egregora_amd.h) and the constants native.py / fatllama_engine.py pass."""
    from egregora_amd import fatllama_engine as fe, native
    txt = (ROOT / "include" / "egregora_amd.h").read_text()
    hdr = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define\s+(EGR_FL_[A-Z_]+)\s+0x([0-9a-fA-F]+)u", txt)}
    want = {"EGR_FL_NORMALIZE": native.FL_NORMALIZE, "EGR_FL_AUTOSCALE": native.FL_AUTOSCALE, "EGR_FL_PCM_IN": native.FL_PCM_IN,
            "EGR_FL_NODE_POST": native.FL_NODE_POST, "EGR_FL_THR_RELATIVE": native.FL_THR_RELATIVE, "EGR_FL_THR_SOFT": native.FL_THR_SOFT,
            "EGR_FL_NO_INIT_THR": native.FL_NO_INIT_THR, "EGR_FL_ZERO_STUFF": native.FL_ZERO_STUFF, "EGR_FL_INTERP_LINSPACE": native.FL_INTERP_LINSPACE,
            "EGR_FL_THR_RECOMPUTE": native.FL_THR_RECOMPUTE, "EGR_FL_DEFER_FINALIZE": native.FL_DEFER_FINALIZE}
    assert hdr == want, (hdr, want)
    assert len(set(want.values())) == len(want) and all(v & (v - 1) == 0 for v in want.values())          # distinct single bits
    assert fe.variant_flags("relative,recompute") == native.FL_THR_RELATIVE | native.FL_THR_RECOMPUTE
    assert fe.variant_flags("") == 0
    with pytest.raises(RuntimeError):
        fe.variant_flags("relative,no_such_variant")


def test_committed_counter_profile_belongs_to_these_sources():
    """profiles/traffic.json (the PMC passes bench.py's roofline objects quote) records the hash of the library sources it was taken on; the
    bench line says whether it matches the sources it runs on.  At the end of a round the two must agree: a kernel change after the last
    profile makes the `traffic` figures somebody else's."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_for_sha", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    assert tj.get("csrc_sha") == mod.csrc_sha(), (tj.get("csrc_sha"), mod.csrc_sha(), tj.get("source"))
