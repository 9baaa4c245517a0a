"""The FlashSR inner boundary of SURVEY.md section 8(b) through ctypes: egr_flashsr_create / egr_flashsr_infer / egr_flashsr_forward /
egr_flashsr_create_from_file / egr_flashsr_destroy (include/egregora_amd.h, csrc/egr_flashsr.cpp) stand in for the reference's
`FlashSR(s, v, vae)` + `model(x, lowpass_input=...)` (egregora_audio_super_resolution.py:346-369).

The library's graph walk must equal the operator-by-operator Python driver (`PyDriverEngine.forward_rows` of tools/flashsr_pydriver.py, itself pinned against
oracle/flashsr_torch.py in tests/test_gpu_flashsr.py) BIT FOR BIT: same kernels, same order, same operands (both pack weights with
csrc/egr_flashsr_pack.hip) -- at the toy size with every stage tapped, at full size, with the input low-pass, in the strict
f32-MFMA / no-Winograd configuration, across pass boundaries, and from a weight-blob file."""
import ctypes as C

import numpy as np
import pytest
import torch


def PyDriverEngine(*a, **k):
    """The operator-by-operator Python walk of the graph (tools/flashsr_pydriver.py: test tooling on top of the product engine)."""
    from tools.flashsr_pydriver import PyDriverEngine as cls
    return cls(*a, **k)

pytestmark = pytest.mark.gpu
STAGES = ("mel", "z_cond", "v", "z0", "mel_hat", "y")


@pytest.fixture(scope="module")
def tiny(pack):
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    # bit-for-bit comparisons between egr_flashsr_infer and the operator walk need both on the same kernels: the handle of this
    # fixture keeps every call on the three-term bf16 kernels (the fp16 scheme of egr_flashsr_infer: tests/test_gpu_split_h2.py)
    old = E.FlashSREngine.SPLIT
    E.FlashSREngine.SPLIT = "bf16x3"
    try:
        e = PyDriverEngine(cfg, P)
    finally:
        E.FlashSREngine.SPLIT = old
    yield e, cfg, P
    e.close()


def rows(cfg, n, seed):
    return (0.3 * torch.randn(n, cfg.chunk, generator=torch.Generator().manual_seed(seed))).cuda()


@pytest.mark.parametrize("lowpass", [False, True])
def test_library_graph_walk_equals_python_driver_bit_for_bit(pack, tiny, lowpass):
    e, cfg, _ = tiny
    x = rows(cfg, 3, 1)
    nz = e.noise(3, torch.tensor([5, 0, 9], dtype=torch.int64, device="cuda"), 11)
    sa, sb = {}, {}
    ya = e.forward_rows(x, nz, stages=sa, lowpass=lowpass)
    yb = e.c_forward(x, nz, stages=sb, lowpass=lowpass)
    torch.cuda.synchronize()
    for k in STAGES:
        assert torch.equal(sa[k].reshape(-1), sb[k].reshape(-1)), k
    assert torch.equal(ya, yb) and bool(torch.isfinite(yb).all())
    assert torch.equal(e.c_forward(x, nz, lowpass=lowpass), yb)           # arena reuse on the second call: same bits


def test_infer_keys_noise_by_row_id_and_is_independent_of_pass_boundaries(pack, tiny, monkeypatch):
    from egregora_amd import flashsr_engine as E
    e, cfg, _ = tiny
    x = rows(cfg, 7, 2)
    ids = torch.tensor([3, 1, 4, 1, 5, 9, 2], dtype=torch.int64, device="cuda")
    monkeypatch.setattr(E, "ROWS_PER_PASS", 32)
    one = E.infer_rows(e, x, ids, 42)
    monkeypatch.setattr(E, "ROWS_PER_PASS", 3)                              # passes of 3 + 3 + 1 rows
    split = E.infer_rows(e, x, ids, 42)
    # the same passes driven operator by operator from Python (tools/flashsr_pydriver.py)
    py = torch.cat([e.forward_rows(x[lo:lo + 3], e.noise(x[lo:lo + 3].shape[0], ids[lo:lo + 3].contiguous(), 42)) for lo in range(0, 7, 3)], 0)
    torch.cuda.synchronize()
    assert torch.equal(split, py)
    # rows are independent; tile choices follow the row count of a pass, so 7-row and 3-row passes agree to fp32 round-off only
    assert float((one - split).abs().max()) <= 2e-4 * float(one.abs().max())
    assert torch.equal(one[1], one[3]) is False                             # same id 1, different input rows
    same_in = torch.cat([x[:1], x[:1]])
    y2 = e.c_infer(same_in, torch.tensor([8, 8], dtype=torch.int64, device="cuda"), 42)
    assert torch.equal(y2[0], y2[1])                                         # same input + same id -> same noise -> same output
    y3 = e.c_infer(same_in, torch.tensor([8, 9], dtype=torch.int64, device="cuda"), 42)
    assert not torch.equal(y3[0], y3[1])
    # implicit ids 0..R-1 continue across passes
    assert torch.equal(e.c_infer(x, None, 7), e.c_infer(x, torch.arange(7, dtype=torch.int64, device="cuda"), 7))


def test_raw_ctypes_call_sequence_like_the_reference_runner(pack, tiny):
    """create -> infer -> destroy with nothing but ctypes and device pointers (what INTEGRATION.md section 2b shows in C)."""
    from egregora_amd import native
    e, cfg, _ = tiny
    L = native.lib()
    named = e.named_tensors()
    descs = (native.TensorDescC * len(named))()
    for d, (k, v) in zip(descs, named.items()):
        d.name, d.data, d.ndim = k.encode(), v.data_ptr(), v.dim()
        for i, n in enumerate(v.shape):
            d.shape[i] = n
    cc = native.flashsr_config_c(cfg)
    h = C.c_void_p()
    # (the module's engine runs the three-term bf16 kernels: the same flag here gives its bits; the default handle -- two fp16 terms
    # per operand -- sits fp32 round-off away)
    x = rows(cfg, 2, 3)
    want = e.c_infer(x, None, 123)
    for flags in (native.FSR_SPLIT_BF16X3, 0):
        assert L.egr_flashsr_create(C.byref(h), C.byref(cc), descs, len(named), flags, None) == 0, native.last_error()
        y = torch.empty_like(x)
        torch.cuda.synchronize()
        assert L.egr_flashsr_infer(h, C.c_void_p(x.data_ptr()), 2, 0, 123, None, C.c_void_p(y.data_ptr()), None) == 0, native.last_error()
        torch.cuda.synchronize()
        if flags:
            assert torch.equal(y, want)
        else:
            assert 0.0 < float((y - want).double().norm() / want.double().norm()) < 2e-5
        assert L.egr_flashsr_scratch_bytes(h) > 0
        assert L.egr_flashsr_destroy(h) == 0
    bad = native.flashsr_config_c(cfg)
    bad.struct_bytes = 12
    assert L.egr_flashsr_create(C.byref(h), C.byref(bad), descs, len(named), 0, None) != 0 and "ABI" in native.last_error()
    assert L.egr_flashsr_create(C.byref(h), C.byref(cc), descs, 5, 0, None) != 0          # tensors missing: loud, no partial handle


def test_weight_blob_file_round_trip(pack, tiny, tmp_path):
    from egregora_amd import flashsr_weights as W, native
    e, cfg, _ = tiny
    path = tmp_path / "flashsr_tiny.egrw"
    W.write_blob(str(path), cfg, e.named_tensors())
    L = native.lib()
    h = C.c_void_p()
    assert L.egr_flashsr_create_from_file(C.byref(h), str(path).encode(), 0, None) == 0, native.last_error()
    x = rows(cfg, 2, 4)
    nz = e.noise(2, None, 1)
    y = torch.empty_like(x)
    assert L.egr_flashsr_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(nz.data_ptr()), 2, 0, C.c_void_p(y.data_ptr()), None, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, e.c_forward(x, nz))
    L.egr_flashsr_destroy(h)
    (tmp_path / "junk.egrw").write_bytes(b"not a blob at all, just bytes")
    assert L.egr_flashsr_create_from_file(C.byref(h), str(tmp_path / "junk.egrw").encode(), 0, None) != 0
    assert "EGRW0001" in native.last_error()


def test_strict_configuration_flags(pack):
    """f32 MFMA everywhere, no Winograd, no thin-end kernels: the handle built with the matching flags equals the Python driver."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 1)
    old = (E.FlashSREngine.MFMA_MODE, E.FlashSREngine.WINO_MIN_CH, E.FlashSREngine.THIN_ENDS)
    try:
        E.FlashSREngine.MFMA_MODE, E.FlashSREngine.WINO_MIN_CH, E.FlashSREngine.THIN_ENDS = "f32", 1 << 30, False
        e = PyDriverEngine(cfg, P)
    finally:
        E.FlashSREngine.MFMA_MODE, E.FlashSREngine.WINO_MIN_CH, E.FlashSREngine.THIN_ENDS = old
    x = rows(cfg, 2, 6)
    nz = e.noise(2, None, 3)
    assert torch.equal(e.forward_rows(x, nz), e.c_forward(x, nz))
    e.close()


def test_profile_and_flop_count_agree_with_the_python_driver(pack, tiny):
    from egregora_amd import native
    e, cfg, _ = tiny
    x = rows(cfg, 2, 7)
    nz = e.noise(2, None, 0)
    e.prof = []
    e.forward_rows(x, nz)
    py = e.prof_summary()
    e.prof = None
    c = e.c_profile(lambda: e.c_forward(x, nz))
    assert set(c) == set(py)
    for k in py:
        assert c[k][0] == py[k][0] and c[k][1] == pytest.approx(py[k][1]) and c[k][2] > 0
    fl = C.c_double()
    native.check(native.lib().egr_flashsr_flop_count(C.c_void_p(e.handle), 2, C.byref(fl), None), "egr_flashsr_flop_count")
    assert fl.value == e.flop_count(2) == pytest.approx(e.flop_count_py(2))


def test_full_size_handle_equals_python_driver_bit_for_bit(pack):
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.FlashSRConfig()
    e = PyDriverEngine(cfg, A.init_params(cfg, 0))
    x = 0.2 * torch.randn(2, cfg.chunk, generator=torch.Generator().manual_seed(5))
    nz = e.noise(2, None, 0)
    sa, sb = {}, {}
    ya = e.forward_rows(x.cuda(), nz, stages=sa)
    yb = e.c_forward(x.cuda(), nz, stages=sb)
    torch.cuda.synchronize()
    for k in STAGES:
        assert torch.equal(sa[k].reshape(-1), sb[k].reshape(-1)), k
    assert torch.equal(ya, yb)
    e.close()


def test_forwards_from_different_streams_never_overlap_on_the_device(pack):
    """ONE handle called from two streams: a handle owns one scratch arena per context and serves one caller stream at a time, so the
    library chains the calls with an event (ForwardGuard, csrc/egr_flashsr.cpp) -- issued back to back on two streams that the
    runtime maps to different hardware queues, both results must equal the single-stream ones bit for bit, 10 rounds."""
    from egregora_amd import flashsr_arch as A, flashsr_engine as E, streams
    cfg = A.FlashSRConfig()
    e = PyDriverEngine(cfg, A.init_params(cfg, 0))
    x = 0.2 * torch.randn(8, cfg.chunk, generator=torch.Generator().manual_seed(9)).cuda()
    nz = e.noise(8, None, 0)
    ref = [e.c_forward(x[:4], nz[:4]).clone(), e.c_forward(x[4:], nz[4:]).clone()]
    torch.cuda.synchronize()
    side = streams.side_streams(1)
    if not side:
        pytest.skip("the runtime gave no stream that overlaps with the current one")
    cur = torch.cuda.current_stream()
    for _ in range(10):
        ready = cur.record_event()
        side[0].wait_event(ready)
        with torch.cuda.stream(side[0]):
            a = e.c_forward(x[:4], nz[:4])
        b = e.c_forward(x[4:], nz[4:])
        torch.cuda.synchronize()
        assert torch.equal(a, ref[0]) and torch.equal(b, ref[1])
    e.close()


def test_two_handles_run_concurrently_and_stay_bit_exact(pack):
    """The co-residency erratum of DESIGN.md 4.4a, closed: packed-fp32 VALU instructions with component-swapped operands return wrong
    values in lanes 48-63 while a bf16-MFMA wave of another kernel shares the SIMD (tools/ubench/coresidency.hip pins the
    instruction class); k_stft_frames next to another stream's k_conv_s3 was the first casualty.  The VALU kernels are now built
    without SLP-formed packed arithmetic, so two full-size forwards of two handles that REALLY overlap on the GPU (guard off, in a
    subprocess because the switch is read once) give the single-stream bits -- 12 rounds."""
    import os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, EGR_FSR_NO_STREAM_GUARD="1")
    r = subprocess.run([sys.executable, str(root / "tools" / "probe_two_stream_forwards.py"), "12"], cwd=str(root), env=env,
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-400:])
    assert r.returncode == 0 and "bad rounds: 0 of 12" in r.stdout, r.stdout[-800:] + r.stderr[-800:]


def test_concurrent_row_groups_inside_infer(pack):
    """egr_flashsr_infer splits a pass of >= 12 rows into row groups that run as simultaneous forwards on the handle's verified side
    streams (own arena each).  Rows are independent and the noise is keyed by row id, so the grouped result equals the one-group
    result up to the fp32 round-off of tile choices that follow a forward's row count -- and equals, bit for bit, separate infer
    calls on the same sub-batches."""
    import ctypes as C
    from egregora_amd import flashsr_arch as A, flashsr_engine as E, native
    cfg = A.tiny_config()
    e = PyDriverEngine(cfg, A.init_params(cfg, 2))
    L = native.lib()
    x = rows(cfg, 14, 12)
    ids = torch.arange(100, 114, dtype=torch.int64, device="cuda")
    h = C.c_void_p(e.handle)
    native.check(L.egr_flashsr_set_streams(h, 1, 6), "set_streams")
    one = e.c_infer(x, ids, 3)
    native.check(L.egr_flashsr_set_streams(h, 2, 6), "set_streams")
    two = e.c_infer(x, ids, 3)
    two_again = e.c_infer(x, ids, 3)
    native.check(L.egr_flashsr_set_streams(h, 1, 6), "set_streams")
    parts = torch.cat([e.c_infer(x[:7], ids[:7], 3), e.c_infer(x[7:], ids[7:], 3)])
    torch.cuda.synchronize()
    assert torch.equal(two, two_again) and torch.equal(two, parts)
    assert float((one - two).abs().max()) <= 2e-4 * float(one.abs().max())
    native.check(L.egr_flashsr_set_streams(h, 2, 6), "set_streams")
    assert torch.equal(e.c_infer(x, None, 3), e.c_infer(x, torch.arange(14, dtype=torch.int64, device="cuda"), 3))     # implicit ids
    e.close()


def test_fat_llama_next_to_a_flashsr_forward_on_another_stream(pack):
    """Product-level form of the erratum check: the Fat-Llama loop (complex arithmetic everywhere) on one stream while a full-size
    FlashSR forward (bf16-MFMA contraction kernels) runs on another -- what two ComfyUI workers, or one node graph with side
    streams, would do.  Before the VALU kernels were rebuilt without packed fp32 this returned a wrong result in EVERY sample
    (profiles/r02/cr19.log); now both results equal their solo runs bit for bit, 6 rounds."""
    from egregora_amd import fatllama_engine as fe, flashsr_arch as A, flashsr_engine as E, streams
    cfg = A.FlashSRConfig()
    e = PyDriverEngine(cfg, A.init_params(cfg, 0))
    x = 0.2 * torch.randn(9, cfg.chunk, generator=torch.Generator().manual_seed(4)).cuda()
    nz = e.noise(9, None, 0)
    a = (3000.0 * torch.randn(1, 480000, generator=torch.Generator().manual_seed(5))).round().cuda()
    ref_sr = e.c_forward(x, nz).clone()
    ref_fl = fe.enhance_device(a, 1, 40, 0.6, False, False, False, False, profile=True).clone()
    torch.cuda.synchronize()
    side = streams.side_streams(1)
    if not side:
        pytest.skip("the runtime gave no stream that overlaps with the current one")
    cur = torch.cuda.current_stream()
    for _ in range(6):
        ready = cur.record_event()
        side[0].wait_event(ready)
        with torch.cuda.stream(side[0]):
            y_sr = e.c_forward(x, nz)
        y_fl = [fe.enhance_device(a, 1, 40, 0.6, False, False, False, False, profile=True) for _ in range(6)]
        torch.cuda.synchronize()
        assert torch.equal(y_sr, ref_sr)
        assert all(torch.equal(y, ref_fl) for y in y_fl)
    e.close()
