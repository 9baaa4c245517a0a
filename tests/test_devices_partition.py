"""In-process multi-GPU sharding of the FlashSR node (flashsr_engine.infer_spans_devices; reference chunk loop
egregora_audio_super_resolution.py:407-420): the partition and the EGREGORA_DEVICES parsing on the CPU (no process group, no gloo),
and on the GPU box the whole path with two handles and two host threads on the one device there is (EGREGORA_DEVICES=0,0)."""
import numpy as np
import pytest
import torch


def test_balanced_bounds(pack):
    from egregora_amd import flashsr_engine as E
    assert [hi - lo for lo, hi in E.balanced_bounds(130, 8)] == [17, 17, 16, 16, 16, 16, 16, 16]
    assert E.balanced_bounds(3, 8)[:4] == [(0, 1), (1, 2), (2, 3), (3, 3)] and E.balanced_bounds(0, 2) == [(0, 0), (0, 0)]
    for n in (1, 2, 13, 130, 391):
        for k in (1, 2, 3, 5, 8):
            b = E.balanced_bounds(n, k)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(k - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_devices_env_parsing(pack, monkeypatch):
    from egregora_amd import flashsr_engine as E
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 2)
    monkeypatch.delenv("EGREGORA_DEVICES", raising=False)
    assert E.devices() == [2]
    monkeypatch.setenv("EGREGORA_DEVICES", "all")
    assert E.devices() == [0, 1, 2, 3]
    monkeypatch.setenv("EGREGORA_DEVICES", " 1, 3 ,0")
    assert E.devices() == [1, 3, 0]
    monkeypatch.setenv("EGREGORA_DEVICES", "0,0")
    assert E.devices() == [0, 0]
    for bad in ("0,7", "x", "-1", ","):
        monkeypatch.setenv("EGREGORA_DEVICES", bad)
        with pytest.raises(RuntimeError, match="EGREGORA_DEVICES"):
            E.devices()


@pytest.mark.gpu
def test_node_shards_over_the_devices_of_one_process(pack, monkeypatch):
    """EGREGORA_DEVICES=0,0 on the one-GPU box: two handles, two host threads, contiguous chunk blocks, predictions written into
    the stitching device's tensor by (here: same-device) copies -- against the single-handle run of the same file.  Rows are
    functions of (weights, row, seed, id) only, so the two runs differ by fp32 round-off from tile choices at most; with ONE device
    listed the path is the plain one, bit for bit.  (More than one physical GPU: unmeasured in this repository's pool.)"""
    from egregora_amd import audio_glue as ag, device_ops as ops, flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    win, hop = cfg.chunk, cfg.chunk - 375
    total = 6 * hop + 1000
    x = (0.3 * torch.randn(2, total, generator=torch.Generator().manual_seed(4))).cuda()
    sp = ag.spans(total, win, hop)
    assert len(sp) == 7
    e0 = E.FlashSREngine(cfg, P)
    try:
        E.set_engine(e0)
        monkeypatch.delenv("EGREGORA_DEVICES", raising=False)
        single = E.infer_spans(x, len(sp), win, hop, False)
        monkeypatch.setenv("EGREGORA_DEVICES", "0")
        assert torch.equal(E.infer_spans(x, len(sp), win, hop, False), single)
        e1, e2 = E.FlashSREngine(cfg, P), E.FlashSREngine(cfg, P)
        E.set_engines([e1, e2], [0, 0])
        monkeypatch.setenv("EGREGORA_DEVICES", "0,0")
        calls = []
        real = E.infer_block
        monkeypatch.setattr(E, "infer_block", lambda *a, **k: (calls.append((a[1], a[2], a[6] if len(a) > 6 else k.get("eng"))), real(*a, **k))[1])
        multi = E.infer_spans(x, len(sp), win, hop, False)
        assert sorted(c[:2] for c in calls) == [(0, 4), (4, 7)] and {id(c[2]) for c in calls} == {id(e1), id(e2)}
        rel = float((multi - single).double().norm() / single.double().norm())
        assert rel < 2e-5, rel
        # each block equals the same block computed alone by a handle of its own (same row count => same bits)
        monkeypatch.setattr(E, "infer_block", real)
        assert torch.equal(multi[:4], real(x, 0, 4, win, hop, False, e0)) and torch.equal(multi[4:], real(x, 4, 7, win, hop, False, e0))
        got = ops.wola_stitch(multi, total, win, hop)
        assert bool(torch.isfinite(got).all())
        # a worker's failure surfaces in the caller
        monkeypatch.setattr(E, "infer_block", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("boom")))
        with pytest.raises(RuntimeError, match="boom"):
            E.infer_spans(x, len(sp), win, hop, False)
    finally:
        E.set_engine(None)
