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


def test_ensure_ready_from_four_threads_loads_the_checkpoints_once(pack, monkeypatch):
    """VERDICT r4 / ADVICE r4: the registry's read-modify-write runs under one lock -- four threads asking for four devices at once
    (the 8-GPU node's first call) read the checkpoints ONCE, build one engine per device, and nobody trips over `_SOURCE = None`."""
    import threading
    import time
    from egregora_amd import flashsr_engine as E, native
    loads, builds = [], []

    def slow_load():
        loads.append(threading.get_ident())
        time.sleep(0.2)
        return "cfg", {"w": torch.zeros(1)}

    def fake_build(cfg, params, dev):
        assert cfg == "cfg" and params is not None
        time.sleep(0.05)
        builds.append(dev)
        return ("engine", dev, len(builds))

    monkeypatch.setattr(E, "_load_source", slow_load)
    monkeypatch.setattr(E, "_build_engine", fake_build)
    monkeypatch.setattr(native, "require_device", lambda: "gfx950")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setenv("EGREGORA_DEVICES", "0,1,2,3")
    E.set_engine(None)
    try:
        got, errs = {}, []

        def ask(d):
            try:
                for _ in range(3):
                    got.setdefault(d, []).append(E.ensure_ready(d))
            except BaseException as ex:      # noqa: BLE001
                errs.append(ex)

        ts = [threading.Thread(target=ask, args=(d,)) for d in (0, 1, 2, 3)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        assert len(loads) == 1 and sorted(builds) == [0, 1, 2, 3]
        assert all(len({id(e) for e in got[d]}) == 1 and got[d][0][1] == d for d in got)
        assert E._SOURCE is None                             # every listed device holds its copy: the host state dict is gone
        # a repeated index is a handle of its own, keyed by position; the first occurrence is the device's engine
        monkeypatch.setenv("EGREGORA_DEVICES", "0,0,0")
        e = E.engines_for([0, 0, 0])
        assert e[0] is got[0][0] and len({id(x) for x in e}) == 3 and len(loads) == 2 and builds[4:] == [0, 0]
        assert E.engines_for([0, 0, 0]) == e and len(loads) == 2
    finally:
        E.set_engine(None)


@pytest.mark.gpu
def test_three_slots_on_one_device_build_their_own_handles(pack, monkeypatch):
    """EGREGORA_DEVICES=0,0,0 WITHOUT set_engines: the node's first call resolves three handles from one (slow) load of the weights on
    the caller's thread, then three host threads run their blocks; same result as the single-handle run."""
    import time
    from egregora_amd import audio_glue as ag, flashsr_arch as A, flashsr_engine as E
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    win, hop = cfg.chunk, cfg.chunk - 375
    total = 6 * hop + 1000
    x = (0.3 * torch.randn(2, total, generator=torch.Generator().manual_seed(4))).cuda()
    n = len(ag.spans(total, win, hop))
    loads = []

    def slow_load():
        loads.append(1)
        time.sleep(0.3)
        return cfg, P

    monkeypatch.setattr(E, "_load_source", slow_load)
    E.set_engine(None)
    try:
        monkeypatch.setenv("EGREGORA_DEVICES", "0,0,0")
        multi = E.infer_spans(x, n, win, hop, False)
        assert len(loads) == 1 and E._SOURCE is None
        engs = E.engines_for([0, 0, 0])
        assert len({e.handle for e in engs}) == 3 and len(loads) == 1
        monkeypatch.delenv("EGREGORA_DEVICES", raising=False)
        single = E.infer_spans(x, n, win, hop, False)        # device 0's engine = slot 0's
        rel = float((multi - single).double().norm() / single.double().norm())
        assert rel < 2e-5, rel
        assert torch.equal(E.infer_spans(x, n, win, hop, False), single)
    finally:
        E.set_engine(None)
