"""Channel-parallel Fat-Llama on the device: two ranks, each with ONE channel of a stereo file behind its own
plan -- egr_fatllama_enhance with EGR_FL_DEFER_FINALIZE, egr_fatllama_joint_peak, ONE all-reduce(MAX) of a float, egr_fatllama_finalize --
against the single-plan call on both channels, bit for bit.  The test box has one GPU: both ranks sit on device 0 and the collective
runs over gloo (RCCL refuses two ranks on one device); the arithmetic and the call sequence are the two-GPU path's.
Reference: the channels of a file meet only in upstream's joint normalise (egregora_fat_llama_gpu.py:213-224 toggle_normalize)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _signal(C, n):
    rng = np.random.Generator(np.random.PCG64(n + C))
    t = np.arange(n) / 48000.0
    x = np.stack([0.4 / (c + 1) * np.sin(2 * np.pi * 440.0 * (c + 1) * t) + 0.02 * rng.standard_normal(n) for c in range(C)])
    return x.astype(np.float32)


def _worker(rank, world, port, C, n, variant, autoscale, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from packload import load_pack
    load_pack()
    from egregora_amd import fatllama_engine as fe
    x = torch.from_numpy(_signal(C, n)).cuda()
    y = fe.enhance_channel_parallel(x, 1, 30, 0.02 if variant else 0.6, True, autoscale, True, True, variant=variant)
    mine = fe.enhance_channel_parallel(x, 1, 30, 0.02 if variant else 0.6, True, autoscale, True, True, variant=variant, gather=False)
    torch.cuda.synchronize()
    q.put((rank, y.cpu().numpy(), mine.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,C,n,variant,autoscale", [(2, 2, 48000, "", False), (2, 2, 48000, "relative,soft", True), (2, 2, 4801, "", True), (3, 2, 9600, "", False)])
def test_two_ranks_one_channel_each_equal_the_single_plan_call(world, C, n, variant, autoscale):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 250 + world * 13 + (n % 17)
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, n, variant, autoscale, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, str(ROOT))
    from packload import load_pack
    load_pack()
    from egregora_amd import fatllama_engine as fe
    x = torch.from_numpy(_signal(C, n)).cuda()
    want = fe.enhance_device(x, 1, 30, 0.02 if variant else 0.6, True, autoscale, True, True, variant=variant).cpu().numpy()
    assert np.isfinite(want).all() and float(np.max(np.abs(want))) > 0.5          # normalised to (about) full scale by the JOINT peak
    rows = 0
    for rank, y, mine in sorted(res, key=lambda r: r[0]):
        if n % 2 == 0:
            np.testing.assert_array_equal(y, want)          # per-channel states: the same kernels on the same data
            np.testing.assert_array_equal(mine, want[rows: rows + mine.shape[0]])
        else:
            # odd length: the single plan packs the two channels into ONE chirp-z state (re / im), a one-channel plan runs its channel with
            # an empty partner -- same arithmetic per element up to float32 round-off of the shared transform, not the same bits
            assert float(np.max(np.abs(y - want))) <= 2.0 / 32768.0 and float(np.mean(y != want)) < 0.05
        rows += mine.shape[0]
    assert rows == C


@pytest.mark.parametrize("n,devices", [(48000, "0,0"), (96000, "0,0,0")])
def test_devices_list_splits_the_channels_inside_one_process(pack, monkeypatch, n, devices):
    """EGREGORA_DEVICES (the in-process form a ComfyUI host uses): the Fat-Llama node gives every position of the list a channel block on a
    host thread of its own -- here two / three positions on the test box's one GPU -- and the result is the single-plan node result bit for
    bit (even lengths); the peaks meet on the host (4 bytes per block)."""
    from egregora_amd import fatllama_engine as fe
    cs = torch.from_numpy(_signal(2, n))
    monkeypatch.delenv("EGREGORA_DEVICES", raising=False)
    want, sr1 = fe.node_run(cs, 48000, 60, 0.6, 1536, True, True)
    monkeypatch.setenv("EGREGORA_DEVICES", devices)
    got, sr2 = fe.node_run(cs, 48000, 60, 0.6, 1536, True, True)
    torch.cuda.synchronize()
    assert sr1 == sr2 == 48000 and got.shape == want.shape and got.device == want.device
    assert torch.equal(got, want)
    x = torch.from_numpy(_signal(2, n)).cuda()
    a = fe.enhance_devices(x, 1, 30, 0.02, True, True, True, True, devs=[0, 0], variant="relative,soft")
    b = fe.enhance_device(x, 1, 30, 0.02, True, True, True, True, variant="relative,soft")
    assert torch.equal(a, b)


@pytest.mark.parametrize("variant,normalize,autoscale,node_post", [("", True, False, True), ("relative,soft", True, True, True), ("", False, True, False), ("", False, False, False)])
def test_deferred_finalize_equals_the_one_call_form_without_a_process_group(pack, variant, normalize, autoscale, node_post):
    """egr_fatllama_enhance with EGR_FL_DEFER_FINALIZE + egr_fatllama_joint_peak + egr_fatllama_finalize on ONE plan (no process group: what
    shard.sharded_channels does for world 1) is the one-call form bit for bit, for every combination of the finalising flags -- including none
    (finalize is then a no-op) and a NULL foreign peak."""
    import ctypes as C
    from egregora_amd import fatllama_engine as fe, native
    x = torch.from_numpy(_signal(2, 24000)).cuda()
    thr = 0.02 if variant else 0.6
    want = fe.enhance_device(x, 1, 20, thr, normalize, autoscale, True, node_post, variant=variant)
    got = fe.enhance_channel_parallel(x, 1, 20, thr, normalize, autoscale, True, node_post, variant=variant)
    assert torch.equal(got, want)
    # the same through the raw entry points with joint_dev = NULL
    L = native.lib()
    flags = ((native.FL_NORMALIZE if normalize else 0) | (native.FL_AUTOSCALE if autoscale else 0) | native.FL_PCM_IN |
             (native.FL_NODE_POST if node_post else 0) | fe.variant_flags(variant))
    plan = fe._plan(24000, 2, 1, 0)
    y = torch.empty_like(x)
    native.check(L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(x), native.ptr(y), 20, thr, flags | native.FL_DEFER_FINALIZE, native.stream_ptr()), "enhance")
    native.check(L.egr_fatllama_finalize(C.c_void_p(plan), native.ptr(y), flags, C.c_void_p(0), native.stream_ptr()), "finalize")
    torch.cuda.synchronize()
    assert torch.equal(y, want)
