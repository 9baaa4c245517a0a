"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise shard.sharded_chunks exactly as the GPU
path uses it (contiguous chunk blocks, one all-gather, results independent of rank count)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, n, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from packload import load_pack
    load_pack()
    from egregora_amd import shard
    calls = []

    def run_block(lo, hi):
        calls.append((lo, hi))
        idx = torch.arange(lo, hi, dtype=torch.float32)
        return (idx[:, None, None] * 10.0 + torch.arange(2)[None, :, None] + torch.zeros(1, 1, 5)).contiguous()

    out = shard.sharded_chunks(run_block, n, (2, 5), torch.device("cpu"))
    q.put((rank, out.numpy(), calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 13), (2, 4), (3, 7), (2, 1)])
def test_sharded_chunks_equals_single_rank(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + world * 7 + n
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (np.arange(n, dtype=np.float32)[:, None, None] * 10.0 + np.arange(2)[None, :, None] + np.zeros((1, 1, 5), np.float32))
    covered = []
    for rank, out, calls in res:
        np.testing.assert_array_equal(out, want)
        covered += calls
    spans = sorted(c for c in covered)
    assert sum(hi - lo for lo, hi in spans) == n          # every chunk computed exactly once across ranks
    assert spans[0][0] == 0 and spans[-1][1] == n


def test_block_bounds():
    sys.path.insert(0, str(ROOT))
    from packload import load_pack
    load_pack()
    from egregora_amd import shard
    # balanced to within one chunk (the ceil-sized blocks of rounds 1-3 left the last of 8 ranks with 11 of 130 chunks: a 7.65x ceiling)
    assert shard.block_bounds(130, 8) == [(0, 17), (17, 34), (34, 50), (50, 66), (66, 82), (82, 98), (98, 114), (114, 130)]
    for n in (0, 1, 7, 13, 130, 391):
        for w in (1, 2, 3, 8):
            b = shard.block_bounds(n, w)
            sizes = [hi - lo for lo, hi in b]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1)) and max(sizes) - min(sizes) <= 1
    assert shard.block_bounds(3, 8)[:4] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert shard.block_bounds(0, 2) == [(0, 0), (0, 0)]


def _wola_worker(rank, world, port, total, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from packload import load_pack
    load_pack()
    from egregora_amd import shard
    from oracle import glue as og
    rng = np.random.Generator(np.random.PCG64(7))
    x = rng.standard_normal((2, total)).astype(np.float32)          # the same file on every rank (replicated input)
    spans = og.chunk_spans(total)

    def run_block(lo, hi):          # a stand-in model that depends on the GLOBAL chunk index, like the row-id-keyed noise
        out = np.zeros((hi - lo, 2, og.WIN), np.float32)
        for k in range(lo, hi):
            s, L = spans[k]
            out[k - lo, :, :L] = x[:, s:s + L] * np.float32(1.0 + 0.01 * k)
        return torch.from_numpy(out)

    preds = shard.sharded_chunks(run_block, len(spans), (2, og.WIN), torch.device("cpu")).numpy()
    y = og.wola([(preds[k], spans[k][0], spans[k][1]) for k in range(len(spans))], total)
    q.put((rank, y))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_then_wola_across_ranks_equals_the_single_rank_stitch(world):
    """Chunks meet only in WOLA (reference egregora_audio_super_resolution.py:407-420): per-rank chunk blocks, ONE all-gather, then
    the reference's stitch on every rank -- bit-identical to one rank doing all chunks, on every rank."""
    from oracle import glue as og
    total = og.WIN + 4 * og.HOP + 12345            # 6 chunks, short last one
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 500) + world
    procs = [ctx.Process(target=_wola_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.Generator(np.random.PCG64(7))
    x = rng.standard_normal((2, total)).astype(np.float32)
    spans = og.chunk_spans(total)
    single = []
    for k, (s, L) in enumerate(spans):
        c = np.zeros((2, og.WIN), np.float32)
        c[:, :L] = x[:, s:s + L] * np.float32(1.0 + 0.01 * k)
        single.append((c, s, L))
    want = og.wola(single, total)
    for rank, y in res:
        np.testing.assert_array_equal(y, want)


class _OracleChannelBackend:
    """shard.sharded_channels with the oracle's arithmetic as the per-rank work (CPU; the device backend is fatllama_engine's
    _ChannelBlockBackend, tests/test_gpu_bench_ranks.py): the loop per channel, the peak after autoscale, and the finalising
    arithmetic of oracle.fatllama.enhance_channels with the JOINT peak handed in."""

    def __init__(self, x, factor, iters, thr, autoscale):
        self.x, self.f, self.iters, self.thr, self.autoscale = x, factor, iters, thr, autoscale
        self.calls = []

    def empty(self, rows):
        return torch.zeros((rows, self.x.shape[1] * self.f), dtype=torch.float32)

    def zero_peak(self):
        return torch.zeros(1, dtype=torch.float32)

    def run_local(self, lo, hi):
        from oracle import fatllama as ofl
        self.calls.append((lo, hi))
        out = ofl.enhance_channels(self.x[lo:hi], self.f, self.iters, self.thr, normalize=False, autoscale=self.autoscale)
        return out

    def joint_peak(self, out):
        return torch.tensor([float(np.max(np.abs(out)))], dtype=torch.float32)

    def finalize(self, out, joint):
        m = float(joint.item())
        return torch.from_numpy((out / np.float32(m)).astype(np.float32) if m > 0 else out)


def _channel_worker(rank, world, port, C, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from packload import load_pack
    load_pack()
    from egregora_amd import shard
    rng = np.random.Generator(np.random.PCG64(9))
    x = np.rint(rng.standard_normal((C, 4800)) * 3000.0 * (1.0 + np.arange(C))[:, None]).astype(np.float32)
    be = _OracleChannelBackend(x, 1, 5, 40.0, autoscale=(C % 2 == 0))
    out = shard.sharded_channels(be, C)
    mine = shard.sharded_channels(be, C, gather=False)
    q.put((rank, out.numpy(), mine.numpy(), be.calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,C", [(2, 2), (2, 1), (3, 2), (2, 5)])
def test_channel_parallel_fatllama_equals_single_rank(world, C):
    """one channel block per rank, ONE all-reduce(MAX) of a float before the joint normalise; ranks beyond
    the channel count idle.  Every rank ends with the single-rank result bit for bit (max is exact)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 100 + world * 11 + C
    procs = [ctx.Process(target=_channel_worker, args=(r, world, port, C, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, str(ROOT))
    from oracle import fatllama as ofl
    rng = np.random.Generator(np.random.PCG64(9))
    x = np.rint(rng.standard_normal((C, 4800)) * 3000.0 * (1.0 + np.arange(C))[:, None]).astype(np.float32)
    want = ofl.enhance_channels(x, 1, 5, 40.0, normalize=True, autoscale=(C % 2 == 0))
    rows = 0
    for rank, out, mine, calls in sorted(res, key=lambda r: r[0]):
        np.testing.assert_array_equal(out, want)
        np.testing.assert_array_equal(mine, want[rows: rows + mine.shape[0]])
        rows += mine.shape[0]
        assert all(hi - lo >= 1 for lo, hi in calls)
    assert rows == C
