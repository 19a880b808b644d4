"""herald_amd.sharded.FramedStep on the HIP engine: the sharded step with fixed frames, replayed from hipGraphs at
world size 1 and run eagerly at world sizes 2 / 4 / 8 as W processes on one GPU (host-staged all-to-all, as in
test_gpu_sharded_multirank.py).  Checked against the oracle's serial PS semantics (oracle/cpu.py sparse_pull /
sparse_push in rank order: PSAgent.h:124-237, PSFHandle.h:101-164)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from herald_amd import synth
from oracle import cpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(rows, n, nb, kind, seed=0):
    out = []
    for k in range(nb):
        if kind == "criteo":
            ids = (synth.criteo_batch(n // 26, seed + k).reshape(-1) % rows).astype(np.float32)
        else:
            g = np.random.default_rng(seed * 1000 + k)
            ids = g.integers(0, rows, size=n).astype(np.float32)
            ids[n // 4: n // 3] = ids[0]
            ids[: n // 8] = np.arange(n // 8) % rows
        out.append(ids)
    return out


@pytest.mark.parametrize("rows,width,n,kind,graphs,block", [(50000, 128, 6656, "criteo", True, 4), (50000, 128, 6656, "criteo", False, 4),
                                                            (3000, 64, 900, "mixed", True, 2), (200, 32, 40, "mixed", True, 1),
                                                            (3000, 64, 900, "mixed", True, 16),
                                                            # beyond 36,864 ids: radix-sorted plans, the push's reduce by unique
                                                            # key (a 4,000-occurrence run in the "mixed" batches)
                                                            (200000, 128, 40014, "criteo", False, 1),
                                                            (60000, 32, 50000, "mixed", False, 1)])
def test_framed_step_world1_replays_from_graphs(dev, rows, width, n, kind, graphs, block):
    """World size 1: pull and push of every step replay from hipGraphs (three routing slots -> three graphs of each
    kind for a fixed set of buffers); every pulled row and the table after every push equal the oracle's."""
    from herald_amd.sharded import FramedStep, ShardedEmbedding
    nb = (9 * block + 3 if block < 16 else 20) if n < 30000 else 5
    ids = _stream(rows, n, nb, kind)
    n = ids[0].size
    rng = np.random.default_rng(5)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    emb = ShardedEmbedding(rows, width, dev, table=torch.from_numpy(table.copy()).to(dev))
    fs = FramedStep(emb, n, graphs=graphs, block=block)
    assert fs.sized == (not graphs)         # eager launches: sized exchanges, own keys served from the shard directly
    LA = fs.LOOKAHEAD
    d_ids = [torch.from_numpy(x).to(dev) for x in ids]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(3)]
    d_grads = [torch.from_numpy(g).to(dev) for g in grads]
    outs = [torch.empty((n, width), device=dev) for _ in range(3)]
    want = table.copy()
    fs.start(d_ids[:LA])
    for k in range(nb):
        got = fs.pull(d_ids[k + LA] if k + LA < nb else None, out=outs[k % 3])
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy().reshape(n, width), want[ids[k].astype(np.int64)],
                                      err_msg="pull of step %d" % k)
        fs.push(d_grads[k % 3], 0.05)
        cpu.sparse_push(want, ids[k], grads[k % 3], 0.05)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(emb.table.cpu().numpy(), want, err_msg="table after step %d" % k)
    assert fs.fallbacks == 0
    if graphs and block < 16:
        assert fs.graphs and sum(1 for g in fs._graphs.values() if g) >= 6, fs._graphs


def test_framed_step_overflow_takes_the_sized_exchange(dev):
    """row_cap below the number of unique keys: the routing flags the batch two steps ahead and that batch runs
    through ShardedEmbedding's sized exchange; results are unchanged."""
    from herald_amd.sharded import FramedStep, ShardedEmbedding
    rows, width, n, nb = 5000, 64, 1200, 6
    ids = _stream(rows, n, nb, "mixed", seed=3)
    ids[1][:] = ids[1][0]                      # one unique key: fits any frame
    rng = np.random.default_rng(6)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    emb = ShardedEmbedding(rows, width, dev, table=torch.from_numpy(table.copy()).to(dev))
    fs = FramedStep(emb, n, row_cap=64, block=2)
    LA = fs.LOOKAHEAD
    d_ids = [torch.from_numpy(x).to(dev) for x in ids]
    want = table.copy()
    for_u64 = torch.from_numpy(ids[4].astype(np.int64)).to(dev)     # a uint64-keyed batch in the same stream
    d_ids[4] = for_u64
    fs.start(d_ids[:LA])
    for k in range(nb):
        g = rng.standard_normal((n, width), dtype=np.float32)
        got = fs.pull(d_ids[k + LA] if k + LA < nb else None)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy().reshape(n, width), want[ids[k].astype(np.int64)])
        fs.push(torch.from_numpy(g).to(dev), 0.1)
        cpu.sparse_push(want, ids[k], g, 0.1)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(emb.table.cpu().numpy(), want)
    assert fs.fallbacks == nb - 1              # every batch but the one-key batch


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _staged(out, inp, out_splits, in_splits, group):
    torch.cuda.current_stream().synchronize()
    o = torch.empty(out.shape, dtype=out.dtype)
    dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
    out.copy_(o)


def _worker(rank, world, port, rows, width, n, kind, row_cap, expect_fallback, sized=True):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from herald_amd.sharded import FramedStep, ShardedEmbedding, partition
    rng = np.random.default_rng(77)
    table_g = rng.standard_normal((rows, width), dtype=np.float32)
    starts = partition(rows, world)
    emb = ShardedEmbedding(rows, width, dev, table=torch.from_numpy(table_g[starts[rank]:starts[rank + 1]].copy()).to(dev),
                           a2a=_staged)
    nb = 5
    streams = [_stream(rows, n, nb, kind, seed=10 + r) for r in range(world)]
    nn = streams[0][0].size
    if world > 2:
        streams[world - 1][2] = streams[world - 1][2][:0]             # an empty batch on the last rank
    fs = FramedStep(emb, nn, row_cap=row_cap, block=2, graphs=False, sized=sized)  # (a host-staged exchange cannot be captured)
    assert fs.sized == sized
    tid = lambda k: torch.from_numpy(streams[rank][k]).to(dev) if k < nb else None
    LA = fs.LOOKAHEAD
    fs.start([tid(k) for k in range(LA)])
    want = table_g.copy()
    for k in range(nb):
        vals = [np.random.default_rng(500 + k * world + r).standard_normal((streams[r][k].size, width), dtype=np.float32)
                for r in range(world)]
        got = fs.pull(tid(k + LA))
        torch.cuda.synchronize()
        mine = streams[rank][k]
        if mine.size:
            np.testing.assert_array_equal(got.cpu().numpy().reshape(-1, width), want[mine.astype(np.int64)],
                                          err_msg="pull, step %d rank %d" % (k, rank))
        fs.push(torch.from_numpy(vals[rank]).to(dev), 0.05)
        torch.cuda.synchronize()
        dist.barrier()
        for r in range(world):
            if streams[r][k].size:
                cpu.sparse_push(want, streams[r][k], vals[r], 0.05)
        np.testing.assert_array_equal(emb.table.cpu().numpy(), want[starts[rank]:starts[rank + 1]],
                                      err_msg="shard after push, step %d rank %d" % (k, rank))
    assert (fs.fallbacks > 0) == expect_fallback, fs.fallbacks
    if sized and not expect_fallback:
        # the row exchanges carried exactly the rows the batches name (the reference's messages hold U_s x d floats per
        # server: PSAgent.h:167-172,217-226) -- no frame padding crosses the fabric
        st = np.asarray(starts)
        real = 0
        for k in range(nb):
            u = [np.unique(cpu.ids_to_keys(streams[r][k])).astype(np.int64) for r in range(world)]
            mine = u[rank]
            real += int(((mine < st[rank]) | (mine >= st[rank + 1])).sum())
            real += sum(int(((u[r] >= st[rank]) & (u[r] < st[rank + 1])).sum()) for r in range(world) if r != rank)
        carried = emb.stats["xgmi_row_bytes"]
        assert carried == 2 * 4 * width * real, (carried, 2 * 4 * width * real)
        assert carried <= 1.15 * 2 * 4 * width * real
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,width,n,kind,row_cap,expect_fallback,sized",
                         [(2, 5000, 64, 1300, "mixed", 300, True, True), (2, 5000, 64, 1300, "mixed", None, False, True),
                          (4, 200000, 128, 6656, "criteo", None, False, True), (8, 33762, 32, 2600, "mixed", 700, False, True),
                          (8, 33762, 32, 2600, "mixed", 400, True, True),
                          # BASELINE configs[1] / configs[2] streams at W = 8 (bs 256 and bs 4096 per rank, 26 fields)
                          (8, 337625, 32, 6656, "criteo", None, False, True), (8, 337625, 16, 106496, "criteo", None, False, True),
                          # fixed row frames (the form that replays from hipGraphs)
                          (2, 5000, 64, 1300, "mixed", None, False, False), (8, 33762, 32, 2600, "mixed", 400, True, False)])
def test_framed_step_at_world_size_gt_1_on_one_gpu(dev, world, rows, width, n, kind, row_cap, expect_fallback, sized):
    mp.spawn(_worker, args=(world, _free_port(), rows, width, n, kind, row_cap, expect_fallback, sized), nprocs=world,
             join=True)


@pytest.mark.parametrize("rows,width,n,kind,block,run", [(50000, 128, 6656, "criteo", 4, 4), (3000, 64, 900, "mixed", 8, 3),
                                                         (200, 32, 40, "mixed", 2, 2)])
def test_framed_step_runs_of_steps_by_one_native_call(dev, rows, width, n, kind, block, run):
    """FramedStep.steps: a run of steps (pull of a batch + push of its gradients) enqueued by ONE library call
    (ha_shard_steps, csrc/shard.hip) -- every pulled row and the table equal the oracle's serial PS semantics
    (oracle/cpu.py sparse_pull / sparse_push: PSAgent.h:124-237, PSFHandle.h:101-164), as for pull / push step by step; the
    runs end where the stream does, and the last steps go step by step again."""
    from herald_amd.sharded import FramedStep, ShardedEmbedding
    nb = 6 * block + 3
    ids = _stream(rows, n, nb, kind, seed=3)
    n = ids[0].size
    rng = np.random.default_rng(15)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    emb = ShardedEmbedding(rows, width, dev, table=torch.from_numpy(table.copy()).to(dev))
    fs = FramedStep(emb, n, graphs=False, block=block)
    assert fs.sized and fs.native_ok()
    LA = fs.LOOKAHEAD
    d_ids = [torch.from_numpy(x).to(dev) for x in ids]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(nb)]
    d_grads = [torch.from_numpy(g).to(dev) for g in grads]
    want = table.copy()
    fs.start(d_ids[:LA])
    k = 0
    lr = 0.05
    while k < nb:
        cnt = min(run, nb - k, block - k % block)
        if k >= nb - 3:              # the tail of the stream: step by step
            out = fs.pull(d_ids[k + LA] if k + LA < nb else None)
            got = [out]
            torch.cuda.synchronize()
            np.testing.assert_array_equal(got[0].cpu().numpy().reshape(-1, width), want[ids[k].astype(np.int64)])
            fs.push(d_grads[k], lr)
            cpu.sparse_push(want, ids[k], grads[k], lr)
            k += 1
            continue
        outs = [torch.empty((n, width), device=dev) for _ in range(cnt)]
        got = fs.steps([d_ids[k + i + LA] if k + i + LA < nb else None for i in range(cnt)], d_grads[k:k + cnt], lr, outs=outs)
        torch.cuda.synchronize()
        for i in range(cnt):
            np.testing.assert_array_equal(got[i].cpu().numpy().reshape(-1, width), want[ids[k + i].astype(np.int64)],
                                          err_msg="rows of batch %d" % (k + i))
            cpu.sparse_push(want, ids[k + i], grads[k + i], lr)
        k += cnt
    torch.cuda.synchronize()
    np.testing.assert_array_equal(emb.table.cpu().numpy(), want)
    with pytest.raises(ValueError):
        fs2 = FramedStep(emb, n, graphs=False, block=2)
        fs2.start(d_ids[:fs2.LOOKAHEAD])
        fs2.steps([None] * 3, d_grads[:3], lr)       # three steps from step 0 cross the boundary of blocks of 2
