"""Multi-rank exchange logic of herald_amd.sharded under gloo (CPU, world_size 2 and 3).

The arithmetic is done by tests/cpu_engine.py (oracle-backed test double); what is under test is the
routing: AveragePartitioner ranges, counts / keys / rows all-to-alls, rank-ordered application."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, rows, width, n, out_dir, side_group=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_engine import CpuEngine
    from herald_amd.sharded import ShardedEmbedding, partition
    from oracle import cpu

    rng = np.random.default_rng(1234)              # same stream on every rank
    table_g = rng.standard_normal((rows, width), dtype=np.float32)
    all_ids = [rng.integers(0, rows, size=n).astype(np.float32) for _ in range(world)]
    for r in range(world):
        all_ids[r][: n // 3] = all_ids[0][: n // 3]          # keys shared between ranks
        all_ids[r][n // 3: n // 2] = all_ids[r][0]           # duplicates inside a rank
    all_vals = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(world)]
    starts = partition(rows, world)
    assert starts == list(cpu.partition(rows, world))
    shard = torch.from_numpy(table_g[starts[rank]:starts[rank + 1]].copy())
    emb = ShardedEmbedding(rows, width, "cpu", engine=CpuEngine(), table=shard, side_group=side_group)
    assert (emb.side_group is not None) == (side_group and world > 1)

    ids = torch.from_numpy(all_ids[rank])
    out = emb.pull(ids)
    np.testing.assert_array_equal(out.numpy(), cpu.sparse_pull(table_g, all_ids[rank]))

    lr = 0.05
    emb.push(ids, torch.from_numpy(all_vals[rank]), lr)
    dist.barrier()
    want = table_g.copy()
    for r in range(world):                                   # rank order
        cpu.sparse_push(want, all_ids[r], all_vals[r], lr)
    np.testing.assert_array_equal(emb.table.numpy(), want[starts[rank]:starts[rank + 1]])

    out2 = emb.pull(ids)                                     # pull after push sees every rank's update
    np.testing.assert_array_equal(out2.numpy(), want[all_ids[rank].astype(np.int64)])

    # software-pipelined schedule of bench.py's N>1 leg: the routing of batch k+1 is prefetched and
    # completed while batch k is pulled and pushed; results equal the serial pull / push sequence
    batches = [[rng.integers(0, rows, size=n).astype(np.float32) for _ in range(world)] for _ in range(4)]
    bvals = [[rng.standard_normal((n, width), dtype=np.float32) for _ in range(world)] for _ in range(4)]
    route = emb.prefetch(torch.from_numpy(batches[0][rank]), after_current=False)
    for k in range(4):
        cur = route
        if k + 1 < 4:
            route = emb.prefetch(torch.from_numpy(batches[k + 1][rank]), after_current=False)
        got = emb.pull(route=cur)
        np.testing.assert_array_equal(got.numpy(), want[batches[k][rank].astype(np.int64)])
        emb.push(None, torch.from_numpy(bvals[k][rank]), lr, route=cur)
        if k + 1 < 4:
            emb.complete(route)
        dist.barrier()
        for r in range(world):
            cpu.sparse_push(want, batches[k][r], bvals[k][r], lr)
        np.testing.assert_array_equal(emb.table.numpy(), want[starts[rank]:starts[rank + 1]])

    # SSPushPull (ParameterServerCommunicate.py:74-76): push this batch, then pull the next one; the pushes
    # are a collective, so every rank's update is in the rows any rank pulls
    pp_ids = [rng.integers(0, rows, size=n).astype(np.float32) for _ in range(world)]
    pp_next = [rng.integers(0, rows, size=n).astype(np.float32) for _ in range(world)]
    pp_vals = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(world)]
    got = emb.push_pull(torch.from_numpy(pp_ids[rank]), torch.from_numpy(pp_vals[rank]), lr,
                        torch.from_numpy(pp_next[rank]))
    for r in range(world):
        cpu.sparse_push(want, pp_ids[r], pp_vals[r], lr)
    np.testing.assert_array_equal(got.numpy(), want[pp_next[rank].astype(np.int64)])
    dist.barrier()
    np.testing.assert_array_equal(emb.table.numpy(), want[starts[rank]:starts[rank + 1]])

    # ragged batches: rank r brings n - 7r ids (the routing frame was agreed as the max over the ranks of the
    # first batch = n), empty batches, and a batch larger than the frame is refused
    rg = [rng.integers(0, rows, size=max(n - 7 * r, 0)).astype(np.float32) for r in range(world)]
    rg[world - 1] = rg[world - 1][:0]                                  # the last rank brings nothing
    got = emb.pull(torch.from_numpy(rg[rank]))
    np.testing.assert_array_equal(got.numpy().reshape(-1, width), want[rg[rank].astype(np.int64)].reshape(-1, width))
    assert emb.max_ids == n
    with pytest.raises(ValueError, match="routing frame"):
        emb.prefetch(torch.zeros(n + 1))
    dist.barrier()

    # 2-D id batches and the checkpoint format round trip
    ids2 = ids[: (n // 4) * 4].reshape(-1, 4)
    assert tuple(emb.pull(ids2).shape) == (ids2.shape[0], 4, width)
    emb.save(os.path.join(out_dir, "emb"))
    before = emb.table.clone()
    emb.table.zero_()
    emb.load(os.path.join(out_dir, "emb"))
    assert torch.equal(before, emb.table)
    assert emb.stats["xgmi_bytes_out"] > 0 or world == 1
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,width,n,side_group", [(2, 1001, 8, 300, False), (3, 50, 4, 64, False),
                                                          (2, 400, 4, 120, True)])
def test_sharded_pull_push_gloo(tmp_path, world, rows, width, n, side_group):
    mp.spawn(_worker, args=(world, _free_port(), rows, width, n, str(tmp_path), side_group), nprocs=world,
             join=True)


def _framed_worker(rank, world, port, rows, width, n, row_cap, block, expect_fallback, sized=True):
    """FramedStep under gloo, with SIZED row exchanges (the default: exactly the rows the batches name travel, own keys
    stay local) and with fixed row frames: a stream of batches with keys shared between the ranks, runs inside a rank,
    an empty batch on the last rank and -- with a small row_cap -- batches that overflow their key frames on SOME rank
    and must take the exchange with a read-back on ALL ranks."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_engine import CpuEngine
    from herald_amd.sharded import FramedStep, ShardedEmbedding, partition
    from oracle import cpu

    rng = np.random.default_rng(99)
    table_g = rng.standard_normal((rows, width), dtype=np.float32)
    starts = partition(rows, world)
    emb = ShardedEmbedding(rows, width, "cpu", engine=CpuEngine(),
                           table=torch.from_numpy(table_g[starts[rank]:starts[rank + 1]].copy()))
    nb = 6
    batches, vals = [], []
    for k in range(nb):
        per = []
        for r in range(world):
            m = n if not (k == 2 and r == world - 1) else 0           # one empty batch on the last rank
            ids = rng.integers(0, rows, size=m).astype(np.float32)
            if m:
                ids[: m // 4] = (np.arange(m // 4) * 3) % rows        # keys every rank names
                ids[m // 4: m // 3] = ids[0]                          # a run
            if k == 4 and m:                                          # everything from rank 0's range: many unique keys of ONE owner
                ids = rng.permutation(starts[1])[:m].astype(np.float32) if starts[1] >= m else ids % starts[1]
            per.append(ids)
        batches.append(per)
        vals.append([rng.standard_normal((b.size, width), dtype=np.float32) for b in per])
    fs = FramedStep(emb, n, row_cap=row_cap, block=block, graphs=False, sized=sized)
    assert fs.sized == sized
    tid = lambda k: torch.from_numpy(batches[k][rank]) if k < nb else None
    LA = fs.LOOKAHEAD
    fs.start([tid(k) for k in range(min(LA, nb))])
    want = table_g.copy()
    lr = 0.05
    for k in range(nb):
        got = fs.pull(tid(k + LA))
        ids = batches[k][rank]
        if ids.size:
            np.testing.assert_array_equal(got.numpy().reshape(-1, width), want[ids.astype(np.int64)],
                                          err_msg="pull, step %d rank %d" % (k, rank))
        else:
            assert got is None
        fs.push(torch.from_numpy(vals[k][rank]), lr)
        dist.barrier()
        for r in range(world):                                        # rank order
            if batches[k][r].size:
                cpu.sparse_push(want, batches[k][r], vals[k][r], lr)
        np.testing.assert_array_equal(emb.table.numpy(), want[starts[rank]:starts[rank + 1]],
                                      err_msg="shard after push, step %d rank %d" % (k, rank))
    assert (fs.fallbacks > 0) == expect_fallback, fs.fallbacks
    if sized and not expect_fallback:
        # what crossed the "fabric" in the row exchanges is exactly the rows the batches name (PSAgent.h:167-172,217-226):
        # per step and direction, the unique keys this rank names of the other owners + the ones the others name of its range
        st = np.asarray(starts)
        real = 0
        for k in range(nb):
            u = [np.unique(cpu.ids_to_keys(b)).astype(np.int64) for b in batches[k]]
            mine = u[rank]
            real += int(((mine < st[rank]) | (mine >= st[rank + 1])).sum())
            real += sum(int(((u[r] >= st[rank]) & (u[r] < st[rank + 1])).sum()) for r in range(world) if r != rank)
        carried = emb.stats["xgmi_row_bytes"]
        assert carried == 2 * 4 * width * real, (carried, 2 * 4 * width * real)       # pull and push
        assert carried <= 1.15 * 2 * 4 * width * real
        assert emb.stats["owner_steps"] == nb and emb.stats["owner_rows_max"] * nb >= emb.stats["owner_rows_sum"] > 0
    with pytest.raises(RuntimeError, match="ended"):
        fs.pull(None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,width,n,row_cap,block,expect_fallback,sized",
                         [(2, 1001, 8, 120, None, 2, True, True),    # default row_cap = n / W: batch 4 overflows owner 0
                          (2, 1001, 8, 120, 120, 1, False, True),    # row_cap = n can never overflow; blocks of one batch
                          (3, 700, 4, 90, 40, 4, True, True),        # the stream ends inside a block
                          (3, 700, 4, 90, 90, 4, False, True),       # three ranks, no overflow: the bytes carried are the rows named
                          (2, 1001, 8, 120, 120, 8, False, True),    # the whole stream is shorter than the lookahead
                          (2, 1001, 8, 120, None, 2, True, False),   # fixed row frames (the graph-replay form)
                          (3, 700, 4, 90, 90, 4, False, False)])
def test_framed_step_gloo(world, rows, width, n, row_cap, block, expect_fallback, sized):
    mp.spawn(_framed_worker, args=(world, _free_port(), rows, width, n, row_cap, block, expect_fallback, sized),
             nprocs=world, join=True)


def test_partition_is_average_partitioner():
    from herald_amd.sharded import partition
    s = partition(33762577, 8)
    assert s[1] - s[0] == 4220323 and all(s[i + 1] - s[i] == 4220322 for i in range(1, 8)) and s[-1] == 33762577
    assert partition(10, 3) == [0, 4, 7, 10]
