"""The HIP engine of herald_amd.sharded at world sizes 2, 4 and 8 on ONE GPU: every rank is a process with
its own shard and HIP context on cuda:0, the collectives run on a gloo group and the row / key buffers
are staged through the host for them (ShardedEmbedding's `a2a` hook; RCCL does not allow two ranks on
one device).  Everything else -- ha_shard_route_* with nshard > 1, owner gathers, occurrence-ordered
reduce, rank-ordered ha_shard_serve_push -- is the product path.  Checked against the oracle's serial
PS semantics (oracle/cpu.py sparse_pull / sparse_push, rank order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def host_staged_a2a(out, inp, out_splits, in_splits, group):
    torch.cuda.current_stream().synchronize()
    o = torch.empty(out.shape, dtype=out.dtype)
    dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
    out.copy_(o)


def _worker(rank, world, port, rows, width, n, ids_kind):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from herald_amd.sharded import ShardedEmbedding, partition
    from herald_amd import synth
    from oracle import cpu

    rng = np.random.default_rng(4321)              # the same stream on every rank
    table_g = rng.standard_normal((rows, width), dtype=np.float32)
    starts = partition(rows, world)
    emb = ShardedEmbedding(rows, width, dev, table=torch.from_numpy(table_g[starts[rank]:starts[rank + 1]].copy()).to(dev),
                           a2a=host_staged_a2a)
    want = table_g.copy()

    def batch(step, r):
        if ids_kind == "criteo":
            return (synth.criteo_batch(n // 26, step * world + r).reshape(-1) % rows).astype(np.float32)
        g = np.random.default_rng(step * 100 + r)
        ids = g.integers(0, rows, size=n).astype(np.float32)
        ids[: n // 4] = np.random.default_rng(step).integers(0, rows, size=n // 4)     # keys shared between ranks
        ids[n // 4: n // 3] = ids[0]                                                   # a long run inside a rank
        return ids

    lr = 0.05
    route = emb.prefetch(torch.from_numpy(batch(0, rank)).to(dev), after_current=False)
    for k in range(4):
        ids_all = [batch(k, r) for r in range(world)]
        nn = ids_all[0].size
        vals_all = [np.random.default_rng(7 + k * world + r).standard_normal((nn, width), dtype=np.float32)
                    for r in range(world)]
        cur = route
        if k + 1 < 4:
            route = emb.prefetch(torch.from_numpy(batch(k + 1, rank)).to(dev), after_current=False)
        got = emb.pull(route=cur)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), cpu.sparse_pull(want, ids_all[rank]),
                                      err_msg="pull, step %d rank %d" % (k, rank))
        emb.push(None, torch.from_numpy(vals_all[rank]).to(dev), lr, route=cur)
        if k + 1 < 4:
            emb.complete(route)
        torch.cuda.synchronize()
        dist.barrier()
        for r in range(world):                               # servers apply in rank order
            cpu.sparse_push(want, ids_all[r], vals_all[r], lr)
        np.testing.assert_array_equal(emb.table.cpu().numpy(), want[starts[rank]:starts[rank + 1]],
                                      err_msg="shard after push, step %d rank %d" % (k, rank))
    # SSPushPull (ParameterServerCommunicate.py:74-76): push this batch, pull the next
    ids_all = [batch(9, r) for r in range(world)]
    nxt_all = [batch(10, r) for r in range(world)]
    vals_all = [np.random.default_rng(90 + r).standard_normal((ids_all[0].size, width), dtype=np.float32)
                for r in range(world)]
    got = emb.push_pull(torch.from_numpy(ids_all[rank]).to(dev), torch.from_numpy(vals_all[rank]).to(dev), lr,
                        torch.from_numpy(nxt_all[rank]).to(dev))
    torch.cuda.synchronize()
    dist.barrier()
    for r in range(world):
        cpu.sparse_push(want, ids_all[r], vals_all[r], lr)
    # the pushes are a collective: every rank's push is applied (rank order) before any rank's pull is served
    np.testing.assert_array_equal(got.cpu().numpy(), cpu.sparse_pull(want, nxt_all[rank]), err_msg="push_pull rows")
    np.testing.assert_array_equal(emb.table.cpu().numpy(), want[starts[rank]:starts[rank + 1]])
    assert emb.stats["xgmi_bytes_out"] > 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,width,n,ids_kind", [(2, 5000, 64, 1300, "mixed"), (4, 200000, 128, 6656, "criteo"),
                                                         (8, 33762, 32, 2600, "mixed")])
def test_hip_engine_at_world_size_gt_1_on_one_gpu(dev, world, rows, width, n, ids_kind):
    mp.spawn(_worker, args=(world, _free_port(), rows, width, n, ids_kind), nprocs=world, join=True)


def test_shard_route_with_8_virtual_shards(dev, lib):
    """ha_shard_route_f32ids with nshard = 8 on the real key space: shard-local keys and per-owner counts
    against AveragePartitioner ranges (oracle partition) + np.searchsorted."""
    import ctypes
    from herald_amd import ops, synth
    from oracle import cpu
    rows, nshard, n = 33762577, 8, 6656
    ids = np.minimum(synth.as_f32_ids(synth.criteo_batch(256, 3, rows=rows)).reshape(-1), rows - 1)
    d_ids = torch.from_numpy(ids).to(dev)
    plan = ops.IndexPlan(n, dev)
    starts = [int(x) for x in cpu.partition(rows, nshard)]
    st = (ctypes.c_int64 * (nshard + 1))(*starts)
    meta = torch.zeros(1 + nshard, dtype=torch.int64, device=dev)
    local = torch.zeros(n, dtype=torch.int32, device=dev)
    rc = lib.ha_shard_route_f32ids(ctypes.c_void_p(d_ids.data_ptr()), n, ctypes.c_void_p(plan.ws.data_ptr()), st,
                                   nshard, ctypes.c_void_p(meta.data_ptr()), ctypes.c_void_p(local.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    uniq = np.unique(cpu.ids_to_keys(ids)).astype(np.int64)
    owner = np.searchsorted(np.array(starts[1:]), uniq, side="right")
    m = meta.cpu().numpy()
    assert m[0] == uniq.size
    np.testing.assert_array_equal(m[1:], np.bincount(owner, minlength=nshard))
    np.testing.assert_array_equal(local.cpu().numpy()[:uniq.size].astype(np.int64), uniq - np.array(starts)[owner])
