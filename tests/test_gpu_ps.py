"""The libps names (libherald_ps.so) on the GPU: the scenario of the reference's
tests/pstests/test_apis.py:105-157 (30x40 duplicated ids into a 2000x1000 table, rows of ones pushed,
rtol 5e-7 there -- bit-exact here against oracle sparse_pull / sparse_push), host and device DLArrays,
SSPushPull, Wait, SaveParam / LoadParam, InitTensor initialisers; and two ranks on one GPU through the
sharded backend."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from herald_amd import ops, ps
from oracle import cpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _h(t):
    return ops.DLHolder(t)


def test_sparse_pull_push_scenario_of_the_reference(dev, tmp_path):
    P = ps.lib()
    node, rows, width = 11, 2000, 1000
    lrs = (ctypes.c_float * 1)(0.1)
    P.InitTensor(node, 2, rows, width, 2, 0.0, 0.01, 123, 0, lrs, 1)     # Normal(0, 0.01), CacheSparse table
    P.Wait(node)
    shard, info = ps.tensor(node, dev)
    assert (info.len, info.width, info.row_start, info.rows_local) == (rows, width, 0, rows)
    t0 = shard.cpu().numpy().copy()
    assert abs(t0.mean()) < 1e-4 and abs(t0.std() - 0.01) < 2e-4          # the initialiser's distribution
    rng = np.random.default_rng(0)
    ids = rng.integers(0, rows, size=(30, 40)).astype(np.float32)
    # device arrays
    d_ids = torch.from_numpy(ids).to(dev)
    out = torch.empty((30, 40, width), dtype=torch.float32, device=dev)
    hi, ho = _h(d_ids), _h(out)
    P.SparsePull(node, hi.handle, ho.handle)
    P.Wait(node)
    np.testing.assert_array_equal(out.cpu().numpy(), cpu.sparse_pull(t0, ids))
    ones = torch.ones((30, 40, width), dtype=torch.float32, device=dev)
    hv = _h(ones)
    P.SparsePush(node, hi.handle, hv.handle, None)
    P.Wait(node)
    want = cpu.sparse_push(t0.copy(), ids.reshape(-1), np.ones((1200, width), np.float32))
    np.testing.assert_array_equal(shard.cpu().numpy(), want)
    # host arrays (the reference's PS ops hand host NDArrays): staged on the tensor's stream
    ids2 = rng.integers(0, rows, size=500).astype(np.float32)
    vals2 = rng.standard_normal((500, width), dtype=np.float32)
    nxt = rng.integers(0, rows, size=500).astype(np.float32)
    out2 = np.empty((500, width), dtype=np.float32)
    hs = [_h(torch.from_numpy(x)) for x in (ids2, vals2, nxt, out2)]
    P.SSPushPull(node, hs[0].handle, hs[1].handle, hs[2].handle, hs[3].handle, None)
    P.Wait(node)
    cpu.sparse_push(want, ids2, vals2)
    np.testing.assert_array_equal(shard.cpu().numpy(), want)
    np.testing.assert_array_equal(out2, cpu.sparse_pull(want, nxt))
    # checkpoint: <address>/<node>_<part>.dat, raw fp32 (PSAgent.h:447-476)
    P.SaveParam(node, str(tmp_path).encode())
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / ("%d_0.dat" % node)), dtype=np.float32).reshape(rows, width), want)
    shard.zero_()
    torch.cuda.synchronize()
    P.LoadParam(node, str(tmp_path).encode())
    np.testing.assert_array_equal(shard.cpu().numpy(), want)
    assert P.rank() == 0 and P.nrank() == 1
    P.Clear(node)


def test_init_tensor_kinds_and_partition_independence(dev):
    P = ps.lib()
    lrs = (ctypes.c_float * 1)(0.0)
    P.InitTensor(21, 1, 5000, 16, 0, 0.5, 0.0, 7, 0, lrs, 1)          # Constant
    P.InitTensor(22, 1, 5000, 16, 1, -2.0, 3.0, 7, 0, lrs, 1)         # Uniform
    P.InitTensor(23, 1, 20000, 16, 3, 1.0, 0.5, 7, 0, lrs, 1)         # TruncatedNormal
    for n in (21, 22, 23):
        P.Wait(n)
    c = ps.tensor(21, dev)[0].cpu().numpy()
    u = ps.tensor(22, dev)[0].cpu().numpy()
    t = ps.tensor(23, dev)[0].cpu().numpy()
    assert (c == 0.5).all()
    assert u.min() >= -2.0 and u.max() <= 3.0 and abs(u.mean() - 0.5) < 0.05
    assert np.abs(t - 1.0).max() <= 1.0 + 1e-6 and abs(t.mean() - 1.0) < 0.01
    for n in (21, 22, 23):
        P.Clear(n)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from test_gpu_sharded_multirank import host_staged_a2a
    from herald_amd.sharded import ShardedEmbedding, partition
    rows, width, n = 3000, 64, 700
    rng = np.random.default_rng(5)
    table_g = rng.standard_normal((rows, width), dtype=np.float32)
    starts = partition(rows, world)
    emb = ShardedEmbedding(rows, width, dev, table=torch.from_numpy(table_g[starts[rank]:starts[rank + 1]].copy()).to(dev),
                           a2a=host_staged_a2a)
    ps.attach_sharded(31, emb, barrier=dist.barrier)
    P = ps.lib()
    assert P.rank() == rank and P.nrank() == world
    ids_all = [np.random.default_rng(10 + r).integers(0, rows, size=n).astype(np.float32) for r in range(world)]
    vals_all = [np.random.default_rng(20 + r).standard_normal((n, width), dtype=np.float32) for r in range(world)]
    d_ids = torch.from_numpy(ids_all[rank]).to(dev)
    out = torch.empty((n, width), dtype=torch.float32, device=dev)
    hi, ho, hv = _h(d_ids), _h(out), _h(torch.from_numpy(vals_all[rank]).to(dev))
    P.SparsePull(31, hi.handle, ho.handle)
    P.Wait(31)
    np.testing.assert_array_equal(out.cpu().numpy(), cpu.sparse_pull(table_g, ids_all[rank]))
    P.SparsePush(31, hi.handle, hv.handle, None)
    P.Wait(31)
    P.BarrierWorker()
    want = table_g.copy()
    for r in range(world):
        cpu.sparse_push(want, ids_all[r], vals_all[r])
    np.testing.assert_array_equal(emb.table.cpu().numpy(), want[starts[rank]:starts[rank + 1]])
    dist.barrier()
    dist.destroy_process_group()


def test_libps_names_over_two_ranks(dev):
    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)
