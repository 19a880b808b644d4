"""The HET cache in front of a store it cannot address (herald_amd.remote_store): the cache kernels fill a
request / an outbox and read an inbox, the store's owner takes kSyncEmbedding's decision and applies
kPushEmbedding (ps-lite/src/PSFhandle_embedding.cc:5-79).  The semantics are those of the direct store, so
the same traces against oracle/cache_model.py must come out bit-identical:
  * LocalStore  -- the protocol alone, on one GPU (every policy, push keys, push_pull);
  * HostStore   -- the table in pinned host DRAM (cold tier, BASELINE configs[4]);
  * ShardedStore -- W processes on one GPU, each with its own cache and shard (host-staged all-to-all under
    gloo), against W cache models sharing one server, pushes applied in rank order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from herald_amd import cache as hcache
from herald_amd import remote_store
from oracle import cache_model

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_cache import _run_trace, run_push_pull_trace  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bind_local(gpu, table, versions):
    gpu.bind_remote(remote_store.LocalStore(table, versions))


@pytest.mark.parametrize("policy", ["lru", "lfu", "lfuopt"])
@pytest.mark.parametrize("pull_bound,push_bound", [(0, 0), (3, 3)])
def test_remote_protocol_traces(dev, policy, pull_bound, push_bound):
    _run_trace(dev, limit=40, rows=400, width=8, n=64, steps=60, pull_bound=pull_bound, push_bound=push_bound,
               policy=policy, seed=11, bind=_bind_local)


@pytest.mark.parametrize("policy", ["lru", "lfu"])
def test_remote_update_of_the_looked_up_keys_reuses_the_plan(dev, policy):
    """same_as_lookup=True through a remote store: the update reuses the index plan ha_cache_lookup_begin left in the
    workspace (ha_cache_update_same_keys, general path: the outbox carries the pushes)."""
    gpu, _ = _run_trace(dev, limit=100, rows=1500, width=8, n=64, steps=50, pull_bound=2, push_bound=2, policy=policy,
                        seed=17, bind=_bind_local, same=True)
    # LRU, limit >= batch: the two-launch update, its pushes and evicted lines in the outbox (cache_update_same_post_kernel)
    assert int(gpu._L.ha_cache_fused_updates(gpu._h)) == (50 if policy == "lru" else 0)
    gpu, _ = _run_trace(dev, limit=16, rows=300, width=4, n=48, steps=30, pull_bound=1, push_bound=1, zipf=False,
                        policy=policy, seed=18, bind=_bind_local, same=True, extra_lookup_every=4)
    assert int(gpu._L.ha_cache_fused_updates(gpu._h)) == 0


def test_remote_update_of_the_looked_up_keys_with_evictions_and_wide_rows(dev):
    gpu, _ = _run_trace(dev, limit=600, rows=5000, width=128, n=416, steps=14, pull_bound=2, push_bound=1, seed=19,
                        check_every=4, bind=_bind_local, same=True, zipf=False)
    assert int(gpu._L.ha_cache_fused_updates(gpu._h)) == 14


def test_remote_heavy_eviction_and_limit_below_batch(dev):
    _run_trace(dev, limit=16, rows=1000, width=4, n=48, steps=40, pull_bound=2, push_bound=2, zipf=False, seed=3,
               bind=_bind_local)
    _run_trace(dev, limit=5, rows=200, width=4, n=40, steps=25, pull_bound=1, push_bound=1, zipf=False, seed=4,
               bind=_bind_local)


@pytest.mark.parametrize("policy", ["lru", "lfu"])
def test_remote_with_push_keys(dev, policy):
    _run_trace(dev, limit=50, rows=500, width=8, n=80, steps=40, pull_bound=3, push_bound=3, push_keys_mode=True,
               seed=5, policy=policy, bind=_bind_local)


def test_remote_criteo_width_and_long_runs(dev):
    _run_trace(dev, limit=300, rows=5000, width=128, n=416, steps=12, pull_bound=2, push_bound=2, seed=6,
               check_every=4, bind=_bind_local)
    _run_trace(dev, limit=500, rows=3000, width=512, n=2000, steps=4, pull_bound=1, push_bound=2, seed=9,
               bind=_bind_local)


@pytest.mark.parametrize("policy", ["lru", "lfu", "lfuopt"])
def test_remote_push_pull_trace(dev, policy):
    run_push_pull_trace(dev, policy, bind=_bind_local)


@pytest.mark.parametrize("own_stream", [False, True])
def test_cold_tier_host_store(dev, own_stream):
    """The table in pinned host memory (HostStore): rows are staged over PCIe by the owner-side kernels (on the caller's
    stream, or on a copy stream handed to the store), versions stay in HBM.  Same trace, same oracle; the host table ends
    up equal to the model's server table."""
    host = {}

    def bind(gpu, table, versions):
        t = torch.empty(tuple(table.shape), dtype=torch.float32, pin_memory=True)
        t.copy_(table)
        st = remote_store.HostStore(table.shape[0], table.shape[1], dev, table=t,
                                    stream=torch.cuda.Stream(device=dev) if own_stream else None)
        host["store"], host["dev_table"], host["dev_versions"] = st, table, versions
        gpu.bind_remote(st)

    rng = np.random.default_rng(8)
    rows, width, n, limit = 20000, 64, 800, 1500
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    server = cache_model.Server(table0)
    model = cache_model.CacheModel("lru", limit, width, server, 2, 2)
    gpu = hcache.LRUCache(limit, rows, width, node_id=0, max_batch=n, device=dev)
    bind(gpu, torch.from_numpy(table0.copy()).to(dev), None)
    gpu.pull_bound = gpu.push_bound = 2
    st = host["store"]
    for step in range(12):
        keys = ((np.minimum(rng.zipf(1.2, size=n) - 1, rows - 1) * 7919) % rows).astype(np.float32)
        want = model.lookup(keys.astype(np.uint64))
        dest = torch.empty((n, width), dtype=torch.float32, device=dev)
        kt = torch.from_numpy(keys).to(dev)
        gpu.embedding_lookup(kt, dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="lookup, step %d" % step)
        g = rng.standard_normal((n, width), dtype=np.float32) * np.float32(-0.01)
        model.update(keys.astype(np.uint64), g)
        if step % 2:            # every other step: the update names the lookup's key tensor (plan reuse)
            gpu.embedding_update(kt, torch.from_numpy(g).to(dev), same_as_lookup=True).wait()
        else:
            gpu.embedding_update(torch.from_numpy(keys).to(dev), torch.from_numpy(g).to(dev)).wait()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(st.versions.cpu().numpy(), server.ver, err_msg="versions, step %d" % step)
        np.testing.assert_array_equal(st.table.numpy(), server.table, err_msg="host table, step %d" % step)


# ---- W caches over one sharded store -------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def host_staged_a2a(out, inp, out_splits, in_splits, group):
    torch.cuda.current_stream().synchronize()
    o = torch.empty(out.shape, dtype=out.dtype)
    dist.all_to_all_single(o, inp.cpu().contiguous(), out_splits, in_splits, group=group)
    out.copy_(o)


def _shard_worker(rank, world, port, rows, width, n, limit, steps, policy, mode, host_local=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from herald_amd.sharded import partition
    rng = np.random.default_rng(99)                       # identical on every rank
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    starts = partition(rows, world)
    if host_local:      # BASELINE configs[4]'s composition: every rank's shard in pinned host DRAM
        shard = torch.from_numpy(table0[starts[rank]:starts[rank + 1]].copy()).pin_memory()
        local = remote_store.HostStore(shard.shape[0], width, dev, table=shard)
    else:
        shard = torch.from_numpy(table0[starts[rank]:starts[rank + 1]].copy()).to(dev)
        local = remote_store.LocalStore(shard)
    store = remote_store.ShardedStore(rows, width, dev, local, a2a=host_staged_a2a)
    cls = {"lru": hcache.LRUCache, "lfu": hcache.LFUCache, "lfuopt": hcache.LFUOptCache}[policy]
    gpu = cls(limit, rows, width, node_id=0, max_batch=n, device=dev)
    gpu.bind_remote(store)
    gpu.pull_bound = gpu.push_bound = 2
    gpu.perf_enabled = True
    # the oracle: one server, one cache model per rank
    server = cache_model.Server(table0)
    models = [cache_model.CacheModel(policy, limit, width, server, 2, 2) for _ in range(world)]

    def keys_of(step, r):
        g = np.random.default_rng(1000 * step + r)
        k = (np.minimum(g.zipf(1.25, size=n) - 1, rows - 1) * 7919) % rows
        k[: n // 5] = (np.random.default_rng(step).integers(0, rows, size=n // 5))     # shared between ranks
        return k.astype(np.float32)

    dest = torch.empty((n, width), dtype=torch.float32, device=dev)
    if mode == "push_pull":
        want0 = [m.lookup(keys_of(0, r).astype(np.uint64)) for r, m in enumerate(models)]
        gpu.embedding_lookup(torch.from_numpy(keys_of(0, rank)).to(dev), dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want0[rank])
    for step in range(steps):
        grads = [np.random.default_rng(5000 * step + r).standard_normal((n, width), dtype=np.float32) *
                 np.float32(-0.01) for r in range(world)]
        if mode == "lookup_update":
            wants = [m.lookup(keys_of(step, r).astype(np.uint64)) for r, m in enumerate(models)]
            gpu.embedding_lookup(torch.from_numpy(keys_of(step, rank)).to(dev), dest).wait()
            np.testing.assert_array_equal(dest.cpu().numpy(), wants[rank], err_msg="lookup, step %d rank %d" % (step, rank))
            for r, m in enumerate(models):                                      # pushes land in rank order
                m.update(keys_of(step, r).astype(np.uint64), grads[r])
            gpu.embedding_update(torch.from_numpy(keys_of(step, rank)).to(dev), torch.from_numpy(grads[rank]).to(dev)).wait()
        else:
            # every rank's push phase reaches the server (rank order) before any rank's sync is served
            wants = [None] * world
            for r, m in enumerate(models):
                m.push_pull_begin(keys_of(step + 1, r).astype(np.uint64), keys_of(step, r).astype(np.uint64), grads[r])
            for r, m in enumerate(models):
                wants[r] = m.push_pull_finish()
            gpu.embedding_push_pull(torch.from_numpy(keys_of(step + 1, rank)).to(dev), dest,
                                    torch.from_numpy(keys_of(step, rank)).to(dev),
                                    torch.from_numpy(grads[rank]).to(dev)).wait()
            np.testing.assert_array_equal(dest.cpu().numpy(), wants[rank], err_msg="push_pull, step %d rank %d" % (step, rank))
        torch.cuda.synchronize()
        dist.barrier()
        np.testing.assert_array_equal(local.versions.cpu().numpy(), server.ver[starts[rank]:starts[rank + 1]],
                                      err_msg="versions, step %d rank %d" % (step, rank))
        np.testing.assert_array_equal(shard.cpu().numpy(), server.table[starts[rank]:starts[rank + 1]],
                                      err_msg="shard, step %d rank %d" % (step, rank))
        got, exp = gpu.perf[-1], models[rank].perf[-1]
        for f in ("type", "num_all", "num_unique", "num_miss", "num_transfered"):
            assert got[f] == exp[f], (step, rank, f, got, exp)
    res, lines = models[rank].resident(), gpu.lines()
    assert sorted(lines.keys()) == sorted(res.keys())
    for k, ln in res.items():
        assert lines[k].version == ln.version and lines[k].updates == ln.updates, k
        np.testing.assert_array_equal(lines[k].data, ln.data)
    # only misses, stale lines and pushed lines crossed the fabric
    assert store.stats["rows_pulled"] < store.stats["keys_synced"]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,policy,mode", [(2, "lru", "lookup_update"), (4, "lru", "lookup_update"),
                                               (4, "lfu", "lookup_update"), (2, "lru", "push_pull")])
def test_caches_over_a_sharded_store(dev, world, policy, mode):
    mp.spawn(_shard_worker, args=(world, _free_port(), 6000, 32, 500, 400, 10, policy, mode), nprocs=world, join=True)


def test_caches_over_a_sharded_host_dram_table(dev):
    """configs[4] composed at small scale: the table row-range sharded over the ranks AND kept in pinned host
    DRAM on each of them, an HBM hot tier per rank in front."""
    mp.spawn(_shard_worker, args=(2, _free_port(), 6000, 64, 500, 400, 8, "lru", "lookup_update", True), nprocs=2,
             join=True)


# ---- BASELINE configs[3] composed: laia scheduler -> LAIADataloader -> cache over the sharded table ----------
def _laia_worker(rank, world, port):
    """Every rank runs its own laia scheduler over the same samples (as run_laia.py does), takes ITS share of
    each global batch and ITS push plan from the LAIADataloader, looks the rows up through its cache over the
    table sharded across the ranks, and pushes with embedding_update_with_push_keys.  Oracle: the laia model's
    stream per rank, one cache model per rank, one server."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from herald_amd import hetu_ops, laia as hlaia, synth
    from herald_amd.sharded import partition
    from oracle import laia_model
    rows, width, mini_bs, limit, T = 60000, 32, 64, 3000, 26
    samples = np.concatenate([synth.criteo_batch(64, step=300 + s, rows=rows, nfields=T) for s in range(world * 8)], axis=0)
    rng = np.random.default_rng(17)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    starts = partition(rows, world)
    local = remote_store.LocalStore(torch.from_numpy(table0[starts[rank]:starts[rank + 1]].copy()).to(dev))
    store = remote_store.ShardedStore(rows, width, dev, local, a2a=host_staged_a2a)
    gpu = hcache.LRUCache(limit, rows, width, node_id=0, max_batch=mini_bs * T, device=dev)
    gpu.bind_remote(store)
    gpu.pull_bound = gpu.push_bound = 3
    sched = hlaia.LAIAScheduler(samples.astype(np.float32), batch_size=mini_bs)
    sched.start(nrank=world, rank=rank, cache_limit=limit, dataset_num=1, epoch_num=1, key_limit=rows)
    dl = hlaia.LAIADataloader(sched, 0, True, samples.astype(np.float32), mini_bs, name="train", device=dev)
    dl.init_states(rank, world)
    grad_op = hetu_ops.EmbeddingLookUp_Gradient((rows, width), enable_push_index=True)
    # oracle side: every rank's stream, cache model and the one server
    server = cache_model.Server(table0)
    models = [cache_model.CacheModel("lru", limit, width, server, 3, 3) for _ in range(world)]
    streams = [laia_model.LaiaSchedulerModel(sched.sparse_data.astype(np.uint64), 1, sched.batch_size, sched.batch_num,
                                             world, r, limit).emit() for r in range(world)]
    for b in range(4):
        ids, plan = dl.get_arr()
        np.testing.assert_array_equal(ids.cpu().numpy(), samples[streams[rank][2 * b + 1]].astype(np.float32))
        np.testing.assert_array_equal(plan.cpu().numpy(), np.asarray(streams[rank][2 * b + 2], dtype=np.float32))
        keys = [samples[streams[r][2 * b + 1]].reshape(-1).astype(np.uint64) for r in range(world)]
        wants = [models[r].lookup(keys[r]) for r in range(world)]
        dest = torch.empty((mini_bs * T, width), dtype=torch.float32, device=dev)
        gpu.embedding_lookup(ids.reshape(-1), dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), wants[rank], err_msg="lookup, batch %d rank %d" % (b, rank))
        grads = [np.random.default_rng(900 + 10 * b + r).standard_normal((mini_bs * T, width), dtype=np.float32) *
                 np.float32(-0.01) for r in range(world)]
        slices = grad_op.compute(torch.from_numpy(grads[rank]).to(dev), (ids, plan))
        gpu.embedding_update_with_push_keys(slices.indices.reshape(-1), slices.push_indices, slices.values).wait()
        for r in range(world):                                                   # pushes land in rank order
            models[r].update_with_push_keys(keys[r], np.asarray(streams[r][2 * b + 2], dtype=np.uint64), grads[r])
        torch.cuda.synchronize()
        dist.barrier()
        np.testing.assert_array_equal(local.versions.cpu().numpy(), server.ver[starts[rank]:starts[rank + 1]],
                                      err_msg="versions, batch %d rank %d" % (b, rank))
        np.testing.assert_array_equal(local.table.cpu().numpy(), server.table[starts[rank]:starts[rank + 1]],
                                      err_msg="shard, batch %d rank %d" % (b, rank))
    sched.sched.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_laia_dataloader_cache_and_sharded_table_composed(dev, world):
    mp.spawn(_laia_worker, args=(world, _free_port()), nprocs=world, join=True)
