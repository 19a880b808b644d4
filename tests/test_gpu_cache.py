"""GPU parity of the device-resident HET cache against oracle/cache_model.py (bit-exact rows,
identical hit/miss/evict/transfer counts, versions and resident sets, step by step)."""
import numpy as np
import pytest
import torch

from herald_amd import cache as hcache
from oracle import cache_model

pytestmark = pytest.mark.gpu


def _compare_state(gpu, model, step):
    res = model.resident()
    lines = gpu.lines()
    assert sorted(lines.keys()) == sorted(res.keys()), "resident set differs at step %d" % step
    for k, ln in res.items():
        g = lines[k]
        assert g.version == ln.version, (step, k, g.version, ln.version)
        assert g.updates == ln.updates, (step, k)
        np.testing.assert_array_equal(g.data, ln.data, err_msg="data of key %d at step %d" % (k, step))
        if ln.grad is not None:
            np.testing.assert_array_equal(g.grad, ln.grad, err_msg="grad of key %d at step %d" % (k, step))


def _run_trace(dev, limit, rows, width, n, steps, pull_bound, push_bound, push_keys_mode=False, seed=0,
               zipf=True, check_every=1, policy="lru", bind=None, same=False, extra_lookup_every=0, ahead=False):
    """bind(gpu_cache, table, versions): how the cache reaches its store (default: bind_store, the table in
    the same HBM; test_gpu_cache_remote.py passes remote stores here).  same: the update names the lookup's key
    tensor (same_as_lookup=True -> ha_cache_update_same_keys); extra_lookup_every=k: every k-th step looks another
    batch up first, so that the evict list is not empty when the step's own lookup starts.  ahead: every step hands the
    NEXT step's key tensor to prefetch_keys between its lookup and its update (ha_cache_sort_ahead -> the next lookup is
    ha_cache_lookup_presorted); ahead=k (an int): every k-th step hands the key tensors of the next k steps to
    prefetch_keys_batch (ha_cache_sort_ahead_batch: one launch sorts them all, the lookups take them in order)."""
    rng = np.random.default_rng(seed)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    server = cache_model.Server(table0)
    model = cache_model.CacheModel(policy, limit, width, server, pull_bound, push_bound)
    table = torch.from_numpy(table0.copy()).to(dev)
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    cls = {"lru": hcache.LRUCache, "lfu": hcache.LFUCache, "lfuopt": hcache.LFUOptCache}[policy]
    gpu = cls(limit, rows, width, node_id=0, max_batch=max(n, 64), device=dev)
    if bind is None:
        gpu.bind_store(table, versions)
    else:
        bind(gpu, table, versions)
    gpu.pull_bound, gpu.push_bound = pull_bound, push_bound
    gpu.perf_enabled = True
    def draw(r):
        if zipf:
            return (np.minimum(r.zipf(1.3, size=n) - 1, rows - 1).astype(np.int64) * 7919) % rows
        return r.integers(0, rows, size=n)

    if ahead:       # the key batches exist before their steps, as a data loader's do
        rk = np.random.default_rng(seed + 1000)
        keys_all = [draw(rk) for _ in range(steps)]
        kts = [torch.from_numpy(k.astype(np.float32)).to(dev) for k in keys_all]
    for step in range(steps):
        keys = keys_all[step] if ahead else draw(rng)
        fk = keys.astype(np.float32)                      # the *_raw entry points take float32 ids
        dest = torch.empty((n, width), dtype=torch.float32, device=dev)
        if extra_lookup_every and step % extra_lookup_every == extra_lookup_every - 1:
            xk = rng.integers(0, rows, size=n).astype(np.float32)
            want = model.lookup(xk.astype(np.uint64))
            gpu.embedding_lookup(torch.from_numpy(xk).to(dev), dest).wait()
            np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="extra lookup rows at step %d" % step)
        if isinstance(ahead, int) and not isinstance(ahead, bool) and step % ahead == 0:
            gpu.prefetch_keys_batch(kts[step:step + ahead])      # the batches of a block of steps, sorted in one launch
        want = model.lookup(fk.astype(np.uint64))
        kt = kts[step] if ahead else torch.from_numpy(fk).to(dev)
        gpu.embedding_lookup(kt, dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="lookup rows at step %d" % step)
        if ahead is True and step + 1 < steps:
            gpu.prefetch_keys(kts[step + 1])
        grads = (rng.standard_normal((n, width), dtype=np.float32) * np.float32(-0.01))
        if push_keys_mode:
            pk = np.unique(rng.choice(keys, size=max(1, n // 3)))
            model.update_with_push_keys(fk.astype(np.uint64), pk.astype(np.uint64), grads)
            gpu.embedding_update_with_push_keys(torch.from_numpy(fk).to(dev),
                                                torch.from_numpy(pk.astype(np.int64)).to(dev),
                                                torch.from_numpy(grads).to(dev)).wait()
        else:
            model.update(fk.astype(np.uint64), grads)
            if same:
                gpu.embedding_update(kt, torch.from_numpy(grads).to(dev), same_as_lookup=True).wait()
            else:
                gpu.embedding_update(torch.from_numpy(fk).to(dev), torch.from_numpy(grads).to(dev)).wait()
        # perf dicts (cache.cc:89-106,179-196)
        for got, exp in zip(gpu.perf[-2:], model.perf[-2:]):
            for f in ("type", "num_all", "num_unique", "num_miss", "num_transfered", "is_full"):
                assert got[f] == exp[f], (step, f, got, exp)
            if exp["type"] == "Push":
                assert got["num_evict"] == exp["num_evict"], (step, got, exp)
            # the stage times of the reference's dict (cache.cc:99-105,189-194), milliseconds from HIP events between the
            # call's launches: present, non-negative, and the stages add up to no more than the whole call
            stages = ("sort_time", "lookup_time", "prepare_time", "transfer_time", "copy_time", "insert_time") \
                if exp["type"] == "Pull" else ("sort_time", "lookup_time", "copy_time", "transfer_time", "cleanup_time")
            assert got["time"] > 0 and all(got[f] >= 0 for f in stages), (step, got)
            # (each stage is its own hipEventElapsedTime difference, rounded by itself: a microsecond of slack)
            assert sum(got[f] for f in stages) <= got["time"] * 1.01 + 1e-3, (step, got)
        np.testing.assert_array_equal(versions.cpu().numpy(), server.ver, err_msg="server versions step %d" % step)
        if step % check_every == 0 or step == steps - 1:
            np.testing.assert_array_equal(table.cpu().numpy(), server.table, err_msg="server table step %d" % step)
            _compare_state(gpu, model, step)
    assert gpu.size() == model.policy.size()
    np.testing.assert_array_equal(gpu.keys(), np.array(model.policy.keys(), dtype=np.uint64))
    return gpu, model


@pytest.mark.parametrize("pull_bound,push_bound", [(0, 0), (3, 3), (100, 100)])
def test_lru_trace_small(dev, pull_bound, push_bound):
    _run_trace(dev, limit=40, rows=400, width=8, n=64, steps=60, pull_bound=pull_bound, push_bound=push_bound)


def test_lru_trace_uniform_heavy_eviction(dev):
    _run_trace(dev, limit=16, rows=1000, width=4, n=48, steps=40, pull_bound=2, push_bound=2, zipf=False, seed=3)


def test_lru_limit_smaller_than_batch(dev):
    _run_trace(dev, limit=5, rows=200, width=4, n=40, steps=25, pull_bound=1, push_bound=1, zipf=False, seed=4)


def test_lru_trace_with_push_keys(dev):
    _run_trace(dev, limit=50, rows=500, width=8, n=80, steps=40, pull_bound=3, push_bound=3, push_keys_mode=True,
               seed=5)


def test_lru_trace_criteo_width(dev):
    _run_trace(dev, limit=300, rows=5000, width=128, n=416, steps=12, pull_bound=2, push_bound=2, seed=6,
               check_every=4)


def test_lru_trace_long_runs_full_width(dev):
    # one key repeats ~600 times per batch: the mapped accumulate takes the cooperative long-run path
    # (8 slices of 64 columns) for both the gradient buffer and the data rows
    _run_trace(dev, limit=500, rows=3000, width=512, n=2000, steps=4, pull_bound=1, push_bound=2, seed=9)


def _fused(gpu):
    return int(gpu._L.ha_cache_fused_updates(gpu._h))


@pytest.mark.parametrize("same,extra", [(True, 0), (False, 0), (True, 3)])
def test_lru_trace_with_the_next_batch_sorted_ahead(dev, same, extra):
    """prefetch_keys: the sort of the next lookup's keys runs on the cache's own stream beside the current update, into the
    second plan workspace -- the same trace, state and reports as the model (and as without it); with `extra`, another
    batch is looked up between a prefetch and its lookup."""
    gpu, _ = _run_trace(dev, limit=100, rows=1500, width=8, n=64, steps=60, pull_bound=2, push_bound=2, seed=21,
                        same=same, extra_lookup_every=extra, ahead=True)
    if same and not extra:
        assert _fused(gpu) == 60


@pytest.mark.parametrize("same,extra,blk", [(True, 0, 16), (False, 0, 5), (True, 3, 7), (True, 0, 1)])
def test_lru_trace_with_a_block_of_batches_sorted_ahead(dev, same, extra, blk):
    """prefetch_keys_batch: the sorts of the next `blk` lookups in ONE launch, into a ring of plan workspaces -- the same
    trace, state and reports as the model; with `extra`, a lookup that was not announced drops the rest of a block (those
    lookups sort by themselves)."""
    gpu, _ = _run_trace(dev, limit=100, rows=1500, width=8, n=64, steps=60, pull_bound=2, push_bound=2, seed=23,
                        same=same, extra_lookup_every=extra, ahead=blk)
    if same and not extra:
        assert _fused(gpu) == 60


def test_lru_trace_with_a_block_sorted_ahead_criteo_width_and_long_runs(dev):
    _run_trace(dev, limit=300, rows=5000, width=128, n=416, steps=12, pull_bound=2, push_bound=2, seed=24,
               check_every=4, ahead=4, same=True)
    _run_trace(dev, limit=2500, rows=6000, width=512, n=2000, steps=5, pull_bound=1, push_bound=2, seed=25, ahead=16,
               same=True)


def test_block_sort_ahead_insists_on_the_announced_order(dev):
    rows, width, n = 500, 8, 96
    rng = np.random.default_rng(6)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    gpu = hcache.LRUCache(200, rows, width, node_id=0, max_batch=n, device=dev)
    gpu.bind_store(torch.from_numpy(table0.copy()).to(dev), torch.zeros(rows, dtype=torch.int64, device=dev))
    ks = [torch.from_numpy(rng.integers(0, rows, size=n).astype(np.float32)).to(dev) for _ in range(3)]
    dest = torch.empty((n, width), device=dev)
    gpu.prefetch_keys_batch(ks)
    from herald_amd import _lib
    import ctypes
    with pytest.raises(Exception):      # the C entry point: batch 1 where batch 0 was announced
        _lib.check(gpu._L.ha_cache_lookup_presorted(gpu._h, ctypes.c_void_p(ks[1].data_ptr()), 0, n,
                                                    ctypes.c_void_p(dest.data_ptr()), None), "ha_cache_lookup_presorted")
    # the Python class: a lookup of another batch drops the block, everything still answers right
    for k in (ks[1], ks[0], ks[2]):
        gpu.embedding_lookup(k, dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), table0[k.cpu().numpy().astype(np.int64)])
    with pytest.raises(ValueError):
        gpu.prefetch_keys_batch(ks * 6)      # more than 16 batches


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
def test_lfu_large_cache_parallel_victim_scan(dev, policy):
    """A cache of 65,536+ slots, full, its lowest use bucket empty at the start of a lookup: the oldest line of the lowest
    non-empty bucket (lfu_cache.cc:31-42, lfuopt_cache.cc:48-60) is found by 512 workgroups in a launch of their own
    (cache_scan_victim_part_kernel) instead of one workgroup walking every line -- same trace as the model."""
    _run_trace(dev, limit=62000, rows=400000, width=4, n=1024, steps=84, pull_bound=2, push_bound=2, zipf=False, seed=31,
               policy=policy, check_every=42, same=False)


def test_lfu_trace_with_the_next_batch_sorted_ahead_criteo_width(dev):
    _run_trace(dev, limit=300, rows=5000, width=128, n=416, steps=10, pull_bound=2, push_bound=2, seed=22, policy="lfu",
               check_every=3, ahead=True)


def test_prefetched_keys_that_change_before_their_lookup_are_sorted_again(dev):
    """A torch-visible write to a prefetched key tensor voids the prefetch: the lookup sorts by itself."""
    rows, width, n = 500, 8, 96
    rng = np.random.default_rng(5)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    table = torch.from_numpy(table0.copy()).to(dev)
    gpu = hcache.LRUCache(200, rows, width, node_id=0, max_batch=n, device=dev)
    gpu.bind_store(table, torch.zeros(rows, dtype=torch.int64, device=dev))
    k = torch.from_numpy(rng.integers(0, rows, size=n).astype(np.float32)).to(dev)
    gpu.prefetch_keys(k)
    k.copy_(torch.from_numpy(rng.integers(0, rows, size=n).astype(np.float32)))
    dest = torch.empty((n, width), device=dev)
    gpu.embedding_lookup(k, dest).wait()
    np.testing.assert_array_equal(dest.cpu().numpy(), table0[k.cpu().numpy().astype(np.int64)])
    # ... and the unchanged tensor takes the presorted path, through the C entry point that insists on the match
    k2 = torch.from_numpy(rng.integers(0, rows, size=n).astype(np.float32)).to(dev)
    gpu.prefetch_keys(k2)
    gpu.embedding_lookup(k2, dest).wait()
    np.testing.assert_array_equal(dest.cpu().numpy(), table0[k2.cpu().numpy().astype(np.int64)])
    with pytest.raises(Exception):
        from herald_amd import _lib
        import ctypes
        _lib.check(gpu._L.ha_cache_lookup_presorted(gpu._h, ctypes.c_void_p(k2.data_ptr()), 0, n,
                                                    ctypes.c_void_p(dest.data_ptr()), None), "ha_cache_lookup_presorted")


@pytest.mark.parametrize("pull_bound,push_bound", [(0, 0), (3, 3), (100, 100)])
@pytest.mark.parametrize("zipf", [True, False])
def test_lru_trace_update_of_the_looked_up_keys(dev, pull_bound, push_bound, zipf):
    """limit >= batch and the update names the lookup's key tensor: accumulate + ONE launch for touch / bounded push /
    evicted lines / commit (cache_update_same_post_kernel) -- the same trace, state and reports as the model."""
    gpu, _ = _run_trace(dev, limit=100, rows=1500, width=8, n=64, steps=80, pull_bound=pull_bound,
                        push_bound=push_bound, zipf=zipf, seed=11, same=True)
    assert _fused(gpu) == 80


def test_lru_trace_update_of_the_looked_up_keys_with_lookups_between(dev):
    # every third step looks two batches up before its update: its evict list holds the victims of both lookups
    # (a key among them may be back in the batch), so that update takes the general path
    gpu, _ = _run_trace(dev, limit=80, rows=300, width=8, n=64, steps=60, pull_bound=1, push_bound=2, zipf=False,
                        seed=12, same=True, extra_lookup_every=3)
    assert _fused(gpu) == 40


def test_lru_trace_update_of_the_looked_up_keys_long_runs_full_width(dev):
    gpu, _ = _run_trace(dev, limit=2500, rows=6000, width=512, n=2000, steps=5, pull_bound=1, push_bound=2, seed=13,
                        same=True)
    assert _fused(gpu) == 5


def test_lru_trace_update_of_the_looked_up_keys_log_compaction(dev):
    gpu, _ = _run_trace(dev, limit=24, rows=96, width=4, n=16, steps=500, pull_bound=0, push_bound=1, seed=14,
                        check_every=50, same=True)
    assert _fused(gpu) == 500
    st = gpu.state()
    assert st["log_tail"] - st["log_head"] <= st["log_cap"]


@pytest.mark.parametrize("limit", [30000, 9000])
def test_lru_trace_many_finish_chunks(dev, limit):
    """20,000 keys per batch: the plan's finish is 20 workgroups that exchange their miss / pull / head counts inside the
    launch (cache_finish_book_kernel); limit >= batch also inserts there, limit < batch leaves the insert to the eviction
    workgroup."""
    gpu, _ = _run_trace(dev, limit=limit, rows=120000, width=4, n=20000, steps=4, pull_bound=1, push_bound=2, zipf=False,
                        seed=31, same=True, check_every=2)
    assert _fused(gpu) == (4 if limit >= 20000 else 0)


def test_lru_limit_smaller_than_batch_never_takes_the_fused_update(dev):
    gpu, _ = _run_trace(dev, limit=5, rows=200, width=4, n=40, steps=10, pull_bound=1, push_bound=1, zipf=False,
                        seed=15, same=True)
    assert _fused(gpu) == 0


def test_log_compaction_keeps_lru_order(dev):
    # a tiny cache driven for many steps forces the stamp log to wrap and compact
    gpu, _ = _run_trace(dev, limit=8, rows=64, width=4, n=16, steps=400, pull_bound=0, push_bound=0, seed=7,
                        check_every=50)
    st = gpu.state()
    assert st["log_tail"] - st["log_head"] <= st["log_cap"]


def test_cache_sparse_table_numpy_and_bounds(dev):
    rows, width = 300, 16
    rng = np.random.default_rng(8)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    table = torch.from_numpy(table0.copy()).to(dev)
    hcache.register_table(7, table)
    t = hcache.CacheSparseTable(limit=32, length=rows, width=width, node_id=7, policy="LRU", bound=0,
                                max_batch=128, device=dev)
    keys = rng.integers(0, rows, size=50).astype(np.uint64)
    dest = np.empty((50, width), dtype=np.float32)
    t.embedding_lookup(keys, dest, sync=True)
    np.testing.assert_array_equal(dest, table0[keys.astype(np.int64)])
    assert t.width == width and t.limit == 32
    with pytest.raises(NotImplementedError):
        hcache.CacheSparseTable(10, rows, width, 7, policy="fifo")
    assert "Cache" in repr(t.cache)


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
@pytest.mark.parametrize("pull_bound,push_bound", [(0, 0), (3, 3)])
def test_lfu_policies_trace(dev, policy, pull_bound, push_bound):
    _run_trace(dev, limit=40, rows=400, width=8, n=64, steps=60, pull_bound=pull_bound, push_bound=push_bound,
               policy=policy, seed=11)


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
def test_lfu_policies_heavy_eviction_and_small_limit(dev, policy):
    _run_trace(dev, limit=16, rows=1000, width=4, n=48, steps=40, pull_bound=2, push_bound=2, zipf=False, seed=12,
               policy=policy)
    _run_trace(dev, limit=5, rows=200, width=4, n=40, steps=25, pull_bound=1, push_bound=1, zipf=False, seed=13,
               policy=policy)


def test_lfuopt_store_fills_up(dev):
    # a hot working set smaller than the limit gets promoted to the never-evicted store; once the store
    # holds `limit` lines new keys are dropped (lfuopt_cache.cc:18-24)
    _run_trace(dev, limit=12, rows=60, width=4, n=40, steps=80, pull_bound=0, push_bound=0, zipf=True, seed=14,
               policy="lfuopt")


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
def test_lfu_with_push_keys(dev, policy):
    _run_trace(dev, limit=50, rows=500, width=8, n=80, steps=30, pull_bound=3, push_bound=3, push_keys_mode=True,
               seed=15, policy=policy)


@pytest.mark.parametrize("policy", ["lru", "lfu", "lfuopt"])
def test_push_pull_trace(dev, policy):
    """embedding_push_pull (ASP prefetch): push batch k while pulling batch k+1, step by step."""
    run_push_pull_trace(dev, policy)


def run_push_pull_trace(dev, policy, bind=None):
    rng = np.random.default_rng(21)
    rows, width, n, limit = 300, 8, 48, 30
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    server = cache_model.Server(table0)
    model = cache_model.CacheModel(policy, limit, width, server, 2, 2)
    table = torch.from_numpy(table0.copy()).to(dev)
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    cls = {"lru": hcache.LRUCache, "lfu": hcache.LFUCache, "lfuopt": hcache.LFUOptCache}[policy]
    gpu = cls(limit, rows, width, node_id=0, max_batch=64, device=dev)
    if bind is None:
        gpu.bind_store(table, versions)
    else:
        bind(gpu, table, versions)
    gpu.pull_bound = gpu.push_bound = 2
    batches = [((np.minimum(rng.zipf(1.3, size=n) - 1, rows - 1) * 31) % rows).astype(np.float32) for _ in range(40)]
    want = model.lookup(batches[0].astype(np.uint64))
    dest = torch.empty((n, width), dtype=torch.float32, device=dev)
    gpu.embedding_lookup(torch.from_numpy(batches[0]).to(dev), dest).wait()
    np.testing.assert_array_equal(dest.cpu().numpy(), want)
    for k in range(len(batches) - 1):
        grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
        want = model.push_pull(batches[k + 1].astype(np.uint64), batches[k].astype(np.uint64), grads)
        gpu.embedding_push_pull(torch.from_numpy(batches[k + 1]).to(dev), dest, torch.from_numpy(batches[k]).to(dev),
                                torch.from_numpy(grads).to(dev)).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="push_pull rows at step %d" % k)
        np.testing.assert_array_equal(versions.cpu().numpy(), server.ver)
        np.testing.assert_array_equal(table.cpu().numpy(), server.table)
        _compare_state(gpu, model, k)


def test_update_of_the_looked_up_key_tensor_reuses_the_plan(dev):
    """embedding_update(keys, same_as_lookup=True) right after embedding_lookup(keys) on the same device
    tensor goes through ha_cache_update_same_keys; the result equals the two-plan path.  The reuse is
    explicit: without the flag the keys are always sorted again, whatever the tensor's identity says."""
    rng = np.random.default_rng(77)
    rows, width, n = 4000, 64, 900
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    results = []
    for reuse in (True, False):
        table = torch.from_numpy(table0.copy()).to(dev)
        versions = torch.zeros(rows, dtype=torch.int64, device=dev)
        hcache.register_table(31 + int(reuse), table, versions)
        c = hcache.CacheSparseTable(300, rows, width, 31 + int(reuse), "LRU", bound=2, max_batch=n, device=dev)
        for step in range(6):
            ids = torch.from_numpy(((np.arange(n) * 7 + step * 13) % 500).astype(np.int64)).to(dev)
            grads = torch.from_numpy(np.full((n, width), 0.25 * (step + 1), dtype=np.float32)).to(dev)
            dest = torch.empty((n, width), dtype=torch.float32, device=dev)
            c.embedding_lookup(ids, dest, sync=True)
            assert (c.cache._last_lookup is not None)
            c.embedding_update(ids, grads, sync=True, same_as_lookup=reuse)
            assert c.cache._last_lookup is None
        torch.cuda.synchronize()
        results.append((table.cpu().numpy(), versions.cpu().numpy(), dest.cpu().numpy()))
    for a, b in zip(results[0], results[1]):
        np.testing.assert_array_equal(a, b)
    # without the flag a key tensor that was rewritten in place (no matter how) never meets a stale plan
    table = torch.from_numpy(table0.copy()).to(dev)
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    hcache.register_table(40, table, versions)
    c = hcache.CacheSparseTable(300, rows, width, 40, "LRU", bound=0, max_batch=n, device=dev)
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    dest = torch.empty((n, width), dtype=torch.float32, device=dev)
    c.embedding_lookup(ids, dest, sync=True)
    ids += 1000
    c.embedding_update(ids, torch.ones((n, width), dtype=torch.float32, device=dev), sync=True)
    with pytest.raises(ValueError):               # the flag without a preceding lookup of that tensor
        c.embedding_update(ids.clone(), torch.ones((n, width), dtype=torch.float32, device=dev), same_as_lookup=True)
    torch.cuda.synchronize()
    t = table.cpu().numpy()
    np.testing.assert_array_equal(t[:1000], table0[:1000])                  # untouched rows
    np.testing.assert_array_equal(t[1000:1000 + n], table0[1000:1000 + n] + 1.0)


def test_phase_clock_of_the_last_lookup(dev):
    """ha_cache_phase_times: the bookkeeping workgroup's stamps ascend, and the insert / eviction workgroup (beside the row
    copies) starts after it."""
    import ctypes
    gpu, _ = _run_trace(dev, limit=100, rows=1500, width=8, n=64, steps=6, pull_bound=1, push_bound=1, seed=21, same=True)
    ph = (ctypes.c_uint64 * 16)()
    hcache.check(gpu._L.ha_cache_phase_times(gpu._h, ph, ctypes.c_void_p(gpu._stream().cuda_stream)), "phase_times")
    assert 0 < ph[0] <= ph[1] <= ph[4] <= ph[8] <= ph[9] <= ph[10] <= ph[12]
    assert ph[14] >= 0 and ph[13] <= 64
