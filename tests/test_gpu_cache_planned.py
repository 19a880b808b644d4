"""The PLANNED flow of the HET cache (csrc/cache_block.hip: ha_cache_plan_block / _lookup_planned / _update_planned) against
oracle/cache_model.py -- the same model, the same comparisons as tests/test_gpu_cache.py holds the call-by-call flow to:
lookup rows bit for bit and the perf dict's counts EVERY step, server table and versions every step, resident set / versions /
update counters / data and gradient rows of every line whenever no bookkeeping has run ahead of the rows (the end of a block
that was planned alone; the end of the stream otherwise).

Reference: CacheBase::_embeddingLookup / _embeddingUpdate (src/hetu_cache/src/cache.cc:60-107, 132-197), LRUCache
(src/hetu_cache/src/lru_cache.cc:5-39), Line (include/embedding.h:18-149), the server (ps-lite/src/PSFhandle_embedding.cc:5-64)."""
import numpy as np
import pytest
import torch

from herald_amd import cache as hcache
from oracle import cache_model
from test_gpu_cache import _compare_state

pytestmark = pytest.mark.gpu


def _draw(r, n, rows, zipf):
    if zipf:
        return (np.minimum(r.zipf(1.3, size=n) - 1, rows - 1).astype(np.int64) * 7919) % rows
    return r.integers(0, rows, size=n)


def _setup(dev, limit, rows, width, n, pull_bound, push_bound, seed, policy="lru"):
    rng = np.random.default_rng(seed)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    server = cache_model.Server(table0)
    model = cache_model.CacheModel(policy, limit, width, server, pull_bound, push_bound)
    table = torch.from_numpy(table0.copy()).to(dev)
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    cls = {"lru": hcache.LRUCache, "lfu": hcache.LFUCache, "lfuopt": hcache.LFUOptCache}[policy]
    gpu = cls(limit, rows, width, node_id=0, max_batch=max(n, 64), device=dev)
    gpu.bind_store(table, versions)
    gpu.pull_bound, gpu.push_bound = pull_bound, push_bound
    gpu.perf_enabled = True
    return rng, server, model, table, versions, gpu


def _check_perf(gpu, model, step):
    for got, exp in zip(gpu.perf[-2:], model.perf[-2:]):
        for f in ("type", "num_all", "num_unique", "num_miss", "num_transfered", "is_full"):
            assert got[f] == exp[f], (step, f, got, exp)
        if exp["type"] == "Push":
            assert got["num_evict"] == exp["num_evict"], (step, got, exp)


STATS = {"own_line_evicted": 0, "own_line_evicted_dirty": 0, "update_misses": 0}


def _planned_step(dev, gpu, model, keys, grads, width, step, versions, server):
    res = model.resident()
    held = {int(k): res[int(k)].updates for k in np.unique(keys) if int(k) in res}
    want = model.lookup(keys.astype(np.uint64))
    gone = [k for k in held if not model.policy.count(k)]        # lines of the batch its own lookup evicted (LFU policies)
    STATS["own_line_evicted"] += len(gone)
    STATS["own_line_evicted_dirty"] += sum(1 for k in gone if held[k] != 0)
    dest = torch.empty((keys.size, width), dtype=torch.float32, device=dev)
    gpu.embedding_lookup_planned(dest).wait()
    np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="lookup rows at step %d" % step)
    model.update(keys.astype(np.uint64), grads)
    gpu.embedding_update_planned(torch.from_numpy(grads).to(dev)).wait()
    STATS["update_misses"] += model.perf[-1]["num_miss"]
    _check_perf(gpu, model, step)
    np.testing.assert_array_equal(versions.cpu().numpy(), server.ver, err_msg="server versions step %d" % step)


def _run_planned(dev, limit, rows, width, n, steps, pull_bound, push_bound, block, seed=0, zipf=True, ahead=True,
                 dtype=np.float32, sizes=None, policy="lru", light=False):
    """ahead=True: block j + 1 is planned when block j starts (two blocks outstanding: its bookkeeping runs beside block j's
    rows); False: a block is planned when the one before is consumed, and the whole state is compared at every block end.
    light=True (the cases with 10^4 .. 10^5 lines): the server table once at the end, the resident SET at the end instead of
    every line's fields (lookup rows, perf counters and server versions still every step)."""
    rng, server, model, table, versions, gpu = _setup(dev, limit, rows, width, n, pull_bound, push_bound, seed, policy)
    sizes = sizes or [n] * steps
    keys_all = [_draw(rng, m, rows, zipf) for m in sizes]
    kts = [torch.from_numpy(k.astype(dtype)).to(dev) for k in keys_all]
    blocks = [list(range(b0, min(b0 + block, steps))) for b0 in range(0, steps, block)]
    if ahead:
        gpu.plan_block([kts[s] for s in blocks[0]])
    for j, blk in enumerate(blocks):
        if ahead and j + 1 < len(blocks):
            gpu.plan_block([kts[s] for s in blocks[j + 1]])
        elif not ahead:
            gpu.plan_block([kts[s] for s in blk])
        for step in blk:
            grads = rng.standard_normal((sizes[step], width), dtype=np.float32) * np.float32(-0.01)
            _planned_step(dev, gpu, model, keys_all[step], grads, width, step, versions, server)
            if not light:
                np.testing.assert_array_equal(table.cpu().numpy(), server.table, err_msg="server table step %d" % step)
        if not ahead or j + 1 == len(blocks):
            assert gpu.plan_pending() == 0
            if not light:
                _compare_state(gpu, model, blk[-1])
    if light:
        np.testing.assert_array_equal(table.cpu().numpy(), server.table, err_msg="server table at the end")
    assert gpu.size() == model.policy.size()
    np.testing.assert_array_equal(gpu.keys(), np.array(model.policy.keys(), dtype=np.uint64))
    return gpu, model


@pytest.mark.parametrize("pull_bound,push_bound", [(0, 0), (3, 3), (100, 100)])
@pytest.mark.parametrize("block,ahead", [(1, False), (4, False), (16, True), (5, True)])
def test_planned_lru_trace_small(dev, pull_bound, push_bound, block, ahead):
    _run_planned(dev, limit=100, rows=1500, width=8, n=64, steps=64, pull_bound=pull_bound, push_bound=push_bound, block=block,
                 seed=11, ahead=ahead)


@pytest.mark.parametrize("ahead", [False, True])
def test_planned_lru_uniform_heavy_eviction_at_limit_equal_batch(dev, ahead):
    # limit == max_batch: every lookup evicts almost as many lines as it brings in
    _run_planned(dev, limit=64, rows=1000, width=4, n=64, steps=48, pull_bound=2, push_bound=2, block=8, seed=3, zipf=False,
                 ahead=ahead)


def test_planned_lru_criteo_width_and_long_runs(dev):
    # width 128 with medium runs; width 512 with one key ~600 times per batch: the cooperative long-run path of the accumulate
    # takes the push epilogue (store row += gradient, gradient = 0) for both destinations' slices
    _run_planned(dev, limit=500, rows=5000, width=128, n=416, steps=12, pull_bound=2, push_bound=2, block=4, seed=24)
    _run_planned(dev, limit=2500, rows=6000, width=512, n=2000, steps=6, pull_bound=1, push_bound=2, block=3, seed=25)


def test_planned_lru_odd_width_takes_the_scalar_kernels(dev):
    _run_planned(dev, limit=120, rows=900, width=10, n=96, steps=20, pull_bound=1, push_bound=1, block=5, seed=8)


def test_planned_lru_many_keys_per_workgroup_slice(dev):
    """20,000 keys per batch: every bookkeeping workgroup owns several keys per thread, the eviction walk takes several
    rounds of 8,192 log entries."""
    _run_planned(dev, limit=30000, rows=120000, width=4, n=20000, steps=6, pull_bound=1, push_bound=2, block=3, seed=31,
                 zipf=False, ahead=True, light=True)


def test_planned_lru_large_cache_log_without_compaction(dev):
    """A cache large enough that the stamp log is NOT compacted in front of every batch (small caches are: their log is
    shorter than the walk's stride), full and evicting ~1,000 lines per lookup."""
    gpu, _ = _run_planned(dev, limit=40000, rows=400000, width=4, n=1024, steps=70, pull_bound=2, push_bound=2, block=16,
                          seed=33, zipf=False, ahead=True, light=True)
    st = gpu.state()
    assert st["size"] == 40000 and st["log_tail"] - st["log_head"] <= st["log_cap"]


def test_planned_lru_ragged_empty_batches_and_uint64_keys(dev):
    sizes = [64, 1, 0, 33, 64, 0, 0, 17, 64, 2]
    _run_planned(dev, limit=100, rows=700, width=8, n=64, steps=len(sizes), pull_bound=1, push_bound=1, block=4, seed=9,
                 dtype=np.int64, sizes=sizes)


def test_planned_and_call_by_call_flows_alternate(dev):
    """Call-by-call steps, a planned block, call-by-call steps, two planned blocks: one trace, the model's state throughout."""
    limit, rows, width, n = 100, 1500, 8, 64
    rng, server, model, table, versions, gpu = _setup(dev, limit, rows, width, n, 2, 2, seed=41)
    step = 0

    def classic(count):
        nonlocal step
        for _ in range(count):
            keys = _draw(rng, n, rows, True)
            fk = torch.from_numpy(keys.astype(np.float32)).to(dev)
            dest = torch.empty((n, width), dtype=torch.float32, device=dev)
            want = model.lookup(keys.astype(np.uint64))
            gpu.embedding_lookup(fk, dest).wait()
            np.testing.assert_array_equal(dest.cpu().numpy(), want)
            grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
            model.update(keys.astype(np.uint64), grads)
            gpu.embedding_update(fk, torch.from_numpy(grads).to(dev), same_as_lookup=bool(step % 2)).wait()
            _compare_state(gpu, model, step)
            step += 1

    def planned(count, blocks):
        nonlocal step
        ks = [[_draw(rng, n, rows, True) for _ in range(count)] for _ in range(blocks)]
        kt = [[torch.from_numpy(k.astype(np.float32)).to(dev) for k in blk] for blk in ks]
        gpu.plan_block(kt[0])
        for b in range(blocks):
            if b + 1 < blocks:
                gpu.plan_block(kt[b + 1])
            for k in ks[b]:
                grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
                _planned_step(dev, gpu, model, k, grads, width, step, versions, server)
                step += 1
        _compare_state(gpu, model, step)

    classic(7)
    planned(5, 1)
    classic(6)
    planned(4, 2)
    classic(3)
    np.testing.assert_array_equal(table.cpu().numpy(), server.table)


def test_planned_pull_decision_sees_what_another_worker_pushed_after_the_plan(dev):
    """The bookkeeping of a block runs before its rows; the staleness-bounded pull (cache.cc:84-93) must still be decided from
    the store's versions AS THE ROWS ARE READ: between two steps of a planned block another worker pushes to rows the cache
    holds (server versions + rows change) -- the next lookups pull exactly the lines the model pulls."""
    limit, rows, width, n = 200, 600, 8, 96
    rng, server, model, table, versions, gpu = _setup(dev, limit, rows, width, n, 2, 100, seed=51)
    keys_all = [_draw(rng, n, rows, True) for _ in range(8)]
    kts = [torch.from_numpy(k.astype(np.float32)).to(dev) for k in keys_all]
    gpu.plan_block(kts)
    torch.cuda.synchronize()            # the whole block is booked before a single row moves
    for step in range(8):
        if step in (2, 5):               # another worker's push: +4 updates on the rows the next batch names
            hot = np.unique(keys_all[step])[::2]
            delta = rng.standard_normal((hot.size, width), dtype=np.float32)
            server.ver[hot] += 4
            server.table[hot] = (server.table[hot] + delta).astype(np.float32)
            versions[torch.from_numpy(hot).to(dev)] += 4
            table[torch.from_numpy(hot).to(dev)] += torch.from_numpy(delta).to(dev)
        grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
        _planned_step(dev, gpu, model, keys_all[step], grads, width, step, versions, server)
    assert sum(r["num_transfered"] for r in gpu.perf if r["type"] == "Pull") > sum(r["num_miss"] for r in gpu.perf if r["type"] == "Pull")
    _compare_state(gpu, model, 8)
    np.testing.assert_array_equal(table.cpu().numpy(), server.table)


def test_planned_flow_refuses_misuse(dev):
    rows, width, n = 500, 8, 64
    rng = np.random.default_rng(6)
    table = torch.from_numpy(rng.standard_normal((rows, width), dtype=np.float32)).to(dev)
    ks = [torch.from_numpy(rng.integers(0, rows, size=n).astype(np.float32)).to(dev) for _ in range(3)]
    dest = torch.empty((n, width), device=dev)
    g = torch.zeros((n, width), device=dev)
    small = hcache.LRUCache(32, rows, width, node_id=0, max_batch=n, device=dev)       # limit < max_batch
    small.bind_store(table, torch.zeros(rows, dtype=torch.int64, device=dev))
    with pytest.raises(Exception, match="limit"):
        small.plan_block(ks[:1])
    lfu = hcache.LFUCache(200, rows, width, node_id=0, max_batch=n, device=dev)
    lfu.bind_store(table, torch.zeros(rows, dtype=torch.int64, device=dev))
    lfu.embedding_lookup(ks[0], dest).wait()
    with pytest.raises(Exception, match="update must follow"):
        lfu.plan_block(ks[:1])
    lfu.embedding_update(ks[1], g).wait()                  # an update of OTHER keys: the lookup's lines stay in the lowest use bucket
    with pytest.raises(Exception, match="lowest use bucket"):
        lfu.plan_block(ks[:1])
    lfu.embedding_update(ks[0], g).wait()
    lfu.plan_block(ks[:1])                                 # ... all of them updated: the planned flow takes over
    lfu.embedding_lookup_planned(dest)
    lfu.embedding_update_planned(g).wait()
    gpu = hcache.LRUCache(200, rows, width, node_id=0, max_batch=n, device=dev)
    gpu.bind_store(table, torch.zeros(rows, dtype=torch.int64, device=dev))
    with pytest.raises(ValueError):
        gpu.embedding_lookup_planned(dest)                 # nothing planned
    gpu.plan_block(ks[:2])
    with pytest.raises(ValueError):
        gpu.embedding_update_planned(g)                    # the lookup comes first
    with pytest.raises(Exception, match="planned"):
        gpu.embedding_lookup(ks[0], dest)                  # call-by-call while planned calls are outstanding
    with pytest.raises(Exception, match="planned"):
        gpu.pull_bound = 7
    gpu.plan_block(ks[2:])
    with pytest.raises(Exception, match="outstanding"):
        gpu.plan_block(ks[:1])                             # a third block
    for _ in range(3):
        gpu.embedding_lookup_planned(dest)
        gpu.embedding_update_planned(g)
    assert gpu.plan_pending() == 0
    gpu.embedding_lookup(ks[0], dest).wait()               # and the call-by-call flow is back
    with pytest.raises(ValueError):
        gpu.plan_block(ks * 6)                             # more than 16 batches


# ---- LFU / LFUOpt (cache_book_lfu_kernel): the same comparisons -----------------------------------------------------------------
@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
@pytest.mark.parametrize("pull_bound,push_bound", [(0, 0), (3, 3), (100, 100)])
@pytest.mark.parametrize("block,ahead", [(1, False), (16, True), (5, True)])
def test_planned_lfu_trace_small(dev, policy, pull_bound, push_bound, block, ahead):
    _run_planned(dev, limit=100, rows=1500, width=8, n=64, steps=64, pull_bound=pull_bound, push_bound=push_bound, block=block,
                 seed=12, ahead=ahead, policy=policy)


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
@pytest.mark.parametrize("limit", [1, 7, 40, 64])
def test_planned_lfu_cache_smaller_than_the_batch(dev, policy, limit):
    """limit < batch: most of a batch's inserts are evicted again by the batch's later inserts (the update finds no line for
    them: a line without data, pushed at once), and the one old line a full cache gives up is often a line of the batch
    itself (read by the lookup, gone for the update, its old gradient pushed BEHIND the batch's own)."""
    before = dict(STATS)
    _run_planned(dev, limit=limit, rows=300, width=8, n=64, steps=40, pull_bound=1, push_bound=3, block=4, seed=5 + limit,
                 ahead=False, policy=policy)
    _run_planned(dev, limit=limit, rows=300, width=8, n=64, steps=40, pull_bound=1, push_bound=0, block=16, seed=6 + limit,
                 zipf=False, ahead=True, policy=policy)
    assert STATS["update_misses"] > before["update_misses"] + 100
    assert STATS["own_line_evicted"] > before["own_line_evicted"], STATS      # (the traces hold the case: 3 .. 24 times)
    if limit in (1, 7):
        assert STATS["own_line_evicted_dirty"] > before["own_line_evicted_dirty"], STATS


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
def test_planned_lfu_hot_keys_reach_the_store_and_fill_it(dev, policy):
    """Few distinct keys, many steps: LFUOpt's lines reach use 10 and move to the never-evicted store until nothing else is
    left (every insert is dropped then, lfuopt_cache.cc:18-24); LFU's use counts grow without bound."""
    _run_planned(dev, limit=48, rows=90, width=4, n=64, steps=60, pull_bound=2, push_bound=2, block=8, seed=17, zipf=True,
                 ahead=True, policy=policy)
    _run_planned(dev, limit=30, rows=64, width=4, n=48, steps=50, pull_bound=0, push_bound=5, block=5, seed=18, zipf=False,
                 ahead=False, policy=policy)


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
def test_planned_lfu_widths_long_runs_ragged_batches(dev, policy):
    _run_planned(dev, limit=500, rows=5000, width=128, n=416, steps=12, pull_bound=2, push_bound=2, block=4, seed=24, policy=policy)
    _run_planned(dev, limit=900, rows=6000, width=512, n=2000, steps=6, pull_bound=1, push_bound=2, block=3, seed=25, policy=policy)
    _run_planned(dev, limit=120, rows=900, width=10, n=96, steps=20, pull_bound=1, push_bound=1, block=5, seed=8, policy=policy)
    sizes = [64, 1, 0, 33, 64, 0, 0, 17, 64, 2]
    _run_planned(dev, limit=100, rows=700, width=8, n=64, steps=len(sizes), pull_bound=1, push_bound=1, block=4, seed=9,
                 dtype=np.int64, sizes=sizes, policy=policy)


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
def test_planned_lfu_many_keys_and_a_large_cache(dev, policy):
    """20,000 keys per batch (several keys per bookkeeping thread); a cache of 40,000 lines (1,250 blocks of the victim tree),
    full, one line leaving per batch."""
    _run_planned(dev, limit=30000, rows=120000, width=4, n=20000, steps=6, pull_bound=1, push_bound=2, block=3, seed=31,
                 zipf=False, ahead=True, policy=policy, light=True)
    gpu, _ = _run_planned(dev, limit=40000, rows=400000, width=4, n=1024, steps=70, pull_bound=2, push_bound=2, block=16,
                          seed=33, zipf=False, ahead=True, policy=policy, light=True)
    assert gpu.state()["size"] == 40000
    # the bench's batch (6,656 keys, every bookkeeping workgroup busy) on a cache of 100,000 lines, full after 16 steps: the
    # answer to "which line leaves" is settled before the fastest workgroups start rewriting lines
    if policy == "lfu":
        gpu, _ = _run_planned(dev, limit=100000, rows=1000000, width=4, n=6656, steps=40, pull_bound=2, push_bound=2, block=16,
                              seed=34, zipf=False, ahead=True, policy=policy, light=True)
        assert gpu.state()["size"] == 100000


@pytest.mark.parametrize("policy", ["lfu", "lfuopt"])
def test_planned_lfu_and_call_by_call_flows_alternate(dev, policy):
    """Call-by-call pairs, a planned block, call-by-call pairs, two planned blocks: the victim tree is rebuilt whenever the
    planned flow takes over."""
    limit, rows, width, n = 100, 1500, 8, 64
    rng, server, model, table, versions, gpu = _setup(dev, limit, rows, width, n, 2, 2, seed=43, policy=policy)
    step = 0

    def classic(count):
        nonlocal step
        for _ in range(count):
            keys = _draw(rng, n, rows, True)
            fk = torch.from_numpy(keys.astype(np.float32)).to(dev)
            dest = torch.empty((n, width), dtype=torch.float32, device=dev)
            want = model.lookup(keys.astype(np.uint64))
            gpu.embedding_lookup(fk, dest).wait()
            np.testing.assert_array_equal(dest.cpu().numpy(), want)
            grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
            model.update(keys.astype(np.uint64), grads)
            gpu.embedding_update(fk, torch.from_numpy(grads).to(dev), same_as_lookup=bool(step % 2)).wait()
            _compare_state(gpu, model, step)
            step += 1

    def planned(count, blocks):
        nonlocal step
        ks = [[_draw(rng, n, rows, True) for _ in range(count)] for _ in range(blocks)]
        kt = [[torch.from_numpy(k.astype(np.float32)).to(dev) for k in blk] for blk in ks]
        gpu.plan_block(kt[0])
        for b in range(blocks):
            if b + 1 < blocks:
                gpu.plan_block(kt[b + 1])
            for k in ks[b]:
                grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
                _planned_step(dev, gpu, model, k, grads, width, step, versions, server)
                step += 1
        _compare_state(gpu, model, step)

    classic(7)
    planned(5, 1)
    classic(6)
    planned(4, 2)
    classic(3)
    np.testing.assert_array_equal(table.cpu().numpy(), server.table)


@pytest.mark.parametrize("policy", ["lru", "lfu"])
def test_planned_flow_over_a_store_in_pinned_host_memory(dev, policy):
    """The cold tier addressed directly: the store's rows in PINNED HOST memory (device-visible), its versions in HBM; the
    planned lookups pull over PCIe, the updates push (read-modify-write) over PCIe -- the model's rows, counters, versions."""
    limit, rows, width, n = 400, 20000, 64, 352
    rng = np.random.default_rng(61)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    server = cache_model.Server(table0)
    model = cache_model.CacheModel(policy, limit, width, server, 2, 2)
    table = torch.from_numpy(table0.copy()).pin_memory()
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    cls = {"lru": hcache.LRUCache, "lfu": hcache.LFUCache}[policy]
    gpu = cls(limit, rows, width, node_id=0, max_batch=n, device=dev)
    gpu.bind_store(table, versions)
    gpu.pull_bound, gpu.push_bound = 2, 2
    gpu.perf_enabled = True
    keys_all = [_draw(rng, n, rows, True) for _ in range(24)]
    kts = [torch.from_numpy(k.astype(np.float32)).to(dev) for k in keys_all]
    blocks = [list(range(b0, b0 + 8)) for b0 in range(0, 24, 8)]
    gpu.plan_block([kts[s] for s in blocks[0]])
    for j, blk in enumerate(blocks):
        if j + 1 < len(blocks):
            gpu.plan_block([kts[s] for s in blocks[j + 1]])
        for step in blk:
            grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(-0.01)
            _planned_step(dev, gpu, model, keys_all[step], grads, width, step, versions, server)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(table.numpy(), server.table, err_msg="the host table after the pushes")
    _compare_state(gpu, model, 24)


def test_pick_side_stream_returns_a_cached_stream_of_the_priority(dev):
    """herald_amd.streams.pick_side_stream: a stream of the asked priority beside the given main stream, the same object on the
    next call (the measurement runs once per (device, main stream, priority)); HA_STREAM_CALIBRATE=0 skips the measurement."""
    from herald_amd import streams
    main = torch.cuda.Stream(device=dev)
    a = streams.pick_side_stream(main, priority=-1)
    assert isinstance(a, torch.cuda.Stream) and a.priority == -1 and a.cuda_stream != main.cuda_stream
    assert streams.pick_side_stream(main, priority=-1) is a
    b = streams.pick_side_stream(main, priority=0)
    assert b is not a and b.priority == 0
    x = torch.zeros(1024, device=dev)
    with torch.cuda.stream(a):
        x.add_(1)
    a.synchronize()
    assert float(x.sum()) == 1024.0


@pytest.mark.parametrize("policy", ["lru", "lfuopt"])
def test_first_planned_block_of_fresh_caches_reads_written_plans(dev, policy):
    """Regression: the zero fill of a fresh plan workspace's header (a null-stream launch the host does not wait for) landed
    after the first plan build on the non-blocking planning stream and wiped n_unique -- the bookkeeping of a NEW cache's first
    block saw an empty batch, the row launches read items nobody wrote (about one new cache in eight; docs/EXPERIMENTS.md
    round 6 section 14).  Many fresh caches, each planning its first block at once: every lookup returns the store's rows and the
    sticky word stays clear.  (At this size the race did not show with the wait removed -- bench.py's cache tier at 1 M rows
    did, three runs in 28; this test holds the first block of a fresh cache, the wait is held by code review.)"""
    limit, rows, width, n = 5000, 200000, 32, 1664
    rng = np.random.default_rng(7)
    table = torch.from_numpy(rng.standard_normal((rows, width), dtype=np.float32)).to(dev)
    cls = {"lru": hcache.LRUCache, "lfuopt": hcache.LFUOptCache}[policy]
    keys = [torch.from_numpy(_draw(rng, n, rows, True).astype(np.float32)).to(dev) for _ in range(4)]
    out = torch.empty((n, width), device=dev)
    grad = torch.zeros((n, width), device=dev)
    torch.cuda.synchronize()
    for trial in range(12):
        versions = torch.zeros(rows, dtype=torch.int64, device=dev)
        gpu = cls(limit, rows, width, node_id=0, max_batch=n, device=dev)
        gpu.bind_store(table, versions)
        gpu.plan_block(keys)
        for k in keys:
            gpu.embedding_lookup_planned(out)
            want = table[k.to(torch.int64)]
            assert torch.equal(out, want), "trial %d: a planned lookup of a fresh cache" % trial
            gpu.embedding_update_planned(grad)
        st = gpu.state()          # (raises when a launch left its sticky word)
        assert 0 < st["size"] <= limit
        del gpu
