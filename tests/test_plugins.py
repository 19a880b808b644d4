"""pybind11 plugin modules `hetu_cache` / `laia_cache`: the reference's plugin surface
(src/hetu_cache/src/python_api.cc:12-79, laia/src/python_binding.cc:8-23) over the C-ABI."""
import os
import sys

import numpy as np
import pytest

PLUG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "herald_amd", "plugins")


def _import():
    import torch  # noqa: F401  one HIP runtime for torch and the plugins
    if PLUG not in sys.path:
        sys.path.insert(0, PLUG)
    import hetu_cache
    import laia_cache
    return hetu_cache, laia_cache


def test_plugin_surface_matches_reference():
    hetu_cache, laia_cache = _import()
    for cls in ("LRUCache", "LFUCache", "LFUOptCache"):
        c = getattr(hetu_cache, cls)
        for name in ("limit", "width", "perf", "pull_bound", "push_bound", "perf_enabled", "bypass", "undo_bypass",
                     "embedding_lookup", "embedding_update", "embedding_lookup_raw", "embedding_update_raw",
                     "embedding_push_pull_raw", "embedding_update_with_push_keys",
                     "embedding_update_with_push_keys_np_raw", "embedding_update_with_push_keys_raw", "count", "lookup",
                     "insert", "size", "keys", "__repr__"):
            assert hasattr(c, name), (cls, name)
    assert hasattr(hetu_cache, "Embedding") and hasattr(hetu_cache, "_waittype") and hasattr(hetu_cache, "debug")
    for name in ("start", "pop", "length"):
        assert hasattr(laia_cache.LaiaScheduler, name)
    for name in ("start", "pop", "pop_from_local_worker", "length"):
        assert hasattr(laia_cache.TopkScheduler, name)


@pytest.mark.gpu
def test_hetu_cache_plugin_against_model(dev):
    import torch
    from oracle import cache_model
    hetu_cache, _ = _import()
    rng = np.random.default_rng(31)
    rows, width, n, limit = 400, 16, 64, 40
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    table = torch.from_numpy(table0.copy()).to(dev)
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    hetu_cache.register_table(3, table.data_ptr(), versions.data_ptr(), rows, 0)
    cache = hetu_cache.LRUCache(limit, rows, width, 3)
    cache.pull_bound = 1
    cache.push_bound = 1
    cache.perf_enabled = True
    server = cache_model.Server(table0)
    model = cache_model.CacheModel("lru", limit, width, server, 1, 1)
    for step in range(25):
        keys = ((np.minimum(rng.zipf(1.3, size=n) - 1, rows - 1) * 17) % rows).astype(np.uint64)
        dest = np.empty((n, width), dtype=np.float32)
        cache.embedding_lookup(keys, dest).wait()                      # numpy (host) entry point
        np.testing.assert_array_equal(dest, model.lookup(keys))
        grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
        if step % 2 == 0:
            cache.embedding_update(keys, grads).wait()
        else:                                                           # raw entry point, device float32 keys
            kf = torch.from_numpy(keys.astype(np.float32)).to(dev)
            g = torch.from_numpy(grads).to(dev)
            cache.embedding_update_raw(kf.data_ptr(), g.data_ptr(), n).wait()
        model.update(keys, grads)
        assert cache.perf[-1]["num_transfered"] == model.perf[-1]["num_transfered"]
        assert cache.perf[-2]["num_miss"] == model.perf[-2]["num_miss"]
    torch.cuda.synchronize()
    np.testing.assert_array_equal(table.cpu().numpy(), server.table)
    np.testing.assert_array_equal(versions.cpu().numpy(), server.ver)
    np.testing.assert_array_equal(cache.keys(), np.array(model.policy.keys(), dtype=np.uint64))
    assert cache.size() == model.policy.size() and cache.limit == limit and cache.width == width
    k0 = int(model.policy.keys()[0])
    e = cache.lookup(k0)
    np.testing.assert_array_equal(e.data, model.resident()[k0].data)
    assert e.version == model.resident()[k0].version and cache.count(k0) == 1 and cache.lookup(10 ** 6) is None
    assert "Cache" in repr(cache) and "hetu.Embedding" in repr(e)
    with pytest.raises(RuntimeError):
        cache.embedding_lookup(np.arange(8, dtype=np.uint64)[::2], np.empty((4, width), dtype=np.float32))


@pytest.mark.gpu
def test_laia_cache_plugin_against_model(dev):
    from oracle import laia_model
    _, laia_cache = _import()
    rng = np.random.default_rng(32)
    S, T, W, rank, mini_bs, batch_num, epochs, cache_size = 1500, 12, 4, 2, 24, 5, 2, 300
    samples = np.stack([j * 500 + np.minimum(rng.zipf(1.3, size=S) - 1, 499) for j in range(T)], axis=1).astype(np.uint64)
    want = laia_model.LaiaSchedulerModel(samples, epochs, mini_bs, batch_num, W, rank, cache_size).emit()
    s = laia_cache.LaiaScheduler()
    s.start(samples.astype(np.intc), S, T, epochs, mini_bs, batch_num, W, rank, cache_size, 16, 24)   # intc as the
    got = []                                                                                          # reference passes
    while True:
        item = s.pop()
        got.append(list(item))
        if got[-1] == [0]:
            break
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("ahead", ["0", "1"])
def test_laia_cache_plugin_device_state_one_batch_ahead(dev, monkeypatch, ahead):
    """A cache that holds a global batch of rows (the scheduler state lives on the device), with and without the launch
    loop announcing its next batch (HA_LAIA_AHEAD, ha_laia_hint_next): the model's stream either way."""
    from oracle import laia_model
    _, laia_cache = _import()
    monkeypatch.setenv("HA_LAIA_AHEAD", ahead)
    rng = np.random.default_rng(34)
    S, T, W, rank, mini_bs, batch_num, epochs, cache_size = 1500, 12, 4, 2, 24, 7, 2, 1400
    samples = np.stack([j * 500 + np.minimum(rng.zipf(1.3, size=S) - 1, 499) for j in range(T)], axis=1).astype(np.uint64)
    want = laia_model.LaiaSchedulerModel(samples, epochs, mini_bs, batch_num, W, rank, cache_size).emit()
    s = laia_cache.LaiaScheduler()
    s.start(samples.astype(np.intc), S, T, epochs, mini_bs, batch_num, W, rank, cache_size, 16, 24)
    got = []
    while True:
        got.append(list(s.pop()))
        if got[-1] == [0]:
            break
    assert got == want


@pytest.mark.gpu
def test_laia_cache_plugin_topk_against_model(dev):
    from oracle import laia_model
    _, laia_cache = _import()
    rng = np.random.default_rng(33)
    S, T, W, mini_bs, batch_num, epochs, cache_size, nt = 1200, 26, 4, 16, 4, 1, 250, 8
    samples = np.stack([j * 300 + np.minimum(rng.zipf(1.3, size=S) - 1, 299) for j in range(T)], axis=1).astype(np.uint64)
    # standalone queue mode
    want = laia_model.TopkSchedulerModel(samples, epochs, mini_bs, batch_num, W, 1, cache_size, nt, "criteo", 20).emit()[1]
    s = laia_cache.TopkScheduler()
    s.start(samples.astype(np.intc), S, T, epochs, mini_bs, batch_num, W, 1, cache_size, nt, "criteo", 20, False, 0, 1)
    got = []
    while True:
        got.append(list(s.pop()))
        if got[-1] == [0]:
            break
    assert got == want
    # local-shared mode: local rank 0 (global rank 2) feeds the rings of the node's two workers
    want = laia_model.TopkSchedulerModel(samples, epochs, mini_bs, batch_num, W, 2, cache_size, nt, "criteo", 20).emit(ranks=[2, 3])
    major, minor = laia_cache.TopkScheduler(), laia_cache.TopkScheduler()
    major.start(samples.astype(np.intc), S, T, epochs, mini_bs, batch_num, W, 2, cache_size, nt, "criteo", 20, True, 0, 2)
    minor.start(samples.astype(np.intc), S, T, epochs, mini_bs, batch_num, W, 2, cache_size, nt, "criteo", 20, True, 1, 2)
    for sched, rank in ((minor, 3), (major, 2)):
        got = []
        while True:
            got.append(list(sched.pop_from_local_worker()))
            if got[-1] == [0]:
                break
        assert got == want[rank]
    with pytest.raises(RuntimeError, match="dataset not supported"):
        laia_cache.TopkScheduler().start(samples.astype(np.intc), S, T, 1, mini_bs, 1, W, 0, 10, 1, "nope", 2, False, 0, 1)
