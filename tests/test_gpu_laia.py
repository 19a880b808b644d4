"""GPU parity of the laia scheduler against oracle/laia_model.py: the emitted [plan, dist] stream
must be identical element by element."""
import numpy as np
import pytest

from herald_amd import laia as hlaia
from oracle import laia_model

pytestmark = pytest.mark.gpu


def _samples(S, T, nkeys, seed):
    rng = np.random.default_rng(seed)
    card = np.maximum(2, (nkeys * rng.dirichlet(np.ones(T))).astype(np.int64))
    off = np.concatenate([[0], np.cumsum(card)[:-1]])
    cols = [off[j] + np.minimum(rng.zipf(1.3, size=S) - 1, card[j] - 1) for j in range(T)]
    return np.stack(cols, axis=1).astype(np.uint64), int(off[-1] + card[-1])


def _run_both(S, T, W, rank, mini_bs, batch_num, epochs, cache_size, nkeys, seed):
    samples, key_limit = _samples(S, T, nkeys, seed)
    want = laia_model.LaiaSchedulerModel(samples, epochs, mini_bs, batch_num, W, rank, cache_size).emit()
    s = hlaia.LaiaScheduler()
    s.start(samples, S, T, epochs, mini_bs, batch_num, W, rank, cache_size, 16, 24, key_limit=key_limit)
    got = []
    while True:
        item = s.pop()
        got.append(item)
        if item == [0]:
            break
    s.close()
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "stream element %d differs (%s)" % (k, "plan" if k % 2 == 0 else "dist")
    return got


@pytest.mark.parametrize("W,rank", [(1, 0), (4, 0), (4, 3), (8, 5)])
def test_laia_stream_matches_model(dev, W, rank):
    got = _run_both(S=2000, T=26, W=W, rank=rank, mini_bs=32, batch_num=6, epochs=2, cache_size=500,
                    nkeys=20000, seed=W * 10 + rank)
    # protocol: [plan, dist] x (batch_num*epochs + 1) then [0]   (laia_scheduler.cc:126-139,168)
    assert len(got) == 2 * (6 * 2 + 1) + 1
    assert all(len(d) == 32 for d in got[1:-1:2])


def test_laia_small_cache_and_wraparound(dev):
    # cache much smaller than a batch's unique rows; batches wrap around the sample array
    _run_both(S=300, T=8, W=3, rank=1, mini_bs=40, batch_num=5, epochs=3, cache_size=25, nkeys=400, seed=7)


def test_laia_python_glue_pairs_dist_with_next_plan(dev):
    samples, key_limit = _samples(1200, 10, 3000, 11)
    sched = hlaia.LAIAScheduler(samples.astype(np.float32), batch_size=20)
    sched.start(nrank=2, rank=0, cache_limit=200, dataset_num=1, epoch_num=2, key_limit=key_limit)
    model = laia_model.LaiaSchedulerModel(sched.sparse_data.astype(np.uint64), 2, sched.batch_size,
                                          sched.batch_num, 2, 0, 200).emit()
    # model stream: p0 d0 p1 d1 ... ; glue discards p0 and pairs d_b with p_{b+1}
    for b in range(5):
        assert sched.get_input_index(b) == model[2 * b + 1]
        assert sched.get_comm_plan(b) == model[2 * b + 2]
    sched.step_forward(0)
    sched.sched.close()


# ---- TopkScheduler (laia/src/topk_scheduler.cc) -------------------------------------------------------
def _drain(pop):
    got = []
    while True:
        item = pop()
        got.append(item)
        if item == [0]:
            return got


@pytest.mark.parametrize("dataset,T,top_k,W,rank,mini_bs,nt,cache", [
    ("criteo", 26, 20, 4, 1, 32, 4, 300), ("criteo", 26, 0, 8, 6, 16, 16, 300), ("avazu", 18, 17, 2, 0, 30, 5, 300),
    ("movie", 2, 2, 3, 2, 12, 1, 300), ("criteosearch", 17, 16, 4, 3, 24, 8, 300),
    # caches that hold a global batch of rows: the scheduler state lives on the DEVICE (stamp-log snapshots, one lane per
    # thread slice assigns, own-sample plans by bitmap compaction)
    ("criteo", 26, 20, 4, 1, 32, 4, 3400), ("criteo", 26, 0, 8, 6, 16, 16, 3500), ("avazu", 18, 17, 2, 0, 30, 5, 1100),
    ("criteosearch", 17, 16, 4, 3, 24, 8, 1700)])
def test_topk_stream_and_counters_match_model(dev, dataset, T, top_k, W, rank, mini_bs, nt, cache):
    samples, key_limit = _samples(1500, T, 9000, seed=W * 7 + T)
    model = laia_model.TopkSchedulerModel(samples, 2, mini_bs, 5, W, rank, cache, nt, dataset, top_k)
    want = model.emit()[rank]
    s = hlaia.TopkScheduler()
    s.start(samples, 1500, T, 2, mini_bs, 5, W, rank, cache, nt, dataset, top_k, key_limit=key_limit)
    got = _drain(s.pop)
    assert len(got) == len(want) == 2 * (5 * 2 + 1) + 1
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "stream element %d differs (%s)" % (k, "plan" if k % 2 == 0 else "dist")
    perf = s.report_cache_perf()
    s.close()
    assert perf["on_device"] == (1 if cache >= W * mini_bs * T else 0)
    assert perf["per_worker"] == {"miss_pull": model.miss_pull, "miss_push": model.miss_push,
                                  "update_pull": model.update_pull, "update_push": model.update_push}
    # every sample of a batch is assigned exactly once
    B = W * mini_bs
    m2 = laia_model.TopkSchedulerModel(samples, 1, mini_bs, 1, W, rank, 300, nt, dataset, top_k)
    _, dist = m2.get_dist(0)
    assert sorted(p for d in dist for p in d) == list(range(B))


def test_topk_rejects_quota_overflow(dev):
    samples, key_limit = _samples(400, 2, 300, seed=3)
    s = hlaia.TopkScheduler()
    # 3 threads do not divide mini_bs 16: threads 1 and 2 get 21 samples for 4 x 5 slots
    s.start(samples, 400, 2, 1, 16, 2, 4, 0, 50, 3, "movie", 2, key_limit=key_limit)
    with pytest.raises(Exception, match="quota"):
        _drain(s.pop)
    s.close()
    with pytest.raises(ValueError):
        laia_model.TopkSchedulerModel(samples, 1, 16, 2, 4, 0, 50, 3, "movie", 2)


def test_topk_local_shared_rings(dev):
    """local rank 0 schedules for the node's workers and hands each its stream through its ring."""
    T, W, mini_bs, nt = 18, 4, 20, 4
    samples, key_limit = _samples(1000, T, 5000, seed=21)
    model = laia_model.TopkSchedulerModel(samples, 1, mini_bs, 4, W, 2, 200, nt, "avazu", 17)
    want = model.emit(ranks=[2, 3])         # node of two workers whose first global rank is 2
    major = hlaia.TopkScheduler()
    major.start(samples, 1000, T, 1, mini_bs, 4, W, 2, 200, nt, "avazu", 17, True, 0, 2, key_limit=key_limit)
    minor = hlaia.TopkScheduler()
    minor.start(samples, 1000, T, 1, mini_bs, 4, W, 2, 200, nt, "avazu", 17, True, 1, 2, key_limit=key_limit)
    got1 = _drain(minor.pop_from_local_worker)
    got0 = _drain(major.pop_from_local_worker)
    minor.close()
    major.close()
    assert got0 == want[2]
    assert got1 == want[3]


# ---- device-resident scheduler state (cache_size >= global batch x tables) ----------------------------------------
def _run_device(monkeypatch, W, rank, T, mini_bs, batch_num, epochs, cache_size, nkeys, S, seed, env=None, host=False):
    """LaiaScheduler whose MiniLRU snapshots, greedy assignment and sorted-unique key lists live on the GPU
    (csrc/laia.hip, laia_next_device): stream, final snapshots and traffic counters against the sequential model."""
    import numpy as _np
    from herald_amd import _lib
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    if host:
        monkeypatch.setenv("HA_LAIA_HOST", "1")
    assert host or cache_size >= W * mini_bs * T
    samples, key_limit = _samples(S, T, nkeys, seed)
    model = laia_model.LaiaSchedulerModel(samples, epochs, mini_bs, batch_num, W, rank, cache_size)
    want = model.emit()
    s = hlaia.LaiaScheduler()
    s.start(samples, S, T, epochs, mini_bs, batch_num, W, rank, cache_size, 16, 24, key_limit=key_limit)
    got = []
    while True:
        item = s.pop()
        got.append(item)
        if item == [0]:
            break
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "stream element %d differs (%s)" % (k, "plan" if k % 2 == 0 else "dist")
    for w in range(W):
        assert s.snapshot_keys(w) == model.snaps[w].keys(), "valid resident keys of worker %d" % w
    cnt = _np.zeros(4 * W, dtype=_np.int64)
    _lib.check(s._L.ha_laia_counters(s._h, cnt.ctypes.data), "ha_laia_counters")
    s.close()
    return got, cnt.reshape(4, W)


@pytest.mark.parametrize("W,rank,T,mini_bs,cache_size,nkeys", [(4, 1, 6, 16, 400, 3000), (1, 0, 5, 32, 200, 900),
                                                               (8, 6, 4, 8, 300, 2500), (3, 2, 7, 20, 500, 1200),
                                                               (2, 0, 26, 64, 3400, 30000)])
def test_laia_device_resident_state_matches_model(dev, monkeypatch, W, rank, T, mini_bs, cache_size, nkeys):
    """Caches a few batches large: every batch evicts, lines near the LRU end are named again (evicted early and
    re-inserted), workers fill their quota at different samples; the counters equal the host mode's."""
    args = dict(W=W, rank=rank, T=T, mini_bs=mini_bs, batch_num=25, epochs=2, cache_size=cache_size, nkeys=nkeys, S=4000,
                seed=100 + W)
    _, cnt_dev = _run_device(monkeypatch, **args)
    _, cnt_host = _run_device(monkeypatch, host=True, **args)
    np.testing.assert_array_equal(cnt_dev, cnt_host)
    assert cnt_dev[0].sum() > 0 and (W == 1 or cnt_dev[3].sum() > 0)


@pytest.mark.parametrize("W,rank,T,mini_bs,cache_size,nkeys", [(4, 1, 6, 16, 400, 3000), (2, 0, 26, 64, 3400, 30000)])
def test_laia_one_batch_ahead_is_the_same_stream(dev, monkeypatch, W, rank, T, mini_bs, cache_size, nkeys):
    """HA_LAIA_AHEAD=1: the scheduler thread announces every next batch (ha_laia_hint_next) and the library enqueues it
    before it hands the current batch over -- across the epoch boundary too (the batch after an epoch's last is batch 0) and
    not beyond the stream's end: stream, final snapshots and counters are those of the plain run and of the model."""
    args = dict(W=W, rank=rank, T=T, mini_bs=mini_bs, batch_num=25, epochs=2, cache_size=cache_size, nkeys=nkeys, S=4000,
                seed=100 + W)
    got_a, cnt_a = _run_device(monkeypatch, env={"HA_LAIA_AHEAD": "1"}, **args)
    monkeypatch.delenv("HA_LAIA_AHEAD")
    got_p, cnt_p = _run_device(monkeypatch, **args)
    assert got_a == got_p
    np.testing.assert_array_equal(cnt_a, cnt_p)


def test_laia_an_announced_batch_must_follow(dev):
    """A hint moves the device state one batch ahead: a call for another batch than the announced one is an error."""
    from herald_amd import _lib
    samples, key_limit = _samples(2000, 6, 3000, 3)
    L = _lib.load()
    h = L.ha_laia_create(samples.ctypes.data, 2000, 6, 2, 400, key_limit, 32)
    assert h
    dist, plan, off = np.empty(32, np.int64), np.empty(32 * 6 * 2 + 16, np.uint64), np.empty(3, np.int64)
    call = lambda b: L.ha_laia_next(h, b, 16, dist.ctypes.data, plan.ctypes.data, plan.size, off.ctypes.data)
    assert L.ha_laia_hint_next(h, 1) == 0 and call(0) == 0       # batch 0 returned, batch 1 enqueued
    assert call(2) != 0 and b"announced" in L.ha_last_error()
    assert call(1) == 0 and call(2) == 0                          # the announced one, then on without hints
    L.ha_laia_destroy(h)


def test_laia_device_log_compaction_and_stamp_renumbering(dev, monkeypatch):
    """A short log (compacted every few batches) and an early wrap of the 32-bit stamp counter (live entries restamped
    1, 2, ...): the stream does not change."""
    args = dict(W=4, rank=3, T=6, mini_bs=16, batch_num=40, epochs=2, cache_size=400, nkeys=3000, S=5000, seed=5)
    _run_device(monkeypatch, env={"HA_LAIA_DEBUG_LOG": "2048"}, **args)
    _run_device(monkeypatch, env={"HA_LAIA_DEBUG_LOG": "2048", "HA_LAIA_DEBUG_WRAP": "3000"}, **args)
