"""BASELINE configs[3] shape for the laia path: global batch 4096 samples (mini batch 1024 x 4 workers),
26 tables, the full 33,762,577-row key space, cache_size = limit = 0.1 x rows -- the scheduler stream
against oracle/laia_model.py element by element, its cost per global batch (GPU part and host part), and
the chain LAIAScheduler -> LAIADataloader -> EmbeddingLookUp_Gradient(enable_push_index) ->
embedding_update_with_push_keys against oracle/cache_model.py."""
import numpy as np
import pytest
import torch

from herald_amd import cache as hcache
from herald_amd import hetu_ops
from herald_amd import laia as hlaia
from herald_amd import synth
from oracle import cache_model, laia_model

pytestmark = pytest.mark.gpu
ROWS = 33762577


def _criteo_samples(nsamples, rows=ROWS):
    per = 256
    parts = [synth.criteo_batch(per, step=5000 + s, rows=rows, nfields=26) for s in range((nsamples + per - 1) // per)]
    return np.concatenate(parts, axis=0)[:nsamples].astype(np.uint64)


def test_laia_stream_at_config_d_size_and_timing(dev):
    W, mini_bs, T, batch_num = 4, 1024, 26, 3
    cache_size = int(0.1 * ROWS)
    samples = _criteo_samples(W * mini_bs * batch_num + 1000)
    S = samples.shape[0]
    rank = 1
    want = laia_model.LaiaSchedulerModel(samples, 1, mini_bs, batch_num, W, rank, cache_size).emit()
    s = hlaia.LaiaScheduler()
    s.start(samples, S, T, 1, mini_bs, batch_num, W, rank, cache_size, 16, 24, key_limit=ROWS)
    got = []
    while True:
        item = s.pop()
        got.append(item)
        if item == [0]:
            break
    tm = s.timing()
    s.close()
    assert len(got) == len(want) == 2 * (batch_num + 1) + 1
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "stream element %d differs (%s)" % (k, "plan" if k % 2 == 0 else "dist")
    assert all(len(d) == mini_bs for d in got[1:-1:2])
    assert tm["batches"] == batch_num + 1 and tm["us_per_batch"] > 0
    print("laia config D: %.0f us per 4096-sample global batch (host assignment %.0f us, host snapshot %.0f us, "
          "GPU + transfers + waits %.0f us)" % (tm["us_per_batch"], tm["host_assign_us"], tm["host_snapshot_us"],
                                                tm["gpu_and_transfer_us"]))


def test_laia_dataloader_drives_update_with_push_keys(dev):
    """The chain of run_laia.py on one worker's side: ids and plan come from the LAIADataloader as a tuple,
    the gradient op turns them into IndexedSlices with push_indices, the cache pushes exactly those keys."""
    rows, width, W, rank, mini_bs = 2_000_000, 64, 4, 2, 1024
    limit = 200_000
    samples = _criteo_samples(W * mini_bs * 8, rows=rows)
    sched = hlaia.LAIAScheduler(samples.astype(np.float32), batch_size=mini_bs)
    sched.start(nrank=W, rank=rank, cache_limit=limit, dataset_num=1, epoch_num=1, key_limit=rows)
    assert sched.batch_size == mini_bs
    dl = hlaia.LAIADataloader(sched, 0, True, samples.astype(np.float32), mini_bs, name="train", device=dev)
    dl.init_states(rank, W)
    assert dl.get_cur_shape() == (mini_bs, 26)

    rng = np.random.default_rng(31)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    server = cache_model.Server(table0)
    model = cache_model.CacheModel("lru", limit, width, server, 3, 3)
    table = torch.from_numpy(table0.copy()).to(dev)
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    hcache.register_table(41, table, versions)
    cst = hcache.CacheSparseTable(limit, rows, width, 41, "LRU", bound=3, max_batch=mini_bs * 26, device=dev)
    cst.perf_enabled(True)
    grad_op = hetu_ops.EmbeddingLookUp_Gradient((rows, width), enable_push_index=True)
    # the model stream of this rank: p0 d0 p1 d1 ...; the glue pairs d_b with p_{b+1}
    stream = laia_model.LaiaSchedulerModel(sched.sparse_data.astype(np.uint64), 1, mini_bs, sched.batch_num, W,
                                           rank, limit).emit()
    for b in range(4):
        ids, plan = dl.get_arr()
        assert isinstance(ids, torch.Tensor) and tuple(ids.shape) == (mini_bs, 26) and ids.dtype == torch.float32
        np.testing.assert_array_equal(ids.cpu().numpy(), samples[stream[2 * b + 1]].astype(np.float32))
        np.testing.assert_array_equal(plan.cpu().numpy(), np.asarray(stream[2 * b + 2], dtype=np.float32))
        dest = torch.empty((mini_bs * 26, width), dtype=torch.float32, device=dev)
        cst.embedding_lookup((ids.reshape(-1), plan), dest).wait()        # cstable accepts the tuple (:38-45)
        keys = ids.cpu().numpy().reshape(-1).astype(np.uint64)
        np.testing.assert_array_equal(dest.cpu().numpy(), model.lookup(keys), err_msg="lookup, batch %d" % b)
        g = (rng.standard_normal((mini_bs * 26, width), dtype=np.float32) * np.float32(-0.01))
        slices = grad_op.compute(torch.from_numpy(g).to(dev), (ids, plan))
        assert slices.push_indices is plan
        cst.embedding_update_with_push_keys(slices.indices.reshape(-1), slices.push_indices, slices.values).wait()
        model.update_with_push_keys(keys, np.asarray(stream[2 * b + 2], dtype=np.uint64), g)
        for got, exp in zip(cst.perf[-2:], model.perf[-2:]):
            for f in ("type", "num_all", "num_unique", "num_miss", "num_transfered", "is_full"):
                assert got[f] == exp[f], (b, f, got, exp)
    np.testing.assert_array_equal(versions.cpu().numpy(), server.ver)
    np.testing.assert_array_equal(table.cpu().numpy(), server.table)
    sched.sched.close()
