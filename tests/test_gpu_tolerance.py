"""Tolerance mode (ha_set_tolerance_mode, include/herald_amd.h): runs of 64 or more occurrences of a key are applied as
row - tree_sum(lr * g) in a FIXED order instead of the reference's serial chain (cpu_SGDOptimizerSparseUpdate,
src/dnnl_ops/Optimizers.cpp:65-72).  Held three ways: bit for bit against the numpy restatement of the tree
(oracle/qstep_model.py, long_min=None / coop_min=64), within BASELINE.json's 1e-5 relative of the accumulated gradient
against the serial chain (oracle/cpu.py), and -- mode off -- bit for bit against the serial chain as before."""
import numpy as np
import pytest
import torch

from herald_amd import ops, synth
from oracle import cpu, qstep_model

pytestmark = pytest.mark.gpu
LR = 0.05


@pytest.fixture()
def tolerance():
    prev = ops.set_tolerance_mode(True)
    yield
    ops.set_tolerance_mode(prev)


def _batch(rows, n, runs, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, rows, size=n).astype(np.float32)
    at = 0
    perm = rng.permutation(n)
    for k, c in enumerate(runs):                 # runs of c occurrences scattered over the batch
        ids[perm[at:at + c]] = np.float32((k * 7919) % rows)
        at += c
    return ids


def _check(table0, got, ids, grads, lr, mode="sgd", chunked=False):
    model = qstep_model.sgd_sparse_update(table0.copy(), ids, grads, lr, long_min=None, coop_min=64, mode=mode, chunked=chunked)
    np.testing.assert_array_equal(got, model)                                   # the tree, bit for bit
    exact = qstep_model.sgd_sparse_update(table0.copy(), ids, grads, lr, long_min=None, coop_min=1 << 30, mode=mode)
    tol = qstep_model.tolerance(ids, grads, lr if mode == "sgd" else 1.0, table0.shape[0], tree_min=64)
    differs = 0
    for k, t in tol.items():
        d = np.abs(got[k].astype(np.float64) - exact[k].astype(np.float64))
        assert np.all(d <= t + 1e-30), (k, d.max(), t.max())
        differs += int(d.max() > 0)
    return differs


@pytest.mark.parametrize("rows,width,n,runs", [(5000, 512, 6656, (64, 65, 200, 700, 2000, 63, 48)), (5000, 128, 6656, (256, 257, 1000)),
                                               (3000, 64, 3000, (100, 1500)), (800, 32, 900, (64, 300)),
                                               (20000, 128, 30000, (5000, 70, 64))])
@pytest.mark.parametrize("entry", ["apply", "apply_finish"])
def test_small_batches_tree_from_64(dev, tolerance, rows, width, n, runs, entry):
    """n <= 36,864 (one wave per sorted position; long runs by their full workgroups): ha_sgd_apply and the fused
    ha_sgd_apply_finish."""
    ids = _batch(rows, n, runs, seed=n + width)
    rng = np.random.default_rng(1)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    t = torch.from_numpy(table.copy()).to(dev)
    d_ids, d_g = torch.from_numpy(ids).to(dev), torch.from_numpy(grads).to(dev)
    if entry == "apply":
        plan = ops.IndexPlan(n, dev).build(d_ids)
        ops.sgd_apply(t, plan, d_g, LR)
    else:
        plan = ops.IndexPlan(n, dev).sort(d_ids)
        ops.sgd_apply_finish(t, plan, d_g, LR)
    torch.cuda.synchronize()
    assert _check(table, t.cpu().numpy(), ids, grads, LR) >= 1       # the tree really ran: some row differs from the chain


@pytest.fixture()
def tolerance_chunked():
    prev = ops.set_tolerance_mode(2)
    yield
    ops.set_tolerance_mode(prev)


@pytest.mark.parametrize("width", [128, 64])
def test_large_batch_by_unique_key_one_tree_per_run(dev, tolerance, width):
    """Mode 1 at BASELINE configs[2]'s per-GPU shape: one tree over every long run."""
    rows = 200000
    ids = (synth.criteo_batch(4096, 5).reshape(-1) % rows).astype(np.float32)
    n = ids.size
    rng = np.random.default_rng(2)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    t = torch.from_numpy(table.copy()).to(dev)
    plan = ops.IndexPlan(n, dev).sort(torch.from_numpy(ids).to(dev))
    ops.sgd_apply_finish(t, plan, torch.from_numpy(grads).to(dev), LR)
    torch.cuda.synchronize()
    assert _check(table, t.cpu().numpy(), ids, grads, LR) >= 1


@pytest.mark.parametrize("width", [128, 64, 256, 512])
def test_large_batch_by_unique_key(dev, tolerance_chunked, width):
    """Mode 2 at BASELINE configs[2]'s per-GPU shape (106,496 ids, 2,250-occurrence runs of the 3-category field, one key 4,096
    times): the by-unique apply of a finished plan.  Rows of up to 256 floats: runs beyond 256 occurrences are cut into chunks of 256 that
    workgroups of their own sum (oracle: tree_coop_chunked); 512: one tree over the whole run, as before."""
    rows = 200000
    ids = (synth.criteo_batch(4096, 5).reshape(-1) % rows).astype(np.float32)
    n = ids.size
    rng = np.random.default_rng(2)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    t = torch.from_numpy(table.copy()).to(dev)
    plan = ops.IndexPlan(n, dev).sort(torch.from_numpy(ids).to(dev))
    ops.sgd_apply_finish(t, plan, torch.from_numpy(grads).to(dev), LR)
    torch.cuda.synchronize()
    chunked = qstep_model.listed_chunking(ids, width)
    assert chunked == (width <= 256)
    assert _check(table, t.cpu().numpy(), ids, grads, LR, chunked=chunked) >= 1
    # the same plan applied again (the chunk counters start from zero in every launch), and the push / reduce forms
    t2 = torch.from_numpy(table.copy()).to(dev)
    ops.sgd_apply(t2, plan, torch.from_numpy(grads).to(dev), LR, finished=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t2.cpu().numpy(), t.cpu().numpy())


@pytest.mark.parametrize("n,width,runs", [(40000, 128, (257, 256, 513, 3000, 64, 48, 47)), (50000, 32, (9000, 300)),
                                          (37000, 256, (1024, 1025)), (60000, 128, ()),
                                          # medium runs (4..47: two columns per lane, 128 per pass) on rows that are not a
                                          # multiple of a pass
                                          (40000, 192, (47, 20, 9, 5, 4)), (38000, 96, (30, 4, 46))])
def test_chunked_runs_of_a_finished_plan(dev, tolerance_chunked, n, width, runs):
    """The chunking rule at its edges (256 / 257 occurrences, a partial last chunk, no long run at all), sgd and push."""
    rows = 90000
    ids = _batch(rows, n, runs, seed=n)
    rng = np.random.default_rng(4)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    d_ids, d_g = torch.from_numpy(ids).to(dev), torch.from_numpy(grads).to(dev)
    chunked = qstep_model.listed_chunking(ids, width)
    assert chunked == any(r > 256 for r in runs)
    plan = ops.IndexPlan(n, dev).build(d_ids)
    t = torch.from_numpy(table.copy()).to(dev)
    ops.sgd_apply(t, plan, d_g, LR, finished=True)
    torch.cuda.synchronize()
    _check(table, t.cpu().numpy(), ids, grads, LR, chunked=chunked)


def test_push_and_reduce_modes(dev, tolerance):
    """ha_push_apply (row + tree_sum(g)) and ha_dedup_reduce_scaled (0 + tree_sum(scale * g)) in tolerance mode."""
    rows, width, n = 4000, 128, 5000
    ids = _batch(rows, n, (64, 500, 90), seed=9)
    rng = np.random.default_rng(3)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    d_ids, d_g = torch.from_numpy(ids).to(dev), torch.from_numpy(grads).to(dev)
    plan = ops.IndexPlan(n, dev).build(d_ids)
    t = torch.from_numpy(table.copy()).to(dev)
    ops.push_apply(t, plan, d_g)
    torch.cuda.synchronize()
    _check(table, t.cpu().numpy(), ids, grads, 1.0, mode="push")
    red = ops.dedup_reduce(plan, d_g, scale=-LR)
    torch.cuda.synchronize()
    uniq = np.unique(ids.astype(np.int64))
    scaled = (grads * np.float32(-LR)).astype(np.float32)
    model = qstep_model.sgd_sparse_update(np.zeros((rows, width), np.float32), ids, scaled, 1.0, long_min=None, coop_min=64,
                                          mode="push")
    np.testing.assert_array_equal(red[:uniq.size].cpu().numpy(), model[uniq])


def test_mode_off_is_the_serial_chain(dev):
    """Default: the same batches are bit-exact to the reference's serial chain."""
    assert ops.set_tolerance_mode(False) is False
    rows, width, n = 5000, 128, 6656
    ids = _batch(rows, n, (64, 200, 2000), seed=11)
    rng = np.random.default_rng(4)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    t = torch.from_numpy(table.copy()).to(dev)
    plan = ops.IndexPlan(n, dev).sort(torch.from_numpy(ids).to(dev))
    ops.sgd_apply_finish(t, plan, torch.from_numpy(grads).to(dev), LR)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t.cpu().numpy(), cpu.sgd_sparse_update(table.copy(), ids, grads, LR))
