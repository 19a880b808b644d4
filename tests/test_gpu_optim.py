"""Sparse optimizers (AdaGrad / Adam / AdamW) on deduplicated slices: scenario of the reference's
tests/test_optimizer.py:117-198 (500x400 table, 100 random duplicated ids, numpy oracle, atol 1e-5)."""
import ctypes

import numpy as np
import pytest
import torch

from herald_amd import ops
from oracle import cpu

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)       # tolerance stated by the reference test (atol=1e-5)


def _setup(dev, seed):
    rng = np.random.default_rng(seed)
    rows, width = 500, 400
    param = rng.standard_normal((rows, width), dtype=np.float32)
    ids = rng.integers(0, rows, size=100).astype(np.float32)
    vals = rng.standard_normal((100, width), dtype=np.float32)
    sl = ops.IndexedSlices(torch.from_numpy(ids).to(dev), torch.from_numpy(vals).to(dev), (rows, width)).deduplicate()
    uniq, _, red = cpu.dedup_reduce(ids, vals)
    np.testing.assert_array_equal(sl.values.cpu().numpy(), red)
    return param, uniq.astype(np.float32), red, sl


def test_adagrad_sparse(dev):
    param, uniq, red, sl = _setup(dev, 1)
    acc = np.abs(np.random.default_rng(2).standard_normal(param.shape, dtype=np.float32))
    p, a = torch.from_numpy(param.copy()).to(dev), torch.from_numpy(acc.copy()).to(dev)
    for _ in range(3):
        ops.dl_call("AdaGradOptimizerSparseUpdate", [p, sl.indices.contiguous(), sl.values.contiguous(), a],
                    scalars=[ctypes.c_float(0.1), ctypes.c_float(1e-7)])
        cpu.adagrad_sparse(param, acc, uniq, red, 0.1, 1e-7)
    np.testing.assert_allclose(p.cpu().numpy(), param, **TOL)
    np.testing.assert_allclose(a.cpu().numpy(), acc, **TOL)


@pytest.mark.parametrize("wd", [None, 0.01])
def test_adam_and_adamw_sparse(dev, wd):
    param, uniq, red, sl = _setup(dev, 3)
    m = np.zeros_like(param)
    v = np.zeros_like(param)
    p, dm, dv = [torch.from_numpy(x.copy()).to(dev) for x in (param, m, v)]
    b1, b2, lr, eps = 0.9, 0.999, 0.01, 1e-7
    for t in range(1, 4):
        sc = [ctypes.c_float(x) for x in (lr, b1, b2, b1 ** t, b2 ** t, eps)]
        if wd is None:
            ops.dl_call("AdamOptimizerSparseUpdate", [p, sl.indices.contiguous(), sl.values.contiguous(), dm, dv],
                        scalars=sc)
        else:
            ops.dl_call("AdamWOptimizerSparseUpdate", [p, sl.indices.contiguous(), sl.values.contiguous(), dm, dv],
                        scalars=sc + [ctypes.c_float(wd)])
        cpu.adam_sparse(param, m, v, uniq, red, lr, b1, b2, np.float32(b1 ** t), np.float32(b2 ** t), eps, wd)
    np.testing.assert_allclose(p.cpu().numpy(), param, **TOL)
    np.testing.assert_allclose(dm.cpu().numpy(), m, **TOL)
    np.testing.assert_allclose(dv.cpu().numpy(), v, **TOL)


def _dedup_case(dev, seed, rows, width, n):
    rng = np.random.default_rng(seed)
    param = rng.standard_normal((rows, width), dtype=np.float32)
    ids = rng.integers(0, rows, size=n).astype(np.float32)
    vals = rng.standard_normal((n, width), dtype=np.float32)
    uniq, _, red = cpu.dedup_reduce(ids, vals)
    return param, uniq.astype(np.float32), red


@pytest.mark.parametrize("rows,width,n", [(500, 400, 100), (300, 7, 64), (2000, 512, 1500), (50, 1030, 20)])
def test_sparse_optimizers_every_width_path(dev, rows, width, n):
    """16-byte vector path (width % 4 == 0), scalar path (width 7), wide rows, more rows than one pass of
    the grid: AdaGrad, Adam, AdamW and sparse L2 against the numpy oracle."""
    param, uniq, red = _dedup_case(dev, rows + width, rows, width, n)
    d_ids, d_red = torch.from_numpy(uniq).to(dev), torch.from_numpy(red).to(dev)
    # sparse L2 (OptimizerLink.py:8-21): grads += l2reg * param[ids]
    g2 = d_red.clone()
    ops.dl_call("AddL2RegularizationSparse", [torch.from_numpy(param).to(dev), d_ids, g2],
                scalars=[ctypes.c_float(0.3)])
    np.testing.assert_allclose(g2.cpu().numpy(), cpu.l2_sparse(param, uniq, red, 0.3), rtol=1e-6, atol=1e-6)
    # AdaGrad
    p0, acc = param.copy(), np.abs(np.random.default_rng(5).standard_normal(param.shape, dtype=np.float32))
    p, a = torch.from_numpy(p0.copy()).to(dev), torch.from_numpy(acc.copy()).to(dev)
    ops.dl_call("AdaGradOptimizerSparseUpdate", [p, d_ids, d_red, a], scalars=[ctypes.c_float(0.1), ctypes.c_float(1e-7)])
    cpu.adagrad_sparse(p0, acc, uniq, red, 0.1, 1e-7)
    np.testing.assert_allclose(p.cpu().numpy(), p0, **TOL)
    np.testing.assert_allclose(a.cpu().numpy(), acc, **TOL)
    # AdamW
    p1, m, v = param.copy(), np.zeros_like(param), np.zeros_like(param)
    dp, dm, dv = [torch.from_numpy(x.copy()).to(dev) for x in (p1, m, v)]
    sc = [ctypes.c_float(x) for x in (0.01, 0.9, 0.999, 0.9, 0.999, 1e-7, 0.01)]
    ops.dl_call("AdamWOptimizerSparseUpdate", [dp, d_ids, d_red, dm, dv], scalars=sc)
    cpu.adam_sparse(p1, m, v, uniq, red, 0.01, 0.9, 0.999, np.float32(0.9), np.float32(0.999), 1e-7, 0.01)
    np.testing.assert_allclose(dp.cpu().numpy(), p1, **TOL)
    np.testing.assert_allclose(dm.cpu().numpy(), m, **TOL)
    np.testing.assert_allclose(dv.cpu().numpy(), v, **TOL)


@pytest.mark.parametrize("nesterov", [False, True])
@pytest.mark.parametrize("rows,width,n", [(500, 400, 100), (64, 128, 900), (40, 6, 50)])
def test_momentum_sparse_with_repeated_ids(dev, nesterov, rows, width, n):
    """MomentumOptimizerSparseUpdate is called WITHOUT deduplication (OptimizerLink.py:37-49): ids repeat.
    Three steps, velocity carried over, dense second phase over the whole table."""
    rng = np.random.default_rng(rows * 3 + width + int(nesterov))
    param = rng.standard_normal((rows, width), dtype=np.float32)
    veloc = np.zeros_like(param)
    p, v = torch.from_numpy(param.copy()).to(dev), torch.from_numpy(veloc.copy()).to(dev)
    for _ in range(3):
        ids = np.minimum(rng.zipf(1.5, size=n) - 1, rows - 1).astype(np.float32)
        g = rng.standard_normal((n, width), dtype=np.float32)
        ops.dl_call("MomentumOptimizerSparseUpdate", [p, torch.from_numpy(ids).to(dev), torch.from_numpy(g).to(dev), v],
                    scalars=[ctypes.c_float(0.01), ctypes.c_float(0.9), ctypes.c_bool(nesterov)])
        cpu.momentum_sparse(param, veloc, ids, g, 0.01, 0.9, nesterov)
    np.testing.assert_allclose(p.cpu().numpy(), param, **TOL)
    np.testing.assert_allclose(v.cpu().numpy(), veloc, **TOL)


@pytest.mark.parametrize("rows,width,n", [(500, 400, 100), (3000, 512, 2500), (60, 10, 30)])
def test_lamb_sparse(dev, rows, width, n):
    param, uniq, red = _dedup_case(dev, 77 + rows, rows, width, n)
    m, v = np.zeros_like(param), np.zeros_like(param)
    dp, dm, dv = [torch.from_numpy(x.copy()).to(dev) for x in (param, m, v)]
    d_ids, d_red = torch.from_numpy(uniq).to(dev), torch.from_numpy(red).to(dev)
    b1, b2 = 0.9, 0.999
    for t in range(1, 4):
        sc = [ctypes.c_float(x) for x in (0.01, b1, b2, b1 ** t, b2 ** t, 1e-7, 0.01)]
        ops.dl_call("LambOptimizerSparseUpdate", [dp, d_ids, d_red, dm, dv], scalars=sc)
        cpu.lamb_sparse(param, m, v, uniq, red, 0.01, b1, b2, np.float32(b1 ** t), np.float32(b2 ** t), 1e-7, 0.01)
    np.testing.assert_allclose(dp.cpu().numpy(), param, **TOL)
    np.testing.assert_allclose(dm.cpu().numpy(), m, **TOL)
    np.testing.assert_allclose(dv.cpu().numpy(), v, **TOL)


def test_cpu_named_symbols_serve_gpu_arrays(dev):
    """cpu_EmbeddingLookup / cpu_SGDOptimizerSparseUpdate (c_runtime_api.h:811-818; hasattr-probed by
    python/hetu/_base.py:8-11,72): GPU arrays run on the HIP kernels and are complete on return."""
    from herald_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(4)
    table = rng.standard_normal((300, 64), dtype=np.float32)
    ids = rng.integers(0, 300, size=(20, 5)).astype(np.float32)
    g = rng.standard_normal((100, 64), dtype=np.float32)
    t, i, gg = [torch.from_numpy(x).to(dev) for x in (table.copy(), ids, g)]
    out = torch.empty((20, 5, 64), dtype=torch.float32, device=dev)
    hs = [ops.DLHolder(x) for x in (t, i, out, gg)]
    assert L.cpu_EmbeddingLookup(hs[0].handle, hs[1].handle, hs[2].handle) == 0
    np.testing.assert_array_equal(out.cpu().numpy(), cpu.embedding_lookup(table, ids))   # no sync needed
    flat = ops.DLHolder(i.reshape(-1))
    assert L.cpu_SGDOptimizerSparseUpdate(hs[0].handle, flat.handle, hs[3].handle, ctypes.c_float(0.1)) == 0
    np.testing.assert_array_equal(t.cpu().numpy(), cpu.sgd_sparse_update(table.copy(), ids.reshape(-1), g, 0.1))


@pytest.mark.parametrize("rows,width", [(300, 64), (40000, 512)])
def test_cpu_named_symbols_serve_host_arrays(dev, rows, width):
    """The same two names on HOST arrays, what the reference's callers pass (EmbeddingLookUp.py:16-17,
    optimizer.py:203-207): the HIP kernels still do the work -- small arrays are copied to the device and back, a
    table of 64 MiB or more (the second case: 78 MiB) is page-locked and mapped, only the named rows cross PCIe.
    Bit-exact against the oracle; a table on the GPU with ids on the host works as well."""
    from herald_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(rows)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    ids = rng.integers(0, rows, size=(20, 5)).astype(np.float32)
    ids[3, :] = ids[0, 0]                      # a run of equal ids
    g = rng.standard_normal((100, width), dtype=np.float32)
    want_out = cpu.embedding_lookup(table, ids)
    want_t = cpu.sgd_sparse_update(table.copy(), ids.reshape(-1), g, 0.1)
    h_t, h_i, h_g = torch.from_numpy(table.copy()), torch.from_numpy(ids), torch.from_numpy(g)
    h_o = torch.full((20, 5, width), -7.0)
    hs = [ops.DLHolder(x) for x in (h_t, h_i, h_o, h_g)]
    assert L.cpu_EmbeddingLookup(hs[0].handle, hs[1].handle, hs[2].handle) == 0, L.ha_last_error()
    np.testing.assert_array_equal(h_o.numpy(), want_out)
    flat = ops.DLHolder(h_i.reshape(-1))
    for _ in range(2):                         # the second call finds the table still registered
        h_t.copy_(torch.from_numpy(table))
        assert L.cpu_SGDOptimizerSparseUpdate(hs[0].handle, flat.handle, hs[3].handle, ctypes.c_float(0.1)) == 0, \
            L.ha_last_error()
        np.testing.assert_array_equal(h_t.numpy(), want_t)
    d_t = torch.from_numpy(table.copy()).to(dev)
    mixed = ops.DLHolder(d_t)
    assert L.cpu_SGDOptimizerSparseUpdate(mixed.handle, flat.handle, hs[3].handle, ctypes.c_float(0.1)) == 0
    np.testing.assert_array_equal(d_t.cpu().numpy(), want_t)
    # a registered host array is unmapped before it is freed (ha_host_unmap; unknown pointers are ignored) ...
    assert L.ha_host_unmap(ctypes.c_void_p(h_t.data_ptr())) == 0 and L.ha_host_unmap(ctypes.c_void_p(h_t.data_ptr())) == 0
    # ... after which the same array is simply registered again by the next call
    h_t.copy_(torch.from_numpy(table))
    assert L.cpu_SGDOptimizerSparseUpdate(hs[0].handle, flat.handle, hs[3].handle, ctypes.c_float(0.1)) == 0, L.ha_last_error()
    np.testing.assert_array_equal(h_t.numpy(), want_t)
    assert L.ha_scratch_release() == 0         # unregisters whatever is left before the arrays are freed


@pytest.mark.parametrize("kind", ["adagrad", "adam", "adamw"])
@pytest.mark.parametrize("rows,width,n", [(500, 400, 100), (3000, 512, 6656), (40, 64, 3000), (300, 7, 200),
                                          (100000, 128, 40000)])
def test_fused_dedup_plus_optimizer_equals_the_two_step_sequence(dev, kind, rows, width, n):
    """ha_sparse_opt_fused_f32ids on the RAW ids (duplicates, hot keys: short, medium and long runs; the
    by-unique path above 36,864 ids) == grad.deduplicate() + the optimizer symbol, bit for bit; and within the
    reference tests' 1e-5 of the numpy oracle."""
    rng = np.random.default_rng(rows + width + n)
    param = rng.standard_normal((rows, width), dtype=np.float32)
    hot = rng.integers(0, rows, size=6)
    ids = np.where(rng.random(n) < 0.4, hot[rng.integers(0, 6, size=n)], rng.integers(0, rows, size=n)).astype(np.float32)
    vals = rng.standard_normal((n, width), dtype=np.float32)
    s1 = np.abs(rng.standard_normal((rows, width), dtype=np.float32)) if kind == "adagrad" else np.zeros_like(param)
    s2 = np.zeros_like(param)
    hyper = dict(lr=0.01, eps=1e-7, beta1=0.9, beta2=0.999, beta1t=0.9, beta2t=0.999, weight_decay=0.01)
    # fused
    fp, f1, f2 = [torch.from_numpy(x.copy()).to(dev) for x in (param, s1, s2)]
    ops.sparse_opt_fused(kind, fp, torch.from_numpy(ids).to(dev), torch.from_numpy(vals).to(dev), f1,
                         None if kind == "adagrad" else f2, **hyper)
    # two steps through the reference-named symbols
    tp, t1, t2 = [torch.from_numpy(x.copy()).to(dev) for x in (param, s1, s2)]
    sl = ops.IndexedSlices(torch.from_numpy(ids).to(dev), torch.from_numpy(vals).to(dev), (rows, width)).deduplicate()
    di, dv = sl.indices.contiguous(), sl.values.contiguous()
    if kind == "adagrad":
        ops.dl_call("AdaGradOptimizerSparseUpdate", [tp, di, dv, t1], scalars=[ctypes.c_float(0.01), ctypes.c_float(1e-7)])
    else:
        sc = [ctypes.c_float(x) for x in (0.01, 0.9, 0.999, 0.9, 0.999, 1e-7)]
        if kind == "adam":
            ops.dl_call("AdamOptimizerSparseUpdate", [tp, di, dv, t1, t2], scalars=sc)
        else:
            ops.dl_call("AdamWOptimizerSparseUpdate", [tp, di, dv, t1, t2], scalars=sc + [ctypes.c_float(0.01)])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(fp.cpu().numpy(), tp.cpu().numpy())
    np.testing.assert_array_equal(f1.cpu().numpy(), t1.cpu().numpy())
    if kind != "adagrad":
        np.testing.assert_array_equal(f2.cpu().numpy(), t2.cpu().numpy())
    # numpy oracle
    uniq, _, red = cpu.dedup_reduce(ids, vals)
    if kind == "adagrad":
        cpu.adagrad_sparse(param, s1, uniq, red, 0.01, 1e-7)
    else:
        cpu.adam_sparse(param, s1, s2, uniq, red, 0.01, 0.9, 0.999, np.float32(0.9), np.float32(0.999), 1e-7,
                        0.01 if kind == "adamw" else None)
    np.testing.assert_allclose(fp.cpu().numpy(), param, **TOL)
    np.testing.assert_allclose(f1.cpu().numpy(), s1, **TOL)
