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
