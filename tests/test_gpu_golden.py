"""GPU parity against the REFERENCE's own compiled CPU operators: the HIP gather and sparse-SGD apply on
the committed fixtures of tests/golden/dnnl_ops.json (cpu_EmbeddingLookup / cpu_SGDOptimizerSparseUpdate,
src/dnnl_ops/EmbeddingLookup.cpp:16-35, Optimizers.cpp:51-74, run by tests/golden/make_golden.py),
bit for bit -- no oracle in between."""
import ctypes
import json
import os
import sys

import numpy as np
import pytest
import torch

from herald_amd import ops

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import formula  # noqa: E402

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLD, "dnnl_ops.json")))


def _arrays(c):
    ids = formula.from_bits(c["ids_bits"], c["ids_shape"])
    n, w = ids.size, c["width"]
    grads = formula.from_bits(c["grads_bits"], (n, w))
    lr = float(formula.from_bits([c["lr_bits"]], (1,))[0])
    out = formula.from_bits(c["out_bits"], tuple(c["ids_shape"]) + (w,))
    touched = formula.from_bits(c["touched_bits"], (len(c["touched_rows"]), w))
    return ids, grads, lr, out, touched


def _device_table(case, dev):
    if case["rows"] <= 1 << 20:
        return torch.from_numpy(formula.table(case["rows"], case["width"])).to(dev)
    t = torch.empty((case["rows"], case["width"]), dtype=torch.float32, device=dev)
    step = 1 << 22
    for s in range(0, case["rows"], step):
        e = min(case["rows"], s + step)
        t[s:e] = torch.from_numpy(formula.rows_of(np.arange(s, e), case["width"])).to(dev)
    return t


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("path", ["unfused", "fused", "dl_symbols"])
def test_hip_matches_reference_dnnl_golden(dev, case, path):
    ids, grads, lr, want_out, want_rows = _arrays(case)
    n, w = ids.size, case["width"]
    table = _device_table(case, dev)
    d_ids = torch.from_numpy(ids).to(dev)
    d_grads = torch.from_numpy(grads).to(dev)
    if path == "unfused":
        out = ops.embedding_lookup(table, d_ids)
        if n:
            plan = ops.IndexPlan(n, dev).build(d_ids.reshape(-1))
            ops.sgd_apply(table, plan, d_grads, lr)
    elif path == "fused":
        plan = ops.IndexPlan(max(n, 1), dev)
        out = ops.lookup_sort(table, d_ids.reshape(-1), plan).reshape(tuple(ids.shape) + (w,))
        ops.sgd_apply_finish(table, plan, d_grads, lr)
    else:
        out = torch.empty(tuple(ids.shape) + (w,), dtype=torch.float32, device=dev)
        if n:
            ops.dl_call("DLGpuEmbeddingLookUp", [table, d_ids, out])
            ops.dl_call("SGDOptimizerSparseUpdate", [table, d_ids.reshape(-1), d_grads], scalars=(ctypes.c_float(lr),))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), want_out.view(np.uint32))
    rows = case["touched_rows"]
    if rows:
        got = table[torch.tensor(rows, device=dev)].cpu().numpy()
        np.testing.assert_array_equal(got.view(np.uint32), want_rows.view(np.uint32))
    # untouched neighbours still hold the closed form
    probe = [r for r in (0, 1, case["rows"] // 2, case["rows"] - 1) if r not in set(rows)]
    if probe:
        got = table[torch.tensor(probe, device=dev)].cpu().numpy()
        np.testing.assert_array_equal(got, formula.rows_of(probe, w))


DEDUP = json.load(open(os.path.join(GOLD, "dedup.json")))


@pytest.mark.parametrize("case", DEDUP, ids=lambda c: c["name"])
@pytest.mark.parametrize("path", ["dedup_reduce", "indexed_slices", "dl_symbol"])
def test_hip_dedup_matches_reference_cpu_deduplicate_golden(dev, case, path):
    """HIP dedup-reduce against tests/golden/dedup.json = the reference's IndexedSlices.cpu_deduplicate
    (python/hetu/ndarray.py:556-576) executed from its own text by make_golden.py: unique ids, reduced rows and
    the deduplicated push indices, bit for bit, through ha_dedup_reduce, through ops.IndexedSlices.deduplicate
    (the reference's GPU call sequence, ndarray.py:532-554) and through the DeduplicateIndexedSlices symbol."""
    ids = formula.from_bits(case["ids_bits"], case["ids_shape"])
    n, w = ids.size, case["width"]
    vals = formula.from_bits(case["values_bits"], (n, w))
    uniq = formula.from_bits(case["uniq_bits"], (len(case["uniq_bits"]),))
    red = formula.from_bits(case["reduced_bits"], (uniq.size, w))
    if n == 0:      # the reference returns empty arrays; the library has nothing to launch
        assert uniq.size == 0 and red.size == 0
        assert ops.IndexedSlices(torch.from_numpy(ids).to(dev), torch.empty(tuple(ids.shape) + (w,), device=dev)) \
            .deduplicate().indices.numel() == 0
        return
    d_ids = torch.from_numpy(ids).to(dev)
    d_vals = torch.from_numpy(vals).to(dev)
    if path == "dedup_reduce":
        plan = ops.IndexPlan(n, dev).build(d_ids.reshape(-1))
        got = ops.dedup_reduce(plan, d_vals)[:plan.n_unique()]
        got_u, _ = plan.export_f32()
    elif path == "indexed_slices":
        push = None
        if case["push_bits"] is not None:
            push = torch.from_numpy(formula.from_bits(case["push_bits"], (len(case["push_bits"]),))).to(dev)
        sl = ops.IndexedSlices(d_ids, d_vals.reshape(tuple(ids.shape) + (w,)), push_indices=push).deduplicate()
        got, got_u = sl.values, sl.indices
        if push is not None:
            want_p = formula.from_bits(case["push_uniq_bits"], (len(case["push_uniq_bits"]),))
            np.testing.assert_array_equal(sl.push_indices.cpu().numpy().view(np.uint32), want_p.view(np.uint32))
    else:
        ru, rinv = np.unique(ids.reshape(-1), return_inverse=True)      # the host step of ndarray.py:534
        got = torch.zeros((ru.size, w), dtype=torch.float32, device=dev)
        ops.dl_call("DeduplicateIndexedSlices",
                    [d_vals, torch.from_numpy(rinv.reshape(-1).astype(np.float32)).to(dev), got])
        got_u = torch.from_numpy(ru.astype(np.float32))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got_u.cpu().numpy().view(np.uint32), uniq.view(np.uint32))
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), red.view(np.uint32))
