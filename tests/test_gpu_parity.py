"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest
import torch

from herald_amd import ops, synth
from oracle import cpu

pytestmark = pytest.mark.gpu


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _u32(t):
    return (t.cpu().numpy().astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)


@pytest.mark.parametrize("width", [1, 3, 4, 16, 64, 128, 200, 512, 1024])
@pytest.mark.parametrize("n", [1, 63, 6656])
def test_gather_bit_exact(dev, width, n):
    rng = np.random.default_rng(width * 7 + n)
    rows = 5000
    table = rng.standard_normal((rows, width), dtype=np.float32)
    ids = rng.integers(0, rows, size=n).astype(np.float32)
    want = cpu.embedding_lookup(table, ids)
    got = ops.embedding_lookup(_dev(table, dev), _dev(ids, dev))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_gather_shapes_and_empty(dev):
    table = _dev(np.arange(25, dtype=np.float32).reshape(5, 5), dev)
    ids = _dev(np.array([[0, 1], [0, 1]], dtype=np.float32), dev)   # tests/test_dnnl_op.py:1120-1135
    out = ops.embedding_lookup(table, ids)
    assert tuple(out.shape) == (2, 2, 5)
    np.testing.assert_array_equal(out.cpu().numpy(), table.cpu().numpy()[[[0, 1], [0, 1]]])
    empty = ops.embedding_lookup(table, torch.empty(0, dtype=torch.float32, device=dev))
    assert tuple(empty.shape) == (0, 5)


def test_gather_u64_ids(dev):
    rng = np.random.default_rng(5)
    table = rng.standard_normal((3000, 64), dtype=np.float32)
    ids = rng.integers(0, 3000, size=(40, 26)).astype(np.int64)
    got = ops.embedding_lookup(_dev(table, dev), _dev(ids, dev))
    np.testing.assert_array_equal(got.cpu().numpy(), table[ids])


def test_gather_through_reference_named_symbol(dev):
    rng = np.random.default_rng(6)
    table = _dev(rng.standard_normal((777, 128), dtype=np.float32), dev)
    ids = _dev(rng.integers(0, 777, size=(256, 26)).astype(np.float32), dev)
    out = torch.empty((256, 26, 128), dtype=torch.float32, device=dev)
    ops.dl_call("DLGpuEmbeddingLookUp", [table, ids, out])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(),
                                  cpu.embedding_lookup(table.cpu().numpy(), ids.cpu().numpy()))


def _check_plan(ids_f32, dev):
    n = ids_f32.size
    plan = ops.IndexPlan(max(n, 1), dev).build(_dev(ids_f32, dev))
    torch.cuda.synchronize()
    keys = cpu.ids_to_keys(ids_f32)
    uniq, inv, cnt = cpu.unique(keys)
    u = plan.n_unique()
    assert u == uniq.size
    if n == 0:
        return plan
    np.testing.assert_array_equal(_u32(plan.keys()), keys)
    np.testing.assert_array_equal(_u32(plan.uniq(u)), uniq)
    np.testing.assert_array_equal(plan.inverse().cpu().numpy().astype(np.int64), inv)
    np.testing.assert_array_equal(plan.counts(u).cpu().numpy().astype(np.int64), cnt)
    # stable argsort: occurrence order inside every run
    perm = plan.perm().cpu().numpy()
    np.testing.assert_array_equal(perm, np.argsort(keys, kind="stable"))
    np.testing.assert_array_equal(_u32(plan.sorted_keys()), keys[perm])
    seg = plan.seg(u).cpu().numpy()
    np.testing.assert_array_equal(seg, np.concatenate([[0], np.cumsum(cnt)]))
    # float32 export == np.unique on the float ids (python/hetu/ndarray.py:534)
    uf, invf = plan.export_f32()
    ru, rinv = np.unique(ids_f32, return_inverse=True)
    np.testing.assert_array_equal(uf.cpu().numpy(), ru)
    np.testing.assert_array_equal(invf.cpu().numpy(), rinv.astype(np.float32))
    return plan


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1000, 6656, 15360, 15361, 26624, 36864])
def test_plan_small_path(dev, n):
    rng = np.random.default_rng(n)
    ids = rng.integers(0, max(2, n // 2 + 1), size=n).astype(np.float32)
    _check_plan(ids, dev)


@pytest.mark.parametrize("n", [36865, 50000, 106496, 300000, 1048576, 1100000])
def test_plan_radix_path(dev, n):
    rng = np.random.default_rng(n)
    ids = synth.as_f32_ids(rng.integers(0, synth.CRITEO_ROWS, size=n))
    ids[: n // 4] = ids[n // 2: n // 2 + n // 4]        # force duplicates
    _check_plan(ids, dev)


def test_plan_radix_path_replays_from_a_graph(dev):
    """The one-launch radix passes exchange their histograms behind flags that the first pass's scatter zeroes: nothing
    of a call survives it, so a captured build replays on new ids (and on the same ones) with the right result."""
    from herald_amd import ops
    n = 106496
    rng = np.random.default_rng(5)
    buf = torch.zeros(n, dtype=torch.float32, device=dev)
    plan = ops.IndexPlan(n, dev)
    side = torch.cuda.Stream(device=dev)
    first = synth.as_f32_ids(rng.integers(0, synth.CRITEO_ROWS, size=n))
    buf.copy_(torch.from_numpy(first).to(dev))
    with torch.cuda.stream(side):
        plan.build(buf, stream=side, key_limit=synth.CRITEO_ROWS)       # first use: eager
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        plan.build(buf, stream=side, key_limit=synth.CRITEO_ROWS)
    for rep in range(4):
        ids = synth.as_f32_ids(rng.integers(0, 1000 if rep == 2 else synth.CRITEO_ROWS, size=n))
        if rep == 3:
            ids = first
        buf.copy_(torch.from_numpy(ids).to(dev))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        keys = cpu.ids_to_keys(ids)
        uniq, inv, cnt = cpu.unique(keys)
        plan._view = None
        u = plan.n_unique()
        assert u == uniq.size
        np.testing.assert_array_equal(_u32(plan.uniq(u)), uniq)
        np.testing.assert_array_equal(plan.inverse().cpu().numpy().astype(np.int64), inv)
        np.testing.assert_array_equal(plan.counts(u).cpu().numpy().astype(np.int64), cnt)
        np.testing.assert_array_equal(plan.perm().cpu().numpy(), np.argsort(keys, kind="stable"))


@pytest.mark.parametrize("n,block,finish_ahead", [(40000, 3, False), (40000, 2, True), (9000, 2, False)])
def test_sort_ahead_pipeline_equals_the_step_by_step_calls(dev, n, block, finish_ahead):
    """ops.SortAheadPipeline (batches beyond the work-queue step's 7,168 ids): plans sorted a block ahead on a side stream,
    gather + apply-and-finish per step -- every output row and the table after every step as the CPU sequence has them."""
    rng = np.random.default_rng(n)
    rows, width, steps = 50000, 16, 7
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    ids = [synth.as_f32_ids(rng.integers(0, rows if k % 2 else 300, size=n)) for k in range(steps)]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(steps)]
    t = _dev(table0, dev)
    pipe = ops.SortAheadPipeline(t, n, 0.25, block=block, finish_ahead=finish_ahead)
    d_ids = [_dev(x, dev) for x in ids]
    d_grads = [_dev(g, dev) for g in grads]
    want = table0.copy()
    starts = list(range(0, steps, block))
    pipe.prepare_block(d_ids[0:block])
    for k in range(steps):
        if k % block == 0 and k + block < steps:
            pipe.prepare_block(d_ids[k + block:k + 2 * block])          # the next block, beside this one's steps
        out = pipe.lookup(k, d_ids[k])
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), want[ids[k].astype(np.int64)], err_msg="lookup of step %d" % k)
        plan = pipe.apply(k, d_grads[k])
        want = cpu.sgd_sparse_update(want, ids[k], grads[k], 0.25)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(t.cpu().numpy(), want, err_msg="table after step %d" % k)
        assert plan.n_unique() == np.unique(ids[k]).size
    assert starts[-1] < steps
    with pytest.raises(ValueError):
        pipe.lookup(steps + 5, d_ids[0])


@pytest.mark.parametrize("n,width", [(40000, 128), (106496, 64)])
def test_sgd_apply_on_a_finished_plan_maps_waves_to_unique_keys(dev, n, width):
    """ha_sgd_apply_finished above 36,864 ids: bit-equal to the CPU chain, and -- with the tolerance mode on -- bit-equal
    to ha_sgd_apply's fixed-order trees (one tree shape, whichever way the waves are mapped); tolerance mode 2: the finished
    plan's apply sums runs beyond 256 occurrences in chunks of 256 (oracle/qstep_model.py listed_chunking)."""
    rng = np.random.default_rng(n + width)
    rows = 300000
    ids = synth.as_f32_ids(rng.integers(0, rows, size=n))
    ids[100:100 + 3000] = 77.0                  # a 3,000-occurrence run
    ids[5000:5050] = 99.0                       # runs around the long-run threshold
    ids[6000:6063] = 1234.0
    ids[7000:7064] = 4321.0
    grads = rng.standard_normal((n, width), dtype=np.float32)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    plan = ops.IndexPlan(n, dev).build(_dev(ids, dev))
    t = _dev(table, dev)
    ops.sgd_apply(t, plan, _dev(grads, dev), 0.3, finished=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t.cpu().numpy(), cpu.sgd_sparse_update(table.copy(), ids, grads, 0.3))
    ops.set_tolerance_mode(True)
    try:
        a, b = _dev(table, dev), _dev(table, dev)
        ops.sgd_apply(a, plan, _dev(grads, dev), 0.3, finished=True)
        ops.sgd_apply(b, plan, _dev(grads, dev), 0.3)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
        ops.set_tolerance_mode(2)
        from oracle import qstep_model
        assert qstep_model.listed_chunking(ids, width)
        a, b = _dev(table, dev), _dev(table, dev)
        ops.sgd_apply(a, plan, _dev(grads, dev), 0.3, finished=True)
        ops.sgd_apply(b, plan, _dev(grads, dev), 0.3)
        torch.cuda.synchronize()
        for got, chunked in ((a, True), (b, False)):
            want = qstep_model.sgd_sparse_update(table.copy(), ids, grads, 0.3, long_min=None, coop_min=64, chunked=chunked)
            np.testing.assert_array_equal(got.cpu().numpy(), want)
    finally:
        ops.set_tolerance_mode(False)


def test_plan_all_equal_and_all_distinct(dev):
    _check_plan(np.full(6656, 12345.0, dtype=np.float32), dev)
    _check_plan(np.arange(6656, dtype=np.float32)[::-1].copy(), dev)
    _check_plan(np.full(20000, 7.0, dtype=np.float32), dev)


def test_plan_criteo_batches(dev):
    for step in range(3):
        ids = synth.as_f32_ids(synth.criteo_batch(256, step)).reshape(-1)
        _check_plan(ids, dev)


@pytest.mark.parametrize("width", [1, 4, 6, 64, 128, 512])
@pytest.mark.parametrize("n", [1, 200, 6656])
def test_sgd_apply_bit_exact(dev, width, n):
    rng = np.random.default_rng(n + width)
    rows = 3000
    table = rng.standard_normal((rows, width), dtype=np.float32)
    ids = rng.integers(0, max(2, n // 3), size=n).astype(np.float32)
    if n >= 200:
        ids[:150] = 3.0                                  # a hot row with >=150 occurrences
    grads = rng.standard_normal((n, width), dtype=np.float32)
    lr = 0.01
    want = cpu.sgd_sparse_update(table.copy(), ids, grads, lr)
    t = _dev(table, dev)
    ops.sgd_sparse_update(t, _dev(ids, dev), _dev(grads, dev), lr)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t.cpu().numpy(), want)


def test_dedup_reduce_large_batch(dev):
    """n > 36,864: the dedup-reduce maps waves to unique keys (runs of thousands included)."""
    rng = np.random.default_rng(41)
    n, width = 60000, 64
    ids = _runs_batch(rng, [3000, 1500, 700, 100, 48, 47, 5], 20000, 500000)
    ids = np.concatenate([ids, rng.integers(0, 500000, size=n - ids.size).astype(np.float32)])
    grads = rng.standard_normal((n, width), dtype=np.float32)
    plan = ops.IndexPlan(n, dev).build(_dev(ids, dev))
    uniq, _, want = cpu.dedup_reduce(ids, grads)
    red = ops.dedup_reduce(plan, _dev(grads, dev))
    np.testing.assert_array_equal(red[:uniq.size].cpu().numpy(), want)
    red2 = ops.dedup_reduce(plan, _dev(grads, dev), scale=0.5)       # scale_values multiplies by -lr
    np.testing.assert_array_equal(red2[:uniq.size].cpu().numpy(), cpu.dedup_reduce(ids, cpu.scale_values(grads, -0.5))[2])


@pytest.mark.parametrize("width", [4, 128, 512])
def test_dedup_reduce_bit_exact(dev, width):
    rng = np.random.default_rng(width)
    n = 6656
    ids = synth.as_f32_ids(synth.criteo_batch(256, 1)).reshape(-1)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    uniq, inv, want = cpu.dedup_reduce(ids, grads)
    plan = ops.IndexPlan(n, dev).build(_dev(ids, dev))
    red = ops.dedup_reduce(plan, _dev(grads, dev))
    torch.cuda.synchronize()
    u = plan.n_unique()
    assert u == uniq.size
    np.testing.assert_array_equal(red[:u].cpu().numpy(), want)


def test_indexed_slices_deduplicate_matches_cpu_deduplicate(dev):
    # scenario of tests/test_optimizer.py:117-198: 500x400 table, 100 random duplicated ids
    rng = np.random.default_rng(3)
    ids = rng.integers(0, 500, size=100).astype(np.float32)
    vals = rng.standard_normal((100, 400), dtype=np.float32)
    s = ops.IndexedSlices(_dev(ids, dev), _dev(vals, dev), dense_shape=(500, 400)).deduplicate()
    ru, rvals = cpu.np_cpu_deduplicate(ids, vals)
    np.testing.assert_array_equal(s.indices.cpu().numpy(), ru)
    np.testing.assert_array_equal(s.values.cpu().numpy(), rvals)
    dense = s.to_dense()
    want = np.zeros((500, 400), dtype=np.float32)
    want[ru.astype(np.int64)] = rvals
    np.testing.assert_array_equal(dense.cpu().numpy(), want)


def test_push_apply_and_reference_named_scatter_ops(dev):
    rng = np.random.default_rng(4)
    rows, width, n = 2000, 128, 1200                     # tests/pstests/test_apis.py:105-157 shape family
    table = rng.standard_normal((rows, width), dtype=np.float32)
    ids = rng.integers(0, rows, size=n).astype(np.float32)
    vals = rng.standard_normal((n, width), dtype=np.float32)
    uniq, inv, red = cpu.dedup_reduce(ids, vals)
    want = cpu.push_apply(table.copy(), uniq, red)
    t = _dev(table, dev)
    ops.dl_call("IndexedSlicesOneSideAdd", [_dev(ids, dev), _dev(vals, dev), t])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t.cpu().numpy(), want)
    # DLGpuEmbeddingLookUp_Gradient: dense grad = scatter-add into zeros
    g = torch.full((rows, width), 7.0, dtype=torch.float32, device=dev)
    ops.dl_call("DLGpuEmbeddingLookUp_Gradient", [_dev(vals, dev), _dev(ids, dev), g])
    want_g = cpu.push_apply(np.zeros((rows, width), dtype=np.float32), uniq, red)
    np.testing.assert_array_equal(g.cpu().numpy(), want_g)
    # SGDOptimizerSparseUpdate through the DLArray ABI
    import ctypes
    t2 = _dev(table, dev)
    ops.dl_call("SGDOptimizerSparseUpdate", [t2, _dev(ids, dev), _dev(vals, dev)],
                scalars=[ctypes.c_float(0.05)])
    want2 = cpu.sgd_sparse_update(table.copy(), ids, vals, 0.05)
    np.testing.assert_array_equal(t2.cpu().numpy(), want2)
    # DeduplicateIndexedSlices with a host-computed inverse, as the reference calls it
    ru, rinv = np.unique(ids, return_inverse=True)
    comp = torch.zeros((ru.size, width), dtype=torch.float32, device=dev)
    ops.dl_call("DeduplicateIndexedSlices", [_dev(vals, dev), _dev(rinv.astype(np.float32), dev), comp])
    np.testing.assert_array_equal(comp.cpu().numpy(), red)


def test_full_size_step_properties(dev):
    """BASELINE config A at full width on a table slice: lookup -> apply round trip and linearity."""
    rows, width, n = 200000, 512, 6656
    rng = np.random.default_rng(9)
    table = torch.from_numpy(rng.standard_normal((rows, width), dtype=np.float32)).to(dev)
    ids_np = (synth.criteo_batch(256, 0).reshape(-1) % rows).astype(np.float32)
    ids = _dev(ids_np, dev)
    before = ops.embedding_lookup(table, ids).clone()
    grads = torch.from_numpy(rng.standard_normal((n, width), dtype=np.float32)).to(dev)
    plan = ops.IndexPlan(n, dev).build(ids)
    t0 = table.clone()
    ops.sgd_apply(table, plan, grads, 0.5)
    # rows not in the batch are untouched
    mask = torch.ones(rows, dtype=torch.bool, device=dev)
    mask[ids.long()] = False
    assert torch.equal(table[mask], t0[mask])
    # a zero gradient is the identity, bit for bit
    t1 = table.clone()
    ops.sgd_apply(table, plan, torch.zeros_like(grads), 0.5)
    assert torch.equal(table, t1)
    # against the oracle on the touched rows only
    want = cpu.sgd_sparse_update(t0.cpu().numpy(), ids_np, grads.cpu().numpy(), 0.5)
    np.testing.assert_array_equal(table.cpu().numpy(), want)
    assert torch.equal(before, ops.embedding_lookup(t0, ids))


@pytest.mark.parametrize("width,n", [(512, 6656), (128, 6656), (64, 100), (512, 15360), (200, 300), (512, 20000),
                                      (512, 26624), (128, 36864), (512, 40000)])
def test_fused_step_matches_oracle_and_unfused(dev, width, n):
    """ha_lookup_sort_* + ha_sgd_apply_finish (two launches) == oracle == the unfused four calls."""
    rng = np.random.default_rng(width + n)
    rows = 40000
    table = rng.standard_normal((rows, width), dtype=np.float32)
    ids = (synth.criteo_batch(max(1, (n + 25) // 26), 3).reshape(-1)[:n] % rows).astype(np.float32)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    lr = 0.05
    want_out = cpu.embedding_lookup(table, ids)
    uniq, inv, cnt = cpu.unique(cpu.ids_to_keys(ids))
    want_t = cpu.sgd_sparse_update(table.copy(), ids, grads, lr)
    t = _dev(table, dev)
    d_ids = _dev(ids, dev)
    plan = ops.IndexPlan(n, dev)
    out = ops.lookup_sort(t, d_ids, plan)
    ops.sgd_apply_finish(t, plan, _dev(grads, dev), lr)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want_out)
    np.testing.assert_array_equal(t.cpu().numpy(), want_t)
    u = plan.n_unique()
    assert u == uniq.size
    np.testing.assert_array_equal(_u32(plan.uniq(u)), uniq)
    np.testing.assert_array_equal(plan.inverse().cpu().numpy().astype(np.int64), inv)
    np.testing.assert_array_equal(plan.counts(u).cpu().numpy().astype(np.int64), cnt)
    # push flavour
    t2 = _dev(table, dev)
    ops.lookup_sort(t2, d_ids, plan)
    ops.push_apply_finish(t2, plan, _dev(grads, dev))
    _, _, red = cpu.dedup_reduce(ids, grads)
    np.testing.assert_array_equal(t2.cpu().numpy(), cpu.push_apply(table.copy(), uniq, red))


def test_backward_with_next_batch_prefetch_is_the_same_backward(dev):
    """ha_sgd_apply_finish_prefetch_f32ids only touches rows of the next batch: table and plan equal the
    plain backward bit for bit (ids of the next batch may exceed the table or the batch length)."""
    rng = np.random.default_rng(17)
    rows, width, n = 20000, 128, 6656
    table = rng.standard_normal((rows, width), dtype=np.float32)
    ids = (synth.criteo_batch(256, 2).reshape(-1) % rows).astype(np.float32)
    nxt = (synth.criteo_batch(300, 3).reshape(-1) % (rows + 500)).astype(np.float32)     # longer, partly out of range
    grads = rng.standard_normal((n, width), dtype=np.float32)
    res = []
    for use_next in (False, True):
        t = _dev(table, dev)
        plan = ops.IndexPlan(n, dev)
        ops.lookup_sort(t, _dev(ids, dev), plan)
        ops.sgd_apply_finish(t, plan, _dev(grads, dev), 0.1, next_ids=_dev(nxt, dev) if use_next else None)
        torch.cuda.synchronize()
        res.append((t.cpu().numpy(), plan.inverse().cpu().numpy(), plan.counts().cpu().numpy()))
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[0][0], cpu.sgd_sparse_update(table.copy(), ids, grads, 0.1))


def test_sharded_single_rank_hip_engine(dev):
    """herald_amd.sharded with the HIP engine at world_size 1: SparsePull / SparsePush semantics."""
    from herald_amd.sharded import ShardedEmbedding
    rng = np.random.default_rng(11)
    rows, width, n = 5000, 128, 2000
    table = rng.standard_normal((rows, width), dtype=np.float32)
    ids = rng.integers(0, rows, size=n).astype(np.float32)
    ids[:500] = ids[0]
    vals = rng.standard_normal((n, width), dtype=np.float32)
    emb = ShardedEmbedding(rows, width, dev, table=_dev(table, dev))
    out, route = emb.pull(_dev(ids, dev), return_route=True)
    np.testing.assert_array_equal(out.cpu().numpy(), cpu.sparse_pull(table, ids))
    emb.push(_dev(ids, dev), _dev(vals, dev), 0.01, route=route)
    want = cpu.sparse_push(table.copy(), ids, vals, 0.01)
    np.testing.assert_array_equal(emb.table.cpu().numpy(), want)
    # a second push without lr scaling (IndexedSlices add semantics)
    emb.push(_dev(ids, dev), _dev(vals, dev))
    want = cpu.sparse_push(want, ids, vals, None)
    np.testing.assert_array_equal(emb.table.cpu().numpy(), want)


def test_shard_bucket_matches_partitioner(dev, lib):
    import ctypes
    rng = np.random.default_rng(12)
    rows, nshard, n = 33762577, 8, 6656
    ids = synth.as_f32_ids(synth.criteo_batch(256, 5)).reshape(-1)
    plan = ops.IndexPlan(n, dev).build(_dev(ids, dev))
    starts = cpu.partition(rows, nshard)
    offsets = torch.empty(nshard + 1, dtype=torch.int32, device=dev)
    local = torch.empty(n, dtype=torch.int32, device=dev)
    st = (ctypes.c_int64 * (nshard + 1))(*[int(x) for x in starts])
    assert lib.ha_shard_bucket(ctypes.c_void_p(plan.ws.data_ptr()), n, st, nshard,
                               ctypes.c_void_p(offsets.data_ptr()), ctypes.c_void_p(local.data_ptr()), None) == 0
    torch.cuda.synchronize()
    uniq = np.unique(cpu.ids_to_keys(ids))
    want_off = np.searchsorted(uniq, starts.astype(np.uint64), side="left")
    want_off[-1] = uniq.size
    np.testing.assert_array_equal(offsets.cpu().numpy(), want_off)
    owner = np.searchsorted(starts, uniq.astype(np.int64), side="right") - 1
    np.testing.assert_array_equal(local.cpu().numpy()[:uniq.size].astype(np.int64),
                                  uniq.astype(np.int64) - starts[owner])


def _runs_batch(rng, run_lengths, n_singles, rows):
    """ids whose sorted order contains one run of every requested length (at whatever alignment the
    lengths before it produce) between single-occurrence keys, in shuffled position order."""
    keys = rng.choice(rows, size=len(run_lengths) + n_singles, replace=False)
    parts = [np.full(L, keys[i], dtype=np.int64) for i, L in enumerate(run_lengths)]
    parts.append(keys[len(run_lengths):])
    ids = np.concatenate(parts)
    rng.shuffle(ids)
    return ids.astype(np.float32)


@pytest.mark.parametrize("width", [64, 200, 512, 1030, 2048])
@pytest.mark.parametrize("mode", ["sgd", "push", "reduce"])
def test_apply_every_run_length_class(dev, width, mode):
    """Short (1-3), medium (4-47) and long (>= 48, full-workgroup cooperative) runs at many lengths and
    alignments, ragged last slices (width 200 / 1030), more slices than full workgroups (width 2048)."""
    rng = np.random.default_rng(width * 3 + len(mode))
    rows = 60000
    lengths = [2, 3, 4, 5, 15, 16, 17, 30, 31, 32, 33, 46, 47, 48, 49, 50, 62, 63, 64, 65, 79, 80, 81, 95, 96,
               97, 127, 128, 129, 143, 144, 160, 200, 255, 256, 257, 300, 511, 512, 513, 1000]
    ids = _runs_batch(rng, lengths, 3000, rows)
    n = ids.size
    assert n <= 15360
    grads = rng.standard_normal((n, width), dtype=np.float32)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    plan = ops.IndexPlan(n, dev).build(_dev(ids, dev))
    if mode == "sgd":
        t = _dev(table, dev)
        ops.sgd_apply(t, plan, _dev(grads, dev), 0.37)
        want = cpu.sgd_sparse_update(table.copy(), ids, grads, 0.37)
        np.testing.assert_array_equal(t.cpu().numpy(), want)
    elif mode == "push":
        t = _dev(table, dev)
        ops.push_apply(t, plan, _dev(grads, dev))
        uniq, _, red = cpu.dedup_reduce(ids, grads)
        np.testing.assert_array_equal(t.cpu().numpy(), cpu.push_apply(table.copy(), uniq, red))
    else:
        red = ops.dedup_reduce(plan, _dev(grads, dev))
        uniq, _, want = cpu.dedup_reduce(ids, grads)
        np.testing.assert_array_equal(red[:uniq.size].cpu().numpy(), want)


@pytest.mark.parametrize("lengths", [[1023, 1024, 1025, 1040, 2047, 2500], [5000, 9000], [15360]])
def test_apply_giant_runs(dev, lengths):
    """Runs longer than the 1024 positions a workgroup scans each way, and a batch that is one run."""
    rng = np.random.default_rng(sum(lengths))
    rows, width = 5000, 128
    ids = _runs_batch(rng, lengths, max(0, 15360 - sum(lengths)) // 3, rows)
    n = ids.size
    grads = rng.standard_normal((n, width), dtype=np.float32)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    t = _dev(table, dev)
    ops.sgd_sparse_update(t, _dev(ids, dev), _dev(grads, dev), 0.5)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t.cpu().numpy(), cpu.sgd_sparse_update(table.copy(), ids, grads, 0.5))


def test_apply_config_c_batch(dev):
    """BASELINE configs[2]: bs=4096 d=128 (n = 106,496 ids: radix sort path, runs of thousands)."""
    rows, width = 2000000, 128
    raw = synth.criteo_batch(4096, 1, rows=rows)
    ids = np.minimum(synth.as_f32_ids(raw).reshape(-1), np.float32(rows - 1))
    n = ids.size
    assert n == 4096 * 26
    rng = np.random.default_rng(5)
    grads = rng.standard_normal((n, width), dtype=np.float32)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    t = _dev(table, dev)
    plan = ops.IndexPlan(n, dev)
    out = ops.lookup_sort(t, _dev(ids, dev), plan)
    np.testing.assert_array_equal(out.cpu().numpy(), cpu.embedding_lookup(table, ids))
    ops.sgd_apply_finish(t, plan, _dev(grads, dev), 0.01)
    torch.cuda.synchronize()
    want = cpu.sgd_sparse_update(table, ids, grads, 0.01)       # in place: the table is 1 GB
    np.testing.assert_array_equal(t.cpu().numpy(), want)
    uniq, inv, cnt = cpu.unique(cpu.ids_to_keys(ids))
    assert plan.n_unique() == uniq.size and int(cnt.max()) > 1000
    np.testing.assert_array_equal(plan.counts(uniq.size).cpu().numpy().astype(np.int64), cnt)


def test_apply_mapped2_equals_two_mapped_passes(dev, lib):
    """ha_apply_mapped2 (one pass, two destinations) == two ha_apply_mapped calls, bit for bit, over
    short / medium / long runs, with un-initialised first-destination rows and skipped second rows."""
    import ctypes
    rng = np.random.default_rng(91)
    width, slots = 192, 6000
    ids = _runs_batch(rng, [2, 3, 4, 9, 17, 40, 47, 48, 49, 100, 300, 700], 1500, 50000)
    n = ids.size
    plan = ops.IndexPlan(n, dev).build(_dev(ids, dev))
    u = plan.n_unique()
    src = _dev(rng.standard_normal((n, width), dtype=np.float32), dev)
    rowmap = rng.permutation(slots)[:u].astype(np.int32)
    rowmap2 = rowmap.copy()
    rowmap2[rng.random(u) < 0.3] = -1                       # lines without a data row
    init = (rng.random(slots) < 0.5).astype(np.uint8)       # lines whose gradient buffer exists already
    a0 = rng.standard_normal((slots, width), dtype=np.float32)
    b0 = rng.standard_normal((slots, width), dtype=np.float32)
    d_rowmap, d_rowmap2, d_init = _dev(rowmap, dev), _dev(rowmap2, dev), _dev(init, dev)
    vp = ctypes.c_void_p
    a1, b1 = _dev(a0, dev), _dev(b0, dev)
    assert lib.ha_apply_mapped(vp(a1.data_ptr()), slots, width, vp(plan.ws.data_ptr()), n, vp(src.data_ptr()),
                               ctypes.c_float(-1.0), vp(d_rowmap.data_ptr()), None, vp(d_init.data_ptr()), None) == 0
    assert lib.ha_apply_mapped(vp(b1.data_ptr()), slots, width, vp(plan.ws.data_ptr()), n, vp(src.data_ptr()),
                               ctypes.c_float(-1.0), vp(d_rowmap2.data_ptr()), None, None, None) == 0
    a2, b2 = _dev(a0, dev), _dev(b0, dev)
    assert lib.ha_apply_mapped2(vp(a2.data_ptr()), slots, vp(b2.data_ptr()), width, vp(plan.ws.data_ptr()), n,
                                vp(src.data_ptr()), ctypes.c_float(-1.0), vp(d_rowmap.data_ptr()),
                                vp(d_rowmap2.data_ptr()), vp(d_init.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    assert not torch.equal(a1, _dev(a0, dev))


def test_random_shapes_sweep(dev):
    """40 random (n, width, key range, skew) configurations: lookup + plan + SGD apply + push apply
    through both the fused and the unfused entry points against the oracle."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        n = int(rng.choice([1, 2, 15, 16, 17, 63, 100, 257, 1000, 3333, 6656, 9000, 15360, 15361, 17000, 36864, 36865]))
        width = int(rng.choice([1, 2, 4, 5, 8, 20, 64, 68, 100, 128, 192, 256, 300]))
        rows = int(rng.choice([3, 50, 1000, 40000]))
        skew = float(rng.choice([0.0, 1.1, 2.0]))
        if skew == 0.0:
            ids = rng.integers(0, rows, size=n)
        else:
            ids = np.minimum(rng.zipf(skew, size=n) - 1, rows - 1)
        ids = ids.astype(np.float32)
        table = rng.standard_normal((rows, width), dtype=np.float32)
        grads = rng.standard_normal((n, width), dtype=np.float32)
        lr = float(rng.choice([1e-3, 0.5]))
        msg = "case %d: n=%d width=%d rows=%d skew=%.1f" % (case, n, width, rows, skew)
        want_out = cpu.embedding_lookup(table, ids)
        want_t = cpu.sgd_sparse_update(table.copy(), ids, grads, lr)
        uniq, inv, cnt = cpu.unique(cpu.ids_to_keys(ids))
        _, _, red = cpu.dedup_reduce(ids, grads)
        want_p = cpu.push_apply(table.copy(), uniq, red)
        d_ids, d_g = _dev(ids, dev), _dev(grads, dev)
        # fused
        t = _dev(table, dev)
        plan = ops.IndexPlan(n, dev)
        out = ops.lookup_sort(t, d_ids, plan)
        ops.sgd_apply_finish(t, plan, d_g, lr)
        np.testing.assert_array_equal(out.cpu().numpy(), want_out, err_msg=msg)
        np.testing.assert_array_equal(t.cpu().numpy(), want_t, err_msg=msg)
        assert plan.n_unique() == uniq.size, msg
        np.testing.assert_array_equal(plan.inverse().cpu().numpy().astype(np.int64), inv, err_msg=msg)
        np.testing.assert_array_equal(plan.counts(uniq.size).cpu().numpy().astype(np.int64), cnt, err_msg=msg)
        # unfused
        t2 = _dev(table, dev)
        plan2 = ops.IndexPlan(n, dev).build(d_ids)
        ops.sgd_apply(t2, plan2, d_g, lr)
        np.testing.assert_array_equal(t2.cpu().numpy(), want_t, err_msg=msg)
        t3 = _dev(table, dev)
        ops.push_apply(t3, plan2, d_g)
        np.testing.assert_array_equal(t3.cpu().numpy(), want_p, err_msg=msg)
        t4 = _dev(table, dev)
        ops.lookup_sort(t4, d_ids, plan)
        ops.push_apply_finish(t4, plan, d_g)
        np.testing.assert_array_equal(t4.cpu().numpy(), want_p, err_msg=msg)


def test_many_criteo_batches_through_the_fused_step(dev):
    """60 consecutive Criteo-shaped batches (bs=256, d=512) through the two fused launches, with the
    next-batch prefetch on, against the oracle after every step (rare interleavings of the cooperative
    long-run path would show up here)."""
    rows, width = 300000, 512
    rng = np.random.default_rng(99)
    table = rng.standard_normal((rows, width), dtype=np.float32)
    want = table.copy()
    t = _dev(table, dev)
    plan = ops.IndexPlan(6656, dev)
    batches = [np.minimum(synth.as_f32_ids(synth.criteo_batch(256, 1000 + k, rows=rows)).reshape(-1),
                          np.float32(rows - 1)) for k in range(61)]
    d_batches = [_dev(b, dev) for b in batches]
    for k in range(60):
        grads = rng.standard_normal((6656, width), dtype=np.float32)
        out = ops.lookup_sort(t, d_batches[k], plan)
        ops.sgd_apply_finish(t, plan, _dev(grads, dev), 0.05, next_ids=d_batches[k + 1])
        np.testing.assert_array_equal(out.cpu().numpy(), cpu.embedding_lookup(want, batches[k]), err_msg="step %d" % k)
        cpu.sgd_sparse_update(want, batches[k], grads, 0.05)
        touched = np.unique(batches[k].astype(np.int64))
        np.testing.assert_array_equal(t[torch.from_numpy(touched).to(dev)].cpu().numpy(), want[touched],
                                      err_msg="step %d" % k)
    np.testing.assert_array_equal(t.cpu().numpy(), want)


def test_sharded_checkpoint_round_trip_chunked(dev, tmp_path):
    """Embedding checkpoint `<name>_<part>.dat` (raw fp32 rows, PSFHandle.h:401-439) of a GPU shard larger
    than the staging buffer: streamed in 64 MiB chunks both ways, byte-identical to the numpy layout the
    reference's servers write, and loadable from a file written by numpy."""
    from herald_amd.sharded import ShardedEmbedding
    rows, width = 300_001, 128                          # 153.6 MB: three chunks, the last one ragged
    emb = ShardedEmbedding(rows, width, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    emb.table.normal_(0, 1, generator=g)
    ref = emb.table.cpu().numpy()
    emb.save(str(tmp_path / "emb"))
    path = str(tmp_path / "emb_0.dat")
    np.testing.assert_array_equal(np.fromfile(path, dtype=np.float32).reshape(rows, width), ref)
    emb.table.zero_()
    emb.load(str(tmp_path / "emb"))
    np.testing.assert_array_equal(emb.table.cpu().numpy(), ref)
    (ref * np.float32(2)).tofile(str(tmp_path / "other_0.dat"))      # a table written elsewhere
    emb.load(str(tmp_path / "other"))
    np.testing.assert_array_equal(emb.table.cpu().numpy(), ref * np.float32(2))
    open(str(tmp_path / "short_0.dat"), "wb").write(b"1234")
    with pytest.raises(ValueError):
        emb.load(str(tmp_path / "short"))


@pytest.mark.parametrize("n,limit,hi", [(18433, 33762577, 33762577), (20000, 1000, 1000), (26624, 33762577, 33762577),
                                        (36864, 1 << 20, 1 << 20), (40000, 5000, 9000), (106496, 33762577, 33762577),
                                        (262144, 1 << 31, 1 << 31), (262145, 33762577, 33762577), (30000, 7, 1)])
def test_bucket_sort_path_with_key_limit(dev, n, limit, hi):
    """ha_plan_*_lim: batches above ~12 k ids with a known key range take the bucket sort (one
    most-significant-digit scatter + rank-by-counting inside the ranges).  Same plan as np.unique / a stable
    argsort: tiny key ranges (every key its own bucket), keys at and beyond the limit (last bucket), all keys
    equal (one bucket streamed through LDS in chunks), Criteo-like skew, and the sizes around both ends."""
    rng = np.random.default_rng(n + limit)
    if hi == 1:
        keys = np.full(n, 5, dtype=np.int64)
    elif limit == 33762577:
        b = max(1, n // 26)
        keys = synth.criteo_batch(b, 3).reshape(-1)[:n]
        keys = np.concatenate([keys, rng.integers(0, hi, size=n - keys.size)])
    else:
        keys = rng.integers(0, hi, size=n)
    d_ids = _dev(keys.astype(np.int64), dev)
    plan = ops.IndexPlan(n, dev).build(d_ids, key_limit=limit)
    torch.cuda.synchronize()
    k64 = keys.astype(np.uint64)
    uniq, inv, cnt = cpu.unique(k64)
    assert plan.n_unique() == uniq.size
    perm = plan.perm().cpu().numpy()
    np.testing.assert_array_equal(perm, np.argsort(k64, kind="stable"))
    np.testing.assert_array_equal(_u32(plan.sorted_keys()), k64[perm])
    np.testing.assert_array_equal(_u32(plan.uniq()), uniq)
    np.testing.assert_array_equal(plan.inverse().cpu().numpy().astype(np.int64), inv)
    np.testing.assert_array_equal(plan.counts().cpu().numpy().astype(np.int64), cnt)
    # float32 ids through the fused forward (gather blocks ride in the first launch) + the SGD apply
    if limit == 33762577 and n <= 106496:
        rows, width = 60000, 32
        fid = (keys % rows).astype(np.float32)
        table = rng.standard_normal((rows, width), dtype=np.float32)
        g = rng.standard_normal((n, width), dtype=np.float32)
        t = _dev(table, dev)
        p2 = ops.IndexPlan(n, dev)
        out = ops.lookup_sort(t, _dev(fid, dev), p2)
        ops.sgd_apply_finish(t, p2, _dev(g, dev), 0.05)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), cpu.embedding_lookup(table, fid))
        np.testing.assert_array_equal(t.cpu().numpy(), cpu.sgd_sparse_update(table.copy(), fid, g, 0.05))


def test_plan_host_readers_wait_for_the_producing_stream(dev):
    """n_unique() / export_f32() / deduplicate() of a plan built on an explicit side stream (behind a long
    kernel on that stream) read the finished values, not stale ones, whatever torch's current stream is."""
    rng = np.random.default_rng(3)
    side = torch.cuda.Stream(device=dev)
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    plan = ops.IndexPlan(6656, dev)
    plan.build(_dev(np.zeros(6656, dtype=np.float32), dev))          # a first content: one unique key
    torch.cuda.synchronize()
    ids = rng.integers(0, 100000, size=6656).astype(np.float32)
    d_ids = _dev(ids, dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(8):
            big.add_(1.0)                                             # keeps the side stream busy for milliseconds
    plan.build(d_ids, stream=side)
    assert plan.n_unique() == np.unique(ids).size                     # current stream is NOT `side`
    uf, invf = plan.export_f32()
    ru, rinv = np.unique(ids, return_inverse=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(uf.cpu().numpy(), ru)
    np.testing.assert_array_equal(invf.cpu().numpy(), rinv.astype(np.float32))


@pytest.mark.parametrize("n,width", [(26624, 64), (50000, 32)])
def test_sort_ahead_schedule_equals_the_single_stream_sequence(dev, n, width):
    """ops.SortAhead: the sort of batch k+1 on a side stream beside lookup + apply of batch k -- same rows, same
    table, same finished plans as the calls on one stream (also under a captured graph)."""
    rng = np.random.default_rng(n)
    rows = 300000
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    steps = 4
    ids = [(synth.criteo_batch((n + 25) // 26, 40 + k).reshape(-1)[:n] % rows).astype(np.float32) for k in range(steps)]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(steps)]
    want = table0.copy()
    t = _dev(table0, dev)
    sa = ops.SortAhead(t, n, 0.05)
    d_ids = [_dev(i, dev) for i in ids]
    sa.begin(d_ids[0])
    for k in range(steps):
        out = sa.lookup(d_ids[k], d_ids[k + 1] if k + 1 < steps else None)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), cpu.embedding_lookup(want, ids[k]))
        cpu.sgd_sparse_update(want, ids[k], grads[k], 0.05)
        plan = sa.apply(_dev(grads[k], dev))
        torch.cuda.synchronize()
        uniq, inv, cnt = cpu.unique(cpu.ids_to_keys(ids[k]))
        assert plan.n_unique() == uniq.size
        np.testing.assert_array_equal(plan.inverse().cpu().numpy().astype(np.int64), inv)
    np.testing.assert_array_equal(t.cpu().numpy(), want)


@pytest.mark.parametrize("kind", ["f32", "u64"])
def test_batched_plans_by_the_one_workgroup_radix_sort(dev, kind):
    """ha_plan_build_batch_*_lim for batches of at most 8,192 ids: every batch is sorted by ONE workgroup (an LSD radix sort in
    LDS, csrc/plan.hip plan_sort_wg_batch_kernel) -- sorted keys, perm (positions ascending inside equal keys: np.argsort
    stable), unique keys, counts and inverse equal numpy's, for sizes around the group / chunk edges, key ranges from 5 bits to
    the pad value, and heavy duplication."""
    import ctypes
    from herald_amd import _lib, ops
    L = _lib.load()
    rng = np.random.default_rng(77)
    sizes = [1, 2, 63, 64, 65, 1023, 1024, 1025, 4097, 6656, 8191, 8192]
    cases = []
    for n in sizes:
        for hi in (17, 300, 33_762_577, (1 << 32) - 3):
            k = rng.integers(0, hi, size=n, dtype=np.uint64)
            if n > 8:
                k[: n // 4] = k[0]                      # a long run
                k[n // 3] = min(hi - 1, 0xFFFFFFFE)
            cases.append(k)
    for c0 in range(0, len(cases), 16):
        grp = cases[c0:c0 + 16]
        if kind == "f32":
            grp = [np.minimum(k, (1 << 24) - 1) for k in grp]          # float32 ids are exact below 2^24
            tens = [torch.from_numpy(k.astype(np.float32)).to(dev) for k in grp]
            fn = L.ha_plan_build_batch_f32ids_lim
        else:
            tens = [torch.from_numpy(k.astype(np.int64)).to(dev) for k in grp]
            fn = L.ha_plan_build_batch_u64ids_lim
        plans = [ops.IndexPlan(max(k.size, 1), dev) for k in grp]
        cnt = len(grp)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        _lib.check(fn((vp * cnt)(*[t.data_ptr() for t in tens]), (i64 * cnt)(*[k.size for k in grp]),
                      (vp * cnt)(*[p.ws.data_ptr() for p in plans]), cnt, ctypes.c_uint64(33_762_577), None), "plan_build_batch")
        torch.cuda.synchronize()
        for k, p in zip(grp, plans):
            p.n, p._view = k.size, None
            keys = np.minimum(k, 0xFFFFFFFE).astype(np.int64)
            order = np.argsort(keys, kind="stable")
            np.testing.assert_array_equal(p.perm().cpu().numpy().astype(np.int64), order)
            np.testing.assert_array_equal(p.sorted_keys().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, keys[order])
            u, inv, cn = np.unique(keys, return_inverse=True, return_counts=True)
            assert p.n_unique() == u.size
            np.testing.assert_array_equal(p.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, u)
            np.testing.assert_array_equal(p.counts().cpu().numpy().astype(np.int64)[:u.size], cn)
            np.testing.assert_array_equal(p.inverse().cpu().numpy().astype(np.int64), inv)


@pytest.mark.parametrize("kind", ["f32", "u64"])
def test_batched_plans_of_medium_batches_by_batched_radix_passes(dev, kind):
    """ha_plan_build_batch_*_lim for batches of 36,865 .. 262,144 ids: the radix passes and the two finish launches of ALL the
    batches of a call per launch (csrc/plan.hip plan_build_batch_radix: blockIdx.y names the sort) -- sorted keys, perm (stable),
    unique keys, counts, inverse and the list of long runs equal numpy's / the one-plan-at-a-time build's, for ragged sizes
    around the tile edges, Criteo-shaped keys with runs of thousands, and full 32-bit keys."""
    import ctypes
    from herald_amd import _lib, ops, synth
    L = _lib.load()
    rng = np.random.default_rng(78)
    rows = 33_762_577
    sizes = [36865, 40960, 40961, 106496, 65536, 106496, 100001, 262144, 53248]
    cases = []
    for j, n in enumerate(sizes):
        if j % 3 == 0:
            k = synth.criteo_batch((n + 25) // 26, 40 + j, rows=rows).reshape(-1)[:n].astype(np.uint64)
        elif j % 3 == 1:
            k = rng.integers(0, rows, size=n, dtype=np.uint64)
            k[: n // 5] = k[0]                         # a run of thousands
        else:
            k = rng.integers(0, (1 << 32) - 3 if kind == "u64" else (1 << 24) - 1, size=n, dtype=np.uint64)
        cases.append(k)
    if kind == "f32":
        cases = [np.minimum(k, (1 << 24) - 1) for k in cases]
        tens = [torch.from_numpy(k.astype(np.float32)).to(dev) for k in cases]
        fn = L.ha_plan_build_batch_f32ids_lim
    else:
        tens = [torch.from_numpy(k.astype(np.int64)).to(dev) for k in cases]
        fn = L.ha_plan_build_batch_u64ids_lim
    plans = [ops.IndexPlan(k.size, dev) for k in cases]
    cnt = len(cases)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    _lib.check(fn((vp * cnt)(*[t.data_ptr() for t in tens]), (i64 * cnt)(*[k.size for k in cases]),
                  (vp * cnt)(*[p.ws.data_ptr() for p in plans]), cnt, ctypes.c_uint64(rows), None), "plan_build_batch")
    torch.cuda.synchronize()
    for k, t, p in zip(cases, tens, plans):
        p.n, p._view = k.size, None
        keys = np.minimum(k, 0xFFFFFFFE).astype(np.int64)
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(p.perm().cpu().numpy().astype(np.int64), order)
        np.testing.assert_array_equal(p.sorted_keys().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, keys[order])
        u, inv, cn = np.unique(keys, return_inverse=True, return_counts=True)
        assert p.n_unique() == u.size
        np.testing.assert_array_equal(p.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, u)
        np.testing.assert_array_equal(p.counts().cpu().numpy().astype(np.int64)[:u.size], cn)
        np.testing.assert_array_equal(p.inverse().cpu().numpy().astype(np.int64), inv)
        # the same plan built alone: every array of the workspace's result equal
        q = ops.IndexPlan(k.size, dev)
        q.build(t, key_limit=rows)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(q.perm().cpu().numpy(), p.perm().cpu().numpy())
        np.testing.assert_array_equal(q.counts().cpu().numpy()[:u.size], p.counts().cpu().numpy()[:u.size])
