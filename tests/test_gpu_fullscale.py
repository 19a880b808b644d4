"""Full-scale addressing parity (BASELINE configs[1] as written): the real 33,762,577 x 512 fp32 table
(69.1 GB) in HBM, filled on the device by a closed form of (row, column); Criteo-shaped batches whose
float32 ids reach beyond 2^24 and whose rows lie beyond the 4 GiB and 64 GiB byte offsets go through
the fused forward / backward launches and through the HET cache at limit 0.1 x rows.  Only the touched
rows come back to the host; the oracle runs on a compact table of exactly those rows."""
import os
import sys

import numpy as np
import pytest
import torch

from herald_amd import cache as hcache
from herald_amd import ops, synth
from oracle import cache_model, cpu, qstep_model

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import formula  # noqa: E402

pytestmark = pytest.mark.gpu

ROWS, WIDTH, FIELDS, BATCH = 33762577, 512, 26, 256


def _need_big_gpu(dev):
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 100 * (1 << 30):
        pytest.skip("needs ~90 GB of free HBM")


@pytest.fixture(scope="module")
def big_table(dev):
    """formula.table(ROWS, WIDTH) computed on the device (no host copy): same integer arithmetic, then
    int -> float32 (exact below 2^24), one float32 multiply and one float32 subtract as separate ops."""
    _need_big_gpu(dev)
    t = torch.empty((ROWS, WIDTH), dtype=torch.float32, device=dev)
    c = torch.arange(WIDTH, dtype=torch.int64, device=dev)[None, :] * 40503
    step = 1 << 18
    for s in range(0, ROWS, step):
        e = min(ROWS, s + step)
        r = torch.arange(s, e, dtype=torch.int64, device=dev)[:, None] * 2654435761
        v = ((r + c) % 2000003).to(torch.float32)
        v = v * np.float32(1e-4)
        t[s:e] = v - np.float32(100.0)
    torch.cuda.synchronize()
    # spot check of the fill against the host formula, including the last row (offset 69.1 GB)
    probe = [0, 1, 2097153, 16777217, 33554433, ROWS - 1]
    got = t[torch.tensor(probe, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(got, formula.rows_of(probe, WIDTH))
    yield t
    del t
    torch.cuda.empty_cache()


def _batches(count, first=0):
    out = []
    for b in range(count):
        f = synth.as_f32_ids(synth.criteo_batch(BATCH, step=first + b, rows=ROWS, nfields=FIELDS)).reshape(-1)
        np.minimum(f, np.float32(ROWS - 1), out=f)
        out.append(f)
    return out


# odd rows above 2^24 are not representable as float32 ids: no batch can touch them
SENTINELS = [2097153, 4194305, 16777217, 20000001, 33554433, 33600001, ROWS - 2]


def test_fused_step_on_the_full_table(dev, big_table):
    table = big_table
    batches = _batches(8)
    keys_all = np.unique(np.concatenate([cpu.ids_to_keys(f) for f in batches]))
    assert keys_all.max() > (1 << 24) and keys_all.max() * WIDTH * 4 > (64 << 30)
    assert (keys_all * WIDTH * 4 > (4 << 30)).sum() > 1000
    compact = formula.rows_of(keys_all, WIDTH)                  # the touched rows, as the oracle's table
    lr = 0.01
    plan = ops.IndexPlan(BATCH * FIELDS, dev)
    rng = np.random.default_rng(11)
    d_ids = [torch.from_numpy(f).to(dev) for f in batches]
    for b, f in enumerate(batches):
        grads = rng.standard_normal((f.size, WIDTH), dtype=np.float32)
        cid = np.searchsorted(keys_all, cpu.ids_to_keys(f)).astype(np.float32)   # < 2^24: exact
        want_out = cpu.embedding_lookup(compact, cid)
        cpu.sgd_sparse_update(compact, cid, grads, lr)
        out = ops.lookup_sort(table, d_ids[b], plan)
        nxt = d_ids[b + 1] if b + 1 < len(batches) else None
        ops.sgd_apply_finish(table, plan, torch.from_numpy(grads).to(dev), lr, next_ids=nxt)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), want_out, err_msg="gather, batch %d" % b)
        u, inv, cnt = cpu.unique(cpu.ids_to_keys(f))
        assert plan.n_unique() == u.size
        np.testing.assert_array_equal((plan.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF), u.astype(np.int64))
        np.testing.assert_array_equal(plan.counts().cpu().numpy().astype(np.int64), cnt)
    got = table[torch.from_numpy(keys_all.astype(np.int64)).to(dev)].cpu().numpy()
    np.testing.assert_array_equal(got, compact, err_msg="updated rows")
    assert not np.array_equal(compact, formula.rows_of(keys_all, WIDTH))      # the updates were visible
    sent = table[torch.tensor(SENTINELS, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(sent, formula.rows_of(SENTINELS, WIDTH), err_msg="untouched sentinel rows")
    # put the touched rows back so that the next test starts from the closed form again
    table[torch.from_numpy(keys_all.astype(np.int64)).to(dev)] = torch.from_numpy(
        formula.rows_of(keys_all, WIDTH)).to(dev)
    torch.cuda.synchronize()


def _restore(table, keys_all, dev):
    table[torch.from_numpy(keys_all.astype(np.int64)).to(dev)] = torch.from_numpy(
        formula.rows_of(keys_all, WIDTH)).to(dev)
    torch.cuda.synchronize()


def _stream_inputs(count, first, seed):
    batches = _batches(count, first=first)
    keys_all = np.unique(np.concatenate([cpu.ids_to_keys(f) for f in batches]))
    assert keys_all.max() > (1 << 24) and keys_all.max() * WIDTH * 4 > (64 << 30)
    assert (keys_all * WIDTH * 4 > (4 << 30)).sum() > 1000
    rng = np.random.default_rng(seed)
    grads = [rng.standard_normal((f.size, WIDTH), dtype=np.float32) for f in batches]
    cids = [np.searchsorted(keys_all, cpu.ids_to_keys(f)).astype(np.float32) for f in batches]   # < 2^24: exact
    return batches, keys_all, grads, cids


def _check_table(table, keys_all, compact, dev):
    got = table[torch.from_numpy(keys_all.astype(np.int64)).to(dev)].cpu().numpy()
    np.testing.assert_array_equal(got, compact, err_msg="updated rows")
    assert not np.array_equal(compact, formula.rows_of(keys_all, WIDTH))      # the updates were visible
    sent = table[torch.tensor(SENTINELS, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(sent, formula.rows_of(SENTINELS, WIDTH), err_msg="untouched sentinel rows")


def test_one_launch_push_pull_step_on_the_full_table(dev, big_table):
    """ha_sgd_push_pull_f32ids = ha::step_kernel (bench.py --engine handoff; round 2's default: in-launch hand-off
    through the pending tables, its own row addressing and `sc1` loads) on the 69.1 GB table: an 8-batch Criteo stream,
    every output row of every step, the touched rows, the sentinels, the pending tables drained and no hand-off
    time-out -- rows beyond the 4 GiB / 64 GiB byte offsets, float32 ids above 2^24."""
    table = big_table
    batches, keys_all, grads, cids = _stream_inputs(8, first=300, seed=12)
    compact = formula.rows_of(keys_all, WIDTH)
    lr = 0.01
    n = BATCH * FIELDS
    plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
    pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
    d_ids = [torch.from_numpy(f).to(dev) for f in batches]
    out = ops.lookup_sort_pend(table, d_ids[0], plans[0], pends[0])
    for b in range(len(batches)):
        torch.cuda.synchronize()
        want_out = cpu.embedding_lookup(compact, cids[b])
        np.testing.assert_array_equal(out.cpu().numpy(), want_out, err_msg="rows of batch %d" % b)
        cpu.sgd_sparse_update(compact, cids[b], grads[b], lr)
        last = b + 1 == len(batches)
        out = ops.sgd_push_pull(table, plans[b % 2], torch.from_numpy(grads[b]).to(dev), lr, pends[b % 2],
                                None if last else d_ids[b + 1], None if last else plans[(b + 1) % 2],
                                None if last else pends[(b + 1) % 2])
        torch.cuda.synchronize()
        u, inv, cnt = cpu.unique(cpu.ids_to_keys(batches[b]))
        pl = plans[b % 2]
        assert pl.n_unique() == u.size
        np.testing.assert_array_equal(pl.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, u.astype(np.int64))
        np.testing.assert_array_equal(pl.counts().cpu().numpy().astype(np.int64), cnt)
        np.testing.assert_array_equal(pl.inverse().cpu().numpy().astype(np.int64), inv)
    assert not plans[0].handoff_timed_out() and not plans[1].handoff_timed_out()
    assert pends[0].is_idle() and pends[1].is_idle()
    _check_table(table, keys_all, compact, dev)
    _restore(table, keys_all, dev)


def test_lookahead_step_pipeline_on_the_full_table(dev, big_table):
    """ha_step_* (ops.StepPipeline: ids three batches ahead, rows forwarded from the applying waves) on the 69.1 GB
    table, same stream shape and checks as the push_pull step."""
    table = big_table
    batches, keys_all, grads, cids = _stream_inputs(9, first=400, seed=13)
    compact = formula.rows_of(keys_all, WIDTH)
    lr = 0.01
    pipe = ops.StepPipeline(table, BATCH * FIELDS, lr)
    d_ids = [torch.from_numpy(f).to(dev) for f in batches]
    B = len(batches)
    out = pipe.start(d_ids[0], d_ids[1], d_ids[2])
    for b in range(B):
        torch.cuda.synchronize()
        want_out = cpu.embedding_lookup(compact, cids[b])
        np.testing.assert_array_equal(out.cpu().numpy().reshape(-1, WIDTH), want_out, err_msg="rows of batch %d" % b)
        cpu.sgd_sparse_update(compact, cids[b], grads[b], lr)
        out = pipe.step(torch.from_numpy(grads[b]).to(dev), d_ids[b + 3] if b + 3 < B else None)
        torch.cuda.synchronize()
        u, inv, cnt = cpu.unique(cpu.ids_to_keys(batches[b]))
        pl = pipe.plan_of(b)
        assert pl.n_unique() == u.size
        np.testing.assert_array_equal(pl.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, u.astype(np.int64))
        np.testing.assert_array_equal(pl.counts().cpu().numpy().astype(np.int64), cnt)
    _check_table(table, keys_all, compact, dev)
    _restore(table, keys_all, dev)


@pytest.mark.parametrize("overlap,block,count,sync", [(True, 16, 20, "flags"), (True, 16, 20, "events"), (False, 1, 9, "events")],
                         ids=["as_bench_flags_block16", "events_block16", "one_stream"])
def test_queue_step_on_the_full_table(dev, big_table, overlap, block, count, sync):
    """THE KERNEL bench.py TIMES BY DEFAULT -- ha::qapply_kernel behind ops.QueueStepPipeline (csrc/qstep.hip) -- on the
    table bench.py uses: 33,762,577 x 512 fp32 (69.1 GB), Criteo batches of 6,656 float32 ids reaching beyond 2^24, rows
    beyond the 4 GiB and 64 GiB byte offsets.  As bench.py drives it (blocks of 16 prepared on a side stream, the two
    streams ordered by epoch tags in the queues and launch-borne events -- sync="flags"; the stream of 20 batches crosses
    a block boundary), with the event pair on the caller's stream instead, and in the serial form.  Every lookup row of every step
      * bit for bit against oracle/qstep_model.py (the kernel's floating-point order), and
      * against the reference's serial chain (oracle/cpu.py, pinned to the compiled cpu_SGDOptimizerSparseUpdate,
        /root/reference/src/dnnl_ops/Optimizers.cpp:51-74; lookups: EmbeddingLookup.cpp:16-35): bit-exact for every key
        that never had 16 or more occurrences in a batch, within 1e-5 x lr x sum|g| (BASELINE.json) for the others;
    the plans group by group against np.unique, the touched rows and the sentinels at the end.  Prints the observed
    maximum of |tree - chain| / (lr x sum|g|) over the tolerance-class rows."""
    table = big_table
    REL = 1e-5
    batches, keys_all, grads, cids = _stream_inputs(count, first=(500 if overlap else 600) + (40 if sync == "flags" else 0),
                                                    seed=14 + block)
    n = BATCH * FIELDS
    model_t = formula.rows_of(keys_all, WIDTH)        # kernel order, on the compact table of the touched rows
    exact_t = model_t.copy()                          # the reference's chain
    lr = 0.01
    drift = {}                                        # compact row -> accumulated tolerance (float64 per column)
    worst = worst_g = 0.0
    pipe = ops.QueueStepPipeline(table, n, lr, block=block, overlap=overlap, sync=sync)
    L = pipe.LOOKAHEAD
    d_ids = [torch.from_numpy(f).to(dev) for f in batches]
    out = pipe.start(d_ids[:L])
    for b in range(count):
        torch.cuda.synchronize()
        ci = cids[b].astype(np.int64)
        got = out.cpu().numpy().reshape(-1, WIDTH)
        np.testing.assert_array_equal(got, model_t[ci], err_msg="rows of batch %d (kernel order)" % b)
        want = exact_t[ci]
        loose = np.array([int(c) in drift for c in ci])
        np.testing.assert_array_equal(got[~loose], want[~loose], err_msg="rows of batch %d without a long run: bit-exact" % b)
        for j in np.flatnonzero(loose):
            d = np.abs(got[j].astype(np.float64) - want[j].astype(np.float64))
            assert (d <= drift[int(ci[j])] + REL * np.abs(want[j])).all(), "batch %d position %d off by %g" % (b, j, d.max())
        qstep_model.sgd_sparse_update(model_t, ci, grads[b], lr)
        cpu.sgd_sparse_update(exact_t, cids[b], grads[b], lr)
        for key, bound in qstep_model.tolerance(ci, grads[b], lr, keys_all.size, REL).items():
            if bound.any():
                drift[key] = drift.get(key, 0) + bound
        out = pipe.step(torch.from_numpy(grads[b]).to(dev), d_ids[b + L] if b + L < count else None)
        torch.cuda.synchronize()
        # the plan of batch b: np.unique's relations with the groups in hash-slot order
        pl = pipe.plan_of(b)
        keys = cpu.ids_to_keys(batches[b]).astype(np.int64)
        u, cnt = np.unique(keys, return_counts=True)
        assert pl.n_unique() == u.size
        got_u = pl.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        order = np.argsort(got_u)
        np.testing.assert_array_equal(got_u[order], u, err_msg="unique keys of batch %d" % b)
        np.testing.assert_array_equal(pl.counts().cpu().numpy().astype(np.int64)[order], cnt, err_msg="dedup counts of batch %d" % b)
        np.testing.assert_array_equal(got_u[pl.inverse().cpu().numpy().astype(np.int64)], keys, err_msg="inverse of batch %d" % b)
        perm = pl.perm().cpu().numpy().astype(np.int64)
        seg = pl.seg().cpu().numpy().astype(np.int64)
        np.testing.assert_array_equal(keys[perm], np.repeat(got_u, pl.counts().cpu().numpy()), err_msg="occurrence lists of batch %d" % b)
        inside = np.ones(keys.size, bool)
        inside[seg[:u.size]] = False
        assert (np.diff(perm)[inside[1:]] > 0).all(), "occurrence order of batch %d" % b
    assert out is None and not pipe.overflowed()
    got_t = table[torch.from_numpy(keys_all.astype(np.int64)).to(dev)].cpu().numpy()
    np.testing.assert_array_equal(got_t, model_t, err_msg="touched rows (kernel order)")
    loose = np.zeros(keys_all.size, bool)
    loose[list(drift.keys())] = True
    assert loose.any() and not loose.all()
    np.testing.assert_array_equal(got_t[~loose], exact_t[~loose], err_msg="touched rows without a long run: the serial chain")
    for j in np.flatnonzero(loose):
        d = np.abs(got_t[j].astype(np.float64) - exact_t[j].astype(np.float64))
        assert (d <= drift[j] + REL * np.abs(exact_t[j])).all(), "row %d off by %g" % (keys_all[j], d.max())
        worst = max(worst, float((d / (drift[j] / REL + np.abs(exact_t[j]))).max()))
        worst_g = max(worst_g, float((d / np.maximum(drift[j] / REL, 1e-30)).max()))
    # (the rows of this table are ~1e2 and lr x sum|g| ~1e-1: one ulp of a row is already 6e-5 of the accumulated
    # gradient, so the figure relative to the gradient alone is dominated by the final subtraction's rounding)
    print("queue step on the full table (%s): %d tolerance-class rows, max |tree - chain| / (lr * sum|g| + |row|) = %.3g "
          "(bound %.0e); relative to lr * sum|g| alone: %.3g" % ("block %d" % block if overlap else "serial",
                                                                 int(loose.sum()), worst, REL, worst_g))
    assert worst <= REL
    assert not np.array_equal(model_t, formula.rows_of(keys_all, WIDTH))
    sent = table[torch.tensor(SENTINELS, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(sent, formula.rows_of(SENTINELS, WIDTH), err_msg="untouched sentinel rows")
    del pipe
    _restore(table, keys_all, dev)


def _fill_formula(t, width, dev):
    """formula.table(rows, width) on the device, as the big_table fixture does it."""
    rows = t.shape[0]
    c = torch.arange(width, dtype=torch.int64, device=dev)[None, :] * 40503
    step = 1 << 20
    for s in range(0, rows, step):
        e = min(rows, s + step)
        r = torch.arange(s, e, dtype=torch.int64, device=dev)[:, None] * 2654435761
        v = ((r + c) % 2000003).to(torch.float32)
        v = v * np.float32(1e-4)
        t[s:e] = v - np.float32(100.0)


@pytest.mark.parametrize("bs,width", [(1024, 512), (4096, 128)], ids=["configs3_26624ids_d512", "configs2_106496ids_d128"])
def test_wide_queue_step_at_the_bench_shapes(dev, big_table, bs, width):
    """THE WIDE PATH AT THE SHAPES bench.py TIMES IT AT (`wide_bs1024_d512`, `wide_bs4096_d128`): batches of 26,624 ids on the
    33,762,577 x 512 table and of 106,496 ids on a 33,762,577 x 128 table (17.3 GB) -- chunked workgroup items x multi-slice rows
    (a key with thousands of occurrences, every 64-column slice of its row), row offsets beyond 4 GiB and 64 GiB, float32 ids
    beyond 2^24 --, six Criteo batches in blocks of four (the stream crosses a block boundary), the two streams ordered by
    epoch tags (sync="flags", as the bench drives it).  The touched rows start from N(0, 0.01) values (the reference's
    init.random_normal(stddev=0.01), wdl_criteo.py:13), so the tolerance classes are held against 1e-5 x lr x sum|g| WITHOUT
    a |row| ~ 100 term beside it.  Every lookup row of every step and the touched rows at the end
      * bit for bit against oracle/qstep_model.py (the kernel's floating-point order, chunk sums included), and
      * against the reference's serial chain (oracle/cpu.py = the compiled cpu_SGDOptimizerSparseUpdate,
        /root/reference/src/dnnl_ops/Optimizers.cpp:51-74; lookups /root/reference/src/dnnl_ops/EmbeddingLookup.cpp:16-35): rows
        whose key never had 16+ occurrences in a batch bit-exact, the others within 1e-5 x (lr x sum|g| + |row|);
    the plans bucket by bucket against np.unique (cpu_deduplicate's contract, /root/reference/python/hetu/ndarray.py:556-576)."""
    from test_gpu_qstep import _check_wide_plan
    REL = 1e-5
    steps, block, lr = 6, 4, 0.01
    if width == WIDTH:
        table = big_table
    else:
        table = torch.empty((ROWS, width), dtype=torch.float32, device=dev)
        _fill_formula(table, width, dev)
    batches = []
    for b in range(steps):
        f = synth.as_f32_ids(synth.criteo_batch(bs, step=700 + b, rows=ROWS, nfields=FIELDS)).reshape(-1)
        np.minimum(f, np.float32(ROWS - 1), out=f)
        batches.append(f)
    n = batches[0].size
    assert n == bs * FIELDS and n > ops.qstep_max_ids()
    keys = [cpu.ids_to_keys(f).astype(np.int64) for f in batches]
    keys_all = np.unique(np.concatenate(keys))
    assert keys_all.max() > (1 << 24) and keys_all.max() * width * 4 > ((64 << 30) if width == 512 else (16 << 30))
    assert (keys_all * width * 4 > (4 << 30)).sum() > 1000
    rng = np.random.default_rng(bs + width)
    start_rows = (rng.standard_normal((keys_all.size, width), dtype=np.float32) * np.float32(0.01))
    d_keys = torch.from_numpy(keys_all).to(dev)
    table[d_keys] = torch.from_numpy(start_rows).to(dev)
    model_t, exact_t = start_rows.copy(), start_rows.copy()
    cids = [np.searchsorted(keys_all, k) for k in keys]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(steps)]
    drift = np.zeros((keys_all.size, width), np.float64)
    pipe = ops.QueueStepPipeline(table, n, lr, overlap=True, block=block, sync="flags")
    assert pipe.wide
    d_ids = [torch.from_numpy(f).to(dev) for f in batches]
    out = pipe.start(d_ids[:pipe.LOOKAHEAD])
    chunked = 0
    for b in range(steps):
        torch.cuda.synchronize()
        ci = cids[b]
        got = out.cpu().numpy().reshape(-1, width)
        np.testing.assert_array_equal(got, model_t[ci], err_msg="rows of batch %d (kernel order)" % b)
        d = np.abs(got.astype(np.float64) - exact_t[ci].astype(np.float64))
        assert (d <= drift[ci] + REL * np.abs(exact_t[ci])).all(), "rows of batch %d vs the serial chain" % b
        loose = drift[ci].any(axis=1)
        np.testing.assert_array_equal(got[~loose], exact_t[ci][~loose], err_msg="rows of batch %d without a long run: bit-exact" % b)
        qstep_model.sgd_sparse_update(model_t, ci, grads[b], lr)
        cpu.sgd_sparse_update(exact_t, ci.astype(np.float32), grads[b], lr)
        cnt = np.bincount(ci, minlength=keys_all.size)
        chunked += int((cnt > qstep_model.CHUNK).sum())
        sumabs = np.zeros((keys_all.size, width), np.float64)
        np.add.at(sumabs, ci, np.abs(grads[b]).astype(np.float64))
        drift += np.where((cnt >= qstep_model.LONG_MIN)[:, None], REL * lr * sumabs, 0.0)
        out = pipe.step(torch.from_numpy(grads[b]).to(dev), None)
        torch.cuda.synchronize()
        assert not pipe.overflowed() and pipe.fallbacks == 0
        _check_wide_plan(pipe.plan_of(b), keys[b], "plan of batch %d" % b)
    assert out is None
    if bs == 4096:
        assert chunked > 0          # keys with more than 256 occurrences: chunk sums x every slice of a 128-column row
    got_t = table[d_keys].cpu().numpy()
    np.testing.assert_array_equal(got_t, model_t, err_msg="touched rows (kernel order)")
    loose = drift.any(axis=1)
    assert loose.any() and not loose.all()
    np.testing.assert_array_equal(got_t[~loose], exact_t[~loose], err_msg="touched rows without a long run: the serial chain")
    d = np.abs(got_t.astype(np.float64) - exact_t.astype(np.float64))
    assert (d <= drift + REL * np.abs(exact_t)).all()
    worst = float((d[loose] / (drift[loose] / REL + np.abs(exact_t[loose]))).max())
    print("wide queue step, bs %d d %d: %d tolerance-class rows, max |tree - chain| / (lr * sum|g| + |row|) = %.3g (bound %.0e; "
          "rows ~1e-2, so the |row| term is negligible)" % (bs, width, int(loose.sum()), worst, REL))
    sents = [k for k in SENTINELS if k not in set(keys_all.tolist())]
    sent = table[torch.tensor(sents, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(sent, formula.rows_of(sents, width), err_msg="untouched sentinel rows")
    del pipe
    if width == WIDTH:
        _restore(table, keys_all, dev)
    else:
        del table
        torch.cuda.empty_cache()


class _LazyRows:
    """Server table of the cache model holding only the rows that were written; the rest is the closed form."""

    def __init__(self, width):
        self.width, self.rows = width, {}

    def __getitem__(self, k):
        k = int(k)
        r = self.rows.get(k)
        return r if r is not None else formula.rows_of([k], self.width)[0]

    def __setitem__(self, k, v):
        self.rows[int(k)] = np.asarray(v, dtype=np.float32)


class _LazyVer(dict):
    def __missing__(self, k):
        return 0


class _LazyServer(cache_model.Server):
    def __init__(self, width):
        self.table = _LazyRows(width)
        self.ver = _LazyVer()


@pytest.mark.parametrize("pull_bound,push_bound", [(2, 2)])
def test_lru_cache_at_limit_0_1_on_the_full_table(dev, big_table, pull_bound, push_bound):
    """configs[1]'s cache tier as written: LRU, limit = 0.1 x rows = 3,376,257 lines, length 33,762,577,
    6,656 keys per batch; rows, perf counters and server versions every step, resident lines at the end."""
    table = big_table
    limit = int(0.1 * ROWS)
    n = BATCH * FIELDS
    server = _LazyServer(WIDTH)
    model = cache_model.CacheModel("lru", limit, WIDTH, server, pull_bound, push_bound)
    versions = torch.zeros(ROWS, dtype=torch.int64, device=dev)
    gpu = hcache.LRUCache(limit, ROWS, WIDTH, node_id=0, max_batch=n, device=dev)
    gpu.bind_store(table, versions)
    gpu.pull_bound, gpu.push_bound = pull_bound, push_bound
    gpu.perf_enabled = True
    rng = np.random.default_rng(21)
    batches = _batches(10, first=100)
    for step, f in enumerate(batches):
        want = model.lookup(cpu.ids_to_keys(f))
        dest = torch.empty((n, WIDTH), dtype=torch.float32, device=dev)
        kt = torch.from_numpy(f).to(dev)
        gpu.embedding_lookup(kt, dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="lookup rows at step %d" % step)
        grads = (rng.standard_normal((n, WIDTH), dtype=np.float32) * np.float32(-0.01))
        model.update(cpu.ids_to_keys(f), grads)
        if step % 2:        # what bench.py's cache_tier leg times: the update names the lookup's key tensor (two launches)
            gpu.embedding_update(kt, torch.from_numpy(grads).to(dev), same_as_lookup=True).wait()
        else:
            gpu.embedding_update(torch.from_numpy(f).to(dev), torch.from_numpy(grads).to(dev)).wait()
        for got, exp in zip(gpu.perf[-2:], model.perf[-2:]):
            for fld in ("type", "num_all", "num_unique", "num_miss", "num_transfered", "is_full"):
                assert got[fld] == exp[fld], (step, fld, got, exp)
    touched = sorted(server.ver.keys())
    tv = versions[torch.tensor(touched, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(tv, np.array([server.ver[k] for k in touched], dtype=np.int64))
    assert int(versions.sum().item()) == sum(server.ver.values())          # nothing else was versioned
    written = sorted(server.table.rows.keys())
    if written:
        got = table[torch.tensor(written, device=dev)].cpu().numpy()
        np.testing.assert_array_equal(got, np.stack([server.table.rows[k] for k in written]),
                                      err_msg="server rows after pushes")
    res, lines = model.resident(), gpu.lines()
    assert sorted(lines.keys()) == sorted(res.keys())
    for k, ln in res.items():
        g = lines[k]
        assert g.version == ln.version and g.updates == ln.updates, k
        np.testing.assert_array_equal(g.data, ln.data, err_msg="data of key %d" % k)
        if ln.grad is not None:
            np.testing.assert_array_equal(g.grad, ln.grad, err_msg="grad of key %d" % k)
    sent = table[torch.tensor(SENTINELS, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(sent, formula.rows_of(SENTINELS, WIDTH))
    assert gpu.size() == model.policy.size()
    assert int(gpu._L.ha_cache_fused_updates(gpu._h)) == len(batches) // 2
    if written:        # the pushed rows back to the closed form: the next test starts from it
        _restore(table, np.array(written, dtype=np.int64), dev)


def test_lru_cache_planned_flow_at_limit_0_1_on_the_full_table(dev, big_table):
    """The same tier through the PLANNED flow (csrc/cache_block.hip; what bench.py's cache_tier line times): blocks of 4
    batches, the next block's bookkeeping planned while this block's rows move; rows, perf counters and server versions every
    step, resident lines at the end."""
    table = big_table
    limit = int(0.1 * ROWS)
    n = BATCH * FIELDS
    server = _LazyServer(WIDTH)
    model = cache_model.CacheModel("lru", limit, WIDTH, server, 2, 2)
    versions = torch.zeros(ROWS, dtype=torch.int64, device=dev)
    gpu = hcache.LRUCache(limit, ROWS, WIDTH, node_id=0, max_batch=n, device=dev)
    gpu.bind_store(table, versions)
    gpu.pull_bound, gpu.push_bound = 2, 2
    gpu.perf_enabled = True
    rng = np.random.default_rng(22)
    batches = _batches(10, first=300)
    kts = [torch.from_numpy(f).to(dev) for f in batches]
    blocks = [list(range(b0, min(b0 + 4, len(batches)))) for b0 in range(0, len(batches), 4)]
    gpu.plan_block([kts[i] for i in blocks[0]])
    for j, blk in enumerate(blocks):
        if j + 1 < len(blocks):
            gpu.plan_block([kts[i] for i in blocks[j + 1]])
        for step in blk:
            f = batches[step]
            want = model.lookup(cpu.ids_to_keys(f))
            dest = torch.empty((n, WIDTH), dtype=torch.float32, device=dev)
            gpu.embedding_lookup_planned(dest).wait()
            np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="lookup rows at step %d" % step)
            grads = (rng.standard_normal((n, WIDTH), dtype=np.float32) * np.float32(-0.01))
            model.update(cpu.ids_to_keys(f), grads)
            gpu.embedding_update_planned(torch.from_numpy(grads).to(dev)).wait()
            for got, exp in zip(gpu.perf[-2:], model.perf[-2:]):
                for fld in ("type", "num_all", "num_unique", "num_miss", "num_transfered", "is_full"):
                    assert got[fld] == exp[fld], (step, fld, got, exp)
    assert gpu.plan_pending() == 0
    touched = sorted(server.ver.keys())
    tv = versions[torch.tensor(touched, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(tv, np.array([server.ver[k] for k in touched], dtype=np.int64))
    assert int(versions.sum().item()) == sum(server.ver.values())
    written = sorted(server.table.rows.keys())
    if written:
        got = table[torch.tensor(written, device=dev)].cpu().numpy()
        np.testing.assert_array_equal(got, np.stack([server.table.rows[k] for k in written]), err_msg="server rows after pushes")
    res, lines = model.resident(), gpu.lines()
    assert sorted(lines.keys()) == sorted(res.keys())
    for k, ln in res.items():
        g = lines[k]
        assert g.version == ln.version and g.updates == ln.updates, k
        np.testing.assert_array_equal(g.data, ln.data, err_msg="data of key %d" % k)
        if ln.grad is not None:
            np.testing.assert_array_equal(g.grad, ln.grad, err_msg="grad of key %d" % k)
    assert gpu.size() == model.policy.size()
    if written:
        _restore(table, np.array(written, dtype=np.int64), dev)


def test_cold_tier_on_an_8gib_pinned_host_table(dev):
    """BASELINE configs[4]'s mechanism at a size beyond 8 GiB: a 33,554,432 x 64 fp32 table (8 GiB) in PINNED
    HOST memory behind an LRU hot tier in HBM (HostStore + remote-store protocol).  The host table is a closed
    form of (row, column); only the touched rows are compared, against oracle/cache_model.py."""
    from herald_amd import remote_store
    rows, width, n = 33554432, 64, 22 * 256
    table = torch.empty((rows, width), dtype=torch.float32, pin_memory=True)
    step = 1 << 20
    for s in range(0, rows, step):
        e = min(rows, s + step)
        table[s:e] = torch.from_numpy(formula.rows_of(np.arange(s, e), width))
    store = remote_store.HostStore(rows, width, dev, table=table)
    limit = int(0.1 * rows)
    server = _LazyServer(width)
    model = cache_model.CacheModel("lru", limit, width, server, 2, 2)
    gpu = hcache.LRUCache(limit, rows, width, node_id=0, max_batch=n, device=dev)
    gpu.bind_remote(store)
    gpu.pull_bound = gpu.push_bound = 2
    gpu.perf_enabled = True
    rng = np.random.default_rng(31)
    for k in range(8):
        ids = synth.criteo_batch(256, 700 + k, rows=rows, nfields=22).reshape(-1)      # int64 keys
        want = model.lookup(ids.astype(np.uint64))
        dest = torch.empty((n, width), dtype=torch.float32, device=dev)
        gpu.embedding_lookup(torch.from_numpy(ids).to(dev), dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="lookup rows at step %d" % k)
        g = rng.standard_normal((n, width), dtype=np.float32) * np.float32(-0.01)
        model.update(ids.astype(np.uint64), g)
        gpu.embedding_update(torch.from_numpy(ids).to(dev), torch.from_numpy(g).to(dev)).wait()
        for got, exp in zip(gpu.perf[-2:], model.perf[-2:]):
            for fld in ("type", "num_all", "num_unique", "num_miss", "num_transfered"):
                assert got[fld] == exp[fld], (k, fld, got, exp)
    torch.cuda.synchronize()
    written = sorted(server.table.rows.keys())
    assert written and max(written) * width * 4 > (4 << 30)                     # rows beyond the 4 GiB offset
    np.testing.assert_array_equal(table[torch.tensor(written)].numpy(), np.stack([server.table.rows[k] for k in written]),
                                  err_msg="host rows after pushes")
    touched = sorted(server.ver.keys())
    np.testing.assert_array_equal(store.versions[torch.tensor(touched, device=dev)].cpu().numpy(),
                                  np.array([server.ver[k] for k in touched], dtype=np.int64))
    probe = [1, rows // 2 + 1, rows - 1]
    probe = [p for p in probe if p not in set(written)]
    np.testing.assert_array_equal(table[torch.tensor(probe)].numpy(), formula.rows_of(probe, width))
    tr = store.traffic()
    assert 0 < tr["rows_pulled"] <= tr["keys_synced"] and tr["lines_pushed"] > 0
    # ---- the same store addressed DIRECTLY (what bench.py's cold_tier.planned times): a second cache bound to the pinned
    # table and the HBM versions, six more batches through the planned flow -- pulls and pushes cross PCIe inside the lookup's /
    # the update's one launch; the server model carries on from the state the first cache's pushes left
    model2 = cache_model.CacheModel("lru", limit, width, server, 2, 2)
    gpu2 = hcache.LRUCache(limit, rows, width, node_id=0, max_batch=n, device=dev)
    gpu2.bind_store(table, store.versions)
    gpu2.pull_bound = gpu2.push_bound = 2
    gpu2.perf_enabled = True
    idl = [synth.criteo_batch(256, 720 + k, rows=rows, nfields=22).reshape(-1) for k in range(6)]
    idl[3] = idl[0].copy()                      # a batch again: hits, lines pushed by their update counters
    kts = [torch.from_numpy(i).to(dev) for i in idl]
    gpu2.plan_block(kts[:3])
    gpu2.plan_block(kts[3:])
    for k in range(6):
        want = model2.lookup(idl[k].astype(np.uint64))
        dest = torch.empty((n, width), dtype=torch.float32, device=dev)
        gpu2.embedding_lookup_planned(dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="planned lookup rows at step %d" % k)
        g = rng.standard_normal((n, width), dtype=np.float32) * np.float32(-0.01)
        model2.update(idl[k].astype(np.uint64), g)
        gpu2.embedding_update_planned(torch.from_numpy(g).to(dev)).wait()
        for got, exp in zip(gpu2.perf[-2:], model2.perf[-2:]):
            for fld in ("type", "num_all", "num_unique", "num_miss", "num_transfered"):
                assert got[fld] == exp[fld], (k, fld, got, exp)
    torch.cuda.synchronize()
    written = sorted(server.table.rows.keys())
    np.testing.assert_array_equal(table[torch.tensor(written)].numpy(), np.stack([server.table.rows[k] for k in written]),
                                  err_msg="host rows after the planned pushes")
    touched = sorted(server.ver.keys())
    np.testing.assert_array_equal(store.versions[torch.tensor(touched, device=dev)].cpu().numpy(),
                                  np.array([server.ver[k] for k in touched], dtype=np.int64))



def _host_budget_bytes():
    """Bytes of host DRAM this test may pin: 40 % of MemAvailable, at most 272 GB."""
    avail = 0
    with open("/proc/meminfo") as fh:
        for line in fh:
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    return min(int(avail * 0.4), 272 << 30)


def _fill_host_table(table, dev):
    """formula.rows_of for every row, computed on the GPU chunk by chunk and copied into the pinned table."""
    rows, width = table.shape
    c = torch.arange(width, dtype=torch.int64, device=dev)[None, :] * 40503
    step = max(1, (1 << 26) // width)
    for s in range(0, rows, step):
        e = min(rows, s + step)
        r = torch.arange(s, e, dtype=torch.int64, device=dev)[:, None] * 2654435761
        v = ((r + c) % 2000003).to(torch.float32)
        v = v * np.float32(1e-4)
        table[s:e].copy_(v - np.float32(100.0), non_blocking=True)
    torch.cuda.synchronize()


class _Pinned:
    """rows x width float32 in page-locked host memory taken straight from hipHostMalloc and given back with
    hipHostFree: torch's pinned allocator rounds requests up to a power of two and keeps freed blocks, which at
    these sizes would pin up to twice the budget."""

    def __init__(self, rows, width):
        import ctypes
        self.hip = hcache._hip()
        self.hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        self.hip.hipHostFree.argtypes = [ctypes.c_void_p]
        self.ptr = ctypes.c_void_p()
        nbytes = rows * width * 4
        rc = self.hip.hipHostMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(nbytes), 0)
        if rc != 0 or not self.ptr.value:
            pytest.skip("hipHostMalloc of %d GB failed (%d)" % (nbytes >> 30, rc))
        buf = (ctypes.c_float * (rows * width)).from_address(self.ptr.value)
        self.tensor = torch.from_numpy(np.ctypeslib.as_array(buf).reshape(rows, width))

    def free(self):
        self.tensor = None
        torch.cuda.synchronize()
        if self.ptr.value:
            self.hip.hipHostFree(self.ptr)
            self.ptr.value = None


def _cold_tier_run(dev, rows, width, steps, seed, first):
    pin = _Pinned(rows, width)
    try:
        return _cold_tier_body(dev, pin.tensor, rows, width, steps, seed, first)
    finally:
        pin.free()


def _cold_tier_body(dev, table, rows, width, steps, seed, first):
    from herald_amd import remote_store
    n = 22 * 256
    assert table.is_pinned()
    _fill_host_table(table, dev)
    probe0 = [0, 1, rows // 3, rows - 1]
    np.testing.assert_array_equal(table[torch.tensor(probe0)].numpy(), formula.rows_of(probe0, width))
    store = remote_store.HostStore(rows, width, dev, table=table)
    limit = int(0.1 * rows)
    server = _LazyServer(width)
    model = cache_model.CacheModel("lru", limit, width, server, 2, 2)
    gpu = hcache.LRUCache(limit, rows, width, node_id=0, max_batch=n, device=dev)
    gpu.bind_remote(store)
    gpu.pull_bound = gpu.push_bound = 2
    gpu.perf_enabled = True
    rng = np.random.default_rng(seed)
    top = 0
    for k in range(steps):
        ids = synth.criteo_batch(256, first + k, rows=rows, nfields=22).reshape(-1)      # int64 keys (u64 entry points)
        top = max(top, int(ids.max()))
        want = model.lookup(ids.astype(np.uint64))
        dest = torch.empty((n, width), dtype=torch.float32, device=dev)
        gpu.embedding_lookup(torch.from_numpy(ids).to(dev), dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="lookup rows at step %d" % k)
        g = rng.standard_normal((n, width), dtype=np.float32) * np.float32(-0.01)
        model.update(ids.astype(np.uint64), g)
        gpu.embedding_update(torch.from_numpy(ids).to(dev), torch.from_numpy(g).to(dev)).wait()
        for got, exp in zip(gpu.perf[-2:], model.perf[-2:]):
            for fld in ("type", "num_all", "num_unique", "num_miss", "num_transfered"):
                assert got[fld] == exp[fld], (k, fld, got, exp)
    torch.cuda.synchronize()
    written = sorted(server.table.rows.keys())
    assert written
    np.testing.assert_array_equal(table[torch.tensor(written)].numpy(), np.stack([server.table.rows[k] for k in written]),
                                  err_msg="host rows after pushes")
    touched = sorted(server.ver.keys())
    np.testing.assert_array_equal(store.versions[torch.tensor(touched, device=dev)].cpu().numpy(),
                                  np.array([server.ver[k] for k in touched], dtype=np.int64))
    assert int(store.versions.sum().item()) == sum(server.ver.values())
    probe = [p for p in (1, rows // 2 + 1, rows - 2) if p not in set(written)]
    np.testing.assert_array_equal(table[torch.tensor(probe)].numpy(), formula.rows_of(probe, width))
    res = model.resident()
    assert sorted(int(k) for k in gpu.keys()) == sorted(res.keys())
    assert gpu.size() == model.policy.size()
    versions = store.versions
    del gpu, store
    torch.cuda.synchronize()
    # ---- the same store addressed directly, through the planned flow (bench.py's cold_tier.planned): three more batches
    model2 = cache_model.CacheModel("lru", limit, width, server, 2, 2)
    gpu2 = hcache.LRUCache(limit, rows, width, node_id=0, max_batch=n, device=dev)
    gpu2.bind_store(table, versions)
    gpu2.pull_bound = gpu2.push_bound = 2
    gpu2.perf_enabled = True
    idl = [synth.criteo_batch(256, first + steps + k, rows=rows, nfields=22).reshape(-1) for k in range(2)]
    idl.append(idl[0].copy())
    gpu2.plan_block([torch.from_numpy(i).to(dev) for i in idl])
    for k in range(3):
        want = model2.lookup(idl[k].astype(np.uint64))
        dest = torch.empty((n, width), dtype=torch.float32, device=dev)
        gpu2.embedding_lookup_planned(dest).wait()
        np.testing.assert_array_equal(dest.cpu().numpy(), want, err_msg="planned lookup rows at step %d" % k)
        g = rng.standard_normal((n, width), dtype=np.float32) * np.float32(-0.01)
        model2.update(idl[k].astype(np.uint64), g)
        gpu2.embedding_update_planned(torch.from_numpy(g).to(dev)).wait()
        for got, exp in zip(gpu2.perf[-2:], model2.perf[-2:]):
            for fld in ("type", "num_all", "num_unique", "num_miss", "num_transfered"):
                assert got[fld] == exp[fld], (k, fld, got, exp)
    torch.cuda.synchronize()
    written2 = sorted(server.table.rows.keys())
    np.testing.assert_array_equal(table[torch.tensor(written2)].numpy(), np.stack([server.table.rows[k] for k in written2]),
                                  err_msg="host rows after the planned pushes")
    assert int(versions.sum().item()) == sum(server.ver.values())
    assert sorted(int(k) for k in gpu2.keys()) == sorted(model2.resident().keys())
    del gpu2, versions
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return top, max(written)


def test_cold_tier_at_one_billion_rows(dev):
    """BASELINE configs[4] at its ROW COUNT: length = 1e9 (direct map 4 GB, versions 8 GB in HBM, LRU hot tier of
    1e8 lines), uint64 ids over the whole key space, the table in pinned host DRAM at d = 64 (256 GB) when the box
    can pin that much, else at the widest d in {32, 16, 8, 4} that fits 40 % of the free host memory (the index side
    -- key space, direct map, versions, eviction -- is at full scale either way).  Closed-form table, touched rows,
    versions and the resident set against oracle/cache_model.py; row offsets beyond 4 GiB (and 64 GiB where d >= 32)."""
    rows = 1_000_000_000
    free, _ = torch.cuda.mem_get_info(dev)
    budget = _host_budget_bytes()
    width = next((w for w in (64, 32, 16, 8, 4) if rows * w * 4 <= budget), None)
    # HBM: direct map 4 B + versions 8 B per row, data + gradient lines of the hot tier
    if width is None or free < rows * 12 + int(0.1 * rows) * width * 8 + (8 << 30):
        pytest.skip("needs %d GB of pinnable host memory (have budget %d GB) and ~%d GB of HBM"
                    % (rows * 4 * 4 >> 30, budget >> 30, (rows * 12 + int(0.1 * rows) * 4 * 8) >> 30))
    top, far = _cold_tier_run(dev, rows, width, steps=6, seed=41, first=900)
    assert top > 900_000_000                                   # keys from the far end of the key space
    assert far * width * 4 > (4 << 30)
    if width >= 32:
        assert far * width * 4 > (64 << 30)
    print("configs[4] at rows=1e9: width %d (%d GB pinned)" % (width, rows * width * 4 >> 30))


def test_cold_tier_d64_at_the_largest_row_count_that_fits(dev):
    """configs[4]'s row width (d = 64) at the largest row count the box can pin (at most 1e9).  Where that is 1e9 --
    test_cold_tier_at_one_billion_rows has then run exactly this -- a second key stream at 1e8 rows (25.6 GB)."""
    budget = _host_budget_bytes()
    rows = min(1_000_000_000, budget // 256)
    if rows == 1_000_000_000:
        rows = 100_000_000
    if rows < 40_000_000:
        pytest.skip("needs more than 10 GB of pinnable host memory")
    top, far = _cold_tier_run(dev, rows, 64, steps=6, seed=43, first=950)
    assert far * 256 > (8 << 30)
    print("configs[4] at d=64: rows %d (%d GB pinned)" % (rows, rows * 256 >> 30))
