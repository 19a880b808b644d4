"""One launch per training step (ha_sgd_push_pull_*): apply(k) beside lookup(k+1), rows handed from the
applying wave to the gathering wave inside the launch.  Bit-exact against the oracle's sequential
gather -> sparse SGD -> gather ..., for every output row of every step, the plans and the final table."""
import numpy as np
import pytest
import torch

from herald_amd import ops, synth
from oracle import cpu

pytestmark = pytest.mark.gpu


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _run_sequence(dev, table0, batches, grads, lr, ids_dtype=np.float32, check_plans=True):
    """batches: list of 1-D id arrays (integers); returns nothing, asserts parity step by step."""
    width = table0.shape[1]
    want_t = table0.copy()
    table = _dev(table0, dev)
    cap = max(max(b.size for b in batches), 1)
    plans = [ops.IndexPlan(cap, dev), ops.IndexPlan(cap, dev)]
    pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
    if ids_dtype == np.float32:
        d_ids = [_dev(b.astype(np.float32), dev) for b in batches]
        f_ids = [b.astype(np.float32) for b in batches]
    else:
        d_ids = [_dev(b.astype(np.int64), dev) for b in batches]
        f_ids = [b.astype(np.float32) for b in batches]     # ids < 2^24 in these tests: exact
    out = ops.lookup_sort_pend(table, d_ids[0], plans[0], pends[0])
    for k in range(len(batches)):
        torch.cuda.synchronize()
        want_out = cpu.embedding_lookup(want_t, f_ids[k]) if batches[k].size else np.zeros((0, width), np.float32)
        np.testing.assert_array_equal(out.cpu().numpy().reshape(-1, width), want_out.reshape(-1, width),
                                      err_msg="lookup rows of batch %d" % k)
        if batches[k].size:
            cpu.sgd_sparse_update(want_t, f_ids[k], grads[k], lr)
        cur, nxt = plans[k % 2], plans[(k + 1) % 2]
        g = _dev(grads[k], dev)
        if k + 1 < len(batches):
            out = ops.sgd_push_pull(table, cur, g, lr, pends[k % 2], d_ids[k + 1], nxt, pends[(k + 1) % 2])
        else:
            ops.sgd_push_pull(table, cur, g, lr, pends[k % 2])
        torch.cuda.synchronize()
        if check_plans and batches[k].size:
            u, inv, cnt = cpu.unique(cpu.ids_to_keys(f_ids[k]))
            assert cur.n_unique() == u.size, k
            np.testing.assert_array_equal(cur.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, u.astype(np.int64))
            np.testing.assert_array_equal(cur.counts().cpu().numpy().astype(np.int64), cnt)
            np.testing.assert_array_equal(cur.inverse().cpu().numpy().astype(np.int64), inv)
        if k + 1 < len(batches):
            assert not nxt.handoff_timed_out(), "hand-off wait timed out at step %d" % k
    np.testing.assert_array_equal(table.cpu().numpy(), want_t, err_msg="table after the sequence")
    assert pends[0].is_idle() and pends[1].is_idle(), "pending tables must drain to zero"


@pytest.mark.parametrize("width", [4, 16, 32, 64, 96, 128, 200, 512, 1024, 2048])
@pytest.mark.parametrize("rows,n", [(40, 700), (5000, 6656), (300, 63), (7, 1)])
def test_push_pull_sequence_bit_exact(dev, width, rows, n):
    """Small tables: almost every row of batch k+1 is updated by batch k (every path of the hand-off:
    short, medium and cooperative long runs)."""
    rng = np.random.default_rng(width * 131 + rows + n)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    steps = 5
    batches = [np.minimum(rng.zipf(1.3, size=n) - 1, rows - 1) if k % 2 else rng.integers(0, rows, size=n)
               for k in range(steps)]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(steps)]
    _run_sequence(dev, table0, batches, grads, 0.05)


def test_push_pull_largest_single_launch_batches(dev):
    """16,640 and 18,432 ids per batch: the upper end of the single-launch regime (rank-by-counting tiles)."""
    rng = np.random.default_rng(44)
    rows, width = 90000, 64
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    for n in (16640, 18432):
        batches = [rng.integers(0, rows, size=n) if k % 2 else np.minimum(rng.zipf(1.2, size=n) - 1, rows - 1)
                   for k in range(4)]
        grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(4)]
        _run_sequence(dev, table0, batches, grads, 0.05)


def test_push_pull_unaligned_table_takes_the_separate_launches(dev):
    """A table that does not start on a 128-byte line cannot use the in-launch hand-off: same results."""
    rng = np.random.default_rng(12)
    rows, width, n = 500, 64, 2000
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    flat = torch.empty(rows * width + 4, dtype=torch.float32, device=dev)
    table = flat[4:].view(rows, width)
    assert table.data_ptr() % 128 != 0
    table.copy_(torch.from_numpy(table0))
    want = table0.copy()
    ids = [rng.integers(0, rows, size=n).astype(np.float32) for _ in range(3)]
    g = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(3)]
    plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
    pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
    out = ops.lookup_sort_pend(table, _dev(ids[0], dev), plans[0], pends[0])
    for k in range(3):
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), cpu.embedding_lookup(want, ids[k]))
        cpu.sgd_sparse_update(want, ids[k], g[k], 0.1)
        nxt = _dev(ids[k + 1], dev) if k < 2 else None
        out = ops.sgd_push_pull(table, plans[k % 2], _dev(g[k], dev), 0.1, pends[k % 2], nxt,
                                plans[(k + 1) % 2] if k < 2 else None, pends[(k + 1) % 2] if k < 2 else None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(table.cpu().numpy(), want)
    assert pends[0].is_idle() and pends[1].is_idle()


def test_push_pull_criteo_stream(dev):
    """24 consecutive Criteo-shaped batches (bs=256, d=512) on a 1M-row table, every output row checked."""
    rows, width, steps = 1_000_000, 512, 24
    rng = np.random.default_rng(77)
    table0 = (rng.standard_normal((rows, width), dtype=np.float32) * np.float32(0.01))
    batches = [synth.criteo_batch(256, step=s).reshape(-1) % rows for s in range(steps)]
    grads = [rng.standard_normal((6656, width), dtype=np.float32) for _ in range(4)]
    _run_sequence(dev, table0, batches, [grads[k % 4] for k in range(steps)], 0.01, check_plans=False)


def test_push_pull_u64_ids_and_out_of_range(dev):
    rows, width, n = 900, 128, 3000
    rng = np.random.default_rng(5)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    batches = [rng.integers(0, rows, size=n) for _ in range(4)]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(4)]
    _run_sequence(dev, table0, batches, grads, 0.1, ids_dtype=np.int64)
    # ids beyond the table: the lookup returns zeros, the update skips them, nothing waits for them
    table = _dev(table0, dev)
    ids0 = np.array([1, rows + 5, 2, 1, 2 ** 31, rows - 1], dtype=np.int64)
    ids1 = np.array([rows + 5, 1, 2, rows - 1, 7], dtype=np.int64)
    p0, p1 = ops.IndexPlan(8, dev), ops.IndexPlan(8, dev)
    q0, q1 = ops.PendingTable(dev), ops.PendingTable(dev)
    out0 = ops.lookup_sort_pend(table, _dev(ids0, dev), p0, q0)
    g0 = rng.standard_normal((6, width), dtype=np.float32)
    out1 = ops.sgd_push_pull(table, p0, _dev(g0, dev), 0.5, q0, _dev(ids1, dev), p1, q1)
    ops.sgd_push_pull(table, p1, _dev(np.zeros((5, width), np.float32), dev), 0.5, q1)
    torch.cuda.synchronize()
    want = table0.copy()
    ok0 = ids0 < rows
    w0 = np.zeros((6, width), np.float32)
    w0[ok0] = want[ids0[ok0]]
    np.testing.assert_array_equal(out0.cpu().numpy(), w0)
    cpu.sgd_sparse_update(want, ids0[ok0].astype(np.float32), g0[ok0], 0.5)
    ok1 = ids1 < rows
    w1 = np.zeros((5, width), np.float32)
    w1[ok1] = want[ids1[ok1]]
    np.testing.assert_array_equal(out1.cpu().numpy(), w1)
    np.testing.assert_array_equal(table.cpu().numpy(), want)
    assert q0.is_idle() and q1.is_idle()


def test_push_pull_mixed_regimes(dev):
    """Batches outside the single-launch regime (more than 36,864 ids; width % 4 != 0; empty) interleaved
    with batches inside it: the separate launches take over, nothing is left registered."""
    rng = np.random.default_rng(9)
    rows, width = 3000, 64
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    sizes = [500, 40000, 800, 0, 300, 36864, 10]
    batches = [rng.integers(0, rows, size=s) for s in sizes]
    grads = [rng.standard_normal((s, width), dtype=np.float32) for s in sizes]
    _run_sequence(dev, table0, batches, grads, 0.02)
    table1 = rng.standard_normal((200, 6), dtype=np.float32)          # width % 4 != 0
    b2 = [rng.integers(0, 200, size=90) for _ in range(3)]
    g2 = [rng.standard_normal((90, 6), dtype=np.float32) for _ in range(3)]
    _run_sequence(dev, table1, b2, g2, 0.02)


def test_push_pull_under_uneven_load(dev):
    """The hand-off under load: a second stream streams 1 GiB copies while 60 steps run; rows still exact."""
    rows, width, steps = 200_000, 512, 60
    rng = np.random.default_rng(3)
    table0 = (rng.standard_normal((rows, width), dtype=np.float32) * np.float32(0.01))
    batches = [synth.criteo_batch(256, step=1000 + s).reshape(-1) % rows for s in range(steps)]
    grads = [rng.standard_normal((6656, width), dtype=np.float32) for _ in range(3)]
    side = torch.cuda.Stream(device=dev)
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    with torch.cuda.stream(side):
        for _ in range(40):
            b.copy_(a, non_blocking=True)
    _run_sequence(dev, table0, batches, [grads[k % 3] for k in range(steps)], 0.01, check_plans=False)
    side.synchronize()
