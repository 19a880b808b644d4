"""ha_step_* (three batches of lookahead): ONE launch applies batch k, writes the rows of batch k+1 after that
update -- forwarded from the applying waves' registers where both batches name a row, copied from the table
otherwise --, finishes the plan of batch k+2 and sorts batch k+3.  Bit-exact against the oracle's sequential gather -> sparse SGD -> gather ...:
every output row of every step, the finished plans and the final table."""
import ctypes

import numpy as np
import pytest
import torch

from herald_amd import _lib, ops, synth
from oracle import cpu

pytestmark = pytest.mark.gpu


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _lookup(want_t, ids_int):
    """cpu_EmbeddingLookup with the library's definition of ids beyond the table (zeros)."""
    width = want_t.shape[1]
    out = np.zeros((ids_int.size, width), np.float32)
    ok = (ids_int >= 0) & (ids_int < want_t.shape[0])
    if ok.any():
        out[ok] = cpu.embedding_lookup(want_t, ids_int[ok].astype(np.float32)).reshape(-1, width)
    return out, ok


def _run_stream(dev, table0, batches, grads, lr, ids_dtype=np.float32, check_plans=True, table=None):
    width = table0.shape[1]
    want_t = table0.copy()
    if table is None:
        table = _dev(table0, dev)
    cap = max(max(b.size for b in batches), 1)
    pipe = ops.StepPipeline(table, cap, lr)
    cast = (lambda b: _dev(b.astype(np.float32), dev)) if ids_dtype == np.float32 else \
        (lambda b: _dev(b.astype(np.int64), dev))
    d_ids = [cast(b) for b in batches]
    B = len(batches)
    out = pipe.start(d_ids[0], d_ids[1] if B > 1 else None, d_ids[2] if B > 2 else None)
    for k in range(B):
        torch.cuda.synchronize()
        want_out, ok = _lookup(want_t, batches[k])
        np.testing.assert_array_equal(out.cpu().numpy().reshape(-1, width), want_out,
                                      err_msg="lookup rows of batch %d" % k)
        if ok.any():
            cpu.sgd_sparse_update(want_t, batches[k][ok].astype(np.float32), grads[k][ok], lr)
        out = pipe.step(_dev(grads[k], dev), d_ids[k + 3] if k + 3 < B else None)
        torch.cuda.synchronize()
        assert (out is None) == (k + 1 >= B)
        if check_plans:
            pl = pipe.plan_of(k)
            u, inv, cnt = cpu.unique(cpu.ids_to_keys(batches[k].astype(np.float32)) if ids_dtype == np.float32
                                     else np.minimum(batches[k], 0xFFFFFFFE).astype(np.uint32))
            assert pl.n_unique() == u.size, k
            np.testing.assert_array_equal(pl.uniq().cpu().numpy().astype(np.int64) & 0xFFFFFFFF, u.astype(np.int64))
            np.testing.assert_array_equal(pl.counts().cpu().numpy().astype(np.int64), cnt)
            np.testing.assert_array_equal(pl.inverse().cpu().numpy().astype(np.int64), inv)
    np.testing.assert_array_equal(table.cpu().numpy(), want_t, err_msg="table after the stream")
    # every key table was cleared by the call after its last reader, except those of the last batches
    return pipe


@pytest.mark.parametrize("width", [4, 16, 32, 64, 96, 128, 200, 512, 1024, 2048])
@pytest.mark.parametrize("rows,n", [(40, 700), (5000, 6656), (300, 63), (7, 1)])
def test_step_stream_bit_exact(dev, width, rows, n):
    """Small tables: almost every row of batch k+1 is updated by batch k (every forwarding path: short,
    medium and cooperative long runs, keys with more than 64 and more than 1024 destinations)."""
    rng = np.random.default_rng(width * 131 + rows + n)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    steps = 6
    batches = [np.minimum(rng.zipf(1.3, size=n) - 1, rows - 1) if k % 2 else rng.integers(0, rows, size=n)
               for k in range(steps)]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(steps)]
    _run_stream(dev, table0, batches, grads, 0.05)


def test_step_rare_in_current_frequent_in_next(dev):
    """A key with 1..3 occurrences now (short path, one wave) and hundreds in the next batch: the forwarding loop
    runs over several chunks of 64 destinations; and the reverse (cooperative long run, one destination)."""
    rng = np.random.default_rng(4)
    rows, width, n = 1000, 512, 4000
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    a = rng.integers(100, rows, size=n)
    a[:2] = 7                       # key 7: twice now ...
    b = rng.integers(100, rows, size=n)
    b[::5] = 7                      # ... 800 times next
    b[1::9] = 8                     # key 8: 445 times in b (long run), once in c
    c = rng.integers(100, rows, size=n)
    c[17] = 8
    c[100:130] = 9                  # medium run now, ~130 destinations next
    d = rng.integers(100, rows, size=n)
    d[::31] = 9
    batches = [a, b, c, d, a]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in batches]
    _run_stream(dev, table0, batches, grads, 0.1)


def test_step_ragged_batches_and_full_tables(dev):
    """Batch sizes 1 .. 12,288 (the limit), all keys distinct at the limit (key tables at their highest load:
    probe chains), interleaved with tiny batches."""
    rng = np.random.default_rng(21)
    rows, width = 60000, 32
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    lim = ops.step_max_ids()
    assert lim == 12288
    sizes = [lim, 1, lim, 300, 2, lim, 77]
    batches = []
    for s in sizes:
        batches.append(rng.permutation(rows)[:s] if s == lim else rng.integers(0, rows, size=s))
    batches[2][: lim // 2] = batches[0][: lim // 2]      # half of a full batch shared with the one before
    grads = [rng.standard_normal((s, width), dtype=np.float32) for s in sizes]
    _run_stream(dev, table0, batches, grads, 0.02)


def test_step_u64_ids_and_out_of_range(dev):
    rows, width, n = 900, 128, 3000
    rng = np.random.default_rng(5)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    batches = [rng.integers(0, rows, size=n) for _ in range(5)]
    for b in batches:
        b[rng.integers(0, n, size=40)] = rows + rng.integers(0, 50, size=40)     # beyond the table
    batches[1][5] = 2 ** 40
    batches[2][6] = 2 ** 31
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(5)]
    _run_stream(dev, table0, batches, grads, 0.1, ids_dtype=np.int64)
    _run_stream(dev, table0, [np.minimum(b, 2 ** 24) for b in batches], grads, 0.1, ids_dtype=np.float32,
                check_plans=False)


def test_step_unaligned_narrow_rows(dev):
    """Rows narrower than a 128-byte line in a table that starts off a line: no special case (nothing written
    in the launch is read in it)."""
    rng = np.random.default_rng(12)
    rows, width, n = 500, 8, 2000
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    flat = torch.empty(rows * width + 4, dtype=torch.float32, device=dev)
    table = flat[4:].view(rows, width)
    assert table.data_ptr() % 128 != 0
    table.copy_(torch.from_numpy(table0))
    batches = [rng.integers(0, rows, size=n) for _ in range(4)]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(4)]
    _run_stream(dev, table0, batches, grads, 0.1, table=table)


def test_step_single_batch_and_two_batches(dev):
    rng = np.random.default_rng(2)
    table0 = rng.standard_normal((100, 64), dtype=np.float32)
    for B in (1, 2, 3):
        batches = [rng.integers(0, 100, size=50) for _ in range(B)]
        grads = [rng.standard_normal((50, 64), dtype=np.float32) for _ in range(B)]
        _run_stream(dev, table0, batches, grads, 0.3)


def test_step_criteo_stream(dev):
    """24 consecutive Criteo-shaped batches (bs=256, d=512) on a 1M-row table, every output row checked."""
    rows, width, steps = 1_000_000, 512, 24
    rng = np.random.default_rng(77)
    table0 = (rng.standard_normal((rows, width), dtype=np.float32) * np.float32(0.01))
    batches = [synth.criteo_batch(256, step=s).reshape(-1) % rows for s in range(steps)]
    grads = [rng.standard_normal((6656, width), dtype=np.float32) for _ in range(4)]
    _run_stream(dev, table0, batches, [grads[k % 4] for k in range(steps)], 0.01, check_plans=False)


def test_step_refuses_what_it_cannot_do(dev):
    L = _lib.load()
    t = torch.zeros((10, 6), device=dev)
    with pytest.raises(ValueError):
        ops.StepPipeline(t, ops.step_max_ids() + 1, 0.1)
    pipe = ops.StepPipeline(t, 16, 0.1)
    with pytest.raises(RuntimeError, match="multiple of 4"):       # width 6
        pipe.start(torch.zeros(4, device=dev))
    with pytest.raises(RuntimeError):
        ops.StepPipeline(torch.zeros((10, 8), device=dev), 16, 0.1).step(torch.zeros((4, 8), device=dev))
    assert L.ha_step_tab_bytes() == 16 << 15


def test_step_replayed_from_a_graph(dev):
    """Twelve steps captured in one hipGraph (tables and plans rotate with period 4) and replayed."""
    rows, width, n = 20000, 256, 3000
    rng = np.random.default_rng(8)
    table0 = rng.standard_normal((rows, width), dtype=np.float32)
    nb = 12
    batches = [np.minimum(rng.zipf(1.2, size=n) - 1, rows - 1) for _ in range(nb)]
    d_ids = [_dev(b.astype(np.float32), dev) for b in batches]
    grads = [rng.standard_normal((n, width), dtype=np.float32) for _ in range(nb)]
    d_g = [_dev(g, dev) for g in grads]
    outs = [torch.empty((n, width), device=dev) for _ in range(nb)]
    table = _dev(table0, dev)
    pipe = ops.StepPipeline(table, n, 0.05)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        pipe.start(d_ids[0], d_ids[1], d_ids[2], out=outs[0], stream=s)
        pipe.step(d_g[0], d_ids[3], out=outs[1], stream=s)      # eager once (LDS attribute outside the capture)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for k in range(1, nb + 1):
                pipe.step(d_g[k % nb], d_ids[(k + 3) % nb], out=outs[(k + 1) % nb], stream=s)
        graph.replay()
        graph.replay()
        s.synchronize()
    want = table0.copy()
    seq = [0] + [k % nb for k in range(1, 2 * nb + 1)]
    for k in seq:
        last = cpu.embedding_lookup(want, batches[k].astype(np.float32))
        cpu.sgd_sparse_update(want, batches[k].astype(np.float32), grads[k], 0.05)
    np.testing.assert_array_equal(table.cpu().numpy(), want)
    # the last launch wrote the rows of the batch after the last applied one
    nxt = (seq[-1] + 1) % nb
    np.testing.assert_array_equal(outs[nxt].cpu().numpy(), cpu.embedding_lookup(want, batches[nxt].astype(np.float32)))
