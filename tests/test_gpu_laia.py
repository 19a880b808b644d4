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
