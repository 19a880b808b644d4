"""CPU checks for the TopkScheduler pieces that need no GPU: the oracle restatement on a hand-worked
case, the thread slicing of topk_scheduler.cc:398-407, and the shared-memory ring of the local-shared
distribution (ha_shm_ring_*, C-ABI, no device call)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import laia_model


def test_thread_slices_follow_the_reference_formula():
    # batch 22 over 4 threads: x=5, y=2 -> thread 0 takes 7, the others 5 starting at y + t*x
    assert laia_model.topk_thread_slices(22, 4) == [(0, 7), (7, 12), (12, 17), (17, 22)]
    assert laia_model.topk_thread_slices(8, 1) == [(0, 8)]


def test_topk_model_hand_case():
    # 2 workers, mini batch 2, one table scored ("movie" order [0, 1], top_k 1), cache of 10 rows
    samples = np.array([[1, 50], [2, 51], [1, 52], [3, 50]], dtype=np.uint64)
    m = laia_model.TopkSchedulerModel(samples, 1, 2, 1, 2, 0, 10, 1, "movie", 1)
    plan0, dist0 = m.get_dist(0)
    # empty snapshots: every score is 0, candidate 0: samples fill worker 0's quota, then worker 1's
    assert dist0 == [[0, 1], [2, 3]] and plan0 == [[], []]
    out = m.emit()[0]
    assert out[0] == [] and out[1] == [0, 1] and out[-1] == [0]
    # after batch 0 worker 0 holds rows {1,2,50,51}, worker 1 {1,3,50,52}.  Batch 1 (same samples):
    # sample 0 (row 1): both score 1, worker 0 reached it first -> worker 0; sample 1 (row 2): only
    # worker 0; sample 2 (row 1): candidate 0 is full -> worker 1; sample 3 (row 3): worker 1.
    assert out[3] == [0, 1]
    # plan = own samples' rows (all tables) valid at the worker
    assert out[2] == [1, 2, 50, 51]
    assert m.update_pull[0] == 4 and m.miss_pull[0] == 4 and m.update_push[0] == 4


@pytest.fixture(scope="module")
def lib():
    from herald_amd import _lib
    return _lib.load()


def test_shm_ring_round_trip_and_backpressure(lib):
    name = ("ha_test_ring_%d" % os.getpid()).encode()
    w = lib.ha_shm_ring_open(name, 1, 64)
    assert w
    r = lib.ha_shm_ring_open(name, 0, 0)
    assert r
    buf = (ctypes.c_uint64 * 64)()
    need = ctypes.c_int64(0)
    assert lib.ha_shm_ring_recv(r, buf, 64, ctypes.byref(need)) == -1          # empty
    msg = (ctypes.c_uint64 * 5)(7, 8, 9, 10, 11)
    sent = 0
    while lib.ha_shm_ring_send(w, msg, 5) == 1:                                 # 6 words per message, 64-word ring
        sent += 1
    assert sent == 10 and lib.ha_shm_ring_send(w, msg, 5) == 0                  # full: try again later
    assert lib.ha_shm_ring_send(w, msg, 64) == -1                               # can never fit
    small = (ctypes.c_uint64 * 2)()
    assert lib.ha_shm_ring_recv(r, small, 2, ctypes.byref(need)) == -2 and need.value == 5
    for _ in range(sent):
        assert lib.ha_shm_ring_recv(r, buf, 64, ctypes.byref(need)) == 5
        assert list(buf[:5]) == [7, 8, 9, 10, 11]
    assert lib.ha_shm_ring_pending_words(r) == 0
    empty = (ctypes.c_uint64 * 1)()
    assert lib.ha_shm_ring_send(w, empty, 0) == 1                               # empty plan: zero-length message
    assert lib.ha_shm_ring_recv(r, buf, 64, ctypes.byref(need)) == 0
    # wrap-around keeps message integrity
    for k in range(40):
        m3 = (ctypes.c_uint64 * 3)(k, k + 1, k + 2)
        assert lib.ha_shm_ring_send(w, m3, 3) == 1
        assert lib.ha_shm_ring_recv(r, buf, 64, ctypes.byref(need)) == 3 and list(buf[:3]) == [k, k + 1, k + 2]
    lib.ha_shm_ring_close(r)
    lib.ha_shm_ring_close(w)
    assert not lib.ha_shm_ring_open(name, 0, 0)                                 # unlinked by its creator
