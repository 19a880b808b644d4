"""The CPU oracle against numpy identities and the reference's own test scenarios (CPU only)."""
import numpy as np
import pytest

from herald_amd import synth
from oracle import cpu


def test_lookup_is_fancy_index():
    # reference test scenario: tests/test_dnnl_op.py:1120-1135 (5x5 table, ids [[0,1],[0,1]])
    table = np.arange(25, dtype=np.float32).reshape(5, 5)
    ids = np.array([[0, 1], [0, 1]], dtype=np.float32)
    out = cpu.embedding_lookup(table, ids)
    assert out.shape == (2, 2, 5)
    np.testing.assert_array_equal(out, table[ids.astype(np.int64)])


@pytest.mark.parametrize("width", [1, 4, 64, 128, 512])
def test_lookup_random(width):
    rng = np.random.default_rng(width)
    table = rng.standard_normal((1000, width), dtype=np.float32)
    ids = rng.integers(0, 1000, size=(37, 26)).astype(np.float32)
    np.testing.assert_array_equal(cpu.embedding_lookup(table, ids), table[ids.astype(np.int64)])


def test_sgd_sparse_update_matches_serial_numpy_with_duplicates():
    # duplicate ids as in tests/test_embedding_op.py:25-89 ([[0,1],[0,1]])
    rng = np.random.default_rng(0)
    table = rng.standard_normal((50, 16), dtype=np.float32)
    ids = np.array([0, 1, 0, 1, 7, 7, 7, 3], dtype=np.float32)
    grads = rng.standard_normal((8, 16), dtype=np.float32)
    lr = np.float32(0.1)
    ref = table.copy()
    for i, k in enumerate(ids.astype(np.int64)):
        ref[k] = ref[k] - lr * grads[i]          # float32 mul, then float32 sub
    got = cpu.sgd_sparse_update(table.copy(), ids, grads, float(lr))
    np.testing.assert_array_equal(got, ref)


def test_unique_matches_np_unique():
    rng = np.random.default_rng(1)
    for n in (1, 2, 100, 6656):
        keys = rng.integers(0, max(2, n // 3), size=n).astype(np.uint64)
        u, inv, cnt = cpu.unique(keys)
        ru, rinv, rcnt = np.unique(keys, return_inverse=True, return_counts=True)
        np.testing.assert_array_equal(u, ru)
        np.testing.assert_array_equal(inv, rinv)
        np.testing.assert_array_equal(cnt, rcnt)


def test_unique_empty():
    u, inv, cnt = cpu.unique(np.zeros(0, dtype=np.uint64))
    assert u.size == 0 and inv.size == 0 and cnt.size == 0


def test_dedup_reduce_matches_literal_cpu_deduplicate():
    # the only reference tests that pin dedup-reduce: tests/test_optimizer.py:117-198 use a
    # 500x400 table with 100 random duplicated ids and a numpy dict oracle
    rng = np.random.default_rng(2)
    ids = rng.integers(0, 500, size=100).astype(np.float32)
    vals = rng.standard_normal((100, 400), dtype=np.float32)
    u, inv, red = cpu.dedup_reduce(ids, vals)
    ru, rred = cpu.np_cpu_deduplicate(ids, vals)
    np.testing.assert_array_equal(u.astype(np.float32), ru)
    np.testing.assert_array_equal(red, rred)
    # dict-based oracle of the reference test (order of += is occurrence order as well)
    acc = {}
    for i, k in enumerate(ids.astype(np.int64)):
        acc[k] = acc.get(k, np.zeros(400, dtype=np.float32)) + vals[i]
    for j, k in enumerate(u.astype(np.int64)):
        np.testing.assert_array_equal(red[j], acc[k])


def test_float_ids_above_2_pow_24_round_like_numpy():
    ids = np.array([16777217, 16777216, 33762576, 33762575], dtype=np.int64)
    f = synth.as_f32_ids(ids)
    keys = cpu.ids_to_keys(f)
    np.testing.assert_array_equal(keys, f.astype(np.uint64))
    assert keys[0] == 16777216          # 2^24+1 is not representable


def test_partition_matches_reference_formula():
    # partitioner.h:46-57; SURVEY: 33,762,577 rows over 8 shards -> 4,220,323 on shard 0, 4,220,322 on 1-7
    s = cpu.partition(33762577, 8)
    lens = np.diff(s)
    assert lens[0] == 4220323 and (lens[1:] == 4220322).all() and s[-1] == 33762577


def test_synth_unique_ratio_calibration():
    r = []
    for step in range(8):
        ids = synth.as_f32_ids(synth.criteo_batch(256, step))
        r.append(np.unique(ids).size / ids.size)
    assert 0.38 < np.mean(r) < 0.44      # reference statistic: 0.411
    ids = synth.criteo_batch(256, 0)
    assert ids.shape == (256, 26) and ids.min() >= 0 and ids.max() < synth.CRITEO_ROWS
