"""ctypes front-end of oracle/oracle.c plus numpy restatements of the reference's Python steps.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c header).  Reference citations are relative to
/root/reference.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            import sys
            sys.path.insert(0, os.path.dirname(_HERE))
            from herald_amd import _build
            _build.build_oracle()
        L = ctypes.CDLL(path)
        c = ctypes
        L.oracle_embedding_lookup.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_void_p]
        L.oracle_sgd_sparse_update.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t,
                                               c.c_void_p, c.c_float]
        L.oracle_unique_u64.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p]
        L.oracle_unique_u64.restype = c.c_size_t
        L.oracle_ids_to_keys.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p]
        L.oracle_dedup_reduce.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_size_t,
                                          c.c_void_p]
        L.oracle_push_apply.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_void_p]
        L.oracle_partition.argtypes = [c.c_size_t, c.c_size_t, c.c_void_p]
        L.oracle_num_threads.restype = c.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def embedding_lookup(table, ids):
    """cpu_EmbeddingLookup (src/dnnl_ops/EmbeddingLookup.cpp:16-35)."""
    table = _f32(table)
    ids = _f32(ids)
    out = np.empty(ids.shape + (table.shape[1],), dtype=np.float32)
    lib().oracle_embedding_lookup(_p(table), table.shape[1], _p(ids), ids.size, _p(out))
    return out


def sgd_sparse_update(table, ids, grads, lr):
    """cpu_SGDOptimizerSparseUpdate (src/dnnl_ops/Optimizers.cpp:51-74); updates `table` in place."""
    assert table.dtype == np.float32 and table.flags.c_contiguous
    ids = _f32(ids)
    grads = _f32(grads)
    lib().oracle_sgd_sparse_update(_p(table), table.shape[1], _p(ids), ids.size, _p(grads),
                                   ctypes.c_float(lr))
    return table


def ids_to_keys(ids):
    ids = _f32(ids).reshape(-1)
    keys = np.empty(ids.size, dtype=np.uint64)
    lib().oracle_ids_to_keys(_p(ids), ids.size, _p(keys))
    return keys


def unique(keys):
    """hetu::Unique<T> / np.unique(return_inverse, return_counts): (uniq, inverse, counts)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
    n = keys.size
    uniq = np.empty(n, dtype=np.uint64)
    inverse = np.empty(n, dtype=np.int64)
    counts = np.empty(n, dtype=np.int64)
    u = lib().oracle_unique_u64(_p(keys), n, _p(uniq), _p(inverse), _p(counts))
    return uniq[:u].copy(), inverse, counts[:u].copy()


def dedup_reduce(ids, grads):
    """IndexedSlices.cpu_deduplicate (python/hetu/ndarray.py:556-576): (uniq_keys, reduced)."""
    keys = ids_to_keys(ids)
    grads = _f32(grads).reshape(keys.size, -1)
    uniq, inverse, _ = unique(keys)
    reduced = np.empty((uniq.size, grads.shape[1]), dtype=np.float32)
    lib().oracle_dedup_reduce(_p(inverse), keys.size, _p(grads), grads.shape[1], uniq.size,
                              _p(reduced))
    return uniq, inverse, reduced


def push_apply(table, uniq, reduced):
    """Server `+=` of SparsePush (ps-lite/include/ps/server/PSFHandle.h:130-164)."""
    uniq = np.ascontiguousarray(uniq, dtype=np.uint64)
    reduced = _f32(reduced)
    lib().oracle_push_apply(_p(table), table.shape[1], _p(uniq), uniq.size, _p(reduced))
    return table


def partition(length, nshard):
    """AveragePartitioner::partitionDense (ps-lite/include/ps/partitioner.h:46-57)."""
    starts = np.empty(nshard + 1, dtype=np.uint64)
    lib().oracle_partition(length, nshard, _p(starts))
    return starts.astype(np.int64)


def num_threads():
    return lib().oracle_num_threads()


# ---- numpy restatements of Python-level reference steps ---------------------------------------
def np_unique(ids):
    """IndexedSlices.deduplicate's host step (python/hetu/ndarray.py:534): np.unique on the float ids."""
    return np.unique(np.asarray(ids), return_inverse=True)


def np_cpu_deduplicate(ids, values):
    """Literal restatement of IndexedSlices.cpu_deduplicate (ndarray.py:556-576), Python loop included.
    Small inputs only."""
    np_indices = np.asarray(ids)
    unique_indices, inverse = np.unique(np_indices, return_inverse=True)
    last_dim = values.shape[-1]
    new_values = np.zeros((unique_indices.shape[0], last_dim)).astype(np.float32)
    flatten = np.asarray(values).reshape((-1, last_dim))
    for i, ind in enumerate(np.asarray(inverse).reshape(-1)):
        new_values[ind] += flatten[i]
    return unique_indices, new_values


# ---- sharded store (PS semantics) ----------------------------------------------------------------
def scale_values(values, lr):
    """`values *= -lr` of ParameterServerCommunicateOp (gpu_ops/ParameterServerCommunicate.py:24,58-59):
    numpy float32 array times a Python float -> float32 product, one rounding."""
    return (np.asarray(values, dtype=np.float32) * np.float32(-lr)).astype(np.float32)


def sparse_pull(table, ids):
    """PSAgent::vecPullSparse + serve(SparsePull): every position receives its row (PSAgent.h:185-237)."""
    return embedding_lookup(table, ids)


def sparse_push(table, ids, values, lr=None):
    """One worker's SparsePush on a global table: scale by -lr, reduce equal ids in position order from 0
    (PSAgent::vecPushSparse, PSAgent.h:124-183), server `+=` (PSFHandle.h:130-164).  In place."""
    vals = scale_values(values, lr) if lr is not None else _f32(values)
    uniq, _, red = dedup_reduce(ids, vals.reshape(np.asarray(ids).size, -1))
    return push_apply(table, uniq, red)


# ---- sparse optimizers on deduplicated rows (numpy, the oracle style of tests/test_optimizer.py:117-198) ----
def adagrad_sparse(param, acc, ids, grads, lr, eps):
    """adagrad_sparse_update, src/ops/OptimizersSparse.cu:331-349 (ids unique).  In place, float32."""
    f = np.float32
    idx = np.asarray(ids).astype(np.int64)
    g = np.asarray(grads, dtype=f)
    a = (acc[idx] + g * g).astype(f)
    acc[idx] = a
    param[idx] = (param[idx] - f(lr) * g / (np.sqrt(a, dtype=f) + f(eps))).astype(f)


def adam_sparse(param, m, v, ids, grads, lr, beta1, beta2, beta1t, beta2t, eps, weight_decay=None):
    """adam_sparse_update / adamw_sparse_update, OptimizersSparse.cu:391-416, 457-484.  In place."""
    f = np.float32
    idx = np.asarray(ids).astype(np.int64)
    g = np.asarray(grads, dtype=f)
    cm = (f(beta1) * m[idx] + (f(1) - f(beta1)) * g).astype(f)
    cv = (f(beta2) * v[idx] + (f(1) - f(beta2)) * g * g).astype(f)
    m[idx] = cm
    v[idx] = cv
    cm = (cm / (f(1) - f(beta1t))).astype(f)
    cv = (cv / (f(1) - f(beta2t))).astype(f)
    if weight_decay is None:
        param[idx] = (param[idx] - f(lr) * cm / (np.sqrt(cv, dtype=f) + f(eps))).astype(f)
    else:
        upd = (cm / (np.sqrt(cv, dtype=f) + f(eps))).astype(f)
        param[idx] = (param[idx] - f(lr) * (upd + f(weight_decay) * param[idx])).astype(f)


def l2_sparse(param, ids, grads, l2reg):
    """add_l2_regularization_sparse, OptimizersSparse.cu:3-18 (ids unique): grads += l2reg * param[ids]."""
    f = np.float32
    idx = np.asarray(ids).astype(np.int64)
    return (np.asarray(grads, dtype=f) + f(l2reg) * param[idx]).astype(f)


def momentum_sparse(param, veloc, ids, grads, lr, momentum, nesterov):
    """MomentumOptimizerSparseUpdate, OptimizersSparse.cu:101-231: ids may repeat; first phase
    velocity[id] += -lr*g per occurrence (Nesterov: param[id] too), here in occurrence order (the
    reference's float atomics pick an arbitrary one); second phase DENSE over the whole arrays.  In place."""
    f = np.float32
    idx = np.asarray(ids).astype(np.int64).reshape(-1)
    g = np.asarray(grads, dtype=f).reshape(idx.size, -1)
    for i, r in enumerate(idx):
        t = (-f(lr) * g[i]).astype(f)
        veloc[r] = (veloc[r] + t).astype(f)
        if nesterov:
            param[r] = (param[r] + t).astype(f)
    if nesterov:
        veloc[...] = (f(momentum) * veloc).astype(f)
        param[...] = (param + veloc).astype(f)
    else:
        param[...] = (param + veloc).astype(f)
        veloc[...] = (f(momentum) * veloc).astype(f)


def lamb_sparse(param, m, v, ids, grads, lr, beta1, beta2, beta1t, beta2t, eps, weight_decay):
    """LambOptimizerSparseUpdate, OptimizersSparse.cu:524-722 (ids unique): norm2 of the indexed parameter
    rows, Adam moments -> update direction, norm2 of the direction, scaled step.  In place."""
    f = np.float32
    idx = np.asarray(ids).astype(np.int64)
    g = np.asarray(grads, dtype=f)
    norm_p = f(np.sqrt(np.sum(param[idx].astype(np.float64) ** 2)))
    cm = (f(beta1) * m[idx] + (f(1) - f(beta1)) * g).astype(f)
    cv = (f(beta2) * v[idx] + (f(1) - f(beta2)) * g * g).astype(f)
    m[idx] = cm
    v[idx] = cv
    cm = (cm / (f(1) - f(beta1t))).astype(f)
    cv = (cv / (f(1) - f(beta2t))).astype(f)
    upd = (cm / (np.sqrt(cv, dtype=f) + f(eps))).astype(f)
    norm_u = f(np.sqrt(np.sum(upd.astype(np.float64) ** 2)))
    param[idx] = (param[idx] - f(lr) * (norm_p / norm_u) * (upd + f(weight_decay) * param[idx])).astype(f)
