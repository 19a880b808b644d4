"""ctypes access to oracle/_ref/libherald_ref.so -- the reference's OWN cache policies, Unique<T> and
MiniLRUCache compiled from /root/reference (oracle/build_ref.sh).  TEST INFRASTRUCTURE ONLY.

PyDLL (GIL held) is required: hetu::CacheBase owns a py::list member (include/cache.h:31)."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libherald_ref.so")
_L = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _L
    if _L is None:
        L = ctypes.PyDLL(_PATH, mode=os.RTLD_LAZY)
        c = ctypes
        L.ref_unique_u64.restype = c.c_size_t
        L.ref_unique_u64.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
        L.ref_policy_new.restype = c.c_void_p
        L.ref_policy_new.argtypes = [c.c_int, c.c_size_t, c.c_size_t]
        L.ref_policy_free.argtypes = [c.c_void_p]
        L.ref_policy_insert.argtypes = [c.c_void_p, c.c_uint64, c.c_int]
        L.ref_policy_lookup.argtypes = [c.c_void_p, c.c_uint64]
        L.ref_policy_count.argtypes = [c.c_void_p, c.c_uint64]
        L.ref_policy_size.restype = c.c_size_t
        L.ref_policy_size.argtypes = [c.c_void_p]
        L.ref_policy_take_evicted.restype = c.c_size_t
        L.ref_policy_take_evicted.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t]
        L.ref_minilru_new.restype = c.c_void_p
        L.ref_minilru_new.argtypes = [c.c_int]
        L.ref_minilru_free.argtypes = [c.c_void_p]
        L.ref_minilru_check.argtypes = [c.c_void_p, c.c_int]
        L.ref_minilru_get.argtypes = [c.c_void_p, c.c_int]
        L.ref_minilru_outdate.argtypes = [c.c_void_p, c.c_int]
        L.ref_minilru_keys.restype = c.c_size_t
        L.ref_minilru_keys.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t]
        _L = L
    return _L


def unique(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    uniq = np.empty(keys.size, dtype=np.uint64)
    inv = np.empty(keys.size, dtype=np.int64)
    u = lib().ref_unique_u64(keys.ctypes.data, keys.size, uniq.ctypes.data, inv.ctypes.data)
    return uniq[:u].copy(), inv


class Policy:
    KINDS = {"lru": 0, "lfu": 1, "lfuopt": 2}

    def __init__(self, kind, limit, width=2):
        self.h = lib().ref_policy_new(self.KINDS[kind], limit, width)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_policy_free(self.h)
            self.h = None

    def insert(self, key, updates=0):
        lib().ref_policy_insert(self.h, int(key), int(updates))

    def lookup(self, key):
        return lib().ref_policy_lookup(self.h, int(key))

    def count(self, key):
        return lib().ref_policy_count(self.h, int(key))

    def size(self):
        return lib().ref_policy_size(self.h)

    def take_evicted(self):
        buf = np.empty(4096, dtype=np.uint64)
        n = lib().ref_policy_take_evicted(self.h, buf.ctypes.data, buf.size)
        return [int(x) for x in buf[:n]]


class MiniLRU:
    def __init__(self, capacity):
        self.h = lib().ref_minilru_new(capacity)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_minilru_free(self.h)
            self.h = None

    def check(self, k):
        return lib().ref_minilru_check(self.h, int(k))

    def get(self, k):
        return lib().ref_minilru_get(self.h, int(k))

    def outdate(self, k):
        lib().ref_minilru_outdate(self.h, int(k))

    def keys(self):
        buf = np.empty(1 << 16, dtype=np.int32)
        n = lib().ref_minilru_keys(self.h, buf.ctypes.data, buf.size)
        return [int(x) for x in buf[:n]]


# ---- the reference's own CPU operators of the hot path (oracle/_ref/libref_dnnl.so) -------------
# src/dnnl_ops/EmbeddingLookup.cpp:16-35 and src/dnnl_ops/Optimizers.cpp:51-74 compiled unchanged
# (oracle/build_ref.sh), called through a ctypes mirror of DLArray (src/common/dlarray.h:40-55).
_DNNL_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_dnnl.so")
_D = None


class _DLContext(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("device_type", ctypes.c_int)]


class _DLArray(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("ctx", _DLContext), ("ndim", ctypes.c_int),
                ("shape", ctypes.POINTER(ctypes.c_int64)), ("stride", ctypes.POINTER(ctypes.c_int64))]


def _dl(a):
    """DLArray view of a C-contiguous float32 numpy array (device_type 1 = CPU)."""
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
    arr = _DLArray(a.ctypes.data, _DLContext(0, 1), a.ndim, shape, None)
    arr._keep = (a, shape)
    return arr


def dnnl_available():
    return os.path.exists(_DNNL_PATH)


def _dnnl():
    global _D
    if _D is None:
        L = ctypes.CDLL(_DNNL_PATH)
        P = ctypes.POINTER(_DLArray)
        L.cpu_EmbeddingLookup.argtypes = [P, P, P]
        L.cpu_EmbeddingLookup.restype = ctypes.c_int
        L.cpu_SGDOptimizerSparseUpdate.argtypes = [P, P, P, ctypes.c_float]
        L.cpu_SGDOptimizerSparseUpdate.restype = ctypes.c_int
        _D = L
    return _D


def dnnl_embedding_lookup(table, ids):
    """The reference's cpu_EmbeddingLookup: out[..., :] = table[size_t(ids[...]), :]."""
    table = np.ascontiguousarray(table, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.float32)
    out = np.empty(ids.shape + (table.shape[1],), dtype=np.float32)
    if ids.size == 0:
        return out
    t, i, o = _dl(table), _dl(ids), _dl(out)
    rc = _dnnl().cpu_EmbeddingLookup(ctypes.byref(t), ctypes.byref(i), ctypes.byref(o))
    assert rc == 0
    return out


def dnnl_sgd_sparse_update(table, ids, grads, lr):
    """The reference's cpu_SGDOptimizerSparseUpdate, in place on `table` (float32, C-contiguous)."""
    assert table.dtype == np.float32 and table.flags["C_CONTIGUOUS"]
    ids = np.ascontiguousarray(ids, dtype=np.float32)
    grads = np.ascontiguousarray(grads, dtype=np.float32)
    t, i, g = _dl(table), _dl(ids), _dl(grads)
    rc = _dnnl().cpu_SGDOptimizerSparseUpdate(ctypes.byref(t), ctypes.byref(i), ctypes.byref(g),
                                              ctypes.c_float(lr))
    assert rc == 0
    return table
