"""libherald_ps.so -- the libps.so names of the reference (ps-lite/src/python_binding.cc:6-151) -- and the
glue that lets them serve a table sharded over several ranks.

The reference's Python binds libps with ctypes and calls `comm.InitTensor / SparsePull / SparsePush /
SSPushPull / Wait / SaveParam / LoadParam / rank / nrank` with DLArray handles
(python/hetu/gpu_ops/ParameterServerCommunicate.py:68-111, initializers.py:28-38).  `lib()` returns the
ctypes library with exactly those symbols.  With one rank they are served natively (csrc/ps.hip).  With
several, `attach_sharded(node_id, emb)` hands the engine a ShardedEmbedding: the shard is served by the same
kernels, the exchange is emb's all-to-all (RCCL over xGMI), registered as the engine's backend.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import DLArray, check

_PS = None
_ATTACHED = {}
_CB_KEEP = []


class DLEvent(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("handle", ctypes.c_void_p)]


class TensorInfo(ctypes.Structure):
    _fields_ = [("table", ctypes.c_void_p), ("len", ctypes.c_int64), ("width", ctypes.c_int64),
                ("row_start", ctypes.c_int64), ("rows_local", ctypes.c_int64), ("stream", ctypes.c_void_p)]


_PULL_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p)
_PUSH_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p)
_BAR_T = ctypes.CFUNCTYPE(ctypes.c_int)


class Backend(ctypes.Structure):
    _fields_ = [("sparse_pull", _PULL_T), ("sparse_push", _PUSH_T), ("barrier", _BAR_T)]


def _engine():
    L = _lib.load()
    if not getattr(L, "_ps_declared", False):
        A = ctypes.POINTER(DLArray)
        c = ctypes
        L.ha_ps_configure.argtypes = [c.c_int, c.c_int]
        L.ha_ps_set_backend.argtypes = [c.c_void_p]
        L.ha_ps_init_tensor.argtypes = [c.c_int, c.c_int, c.c_int64, c.c_int64, c.c_int, c.c_double, c.c_double,
                                        c.c_uint64]
        L.ha_ps_attach_tensor.argtypes = [c.c_int, c.c_void_p, c.c_int64, c.c_int64]
        L.ha_ps_tensor.argtypes = [c.c_int, c.POINTER(TensorInfo)]
        L.ha_ps_sparse_pull.argtypes = [c.c_int, A, A]
        L.ha_ps_sparse_push.argtypes = [c.c_int, A, A]
        L.ha_ps_dense_pull.argtypes = [c.c_int, A]
        for n in ("ha_ps_wait", "ha_ps_clear"):
            getattr(L, n).argtypes = [c.c_int]
        for n in ("ha_ps_save", "ha_ps_load"):
            getattr(L, n).argtypes = [c.c_int, c.c_char_p]
        L._ps_declared = True
    return L


def lib():
    """ctypes handle of libherald_ps.so with the reference's argument types declared."""
    global _PS
    if _PS is None:
        _engine()
        if not os.path.exists(_lib.PS_LIB_PATH):
            raise _lib.HeraldAmdError("libherald_ps.so is not built (%s)" % _lib.PS_LIB_PATH)
        P = ctypes.CDLL(_lib.PS_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        A, E = ctypes.POINTER(DLArray), ctypes.POINTER(DLEvent)
        c = ctypes
        P.InitTensor.argtypes = [c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_double, c.c_double, c.c_ulonglong,
                                 c.c_int, c.POINTER(c.c_float), c.c_int]
        P.SparsePull.argtypes = [c.c_int, A, A]
        P.SparsePush.argtypes = [c.c_int, A, A, E]
        P.SSPushPull.argtypes = [c.c_int, A, A, A, A, E]
        P.SDPushPull.argtypes = [c.c_int, A, A, A, E]
        P.Pull.argtypes = [c.c_int, A]
        for n in ("Wait", "Clear", "ClearOnServer"):
            getattr(P, n).argtypes = [c.c_int]
        for n in ("SaveParam", "LoadParam"):
            getattr(P, n).argtypes = [c.c_int, c.c_char_p]
        for n in ("InitTensor", "SparsePull", "SparsePush", "SSPushPull", "SDPushPull", "Pull", "Wait", "Clear",
                  "ClearOnServer", "SaveParam", "LoadParam", "Init", "Finalize", "BarrierWorker"):
            getattr(P, n).restype = None
        P.rank.restype = c.c_int
        P.nrank.restype = c.c_int
        _PS = P
    return _PS


def configure(rank, nrank):
    check(_engine().ha_ps_configure(int(rank), int(nrank)), "ha_ps_configure")


def tensor(node_id, device=None):
    """This rank's shard of a tensor created by InitTensor, as a torch tensor (no copy)."""
    from .cache import _dev_view
    info = TensorInfo()
    check(_engine().ha_ps_tensor(int(node_id), ctypes.byref(info)), "ha_ps_tensor")
    dev = torch.device(device if device is not None else "cuda")
    return _dev_view(info.table, (info.rows_local, info.width), torch.float32, dev), info


def attach_sharded(node_id, emb, barrier=None):
    """Serve `emb` (a herald_amd.sharded.ShardedEmbedding) under `node_id`: SparsePull / SparsePush of
    libherald_ps.so on that node go through emb's routing and all-to-all; the shard is emb.table."""
    from .cache import _dev_view
    L = _engine()
    if not _ATTACHED:
        if L.ha_ps_nrank() != emb.world or L.ha_ps_rank() != emb.rank:
            configure(emb.rank, emb.world)
    check(L.ha_ps_attach_tensor(int(node_id), ctypes.c_void_p(emb.table.data_ptr()), emb.rows, emb.width),
          "ha_ps_attach_tensor")
    _ATTACHED[int(node_id)] = emb
    dev = emb.device

    def served(node):
        """The store behind `node`: attached explicitly, or -- a tensor InitTensor created on every rank with this
        engine's partition (ha_ps_init_tensor allocates exactly the AveragePartitioner range) -- wrapped on first
        use in a ShardedEmbedding with the exchange settings of the store this backend was registered with.  Every
        rank reaches this at the same call (SparsePull / SparsePush are issued by all workers)."""
        e = _ATTACHED.get(node)
        if e is None:
            from .sharded import ShardedEmbedding
            shard, info = tensor(node, dev)
            e = ShardedEmbedding(int(info.len), int(info.width), dev, group=emb.group, table=shard,
                                 a2a=emb._a2a_fn, side_stream=emb.side_stream)
            _ATTACHED[node] = e
        return e

    def pull_cb(node, ids_ptr, n, out_ptr, stream):
        try:
            e = served(node)
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                ids = _dev_view(ids_ptr, (n,), torch.float32, dev)
                out = _dev_view(out_ptr, (n, e.width), torch.float32, dev)
                out.copy_(e.pull(ids))
            return 0
        except Exception as ex:   # surfaced through ha_last_error-less path: print, report failure
            print("[herald_ps] SparsePull backend failed: %r" % (ex,))
            return -1

    def push_cb(node, ids_ptr, n, vals_ptr, stream):
        try:
            e = served(node)
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                ids = _dev_view(ids_ptr, (n,), torch.float32, dev)
                vals = _dev_view(vals_ptr, (n, e.width), torch.float32, dev)
                e.push(ids, vals)          # values arrive already scaled by -lr (ParameterServerCommunicate.py:58-59)
            return 0
        except Exception as ex:
            print("[herald_ps] SparsePush backend failed: %r" % (ex,))
            return -1

    def barrier_cb():
        if barrier is not None:
            barrier()
        return 0

    b = Backend(_PULL_T(pull_cb), _PUSH_T(push_cb), _BAR_T(barrier_cb))
    _CB_KEEP.append(b)
    check(L.ha_ps_set_backend(ctypes.byref(b)), "ha_ps_set_backend")
