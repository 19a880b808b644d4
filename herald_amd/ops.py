"""Host-side mirror of the reference's embedding operators, on torch-allocated device memory.

Mirrors (reference paths relative to /root/reference):
  embedding_lookup            python/hetu/gpu_links/EmbeddingLookUpLink.py:8-13 -> DLGpuEmbeddingLookUp
  sgd_sparse_update           python/hetu/gpu_links/OptimizerLink.py:23-33      -> SGDOptimizerSparseUpdate
  indexedslices_oneside_add   python/hetu/gpu_links/IndexedSlicesLink.py         -> IndexedSlicesOneSideAdd
  IndexedSlices               python/hetu/ndarray.py:503-611 (deduplicate on the GPU instead of a
                              host np.unique round trip)
  IndexPlan                   the per-batch sorted-unique/inverse/counts plan (np.unique /
                              hetu::Unique<T> semantics) that gather, dedup-reduce and apply share

torch is plumbing only: it owns the device allocations and the stream; every computation below is a
call into libherald_amd.so.  There is no fallback: a missing library raises HeraldAmdError.
"""
import ctypes
import os

import numpy as np

import torch

from . import _lib
from ._lib import DLArray, DLContext, DLStream, PlanView, check


def _stream_ptr(stream=None):
    if stream is None:      # raw handle of torch's current stream, without building a Stream object
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    if isinstance(stream, int):
        return ctypes.c_void_p(stream)
    return ctypes.c_void_p(stream.cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _raw_stream(stream=None):
    """Raw hipStream_t value of `stream` (None = torch's current stream)."""
    p = _stream_ptr(stream).value
    return p if p is not None else 0


def _require(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP torch tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


# ---- DLArray plumbing (for the reference-named symbols) ------------------------------------------
class DLHolder:
    """Keeps the ctypes shape buffer alive next to the DLArray struct."""

    def __init__(self, t):
        self.tensor = t
        self.shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
        self.arr = DLArray(ctypes.c_void_p(t.data_ptr()),
                           DLContext(t.device.index or 0, 2 if t.is_cuda else 1), t.dim(),
                           ctypes.cast(self.shape, ctypes.POINTER(ctypes.c_int64)), None)

    @property
    def handle(self):
        return ctypes.byref(self.arr)


class DLStreamHolder:
    """DLStream whose handle points AT a hipStream_t, as the reference expects
    (`*(cudaStream_t *)stream_handle->handle`, src/ops/EmbeddingLookup.cu:46)."""

    def __init__(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream()
        self._raw = ctypes.c_void_p(stream.cuda_stream)
        self.s = DLStream(stream.device.index or 0, ctypes.cast(ctypes.pointer(self._raw), ctypes.c_void_p))

    @property
    def handle(self):
        return ctypes.byref(self.s)


# ---- tolerance mode --------------------------------------------------------------------------------------
def set_tolerance_mode(on):
    """Process-wide (include/herald_amd.h, ha_set_tolerance_mode): runs of 64 or more occurrences of a key are applied
    as `row - tree_sum(lr * g)` in a fixed order (within BASELINE.json's 1e-5 on accumulated gradients) instead of the
    reference's serial chain; shorter runs stay bit-exact.  Default off.  on = 2 (or "chunked"): also cut runs beyond 256
    occurrences into chunks in the applies of a finished plan (header).  Returns the previous setting (False / True / 2)."""
    L = _lib.load()
    prev = int(L.ha_get_tolerance_mode())
    mode = 2 if on in (2, "chunked") and on is not True else (1 if on else 0)
    check(L.ha_set_tolerance_mode(mode), "ha_set_tolerance_mode")
    return 2 if prev == 2 else bool(prev)


# ---- forward gather ----------------------------------------------------------------------------------
def embedding_lookup(table, ids, out=None, stream=None):
    """out[..., :] = table[(size_t)ids[...], :]; ids float32 (operator boundary) or int64/uint64 keys."""
    L = _lib.load()
    _require(table, torch.float32, "table")
    if table.dim() != 2:
        raise ValueError("table must be 2-D")
    if not ids.is_cuda or not ids.is_contiguous():
        raise ValueError("ids must be a contiguous device tensor")
    width = table.shape[1]
    if out is None:
        out = torch.empty(tuple(ids.shape) + (width,), dtype=torch.float32, device=table.device)
    _require(out, torch.float32, "out")
    if out.numel() != ids.numel() * width:
        raise ValueError("out has the wrong size")
    if ids.dtype == torch.float32:
        fn = L.ha_gather_f32ids
    elif ids.dtype in (torch.int64, torch.uint64):
        fn = L.ha_gather_u64ids
    else:
        raise TypeError("ids must be float32 or (u)int64")
    check(fn(_ptr(table), table.shape[0], width, _ptr(ids), ids.numel(), _ptr(out), _stream_ptr(stream)),
          "ha_gather")
    return out


# ---- index plan ----------------------------------------------------------------------------------------
class IndexPlan:
    """Device-resident np.unique(return_inverse, return_counts) + occurrence lists of one id batch."""

    def __init__(self, capacity, device=None):
        L = _lib.load()
        self.capacity = int(capacity)
        self.device = torch.device(device if device is not None else "cuda")
        self.nbytes = L.ha_plan_bytes(self.capacity)
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)
        self.ws[:256].zero_()      # the header: n_unique and the (sticky) hand-off time-out flag
        torch.cuda.current_stream(self.device).synchronize()   # users launch on streams of their own
        self.n = 0
        self._view = None
        self._route_cache = None
        self._produced_on = None     # raw stream of the launch that last wrote the plan

    def produced_on(self, stream=None):
        """Record the stream of the launch that (re)wrote this plan; the host-side readers below wait for it."""
        self._produced_on = _raw_stream(stream)

    def _sync_producer(self):
        if self._produced_on is not None:
            torch.cuda.ExternalStream(self._produced_on, device=self.device).synchronize() \
                if self._produced_on else torch.cuda.synchronize(self.device)

    def sort(self, ids, stream=None, key_limit=None):
        """Stable sort only (keys / sorted / perm): all that sgd_apply and push_apply consume."""
        return self.build(ids, stream, sort_only=True, key_limit=key_limit)

    def finish(self, stream=None):
        """Second phase after sort(): n_unique, uniq, counts, seg, inverse, upos."""
        check(_lib.load().ha_plan_finish(_ptr(self.ws), self.n, _stream_ptr(stream)), "ha_plan_finish")
        self.produced_on(stream)
        return self

    def build(self, ids, stream=None, sort_only=False, key_limit=None):
        """key_limit: a bound on the valid keys (the table's row count) -- lets batches of 18,433 .. 36,864 ids take
        the bucket sort (ha_plan_*_lim); results are identical with and without it."""
        L = _lib.load()
        n = ids.numel()
        if n > self.capacity:
            raise ValueError("plan capacity %d < %d ids" % (self.capacity, n))
        if not ids.is_cuda or not ids.is_contiguous():
            raise ValueError("ids must be a contiguous device tensor")
        # the workspace layout depends on n, so a view is per build
        if ids.dtype not in (torch.float32, torch.int64, torch.uint64):
            raise TypeError("ids must be float32 or (u)int64")
        kind = "f32ids" if ids.dtype == torch.float32 else "u64ids"
        name = "ha_plan_%s_%s" % ("sort" if sort_only else "build", kind)
        if key_limit is not None:
            rc = getattr(L, name + "_lim")(_ptr(ids), n, _ptr(self.ws), int(key_limit), _stream_ptr(stream))
        elif ids.dtype == torch.float32:
            rc = getattr(L, name)(_ptr(ids), n, _ptr(self.ws), _stream_ptr(stream))
        elif ids.dtype in (torch.int64, torch.uint64):
            rc = getattr(L, name)(_ptr(ids), n, _ptr(self.ws), _stream_ptr(stream))
        else:
            raise TypeError("ids must be float32 or (u)int64")
        check(rc, "ha_plan_build")
        self.n = n
        self._view = None
        self._route_cache = None
        self.produced_on(stream)
        return self

    # -- typed views into the workspace (no copies) --
    def view(self):
        if self._view is None:
            v = PlanView()
            check(_lib.load().ha_plan_view_of(_ptr(self.ws), self.n, ctypes.byref(v)), "ha_plan_view_of")
            self._view = v
        return self._view

    def _slice(self, addr, count, dtype):
        itemsize = torch.empty(0, dtype=dtype).element_size()
        off = addr - self.ws.data_ptr()
        return self.ws[off:off + count * itemsize].view(dtype)

    def n_unique(self):
        """Host int.  Waits for the stream of the launch that last wrote the plan (whichever stream the caller
        passed to build / finish / lookup_sort / sgd_apply_finish / sgd_push_pull), then reads the count."""
        self._sync_producer()
        self._raise_if_handoff_timed_out()
        return int(self._slice(self.view().n_unique, 1, torch.int64).item())

    def _raise_if_handoff_timed_out(self):
        addr = _lib.load().ha_plan_handoff_timeout(_ptr(self.ws))
        if bool(self._slice(addr, 1, torch.int64).item()):
            raise _lib.HeraldAmdError(
                "a wait inside a launch that wrote this plan timed out -- the hand-off of ha_sgd_push_pull (the rows that "
                "lookup returned may predate the update of the previous batch) or the histogram exchange of a radix pass "
                "(the order of the plan is not to be trusted): include/herald_amd.h, ha_plan_handoff_timeout")

    def handoff_timed_out(self):
        """True if a hand-off wait of the sgd_push_pull launch that sorted this plan gave up (host sync)."""
        addr = _lib.load().ha_plan_handoff_timeout(_ptr(self.ws))
        self._sync_producer()
        return bool(self._slice(addr, 1, torch.int64).item())

    def n_unique_dev(self):
        return self._slice(self.view().n_unique, 1, torch.int64)

    def keys(self):
        return self._slice(self.view().keys, self.n, torch.int32)

    def sorted_keys(self):
        return self._slice(self.view().sorted, self.n, torch.int32)

    def perm(self):
        return self._slice(self.view().perm, self.n, torch.int32)

    def inverse(self):
        return self._slice(self.view().inverse, self.n, torch.int32)

    def uniq(self, u=None):
        u = self.n_unique() if u is None else u
        return self._slice(self.view().uniq, u, torch.int32)

    def counts(self, u=None):
        u = self.n_unique() if u is None else u
        return self._slice(self.view().counts, u, torch.int32)

    def seg(self, u=None):
        u = self.n_unique() if u is None else u
        return self._slice(self.view().seg, u + 1, torch.int32)

    def export_f32(self, stream=None):
        """(uniq_f32[U], inverse_f32[n]) as IndexedSlices.deduplicate hands them on (ndarray.py:534-545)."""
        u = self.n_unique()
        uniq = torch.empty(self.n, dtype=torch.float32, device=self.device)
        inv = torch.empty(self.n, dtype=torch.float32, device=self.device)
        check(_lib.load().ha_plan_export_f32(_ptr(self.ws), self.n, _ptr(uniq), _ptr(inv),
                                             _stream_ptr(stream)), "ha_plan_export_f32")
        return uniq[:u], inv


# ---- backward ---------------------------------------------------------------------------------------------
def dedup_reduce(plan, grads, out=None, stream=None, scale=None):
    """reduced[u,:] = sum of (scale *) grads rows of unique key u in occurrence order (cpu_deduplicate
    order; with scale = -lr this is the worker side of a PS sparse push)."""
    _require(grads, torch.float32, "grads")
    n = plan.n
    width = grads.numel() // max(n, 1) if n else (grads.shape[-1] if grads.dim() else 1)
    if out is None:
        out = torch.empty((max(n, 1), width), dtype=torch.float32, device=grads.device)
    if scale is None:
        check(_lib.load().ha_dedup_reduce(_ptr(plan.ws), n, _ptr(grads), width, _ptr(out),
                                          _stream_ptr(stream)), "ha_dedup_reduce")
    else:
        check(_lib.load().ha_dedup_reduce_scaled(_ptr(plan.ws), n, _ptr(grads), width, ctypes.c_float(scale),
                                                 _ptr(out), _stream_ptr(stream)), "ha_dedup_reduce_scaled")
    return out


def sgd_apply(table, plan, grads, lr, stream=None, finished=False):
    """table[key,:] -= lr*grads[i,:] per occurrence, occurrence order (bit-exact cpu_SGDOptimizerSparseUpdate).
    finished=True: the plan went through build() / finish() -- batches of more than 36,864 ids then map waves to unique
    keys (ha_sgd_apply_finished); same results."""
    _require(table, torch.float32, "table")
    _require(grads, torch.float32, "grads")
    fn = _lib.load().ha_sgd_apply_finished if finished else _lib.load().ha_sgd_apply
    check(fn(_ptr(table), table.shape[0], table.shape[1], _ptr(plan.ws), plan.n,
             _ptr(grads), ctypes.c_float(lr), _stream_ptr(stream)), "ha_sgd_apply")
    return table


def push_apply(table, plan, grads, stream=None):
    """table[key,:] += (0 + g_i0 + g_i1 ...)  -- worker-side reduce + server-side `+=` of a sparse push."""
    _require(table, torch.float32, "table")
    _require(grads, torch.float32, "grads")
    check(_lib.load().ha_push_apply(_ptr(table), table.shape[0], table.shape[1], _ptr(plan.ws), plan.n,
                                    _ptr(grads), _stream_ptr(stream)), "ha_push_apply")
    return table


def sgd_sparse_update(table, ids, grads, lr, stream=None):
    """One call: plan + apply (reference SGDOptimizerSparseUpdate / cpu_SGDOptimizerSparseUpdate)."""
    _require(table, torch.float32, "table")
    _require(ids, torch.float32, "ids")
    _require(grads, torch.float32, "grads")
    check(_lib.load().ha_sgd_sparse_update_f32ids(_ptr(table), table.shape[0], table.shape[1], _ptr(ids),
                                                  ids.numel(), _ptr(grads), ctypes.c_float(lr),
                                                  _stream_ptr(stream)), "ha_sgd_sparse_update")
    return table


# ---- fused launches: two per training step --------------------------------------------------------
def lookup_sort(table, ids, plan, out=None, stream=None):
    """Forward of one batch in ONE launch: out = table[ids] and plan.sort(ids)."""
    L = _lib.load()
    _require(table, torch.float32, "table")
    n = ids.numel()
    width = table.shape[1]
    if n > plan.capacity:
        raise ValueError("plan capacity %d < %d ids" % (plan.capacity, n))
    if out is None:
        out = torch.empty(tuple(ids.shape) + (width,), dtype=torch.float32, device=table.device)
    if ids.dtype == torch.float32:
        fn = L.ha_lookup_sort_f32ids
    elif ids.dtype in (torch.int64, torch.uint64):
        fn = L.ha_lookup_sort_u64ids
    else:
        raise TypeError("ids must be float32 or (u)int64")
    check(fn(_ptr(table), table.shape[0], width, _ptr(ids), n, _ptr(out), _ptr(plan.ws),
             _stream_ptr(stream)), "ha_lookup_sort")
    plan.n = n
    plan._view = None
    plan.produced_on(stream)
    return out


def sgd_apply_finish(table, plan, grads, lr, stream=None, next_ids=None):
    """Backward of one batch in ONE launch: sgd_apply + plan.finish().  next_ids (float32): the ids of
    the batch whose lookup follows -- their rows are touched on the way out so that the lookup finds
    them in the Infinity Cache (ha_sgd_apply_finish_prefetch_f32ids)."""
    _require(table, torch.float32, "table")
    _require(grads, torch.float32, "grads")
    if next_ids is not None and next_ids.dtype == torch.float32:
        _require(next_ids, torch.float32, "next_ids")
        check(_lib.load().ha_sgd_apply_finish_prefetch_f32ids(
            _ptr(table), table.shape[0], table.shape[1], _ptr(plan.ws), plan.n, _ptr(grads), ctypes.c_float(lr),
            _ptr(next_ids), next_ids.numel(), _stream_ptr(stream)), "ha_sgd_apply_finish_prefetch_f32ids")
        plan.produced_on(stream)
        return table
    check(_lib.load().ha_sgd_apply_finish(_ptr(table), table.shape[0], table.shape[1], _ptr(plan.ws),
                                          plan.n, _ptr(grads), ctypes.c_float(lr), _stream_ptr(stream)),
          "ha_sgd_apply_finish")
    plan.produced_on(stream)
    return table


# ---- fused deduplicate + optimizer step ------------------------------------------------------------------
_OPT_KINDS = {"adagrad": 0, "adam": 1, "adamw": 2}


def sparse_opt_fused(kind, param, ids, grads, state1, state2=None, lr=0.01, eps=1e-7, beta1=0.9, beta2=0.999,
                     beta1t=0.9, beta2t=0.999, weight_decay=0.0, plan=None, stream=None):
    """grad.deduplicate() + {AdaGrad,Adam,AdamW}OptimizerSparseUpdate (OptimizerLink.py:52-100) in one call on
    the RAW (not deduplicated) float32 ids and their gradient rows; bit-identical to the two-step sequence."""
    _require(param, torch.float32, "param")
    _require(ids, torch.float32, "ids")
    _require(grads, torch.float32, "grads")
    n = ids.numel()
    if plan is None:
        plan = IndexPlan(max(n, 1), param.device)
    hyper = (ctypes.c_float * 7)(lr, eps, beta1, beta2, beta1t, beta2t, weight_decay)
    check(_lib.load().ha_sparse_opt_fused_f32ids(_OPT_KINDS[kind], _ptr(param), param.shape[0], param.shape[1],
                                                 _ptr(ids), n, _ptr(grads), _ptr(state1),
                                                 _ptr(state2) if state2 is not None else None, hyper, _ptr(plan.ws),
                                                 _stream_ptr(stream)), "ha_sparse_opt_fused_f32ids")
    return param


# ---- one launch per training step: backward of batch k beside the forward of batch k+1 ---------------
class PendingTable:
    """Per-batch hand-off table of ha_sgd_push_pull_* (include/herald_amd.h): all-zero when idle; the
    launch that sorts a batch registers its keys here, the launch that applies the batch drains it."""

    def __init__(self, device=None):
        L = _lib.load()
        self.device = torch.device(device if device is not None else "cuda")
        self.buf = torch.zeros(L.ha_pend_bytes(), dtype=torch.uint8, device=self.device)
        # the zero fill ran on torch's current stream; launches that register keys here come on streams
        # of the caller's choice, so it must have landed before the constructor returns
        torch.cuda.current_stream(self.device).synchronize()

    def reset(self, stream=None):
        check(_lib.load().ha_pend_reset(_ptr(self.buf), _stream_ptr(stream)), "ha_pend_reset")
        return self

    def is_idle(self):
        """Host check (synchronises): every registered unit was given back."""
        return not bool(self.buf.view(torch.int32).any().item())


def _ids_kind(ids):
    if ids.dtype == torch.float32:
        return "f32ids"
    if ids.dtype in (torch.int64, torch.uint64):
        return "u64ids"
    raise TypeError("ids must be float32 or (u)int64")


def lookup_sort_pend(table, ids, plan, pend, out=None, stream=None):
    """lookup_sort that also registers the batch in `pend` (first batch of a push_pull sequence)."""
    L = _lib.load()
    _require(table, torch.float32, "table")
    n = ids.numel()
    width = table.shape[1]
    if n > plan.capacity:
        raise ValueError("plan capacity %d < %d ids" % (plan.capacity, n))
    if out is None:
        out = torch.empty(tuple(ids.shape) + (width,), dtype=torch.float32, device=table.device)
    fn = getattr(L, "ha_lookup_sort_pend_" + _ids_kind(ids))
    check(fn(_ptr(table), table.shape[0], width, _ptr(ids), n, _ptr(out), _ptr(plan.ws), _ptr(pend.buf),
             _stream_ptr(stream)), "ha_lookup_sort_pend")
    plan.n = n
    plan._view = None
    plan.produced_on(stream)
    return out


def sgd_push_pull(table, plan_cur, grads, lr, pend_cur, next_ids=None, plan_next=None, pend_next=None,
                  next_out=None, stream=None):
    """ONE launch: sgd_apply_finish(plan_cur, grads) and, behind it in effect, lookup_sort(next_ids) ->
    (next_out, plan_next).  plan_cur must have been sorted by lookup_sort_pend / sgd_push_pull with
    pend_cur as its pending table.  next_ids=None: the last batch (apply only).  Returns next_out."""
    L = _lib.load()
    _require(table, torch.float32, "table")
    _require(grads, torch.float32, "grads")
    width = table.shape[1]
    if next_ids is None:
        check(L.ha_sgd_push_pull_f32ids(_ptr(table), table.shape[0], width, _ptr(plan_cur.ws), plan_cur.n,
                                        _ptr(grads), ctypes.c_float(lr), _ptr(pend_cur.buf), None, 0, None,
                                        None, None, _stream_ptr(stream)), "ha_sgd_push_pull")
        plan_cur.produced_on(stream)
        return None
    n = next_ids.numel()
    if n > plan_next.capacity:
        raise ValueError("plan capacity %d < %d ids" % (plan_next.capacity, n))
    if next_out is None:
        next_out = torch.empty(tuple(next_ids.shape) + (width,), dtype=torch.float32, device=table.device)
    fn = getattr(L, "ha_sgd_push_pull_" + _ids_kind(next_ids))
    check(fn(_ptr(table), table.shape[0], width, _ptr(plan_cur.ws), plan_cur.n, _ptr(grads),
             ctypes.c_float(lr), _ptr(pend_cur.buf), _ptr(next_ids), n, _ptr(next_out), _ptr(plan_next.ws),
             _ptr(pend_next.buf), _stream_ptr(stream)), "ha_sgd_push_pull")
    plan_next.n = n
    plan_next._view = None
    plan_cur.produced_on(stream)
    plan_next.produced_on(stream)
    return next_out


def check_handoff(plans):
    """Raises if any of `plans` carries the sticky hand-off time-out flag of ha_sgd_push_pull_* (a gather wave gave
    up waiting for a row of the previous batch, so a lookup may have returned stale rows).  Synchronises with the
    streams that produced the plans: call it where the training loop synchronises anyway (logging, evaluation)."""
    for p in plans:
        p._sync_producer()
        p._raise_if_handoff_timed_out()


class SortAhead:
    """Schedule for batches beyond the one-launch regime (more than 18,432 ids): the stable sort of batch k+1 runs
    on a side stream beside the lookup and the sparse SGD of batch k -- it needs nothing from them, and all of them
    are chains of small latency-bound launches (112.8 -> 80.0 us per step at 106,496 ids,
    62.4 -> 55.8 us at 26,624).  Results are those of lookup + sort + sgd_apply_finish on one stream.

        sa = SortAhead(table, capacity, lr)
        sa.begin(ids0)                         # sort of the first batch
        for k in ...:
            out = sa.lookup(ids_k, next_ids=ids_{k+1})   # rows of batch k; the sort of batch k+1 starts beside it
            ... dense model ...
            sa.apply(grads_k)                  # sparse SGD + plan finish of batch k; joins the side stream
    `sa.plan` is the plan of the batch last applied (finished)."""

    def __init__(self, table, capacity, lr, device=None):
        _require(table, torch.float32, "table")
        self.table, self.lr = table, float(lr)
        self.device = table.device if device is None else torch.device(device)
        self.plans = [IndexPlan(capacity, self.device), IndexPlan(capacity, self.device)]
        self.side = torch.cuda.Stream(device=self.device)
        self.k = 0
        self.plan = None

    def _sort(self, plan, ids):
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)      # the plan buffer is free once everything queued so far has run
        plan.sort(ids.reshape(-1), stream=self.side, key_limit=self.table.shape[0])

    def begin(self, ids0):
        self.k = 0
        self._sort(self.plans[0], ids0)
        torch.cuda.current_stream(self.device).wait_stream(self.side)

    def lookup(self, ids, next_ids=None, out=None):
        if next_ids is not None:
            self._sort(self.plans[(self.k + 1) % 2], next_ids)
        return embedding_lookup(self.table, ids, out=out)

    def apply(self, grads):
        cur = torch.cuda.current_stream(self.device)
        plan = self.plans[self.k % 2]
        sgd_apply_finish(self.table, plan, grads, self.lr)
        cur.wait_stream(self.side)      # the next step's apply needs the sort that ran beside this one
        self.plan = plan
        self.k += 1
        return plan


# ---- the step with two batches of lookahead: nothing waits inside the launch --------------------------
def step_max_ids():
    return int(_lib.load().ha_step_max_ids())


class StepPipeline:
    """Drives ha_step_* (include/herald_amd.h) over a stream of id batches: one launch per training step that
    applies the sparse SGD of batch k (cpu_SGDOptimizerSparseUpdate order), writes the rows of batch k+1 after
    that update (forwarded from the applying waves' registers where both batches name a row), finishes the plan
    of batch k+2 and sorts batch k+3.  Four plans and four key tables rotate.

        pipe = StepPipeline(table, capacity, lr)
        out0 = pipe.start(ids0, ids1, ids2)           # rows of batch 0 (three launches)
        out1 = pipe.step(grads0, ids3)                # apply 0, rows of batch 1, sort 3   (one launch)
        out2 = pipe.step(grads1, ids4) ...            # ahead_ids=None at the end of the stream
    `pipe.plan_of(j)` is the plan of batch j (finished -- unique keys / inverse / counts -- once the call that
    applies batch j-2 has been made)."""

    NPLAN, NTAB = 4, 4

    def __init__(self, table, capacity, lr, device=None):
        L = _lib.load()
        _require(table, torch.float32, "table")
        self.table, self.lr = table, float(lr)
        self.device = table.device if device is None else torch.device(device)
        if capacity > step_max_ids():
            raise ValueError("ha_step_* takes at most %d ids per batch (got capacity %d): use sgd_push_pull / "
                             "lookup_sort + sgd_apply_finish" % (step_max_ids(), capacity))
        self.capacity = capacity
        self.plans = [IndexPlan(capacity, self.device) for _ in range(self.NPLAN)]
        self.tab_bytes = int(L.ha_step_tab_bytes())
        self.tabs = torch.empty(self.NTAB * self.tab_bytes, dtype=torch.uint8, device=self.device)
        self.c = None            # index of the batch the next step() applies
        self.n = {}              # batch index -> number of ids (batches in flight)
        self.shape = {}
        self.reset()

    def _tab(self, c):
        return self.tabs.data_ptr() + (c % self.NTAB) * self.tab_bytes

    def plan_of(self, c):
        return self.plans[c % self.NPLAN]

    def reset(self, stream=None):
        L = _lib.load()
        for t in range(self.NTAB):
            check(L.ha_step_tab_reset(self._tab(t), _stream_ptr(stream)), "ha_step_tab_reset")
        if stream is None:
            torch.cuda.current_stream(self.device).synchronize()
        self.c, self.n, self.shape = None, {}, {}
        return self

    def launch(self, c, n_cur, grads, n_next, out, n_fin, ahead_ids, stream=None):
        """Call c of the stream, stateless (for callers that replay captured launches and keep the batch sizes
        themselves): batch c (n_cur ids) is applied, the rows of batch c+1 (n_next ids) go to `out`, the plan of
        batch c+2 (n_fin ids) is finished and its key table filled, `ahead_ids` = batch c+3 is sorted."""
        L = _lib.load()
        t = self.table
        width = t.shape[1]
        n_ahead = 0 if ahead_ids is None else ahead_ids.numel()
        if max(n_cur, n_next, n_fin, n_ahead) > self.capacity:
            raise ValueError("plan capacity %d < %d ids" % (self.capacity, max(n_cur, n_next, n_fin, n_ahead)))
        if n_cur:
            _require(grads, torch.float32, "grads")
            if grads.numel() != n_cur * width:
                raise ValueError("grads must hold %d x %d values" % (n_cur, width))
        if n_next:
            _require(out, torch.float32, "out")
            if out.numel() != n_next * width:
                raise ValueError("out must hold %d x %d values" % (n_next, width))
        kind = _ids_kind(ahead_ids) if n_ahead else "f32ids"
        fn = getattr(L, "ha_step_" + kind)
        check(fn(_ptr(t), t.shape[0], width,
                 _ptr(self.plan_of(c).ws) if n_cur else None, n_cur, _ptr(grads) if n_cur else None,
                 ctypes.c_float(self.lr), self._tab(c) if n_cur else None,
                 _ptr(self.plan_of(c + 1).ws) if n_next else None, n_next, _ptr(out) if n_next else None,
                 self._tab(c + 1) if n_next else None,
                 _ptr(self.plan_of(c + 2).ws) if n_fin else None, n_fin, self._tab(c + 2) if n_fin else None,
                 _ptr(ahead_ids) if n_ahead else None, n_ahead,
                 _ptr(self.plan_of(c + 3).ws) if n_ahead else None,
                 self._tab(c + 3), _stream_ptr(stream)), "ha_step")
        if n_fin:
            self.plan_of(c + 2).produced_on(stream)
        if n_ahead:
            pl = self.plan_of(c + 3)
            pl.n = n_ahead
            pl._view = None
            pl.produced_on(stream)

    def _call(self, c, grads, ahead_ids, out, stream):
        n_cur, n_next, n_fin = self.n.get(c, 0), self.n.get(c + 1, 0), self.n.get(c + 2, 0)
        n_ahead = 0 if ahead_ids is None else ahead_ids.numel()
        if n_next and out is None:
            out = torch.empty(tuple(self.shape[c + 1]) + (self.table.shape[1],), dtype=torch.float32,
                              device=self.table.device)
        self.launch(c, n_cur, grads, n_next, out, n_fin, ahead_ids, stream)
        if n_cur:
            del self.n[c]
        if n_ahead:
            self.n[c + 3] = n_ahead
            self.shape[c + 3] = tuple(ahead_ids.shape)
        self.c = c + 1
        return out if n_next else None

    def start(self, ids0, ids1=None, ids2=None, out=None, stream=None):
        """Sorts batches 0, 1 and 2 and returns the rows of batch 0."""
        self._call(-3, None, ids0, None, stream)
        self._call(-2, None, ids1, None, stream)
        return self._call(-1, None, ids2, out, stream)

    def step(self, grads, ahead_ids=None, out=None, stream=None):
        """Applies `grads` of the current batch; returns the rows of the next batch (None at the end)."""
        if self.c is None or self.c not in self.n:
            raise RuntimeError("StepPipeline.step without a batch in flight (call start first)")
        return self._call(self.c, grads, ahead_ids, out, stream)


def qbig_max_ids():
    """Largest batch of the work-queue step's wide path (hash buckets of at most qstep_max_ids() ids each)."""
    return int(_lib.load().ha_qbig_max_ids())


class WidePlan:
    """The plan of a batch of more than qstep_max_ids() ids as ha_qbig_plan_batch_* leaves it: the batch cut into hash
    buckets, per bucket the relations of an index plan (unique keys in hash-slot order, counts, segment starts local to
    the bucket) and ONE list of occurrence lists for the whole batch (positions of the batch, ascending inside every
    key's list)."""

    def __init__(self, capacity, device):
        L = _lib.load()
        self.capacity, self.device = int(capacity), device
        self.ws = torch.zeros(int(L.ha_qbig_plan_bytes(self.capacity)), dtype=torch.uint8, device=device)
        torch.cuda.current_stream(device).synchronize()   # (the zero fill: users launch on streams of their own)
        self.buckets = int(L.ha_qbig_buckets(self.capacity))
        self.n = 0
        self._produced_on, self._view = 0, None

    def groups(self):
        """(unique keys, counts, occurrence lists) of the whole batch, bucket after bucket (host copies; synchronises)."""
        L = _lib.load()
        ptr = [ctypes.c_void_p() for _ in range(7)]
        check(L.ha_qbig_plan_view(_ptr(self.ws), self.capacity, *[ctypes.byref(p) for p in ptr]), "ha_qbig_plan_view")
        torch.cuda.synchronize(self.device)
        base = self.ws.data_ptr()

        def arr(p, count, dtype):
            off = p.value - base
            return self.ws[off:off + count * dtype.itemsize].view(dtype).cpu().numpy()
        P, n = self.buckets, self.n
        boff = arr(ptr[0], P + 1, torch.int32).astype(np.int64)
        hdr = arr(ptr[1], P * 32, torch.int64).reshape(P, 32)
        uniq = arr(ptr[2], max(n, 1), torch.int32).astype(np.int64) & 0xFFFFFFFF
        counts = arr(ptr[3], max(n, 1), torch.int32).astype(np.int64)
        gperm = arr(ptr[5], max(n, 1), torch.int32).astype(np.int64)
        meta = arr(ptr[6], 1, torch.int32)
        keys, cnts, lists = [], [], []
        for p in range(P):
            U = int(hdr[p, 0])
            o = int(boff[p])
            keys.append(uniq[o:o + U])
            cnts.append(counts[o:o + U])
            lists.append(gperm[o:int(boff[p + 1])])
        return {"overflow": int(meta[0]), "bucket_sizes": np.diff(boff), "uniq": np.concatenate(keys) if keys else uniq[:0],
                "counts": np.concatenate(cnts) if cnts else counts[:0], "perm": np.concatenate(lists) if lists else gperm[:0]}


def qstep_max_ids():
    return int(_lib.load().ha_qstep_max_ids())


class SortAheadPipeline:
    """Training steps on batches too large for the work-queue step (more than ha_qstep_max_ids ids: BASELINE configs[2] /
    [3] hand a GPU 106,496 / 26,624 ids per step): per step one gather launch and one apply + finish launch on the
    caller's stream, with the index plans SORTED a block of batches ahead on a side stream.  The sort depends on the ids
    only (the data loader has them a block ahead, as for QueueStepPipeline), so the two streams meet once per BLOCK -- one
    event each way -- instead of twice per step: a cross-stream dependency costs 10-20 us on this part, as much as the
    sort of a 26,624-id batch it is meant to hide.

        pipe = SortAheadPipeline(table, max_ids, lr, block=8)
        pipe.prepare_block(ids_of_steps_0_to_7)              # side stream
        for k in range(8):                                   # caller's stream
            if k == 0: pipe.prepare_block(ids_of_steps_8_to_15)
            out = pipe.lookup(k, ids[k]);  ...;  pipe.apply(k, grads)

    Results are those of embedding_lookup + sgd_apply_finish step by step (bit-exact; tolerance mode as set)."""

    NBLOCKS = 3          # blocks of plans in flight: being applied, sorted and waiting, being sorted

    def __init__(self, table, max_ids, lr, block=8, key_limit=None, device=None, finish_ahead=None):
        _require(table, torch.float32, "table")
        self.table, self.lr, self.block = table, float(lr), int(block)
        self.device = table.device if device is None else torch.device(device)
        self.key_limit = int(table.shape[0]) if key_limit is None else int(key_limit)
        # finish_ahead=True: sort AND finish on the side stream, the apply maps its waves to unique keys
        # (ha_sgd_apply_finished).  Alone that apply is a third faster than the apply-and-finish launch at 106,496 ids;
        # beside the sorts it is not -- its 512 resident workgroups leave the sort's launches no room, 70 against 60 us per
        # step (tools/cfgc_bench.py) -- so it is off unless asked for.
        self.finish_ahead = bool(finish_ahead)
        self.plans = [[IndexPlan(max_ids, self.device) for _ in range(self.block)] for _ in range(self.NBLOCKS)]
        self.side = torch.cuda.Stream(device=self.device)      # (a high-priority side stream: no difference, 47.4 / 47.2 us)
        self._sorted = [torch.cuda.Event() for _ in range(self.NBLOCKS)]      # side -> main: the block's plans are sorted
        self._freed = [None] * self.NBLOCKS                                   # main -> side: the block's plans were applied
        self._first = [0] * self.NBLOCKS       # number of the first step of the block each slot holds
        self._count = [0] * self.NBLOCKS
        self._next_block = 0
        self._waited = -1

    def prepare_block(self, ids_list):
        """Sort the plans of the next `len(ids_list)` (<= block) steps on the side stream."""
        if not 0 < len(ids_list) <= self.block:
            raise ValueError("a block holds 1..%d batches" % self.block)
        b = self._next_block
        slot = b % self.NBLOCKS
        # the side stream follows the caller's stream up to here (one event per block; also what makes the pair of streams
        # capturable into one hipGraph): the sorts run beside the steps enqueued after this call
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            if self._freed[slot] is not None:
                self.side.wait_event(self._freed[slot])      # its plans' previous users have finished
            for i, ids in enumerate(ids_list):
                if self.finish_ahead:       # sort AND finish beside the steps: the apply then goes by unique key
                    self.plans[slot][i].build(ids, stream=self.side, key_limit=self.key_limit)
                else:
                    self.plans[slot][i].sort(ids, stream=self.side, key_limit=self.key_limit)
            self._sorted[slot].record(self.side)
        self._first[slot] = (self._first[(b - 1) % self.NBLOCKS] + self._count[(b - 1) % self.NBLOCKS]) if b else 0
        self._count[slot] = len(ids_list)
        self._next_block = b + 1
        return b

    def _plan(self, k, stream):
        for b in range(max(self._next_block - self.NBLOCKS, 0), self._next_block):
            slot = b % self.NBLOCKS
            if self._first[slot] <= k < self._first[slot] + self._count[slot]:
                if b > self._waited:           # once per block
                    (stream or torch.cuda.current_stream(self.device)).wait_event(self._sorted[slot])
                    self._waited = b
                return slot, k - self._first[slot]
        raise ValueError("step %d is not in a prepared block" % k)

    def lookup(self, k, ids, out=None, stream=None):
        self._plan(k, stream)                  # the block's sorts are ordered before anything of the step
        return embedding_lookup(self.table, ids, out=out, stream=stream)

    def apply(self, k, grads, stream=None):
        slot, i = self._plan(k, stream)
        if self.finish_ahead:
            sgd_apply(self.table, self.plans[slot][i], grads, self.lr, stream=stream, finished=True)
            self.plans[slot][i].produced_on(self.side)
        else:
            sgd_apply_finish(self.table, self.plans[slot][i], grads, self.lr, stream=stream)
        if i == self._count[slot] - 1:         # last step of the block: its plans may be overwritten
            ev = torch.cuda.Event()
            ev.record(stream or torch.cuda.current_stream(self.device))
            self._freed[slot] = ev
        return self.plans[slot][i]


class QueueStepPipeline:
    """Drives ha_qapply / ha_qplan_batch_* / ha_qqueue_batch (include/herald_amd.h, csrc/qstep.hip) over a stream of id
    batches.  Every training step is ONE launch on the caller's stream: it applies the sparse SGD of batch c
    (cpu_SGDOptimizerSparseUpdate order) and writes the rows of batch c+1 after that update, item by item from a work
    queue (one wave per unique key / column slice, no probing, no waiting).  The queues and the per-batch plans behind
    them are prepared a BLOCK of `block` steps at a time: at the start of block b the plans of the batches of block b+2
    (one workgroup each, one launch) and the queues of the steps of block b+1 (two workgroups each, one launch) go to a
    side stream and run beside the steps of block b.  Ids are needed LOOKAHEAD = 3 * block batches ahead.

        overlap=False  block = 1 and everything on the caller's stream (three launches per step, LOOKAHEAD = 3).

        pipe = QueueStepPipeline(table, capacity, lr, block=8)
        out0 = pipe.start(ids[:pipe.LOOKAHEAD])       # the first LOOKAHEAD batches; returns the rows of batch 0
        out1 = pipe.step(grads0, ids[LOOKAHEAD])      # apply 0, rows of batch 1; one more batch enters the pipeline
        ...                                           # ahead_ids=None once the stream of batches ends

    Keys with 16 or more occurrences in a batch are applied as `row - tree_sum(lr * g)` (deterministic, within the 1e-5
    relative BASELINE.json allows for accumulated gradients); below 16 occurrences the result is the reference's serial
    chain bit for bit.  `plan_of(j)` holds the unique keys / counts / inverse / occurrence lists of batch j with the
    unique keys in hash-slot order (not np.unique's order) once its block has been prepared."""

    def __init__(self, table, capacity, lr, device=None, block=8, overlap=True, sync="events", min_flags_block=8):
        """sync: how the preparation stream and the caller's stream are ordered when overlap is on.
        "events" (default): an event record and an event wait on the caller's stream at every block start -- safe for any
        caller (ids produced by work queued on the caller's stream, steps captured into hipGraphs).
        "flags": NOTHING but apply launches on the caller's stream -- every queue carries the epoch of its step and the
        apply checks it before its first item, the last launch of a block completes an event of its own that the
        preparation stream waits for (ha_qapply_steps_sync, include/herald_amd.h): ~1 us per step less at blocks of 16.
        Requirements: the ids handed to prepare_block / step are COMPLETE on the device when they are handed over (nothing
        orders them behind work on the caller's stream), and the steps are enqueued eagerly (not captured)."""
        L = _lib.load()
        _require(table, torch.float32, "table")
        self.table, self.lr = table, float(lr)
        self.device = table.device if device is None else torch.device(device)
        # batches beyond one plan workgroup's reach take the WIDE path: hash buckets of the batch planned and joined side
        # by side (ha_qbig_*, csrc/qstep.hip), the same apply launch
        self.wide = capacity > qstep_max_ids()
        if capacity > qbig_max_ids():
            raise ValueError("ha_qstep_* takes at most %d ids per batch (got capacity %d): use SortAheadPipeline / "
                             "lookup_sort + sgd_apply_finish" % (qbig_max_ids(), capacity))
        if table.shape[1] % 4 != 0:
            raise ValueError("ha_qstep_* needs rows of a multiple of 4 floats")
        with torch.cuda.device(self.device):
            if L.ha_qstep_init() < 0:      # LDS attributes + the lane-order probe, outside any stream capture
                check(-1, "ha_qstep_init")
        self.capacity = max(int(capacity), 1)
        self.overlap = bool(overlap)
        if sync not in ("events", "flags"):
            raise ValueError("sync must be 'events' or 'flags'")
        self.sync = sync if self.overlap else "events"
        # "flags" needs the queues' builder a comfortable block ahead of the steps: an apply launch that has to WAIT for its queue
        # polls with every compute unit taken, and the builder's workgroups start only on an empty one (csrc/qstep.hip,
        # qapply_lists) -- sporadic 2 s time-outs at blocks of 2 steps, 43 us per step at blocks of 4 (14 by events).  Blocks of
        # fewer than 8 steps are ordered by events whatever was asked for.
        # (min_flags_block: tests of the mechanism itself lower it)
        if self.sync == "flags" and int(block) < int(min_flags_block):
            self.sync = "events"
        self._done_ev, self._cev_pool = {}, []       # flags: block -> library event completed by its last apply launch
        # done-events the preparation stream has been told to wait for, with a marker recorded on that stream behind the wait:
        # an event is handed to a later launch only once its marker has completed (the wait has certainly been consumed -- the
        # host runs many blocks ahead of the device, and a wait that is resolved against an event RE-BOUND to a later launch
        # waits for a step whose queue this very stream has yet to build: a deadlock that ends in the apply's 2 s time-out;
        # seen at blocks of 2 steps)
        self._cev_busy = []
        self.block = int(block) if self.overlap else 1
        if not 1 <= self.block <= 64:
            raise ValueError("block must be 1..64")
        self.LOOKAHEAD = 3 * self.block
        self.NPLAN, self.NQUEUE = 4 * self.block, 2 * self.block
        self.ROTATION = 4 * self.block     # plans and queues of call c and call c + ROTATION are the same
        self.plans = [(WidePlan if self.wide else IndexPlan)(self.capacity, self.device) for _ in range(self.NPLAN)]
        self.queue_bytes = int(L.ha_qstep_queue_bytes(self.capacity, table.shape[1]))
        self.queues = torch.zeros(self.NQUEUE * self.queue_bytes, dtype=torch.uint8, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()   # (the zero fill: the builders run on the side stream)
        # the preparation runs beside the steps: its stream may be given the LOWEST priority (HA_QSIDE_PRIO=low; =high: the highest -- A/B knobs), so that the
        # apply launches win the dispatcher whenever both have workgroups to place
        side_prio = 0
        if os.environ.get("HA_QSIDE_PRIO") in ("low", "high"):
            try:
                lo_hi = torch.cuda.Stream.priority_range()
                side_prio = max(lo_hi) if os.environ["HA_QSIDE_PRIO"] == "low" else min(lo_hi)
            except Exception:      # noqa: BLE001
                side_prio = 0
        self.side = torch.cuda.Stream(device=self.device, priority=side_prio) if self.overlap else None
        # {wave items, workgroup items, copy items} (+ 1; 0 = not built yet) of the queue of step c, written to pinned host
        # memory by the launch that builds it, in a ring long enough that a build still in flight cannot write into the
        # slot of a later step: queues are built a block ahead, so the host usually knows the numbers when it enqueues
        # the step and sizes the launch by them (a hint: a missing or stale one costs time, never correctness)
        self.COUNTS = 8192
        self.counts = torch.zeros((self.COUNTS, 4), dtype=torch.int32).pin_memory()
        self._counts_np = self.counts.numpy()
        self._counts_c = (ctypes.c_int32 * (self.COUNTS * 4)).from_address(self.counts.data_ptr())   # cheap scalar reads
        self._counts_base = self.counts.data_ptr()
        self._L = L
        self._plan_ptr = [p.ws.data_ptr() for p in self.plans]
        self._queue_ptr = [self.queues.data_ptr() + q * self.queue_bytes for q in range(self.NQUEUE)]
        self._ev_pool = []
        self._last_items = 0
        self.fallbacks, self._fb_plan, self._wide_ids = 0, None, {}
        self._enq_last = None        # the last step handed to the device (ordering fallback of the flags mode)
        self.reset()

    def close(self):
        """Releases the library events of the flags mode (ha_event_create)."""
        for ev in list(self._done_ev.values()) + self._cev_pool + [e for e, _ in self._cev_busy]:
            self._L.ha_event_destroy(ev)
        self._done_ev, self._cev_pool, self._cev_busy = {}, [], []

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass

    # ---- bookkeeping ------------------------------------------------------------------------------------------
    def reset(self, stream=None):
        if self._done_ev:
            if self.side is not None:       # (nothing may be waiting for them, but an event in flight is not handed out either)
                marker = self._event() if hasattr(self, "_ev_pool") else torch.cuda.Event()
                marker.record(self.side)
                self._cev_busy.extend((e, marker) for e in self._done_ev.values())
            else:
                self._cev_pool.extend(self._done_ev.values())
        self._done_ev = {}
        self.c, self.n, self.shape, self.ids = None, {}, {}, {}
        self._enq_last, self._covered = None, -(1 << 60)
        self._ev_side = {}          # block index -> event behind the side work launched at its start
        self._held = {}             # block index -> id tensors its plan launch reads (kept alive, not record_stream'ed)
        return self

    def plan_of(self, b):
        return self.plans[b % self.NPLAN]

    def _queue(self, c):
        return self.queues.data_ptr() + (c % self.NQUEUE) * self.queue_bytes

    def queue_header(self, c):
        """{wave items, workgroup items, long, medium, small, copy items} of the queue step c reads (host copy;
        synchronises)."""
        off = (c % self.NQUEUE) * self.queue_bytes
        torch.cuda.synchronize(self.device)
        h = self.queues[off:off + 40].view(torch.int32).cpu().tolist()
        d = dict(zip(("wave_items", "workgroup_items", "long", "medium", "small", "copy_items"), h))
        d["overflow"] = int(h[8] != 0 or h[9] != 0)
        return d

    def overflowed(self):
        """True if any queue built so far counted more items than it holds (the layout's bounds exclude it; the builder
        raises a sticky word in pinned memory instead of dropping items silently)."""
        return bool((self._counts_np[:, 3] & ~4).any())      # (4 = "this step takes the sorted plan": handled, not an error)

    def _raise_if_failed(self, steps):
        """The pinned error words of `steps`: 8 = an apply launch gave up waiting for its queue's epoch tag."""
        cc, ring = self._counts_c, self.COUNTS
        for j in steps:
            f = cc[4 * (j % ring) + 3]
            if f & ~4:
                raise RuntimeError("QueueStepPipeline: step %d failed on the device (flags %d: 1 = queue overflow, 2 = an "
                                   "occurrence list out of position order, 8 = an apply launch gave up waiting -- its queue was "
                                   "never completed, or an item it depends on never finished; the table is partly updated)"
                                   % (j, f))

    # ---- the preparation of a block ------------------------------------------------------------------------------
    def prepare_block(self, b, ids_of, stream=None, ph=None):
        """Start of block b (call before the first step of the block, NOT inside a stream capture when overlap is on):
        the plans of the batches of block b+2 and the queues of the steps of block b+1 are enqueued (side stream), and the
        caller's stream is made to wait for the side work enqueued at the start of block b-1 -- the queues of block b.
        `ids_of(j)` = device id tensor of batch j, or None outside the stream of batches.  (Host cost matters here: the
        call sits between two steps of the caller's stream -- ~60 us for a block of 16.)"""
        L = self._L
        rows, width = self.table.shape
        main = stream if stream is not None else torch.cuda.current_stream(self.device)
        B = self.block
        s = self.side if self.overlap else main
        sp = _stream_ptr(s)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        flags = self.sync == "flags"
        # a launch of the block before last that gave up (flags: a queue that never became ready) has left its word in pinned
        # memory by now: stop here
        self._raise_if_failed(range(max((b - 2) * B, 0), max((b - 1) * B, 0)))
        if self.overlap and flags:
            # the steps of block b-1 are complete (their last launch carries the event): the plans / queues about to be
            # rewritten are free.  Nothing is enqueued on the caller's stream.
            cev = self._done_ev.pop(b - 1, None)
            if cev is not None:
                check(L.ha_stream_wait_event(sp, cev), "ha_stream_wait_event")
                marker = self._event()
                marker.record(s)
                self._cev_busy.append((cev, marker))
                self._covered = b * B - 1
            elif self._enq_last is not None and self._enq_last > self._covered:
                # steps that may still read what is about to be rewritten were enqueued without an event of their own (a
                # debug launch, a chunk that did not end with its block): order the side stream behind the caller's stream
                ev = self._event()
                ev.record(main)
                s.wait_event(ev)
                self._ev_pool.append(ev)
                self._covered = self._enq_last
        elif self.overlap:
            ev = self._event()
            ev.record(main)                    # the buffers about to be rewritten are free, the ids are there
            s.wait_event(ev)
        # plans of block b+2
        sel = {"f32ids": [], "u64ids": []}
        held = []
        for j in range((b + 2) * B, (b + 3) * B):
            t = ids_of(j)
            if t is None:
                continue
            m = t.numel()
            if m == 0:
                continue
            if m > self.capacity:
                raise ValueError("plan capacity %d < %d ids" % (self.capacity, m))
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("ids must be contiguous device tensors")
            sel[_ids_kind(t)].append((j, t, m))
            held.append(t)
            self.n[j] = m
            if self.wide:            # (a step whose batch cannot be bucketed falls back to a sorted plan of the ids)
                self._wide_ids[j] = t
                self._wide_ids.pop(j - 5 * B, None)
        for kind, lst in sel.items():
            if not lst:
                continue
            cnt = len(lst)
            ids_arr = (vp * cnt)(*[t.data_ptr() for _, t, _ in lst])
            n_arr = (i64 * cnt)(*[m for _, _, m in lst])
            pl_arr = (vp * cnt)(*[self._plan_ptr[j % self.NPLAN] for j, _, _ in lst])
            if ph is not None and kind == "f32ids":
                j0, t0, m0 = lst[0]
                check(L.ha_debug_qprep_f32ids(rows, width, _ptr(t0), m0, _ptr(self.plan_of(j0).ws), None, 0,
                                              None, 0, None, self.capacity, _ptr(ph), sp), "ha_debug_qprep")
            if self.wide:
                check(getattr(L, "ha_qbig_plan_batch_" + kind)(ids_arr, n_arr, pl_arr, self.capacity, cnt, sp),
                      "ha_qbig_plan_batch")
            else:
                check(getattr(L, "ha_qplan_batch_" + kind)(ids_arr, n_arr, pl_arr, cnt, sp), "ha_qplan_batch")
            for j, _, m in lst:
                pl = self.plans[j % self.NPLAN]
                pl.n = m
                pl._view = None
                pl._produced_on = sp.value if sp.value is not None else 0
        # the ids are read on the side stream: keep them alive until the plans of the block after next are enqueued
        self._held[b] = held
        self._held.pop(b - 2, None)
        # queues of the steps of block b+1 (step j: batch j applied, batch j+1 looked up)
        nget = self.n.get
        steps = [j for j in range((b + 1) * B, (b + 2) * B) if nget(j, 0) or nget(j + 1, 0)]
        if steps:
            cnt = len(steps)
            pp, NP, NQ = self._plan_ptr, self.NPLAN, self.NQUEUE
            pa = (vp * cnt)(*[pp[j % NP] if nget(j, 0) else None for j in steps])
            na = (i64 * cnt)(*[nget(j, 0) for j in steps])
            pg = (vp * cnt)(*[pp[(j + 1) % NP] if nget(j + 1, 0) else None for j in steps])
            ng = (i64 * cnt)(*[nget(j + 1, 0) for j in steps])
            qs = (vp * cnt)(*[self._queue_ptr[j % NQ] for j in steps])
            if ph is not None:
                check(L.ha_debug_qprep_f32ids(rows, width, None, 0, None, pa[0], na[0], pg[0], ng[0], qs[0], self.capacity,
                                              _ptr(ph), sp), "ha_debug_qprep")
            base, ring, cc = self._counts_base, self.COUNTS, self._counts_c
            for j in steps:
                at = 4 * (j % ring)
                cc[at] = 0
                cc[at + 2] = 0
                cc[at + 3] &= ~4         # "takes the sorted plan" belongs to the step that had the slot before
            cs = (vp * cnt)(*[base + 16 * (j % ring) for j in steps])
            eps = (ctypes.c_uint32 * cnt)(*[self._epoch(j) for j in steps])
            if self.wide:
                check(L.ha_qbig_queue_batch(rows, width, pa, na, pg, ng, qs, self.capacity, cnt, cs, eps, sp),
                      "ha_qbig_queue_batch")
            else:
                check(L.ha_qqueue_batch_epochs(rows, width, pa, na, pg, ng, qs, self.capacity, cnt, cs, eps, sp),
                      "ha_qqueue_batch")
        if self.overlap and not flags:
            ev = self._event()
            ev.record(s)
            self._ev_side[b] = ev
            ready = self._ev_side.pop(b - 1, None)
            if ready is not None:
                main.wait_event(ready)
                self._ev_pool.append(ready)

    def _event(self):
        return self._ev_pool.pop() if self._ev_pool else torch.cuda.Event()

    @staticmethod
    def _epoch(j):
        """The tag of step j's queue (non-zero, distinct for the steps that share a queue slot)."""
        return (j + (1 << 24)) & 0xFFFFFFFF

    def _block_done_event(self, c):
        """flags: the library event the launch of step c has to complete if c is the last step of its block, else None."""
        if self.sync != "flags" or (c + 1) % self.block != 0:
            return None
        if not self._cev_pool and self._cev_busy and self._cev_busy[0][1].query():
            old_ev, marker = self._cev_busy.pop(0)      # the preparation stream is past its wait for this one
            self._ev_pool.append(marker)
            self._cev_pool.append(old_ev)
        ev = self._cev_pool.pop() if self._cev_pool else ctypes.c_void_p(self._L.ha_event_create())
        if not ev:
            raise RuntimeError("ha_event_create failed")
        old = self._done_ev.pop(c // self.block, None)
        if old is not None:
            self._cev_pool.append(old)
        self._done_ev[c // self.block] = ev
        return ev

    def _err_ptr(self, c):
        return ctypes.c_void_p(self._counts_base + 16 * (c % self.COUNTS) + 12)

    def wave_items(self, c):
        """Wave + copy items of the queue of step c if its numbers have landed in pinned memory, else -1."""
        at = 4 * (c % self.COUNTS)
        if self._counts_c[at + 3]:
            raise RuntimeError("ha_qqueue_batch: the work queue of step %d is unusable (flags %d: 1 = overflow, 2 = an "
                               "occurrence list out of position order, 4 = a hash bucket of a wide batch holds more than "
                               "%d ids)" % (c, self._counts_c[at + 3], qstep_max_ids()))
        if os.environ.get("HA_QHINT") == "0":
            return -1
        w, cp = self._counts_c[at], self._counts_c[at + 2]
        return w + cp - 2 if w > 0 and cp > 0 else -1

    # ---- one step ----------------------------------------------------------------------------------------------------
    def apply(self, c, grads, out, stream=None, dbg=None, n_cur=None, n_next=None):
        """Step c on `stream`: batch c applied with `grads`, the rows of batch c+1 to `out`.  n_cur / n_next: the ids of
        the two batches (default: as recorded when their plans were prepared)."""
        L = self._L
        t = self.table
        rows, width = t.shape
        n_cur = self.n.get(c, 0) if n_cur is None else int(n_cur)
        n_next = self.n.get(c + 1, 0) if n_next is None else int(n_next)
        if n_cur:
            _require(grads, torch.float32, "grads")
            if grads.numel() != n_cur * width:
                raise ValueError("grads must hold %d x %d values" % (n_cur, width))
        if n_next:
            _require(out, torch.float32, "out")
            if out.numel() != n_next * width:
                raise ValueError("out must hold %d x %d values" % (n_next, width))
        flags = self.sync == "flags" and dbg is None
        capturing = flags and torch.cuda.is_current_stream_capturing()
        if capturing:
            raise RuntimeError("QueueStepPipeline(sync='flags'): steps cannot be captured into a hipGraph (use sync='events')")
        done = self._block_done_event(c) if flags else None
        self._enq_last = c if self._enq_last is None else max(self._enq_last, c)
        if not (n_cur or n_next):
            if done is not None:         # the block's last step launches nothing: mark the point on the stream instead
                check(L.ha_event_record(done, _stream_ptr(stream)), "ha_event_record")
            return
        args = [_ptr(t), rows, width, _ptr(self.plan_of(c).ws) if n_cur else None, n_cur,
                _ptr(grads) if n_cur else None, ctypes.c_float(self.lr),
                _ptr(self.plan_of(c + 1).ws) if n_next else None, n_next, _ptr(out) if n_next else None,
                self._queue(c), self.capacity]
        if self.wide:
            # The host waits until queue c is complete (the builder runs a block ahead; its last bucket workgroup writes the
            # counts to pinned memory): only then is it known whether the batches of this step could be planned at all -- a
            # hash bucket that holds more than qstep_max_ids() ids cannot (flag 4), and such a step takes the sorted plan.
            flag, coop = self._wide_built(c)
            if flag & 4:
                self._wide_fallback(c, grads, out, stream, n_cur, n_next, done)
                return
            if flag:
                self.wave_items(c)       # raises
            check(L.ha_qbig_apply(*args, coop, self._epoch(c) if flags else 0, self._err_ptr(c) if flags else None, done,
                                  _stream_ptr(stream)), "ha_qbig_apply")
        elif flags:
            check(L.ha_qapply_sync(*args, self.wave_items(c), self._epoch(c), self._err_ptr(c), done, _stream_ptr(stream)),
                  "ha_qapply_sync")
        elif dbg is None:
            check(L.ha_qapply_sized(*args, self.wave_items(c), _stream_ptr(stream)), "ha_qapply")
        else:
            check(L.ha_debug_qapply(*args, _ptr(dbg), _stream_ptr(stream)), "ha_debug_qapply")

    def _wide_built(self, c, timeout=20.0):
        """Wide path: wait (host) for the pinned counts of queue c; -> (flags, workgroup items)."""
        import time
        at = 4 * (c % self.COUNTS)
        cc = self._counts_c
        if cc[at] == 0:
            t0 = time.perf_counter()
            while cc[at] == 0:
                if time.perf_counter() - t0 > timeout:
                    raise RuntimeError("QueueStepPipeline: the queue of step %d was not built within %.0f s (was its block "
                                       "prepared?)" % (c, timeout))
        return cc[at + 3], max(cc[at + 1] - 1, 0)

    def _wide_fallback(self, c, grads, out, stream, n_cur, n_next, done):
        """Step c without its queue: a batch it touches has a hash bucket beyond one plan workgroup's reach.  Batch c is
        applied through a SORTED plan (the reference's serial chain for every key: stricter than the queue's tolerance
        classes), then the rows of batch c+1 are gathered -- two launches + a sort instead of one launch, for this step only."""
        self.fallbacks += 1
        if n_cur:
            if self._fb_plan is None:
                self._fb_plan = IndexPlan(self.capacity, self.device)
            self._fb_plan.build(self._wide_ids[c], stream=stream)
            sgd_apply(self.table, self._fb_plan, grads.reshape(n_cur, -1), self.lr, stream=stream, finished=True)
        if n_next:
            embedding_lookup(self.table, self._wide_ids[c + 1].reshape(-1), out=out.reshape(n_next, -1), stream=stream)
        if done is not None:
            check(self._L.ha_event_record(done, _stream_ptr(stream)), "ha_event_record")

    def apply_call(self, c, grads, out, stream, n_cur, n_next, sized=True):
        """-> callable(step index) that enqueues the launch of the steps c, c + ROTATION, ... with the arguments converted
        once (the steps of a long run repeat every ROTATION steps with a fixed set of buffers; the per-call ctypes
        conversion of apply() costs more host time than the launch itself)."""
        if self.sync == "flags":
            raise RuntimeError("apply_call: use apply / apply_steps_call with sync='flags'")
        L = self._L
        t = self.table
        rows, width = t.shape
        hint = ctypes.c_int64(-1)
        args = (ctypes.c_void_p(t.data_ptr()), ctypes.c_int64(rows), ctypes.c_int64(width),
                ctypes.c_void_p(self.plan_of(c).ws.data_ptr() if n_cur else None), ctypes.c_int64(n_cur),
                ctypes.c_void_p(grads.data_ptr() if n_cur else None), ctypes.c_float(self.lr),
                ctypes.c_void_p(self.plan_of(c + 1).ws.data_ptr() if n_next else None), ctypes.c_int64(n_next),
                ctypes.c_void_p(out.data_ptr() if n_next else None), ctypes.c_void_p(self._queue(c)),
                ctypes.c_int64(self.capacity), hint, _stream_ptr(stream))
        fn = L.ha_qapply_sized

        def call(k):
            self._enq_last = k if self._enq_last is None else max(self._enq_last, k)
            if sized:       # not inside a graph capture: a captured launch keeps the grid it was captured with
                hint.value = self.wave_items(k)
            if fn(*args) != 0:
                check(-1, "ha_qapply")
        return call

    def apply_steps_call(self, c0, grads_list, out_list, stream, n, sized=True):
        """-> callable(first step index) that enqueues len(grads_list) consecutive steps c0, c0 + 1, ... (and the same
        phases ROTATION steps later) by ONE library call, arguments converted once; every batch has n ids."""
        L = self._L
        t = self.table
        rows, width = t.shape
        cnt = len(grads_list)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        pc = (vp * cnt)(*[self.plan_of(c0 + i).ws.data_ptr() for i in range(cnt)])
        pn = (vp * cnt)(*[self.plan_of(c0 + i + 1).ws.data_ptr() for i in range(cnt)])
        ns = (i64 * cnt)(*[n] * cnt)
        gs = (vp * cnt)(*[g.data_ptr() for g in grads_list])
        os_ = (vp * cnt)(*[o.data_ptr() for o in out_list])
        qs = (vp * cnt)(*[self._queue(c0 + i) for i in range(cnt)])
        hints = (i64 * cnt)(*[-1] * cnt)
        head = (vp(t.data_ptr()), i64(rows), i64(width), ctypes.c_float(self.lr), i64(self.capacity), i64(cnt))
        sp = _stream_ptr(stream)
        fn = L.ha_qapply_steps
        counts, ring = self._counts_c, self.COUNTS
        flags = self.sync == "flags"
        eps = (ctypes.c_uint32 * cnt)()
        fn_sync, epoch, last_err = L.ha_qapply_steps_counts, self._epoch, self._err_ptr
        B = self.block
        cs = (vp * cnt)()
        base = self._counts_base
        use_counts = os.environ.get("HA_QCOUNTS", "0") != "0"      # (same-box A/B: no gain for one launch per step)

        def call(k0):
            if flags and k0 // B != (k0 + cnt - 1) // B:
                # the block's done-event rides on the LAST launch of a call: a call that runs over a block boundary would
                # leave the block before without one (and the side stream would rewrite queues that steps still read)
                raise RuntimeError("apply_steps_call(sync='flags'): steps %d..%d cross a block boundary (block = %d)"
                                   % (k0, k0 + cnt - 1, B))
            self._enq_last = k0 + cnt - 1 if self._enq_last is None else max(self._enq_last, k0 + cnt - 1)
            if sized:
                for i in range(cnt):
                    at = 4 * ((k0 + i) % ring)
                    w, cp = counts[at], counts[at + 2]
                    if counts[at + 3]:
                        raise RuntimeError("ha_qqueue_batch: the work queue of step %d is unusable (flags %d: 1 = overflow, "
                                           "2 = an occurrence list out of position order)" % (k0 + i, counts[at + 3]))
                    if w > 0 and cp > 0:
                        self._last_items = w + cp - 2
                    # the queue of this step may not be built yet when the host is far ahead of the device: item counts
                    # vary by a few per cent from batch to batch, so the last known count (+ 6 %) sizes the launch then
                    hints[i] = w + cp - 2 if w > 0 and cp > 0 else (self._last_items * 17) // 16 if self._last_items > 0 else -1
                    cs[i] = base + 4 * at if use_counts else None
            if flags:
                for i in range(cnt):
                    eps[i] = epoch(k0 + i)
                done = self._block_done_event(k0 + cnt - 1)
                if fn_sync(*head, pc, ns, gs, pn, ns, os_, qs, hints, cs if sized else None, eps, last_err(k0 + cnt - 1), done,
                           sp) != 0:
                    check(-1, "ha_qapply_steps_counts")
            elif fn(*head, pc, ns, gs, pn, ns, os_, qs, hints, sp) != 0:
                check(-1, "ha_qapply_steps")
        return call

    # ---- the stream protocol --------------------------------------------------------------------------------------
    def _call(self, c, grads, ahead_ids, out, stream):
        if c % self.block == 0:
            self.prepare_block(c // self.block, lambda j: self.ids.get(j), stream)
            for j in [j for j in self.ids if j < c + 3 * self.block]:
                del self.ids[j]            # planned (enqueued): the plan holds what the steps need
        if ahead_ids is not None and ahead_ids.numel():
            self.ids[c + self.LOOKAHEAD] = ahead_ids
            self.shape[c + self.LOOKAHEAD] = tuple(ahead_ids.shape)
        n_next = self.n.get(c + 1, 0)
        if n_next and out is None:
            out = torch.empty(tuple(self.shape[c + 1]) + (self.table.shape[1],), dtype=torch.float32,
                              device=self.table.device)
        self.apply(c, grads, out, stream)
        for d in (self.n, self.shape):
            d.pop(c - 1, None)
        self.c = c + 1
        return out if n_next else None

    def start(self, ids, out=None, stream=None):
        """`ids`: the first LOOKAHEAD batches of the stream (a shorter list if the stream is shorter; None / empty
        entries are empty batches).  Returns the rows of batch 0."""
        ids = list(ids)
        if len(ids) > self.LOOKAHEAD:
            raise ValueError("start takes the first %d batches" % self.LOOKAHEAD)
        ids = ids + [None] * (self.LOOKAHEAD - len(ids))
        self.reset()
        res = None
        for k, b in enumerate(ids):
            c = k - self.LOOKAHEAD
            res = self._call(c, None, b, out if c == -1 else None, stream)
        return res

    def step(self, grads, ahead_ids=None, out=None, stream=None):
        """Applies `grads` of the current batch (None if that batch is empty); `ahead_ids` = the batch LOOKAHEAD ahead of
        it (None once the stream ends).  Returns the rows of the next batch (None at the end of the stream or if the
        next batch is empty)."""
        if self.c is None:
            raise RuntimeError("QueueStepPipeline.step before start")
        return self._call(self.c, grads, ahead_ids, out, stream)


def push_apply_finish(table, plan, grads, stream=None):
    _require(table, torch.float32, "table")
    _require(grads, torch.float32, "grads")
    check(_lib.load().ha_push_apply_finish(_ptr(table), table.shape[0], table.shape[1], _ptr(plan.ws),
                                           plan.n, _ptr(grads), _stream_ptr(stream)),
          "ha_push_apply_finish")
    plan.produced_on(stream)
    return table


# ---- reference-named symbols through the DLArray ABI ---------------------------------------------
def dl_call(name, arrays, scalars=(), stream=None):
    """Call a reference-named symbol: arrays -> DLArray*, then scalars, then DLStream*."""
    L = _lib.load()
    holders = [DLHolder(a) for a in arrays]
    sh = DLStreamHolder(stream)
    args = [h.handle for h in holders] + list(scalars) + [sh.handle]
    check(getattr(L, name)(*args), name)


class IndexedSlices:
    """Mirror of python/hetu/ndarray.py:503-611 with device-side dedup.

    indices: float32 device tensor (any shape), values: float32 [..., width].
    deduplicate() replaces (indices, values) by (sorted unique indices, occurrence-order row sums),
    exactly what the reference computes with np.unique + DeduplicateIndexedSlices / cpu_deduplicate.
    """

    def __init__(self, indices=None, values=None, dense_shape=None, push_indices=None):
        self.indices = indices
        self.values = values
        self.dense_shape = dense_shape
        self.push_indices = push_indices
        self.deduplicated = False

    def get_dense_shape(self):
        assert self.dense_shape is not None
        return self.dense_shape

    def get_sparse_shape(self):
        return tuple(self.values.shape)

    def update(self, indices, values, dense_shape, push_indices=None):
        self.indices = indices
        self.push_indices = push_indices
        self.values = values
        if self.dense_shape is not None:
            assert tuple(self.dense_shape) == tuple(dense_shape)
        else:
            self.dense_shape = dense_shape
        self.deduplicated = False

    def deduplicate(self, stream=None):
        ids = self.indices.reshape(-1)
        n = ids.numel()
        width = self.values.shape[-1]
        plan = IndexPlan(max(n, 1), device=ids.device).build(ids, stream)
        reduced = dedup_reduce(plan, self.values.reshape(n, width), stream=stream)
        uniq_f32, _ = plan.export_f32(stream)
        u = uniq_f32.numel()
        self.indices = uniq_f32
        self.values = reduced[:u]
        if self.push_indices is not None:
            pids = self.push_indices.reshape(-1)
            pplan = IndexPlan(max(pids.numel(), 1), device=pids.device).build(pids, stream)
            self.push_indices, _ = pplan.export_f32(stream)
        self.deduplicated = True
        return self

    def to_dense(self, stream=None):
        dense = torch.zeros(tuple(self.get_dense_shape()), dtype=torch.float32, device=self.values.device)
        dl_call("IndexedSlices2Dense", [self.values.contiguous(), self.indices.contiguous(), dense],
                stream=stream)
        return dense
