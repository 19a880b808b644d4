"""Host-side mirror of the `hetu_cache` plugin and of `CacheSparseTable`, on the device-resident cache.

Reference surfaces mirrored (paths relative to /root/reference):
  hetu_cache.LRUCache / LFUCache / LFUOptCache(limit, len, width, node_id)
        src/hetu_cache/src/python_api.cc:12-79 -- properties limit, width, perf, pull_bound, push_bound,
        perf_enabled; methods bypass, undo_bypass, embedding_lookup, embedding_update,
        embedding_update_with_push_keys, *_raw variants, count, size, keys, lookup, __repr__;
        every batch method returns a wait handle with .wait().
  CacheSparseTable(limit, length, width, node_id, policy, bound)   python/hetu/cstable.py:20-36

Differences that the boundary makes explicit:
  * the server is not reached through libps: `bind_store(table, versions)` hands the cache the table
    shard (device tensor) and its per-row version array, or pass `node_id` of a table registered with
    `herald_amd.cache.register_table` (the role InitTensor plays in the reference);
  * keys / dest / grads may be device tensors (no copy) or host numpy arrays (staged over PCIe);
  * calls are asynchronous on a HIP stream; the returned handle's wait() synchronises it.
All computation is in libherald_amd.so (csrc/cache.hip); this file only marshals arguments.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check

_TABLES = {}


def register_table(node_id, table, versions=None, row_start=0):
    """The in-node equivalent of InitTensor(node_id, ptype=CacheTable, ...) (python/hetu/initializers.py:28-38):
    associates a device table shard (and its row versions) with a node id."""
    if versions is None:
        versions = torch.zeros(table.shape[0], dtype=torch.int64, device=table.device)
        if versions.is_cuda:
            torch.cuda.current_stream(versions.device).synchronize()   # (the cache's calls come on streams of the caller's choice)
    _TABLES[int(node_id)] = (table, versions, int(row_start))
    return versions

_PLAN_STREAMS = {}      # (device index, priority, row stream) -> the planning stream the caches of this process share


class _LookupMark:
    """What embedding_lookup remembers of its key tensor so that an update can prove it was handed the same one:
    a STRONG reference (the caching allocator cannot hand the block to another tensor meanwhile), the address /
    length, and the autograd version counter (views share it: an in-place write through torch bumps it).  Raw-pointer
    writers are invisible to it -- which is why same_as_lookup stays an explicit statement of the caller."""
    __slots__ = ("tensor", "ptr", "numel", "version")

    def __init__(self, k):
        self.tensor, self.ptr, self.numel, self.version = k, k.data_ptr(), k.numel(), k._version


def _is_marked(mark, k):
    return (mark is not None and torch.is_tensor(k) and mark.ptr == k.data_ptr() and mark.numel == k.numel() and
            mark.tensor.dtype == k.dtype and mark.tensor.untyped_storage().data_ptr() == k.untyped_storage().data_ptr()
            and mark.version == k._version and mark.tensor._version == mark.version)


class Wait:
    """The `_waittype` of the reference (a shared_future): wait() blocks until the call is done."""

    def __init__(self, stream, keep=()):
        self._event = torch.cuda.Event()
        self._event.record(stream)
        self._keep = keep
        self._after = None

    def wait(self):
        self._event.synchronize()
        if self._after is not None:
            self._after()
            self._after = None
        self._keep = ()


class Embedding:
    """The `Embedding` view of one cache line (python_api.cc:60-75)."""

    def __init__(self, key, version, data, grad=None, updates=0):
        self.key, self.version, self.updates = int(key), int(version), int(updates)
        self.data, self.grad = data, grad

    def mean(self):
        return float(np.mean(self.data.astype(np.float64)))

    def var(self):
        return float(np.var(self.data.astype(np.float64)))

    def __repr__(self):
        return "<hetu.Embedding : key:%d, len:%d, version:%d, mean:%g, var:%g>" % (
            self.key, self.data.size, self.version, self.mean(), self.var())


class _CacheBase:
    POLICY = None
    NAME = None

    def __init__(self, limit, length, width, node_id=0, max_batch=None, device=None, stream=None):
        self._L = _lib.load()
        self.device = torch.device(device if device is not None else "cuda")
        self._limit, self._length, self._width, self.node_id = int(limit), int(length), int(width), int(node_id)
        self._max_batch = int(max_batch) if max_batch else 1 << 17
        with torch.cuda.device(self.device):
            self._h = self._L.ha_cache_create(self.POLICY, self._limit, self._length, self._width,
                                              self._max_batch)
        if not self._h:
            raise _lib.HeraldAmdError("ha_cache_create failed: %s" % self._L.ha_last_error().decode())
        self._pull_bound = self._push_bound = 5
        self._perf_enabled = False
        self.perf_times = False
        self._perf = []
        self._last_lookup = None
        self._ahead = None
        self._ahead_ring = []
        self._store = None
        self._remote = None
        self.stream = stream
        if self.node_id in _TABLES:
            self.bind_store(*_TABLES[self.node_id])

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.ha_cache_destroy(h)

    # ---- store ---------------------------------------------------------------------------------------
    def bind_store(self, table, versions, row_start=0):
        """The store ("server") the cache fronts: rows [row_start, row_start + table.shape[0]) of the global table and their
        versions (device int64).  `table`: device memory, or PINNED host memory (device-visible: the cache's kernels then reach
        the store's rows over PCIe -- the cold tier of BASELINE configs[4] with the store addressed directly)."""
        assert (table.is_cuda or table.is_pinned()) and table.dtype == torch.float32 and table.is_contiguous()
        assert versions.is_cuda and versions.dtype == torch.int64 and versions.numel() == table.shape[0]
        assert table.shape[1] == self._width
        self._store = (table, versions)
        check(self._L.ha_cache_bind_store(self._h, ctypes.c_void_p(table.data_ptr()),
                                          ctypes.c_void_p(versions.data_ptr()), table.shape[0], int(row_start)),
              "ha_cache_bind_store")

    # ---- remote store (rows owned by other ranks / kept in host memory) -----------------------------------
    def bind_remote(self, store):
        """Put the cache in front of a store it cannot address: `store` implements
            sync(keys, versions, bound, pull, idx, ver_out, rows_out)   the owner's kSyncEmbedding
            push(keys, updates, rows)                                     the owner's kPushEmbedding
        on device tensors (herald_amd.remote_store: LocalStore, ShardedStore, HostStore).  The cache kernels
        then fill a request / an outbox and read an inbox (include/herald_amd.h, ha_cache_remote)."""
        check(self._L.ha_cache_set_remote(self._h), "ha_cache_set_remote")
        r = _lib.CacheRemote()
        check(self._L.ha_cache_remote_buffers(self._h, ctypes.byref(r)), "ha_cache_remote_buffers")
        nb, w, cap = int(r.max_batch), self._width, int(r.out_capacity)
        dv = self.device
        self._rb = {
            "req_keys": _dev_view(r.req_keys, (nb,), torch.int32, dv),
            "req_ver": _dev_view(r.req_versions, (nb,), torch.int64, dv),
            "pull": _dev_view(r.inbox_pull, (nb,), torch.int32, dv),
            "idx": _dev_view(r.inbox_idx, (nb,), torch.int32, dv),
            "ver": _dev_view(r.inbox_versions, (nb,), torch.int64, dv),
            "rows": _dev_view(r.inbox_rows, (nb, w), torch.float32, dv),
            "out_keys": _dev_view(r.out_keys, (cap,), torch.int32, dv),
            "out_upd": _dev_view(r.out_updates, (cap,), torch.int32, dv),
            "out_rows": _dev_view(r.out_rows, (cap, w), torch.float32, dv),
        }
        self._remote = store
        self._out_cap = cap
        self._evict_bound = 0     # keys of the lookups since the last update: bounds their pending evictions

    def _sync_request(self, u):
        b = self._rb
        self._remote.sync(b["req_keys"][:u], b["req_ver"][:u], self._pull_bound, b["pull"][:u], b["idx"][:u],
                          b["ver"][:u], b["rows"])

    def _host_counts(self):
        """Does the store size its work on the host (an exchange between ranks)?  Otherwise the request and the
        outbox are handed over padded and no call of the step reads anything back (include/herald_amd.h)."""
        return getattr(self._remote, "host_counts", True)

    def _outbox_bound(self, n):
        """Entries that bound U + E of an update of n keys: its own unique keys plus the evictions of every
        lookup since the last update (each at most its number of keys); None: beyond the outbox, count on the host."""
        m = n + self._evict_bound
        return m if m <= self._out_cap else None

    def _push_outbox(self, s, bound=None, distinct=False):
        b = self._rb
        if bound is None:
            cnt = ctypes.c_int64(0)
            check(self._L.ha_cache_outbox_count(self._h, ctypes.byref(cnt), ctypes.c_void_p(s.cuda_stream)),
                  "ha_cache_outbox_count")
            bound = int(cnt.value)
        if distinct:
            self._remote.push(b["out_keys"][:bound], b["out_upd"][:bound], b["out_rows"][:bound], distinct=True)
        else:
            self._remote.push(b["out_keys"][:bound], b["out_upd"][:bound], b["out_rows"][:bound])

    def _lookup_remote(self, keys, dest):
        keep = []
        s = self._stream()
        with torch.cuda.stream(s):
            k, kind = self._keys(keys, keep)
            host_dest = None
            if isinstance(dest, np.ndarray):
                host_dest = dest
                dest = torch.empty((k.numel(), self._width), dtype=torch.float32, device=self.device)
            assert dest.numel() == k.numel() * self._width and dest.dtype == torch.float32
            if self._host_counts():
                u = ctypes.c_int64(0)
                check(self._L.ha_cache_lookup_begin(self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(),
                                                    ctypes.byref(u), ctypes.c_void_p(s.cuda_stream)),
                      "ha_cache_lookup_begin")
                self._sync_request(int(u.value))
            else:       # the request is padded to the batch size: nothing to read back
                check(self._L.ha_cache_lookup_begin(self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(), None,
                                                    ctypes.c_void_p(s.cuda_stream)), "ha_cache_lookup_begin")
                self._sync_request(k.numel())
            self._evict_bound += k.numel()
            check(self._L.ha_cache_lookup_finish(self._h, k.numel(), ctypes.c_void_p(dest.data_ptr()),
                                                 ctypes.c_void_p(s.cuda_stream)), "ha_cache_lookup_finish")
            # the plan of ha_cache_lookup_begin stays in the workspace: an update of the same key tensor can reuse it
            self._last_lookup = None if keep else _LookupMark(k)
            if self.perf_enabled:
                self._perf_record(0)
            w = Wait(s, keep + [k, dest])
            if host_dest is not None:
                w._after = (lambda hd=host_dest, dd=dest: np.copyto(hd.reshape(dd.shape), dd.cpu().numpy()))
        return w

    def _update_remote(self, keys, grads, push_keys, same_as_lookup=False):
        keep = []
        s = self._stream()
        with torch.cuda.stream(s):
            k, kind = self._keys(keys, keep)
            g = self._grads(grads, keep)
            assert g.numel() == k.numel() * self._width
            same = bool(same_as_lookup) and k.numel() > 0 and _is_marked(self._last_lookup, k)
            if same_as_lookup and not same:
                raise ValueError("same_as_lookup=True, but no embedding_lookup of this key tensor precedes")
            self._last_lookup = None
            bound = None if self._host_counts() else self._outbox_bound(k.numel())
            check(self._L.ha_cache_outbox_pad(self._h, bound or 0), "ha_cache_outbox_pad")
            fused0 = int(self._L.ha_cache_fused_updates(self._h))
            if push_keys is None and same:       # the lookup's index plan is still in the workspace
                check(self._L.ha_cache_update_same_keys(self._h, k.numel(), ctypes.c_void_p(g.data_ptr()),
                                                        ctypes.c_void_p(s.cuda_stream)), "ha_cache_update_same_keys")
            elif push_keys is None:
                check(self._L.ha_cache_update(self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(),
                                              ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(s.cuda_stream)),
                      "ha_cache_update")
            else:
                pk, pkind = self._keys(push_keys, keep)
                check(self._L.ha_cache_update_with_push_keys(
                    self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(), ctypes.c_void_p(pk.data_ptr()), pkind,
                    pk.numel(), ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(s.cuda_stream)),
                    "ha_cache_update_with_push_keys")
            # an update that took the two-launch path leaves pairwise distinct keys in the outbox (the batch's unique keys
            # and the victims of one lookup): the store applies them without sorting
            self._push_outbox(s, bound, distinct=int(self._L.ha_cache_fused_updates(self._h)) > fused0)
            self._evict_bound = 0
            if self.perf_enabled:
                self._perf_record(1)
            return Wait(s, keep + [k, g])

    def _push_pull_remote(self, pullkeys, dest, pushkeys, grads):
        keep = []
        self._last_lookup = None
        s = self._stream()
        with torch.cuda.stream(s):
            pk, pkind = self._keys(pullkeys, keep)
            sk, skind = self._keys(pushkeys, keep)
            g = self._grads(grads, keep)
            assert dest.is_cuda and dest.dtype == torch.float32 and dest.numel() == pk.numel() * self._width
            hc = self._host_counts()
            bound = None if hc else self._outbox_bound(sk.numel())
            check(self._L.ha_cache_outbox_pad(self._h, bound or 0), "ha_cache_outbox_pad")
            u = ctypes.c_int64(0)
            check(self._L.ha_cache_push_pull_begin(self._h, ctypes.c_void_p(pk.data_ptr()), pkind, pk.numel(),
                                                   ctypes.c_void_p(sk.data_ptr()), skind, sk.numel(),
                                                   ctypes.c_void_p(g.data_ptr()), ctypes.byref(u) if hc else None,
                                                   ctypes.c_void_p(s.cuda_stream)), "ha_cache_push_pull_begin")
            self._push_outbox(s, bound)          # the server pushes before it syncs
            self._sync_request(int(u.value) if hc else pk.numel())
            self._evict_bound = pk.numel()       # the pull phase's evictions stay pending until the next update
            check(self._L.ha_cache_push_pull_finish(self._h, ctypes.c_void_p(dest.data_ptr()),
                                                    ctypes.c_void_p(s.cuda_stream)), "ha_cache_push_pull_finish")
            return Wait(s, keep + [pk, sk, g, dest])

    # ---- properties of the reference ----------------------------------------------------------------------
    @property
    def limit(self):
        return self._limit

    @property
    def width(self):
        return self._width

    @property
    def pull_bound(self):
        return self._pull_bound

    @pull_bound.setter
    def pull_bound(self, v):
        self._pull_bound = int(v)
        check(self._L.ha_cache_set_bounds(self._h, self._pull_bound, self._push_bound), "set_bounds")

    @property
    def push_bound(self):
        return self._push_bound

    @push_bound.setter
    def push_bound(self, v):
        self._push_bound = int(v)
        check(self._L.ha_cache_set_bounds(self._h, self._pull_bound, self._push_bound), "set_bounds")

    @property
    def perf(self):
        return self._perf

    def bypass(self):
        check(self._L.ha_cache_set_bypass(self._h, 1), "bypass")

    def undo_bypass(self):
        check(self._L.ha_cache_set_bypass(self._h, 0), "undo_bypass")

    # ---- argument marshalling ----------------------------------------------------------------------------------
    def _stream(self):
        return self.stream if self.stream is not None else torch.cuda.current_stream(self.device)

    def _keys(self, keys, keep):
        """-> (device tensor, key_kind).  float32 -> kind 0 (raw entry points), (u)int64 -> kind 1."""
        if isinstance(keys, np.ndarray):
            if not keys.flags.c_contiguous:
                raise RuntimeError("Numpy Array is not contiguous")          # binding.h:51-57
            if keys.dtype == np.uint64:
                keys = keys.view(np.int64)
            t = torch.from_numpy(keys).to(self.device, non_blocking=False)
            keep.append(t)
            keys = t
        if keys.dtype == torch.float32:
            kind = 0
        elif keys.dtype in (torch.int64, torch.uint64):
            kind = 1
        else:
            raise TypeError("keys must be float32 or (u)int64, got %s" % keys.dtype)
        if not keys.is_contiguous():
            raise RuntimeError("keys are not contiguous")
        return keys, kind

    @property
    def perf_enabled(self):
        return self._perf_enabled

    @perf_enabled.setter
    def perf_enabled(self, on):
        """The reference's perf_enabled_: one dict per call in `perf` (counts + stage times)."""
        self._perf_enabled = bool(on)
        self.perf_times = bool(on)
        check(self._L.ha_cache_set_timing(self._h, 1 if on else 0), "ha_cache_set_timing")

    def _perf_record(self, kind):
        out = (ctypes.c_int64 * 8)()
        check(self._L.ha_cache_perf(self._h, out, ctypes.c_void_p(self._stream().cuda_stream)), "ha_cache_perf")
        d = {"type": "Pull" if out[0] == 0 else "Push", "num_all": out[1], "num_unique": out[2],
             "num_miss": out[3], "num_transfered": out[4], "is_full": bool(out[6])}
        if out[0] == 1:
            d["num_evict"] = out[5]
        if self.perf_times:
            # the reference's stage times (milliseconds; cache.cc:99-105 Pull, 189-194 Push), from HIP events between the
            # call's launches: what its CPU stages are here -- sort = the index plan; lookup (+ prepare) = probe, miss scan
            # and slot assignment; copy = rows to dest with the pulls, the insert and the eviction in the same launch
            # (Push: the accumulate); transfer = the exchange with a remote store (Push: the push launches)
            ms = (ctypes.c_double * 6)()
            check(self._L.ha_cache_stage_times(self._h, ms), "ha_cache_stage_times")
            t = [max(v, 0.0) for v in ms]
            d.update({"time": t[0], "sort_time": t[1], "lookup_time": t[2], "copy_time": t[3], "transfer_time": t[4]})
            if out[0] == 0:
                d.update({"prepare_time": 0.0, "copy_time": t[3] + t[5], "insert_time": 0.0})
            else:
                d["cleanup_time"] = t[5]
        self._perf.append(d)

    # ---- batch API ---------------------------------------------------------------------------------------------------
    def prefetch_keys(self, keys):
        """State that `keys` (a device tensor) is the batch of the NEXT embedding_lookup: its sort runs now, on a stream of
        the cache's own, beside the calls of the current batch (the data loader has the ids a batch early,
        dataloader.py:63-98).  The tensor must not be written to until that lookup -- torch-visible writes are detected and
        the lookup then sorts by itself; raw-pointer writers are the caller's responsibility, as with same_as_lookup.
        Local stores only (a remote store's lookup is split around an exchange and sorts at its start)."""
        self._ahead = None
        self._drop_ahead_ring()
        if self._remote is not None or not torch.is_tensor(keys) or keys.numel() == 0:
            return
        s = self._stream()
        k, kind = self._keys(keys, [])
        check(self._L.ha_cache_sort_ahead(self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(),
                                          ctypes.c_void_p(s.cuda_stream)), "ha_cache_sort_ahead")
        self._ahead = _LookupMark(k)

    def prefetch_keys_batch(self, keys_list):
        """State that the device tensors of `keys_list` (at most 16, one dtype, 1 .. 36,864 keys each) are the batches of the
        NEXT embedding_lookup calls, in this order: their sorts run now, in ONE launch on the cache's stream (a caller that has
        its ids a block of batches early; no second stream, so the calls capture into a hipGraph without a fork).  Same
        conditions on the tensors as prefetch_keys; a lookup of anything else drops what is left of the block."""
        self._ahead = None
        self._ahead_ring = []
        if self._remote is not None or not keys_list:
            return
        s = self._stream()
        ks = [self._keys(k, []) for k in keys_list]
        kinds = {kind for _, kind in ks}
        if len(kinds) != 1 or len(ks) > 16 or any(k.numel() == 0 for k, _ in ks):
            raise ValueError("prefetch_keys_batch: 1-16 non-empty key tensors of one dtype")
        ptrs = (ctypes.c_void_p * len(ks))(*[k.data_ptr() for k, _ in ks])
        ns = (ctypes.c_int64 * len(ks))(*[k.numel() for k, _ in ks])
        check(self._L.ha_cache_sort_ahead_batch(self._h, ptrs, kinds.pop(), ns, len(ks), ctypes.c_void_p(s.cuda_stream)),
              "ha_cache_sort_ahead_batch")
        self._ahead_ring = [_LookupMark(k) for k, _ in ks]

    def _drop_ahead_ring(self):
        if self._ahead_ring:
            self._ahead_ring = []
            check(self._L.ha_cache_sort_ahead_batch(self._h, None, 0, None, 0, ctypes.c_void_p(self._stream().cuda_stream)),
                  "ha_cache_sort_ahead_batch")

    # ---- the planned flow: the bookkeeping of a block of batches ahead, ONE launch per lookup / update ----------------------
    def plan_block(self, keys_list, side=None):
        """State that the device tensors of `keys_list` (1..16, one dtype, at most min(max_batch, 36,864) keys each) are the
        batches of the NEXT embedding_lookup_planned / embedding_update_planned pairs, in this order (ha_cache_plan_block,
        csrc/cache_block.hip: a local store; LRU: limit >= the batch; LFU / LFUOpt: every resident line updated since its lookup, as
        after any lookup + update pair).  Their index plans and the bookkeeping of the whole
        block -- hits, misses, slots, evictions, update counters, the bounded push: all of it follows from the ids -- run NOW on
        `side` (default: a stream of the cache's own), beside whatever rows the cache's stream is still moving; every lookup and
        every update of these batches is then ONE launch.  Plan block b + 1 when block b starts (two blocks may be outstanding).
        Until the planned pairs are consumed the call-by-call methods raise; lines() / keys() / size() show the bookkeeping of
        every planned batch (meaningful at the end of a block)."""
        if self._remote is not None or not keys_list:
            raise ValueError("plan_block: a non-empty list of key tensors, local store")
        s = self._stream()
        if side is None:
            if getattr(self, "_plan_side", None) is None:
                # A stream of another PRIORITY than the row stream: HIP multiplexes the streams of one priority onto a few
                # hardware queues, and two streams that share a queue run in submission order -- the bookkeeping of block b + 1
                # then sits between the rows of block b - 1 and block b instead of beside them (measured: 183 us of idle row
                # stream per block in bench.py, where a dozen streams exist; none in a process with two).  HA_CACHE_PLAN_PRIO:
                # high (default) / low / normal.
                import os
                prio = 0
                want = os.environ.get("HA_CACHE_PLAN_PRIO", "high")
                try:
                    lo_hi = torch.cuda.Stream.priority_range()
                    prio = min(lo_hi) if want == "high" else max(lo_hi) if want == "low" else 0
                except Exception:      # noqa: BLE001
                    prio = 0
                # ONE planning stream per device for all caches of the process (HA_CACHE_PLAN_SHARED=0: one per cache): which
                # hardware queue a newly created stream lands on depends on what the process created before, and an unlucky
                # pairing with the row stream's queue stretched the ROW launches threefold (bench.py's second / third cache:
                # profiles/r06/cache_tier_third_instance.txt) -- the pair that the first cache got is kept.
                shared = os.environ.get("HA_CACHE_PLAN_SHARED", "1") == "1"
                key = (torch.device(self.device).index, prio, s.cuda_stream)
                if shared and key in _PLAN_STREAMS:
                    self._plan_side = _PLAN_STREAMS[key]
                else:
                    # (and not an unlucky partner of the row stream: streams.pick_side_stream measures a few candidates once)
                    from . import streams
                    self._plan_side = streams.pick_side_stream(s, prio) if shared else \
                        torch.cuda.Stream(device=self.device, priority=prio)
                    if shared:
                        _PLAN_STREAMS[key] = self._plan_side
            side = self._plan_side
        ks = [self._keys(k, []) for k in keys_list]
        kinds = {kind for _, kind in ks}
        if len(kinds) != 1 or len(ks) > 16:
            raise ValueError("plan_block: 1-16 key tensors of one dtype")
        ptrs = (ctypes.c_void_p * len(ks))(*[k.data_ptr() if k.numel() else None for k, _ in ks])
        ns = (ctypes.c_int64 * len(ks))(*[k.numel() for k, _ in ks])
        self._ahead = None
        self._drop_ahead_ring()
        self._last_lookup = None
        check(self._L.ha_cache_plan_block(self._h, ptrs, kinds.pop(), ns, len(ks), ctypes.c_void_p(side.cuda_stream),
                                          ctypes.c_void_p(s.cuda_stream)), "ha_cache_plan_block")
        if not hasattr(self, "_planned"):
            self._planned = []
        self._planned.extend([k, False] for k, _ in ks)      # [key tensor (kept alive), its lookup done?]

    def plan_pending(self):
        """Planned calls (lookups + updates) still to be made."""
        return int(self._L.ha_cache_plan_pending(self._h))

    def embedding_lookup_planned(self, dest):
        """The lookup of the next planned batch (cache.cc:60-107), ONE launch; dest: float32 device tensor [n, width]."""
        if not getattr(self, "_planned", None) or self._planned[0][1]:
            raise ValueError("embedding_lookup_planned: no planned batch is due for its lookup")
        k = self._planned[0][0]
        s = self._stream()
        assert dest.is_cuda and dest.dtype == torch.float32 and dest.numel() == k.numel() * self._width
        check(self._L.ha_cache_lookup_planned(self._h, k.numel(), ctypes.c_void_p(dest.data_ptr() if k.numel() else None),
                                              ctypes.c_void_p(s.cuda_stream)), "ha_cache_lookup_planned")
        self._planned[0][1] = True
        if self.perf_enabled:
            self._perf_record(0)
        return Wait(s, [k, dest]) if self._planned_waits else None

    def embedding_update_planned(self, grads):
        """The update of the planned batch whose lookup was the last planned call (cache.cc:132-197), ONE launch."""
        if not getattr(self, "_planned", None) or not self._planned[0][1]:
            raise ValueError("embedding_update_planned: the lookup of the planned batch comes first")
        k = self._planned[0][0]
        s = self._stream()
        assert grads.is_cuda and grads.dtype == torch.float32 and grads.is_contiguous() and grads.numel() == k.numel() * self._width
        check(self._L.ha_cache_update_planned(self._h, k.numel(), ctypes.c_void_p(grads.data_ptr() if k.numel() else None),
                                              ctypes.c_void_p(s.cuda_stream)), "ha_cache_update_planned")
        self._planned.pop(0)
        if self.perf_enabled:
            self._perf_record(1)
        return Wait(s, [k, grads]) if self._planned_waits else None

    def run_planned_pairs(self, dests, grads):
        """The next len(dests) planned pairs by ONE library call (ha_cache_run_planned_pairs): lookup into dests[k], update with
        grads[k].  For callers that have the pairs' gradient buffers at hand (a benchmark loop); no perf records."""
        cnt = len(dests)
        pl = getattr(self, "_planned", None) or []
        if cnt > len(pl) or len(grads) != cnt or (pl and pl[0][1]):
            raise ValueError("run_planned_pairs: %d pairs, %d planned" % (cnt, len(pl)))
        s = self._stream()
        ns = (ctypes.c_int64 * cnt)(*[pl[k][0].numel() for k in range(cnt)])
        dp = (ctypes.c_void_p * cnt)(*[d.data_ptr() for d in dests])
        gp = (ctypes.c_void_p * cnt)(*[g.data_ptr() for g in grads])
        check(self._L.ha_cache_run_planned_pairs(self._h, cnt, ns, dp, gp, ctypes.c_void_p(s.cuda_stream)),
              "ha_cache_run_planned_pairs")
        del pl[:cnt]

    _planned_waits = True      # False: the planned calls return None instead of a wait handle (no event per call: bench loops)

    def embedding_lookup(self, keys, dest):
        """dest[i,:] = line(keys[i]).data after the staleness-bounded pull (cache.cc:60-107)."""
        if self._remote is not None:
            return self._lookup_remote(keys, dest)
        keep = []
        s = self._stream()
        with torch.cuda.stream(s):
            k, kind = self._keys(keys, keep)
            lookup = self._L.ha_cache_lookup
            if self._ahead_ring:
                if _is_marked(self._ahead_ring[0], k):
                    lookup = self._L.ha_cache_lookup_presorted
                    self._ahead_ring.pop(0)
                else:           # not the batch that was announced: the rest of the block sorts by itself
                    self._drop_ahead_ring()
            elif self._ahead is not None and _is_marked(self._ahead, k):
                lookup, self._ahead = self._L.ha_cache_lookup_presorted, None
            host_dest = None
            if isinstance(dest, np.ndarray):
                if not dest.flags.c_contiguous:
                    raise RuntimeError("Numpy Array is not contiguous")
                host_dest = dest
                dest = torch.empty((k.numel(), self._width), dtype=torch.float32, device=self.device)
            assert dest.numel() == k.numel() * self._width and dest.dtype == torch.float32
            check(lookup(self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(),
                         ctypes.c_void_p(dest.data_ptr()), ctypes.c_void_p(s.cuda_stream)), "ha_cache_lookup")
            # an update of the very same (unmodified) device key tensor can reuse this call's index plan
            self._last_lookup = None if keep else _LookupMark(k)
            if self.perf_enabled:
                self._perf_record(0)
            w = Wait(s, keep + [k, dest])
            if host_dest is not None:
                w._after = (lambda hd=host_dest, dd=dest: np.copyto(hd.reshape(dd.shape), dd.cpu().numpy()))
        return w

    def _grads(self, grads, keep):
        if isinstance(grads, np.ndarray):
            if not grads.flags.c_contiguous:
                raise RuntimeError("Numpy Array is not contiguous")
            grads = torch.from_numpy(grads).to(self.device)
            keep.append(grads)
        assert grads.dtype == torch.float32 and grads.is_contiguous()
        return grads

    def embedding_update(self, keys, grads, same_as_lookup=False):
        """Line::accumulate per occurrence + bounded push (cache.cc:132-197).  same_as_lookup=True: the
        caller states that `keys` is the very batch of the embedding_lookup that precedes this call (same
        tensor, contents unchanged); the update then reuses that call's index plan instead of sorting the
        keys again.  It is explicit because a tensor's identity says nothing about its contents when
        raw-pointer writers (HIP kernels, DLPack aliases) fill it."""
        if self._remote is not None:
            return self._update_remote(keys, grads, None, same_as_lookup)
        keep = []
        s = self._stream()
        with torch.cuda.stream(s):
            k, kind = self._keys(keys, keep)
            g = self._grads(grads, keep)
            assert g.numel() == k.numel() * self._width
            same = bool(same_as_lookup) and k.numel() > 0 and _is_marked(self._last_lookup, k)
            if same_as_lookup and not same:
                raise ValueError("same_as_lookup=True, but no embedding_lookup of this key tensor precedes")
            self._last_lookup = None
            if same:
                check(self._L.ha_cache_update_same_keys(self._h, k.numel(), ctypes.c_void_p(g.data_ptr()),
                                                        ctypes.c_void_p(s.cuda_stream)),
                      "ha_cache_update_same_keys")
            else:
                check(self._L.ha_cache_update(self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(),
                                              ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(s.cuda_stream)),
                      "ha_cache_update")
            if self.perf_enabled:
                self._perf_record(1)
            return Wait(s, keep + [k, g])

    def embedding_update_with_push_keys(self, keys, push_keys, grads):
        """As embedding_update, the push set being the lines listed in the (sorted) push keys (cache.cc:248-335)."""
        if self._remote is not None:
            return self._update_remote(keys, grads, push_keys)
        self._last_lookup = None
        keep = []
        s = self._stream()
        with torch.cuda.stream(s):
            k, kind = self._keys(keys, keep)
            pk, pkind = self._keys(push_keys, keep)
            g = self._grads(grads, keep)
            check(self._L.ha_cache_update_with_push_keys(
                self._h, ctypes.c_void_p(k.data_ptr()), kind, k.numel(), ctypes.c_void_p(pk.data_ptr()), pkind,
                pk.numel(), ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(s.cuda_stream)),
                "ha_cache_update_with_push_keys")
            if self.perf_enabled:
                self._perf_record(1)
            return Wait(s, keep + [k, pk, g])

    def embedding_push_pull(self, pullkeys, dest, pushkeys, grads):
        """Push the gradients of pushkeys, then pull pullkeys into dest (cache.cc:356-422)."""
        if self._remote is not None:
            return self._push_pull_remote(pullkeys, dest, pushkeys, grads)
        self._last_lookup = None
        keep = []
        s = self._stream()
        with torch.cuda.stream(s):
            pk, pkind = self._keys(pullkeys, keep)
            sk, skind = self._keys(pushkeys, keep)
            g = self._grads(grads, keep)
            assert dest.is_cuda and dest.dtype == torch.float32 and dest.numel() == pk.numel() * self._width
            check(self._L.ha_cache_push_pull(self._h, ctypes.c_void_p(pk.data_ptr()), pkind, pk.numel(),
                                             ctypes.c_void_p(dest.data_ptr()), ctypes.c_void_p(sk.data_ptr()), skind,
                                             sk.numel(), ctypes.c_void_p(g.data_ptr()),
                                             ctypes.c_void_p(s.cuda_stream)), "ha_cache_push_pull")
            return Wait(s, keep + [pk, sk, g, dest])

    embedding_push_pull_raw = embedding_push_pull

    # the *_raw spellings of the reference take addresses; here they take the tensors themselves
    embedding_lookup_raw = embedding_lookup
    embedding_update_raw = embedding_update
    embedding_update_with_push_keys_raw = embedding_update_with_push_keys
    embedding_update_with_push_keys_np_raw = embedding_update_with_push_keys

    # ---- inspection (debug API of the reference: count / size / keys / lookup) ----------------------------------------------
    def _snapshot(self):
        cap = self._limit + 8
        dev = self.device
        keys = torch.empty(cap, dtype=torch.int32, device=dev)
        ver = torch.empty(cap, dtype=torch.int64, device=dev)
        upd = torch.empty(cap, dtype=torch.int32, device=dev)
        stamp = torch.empty(cap, dtype=torch.int64, device=dev)
        slots = torch.empty(cap, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        s = self._stream()
        with torch.cuda.stream(s):
            check(self._L.ha_cache_snapshot(self._h, cap, *[ctypes.c_void_p(t.data_ptr()) for t in
                                                            (keys, ver, upd, stamp, slots, cnt)],
                                            ctypes.c_void_p(s.cuda_stream)), "ha_cache_snapshot")
            s.synchronize()
        n = int(cnt.item())
        k = (keys[:n].cpu().numpy().astype(np.int64) & 0xFFFFFFFF)
        order = np.argsort(k, kind="stable")
        return {"keys": k[order], "version": ver[:n].cpu().numpy()[order], "updates": upd[:n].cpu().numpy()[order],
                "stamp": stamp[:n].cpu().numpy()[order], "slots": slots[:n].cpu().numpy()[order]}

    def _rows(self, which, slots):
        """Rows of the cache's data / grad arrays (device -> host copies; inspection only)."""
        fn = self._L.ha_cache_data if which == "data" else self._L.ha_cache_grad
        base = fn(self._h)
        out = np.empty((len(slots), self._width), dtype=np.float32)
        torch.cuda.synchronize(self.device)
        hip = _hip()
        for i, sl in enumerate(slots):
            rc = hip.hipMemcpy(ctypes.c_void_p(out[i].ctypes.data), ctypes.c_void_p(base + int(sl) * self._width * 4),
                               ctypes.c_size_t(self._width * 4), ctypes.c_int(2))   # hipMemcpyDeviceToHost
            if rc != 0:
                raise _lib.HeraldAmdError("hipMemcpy failed: %d" % rc)
        return out

    def size(self):
        out = (ctypes.c_int64 * 8)()
        check(self._L.ha_cache_state(self._h, out, ctypes.c_void_p(self._stream().cuda_stream)), "ha_cache_state")
        return int(out[0])

    def state(self):
        out = (ctypes.c_int64 * 8)()
        check(self._L.ha_cache_state(self._h, out, ctypes.c_void_p(self._stream().cuda_stream)), "ha_cache_state")
        return dict(zip(("size", "pending_evictions", "free_slots", "log_head", "log_tail", "clock", "slots",
                         "log_cap"), [int(x) for x in out]))

    def keys(self):
        return self._snapshot()["keys"].astype(np.uint64)

    def count(self, k):
        return int(int(k) in set(self._snapshot()["keys"].tolist()))

    def lookup(self, k):
        snap = self._snapshot()
        idx = np.nonzero(snap["keys"] == int(k))[0]
        if idx.size == 0:
            return None
        i = int(idx[0])
        data = self._rows("data", [snap["slots"][i]])[0]
        grad = self._rows("grad", [snap["slots"][i]])[0]
        return Embedding(k, snap["version"][i], data, grad, snap["updates"][i])

    def insert(self, key, embedding=None):
        """insert(EmbeddingPT) of the policies (lru_cache.cc:9-25, lfu_cache.cc:20-45, lfuopt_cache.cc:13-42): the line
        enters the cache like a lookup miss would bring it in (same touch / evict bookkeeping, a dirty victim joins the
        pending evictions), then takes the given version and data.  `insert(Embedding)` or `insert(key, row)`."""
        if isinstance(key, Embedding):
            e = key
        else:
            row = np.asarray(embedding, dtype=np.float32).reshape(-1)
            e = Embedding(int(key), -1, row)
        if e.data.size != self._width:
            raise ValueError("embedding width %d, cache width %d" % (e.data.size, self._width))
        dest = torch.empty((1, self._width), dtype=torch.float32, device=self.device)
        k = torch.tensor([int(e.key)], dtype=torch.int64, device=self.device)
        self.embedding_lookup(k, dest).wait()
        row = torch.from_numpy(np.ascontiguousarray(e.data, dtype=np.float32)).to(self.device)
        s = self._stream()
        check(self._L.ha_cache_set_line(self._h, int(e.key), int(e.version), ctypes.c_void_p(row.data_ptr()),
                                        ctypes.c_void_p(s.cuda_stream)), "ha_cache_set_line")
        Wait(s, keep=(row,)).wait()

    def lines(self):
        """All resident lines as {key: Embedding} (test helper)."""
        snap = self._snapshot()
        data = self._rows("data", snap["slots"])
        grad = self._rows("grad", snap["slots"])
        return {int(k): Embedding(k, snap["version"][i], data[i], grad[i], snap["updates"][i])
                for i, k in enumerate(snap["keys"])}

    def __repr__(self):
        return "<Cache : %d/%d , id:%d , width:%d , bound:%d %d>" % (
            self.size(), self._limit, self.node_id, self._width, self._pull_bound, self._push_bound)


class _DevPtr:
    """__cuda_array_interface__ holder: lets torch wrap a device buffer the library owns (no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _dev_view(ptr, shape, dtype, device):
    typestr = {torch.int32: "<i4", torch.int64: "<i8", torch.float32: "<f4"}[dtype]
    if 0 in shape:
        return torch.empty(shape, dtype=dtype, device=device)
    return torch.as_tensor(_DevPtr(ptr, shape, typestr), device=device)


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        import os
        _HIP = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        _HIP.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _HIP.hipMemcpy.restype = ctypes.c_int
    return _HIP


class LRUCache(_CacheBase):
    POLICY, NAME = 0, "LRU"


class LFUCache(_CacheBase):
    POLICY, NAME = 1, "LFU"


class LFUOptCache(_CacheBase):
    POLICY, NAME = 2, "LFUOpt"


class CacheSparseTable:
    """Mirror of python/hetu/cstable.py:20-150."""

    def __init__(self, limit, length, width, node_id, policy="LRU", bound=100, max_batch=None, device=None):
        policy = policy.lower()
        cls = {"lru": LRUCache, "lfu": LFUCache, "lfuopt": LFUOptCache}.get(policy)
        if cls is None:
            raise NotImplementedError(policy)
        self.cache = cls(limit, length, width, node_id, max_batch=max_batch, device=device)
        self.cache.pull_bound = bound
        self.cache.push_bound = bound

    @classmethod
    def wrap(cls, cache):
        """CacheSparseTable around an existing cache object (e.g. one bound to a remote store)."""
        t = cls.__new__(cls)
        t.cache = cache
        return t

    def _finish(self, wait, sync):
        if sync:
            wait.wait()
        return wait

    def embedding_lookup(self, keys, dest, sync=False):
        if isinstance(keys, tuple):
            keys = keys[0]
        return self._finish(self.cache.embedding_lookup(keys, dest), sync)

    def prefetch_keys(self, keys):
        """The next embedding_lookup's key batch, handed over a batch early (its sort overlaps the current batch)."""
        self.cache.prefetch_keys(keys)

    def prefetch_keys_batch(self, keys_list):
        """The key batches of the next (up to 16) embedding_lookup calls, in order: sorted in one launch now."""
        self.cache.prefetch_keys_batch(keys_list)

    def embedding_update(self, keys, grads, sync=False, same_as_lookup=False):
        return self._finish(self.cache.embedding_update(keys, grads, same_as_lookup=same_as_lookup), sync)

    # the planned flow (csrc/cache_block.hip): the ids of a block of batches a block early -- bookkeeping ahead on a side stream,
    # ONE launch per lookup and per update
    def plan_block(self, keys_list, side=None):
        self.cache.plan_block([k[0] if isinstance(k, tuple) else k for k in keys_list], side)

    def embedding_lookup_planned(self, dest, sync=False):
        w = self.cache.embedding_lookup_planned(dest)
        return self._finish(w, sync) if w is not None else None

    def embedding_update_planned(self, grads, sync=False):
        w = self.cache.embedding_update_planned(grads)
        return self._finish(w, sync) if w is not None else None

    def run_planned_pairs(self, dests, grads):
        self.cache.run_planned_pairs(dests, grads)

    def looked_up_last(self, keys):
        """True when `keys` is the device tensor the cache's last operation, an embedding_lookup, was given (same storage,
        same length).  A caller that also knows the CONTENTS are unchanged since -- an executor between the lookup of a
        batch and the push of its gradients -- may then say embedding_update(..., same_as_lookup=True)."""
        return (torch.is_tensor(keys) and keys.is_cuda and keys.numel() > 0 and
                _is_marked(self.cache._last_lookup, keys))

    def embedding_update_with_push_keys(self, keys, push_keys, grads, sync=False):
        return self._finish(self.cache.embedding_update_with_push_keys(keys, push_keys, grads), sync)

    def embedding_push_pull(self, pullkeys, dest, pushkeys, grads, sync=False):
        return self._finish(self.cache.embedding_push_pull(pullkeys, dest, pushkeys, grads), sync)

    @property
    def width(self):
        return self.cache.width

    @property
    def limit(self):
        return self.cache.limit

    def perf_enabled(self, enable=True):
        self.cache.perf_enabled = enable

    @property
    def perf(self):
        """One dict per call while the counters are on: type ("Pull" / "Push"), is_full, num_all, num_unique,
        num_miss, num_evict, num_transfered, time (cstable.py:160-168, filled by cache.cc:89-106,179-196)."""
        return self.cache.perf

    # ---- the rest of the reference wrapper's surface (python/hetu/cstable.py:170-248) ----
    def bypass(self):
        """Every key misses from now on: lookups and updates go straight to the store (cache.h bypass_)."""
        self.cache.bypass()

    def undobypass(self):
        self.cache.undo_bypass()

    def __repr__(self):
        return repr(self.cache)

    # single-key debugging calls
    def lookup(self, key):
        """The resident line of `key` as an Embedding (key, version, data, grad), or None."""
        return self.cache.lookup(key)

    def count(self, key):
        return self.cache.count(key)

    def insert(self, key, embedding=None):
        """Enter one line (the policies' insert(EmbeddingPT), lru_cache.cc:9-25): `insert(Embedding)` as the
        plugin takes it, or `insert(key, row)`."""
        return self.cache.insert(key, embedding)

    def keys(self):
        return self.cache.keys()

    def get_perf(self):
        return self.perf

    def _perf_rows(self, include_cold_start):
        return [r for r in self.perf if include_cold_start or r["is_full"]]

    def overall_miss_rate(self, include_cold_start=False):
        """Unique-key miss rate over the recorded lookups; -1 without records (cstable.py:202-212)."""
        rows = self._perf_rows(include_cold_start)
        if not rows:
            return -1
        pulls = [r for r in rows if r["type"] == "Pull"]
        return float(np.sum([r["num_miss"] for r in pulls]) / np.sum([r["num_unique"] for r in pulls]))

    def overall_data_rate(self, include_cold_start=False):
        """Rows that crossed to / from the store per key handed in, lookups and updates alike: the traffic relative
        to a cache-less sparse pull / push (cstable.py:214-224)."""
        rows = self._perf_rows(include_cold_start)
        if not rows:
            return -1
        return float(np.sum([r["num_transfered"] for r in rows]) / np.sum([r["num_all"] for r in rows]))

    def debug_keys(self, comm=None):
        """Overlap of the resident key sets between workers: rt[i][j] = |keys_i & keys_j| / |keys_i| on rank 0
        (cstable.py:226-248 exchanges them through a file and BarrierWorker; here through `comm`, a
        torch.distributed process group or the default one; single process: a 1 x 1 matrix)."""
        mine = sorted(int(k) for k in self.keys())
        import torch.distributed as dist
        if comm is None and not (dist.is_available() and dist.is_initialized()):
            sets, rank = [set(mine)], 0
        else:
            world = dist.get_world_size(group=comm)
            gathered = [None] * world
            dist.all_gather_object(gathered, mine, group=comm)
            sets, rank = [set(g) for g in gathered], dist.get_rank(group=comm)
        if rank != 0:
            return None
        rt = np.zeros((len(sets), len(sets)))
        for i, a in enumerate(sets):
            for j, b in enumerate(sets):
                if a:
                    rt[i][j] = len(a & b) / len(a)
        return rt
