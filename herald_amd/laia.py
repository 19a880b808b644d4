"""Host-side mirror of the `laia_cache` plugin and of its Python glue.

Reference surfaces mirrored (paths relative to /root/reference):
  laia_cache.LaiaScheduler().start(samples, num_sample, num_table, epoch_num, mini_batch_size,
        batch_num, nrank, rank, cache_size, num_threads, top_k_table) / .pop() / .length()
        laia/src/python_binding.cc:8-23, laia/src/laia_scheduler.cc:31-113.  The stream protocol is the
        reference's: alternating [plan, dist] lists, terminated by [0]; pop() blocks.
  laia_cache.TopkScheduler().start(samples, num_sample, num_table, epoch_num, mini_batch_size,
        batch_num, nrank, rank, cache_size, num_threads, dataset, top_k_table, local_shared,
        local_rank, local_size) / .pop() / .pop_from_local_worker() / .length()
        laia/src/python_binding.cc:16-22, laia/src/topk_scheduler.cc:47-354.  With local_shared the
        scheduler runs on local rank 0 only and every local worker i receives its stream through the
        shared-memory ring "laia_cache_<i>" (laia/include/share_mem.h:40-193).
  LAIAScheduler(sparse_data, batch_size, ...)   python/hetu/laia/laia_dataloader.py:29-169
        (5-deep queue, the first plan is discarded so that dist(b) is paired with plan(b+1)).

The per-batch work is ha_laia_next in libherald_amd.so (csrc/laia.hip): probing and plan extraction on
the GPU, assignment and snapshot bookkeeping on the scheduler's own host thread -- like the
reference, the scheduler runs ahead of training in a background thread.
"""
import ctypes
import os
import queue
import threading
import time

import numpy as np

from . import _lib


class LaiaScheduler:
    def __init__(self):
        self._L = _lib.load()
        self._h = None
        self._q = queue.Queue()
        self._thread = None
        self._close = False
        self._error = None

    def start(self, samples, num_sample, num_table, epoch_num, mini_batch_size, batch_num, nrank, rank,
              cache_size, num_threads=16, top_k_table=24, key_limit=None, device=None, ahead=None):
        """ahead (default: the environment's HA_LAIA_AHEAD == "1"): the scheduler thread announces every next batch to the
        library (ha_laia_hint_next), which then works on batch k+1 while this thread queues plan and dist of batch k."""
        samples = np.ascontiguousarray(np.asarray(samples).astype(np.uint64))   # pybind force-cast
        if samples.ndim != 2:
            raise RuntimeError("Input should be 2D numpy array")
        assert samples.shape == (num_sample, num_table)
        if key_limit is None:
            key_limit = int(samples.max()) + 1
        import os
        self._ahead = (os.environ.get("HA_LAIA_AHEAD") == "1") if ahead is None else bool(ahead)
        self._wall = (0.0, 0)
        import torch
        if device is not None:
            torch.cuda.set_device(device)
        self._cfg = (int(epoch_num), int(mini_batch_size), int(batch_num), int(nrank), int(rank))
        self._h = self._L.ha_laia_create(samples.ctypes.data, int(num_sample), int(num_table), int(nrank),
                                         int(cache_size), int(key_limit), int(mini_batch_size) * int(nrank))
        if not self._h:
            raise _lib.HeraldAmdError("ha_laia_create failed: %s" % self._L.ha_last_error().decode())
        self._num_table = int(num_table)
        self._device = torch.cuda.current_device()
        self._thread = threading.Thread(target=self._launch, daemon=True)
        self._thread.start()

    def _launch(self):
        """LaiaScheduler::launch (laia_scheduler.cc:115-169)."""
        import torch
        torch.cuda.set_device(self._device)
        epoch_num, mini_bs, batch_num, W, rank = self._cfg
        dist = np.empty(W * mini_bs, dtype=np.int64)
        cap = W * mini_bs * self._num_table * max(W - 1, 1) + 16
        plan = np.empty(cap, dtype=np.uint64)
        off = np.empty(W + 1, dtype=np.int64)
        epoch_id = 0
        import time
        t0, done = time.perf_counter(), 0
        try:
            while epoch_id < epoch_num and not self._close:
                batch_id = 0
                epoch_id += 1
                if epoch_id == epoch_num:
                    batch_num += 1          # one more allocation for the cache prefetch (:126-128)
                while batch_id < batch_num and not self._close:
                    if getattr(self, "_ahead", False):
                        # the batch of the NEXT call: the following one of this epoch, the first of the next epoch, or none
                        nxt = batch_id + 1 if batch_id + 1 < batch_num else (0 if epoch_id < epoch_num else -1)
                        self._L.ha_laia_hint_next(self._h, nxt)
                    rc = self._next(batch_id, mini_bs, dist, plan, cap, off)
                    if rc != 0:
                        raise _lib.HeraldAmdError("ha_laia_next failed: %s" % self._L.ha_last_error().decode())
                    self._emit(plan, dist, off, mini_bs, rank)
                    batch_id += 1
                    done += 1
        except Exception as e:   # surfaced by pop()
            self._error = e
        self._wall = (time.perf_counter() - t0, done)
        self._finish()

    def _next(self, batch_id, mini_bs, dist, plan, cap, off):
        # only this rank's plan is queued (laia_scheduler.cc:140-168): with the state on the device only its rows come back
        return self._L.ha_laia_next_for_rank(self._h, batch_id, mini_bs, self._cfg[4], dist.ctypes.data, plan.ctypes.data,
                                             cap, off.ctypes.data)

    def _emit(self, plan, dist, off, mini_bs, rank):
        # (arrays, copied out of the buffers the next call overwrites: turning ~20,000 keys into Python ints took the
        # scheduler thread as long as computing them -- `pop` does it for callers that want the reference's lists)
        self._q.put(plan[off[rank]:off[rank + 1]].copy())
        self._q.put(dist[rank * mini_bs:(rank + 1) * mini_bs].copy())

    def _finish(self):
        self._q.put(np.zeros(1, dtype=np.int64))

    def pop_arrays(self):
        """`pop` without the conversion: the plan as a uint64 array, the dist as an int64 array, the terminator as [0]."""
        item = self._q.get()
        if self._error is not None:
            raise self._error
        return item

    def pop(self):
        """Blocking; a list of Python ints, as the reference's pybind binding returns (laia/src/python_binding.cc:8-23)."""
        return self.pop_arrays().tolist()

    def length(self):
        return self._q.qsize()

    def snapshot_keys(self, worker):
        buf = np.empty(1 << 22, dtype=np.int32)
        n = self._L.ha_laia_snapshot_keys(self._h, int(worker), buf.ctypes.data, buf.size)
        return buf[:n].tolist()

    def timing(self):
        """Scheduler cost per global batch (us): whole ha_laia_next call, of which host greedy assignment and
        host snapshot (MiniLRU) bookkeeping; the rest is GPU kernels + transfers + waits."""
        t = np.zeros(4, dtype=np.float64)
        _lib.check(self._L.ha_laia_timing(self._h, t.ctypes.data), "ha_laia_timing")
        calls = max(t[0], 1.0)
        wall, done = getattr(self, "_wall", (0.0, 0))
        td = np.zeros(4, dtype=np.float64)
        _lib.check(self._L.ha_laia_timing_device(self._h, td.ctypes.data), "ha_laia_timing_device")
        return {"issue_us": td[1] / calls, "wait_us": td[2] / calls, "unpack_us": td[3] / calls,
                "batches": int(t[0]), "us_per_batch": t[1] / calls, "host_assign_us": t[2] / calls,
                "host_snapshot_us": t[3] / calls, "gpu_and_transfer_us": (t[1] - t[2] - t[3]) / calls,
                # the scheduler thread's loop as a whole (library calls + queueing plan and dist as Python lists)
                "thread_wall_us_per_batch": 1e6 * wall / done if done else None}

    def close(self):
        self._close = True
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._h:
            self._L.ha_laia_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# pre-profiled table orders (laia/src/topk_scheduler.cc:151-165) and how many of them count
# (python/hetu/laia/laia_dataloader.py:19-24)
TOPK_TABLE_ORDER = {
    "criteo": [9, 13, 22, 20, 12, 21, 17, 14, 24, 3, 5, 10, 16, 15, 19, 2, 4, 11, 7, 25, 23, 18, 8, 1, 0, 6],
    "avazu": [1, 2, 4, 5, 15, 7, 6, 16, 12, 0, 17, 8, 14, 10, 9, 11, 13, 3],
    "movie": [0, 1],
    "criteosearch": [0, 11, 3, 4, 5, 14, 1, 6, 2, 13, 16, 9, 8, 10, 12, 7, 15],
}
top_k_table = {"criteo": 20, "avazu": 17, "movie": 2, "criteosearch": 16}
local_worker_num = 8
_RING_WORDS = 1 << 24   # 128 MiB of uint64 words per local worker (the reference maps 1 GiB, :76)


class _Ring:
    """ha_shm_ring_* : message-framed single-producer / single-consumer ring in POSIX shared memory."""

    def __init__(self, L, name, create):
        self._L = L
        self._h = L.ha_shm_ring_open(name.encode(), 1 if create else 0, _RING_WORDS if create else 0)
        if not self._h:
            raise _lib.HeraldAmdError("ha_shm_ring_open(%s): %s" % (name, L.ha_last_error().decode()))
        self._buf = np.empty(1 << 16, dtype=np.uint64)

    def send(self, words, closing=lambda: False):
        """Blocks (polling every 10 us like push_to_local_worker, topk_scheduler.cc:204-222)."""
        a = np.ascontiguousarray(np.asarray(words, dtype=np.uint64))
        while not closing():
            rc = self._L.ha_shm_ring_send(self._h, a.ctypes.data, a.size)
            if rc == 1:
                return True
            if rc < 0:
                raise _lib.HeraldAmdError("message of %d words does not fit the ring" % a.size)
            time.sleep(10e-6)
        return False

    def recv(self):
        """Blocks until a message is there (pop_from_local_worker, topk_scheduler.cc:236-260)."""
        return self.recv_array().tolist()

    def recv_array(self):
        need = ctypes.c_int64(0)
        while True:
            n = self._L.ha_shm_ring_recv(self._h, self._buf.ctypes.data, self._buf.size, ctypes.byref(need))
            if n >= 0:
                return self._buf[:n].copy()
            if n == -2:
                self._buf = np.empty(int(need.value) + 16, dtype=np.uint64)
                continue
            time.sleep(10e-6)

    def pending(self):
        return int(self._L.ha_shm_ring_pending_words(self._h))

    def close(self):
        if self._h:
            self._L.ha_shm_ring_close(self._h)
            self._h = None


class TopkScheduler(LaiaScheduler):
    """laia_cache.TopkScheduler: LaiaScheduler restricted to the dataset's pre-profiled top-k tables,
    with per-thread quotas and own-sample plans (ha_laia_next_topk), standalone or local-shared."""

    def start(self, samples, num_sample, num_table, epoch_num, mini_batch_size, batch_num, nrank, rank,
              cache_size, num_threads, dataset, top_k_table, local_shared=False, local_rank=0, local_size=1,
              key_limit=None, device=None, ahead=None):
        if dataset not in TOPK_TABLE_ORDER:
            raise ValueError("dataset not supported")                           # :163-166
        order = TOPK_TABLE_ORDER[dataset]
        k = int(top_k_table) if top_k_table else int(num_table)                  # :131-133
        k = min(k, len(order))                                                   # :167-168
        self._order = np.ascontiguousarray(np.asarray(order[:k], dtype=np.int32))
        self._nt = int(num_threads)
        self._local_shared, self._local_rank, self._local_size = bool(local_shared), int(local_rank), int(local_size)
        self._rings, self._my_ring = [], None
        if self._local_shared:
            if self._local_rank == 0:
                # rank 0 creates every local worker's ring, then opens its own (:68-84)
                self._rings = [_Ring(self._L, "laia_cache_%d" % i, True) for i in range(self._local_size)]
            self._my_ring = _Ring(self._L, "laia_cache_%d" % self._local_rank, False)
            if self._local_rank != 0:
                return                                                           # only local rank 0 schedules (:176-180)
        super().start(samples, num_sample, num_table, epoch_num, mini_batch_size, batch_num, nrank, rank,
                      cache_size, num_threads, k, key_limit=key_limit, device=device, ahead=ahead)

    def _next(self, batch_id, mini_bs, dist, plan, cap, off):
        return self._L.ha_laia_next_topk(self._h, batch_id, mini_bs, self._order.ctypes.data, self._order.size,
                                         self._nt, dist.ctypes.data, plan.ctypes.data, cap, off.ctypes.data)

    def _emit(self, plan, dist, off, mini_bs, rank):
        if not self._local_shared:
            return super()._emit(plan, dist, off, mini_bs, rank)
        for i in range(self._local_size):                                        # :308-318
            w = rank + i
            if not (self._rings[i].send(plan[off[w]:off[w + 1]], lambda: self._close) and
                    self._rings[i].send(dist[w * mini_bs:(w + 1) * mini_bs], lambda: self._close)):
                return

    def _finish(self):
        if not self._local_shared:
            return super()._finish()
        for r in self._rings:                                                    # :350-353
            r.send([0], lambda: self._close)

    def pop_from_local_worker_arrays(self):
        assert self._local_shared
        item = self._my_ring.recv_array()
        if self._error is not None:
            raise self._error
        return item

    def pop_from_local_worker(self):
        return self.pop_from_local_worker_arrays().tolist()

    def length(self):
        if not self._local_shared:                                               # :187-193
            return super().length()
        return self._my_ring.pending()

    def report_cache_perf(self):
        """Average miss_pull / miss_push / update_pull / update_push per worker (:504-527)."""
        if self._thread is not None:
            self._thread.join()
        W = self._cfg[3]
        c = np.zeros(4 * W, dtype=np.int64)
        _lib.check(self._L.ha_laia_counters(self._h, c.ctypes.data), "ha_laia_counters")
        c = c.reshape(4, W)
        return {"miss_pull": int(c[0].sum() // W), "miss_push": int(c[1].sum() // W),
                "update_pull": int(c[2].sum() // W), "update_push": int(c[3].sum() // W),
                "per_worker": {"miss_pull": c[0].tolist(), "miss_push": c[1].tolist(),
                               "update_pull": c[2].tolist(), "update_push": c[3].tolist()},
                "on_device": int(self._L.ha_laia_on_device(self._h))}

    def close(self):
        super().close()
        if self._my_ring is not None:
            self._my_ring.close()
            self._my_ring = None
        for r in self._rings:
            r.close()
        self._rings = []


def native_plugin():
    """The pybind11 module `laia_cache` (csrc/py_laia_cache.cpp: the reference's plugin surface, laia/src/python_binding.cc:8-23,
    with the scheduler loop in a C++ thread of its own, as the reference's launch()) if it has been built, else None --
    the classes above run the same loop in a Python thread."""
    import importlib
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugins")
    try:
        import torch  # noqa: F401  (one HIP runtime for torch and the plugin)
        if d not in sys.path:
            sys.path.insert(0, d)
        return importlib.import_module("laia_cache")
    except Exception:      # noqa: BLE001 -- not built / not loadable: the Python classes serve
        return None


def topk_num_threads(mini_batch_size, wanted=80):
    """The reference starts TopkScheduler with 80 pool threads (laia_dataloader.py:85); its per-thread
    quotas only add up when the thread count divides the mini batch (otherwise it writes dist[-1]),
    so the glue takes the largest divisor of mini_batch_size that is <= wanted."""
    for nt in range(min(wanted, mini_batch_size), 0, -1):
        if mini_batch_size % nt == 0:
            return nt
    return 1


class _PlanDistStream:
    """The scheduler's output as (dist, plan) pairs.  The native schedulers emit `plan(0), dist(0), plan(1), dist(1), ...`
    and finally `[0]` (laia/src/laia_scheduler.cc:138-139, 166); the training loop wants dist(b) together with
    plan(b+1) -- the rows this worker holds that somebody else touches next step -- so the very first plan is
    dropped and every pair is (dist(b), plan(b+1)) (python/hetu/laia/laia_dataloader.py:108-114)."""

    def __init__(self, pop, ready):
        self._pop, self._ready = pop, ready
        self.closed = False
        self._take()                      # plan(0): nobody pushes before the first step

    def _take(self):
        if self.closed:
            raise RuntimeError("the laia scheduler's stream has ended; nothing left to read")
        msg = self._pop()
        if not isinstance(msg, (list, np.ndarray)):
            raise TypeError("laia scheduler returned %r, expected a list or an array" % type(msg))
        if len(msg) == 1 and int(msg[0]) == 0:     # terminator
            self.closed = True
            return msg[:0]
        return msg

    def pair(self):
        dist = self._take()
        plan = dist[:0] if self.closed else self._take()
        return dist, plan

    def pair_ready(self):
        return self._ready() >= 2


class LAIAScheduler:
    """The host side of a laia-scheduled epoch: which samples this worker trains on in batch b and which cached rows
    it pushes after it.  Public surface of python/hetu/laia/laia_dataloader.py:29-169 (`start`, `get_input_index`,
    `get_comm_plan`, `step_forward`, `samples_num / batch_num / batch_size`), own body: a window of WINDOW
    consecutive batches {batch id -> (dist, plan)} over the native scheduler's stream; a batch leaves the window once
    EVERY data loader that shares the scheduler has stepped past it, and the window then takes the next pair --
    without blocking the training loop while it still holds a batch the slowest loader has not consumed."""

    WINDOW = 5

    def __init__(self, sparse_data, batch_size, drop_last=True, dataset="criteo", local_shared=False):
        # the reference hands float32 ids over and the scheduler reads C ints (laia_dataloader.py:31)
        self.sparse_data = np.asarray(sparse_data, dtype=np.float32).astype(np.intc)
        self.batch_size, self.drop_last = batch_size, drop_last
        self.dataset, self.local_shared = dataset, local_shared
        self.sched = None

    def _native(self, nrank, rank, cache_limit, epoch_num, key_limit, local_rank, local_size):
        data = self.sparse_data
        head = (data, data.shape[0], data.shape[1], epoch_num, self.batch_size, self.batch_num, int(nrank), int(rank),
                int(cache_limit))
        plug = native_plugin() if os.environ.get("HA_LAIA_PYTHON_THREAD") != "1" else None
        if not self.local_shared:
            if plug is not None and key_limit is None:
                # the scheduler loop in the plugin's C++ thread: no interpreter between two global batches
                s = plug.LaiaScheduler()
                s.start(np.ascontiguousarray(data, dtype=np.uint64), *head[1:], 16, 24)
                return s, s.pop_arrays
            s = LaiaScheduler()
            s.start(*head, 16, 24, key_limit=key_limit)
            return s, s.pop_arrays
        # one scheduler per node: local rank 0 computes and feeds the others' shared-memory rings, so it has to
        # be up before they open theirs (laia_dataloader.py:72-95)
        s = TopkScheduler()
        if local_rank != 0:
            time.sleep(3)
        s.start(*head, topk_num_threads(self.batch_size), self.dataset, int(top_k_table[self.dataset]), True,
                int(local_rank), int(local_size), key_limit=key_limit)
        return s, s.pop_from_local_worker_arrays

    def start(self, nrank, rank, cache_limit, dataset_num=3, epoch_num=-1, key_limit=None, local_rank=0,
              local_size=local_worker_num):
        if self.sched is not None:
            raise RuntimeError("LAIAScheduler.start may be called once")
        self.samples_num = len(self.sparse_data) // nrank
        # the window must never hold the same batch id twice: at least WINDOW batches per epoch
        self.batch_size = min(int(self.batch_size), self.samples_num // self.WINDOW)
        if self.batch_size <= 0:
            raise ValueError("batch size %d is not usable with %d samples per worker" % (self.batch_size, self.samples_num))
        full, rest = divmod(self.samples_num, self.batch_size)
        self.batch_num = full if (self.drop_last or rest == 0) else full + 1
        self.sched, pop = self._native(nrank, rank, cache_limit, epoch_num, key_limit, local_rank, local_size)
        self._stream = _PlanDistStream(pop, self.sched.length)
        self._window = {b: self._stream.pair() for b in range(self.WINDOW)}
        self._cursor = [0] * dataset_num      # batches each data loader has consumed
        self._released = 0                    # batches that have left the window

    @property
    def init(self):
        return self.sched is not None

    @property
    def channel_close(self):
        return self._stream.closed

    def get_input_index(self, batch_id):
        """The samples of this worker in batch `batch_id`: a list, as the reference's (laia_dataloader.py:116-121)."""
        d = self._window[batch_id][0]
        return d.tolist() if isinstance(d, np.ndarray) else d

    def get_comm_plan(self, batch_id):
        p = self._window[batch_id][1]
        return p.tolist() if isinstance(p, np.ndarray) else p

    def get_input_index_array(self, batch_id):
        """... as an integer array (what LAIADataloader indexes with: no Python ints in between)."""
        return np.asarray(self._window[batch_id][0])

    def get_comm_plan_array(self, batch_id):
        return np.asarray(self._window[batch_id][1])

    def step_forward(self, dataset_id):
        self._cursor[dataset_id] += 1
        done = min(self._cursor)
        while self._released < done and not self._stream.closed:
            backlog = done - self._released
            if backlog < self.WINDOW and not self._stream.pair_ready():
                break                          # the scheduler is still computing and nobody is starved yet
            oldest = self._released % self.batch_num
            nxt = self._stream.pair()          # may block (or raise) before anything is changed
            del self._window[oldest]
            self._window[(oldest + self.WINDOW) % self.batch_num] = nxt
            self._released += 1


class LAIADataloader:
    """python/hetu/laia/laia_dataloader.py:152-230: the data loader of a laia-scheduled training set.  The
    samples of a batch are the ones the scheduler assigned to this worker (`get_input_index`); a sparse
    loader returns the TUPLE `(ids, comm_plan)` -- the plan is the one the scheduler paired with this
    batch (LAIAScheduler.start discards the first plan, so dist(b) travels with plan(b+1)) and becomes
    `IndexedSlices.push_indices` in EmbeddingLookUp_Gradient (EmbeddingLookUp.py:95-103), i.e. the push
    keys of `embedding_update_with_push_keys`.  The constructor fields are those of the reference's
    `Dataloader` base (python/hetu/dataloader.py:12-19).  Arrays are float32 like every reference
    NDArray (ndarray.array default dtype); with `device` they are torch tensors resident on it (the
    reference returns host NDArrays and lets the executor copy them), else numpy arrays."""

    def __init__(self, sched, sched_id, is_sparse, raw_data, batch_size, name="default", func=None,
                 drop_last=True, device=None):
        self.func = func if func else lambda x: x
        self.raw_data = np.array(self.func(raw_data), np.float32)
        self.batch_size = batch_size
        self.drop_last = drop_last
        self.name = str(name)
        assert drop_last, "drop_last must be True"
        self.sched = sched
        self.sched_id = sched_id
        self.is_sparse = is_sparse
        self.device = device
        self._raw_dev = None

    def init_states(self, rank=None, nrank=None):
        if nrank is None:
            nrank = 1
        self.samples_num = self.sched.samples_num
        self.batch_num = self.sched.batch_num
        self.batch_size = self.sched.batch_size
        self.batch_index = 0
        self.rank = rank
        if self.device is not None:
            import torch
            self._raw_dev = torch.from_numpy(self.raw_data).to(self.device)

    def _rows(self, idx):
        if self._raw_dev is None:
            return self.raw_data[idx]
        import torch
        return self._raw_dev[torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.device)]

    def _get_arr(self, batchind):
        idx = self.sched.get_input_index_array(self.batch_index)
        if not self.is_sparse:
            return self._rows(idx)
        plan = self.sched.get_comm_plan_array(self.batch_index).astype(np.float32)
        if self._raw_dev is not None:
            import torch
            plan = torch.from_numpy(plan).to(self.device)
        return (self._rows(idx), plan)

    def get_arr(self):
        """The current batch; steps the scheduler's queue forward (laia_dataloader.py:201-206)."""
        res = self._get_arr(self.batch_index)
        self.batch_index = (self.batch_index + 1) % self.batch_num
        self.sched.step_forward(self.sched_id)
        return res

    def get_next_arr(self):
        return self._get_arr(self.batch_index)

    def get_cur_shape(self):
        return tuple([len(self.sched.get_input_index_array(self.batch_index))] + list(self.raw_data.shape[1:]))
