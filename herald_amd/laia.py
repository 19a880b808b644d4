"""Host-side mirror of the `laia_cache` plugin and of its Python glue.

Reference surfaces mirrored (paths relative to /root/reference):
  laia_cache.LaiaScheduler().start(samples, num_sample, num_table, epoch_num, mini_batch_size,
        batch_num, nrank, rank, cache_size, num_threads, top_k_table) / .pop() / .length()
        laia/src/python_binding.cc:8-23, laia/src/laia_scheduler.cc:31-113.  The stream protocol is the
        reference's: alternating [plan, dist] lists, terminated by [0]; pop() blocks.
  LAIAScheduler(sparse_data, batch_size, ...)   python/hetu/laia/laia_dataloader.py:29-169
        (5-deep queue, the first plan is discarded so that dist(b) is paired with plan(b+1)).

The per-batch work is ha_laia_next in libherald_amd.so (csrc/laia.hip): probing and plan extraction on
the GPU, assignment and snapshot bookkeeping on the scheduler's own host thread -- like the
reference, the scheduler runs ahead of training in a background thread.
"""
import ctypes
import queue
import threading

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
              cache_size, num_threads=16, top_k_table=24, key_limit=None, device=None):
        samples = np.ascontiguousarray(np.asarray(samples).astype(np.uint64))   # pybind force-cast
        if samples.ndim != 2:
            raise RuntimeError("Input should be 2D numpy array")
        assert samples.shape == (num_sample, num_table)
        if key_limit is None:
            key_limit = int(samples.max()) + 1
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
        try:
            while epoch_id < epoch_num and not self._close:
                batch_id = 0
                epoch_id += 1
                if epoch_id == epoch_num:
                    batch_num += 1          # one more allocation for the cache prefetch (:126-128)
                while batch_id < batch_num and not self._close:
                    rc = self._L.ha_laia_next(self._h, batch_id, mini_bs, dist.ctypes.data, plan.ctypes.data, cap,
                                              off.ctypes.data)
                    if rc != 0:
                        raise _lib.HeraldAmdError("ha_laia_next failed: %s" % self._L.ha_last_error().decode())
                    self._q.put([int(x) for x in plan[off[rank]:off[rank + 1]]])
                    self._q.put([int(x) for x in dist[rank * mini_bs:(rank + 1) * mini_bs]])
                    batch_id += 1
        except Exception as e:   # surfaced by pop()
            self._error = e
        self._q.put([0])

    def pop(self):
        item = self._q.get()
        if self._error is not None:
            raise self._error
        return item

    def length(self):
        return self._q.qsize()

    def snapshot_keys(self, worker):
        buf = np.empty(1 << 22, dtype=np.int32)
        n = self._L.ha_laia_snapshot_keys(self._h, int(worker), buf.ctypes.data, buf.size)
        return buf[:n].tolist()

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


class LAIAScheduler:
    """python/hetu/laia/laia_dataloader.py:29-169 on top of the LaiaScheduler above."""

    def __init__(self, sparse_data, batch_size, drop_last=True, dataset="criteo", local_shared=False):
        self.sparse_data = np.array(sparse_data, np.float32).astype(np.intc)
        self.batch_size = batch_size
        self.drop_last = drop_last
        self.init = False
        self.dataset = dataset
        if local_shared:
            raise NotImplementedError("TopkScheduler / local-shared distribution is not built yet (SURVEY 8f.1)")

    def start(self, nrank, rank, cache_limit, dataset_num=3, epoch_num=-1, key_limit=None):
        assert not self.init, "LAIA scheduler can only be initialized once"
        self.samples_num = len(self.sparse_data) // nrank
        self.queue_size = 5
        self.batch_size = min(int(self.batch_size), self.samples_num // self.queue_size)
        assert self.batch_size > 0, "Batch size %d invalid." % self.batch_size
        self.batch_num = (int(np.ceil(self.samples_num / self.batch_size)) if not self.drop_last
                          else self.samples_num // self.batch_size)
        self.sched = LaiaScheduler()
        self.sched.start(self.sparse_data, self.sparse_data.shape[0], self.sparse_data.shape[1], epoch_num,
                         self.batch_size, self.batch_num, int(nrank), int(rank), int(cache_limit), 16, 24,
                         key_limit=key_limit)
        self.channel_close = False
        self.input_index, self.comm_plan, self.arr_map = [], [], {}
        for i in range(self.queue_size):
            if i == 0:
                self._channel_get()          # discard the first comm_plan
            self.input_index.append(self._channel_get())
            self.comm_plan.append(self._channel_get())
            self.arr_map[i] = i
        self.step = [0] * dataset_num
        self.cur_min_step = 0
        self.init = True

    def _channel_get(self):
        if self.channel_close:
            raise RuntimeError("Channle have been closed, but still try to get value from it")
        res = self.sched.pop()
        assert isinstance(res, list)
        if len(res) == 1 and res[0] == 0:
            self.channel_close = True
            return []
        return res

    def get_input_index(self, batch_id):
        return self.input_index[self.arr_map[batch_id]]

    def get_comm_plan(self, batch_id):
        return self.comm_plan[self.arr_map[batch_id]]

    def step_forward(self, dataset_id):
        self.step[dataset_id] += 1
        new_min_step = min(self.step)
        while self.cur_min_step < new_min_step:
            if self.channel_close or (self.sched.length() < 2 and new_min_step - self.cur_min_step < self.queue_size):
                break
            min_batch_id = self.cur_min_step % self.batch_num
            arr_index = self.arr_map.pop(min_batch_id)
            self.input_index[arr_index] = self._channel_get()
            self.comm_plan[arr_index] = self._channel_get()
            new_batch_id = (min_batch_id + self.queue_size) % self.batch_num
            self.arr_map[new_batch_id] = arr_index
            self.cur_min_step += 1
