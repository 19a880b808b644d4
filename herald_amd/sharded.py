"""Row-range sharded embedding table: the in-node replacement of Herald/Hetu's PS/worker split.

The reference keeps the embedding table on parameter servers and moves rows over ZMQ/ps-lite
(paths relative to /root/reference):
  * partition  : AveragePartitioner::partitionDense, ps-lite/include/ps/partitioner.h:46-57
                 (shard i holds len/S + (i < len%S) contiguous rows);
  * SparsePull : worker dedups the ids with a std::map (sorted unique), routes the unique keys to
                 their shards as shard-local offsets, servers gather, the worker scatters every
                 returned row to all its positions
                 (PSAgent::vecPullSparse, ps-lite/include/ps/worker/PSAgent.h:185-237;
                  PSHandler::serve(SparsePull), ps-lite/include/ps/server/PSFHandle.h:101-128);
  * SparsePush : worker reduces the values of equal ids in position order from 0, servers `+=`
                 (PSAgent::vecPushSparse, PSAgent.h:124-183; serve(SparsePush), PSFHandle.h:130-164),
                 after the Python op multiplied the values by -lr
                 (python/hetu/gpu_ops/ParameterServerCommunicate.py:58-59).

Here every rank owns one shard in its HBM and is a worker at the same time; the exchange is all-to-all
over RCCL (torch.distributed "nccl" backend == RCCL over xGMI):
    route: [count | shard-local keys] frames, ONE equal-split exchange   (once per batch, shared by pull and push)
    pull : [owner gathers]  ->  rows back  ->  [expand to the positions]
    push : [worker dedup-reduces]  ->  reduced rows  ->  [owner applies, rank order]
The routing of a batch depends on its ids only, and the ids are known one step ahead (the reference
prefetches them too: ParameterServerCommunicate.py:96-139), so `prefetch(ids)` enqueues it ahead of the
row exchanges of the current batch; the only host read-back of a step (n_unique and the 2W counts, which
size the row exchanges) is then long complete when the host needs it.
The reference's servers apply concurrent pushes in arrival order (non-deterministic); here an owner
applies the W incoming sorted lists in RANK order, `row = (row + r_0) + r_1 ...`, which is one of the
orders the reference can produce and makes the result reproducible.

All arithmetic goes through an *engine*: `HipEngine` (libherald_amd kernels; the product path) or a
test double injected by the CPU/gloo tests.  There is no CPU fallback in this module: without an
engine argument the HIP engine is used and fails loudly when the library or a GPU is missing.
"""
import ctypes

import torch
import torch.distributed as dist


def partition(rows, nshard):
    """AveragePartitioner::partitionDense (partitioner.h:46-57): starts[nshard+1]."""
    per, rem = divmod(int(rows), int(nshard))
    starts = [0]
    for i in range(nshard):
        starts.append(starts[-1] + per + (1 if i < rem else 0))
    return starts


class _SideStream:
    """`with` block that makes a side stream current (torch.cuda.set_stream both ways: the generic
    torch.cuda.stream() context costs several microseconds per use on the step's host path)."""
    __slots__ = ("side", "prev")

    def __init__(self, side, after_current):
        self.side = side
        self.prev = torch.cuda.current_stream()
        if after_current:
            side.wait_stream(self.prev)

    def __enter__(self):
        torch.cuda.set_stream(self.side)

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


class HipEngine:
    """Device compute of the sharded store through the C-ABI (libherald_amd)."""

    NSLOT = 3   # routing workspaces in rotation: current batch, prefetched batch, one spare

    def __init__(self, device):
        from . import _lib, ops
        self.ops = ops
        self.lib = _lib.load()
        self.check = _lib.check
        self.device = torch.device(device)
        self._devidx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.side = torch.cuda.Stream(device=self.device)
        self._slots = {}
        self._owner_plan = None
        self._bufs = {}

    def _buf(self, name, rows, width):
        """Grow-only scratch rows that never leave the store (exchange staging)."""
        b = self._bufs.get(name)
        if b is None or b.shape[0] < rows or b.shape[1] != width:
            b = torch.empty((max(rows, 1) * 5 // 4 + 16, width), dtype=torch.float32, device=self.device)
            self._bufs[name] = b
        return b[:rows]

    def _stream(self):
        # raw handle of torch's current stream (torch.cuda.current_stream() builds a Python object and
        # resolves the device every time: ~3 us, a dozen times per step)
        return torch._C._cuda_getCurrentRawStream(self._devidx)

    # -- routing -------------------------------------------------------------------------------------
    def route_issue(self, ids, starts, slot, cap):
        """Enqueue plan build + routing frames for a batch of ids (float32 or int64) into the routing
        workspace `slot`.  -> RouteBuffers: plan, send / recv (int32[W, 1+cap] frames: count, shard-local
        keys), meta_all (int64[1+2W] = n_unique, send counts, receive counts), host (pinned copy of it),
        keys_recv (int32[W*cap]: the received key lists in rank order), free (event of its last consumer)."""
        n = ids.numel()
        w = len(starts) - 1
        st = self._slots.get(slot)
        if st is None or st.plan.capacity < n or st.meta.numel() != 1 + w or st.cap != cap:
            st = RouteBuffers()
            st.cap = cap
            st.plan = self.ops.IndexPlan(max(cap, 1), self.device)
            st.send = torch.empty((w, 1 + cap), dtype=torch.int32, device=self.device)
            st.recv = torch.empty((w, 1 + cap), dtype=torch.int32, device=self.device)
            st.meta_all = torch.empty(1 + 2 * w, dtype=torch.int64, device=self.device)
            st.meta = st.meta_all[:1 + w]       # n_unique, send counts (written by ha_shard_route_pack_*)
            st.recv_cnt = st.meta_all[1 + w:]   # receive counts (written by ha_shard_route_unpack)
            st.host = torch.empty(1 + 2 * w, dtype=torch.int64, pin_memory=True)
            st.ev_host, st.ev_ready, st.ev_free = (torch.cuda.Event() for _ in range(3))
            st.keys_recv = torch.empty(max(w * cap, 1), dtype=torch.int32, device=self.device)
            st.starts = (ctypes.c_int64 * (w + 1))(*starts)
            self._slots[slot] = st
        self.wait_event(st.free)   # the previous batch routed through this workspace is fully consumed
        st.free = None
        plan = st.plan
        if ids.dtype == torch.float32:
            fn = self.lib.ha_shard_route_pack_f32ids
        elif ids.dtype in (torch.int64, torch.uint64):
            fn = self.lib.ha_shard_route_pack_u64ids
        else:
            raise TypeError("ids must be float32 or (u)int64")
        self.check(fn(ids.data_ptr(), n, plan.ws.data_ptr(), st.starts, w, cap, st.meta.data_ptr(),
                      st.send.data_ptr(), self._stream()), "ha_shard_route_pack")
        plan.n = n
        plan._view = None
        plan._produced_on = self._stream()
        return st

    def route_unpack(self, st):
        """Received frames -> key lists in rank order + receive counts (device)."""
        self.check(self.lib.ha_shard_route_unpack(st.recv.data_ptr(), st.recv.shape[0], st.cap,
                                                  st.recv_cnt.data_ptr(), st.keys_recv.data_ptr(), self._stream()),
                   "ha_shard_route_unpack")

    def on_side(self, after_current=True):
        """Context: the side stream is current; with after_current it first waits for the work queued
        on the main stream (inputs produced there)."""
        return _SideStream(self.side, after_current)

    def record(self, ev=None):
        if ev is None:
            ev = torch.cuda.Event()
        ev.record()
        return ev

    def wait_event(self, ev):
        if ev is not None:
            ev.wait()      # on torch's current stream

    def to_host(self, b):
        """Asynchronous device -> pinned host copy of a RouteBuffers' [n_unique | send | recv]; -> event."""
        b.host.copy_(b.meta_all, non_blocking=True)
        return self.record(b.ev_host)

    def host_sync(self, ev):
        if ev is not None:
            ev.synchronize()

    # -- rows ----------------------------------------------------------------------------------------
    def gather_keys(self, table, keys_i32, scratch=None):
        """rows[j,:] = table[keys[j],:] for shard-local uint32 keys held in an int32 tensor; `scratch`
        names a reusable staging buffer (the result is then only valid until its next use)."""
        n = keys_i32.numel()
        out = (self._buf(scratch, n, table.shape[1]) if scratch else
               torch.empty((n, table.shape[1]), dtype=torch.float32, device=self.device))
        self.check(self.lib.ha_gather_u32keys(table.data_ptr(), table.shape[0], table.shape[1],
                                              keys_i32.data_ptr(), n, out.data_ptr(), self._stream()),
                   "ha_gather_u32keys")
        return out

    def expand(self, rows, plan):
        """out[i,:] = rows[inverse[i],:] -- every position receives its unique row (sparse.h:17-31)."""
        return self.gather_keys(rows, plan.inverse())

    def rows_buffer(self, name, rows, width):
        return self._buf(name, rows, width)

    def reduce_scaled(self, plan, values, scale):
        out = self._buf("reduced", max(plan.n, 1), values.shape[1])
        self.check(self.lib.ha_dedup_reduce_scaled(plan.ws.data_ptr(), plan.n, values.data_ptr(), values.shape[1],
                                                   ctypes.c_float(scale), out.data_ptr(), self._stream()),
                   "ha_dedup_reduce_scaled")
        return out

    def acc_apply(self, table, keys_i32, values):
        """table[key,:] = (table[key,:] + v_a) + v_b ... in the order the (key, value) pairs are listed."""
        n = keys_i32.numel()
        if n == 0:
            return
        p = self._owner_plan
        if p is None or p.capacity < n:
            p = self._owner_plan = self.ops.IndexPlan(max(n, 1) * 5 // 4 + 16, self.device)
        self.check(self.lib.ha_shard_serve_push(table.data_ptr(), table.shape[0], table.shape[1],
                                                keys_i32.data_ptr(), n, values.data_ptr(), p.ws.data_ptr(),
                                                self._stream()), "ha_shard_serve_push")


    # -- fixed frames (FramedStep): no launch or exchange size depends on a device-side count ---------------------
    def frames_buffers(self, w, rcap, n_cap, width):
        """Persistent buffers of one routing slot of a FramedStep."""
        fb = FrameBuffers()
        fb.w, fb.rcap, fb.n_cap = w, rcap, n_cap
        fb.plan = self.ops.IndexPlan(max(n_cap, 1), self.device)
        fb.ksend = torch.empty((w, 2 + rcap), dtype=torch.int32, device=self.device)
        fb.krecv = torch.empty((w, 2 + rcap), dtype=torch.int32, device=self.device)
        fb.keys_fixed = torch.empty(w * rcap, dtype=torch.int32, device=self.device)
        fb.rowmap = torch.empty(max(n_cap, 1), dtype=torch.int32, device=self.device)
        fb.posmap = torch.empty(max(n_cap, 1), dtype=torch.int32, device=self.device)
        fb.state = torch.zeros(2, dtype=torch.int32, device=self.device)
        fb.state_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        fb.ids = {}           # dtype -> static copy of the batch's ids (the graphs read this address)
        fb.n, fb.cur, fb.shape, fb.routed = 0, None, (0,), False
        return fb

    def frames_ids(self, fb, ids):
        """Copy a batch's ids to the slot's static buffer (the captured routing reads it from there)."""
        t = fb.ids.get(ids.dtype)
        if t is None:
            t = fb.ids[ids.dtype] = torch.zeros(max(fb.n_cap, 1), dtype=ids.dtype, device=self.device)
        fb.n = ids.numel()
        fb.cur = t[:fb.n]
        fb.cur.copy_(ids.reshape(-1), non_blocking=True)
        return fb.cur

    def frames_route(self, fb, starts):
        ids = fb.cur
        fn = self.lib.ha_shard_frames_route_f32ids if ids.dtype == torch.float32 else self.lib.ha_shard_frames_route_u64ids
        st = (ctypes.c_int64 * len(starts))(*starts)
        self.check(fn(ids.data_ptr(), fb.n, fb.plan.ws.data_ptr(), st, fb.w, fb.rcap, fb.ksend.data_ptr(),
                      fb.rowmap.data_ptr(), fb.posmap.data_ptr(), self._stream()), "ha_shard_frames_route")
        fb.plan.n = fb.n
        fb.plan._view = None

    def frames_unpack(self, fb, krecv):
        self.check(self.lib.ha_shard_frames_unpack(krecv.data_ptr(), fb.w, fb.rcap, fb.keys_fixed.data_ptr(),
                                                   fb.state.data_ptr(), self._stream()), "ha_shard_frames_unpack")
        fb.state_host.copy_(fb.state, non_blocking=True)

    def frames_overflowed(self, fb):
        return bool(fb.state_host[0].item())

    def frames_serve_pull(self, table, fb, rows_send):
        """rows_send[g * rcap + j, :] = table[key j of rank g] (zero rows in the unused slots)."""
        m = fb.w * fb.rcap
        self.check(self.lib.ha_gather_u32keys(table.data_ptr(), table.shape[0], table.shape[1], fb.keys_fixed.data_ptr(),
                                              m, rows_send.data_ptr(), self._stream()), "ha_gather_u32keys")

    def frames_expand(self, rows_recv, fb, out):
        """out[i, :] = the pulled row of position i (a slot index beyond the frames reads as a zero row)."""
        if fb.n:
            self.check(self.lib.ha_gather_u32keys(rows_recv.data_ptr(), rows_recv.shape[0], rows_recv.shape[1],
                                                  fb.posmap.data_ptr(), fb.n, out.data_ptr(), self._stream()),
                       "ha_gather_u32keys")

    def frames_reduce(self, fb, values, scale, rows_send, zero_flags):
        """rows_send[slot of unique key u, :] = 0 + scale * v_a + scale * v_b ... over the positions of u in order
        (PSAgent::vecPushSparse's worker-side reduce, PSAgent.h:124-183), straight into the push frames."""
        if fb.n:
            self.check(self.lib.ha_apply_mapped(rows_send.data_ptr(), fb.w * fb.rcap, rows_send.shape[1],
                                                fb.plan.ws.data_ptr(), fb.n, values.data_ptr(), ctypes.c_float(-scale),
                                                fb.rowmap.data_ptr(), None, zero_flags.data_ptr(), self._stream()),
                       "ha_apply_mapped")

    def frames_serve_push(self, table, fb, rows_recv):
        """Owner side: the w received lists applied in rank order; unused slots carry a key beyond any table."""
        m = fb.w * fb.rcap
        p = self._owner_plan
        if p is None or p.capacity < m:
            p = self._owner_plan = self.ops.IndexPlan(m + 16, self.device)
        self.check(self.lib.ha_shard_serve_push(table.data_ptr(), table.shape[0], table.shape[1],
                                                fb.keys_fixed.data_ptr(), m, rows_recv.data_ptr(), p.ws.data_ptr(),
                                                self._stream()), "ha_shard_serve_push")

    def graph_capture(self, fn):
        """Capture `fn()` (enqueues work on the current stream) into a hipGraph; -> object with .replay()."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):          # torch's capture stream becomes current: _stream() follows it
            fn()
        return g

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def empty_rows(self, rows, width):
        return torch.empty((rows, width), dtype=torch.float32, device=self.device)


class FrameBuffers:
    """Persistent buffers of one routing slot of a FramedStep (see HipEngine.frames_buffers)."""
    __slots__ = ("w", "rcap", "n_cap", "plan", "ksend", "krecv", "keys_fixed", "rowmap", "posmap", "state",
                 "state_host", "ids", "cur", "n", "shape", "routed")


class RouteBuffers:
    """Persistent device / pinned-host buffers of one routing workspace (see HipEngine.route_issue)."""
    __slots__ = ("plan", "cap", "send", "recv", "meta_all", "meta", "recv_cnt", "host", "keys_recv", "starts",
                 "free", "ev_host", "ev_ready", "ev_free")

    def __init__(self):
        self.free = None
        self.ev_host = self.ev_ready = self.ev_free = None


class Route:
    """Routing of one id batch: its plan, which owner gets which of its unique keys, and the keys this
    rank will be asked for.  Shared by the pull and the push of the batch."""
    __slots__ = ("buf", "plan", "pending", "u", "send_cnt", "recv_cnt", "keys_recv", "ready", "shape", "released")

    def __init__(self):
        self.pending = None
        self.ready = None
        self.released = False


class ShardedEmbedding:
    """One row-range shard per rank + all-to-all pull/push.  `table` is this rank's shard.

    All collectives of a store go through ONE communicator in program order (identical on every rank),
    so the prefetched routing exchanges can never cross the row exchanges differently on two ranks.
    side_group=True creates a second process group for the routing exchanges (they then overlap the
    row exchanges instead of queueing between them); construct the store on every rank of `group` at
    the same point of the program in that case."""

    def __init__(self, rows, width, device, group=None, engine=None, table=None, side_group=False, a2a=None,
                 max_ids=None, side_stream=None):
        """a2a: optional replacement of torch.distributed.all_to_all_single with the same arguments
        (out, inp, out_splits, in_splits, group) -- e.g. a host-staged exchange where the process group's
        backend cannot move device tensors (several ranks sharing one GPU under gloo in the tests).
        max_ids: the largest batch (ids per rank) any rank will route -- the size of the fixed routing frame,
        which must be the same on every rank; None: agreed at the first routing call (max over the ranks of
        their first batch; a later, larger batch is then an error).
        side_stream: True runs the routing on the engine's side stream (it then overlaps the row exchanges on
        the device), False on the caller's stream (fewer events and stream switches on the host: the step is
        host-bound at small batches).  Default: False (HA_SHARD_SIDE_STREAM=1 turns it on)."""
        import os
        self.group = group
        self._a2a_fn = a2a
        self.max_ids = None if max_ids is None else int(max_ids)
        self.side_stream = (os.environ.get("HA_SHARD_SIDE_STREAM") == "1") if side_stream is None else bool(side_stream)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows, self.width = int(rows), int(width)
        self.device = torch.device(device)
        self.starts = partition(rows, self.world)
        self.local_rows = self.starts[self.rank + 1] - self.starts[self.rank]
        self.engine = engine if engine is not None else HipEngine(self.device)
        if table is None:
            table = torch.zeros((self.local_rows, width), dtype=torch.float32, device=self.device)
        assert tuple(table.shape) == (self.local_rows, width)
        self.table = table
        self.stats = {"xgmi_bytes_out": 0, "xgmi_bytes_in": 0}
        self._slot = 0
        self._live = {}
        self.side_group = None
        if self.world > 1 and side_group:
            ranks = None if group is None else dist.get_process_group_ranks(group)
            self.side_group = dist.new_group(ranks=ranks)

    # -- exchange plumbing ---------------------------------------------------------------------------
    def _a2a(self, out, inp, out_splits, in_splits, group=None):
        if self.world == 1:
            out.copy_(inp)
        elif self._a2a_fn is not None:
            self._a2a_fn(out, inp, out_splits, in_splits, group if group is not None else self.group)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=group if group is not None else self.group)
        return out

    # -- routing: phase 1 (enqueue, no host wait) and phase 2 (host reads 1+2W counts, keys exchange) ----
    def _frame(self, n):
        """Keys per owner in the routing frame: fixed for the life of the store, identical on all ranks."""
        if self.max_ids is None:
            cap = int(n)
            if self.world > 1:
                t = torch.tensor([cap], dtype=torch.int64, device=self.device)
                if self._a2a_fn is not None and self.device.type == "cuda":
                    t = t.cpu()      # the group's backend cannot move device tensors (see a2a)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                cap = int(t.item())
            self.max_ids = max(cap, 1)
        return self.max_ids

    def _on_route_stream(self, after_current):
        import contextlib
        return self.engine.on_side(after_current) if self.side_stream else contextlib.nullcontext()

    def prefetch(self, ids, after_current=True):
        """Start the routing of a batch of ids: plan, routing frames ([count | shard-local keys] per owner),
        ONE equal-split all-to-all, received key lists, and the asynchronous read-back of the counts; no
        host wait.  Returns a Route for pull(route=) / push(route=).  after_current=False (side-stream
        routing only) when `ids` is not produced by work queued on the current stream."""
        eng = self.engine
        cap = self._frame(ids.numel())
        if ids.numel() > cap:
            raise ValueError("a batch of %d ids does not fit the routing frame of %d keys agreed for this store "
                             "(construct ShardedEmbedding with max_ids >= the largest batch)" % (ids.numel(), cap))
        r = Route()
        r.shape = tuple(ids.shape)
        slot = self._slot
        self._slot = (self._slot + 1) % eng.NSLOT
        old = self._live.get(slot)
        if old is not None and not old.released:
            raise RuntimeError("routing workspace %d is still in use by a batch that was prefetched but "
                               "neither pulled nor pushed (at most %d batches in flight)" % (slot, eng.NSLOT))
        self._live[slot] = r
        with self._on_route_stream(after_current):
            b = r.buf = eng.route_issue(ids.reshape(-1), self.starts, slot, cap)
            r.plan = b.plan
            self._a2a(b.recv, b.send, None, None, group=self.side_group)
            eng.route_unpack(b)
            r.pending = eng.to_host(b)
            r.ready = eng.record(b.ev_ready) if self.side_stream else None
        return r

    def complete(self, r):
        """Second phase of a route: the host reads n_unique and the 2W counts (waits for the read-back of
        prefetch); nothing is enqueued.  pull / push call it when needed."""
        if r.pending is False:
            return r
        b = r.buf
        self.engine.host_sync(r.pending)
        vals = b.host.tolist()
        w = self.world
        r.u, r.send_cnt, r.recv_cnt = vals[0], vals[1:1 + w], vals[1 + w:1 + 2 * w]
        r.keys_recv = b.keys_recv[:sum(r.recv_cnt)]
        r.pending = False
        return r

    def _release(self, r):
        """A consumer of a route's buffers is queued on the current stream.  The workspace may be reused
        once the LAST of them has run: pull and push of one batch both record here, the later record
        overwrites the earlier one (same stream, program order), and route_issue waits for it.  A route
        is live from prefetch() until its push (or, for pull-only use, its pull); with NSLOT = 3 workspaces
        at most two newer batches may be prefetched meanwhile -- prefetch() asserts that."""
        if self.side_stream:       # on one stream the program order already protects the workspace
            r.buf.free = self.engine.record(r.buf.ev_free)
        r.released = True

    def _account(self, send_cnt, recv_cnt, bytes_per_key_out, bytes_per_key_in):
        r = self.rank
        out_keys = sum(c for g, c in enumerate(send_cnt) if g != r)
        in_keys = sum(c for g, c in enumerate(recv_cnt) if g != r)
        self.stats["xgmi_bytes_out"] += out_keys * bytes_per_key_out + in_keys * bytes_per_key_in
        self.stats["xgmi_bytes_in"] += in_keys * bytes_per_key_out + out_keys * bytes_per_key_in

    # -- SparsePull -------------------------------------------------------------------------------------
    def pull(self, ids=None, route=None, return_route=False):
        """out[i,:] = table_global[ids[i],:] for this rank's batch of ids (or of a prefetched route)."""
        eng = self.engine
        if route is None:
            route = self.prefetch(ids)
        r = self.complete(route)
        eng.wait_event(r.ready)
        rows_send = eng.gather_keys(self.table, r.keys_recv, scratch="pull_send")
        rows_recv = eng.rows_buffer("pull_recv", r.u, self.width)
        self._a2a(rows_recv, rows_send, r.send_cnt, r.recv_cnt)
        out = eng.expand(rows_recv, r.plan)
        self._account(r.send_cnt, r.recv_cnt, 4, 4 * self.width)
        out = out.reshape(r.shape + (self.width,))
        self._release(r)
        return (out, r) if return_route else out

    # -- SparsePush -------------------------------------------------------------------------------------
    def push(self, ids, values, lr=None, route=None):
        """table_global[id,:] += sum over this rank's positions of (-lr * values)  (scale 1 if lr is None),
        pushes of different ranks applied in rank order.  `route` = the Route of the same ids (from
        prefetch / pull(return_route=True)) saves the routing exchange."""
        eng = self.engine
        if route is None:
            route = self.prefetch(ids)
        r = self.complete(route)
        scale = 1.0 if lr is None else -float(lr)
        eng.wait_event(r.ready)   # the plan and the keys were produced on the side stream
        reduced = eng.reduce_scaled(r.plan, values.reshape(-1, self.width), scale)
        rows_send = reduced[:r.u]
        rows_recv = eng.rows_buffer("push_recv", sum(r.recv_cnt), self.width)
        self._a2a(rows_recv, rows_send, r.recv_cnt, r.send_cnt)
        eng.acc_apply(self.table, r.keys_recv, rows_recv)
        self._account(r.send_cnt, r.recv_cnt, 4 * self.width, 0)
        self._release(r)

    # -- SSPushPull (push this batch, pull the next one): ParameterServerCommunicate.py:74-76 ------------
    def push_pull(self, push_ids, values, lr, pull_ids):
        self.push(push_ids, values, lr)
        return self.pull(pull_ids)

    # -- checkpoint format of the reference: raw fp32 `<name>_<part>.dat` per shard ------------------------
    CKPT_CHUNK_BYTES = 64 << 20

    def _ckpt_chunk_rows(self):
        return max(1, self.CKPT_CHUNK_BYTES // (4 * self.width))

    def save(self, path_prefix):
        """PSAgent ParamSave (PSAgent.h:447-476, PSFHandle.h:401-439): raw little-endian fp32 rows in
        `<prefix>_<rank>.dat`.  The shard is streamed through one 64 MiB staging buffer (pinned when the
        shard lives on a GPU): no whole-shard host copy, a 35 GB shard needs 64 MiB of host memory."""
        chunk = self._ckpt_chunk_rows()
        pin = self.table.is_cuda
        stage = torch.empty((min(chunk, max(self.local_rows, 1)), self.width), dtype=torch.float32, pin_memory=pin)
        with open("%s_%d.dat" % (path_prefix, self.rank), "wb") as f:
            for s in range(0, self.local_rows, chunk):
                e = min(self.local_rows, s + chunk)
                stage[:e - s].copy_(self.table[s:e])          # synchronous for pinned destinations
                f.write(memoryview(stage[:e - s].numpy()).cast("B"))

    def load(self, path_prefix):
        """Inverse of save; also reads tables written by the reference's servers (same raw layout)."""
        import os
        path = "%s_%d.dat" % (path_prefix, self.rank)
        want = self.local_rows * self.width * 4
        if os.path.getsize(path) != want:
            raise ValueError("%s holds %d bytes, this shard needs %d" % (path, os.path.getsize(path), want))
        chunk = self._ckpt_chunk_rows()
        pin = self.table.is_cuda
        stage = torch.empty((min(chunk, max(self.local_rows, 1)), self.width), dtype=torch.float32, pin_memory=pin)
        with open(path, "rb") as f:
            for s in range(0, self.local_rows, chunk):
                e = min(self.local_rows, s + chunk)
                got = f.readinto(memoryview(stage[:e - s].numpy()).cast("B"))
                assert got == (e - s) * self.width * 4
                self.table[s:e].copy_(stage[:e - s])
                if pin:
                    torch.cuda.current_stream().synchronize()   # the staging buffer is reused


class FramedStep:
    """The sharded step with FIXED frames: every launch and every exchange has a size the host knows without reading
    anything back, so a step replays from hipGraphs (pull and push are one graph each; the routing of the batch two
    steps ahead rides in the pull graph).  Same semantics as ShardedEmbedding.pull / push (PSAgent::vecPullSparse /
    vecPushSparse, PSAgent.h:124-237; rank-ordered server `+=`, PSFHandle.h:130-164).

        fs = FramedStep(emb, max_ids)                  # row_cap: rows per owner and exchange (default max_ids / W)
        fs.start(ids0, ids1)                           # routes the first two batches
        rows0 = fs.pull(ids2)                          # rows of batch 0; batch 2 is routed meanwhile
        fs.push(grads0, lr)                            # batch 0 applied on its owners (rank order)
        rows1 = fs.pull(ids3) ...                      # ahead ids = None once the stream of batches ends

    Per owner g a batch may name at most row_cap unique keys.  A batch that names more on ANY rank is detected by the
    routing itself (the flag travels in the key frames, so all ranks agree) two steps before it is pulled; that batch
    alone takes ShardedEmbedding's sized exchange (one host read-back), the others keep replaying.  The host reads one
    pinned word per step, written two steps earlier -- it never waits for the step in flight.

    graphs=False enqueues the same kernels and exchanges eagerly (exchanges that cannot be captured: the host-staged
    all-to-all of the one-GPU multi-rank tests, the CPU engine of the gloo tests)."""

    NSLOT = 3

    def __init__(self, emb, max_ids, row_cap=None, graphs=True):
        self.emb, self.eng = emb, emb.engine
        w = emb.world
        self.max_ids = int(max_ids)
        self.rcap = int(row_cap) if row_cap is not None else max(-(-self.max_ids // w), 1)
        self.graphs = bool(graphs)
        self.slots = [self.eng.frames_buffers(w, self.rcap, self.max_ids, emb.width) for _ in range(self.NSLOT)]
        m = w * self.rcap
        self.pull_send = self.eng.empty_rows(m, emb.width)
        self.push_send = self.eng.empty_rows(m, emb.width)
        self.pull_recv = self.eng.empty_rows(m, emb.width)
        self.push_recv = self.eng.empty_rows(m, emb.width)
        self.zero_flags = self.eng.zeros((m,), torch.uint8)
        self._graphs = {}
        self._ev = {}                # batch index -> event behind the enqueue of its routing
        self.k = None
        self.fallbacks = 0
        emb._frame(self.max_ids)     # the sized path (overflowed batches) agrees on its frame now, on every rank

    # -- plumbing ---------------------------------------------------------------------------------------------------
    def _exchange(self, out, inp):
        """Equal-split all-to-all of whole frames; at world size 1 the frames ARE the received frames."""
        if self.emb.world == 1:
            return inp
        self.emb._a2a(out, inp, None, None)
        return out

    def _slot(self, j):
        return self.slots[j % self.NSLOT]

    def _route(self, j):
        """Enqueue the routing of batch j (its ids are in the slot's static buffer)."""
        fb = self._slot(j)
        self.eng.frames_route(fb, self.emb.starts)
        krecv = self._exchange(fb.krecv, fb.ksend)
        self.eng.frames_unpack(fb, krecv)

    def _pull(self, j, out):
        fb = self._slot(j)
        self.eng.frames_serve_pull(self.emb.table, fb, self.pull_send)
        got = self._exchange(self.pull_recv, self.pull_send)
        self.eng.frames_expand(got, fb, out)

    def _push(self, j, values, scale):
        fb = self._slot(j)
        self.eng.frames_reduce(fb, values, scale, self.push_send, self.zero_flags)
        got = self._exchange(self.push_recv, self.push_send)
        self.eng.frames_serve_push(self.emb.table, fb, got)

    def _run(self, key, fn):
        """Enqueue `fn`: eagerly at the first use of `key` (lazy one-time initialisation -- kernel attributes, scratch
        allocations -- must not fall into a capture), captured into a hipGraph at the second, replayed from then on."""
        if not self.graphs:
            fn()
            return
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 256:
                self._graphs.clear()
            self._graphs[key] = False
            fn()
            return
        if g is False:
            try:
                g = self._graphs[key] = self.eng.graph_capture(fn)
            except Exception as e:          # an exchange that cannot be captured: stay eager from here on
                import warnings
                warnings.warn("FramedStep: hipGraph capture failed (%s); continuing without graphs" % (e,))
                self.graphs = False
                fn()
                return
        g.replay()

    def _stage(self, j, ids):
        """Batch j enters the pipeline: its ids go to the static buffer of its slot.  ids = None: the stream of batches
        has ended (on every rank); an EMPTY tensor is a batch in which this rank names nothing -- it still takes part
        in the exchanges."""
        fb = self._slot(j)
        if ids is None:
            fb.n, fb.cur, fb.shape, fb.routed = 0, None, (0,), False
            return False
        if ids.numel() > self.max_ids:
            raise ValueError("a batch of %d ids exceeds max_ids = %d of this FramedStep" % (ids.numel(), self.max_ids))
        self.eng.frames_ids(fb, ids)
        fb.shape = tuple(ids.shape)
        fb.routed = True
        return True

    def _account(self):
        emb = self.emb
        w = emb.world
        if w > 1:
            per_peer = 4 * (2 + self.rcap) + 2 * 4 * self.rcap * emb.width     # key frame + pull rows + push rows
            emb.stats["xgmi_bytes_out"] += (w - 1) * per_peer
            emb.stats["xgmi_bytes_in"] += (w - 1) * per_peer

    # -- the stream protocol ----------------------------------------------------------------------------------------
    def start(self, ids0, ids1=None):
        """Route the first two batches of the stream (ids1 = None: a stream of one batch)."""
        self.k = 0
        self._ev = {}
        for j, ids in ((0, ids0), (1, ids1)):
            if self._stage(j, ids):
                fb = self._slot(j)
                self._run(("route", j % self.NSLOT, fb.n, fb.cur.dtype), lambda j=j: self._route(j))
            self._ev[j] = self.eng.record()
        return self

    def _overflowed(self, j):
        """Host: did any rank overflow its frames for batch j?  Reads the pinned word its routing wrote (enqueued two
        steps ago: the wait is for work that is long complete in steady state)."""
        fb = self._slot(j)
        if not fb.routed:
            return False
        self.eng.host_sync(self._ev.pop(j, None))
        return self.eng.frames_overflowed(fb)

    def pull(self, ahead_ids=None, out=None):
        """Rows of the current batch k; `ahead_ids` = batch k+2 (routed in the same graph), None at the end."""
        if self.k is None:
            raise RuntimeError("FramedStep.pull before start")
        k = self.k
        fb = self._slot(k)
        if not fb.routed:
            raise RuntimeError("FramedStep.pull: the stream of batches has ended")
        self._over = self._overflowed(k)
        ahead = self._stage(k + 2, ahead_ids)
        fa = self._slot(k + 2)
        width = self.emb.width
        if fb.n and out is None:
            out = self.eng.empty_rows(fb.n, width)
        if self._over:
            # sized exchange for this batch only (collective: every rank saw the flag); the routing ahead still runs
            self.fallbacks += 1
            if ahead:
                self._run(("route", (k + 2) % self.NSLOT, fa.n, fa.cur.dtype), lambda: self._route(k + 2))
            self._sized = self.emb.prefetch(fb.cur)
            rows = self.emb.pull(route=self._sized, return_route=False)
            if fb.n:
                out.copy_(rows.reshape(out.shape))
        else:
            def seg():
                if ahead:
                    self._route(k + 2)
                self._pull(k, out)
            self._run(("pull", k % self.NSLOT, fb.n, fa.n if ahead else -1, fa.cur.dtype if ahead else None,
                       out.data_ptr() if fb.n else 0), seg)
            self._account()
        self._ev[k + 2] = self.eng.record()
        return out.reshape(fb.shape + (width,)) if fb.n else None

    def push(self, values, lr=None):
        """Apply the gradients `values` of the current batch on its owners (scale -lr; 1 if lr is None)."""
        k = self.k
        fb = self._slot(k)
        scale = 1.0 if lr is None else -float(lr)
        if self._over:
            self.emb.push(None, values, lr, route=self._sized)
            self._sized = None
        else:
            v = values.reshape(-1, self.emb.width) if fb.n else None
            self._run(("push", k % self.NSLOT, fb.n, v.data_ptr() if fb.n else 0, scale),
                      lambda: self._push(k, v, scale))
        self.k = k + 1
