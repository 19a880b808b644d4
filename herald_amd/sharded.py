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
import os

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
    RADIX_FROM = 36864   # ids per batch above which the index plan is a radix sort (csrc/plan_dev.h kSmallMax)

    def __init__(self, device):
        from . import _lib, ops
        self.ops = ops
        self.lib = _lib.load()
        self.check = _lib.check
        self.device = torch.device(device)
        self._devidx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.side = torch.cuda.Stream(device=self.device)
        self._side_normal, self._side_high = self.side, None
        self._slots = {}
        self._owner_plan = None
        self._bufs = {}
        self._sort_streams = []

    def use_side_priority(self, high):
        """The routing stream at ANOTHER priority than the step's stream (high=True), or at the same.  HIP multiplexes the
        streams of one priority onto a few hardware queues, and two streams that share a queue run in submission order: the
        routing of block b + 1 then sits BETWEEN the steps of two blocks instead of beside them (kernel trace of bench.py's
        sharded leg at world size 1: every launch on one queue; docs/EXPERIMENTS.md round 6): 20.3 -> 18.5 us per step at
        configs[1]'s shape.  (While the plans of 106,496-id batches were sorted one batch at a time -- 16 x 6 chip-wide launches
        per block -- the higher priority pushed the steps aside, 74 -> 102 us at configs[2]'s shape; with the radix passes of a
        block's batches in six launches, plan_build_batch_radix, it is 57.6 against 58.4 at the steps' own priority.)"""
        import os
        if os.environ.get("HA_SHARD_SIDE_PRIO") in ("high", "normal"):
            high = os.environ["HA_SHARD_SIDE_PRIO"] == "high"
        if high and self._side_high is None:
            try:
                # (not an unlucky partner of the step stream: herald_amd/streams.py measures a few candidates once)
                from . import streams
                self._side_high = streams.pick_side_stream(torch.cuda.current_stream(self.device),
                                                           priority=min(torch.cuda.Stream.priority_range()))
            except Exception:      # noqa: BLE001
                self._side_high = self._side_normal
        new = self._side_high if high else self._side_normal
        if new is not self.side:
            new.wait_stream(self.side)        # whatever the other one still holds comes first
            self.side = new

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
    def frames_block(self, w, rcap, n_cap, block):
        """Persistent buffers of one routing block of a FramedStep: the key frames of `block` batches, laid out
        [owner][batch][2 + rcap] so that they travel in one equal-split all-to-all, and per batch the plan, the received
        keys in rank order, the row / position maps and the pinned overflow word."""
        blk = FrameBlock()
        blk.ksend = torch.empty((w, block, 2 + rcap), dtype=torch.int32, device=self.device)
        blk.krecv = torch.empty((w, block, 2 + rcap), dtype=torch.int32, device=self.device)
        blk.kgot, blk.ev, blk.ev_obj, blk.live, blk.synced, blk.carrays = None, None, None, False, False, None
        blk.slots = []
        for i in range(block):
            fb = FrameBuffers()
            fb.i, fb.w, fb.rcap, fb.stride = i, w, rcap, block * (2 + rcap)
            fb.plan = self.ops.IndexPlan(max(n_cap, 1), self.device)
            fb.keys_fixed = torch.empty(w * rcap, dtype=torch.int32, device=self.device)
            fb.rowmap = torch.empty(max(n_cap, 1), dtype=torch.int32, device=self.device)
            fb.posmap = torch.empty(max(n_cap, 1), dtype=torch.int32, device=self.device)
            # {overflow, keys received, send counts[w], receive counts[w]}: written by the routing kernels straight to
            # pinned host memory (sized frames read all of it, fixed frames the first two words), and its device copy
            fb.state_host = torch.zeros(2 + 2 * w, dtype=torch.int32).pin_memory()
            fb.state_c = (ctypes.c_int32 * (2 + 2 * w)).from_address(fb.state_host.data_ptr())
            fb.meta_dev = torch.zeros(2 + 2 * w, dtype=torch.int32, device=self.device)
            fb.n, fb.ids, fb.shape, fb.routed = 0, None, (0,), False
            blk.slots.append(fb)
        torch.cuda.current_stream(self.device).synchronize()   # (the zero fills: the routing runs on a side stream)
        return blk

    def _frame_ptr(self, t, fb):
        return t.data_ptr() + fb.i * (2 + fb.rcap) * 4

    def frames_plan(self, fb, starts):
        """First half of the routing of a batch: the index plan of its ids (keys < total rows)."""
        ids = fb.ids
        if fb.n:
            fn = self.lib.ha_plan_build_f32ids_lim if ids.dtype == torch.float32 else self.lib.ha_plan_build_u64ids_lim
            self.check(fn(ids.data_ptr(), fb.n, fb.plan.ws.data_ptr(), starts[-1], self._stream()), "ha_plan_build")
        fb.plan.n = fb.n
        fb.plan._view = None

    def frames_pack(self, blk, fb, starts):
        """Second half: key frames, row map (reduce) and position map (expand)."""
        st = (ctypes.c_int64 * len(starts))(*starts)
        self.check(self.lib.ha_shard_frames_pack(fb.plan.ws.data_ptr(), fb.n, st, fb.w, fb.rcap, fb.stride,
                                                 self._frame_ptr(blk.ksend, fb), fb.rowmap.data_ptr(),
                                                 fb.posmap.data_ptr(), self._stream()), "ha_shard_frames_pack")

    def frames_unpack(self, blk, fb):
        """Received key frames -> keys_fixed and the overflow word, written straight to pinned host memory (no copy
        in the stream: the host reads it a block later)."""
        self.check(self.lib.ha_shard_frames_unpack(self._frame_ptr(blk.kgot, fb), fb.w, fb.rcap, fb.stride,
                                                   fb.keys_fixed.data_ptr(), fb.state_host.data_ptr(), self._stream()),
                   "ha_shard_frames_unpack")

    def frames_route_block(self, blk, starts, exchange, sized_rank=None):
        """The routing of a block of batches in four launches and one key exchange: plans (two launches for all batches
        of one id dtype), key frames + maps (one), `exchange(krecv, ksend)` -> the received frames, received keys and
        overflow words (one).  Slots without a batch send empty frames.  sized_rank = this rank: the SIZED maps and the
        per-owner counts (device + pinned host) instead of the fixed-frame maps."""
        vp = ctypes.c_void_p
        slots = blk.slots
        big = [fb for fb in slots if fb.n > self.RADIX_FROM]
        # (round 6: ha_plan_build_batch_* runs the radix passes of ALL the block's batches per launch -- csrc/plan.hip
        # plan_build_batch_radix: six launches of 16 x 26 workgroups instead of 16 x 6 of 26; the four side streams below are what
        # the engine did before, HA_SHARD_BIG_STREAMS=1)
        if os.environ.get("HA_SHARD_BIG_STREAMS") != "1":
            big = []
        if len(big) > 1:
            # Batches beyond the counting sort's reach are radix-sorted: four launches of ~26 workgroups each per batch
            # -- a latency chain that leaves nine tenths of the chip idle (43 us at 106,496 ids).  The batches of a block
            # are independent: their sorts go to a few streams side by side and meet again before the key frames are packed.
            cur = torch.cuda.current_stream(self.device)
            if not self._sort_streams:
                self._sort_streams = [torch.cuda.Stream(device=self.device) for _ in range(4)]
            used = self._sort_streams[:min(4, len(big))]
            for st_ in used:
                st_.wait_stream(cur)
            for i, fb in enumerate(big):
                fn1 = self.lib.ha_plan_build_f32ids_lim if fb.ids.dtype == torch.float32 else self.lib.ha_plan_build_u64ids_lim
                st_ = used[i % len(used)]
                self.check(fn1(fb.ids.data_ptr(), fb.n, fb.plan.ws.data_ptr(), starts[-1], st_.cuda_stream), "ha_plan_build")
                fb.ids.record_stream(st_)
            for st_ in used:
                cur.wait_stream(st_)
        f32, u64 = [], []
        many_big = len(big) > 1
        for fb in slots:
            if fb.n and not (many_big and fb.n > self.RADIX_FROM):
                (f32 if fb.ids.dtype is torch.float32 else u64).append(fb)
        for sel, fn in ((f32, self.lib.ha_plan_build_batch_f32ids_lim), (u64, self.lib.ha_plan_build_batch_u64ids_lim)):
            if sel:
                cnt = len(sel)
                self.check(fn((vp * cnt)(*[fb.ids.data_ptr() for fb in sel]), (ctypes.c_int64 * cnt)(*[fb.n for fb in sel]),
                              (vp * cnt)(*[fb.plan.ws.data_ptr() for fb in sel]), cnt, starts[-1], self._stream()),
                           "ha_plan_build_batch")
        for fb in slots:
            fb.plan.n = fb.n
            fb.plan._view = None
        cnt = len(slots)
        fb0 = slots[0]
        c = blk.carrays            # the addresses of a block's buffers never change: converted once
        if c is None or c[0] != tuple(starts):
            c = blk.carrays = (tuple(starts), (ctypes.c_int64 * len(starts))(*starts),
                               (vp * cnt)(*[fb.plan.ws.data_ptr() for fb in slots]),
                               (vp * cnt)(*[fb.rowmap.data_ptr() for fb in slots]),
                               (vp * cnt)(*[fb.posmap.data_ptr() for fb in slots]),
                               (vp * cnt)(*[fb.keys_fixed.data_ptr() for fb in slots]),
                               (vp * cnt)(*[fb.state_host.data_ptr() for fb in slots]),
                               (vp * cnt)(*[fb.meta_dev.data_ptr() for fb in slots]),
                               (ctypes.c_int64 * cnt)())
        _, st, plans, rowmaps, posmaps, keysf, hosts, metas, ns = c
        for i, fb in enumerate(slots):
            ns[i] = fb.n
        if sized_rank is None:
            self.check(self.lib.ha_shard_frames_pack_batch(plans, ns, cnt, st, fb0.w, fb0.rcap, fb0.stride,
                                                           blk.ksend.data_ptr(), rowmaps, posmaps, self._stream()),
                       "ha_shard_frames_pack_batch")
            blk.kgot = exchange(blk.krecv, blk.ksend)
            self.check(self.lib.ha_shard_frames_unpack_batch(blk.kgot.data_ptr(), cnt, fb0.w, fb0.rcap, fb0.stride, keysf,
                                                             hosts, self._stream()), "ha_shard_frames_unpack_batch")
            return
        self.check(self.lib.ha_shard_frames_pack_batch_sized(plans, ns, cnt, st, fb0.w, int(sized_rank), fb0.rcap, fb0.stride,
                                                             blk.ksend.data_ptr(), rowmaps, posmaps, metas, hosts,
                                                             self._stream()), "ha_shard_frames_pack_batch_sized")
        blk.kgot = exchange(blk.krecv, blk.ksend)
        self.check(self.lib.ha_shard_frames_unpack_batch_sized(blk.kgot.data_ptr(), cnt, fb0.w, fb0.rcap, fb0.stride, keysf,
                                                               metas, hosts, self._stream()),
                   "ha_shard_frames_unpack_batch_sized")

    def frames_overflowed(self, fb):
        return fb.state_c[0] != 0

    def frames_serve_pull(self, table, blk, fb, rows_send):
        """rows_send[g * rcap + j, :] = table[key j of rank g] for the live slots of the received key frames."""
        self.check(self.lib.ha_shard_frames_serve_pull(table.data_ptr(), table.shape[0], table.shape[1],
                                                       self._frame_ptr(blk.kgot, fb), fb.w, fb.rcap, fb.stride,
                                                       rows_send.data_ptr(), fb.keys_fixed.data_ptr(),
                                                       fb.state_host.data_ptr(), self._stream()),
                   "ha_shard_frames_serve_pull")

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
        self.check(self.lib.ha_shard_frames_serve_push(table.data_ptr(), table.shape[0], table.shape[1],
                                                       fb.keys_fixed.data_ptr(), fb.w, fb.rcap, rows_recv.data_ptr(),
                                                       p.ws.data_ptr(), self._stream()), "ha_shard_frames_serve_push")

    # -- sized frames: rows in compact rank-ordered lists, exchanges sized by the real counts, own keys served locally ----
    def frames_counts(self, fb):
        """(send counts, receive counts) of a routed batch from the pinned words its routing wrote (host; call behind
        the block's event)."""
        w, c = fb.w, fb.state_c
        return [c[2 + g] for g in range(w)], [c[2 + w + g] for g in range(w)]

    def sized_serve_pull_call(self, table, fb, rank, rows_send):
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        return self._call(self.lib.ha_shard_sized_serve_pull, "ha_shard_sized_serve_pull", vp(table.data_ptr()),
                          i64(table.shape[0]), i64(table.shape[1]), vp(fb.keys_fixed.data_ptr()), ctypes.c_int(fb.w),
                          ctypes.c_int(rank), i64(fb.rcap), vp(fb.meta_dev.data_ptr()), vp(rows_send.data_ptr()),
                          vp(self._stream()))

    def sized_expand_call(self, table, rows_recv, fb, out):
        """out[i, :] = the row of position i: from this rank's own shard, or from the rows the owners sent."""
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        return self._call(self.lib.ha_gather2_u32map, "ha_gather2_u32map", vp(table.data_ptr()), i64(table.shape[0]),
                          vp(rows_recv.data_ptr()), i64(rows_recv.shape[0]), i64(table.shape[1]), vp(fb.posmap.data_ptr()),
                          i64(fb.n), vp(out.data_ptr()), vp(self._stream()))

    def sized_reduce_call(self, fb, values, scale, push_buf, zero_flags):
        """Occurrence-ordered reduce of scale * values by unique key straight into the push buffer (region A: keys of
        the other owners, compact; region S: this rank's own keys)."""
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        return self._call(self.lib.ha_apply_mapped, "ha_apply_mapped", vp(push_buf.data_ptr()), i64(push_buf.shape[0]),
                          i64(push_buf.shape[1]), vp(fb.plan.ws.data_ptr()), i64(fb.n), vp(values.data_ptr()),
                          ctypes.c_float(-scale), vp(fb.rowmap.data_ptr()), vp(None), vp(zero_flags.data_ptr()),
                          vp(self._stream()))

    def sized_serve_push_call(self, table, fb, rank, total, push_buf):
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        m = fb.w * fb.rcap
        p = self._owner_plan
        if p is None or p.capacity < m:
            p = self._owner_plan = self.ops.IndexPlan(m + 16, self.device)
        return self._call(self.lib.ha_shard_sized_serve_push, "ha_shard_sized_serve_push", vp(table.data_ptr()),
                          i64(table.shape[0]), i64(table.shape[1]), vp(fb.keys_fixed.data_ptr()), ctypes.c_int(fb.w),
                          ctypes.c_int(rank), i64(fb.rcap), vp(fb.meta_dev.data_ptr()), i64(total), vp(push_buf.data_ptr()),
                          vp(p.ws.data_ptr()), vp(self._stream()))

    def sized_push_alone_call(self, table, fb, values, scale):
        """World size 1: reduce + server add of the batch in one launch (every key is this rank's own)."""
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        return self._call(self.lib.ha_push_apply_scaled_finished, "ha_push_apply_scaled_finished", vp(table.data_ptr()),
                          i64(table.shape[0]), i64(table.shape[1]), vp(fb.plan.ws.data_ptr()), i64(fb.n),
                          vp(values.data_ptr()), ctypes.c_float(scale), vp(self._stream()))

    def sized_serve_pull(self, table, fb, rank, rows_send):
        self.sized_serve_pull_call(table, fb, rank, rows_send)()

    def sized_expand(self, table, rows_recv, fb, out):
        if fb.n:
            self.sized_expand_call(table, rows_recv, fb, out)()

    def sized_reduce(self, fb, values, scale, push_buf, zero_flags):
        if fb.n:
            self.sized_reduce_call(fb, values, scale, push_buf, zero_flags)()

    def sized_serve_push(self, table, fb, rank, total, push_buf):
        self.sized_serve_push_call(table, fb, rank, total, push_buf)()

    def sized_push_alone(self, table, fb, values, scale):
        if fb.n:
            self.sized_push_alone_call(table, fb, values, scale)()

    # the same launches as callables with their arguments converted once (a step is five launches of 4-9 us: converting
    # the arguments through ctypes on every call costs more host time than the kernels take)
    def _call(self, fn, what, *args):
        check = self.check

        def call():
            if fn(*args) != 0:
                check(-1, what)
        return call

    def frames_serve_pull_call(self, table, blk, fb, rows_send):
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        return self._call(self.lib.ha_shard_frames_serve_pull, "ha_shard_frames_serve_pull", vp(table.data_ptr()),
                          i64(table.shape[0]), i64(table.shape[1]), vp(self._frame_ptr(blk.kgot, fb)), ctypes.c_int(fb.w),
                          i64(fb.rcap), i64(fb.stride), vp(rows_send.data_ptr()), vp(fb.keys_fixed.data_ptr()),
                          vp(fb.state_host.data_ptr()), vp(self._stream()))

    def frames_expand_call(self, rows_recv, fb, out):
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        return self._call(self.lib.ha_gather_u32keys, "ha_gather_u32keys", vp(rows_recv.data_ptr()), i64(rows_recv.shape[0]),
                          i64(rows_recv.shape[1]), vp(fb.posmap.data_ptr()), i64(fb.n), vp(out.data_ptr()),
                          vp(self._stream()))

    def frames_reduce_call(self, fb, values, scale, rows_send, zero_flags):
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        return self._call(self.lib.ha_apply_mapped, "ha_apply_mapped", vp(rows_send.data_ptr()), i64(fb.w * fb.rcap),
                          i64(rows_send.shape[1]), vp(fb.plan.ws.data_ptr()), i64(fb.n), vp(values.data_ptr()),
                          ctypes.c_float(-scale), vp(fb.rowmap.data_ptr()), vp(None), vp(zero_flags.data_ptr()),
                          vp(self._stream()))

    def frames_serve_push_call(self, table, fb, rows_recv):
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        m = fb.w * fb.rcap
        p = self._owner_plan
        if p is None or p.capacity < m:
            p = self._owner_plan = self.ops.IndexPlan(m + 16, self.device)
        return self._call(self.lib.ha_shard_frames_serve_push, "ha_shard_frames_serve_push", vp(table.data_ptr()),
                          i64(table.shape[0]), i64(table.shape[1]), vp(fb.keys_fixed.data_ptr()), ctypes.c_int(fb.w),
                          i64(fb.rcap), vp(rows_recv.data_ptr()), vp(p.ws.data_ptr()), vp(self._stream()))

    def hold_for_side(self, t):
        if t is not None and t.is_cuda:
            t.record_stream(self.side)

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


class FrameBlock:
    """Key frames of one routing block of a FramedStep (see HipEngine.frames_block)."""
    __slots__ = ("ksend", "krecv", "kgot", "ev", "ev_obj", "live", "slots", "synced", "carrays")


class FrameBuffers:
    """Per-batch buffers inside a FrameBlock."""
    __slots__ = ("i", "w", "rcap", "stride", "plan", "keys_fixed", "rowmap", "posmap", "state_host", "state_c", "meta_dev", "out_shape",
                 "ids", "n", "shape", "routed", "send_cnt", "recv_cnt", "pos_local")


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


class NativeExchange:
    """The all-to-all of a store's keys and rows as RCCL point-to-point calls made by the LIBRARY on the caller's stream
    (ha_xchg_*, csrc/xchg.hip): what torch.distributed.all_to_all_single does -- a group of ncclSend / ncclRecv, one pair
    per peer with a non-zero count -- without the Python call, the split-size lists and c10d's own stream with its event
    hop there and back.  The communicator is the library's own: rank 0 of `group` makes the unique id, torch.distributed
    carries it over.  `create` returns None where no RCCL can be resolved, the group's backend is not nccl, or the new
    exchange does not reproduce torch's result on a first pattern (checked on every rank, agreed by an all-reduce)."""

    @classmethod
    def create(cls, group, device):
        """Every rank of `group` calls this at the same point of the program.  The ranks AGREE (all-reduce) before every step
        that is collective in RCCL or torch.distributed -- whether an RCCL library can be resolved at all, whether rank 0 got a
        unique id, whether every communicator came up, whether the first exchange reproduced torch's -- so that a rank that
        fails locally never leaves the others waiting inside a collective."""
        import os
        if os.environ.get("HA_NATIVE_XCHG") == "0" or not dist.is_initialized():
            return None
        try:
            if dist.get_backend(group) != "nccl":
                return None
        except Exception:      # noqa: BLE001
            return None

        def agree(flag):
            t = torch.tensor([1 if flag else 0], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        L = None
        try:
            from . import _lib
            L = _lib.load()
            have = bool(L.ha_xchg_available())
        except Exception:      # noqa: BLE001
            have = False
        if not agree(have):
            return None
        x = cls.__new__(cls)
        x.L, x.group, x.device = L, group, torch.device(device)
        x.world, x.rank = dist.get_world_size(group), dist.get_rank(group)
        x.h = None
        x._i64 = ctypes.c_int64 * x.world
        # rank 0's unique id (all zeros + a flag byte if it could not be made), carried over by torch.distributed
        uid = torch.zeros(129, dtype=torch.uint8)
        if x.rank == 0:
            try:
                if L.ha_xchg_unique_id(ctypes.c_void_p(uid.data_ptr())) == 0:
                    uid[128] = 1
            except Exception:      # noqa: BLE001
                pass
        uid = uid.to(x.device)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(uid, src=src, group=group)
        uid = uid.cpu()
        if int(uid[128]) != 1:      # (the same word on every rank)
            return None
        try:
            with torch.cuda.device(x.device):
                x.h = L.ha_xchg_create(ctypes.c_void_p(uid.data_ptr()), x.world, x.rank)
        except Exception:      # noqa: BLE001
            x.h = None
        if not agree(bool(x.h)):
            x.close()
            return None
        if not agree(x._self_check()):
            x.close()
            return None
        return x

    def close(self):
        if getattr(self, "h", None):
            self.L.ha_xchg_destroy(ctypes.c_void_p(self.h))
            self.h = None

    def all_to_all(self, out, inp, out_splits, in_splits):
        """torch.distributed.all_to_all_single(out, inp, out_splits, in_splits) on the current stream: splits in rows of
        dim 0 (None: equal parts)."""
        w = self.world
        row_in = inp.element_size() * (inp.numel() // inp.shape[0] if inp.shape[0] else 0)
        row_out = out.element_size() * (out.numel() // out.shape[0] if out.shape[0] else 0)
        if in_splits is None:
            in_splits = [inp.shape[0] // w] * w
        if out_splits is None:
            out_splits = [out.shape[0] // w] * w
        sb = self._i64(*[int(c) * row_in for c in in_splits])
        rb = self._i64(*[int(c) * row_out for c in out_splits])
        s = torch.cuda.current_stream(self.device).cuda_stream
        if self.L.ha_xchg_bytes(ctypes.c_void_p(self.h), ctypes.c_void_p(inp.data_ptr()), sb, ctypes.c_void_p(out.data_ptr()),
                                rb, ctypes.c_void_p(s)) != 0:
            raise RuntimeError("ha_xchg_bytes: " + self.L.ha_last_error().decode())
        return out

    def _self_check(self):
        """A small exchange with a different row count for every (sender, receiver) pair, through this object and through
        torch.distributed: equal on this rank?"""
        try:
            w, r, width = self.world, self.rank, 8
            ins = [0 if g == r else 1 + (r + 2 * g) % 3 for g in range(w)]
            outs = [0 if g == r else 1 + (g + 2 * r) % 3 for g in range(w)]
            inp = torch.cat([torch.full((c, width), float(100 * r + g), device=self.device) for g, c in enumerate(ins)] +
                            [torch.zeros((0, width), device=self.device)])
            want = torch.empty((sum(outs), width), device=self.device)
            got = torch.full((sum(outs), width), -1.0, device=self.device)
            dist.all_to_all_single(want, inp, outs, ins, group=self.group)
            self.all_to_all(got, inp, outs, ins)
            torch.cuda.synchronize(self.device)
            return bool(torch.equal(got, want))
        except Exception:      # noqa: BLE001
            return False


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
        # the exchanges as RCCL calls made by the library (csrc/xchg.hip) where that is possible and reproduces torch's
        # exchange; one communicator per process group in use (the routing's exchanges run on another stream)
        self.native = self.native_side = None
        if self.world > 1 and a2a is None:
            self.native = NativeExchange.create(group, self.device)
            if self.native is not None and self.side_group is not None:
                self.native_side = NativeExchange.create(self.side_group, self.device)

    # -- exchange plumbing ---------------------------------------------------------------------------
    def _a2a(self, out, inp, out_splits, in_splits, group=None):
        if self.world == 1:
            out.copy_(inp)
        elif self._a2a_fn is not None:
            self._a2a_fn(out, inp, out_splits, in_splits, group if group is not None else self.group)
        elif self.native is not None and (group is None or group is self.group):
            self.native.all_to_all(out, inp, out_splits, in_splits)
        elif self.native_side is not None and group is self.side_group:
            self.native_side.all_to_all(out, inp, out_splits, in_splits)
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
    anything back, so the pull and the push of a step replay from one hipGraph each (three launches and one
    equal-split all-to-all per graph).  Same semantics as ShardedEmbedding.pull / push (PSAgent::vecPullSparse /
    vecPushSparse, PSAgent.h:124-237; rank-ordered server `+=`, PSFHandle.h:130-164).

    The ROUTING of the batches -- index plan, key frames, key exchange -- depends on their ids only, and the reference
    prefetches ids too (ParameterServerCommunicate.py:147-185).  It runs a BLOCK of `block` batches at a time, one
    block ahead of the steps that use it, on the engine's side stream beside them: per block one launch chain per
    batch and ONE key exchange for all of them (frames laid out [owner][batch][...]), instead of one exchange per step.
    Ids are needed LOOKAHEAD = 2 * block batches ahead.

        fs = FramedStep(emb, max_ids, block=8)
        fs.start(ids[:fs.LOOKAHEAD])                   # the first LOOKAHEAD batches (fewer if the stream is shorter)
        rows0 = fs.pull(ids[LOOKAHEAD])                # rows of batch 0; one more batch enters the pipeline
        fs.push(grads0, lr)                            # batch 0 applied on its owners (rank order)
        rows1 = fs.pull(ids[LOOKAHEAD + 1]) ...        # ahead ids = None once the stream of batches has ended

    Per owner a batch may name at most row_cap unique keys (default 1.5 x max_ids / W, at most max_ids).  A batch that names more on ANY rank
    is detected by the routing itself (the flag travels in the key frames, so all ranks agree) a block before it is
    pulled; that batch alone takes ShardedEmbedding's sized exchange (one host read-back), the others keep replaying.

    graphs=False enqueues the same kernels and exchanges eagerly (exchanges that cannot be captured: the host-staged
    all-to-all of the one-GPU multi-rank tests, the CPU engine of the gloo tests).  At world size > 1 the routing runs
    beside the steps only if the store has a second communicator (side_group=True); otherwise at the block start on
    the caller's stream."""

    def __init__(self, emb, max_ids, row_cap=None, block=8, graphs=True, sized=None):
        """sized (default: on unless graphs): the row exchanges carry exactly the rows the batch names -- the per-owner
        counts of a batch reach pinned host memory with its routing, a block before the step that needs them, so sizing
        the all-to-alls by them stalls nothing -- in compact rank-ordered lists, and the keys a rank owns itself are
        served from / applied to its shard without entering a frame.  sized=False keeps the FIXED row frames of
        row_cap rows per peer (equal-split exchanges whose sizes never depend on a count: the form that replays from
        hipGraphs)."""
        self.emb, self.eng = emb, emb.engine
        w = emb.world
        self.max_ids = int(max_ids)
        # default: 1.5 x the even share -- row-range shards are not evenly loaded (at W = 8 the owner of the small Criteo
        # tables is named 944 unique keys by a 6,656-id batch whose even share is 832), and a batch beyond the cap costs a
        # sized exchange on every rank
        self.rcap = int(row_cap) if row_cap is not None else min(self.max_ids, max(-(-3 * self.max_ids // (2 * w)), 1))
        self.block = int(block)
        if self.block < 1:
            raise ValueError("block must be >= 1")
        self.LOOKAHEAD = 2 * self.block
        self.graphs = bool(graphs)
        self.sized = (not self.graphs) if sized is None else bool(sized)
        if self.sized and self.graphs:
            raise ValueError("FramedStep: sized exchanges are enqueued per step (graphs=True needs sized=False)")
        self.blocks = [self.eng.frames_block(w, self.rcap, self.max_ids, self.block) for _ in range(3)]
        m = w * self.rcap
        self.pull_send = self.eng.empty_rows(m, emb.width)
        self.pull_recv = self.eng.empty_rows(m, emb.width)
        if self.sized:
            # regions A (rows for the other owners) | S (own keys) | B (rows received): see csrc/shard.hip
            self.push_buf = self.eng.empty_rows((2 * w + 1) * self.rcap, emb.width)
            self.zero_flags = self.eng.zeros(((2 * w + 1) * self.rcap,), torch.uint8)
        else:
            self.push_send = self.eng.empty_rows(m, emb.width)
            self.push_recv = self.eng.empty_rows(m, emb.width)
            self.zero_flags = self.eng.zeros((m,), torch.uint8)
        self.side = w == 1 or emb.side_group is not None
        if hasattr(self.eng, "use_side_priority"):
            self.eng.use_side_priority(True)
        self._graphs = {}
        self._calls = {}
        self._fast_ok = hasattr(self.eng, "frames_serve_pull_call")
        self.k = None
        self.fallbacks = 0
        self._ring = 3 * self.block
        self._slot_of = [(self.blocks[(j // self.block) % 3], self.blocks[(j // self.block) % 3].slots[j % self.block])
                         for j in range(3 * self.block)]
        emb._frame(self.max_ids)     # the sized path (overflowed batches) agrees on its frame now, on every rank

    # -- plumbing ---------------------------------------------------------------------------------------------------
    def _exchange(self, out, inp, group=None):
        """Equal-split all-to-all of whole frames; at world size 1 the frames ARE the received frames."""
        if self.emb.world == 1:
            return inp
        self.emb._a2a(out, inp, None, None, group=group)
        return out

    def _fb(self, j):
        return self._slot_of[j % self._ring]

    def _stage(self, j, ids):
        """Batch j enters the pipeline.  ids = None: the stream of batches has ended (on every rank); an EMPTY tensor is
        a batch in which this rank names nothing -- it still takes part in the exchanges."""
        _, fb = self._fb(j)
        if ids is None:
            fb.n, fb.ids, fb.shape, fb.routed, fb.out_shape = 0, None, (0,), False, (0, self.emb.width)
            return
        n = ids.numel()
        if n > self.max_ids:
            raise ValueError("a batch of %d ids exceeds max_ids = %d of this FramedStep" % (n, self.max_ids))
        dt = ids.dtype
        if dt is not torch.float32 and dt is not torch.int64 and dt is not torch.uint64:
            raise TypeError("ids must be float32 or (u)int64")
        shape = tuple(ids.shape)
        fb.ids, fb.n, fb.shape, fb.routed = (ids if len(shape) == 1 else ids.reshape(-1)), n, shape, True
        fb.out_shape = shape + (self.emb.width,)

    def _route_block(self, b, after=None):
        """Enqueue the routing of block b: plans and key frames of its batches, one key exchange, received keys.  after: an
        event of the step stream the routing stream waits for INSTEAD of everything queued on the step stream so far (steps():
        the block's steps are enqueued first, its successor's routing follows them onto the device without waiting for them)."""
        import contextlib
        blk = self.blocks[b % 3]
        blk.live = any(fb.routed for fb in blk.slots)
        blk.synced = False
        if not blk.live:
            blk.ev = None
            return
        eng, starts = self.eng, self.emb.starts
        if after is not None and self.side:
            eng.side.wait_event(after)
        ctx = eng.on_side(after_current=after is None) if self.side else contextlib.nullcontext()
        with ctx:
            # (the id tensors of the block stay referenced by their slots until they are restaged three blocks later, long
            # after this routing has completed: the allocator cannot hand their memory out under the side stream)
            exchange = lambda out, inp: self._exchange(out, inp, group=self.emb.side_group if self.side else None)
            sized_rank = self.emb.rank if self.sized else None
            if hasattr(eng, "frames_route_block"):
                eng.frames_route_block(blk, starts, exchange, sized_rank)
            else:
                for fb in blk.slots:       # a slot without a batch (the stream ended inside the block): an empty frame
                    eng.frames_plan(fb, starts)
                    eng.frames_pack(blk, fb, starts, sized_rank)
                blk.kgot = exchange(blk.krecv, blk.ksend)
                for fb in blk.slots:
                    if fb.routed:
                        eng.frames_unpack(blk, fb)
            blk.ev = blk.ev_obj = eng.record(blk.ev_obj)      # (one event object per block, re-recorded)
        for fb in blk.slots:
            fb.send_cnt = fb.recv_cnt = None
        if self.emb.world > 1:
            per_peer = 4 * self.block * (2 + self.rcap)
            self.emb.stats["xgmi_bytes_out"] += (self.emb.world - 1) * per_peer
            self.emb.stats["xgmi_bytes_in"] += (self.emb.world - 1) * per_peer
            self.emb.stats["xgmi_key_frame_bytes"] = self.emb.stats.get("xgmi_key_frame_bytes", 0) + (self.emb.world - 1) * per_peer

    def _block_start(self, b):
        """First step of block b: its routing (enqueued a block ago) must be complete before its first pull; the routing
        of block b+1 starts beside the steps of block b."""
        self.eng.wait_event(self.blocks[b % 3].ev)
        self._route_block(b + 1)

    def _pull(self, j, out):
        blk, fb = self._fb(j)
        self.eng.frames_serve_pull(self.emb.table, blk, fb, self.pull_send)
        got = self._exchange(self.pull_recv, self.pull_send)
        self.eng.frames_expand(got, fb, out)

    def _push(self, j, values, scale):
        _, fb = self._fb(j)
        self.eng.frames_reduce(fb, values, scale, self.push_send, self.zero_flags)
        got = self._exchange(self.push_recv, self.push_send)
        self.eng.frames_serve_push(self.emb.table, fb, got)

    # -- sized exchanges -----------------------------------------------------------------------------------------------
    def _counts(self, fb):
        if fb.send_cnt is None:
            fb.send_cnt, fb.recv_cnt = self.eng.frames_counts(fb)
        return fb.send_cnt, fb.recv_cnt

    def _pull_sized(self, j, out, stream_key):
        blk, fb = self._slot_of[j % self._ring]
        eng, emb = self.eng, self.emb
        w, r = emb.world, emb.rank
        if w == 1 and self._fast_ok:       # the whole pull is one launch; its arguments are converted once per buffer set
            if fb.n:
                key = ("sexp", j % self._ring, fb.n, out.data_ptr(), stream_key)
                c = self._calls.get(key)
                if c is None:
                    c = self._fast(key, lambda: eng.sized_expand_call(emb.table, self.pull_recv, fb, out))
                c()
            return
        if w > 1:
            send_cnt, recv_cnt = self._counts(fb)
            ins = [c if g != r else 0 for g, c in enumerate(recv_cnt)]     # rows this rank serves to peer g
            outs = [c if g != r else 0 for g, c in enumerate(send_cnt)]    # rows owner g sends back
            if self._fast_ok:
                self._fast(("spull", j % self._ring, stream_key),
                           lambda: eng.sized_serve_pull_call(emb.table, fb, r, self.pull_send))()
            else:
                eng.sized_serve_pull(emb.table, fb, r, self.pull_send)
            emb._a2a(self.pull_recv[:sum(outs)], self.pull_send[:sum(ins)], outs, ins)
            self._account_sized(ins, outs, send_cnt, recv_cnt)
        if fb.n:
            if self._fast_ok:
                self._fast(("sexp", j % self._ring, fb.n, out.data_ptr(), stream_key),
                           lambda: eng.sized_expand_call(emb.table, self.pull_recv, fb, out))()
            else:
                eng.sized_expand(emb.table, self.pull_recv, fb, out)

    def _push_sized(self, j, values, scale, stream_key):
        _, fb = self._slot_of[j % self._ring]
        eng, emb = self.eng, self.emb
        w, r = emb.world, emb.rank
        if w == 1:          # nobody else pushes: reduce + server add of the own keys in one launch
            if fb.n:
                if self._fast_ok:
                    key = ("salone", j % self._ring, fb.n, values.data_ptr(), scale, stream_key)
                    c = self._calls.get(key)
                    if c is None:
                        c = self._fast(key, lambda: eng.sized_push_alone_call(emb.table, fb, values, scale))
                    c()
                else:
                    eng.sized_push_alone(emb.table, fb, values, scale)
            return
        send_cnt, recv_cnt = self._counts(fb)
        if fb.n:
            if self._fast_ok:
                self._fast(("sred", j % self._ring, fb.n, values.data_ptr(), scale, stream_key),
                           lambda: eng.sized_reduce_call(fb, values, scale, self.push_buf, self.zero_flags))()
            else:
                eng.sized_reduce(fb, values, scale, self.push_buf, self.zero_flags)
        ins = [c if g != r else 0 for g, c in enumerate(send_cnt)]          # reduced rows for owner g
        outs = [c if g != r else 0 for g, c in enumerate(recv_cnt)]         # rows peer g pushes to this rank
        b0 = (w + 1) * self.rcap
        emb._a2a(self.push_buf[b0:b0 + sum(outs)], self.push_buf[:sum(ins)], outs, ins)
        self._account_sized(ins, outs, send_cnt, recv_cnt, pull=False)
        eng.sized_serve_push(emb.table, fb, r, sum(recv_cnt), self.push_buf)

    def _account_sized(self, ins, outs, send_cnt, recv_cnt, pull=True):
        """Bytes that crossed the fabric in one sized row exchange: every one of them a row some batch names."""
        st = self.emb.stats
        bo, bi = 4 * self.emb.width * sum(ins), 4 * self.emb.width * sum(outs)
        st["xgmi_bytes_out"] += bo
        st["xgmi_bytes_in"] += bi
        st["xgmi_row_bytes"] = st.get("xgmi_row_bytes", 0) + bo + bi
        st["xgmi_row_bytes_out"] = st.get("xgmi_row_bytes_out", 0) + bo
        # useful egress: every row carried is one a batch names; of the key frames, the words in use (count, flag, keys)
        st["xgmi_useful_bytes_out"] = st.get("xgmi_useful_bytes_out", 0) + bo + \
            (4 * (2 * (self.emb.world - 1) + sum(outs)) if pull else 0)
        if pull:        # how unevenly the owners are named (row-range shards of a skewed key space)
            named = sum(recv_cnt)
            st["owner_rows_sum"] = st.get("owner_rows_sum", 0) + named
            st["owner_rows_max"] = max(st.get("owner_rows_max", 0), named)
            st["owner_steps"] = st.get("owner_steps", 0) + 1

    def _fast(self, key, build):
        """Plain launches with arguments converted once per key (engines that offer it; not with graphs)."""
        c = self._calls.get(key)
        if c is None:
            if len(self._calls) >= 4096:
                self._calls.clear()
            c = self._calls[key] = build()
        return c

    def _pull_fast(self, j, out, stream_key):
        blk, fb = self._fb(j)
        eng, emb = self.eng, self.emb
        rows_in = self.pull_send if emb.world == 1 else self.pull_recv
        a, b = self._fast(("pull", j % self._ring, fb.n, out.data_ptr() if fb.n else 0, blk.kgot.data_ptr(), stream_key),
                          lambda: (eng.frames_serve_pull_call(emb.table, blk, fb, self.pull_send),
                                   eng.frames_expand_call(rows_in, fb, out) if fb.n else None))
        a()
        self._exchange(self.pull_recv, self.pull_send)
        if b is not None:
            b()

    def _push_fast(self, j, values, scale, stream_key):
        _, fb = self._fb(j)
        eng, emb = self.eng, self.emb
        rows_in = self.push_send if emb.world == 1 else self.push_recv
        a, b = self._fast(("push", j % self._ring, fb.n, values.data_ptr() if fb.n else 0, scale, stream_key),
                          lambda: (eng.frames_reduce_call(fb, values, scale, self.push_send, self.zero_flags) if fb.n else None,
                                   eng.frames_serve_push_call(emb.table, fb, rows_in)))
        if a is not None:
            a()
        self._exchange(self.push_recv, self.push_send)
        b()

    def _run(self, key, fn):
        """Enqueue `fn`: eagerly at the first use of `key` (lazy one-time initialisation -- kernel attributes, scratch
        allocations -- must not fall into a capture), captured into a hipGraph at the second, replayed from then on."""
        if not self.graphs:
            fn()
            return
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 512:
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

    def _account(self):
        emb = self.emb
        w = emb.world
        if w > 1:
            per_peer = 2 * 4 * self.rcap * emb.width     # pull rows + push rows: whole frames travel, padding included
            emb.stats["xgmi_bytes_out"] += (w - 1) * per_peer
            emb.stats["xgmi_bytes_in"] += (w - 1) * per_peer
            emb.stats["xgmi_frame_padded_bytes"] = emb.stats.get("xgmi_frame_padded_bytes", 0) + 2 * (w - 1) * per_peer

    # -- the stream protocol ----------------------------------------------------------------------------------------
    def start(self, ids_list):
        """`ids_list`: the first LOOKAHEAD batches of the stream (fewer if the stream is shorter)."""
        ids_list = list(ids_list)
        if len(ids_list) > self.LOOKAHEAD:
            raise ValueError("start takes the first %d batches" % self.LOOKAHEAD)
        for j in range(3 * self.block):
            self._stage(j, ids_list[j] if j < len(ids_list) else None)
        self.k = 0
        self._pending = False
        self._route_block(0)
        return self

    def _overflowed(self, j):
        """Host: did any rank overflow its frames for batch j?  Reads the pinned word its routing wrote (enqueued a
        block ago: the wait is for work that is long complete in steady state)."""
        blk, fb = self._fb(j)
        if not blk.synced:
            self.eng.host_sync(blk.ev)
            blk.synced = True
        return self.eng.frames_overflowed(fb)

    def pull(self, ahead_ids=None, out=None, _block_started=False):
        """Rows of the current batch k; `ahead_ids` = batch k + LOOKAHEAD (None once the stream has ended)."""
        if self.k is None:
            raise RuntimeError("FramedStep.pull before start")
        if self._pending:
            raise RuntimeError("FramedStep: pull and push alternate (push the current batch first)")
        k = self.k
        blk, fb = self._slot_of[k % self._ring]
        if not fb.routed:
            raise RuntimeError("FramedStep.pull: the stream of batches has ended")
        if k % self.block == 0 and not _block_started:
            self._block_start(k // self.block)
        self._stage(k + self.LOOKAHEAD, ahead_ids)
        if not blk.synced:          # the pinned words of the block's routing (enqueued a block ago)
            self.eng.host_sync(blk.ev)
            blk.synced = True
        self._over = over = self.eng.frames_overflowed(fb)
        width = self.emb.width
        if fb.n and out is None:
            out = self.eng.empty_rows(fb.n, width)
        if over:
            # sized exchange for this batch only (a collective: every rank saw the flag)
            self.fallbacks += 1
            self._sized = self.emb.prefetch(fb.ids)
            rows = self.emb.pull(route=self._sized, return_route=False)
            if fb.n:
                out.copy_(rows.reshape(out.shape))
        elif self.sized:
            self._pull_sized(k, out, self.eng._stream() if self._fast_ok else 0)
        else:
            if self._fast_ok and not self.graphs:
                self._pull_fast(k, out, self.eng._stream())
            else:
                self._run(("pull", k % self._ring, fb.n, out.data_ptr() if fb.n else 0), lambda: self._pull(k, out))
            self._account()
        self._pending = True
        if not fb.n:
            return None
        return out if out.shape == fb.out_shape else out.reshape(fb.out_shape)

    def push(self, values, lr=None):
        """Apply the gradients `values` of the current batch on its owners (scale -lr; 1 if lr is None)."""
        if not self._pending:
            raise RuntimeError("FramedStep: pull and push alternate (pull the current batch first)")
        k = self.k
        _, fb = self._slot_of[k % self._ring]
        scale = 1.0 if lr is None else -float(lr)
        if self._over:
            self.emb.push(None, values, lr, route=self._sized)
            self._sized = None
        else:
            v = None
            if fb.n:        # the launches read `values` as [n, width] rows
                v = values if values.dim() == 2 and values.is_contiguous() else values.reshape(-1, self.emb.width).contiguous()
            if self.sized:
                self._push_sized(k, v, scale, self.eng._stream() if self._fast_ok else 0)
            elif self._fast_ok and not self.graphs:
                self._push_fast(k, v, scale, self.eng._stream())
            else:
                self._run(("push", k % self._ring, fb.n, v.data_ptr() if fb.n else 0, scale),
                          lambda: self._push(k, v, scale))
        self._pending = False
        self.k = k + 1

    # -- the step as ONE native call (ha_shard_step / ha_shard_steps, csrc/shard.hip) ---------------------------------------
    def native_ok(self):
        """Can a step be handed to the library as one call?  Sized exchanges, the HIP engine, and either nobody to exchange
        with (world size 1) or the library's own RCCL exchange (sharded.NativeExchange)."""
        return bool(self.sized and self._fast_ok and not self.graphs and hasattr(self.eng.lib, "ha_shard_steps") and
                    (self.emb.world == 1 or getattr(self.emb, "native", None) is not None))

    def _slot_desc(self, j):
        """ha_shard_slot of ring slot j % ring (its buffers never move; n follows the batch that sits in it)."""
        from . import _lib
        i = j % self._ring
        blk, fb = self._slot_of[i]
        if not hasattr(self, "_descs"):
            self._descs = {}
        d = self._descs.get(i)
        if d is None:
            emb = self.emb
            d = _lib.ShardSlot()
            d.world, d.rank, d.rcap = emb.world, emb.rank, self.rcap
            d.plan_ws = fb.plan.ws.data_ptr()
            d.keys_fixed, d.meta_dev = fb.keys_fixed.data_ptr(), fb.meta_dev.data_ptr()
            d.posmap, d.rowmap = fb.posmap.data_ptr(), fb.rowmap.data_ptr()
            d.counts_host = fb.state_host.data_ptr()
            if emb.world > 1:
                p = self.eng._owner_plan
                m = emb.world * self.rcap
                if p is None or p.capacity < m:
                    p = self.eng._owner_plan = self.eng.ops.IndexPlan(m + 16, self.eng.device)
                d.owner_plan_ws = p.ws.data_ptr()
            self._descs[i] = d
        d.n = fb.n
        return d

    def steps(self, ahead_ids_list, values_list, lr=None, outs=None):
        """len(values_list) consecutive steps -- pull of the current batch, push of its gradients values_list[i] -- by ONE
        library call (ha_shard_steps): the reference's worker does a pull or a push inside one C++ call
        (PSAgent::vecPullSparse / vecPushSparse, PSAgent.h:124-237); here the launches and the row exchanges of a whole run
        of steps are enqueued without Python in between.  For callers that hold the gradients of a batch when they ask for its
        rows (a benchmark loop, a pipeline with staleness); the steps must lie in ONE routing block.  ahead_ids_list[i] = batch
        k + i + LOOKAHEAD (None entries once the stream has ended).  Falls back to pull / push step by step where the native
        step is not available or a batch of the run overflowed its key frames.  Returns the rows of the batches."""
        import ctypes
        if self.k is None:
            raise RuntimeError("FramedStep.steps before start")
        if self._pending:
            raise RuntimeError("FramedStep: push the current batch first")
        cnt, k0 = len(values_list), self.k
        ahead_ids_list = list(ahead_ids_list) if ahead_ids_list is not None else [None] * cnt
        if cnt == 0:
            return []
        if k0 // self.block != (k0 + cnt - 1) // self.block:
            raise ValueError("steps %d..%d cross a routing block boundary (block = %d)" % (k0, k0 + cnt - 1, self.block))
        eng, emb, width = self.eng, self.emb, self.emb.width
        fbs = [self._slot_of[(k0 + i) % self._ring][1] for i in range(cnt)]
        if outs is None:
            outs = [eng.empty_rows(fb.n, width) if fb.n else None for fb in fbs]
        native = native_tried = self.native_ok() and all(fb.routed for fb in fbs)
        # The host's work at a block boundary used to sit between two blocks' steps with the device idle: the routing of block
        # b + 1 was enqueued (and, the device having nothing else, executed) BEFORE the steps of block b -- 95 us of a 290 us block
        # at configs[1]'s shape (kernel trace, docs/EXPERIMENTS.md round 6 section 13).  Now the block's steps go first; the
        # routing follows them onto the device and waits only for what it needs of the step stream: the previous block's end
        # (the ids it reads were staged by then).
        route_after = None
        if native:
            if k0 % self.block == 0:
                self.eng.wait_event(self.blocks[(k0 // self.block) % 3].ev)
                route_after = getattr(self, "_ev_block_end", None)
                if route_after is None or not self.side:
                    self._route_block(k0 // self.block + 1)          # (the first block, or no side stream: as before)
                    route_after = None
                else:
                    route_after = (k0 // self.block + 1, route_after)
            blk = self._slot_of[k0 % self._ring][0]
            if not blk.synced:
                eng.host_sync(blk.ev)
                blk.synced = True
            native = not any(eng.frames_overflowed(fb) for fb in fbs)
        started = native_tried and k0 % self.block == 0       # (the fallback below must not start the block again)
        if not native and route_after is not None:
            self._route_block(route_after[0])        # (the fallback pulls and pushes step by step: the routing in its old place)
            route_after = None
        if not native:
            res = []
            for i in range(cnt):
                res.append(self.pull(ahead_ids_list[i], out=outs[i], _block_started=started and i == 0))
                self.push(values_list[i], lr)
            return res
        for i in range(cnt):
            self._stage(k0 + i + self.LOOKAHEAD, ahead_ids_list[i])
        scale = 1.0 if lr is None else -float(lr)
        vp = ctypes.c_void_p
        descs = [self._slot_desc(k0 + i) for i in range(cnt)]
        vals = []
        for i, fb in enumerate(fbs):
            v = values_list[i]
            if fb.n:
                v = v if v.dim() == 2 and v.is_contiguous() else v.reshape(-1, width).contiguous()
            vals.append(v)
        sl = (ctypes.POINTER(type(descs[0])) * cnt)(*[ctypes.pointer(d) for d in descs])
        op = (vp * cnt)(*[o.data_ptr() if (o is not None and fb.n) else None for o, fb in zip(outs, fbs)])
        gp = (vp * cnt)(*[v.data_ptr() if fb.n else None for v, fb in zip(vals, fbs)])
        xh = vp(emb.native.h) if emb.world > 1 else None
        rc = eng.lib.ha_shard_steps(vp(emb.table.data_ptr()), emb.table.shape[0], width, cnt, sl, xh,
                                    vp(self.pull_send.data_ptr()), vp(self.pull_recv.data_ptr()), self.pull_recv.shape[0],
                                    vp(self.push_buf.data_ptr()), self.push_buf.shape[0], vp(self.zero_flags.data_ptr()), op, gp,
                                    ctypes.c_float(scale), vp(eng._stream()))
        if rc != 0:
            eng.check(-1, "ha_shard_steps")
        if route_after is not None:
            self._route_block(route_after[0], after=route_after[1])
        if (k0 + cnt) % self.block == 0:           # the block's last step is enqueued: what its successor's successor's routing waits for
            self._ev_block_end = eng.record(getattr(self, "_ev_block_end_obj", None))
            self._ev_block_end_obj = self._ev_block_end
        if emb.world > 1:
            r = emb.rank
            for fb in fbs:
                send_cnt, recv_cnt = self._counts(fb)
                self._account_sized([c if g != r else 0 for g, c in enumerate(recv_cnt)],
                                    [c if g != r else 0 for g, c in enumerate(send_cnt)], send_cnt, recv_cnt)
                self._account_sized([c if g != r else 0 for g, c in enumerate(send_cnt)],
                                    [c if g != r else 0 for g, c in enumerate(recv_cnt)], send_cnt, recv_cnt, pull=False)
        self.k = k0 + cnt
        self._keep = (vals, outs)        # (the launches read them: alive until the next call)
        return [None if not fb.n else (o if o.shape == fb.out_shape else o.reshape(fb.out_shape)) for o, fb in zip(outs, fbs)]

    # -- measurement aid (bench.py's N>1 leg): the five launches of a step one by one -----------------------------------
    def kernel_times(self, values, lr, reps=30):
        """Average duration (us, HIP events on the current stream, `reps` back-to-back launches) of each launch of the
        CURRENT batch's pull and push, without the exchanges.  Call between a pull and its push; the push launches do
        apply `values` to the table `reps` times (synthetic benchmarks only)."""
        if not self._pending:
            raise RuntimeError("kernel_times: call after pull, before push")
        k = self.k
        blk, fb = self._fb(k)
        eng, emb = self.eng, self.emb
        out = eng.empty_rows(max(fb.n, 1), emb.width)
        v = values.reshape(-1, emb.width)
        scale = 1.0 if lr is None else -float(lr)
        if self.sized:
            r, total = emb.rank, sum(self._counts(fb)[1])
            calls = []
            if emb.world > 1:
                calls.append(("serve_pull (owner gather of the other ranks' keys, compact)",
                              lambda: eng.sized_serve_pull(emb.table, fb, r, self.pull_send)))
            calls.append(("expand (own keys from the shard, the others from the received rows -> positions)",
                          lambda: eng.sized_expand(emb.table, self.pull_recv, fb, out)))
            if emb.world > 1:
                calls.append(("reduce (gradients -> push buffer, own keys apart)",
                              lambda: eng.sized_reduce(fb, v, scale, self.push_buf, self.zero_flags)))
                calls.append(("serve_push (merge of the received lists + rank-ordered apply)",
                              lambda: eng.sized_serve_push(emb.table, fb, r, total, self.push_buf)))
            else:
                calls.append(("push_alone (reduce + server add of the own keys, one launch)",
                              lambda: eng.sized_push_alone(emb.table, fb, v, scale)))
            return self._time_calls(calls, reps)
        rows_in = self.pull_send if emb.world == 1 else self.pull_recv
        grads_in = self.push_send if emb.world == 1 else self.push_recv
        calls = (("serve_pull (owner gather into row frames)", lambda: eng.frames_serve_pull(emb.table, blk, fb, self.pull_send)),
                 ("expand (row frames -> positions)", lambda: eng.frames_expand(rows_in, fb, out)),
                 ("reduce (gradients -> push frames)", lambda: eng.frames_reduce(fb, v, scale, self.push_send, self.zero_flags)),
                 ("serve_push (sort received keys + rank-ordered apply)", lambda: eng.frames_serve_push(emb.table, fb, grads_in)))
        return self._time_calls(calls, reps)

    @staticmethod
    def _time_calls(calls, reps):
        res = {}
        for name, fn in calls:
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            b.synchronize()
            res[name] = a.elapsed_time(b) * 1e3 / reps
        return res
