"""Test double for herald_amd.sharded's engine: the same interface as HipEngine, computed by the CPU
oracle on CPU tensors, so that the multi-rank exchange logic can run under gloo without a GPU.
Lives in tests/ only -- the product has no CPU path."""
import contextlib

import numpy as np
import torch

from oracle import cpu


class _Plan:
    def __init__(self, ids):
        keys = cpu.ids_to_keys(ids.numpy().reshape(-1)) if ids.dtype == torch.float32 \
            else ids.numpy().reshape(-1).astype(np.uint64)
        self.uniq, self.inv, self.cnt = cpu.unique(keys)
        self.n = keys.size

    def inverse(self):
        return torch.from_numpy(self.inv.astype(np.int32))


class CpuEngine:
    NSLOT = 3

    def __init__(self):
        self.device = torch.device("cpu")

    # -- routing (same contract as HipEngine.route_issue; streams and events are no-ops on the CPU) --
    def route_issue(self, ids, starts, slot, cap):
        plan = _Plan(ids)
        assert plan.n <= cap
        st = np.asarray(starts, dtype=np.uint64)
        offsets = np.searchsorted(plan.uniq, st, side="left").astype(np.int64)
        offsets[-1] = plan.uniq.size
        owner = np.searchsorted(st, plan.uniq, side="right") - 1
        local = (plan.uniq - st[owner]).astype(np.int32)
        w = len(starts) - 1
        send = np.full((w, 1 + cap), -7, dtype=np.int32)        # padding is never read
        for g in range(w):
            c = int(offsets[g + 1] - offsets[g])
            send[g, 0] = c
            send[g, 1:1 + c] = local[offsets[g]:offsets[g + 1]]
        meta = np.concatenate([[plan.uniq.size], offsets[1:] - offsets[:-1]]).astype(np.int64)
        from herald_amd.sharded import RouteBuffers
        b = RouteBuffers()
        b.plan, b.cap = plan, cap
        b.send = torch.from_numpy(send)
        b.recv = torch.empty((w, 1 + cap), dtype=torch.int32)
        b.meta_all = torch.cat([torch.from_numpy(meta), torch.zeros(w, dtype=torch.int64)])
        b.meta, b.recv_cnt = b.meta_all[:1 + w], b.meta_all[1 + w:]
        b.host = torch.empty(1 + 2 * w, dtype=torch.int64)
        b.keys_recv = torch.empty(max(w * cap, 1), dtype=torch.int32)
        return b

    def route_unpack(self, b):
        r = b.recv.numpy()
        at = 0
        for g in range(r.shape[0]):
            c = int(r[g, 0])
            b.recv_cnt[g] = c
            b.keys_recv[at:at + c] = torch.from_numpy(r[g, 1:1 + c].copy())
            at += c

    def on_side(self, after_current=True):
        return contextlib.nullcontext()

    def record(self, ev=None):
        return None

    def wait_event(self, ev):
        pass

    def to_host(self, b):
        b.host.copy_(b.meta_all)
        return None

    def host_sync(self, ev):
        pass

    def gather_keys(self, table, keys_i32, scratch=None):
        return torch.from_numpy(table.numpy()[keys_i32.numpy().astype(np.int64)])

    def rows_buffer(self, name, rows, width):
        return torch.empty((rows, width), dtype=torch.float32)

    def expand(self, rows, plan):
        return torch.from_numpy(rows.numpy()[plan.inv])

    def reduce_scaled(self, plan, values, scale):
        v = (values.numpy() * np.float32(scale)).astype(np.float32)
        red = np.zeros((max(plan.n, 1), v.shape[1]), dtype=np.float32)
        for i, u in enumerate(plan.inv):
            red[u] += v[i]
        return torch.from_numpy(red)

    def acc_apply(self, table, keys_i32, values):
        t = table.numpy()
        v = values.numpy()
        for j, k in enumerate(keys_i32.numpy().astype(np.int64)):
            t[k] = t[k] + v[j]

    # -- fixed frames (herald_amd.sharded.FramedStep; same contract as HipEngine.frames_*) ---------------------------
    def frames_block(self, w, rcap, n_cap, block):
        from herald_amd.sharded import FrameBlock, FrameBuffers
        blk = FrameBlock()
        blk.ksend = torch.full((w, block, 2 + rcap), -99, dtype=torch.int32)
        blk.krecv = torch.full((w, block, 2 + rcap), -99, dtype=torch.int32)
        blk.kgot, blk.ev, blk.ev_obj, blk.live, blk.slots, blk.synced = None, None, None, False, [], False
        for i in range(block):
            fb = FrameBuffers()
            fb.i, fb.w, fb.rcap = i, w, rcap
            fb.keys_fixed = torch.zeros(w * rcap, dtype=torch.int32)
            fb.state_host = torch.zeros(2 + 2 * w, dtype=torch.int32)
            fb.send_cnt = fb.recv_cnt = None
            fb.n, fb.ids, fb.shape, fb.routed = 0, None, (0,), False
            blk.slots.append(fb)
        return blk

    def frames_plan(self, fb, starts):
        fb.plan = _Plan(fb.ids if fb.n else torch.zeros(0))

    def frames_pack(self, blk, fb, starts, sized_rank=None):
        plan = fb.plan
        st = np.asarray(starts, dtype=np.uint64)
        off = np.searchsorted(plan.uniq, st, side="left").astype(np.int64)
        off[-1] = plan.uniq.size
        cnt = off[1:] - off[:-1]
        send = np.full((fb.w, 2 + fb.rcap), -1, dtype=np.int32)
        send[:, 0] = cnt
        send[:, 1] = int((cnt > fb.rcap).any())
        rowmap = np.full(plan.uniq.size, -1, dtype=np.int64)
        for g in range(fb.w):
            c = int(min(cnt[g], fb.rcap))
            send[g, 2:2 + c] = (plan.uniq[off[g]:off[g] + c] - st[g]).astype(np.int32)
            rowmap[off[g]:off[g] + c] = g * fb.rcap + np.arange(c)
        blk.ksend[:, fb.i, :] = torch.from_numpy(send)
        fb.rowmap = rowmap
        fb.posmap = rowmap[plan.inv] if plan.n else np.zeros(0, np.int64)
        if sized_rank is not None:
            # sized frames (csrc/shard.hip): compact rank-ordered lists, own keys in a region of their own / read locally
            r, U = int(sized_rank), plan.uniq.size
            own0, own1 = int(off[r]), int(off[r + 1])
            u = np.arange(U)
            over = bool((cnt > fb.rcap).any())
            rm = np.where(u < own0, u, np.where(u < own1, fb.w * fb.rcap + (u - own0), u - (own1 - own0)))
            fb.rowmap = np.full(U, -1, np.int64) if over else rm
            local = (u >= own0) & (u < own1)
            fb.pos_local = (local[plan.inv] if plan.n else np.zeros(0, bool))
            key_local = (plan.uniq.astype(np.int64) - int(st[r]))
            idx = np.where(local, key_local, np.where(u < own0, u, u - (own1 - own0)))
            fb.posmap = np.full(plan.n, -1, np.int64) if over else (idx[plan.inv] if plan.n else np.zeros(0, np.int64))
            fb.state_host[2:2 + fb.w] = torch.from_numpy(cnt.astype(np.int32))

    def frames_unpack(self, blk, fb):
        r = blk.kgot[:, fb.i, :].numpy()
        keys = np.full(fb.w * fb.rcap, -1, dtype=np.int32)
        over = 0
        for g in range(fb.w):
            c = int(r[g, 0])
            over |= int(r[g, 1] != 0 or c > fb.rcap)
            c = min(c, fb.rcap)
            keys[g * fb.rcap:g * fb.rcap + c] = r[g, 2:2 + c]
        fb.keys_fixed.copy_(torch.from_numpy(keys))
        fb.state_host[0] = over
        got = np.minimum(np.maximum(r[:, 0], 0), fb.rcap).astype(np.int32)
        fb.state_host[1] = int(got.sum())
        fb.state_host[2 + fb.w:2 + 2 * fb.w] = torch.from_numpy(got)

    def frames_overflowed(self, fb):
        return bool(fb.state_host[0].item())

    def frames_serve_pull(self, table, blk, fb, rows_send):
        self.frames_unpack(blk, fb)            # the HIP kernel rewrites keys_fixed / state as well
        t, k = table.numpy(), fb.keys_fixed.numpy()
        live = k >= 0
        rows_send.numpy()[live] = t[k[live].astype(np.int64)]      # unused slots are not written

    def frames_expand(self, rows_recv, fb, out):
        if fb.n:
            r = rows_recv.numpy()
            o = out.numpy().reshape(fb.n, -1)
            ok = fb.posmap >= 0
            o[:] = 0
            o[ok] = r[fb.posmap[ok]]

    def frames_reduce(self, fb, values, scale, rows_send, zero_flags):
        if not fb.n:
            return
        v = (values.numpy().reshape(fb.n, -1) * np.float32(scale)).astype(np.float32)
        red = np.zeros((fb.plan.uniq.size, v.shape[1]), dtype=np.float32)
        for i, u in enumerate(fb.plan.inv):
            red[u] += v[i]
        ok = fb.rowmap >= 0
        rows_send.numpy()[fb.rowmap[ok]] = red[ok]

    def frames_serve_push(self, table, fb, rows_recv):
        t, v = table.numpy(), rows_recv.numpy()
        for p, k in enumerate(fb.keys_fixed.numpy().astype(np.int64)):
            if k >= 0:
                t[k] = t[k] + v[p]

    # -- sized frames (same contract as HipEngine.sized_*) ---------------------------------------------------------------
    def frames_counts(self, fb):
        s = fb.state_host.tolist()
        return s[2:2 + fb.w], s[2 + fb.w:2 + 2 * fb.w]

    @staticmethod
    def _roff(recv_cnt, rank, g):
        return sum(c for t, c in enumerate(recv_cnt[:g]) if t != rank)

    def sized_serve_pull(self, table, fb, rank, rows_send):
        _, recv_cnt = self.frames_counts(fb)
        t, k, o = table.numpy(), fb.keys_fixed.numpy().astype(np.int64), rows_send.numpy()
        for g in range(fb.w):
            if g == rank:
                continue
            at = self._roff(recv_cnt, rank, g)
            for j in range(recv_cnt[g]):
                o[at + j] = t[k[g * fb.rcap + j]]

    def sized_expand(self, table, rows_recv, fb, out):
        if fb.n:
            o = out.numpy().reshape(fb.n, -1)
            t, r = table.numpy(), rows_recv.numpy()
            for i in range(fb.n):
                m = int(fb.posmap[i])
                o[i] = 0 if m < 0 else (t[m] if fb.pos_local[i] else r[m])

    def sized_reduce(self, fb, values, scale, push_buf, zero_flags):
        if not fb.n:
            return
        v = (values.numpy().reshape(fb.n, -1) * np.float32(scale)).astype(np.float32)
        red = np.zeros((fb.plan.uniq.size, v.shape[1]), dtype=np.float32)
        for i, u in enumerate(fb.plan.inv):
            red[u] += v[i]
        ok = fb.rowmap >= 0
        push_buf.numpy()[fb.rowmap[ok]] = red[ok]

    def sized_serve_push(self, table, fb, rank, total, push_buf):
        _, recv_cnt = self.frames_counts(fb)
        assert total == sum(recv_cnt)
        t, v, k = table.numpy(), push_buf.numpy(), fb.keys_fixed.numpy().astype(np.int64)
        s0, b0 = fb.w * fb.rcap, (fb.w + 1) * fb.rcap
        for g in range(fb.w):                                     # rank order
            at = s0 if g == rank else b0 + self._roff(recv_cnt, rank, g)
            for j in range(recv_cnt[g]):
                key = k[g * fb.rcap + j]
                t[key] = t[key] + v[at + j]

    def sized_push_alone(self, table, fb, values, scale):
        if not fb.n:
            return
        v = (values.numpy().reshape(fb.n, -1) * np.float32(scale)).astype(np.float32)
        red = np.zeros((fb.plan.uniq.size, v.shape[1]), dtype=np.float32)
        for i, u in enumerate(fb.plan.inv):
            red[u] += v[i]
        t = table.numpy()
        for u, key in enumerate(fb.plan.uniq.astype(np.int64)):
            t[key] = t[key] + red[u]

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def empty_rows(self, rows, width):
        return torch.full((rows, width), float("nan"), dtype=torch.float32)      # stale frame slots must never be used
