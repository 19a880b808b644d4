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
    def route_issue(self, ids, starts, slot):
        plan = _Plan(ids)
        st = np.asarray(starts, dtype=np.uint64)
        offsets = np.searchsorted(plan.uniq, st, side="left").astype(np.int64)
        offsets[-1] = plan.uniq.size
        owner = np.searchsorted(st, plan.uniq, side="right") - 1
        local = np.zeros(max(plan.n, 1), dtype=np.int32)
        local[:plan.uniq.size] = (plan.uniq - st[owner]).astype(np.int32)
        w = len(starts) - 1
        meta = np.concatenate([[plan.uniq.size], offsets[1:] - offsets[:-1]]).astype(np.int64)
        from herald_amd.sharded import RouteBuffers
        b = RouteBuffers()
        b.plan, b.local, b.meta = plan, torch.from_numpy(local), torch.from_numpy(meta)
        b.meta_all = torch.cat([b.meta, torch.zeros(w, dtype=torch.int64)])
        b.meta, b.recv = b.meta_all[:1 + w], b.meta_all[1 + w:]
        b.host = torch.empty(1 + 2 * w, dtype=torch.int64)
        b.keys_recv = torch.empty(max(w * plan.n, 1), dtype=torch.int32)
        return b

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
