"""Test double for herald_amd.sharded's engine: the same interface as HipEngine, computed by the CPU
oracle on CPU tensors, so that the multi-rank exchange logic can run under gloo without a GPU.
Lives in tests/ only -- the product has no CPU path."""
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
    def __init__(self):
        self.device = torch.device("cpu")

    def plan(self, ids, slot="batch"):
        return _Plan(ids)

    def bucket(self, plan, starts):
        st = np.asarray(starts, dtype=np.uint64)
        offsets = np.searchsorted(plan.uniq, st, side="left").astype(np.int32)
        offsets[-1] = plan.uniq.size
        owner = np.searchsorted(st, plan.uniq, side="right") - 1
        local = np.zeros(max(plan.n, 1), dtype=np.int32)
        local[:plan.uniq.size] = (plan.uniq - st[owner]).astype(np.int32)
        return torch.from_numpy(offsets), torch.from_numpy(local)

    def gather_keys(self, table, keys_i32):
        return torch.from_numpy(table.numpy()[keys_i32.numpy().astype(np.int64)])

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

    def n_unique_and(self, plan, *tensors):
        return int(plan.uniq.size), [t.reshape(-1).tolist() for t in tensors]
