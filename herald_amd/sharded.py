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

Here every rank owns one shard in its HBM and is a worker at the same time; the exchange is two
all-to-alls per direction over RCCL (torch.distributed "nccl" backend == RCCL over xGMI):
    pull:  counts  ->  shard-local keys  ->  [owner gathers]  ->  rows back
    push:  counts  ->  keys + worker-reduced rows  ->  [owner applies, rank order]
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


class HipEngine:
    """Device compute of the sharded store through the C-ABI (herald_amd.ops)."""

    def __init__(self, device):
        from . import _lib, ops
        self.ops = ops
        self.lib = _lib.load()
        self.device = torch.device(device)
        self._plans = {}

    def _plan(self, slot, n):
        p = self._plans.get(slot)
        if p is None or p.capacity < n:
            p = self.ops.IndexPlan(max(n, 1), self.device)
            self._plans[slot] = p
        return p

    def plan(self, ids, slot="batch"):
        """Sorted-unique plan of a batch of ids (float32 or int64)."""
        return self._plan(slot, ids.numel()).build(ids.reshape(-1))

    def bucket(self, plan, starts):
        """-> (offsets int32[W+1] device, local_keys int32[n] device); no host sync."""
        w = len(starts) - 1
        offsets = torch.empty(w + 1, dtype=torch.int32, device=self.device)
        local = torch.empty(max(plan.n, 1), dtype=torch.int32, device=self.device)
        st = (ctypes.c_int64 * (w + 1))(*starts)
        from ._lib import check
        check(self.lib.ha_shard_bucket(ctypes.c_void_p(plan.ws.data_ptr()), plan.n, st, w,
                                       ctypes.c_void_p(offsets.data_ptr()), ctypes.c_void_p(local.data_ptr()),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "ha_shard_bucket")
        return offsets, local

    def gather_keys(self, table, keys_i32):
        """rows[j,:] = table[keys[j],:] for shard-local uint32 keys held in an int32 tensor."""
        from ._lib import check
        n = keys_i32.numel()
        out = torch.empty((n, table.shape[1]), dtype=torch.float32, device=self.device)
        check(self.lib.ha_gather_u32keys(ctypes.c_void_p(table.data_ptr()), table.shape[0], table.shape[1],
                                         ctypes.c_void_p(keys_i32.data_ptr()), n,
                                         ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "ha_gather_u32keys")
        return out

    def expand(self, rows, plan):
        """out[i,:] = rows[inverse[i],:] -- every position receives its unique row (sparse.h:17-31)."""
        return self.gather_keys(rows, plan.inverse())

    def reduce_scaled(self, plan, values, scale):
        return self.ops.dedup_reduce(plan, values, scale=scale)

    def acc_apply(self, table, keys_i32, values):
        """table[key,:] = (table[key,:] + v_a) + v_b ... in the order the (key, value) pairs are listed."""
        from ._lib import check
        n = keys_i32.numel()
        if n == 0:
            return
        p = self._plan("owner", n)
        check(self.lib.ha_plan_build_u32keys(ctypes.c_void_p(keys_i32.data_ptr()), n,
                                             ctypes.c_void_p(p.ws.data_ptr()), 32,
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "ha_plan_build_u32keys")
        p.n = n
        p._view = None
        self.ops.sgd_apply(table, p, values, -1.0)

    def n_unique_and(self, plan, *tensors):
        """One host sync: (n_unique, [tensor.tolist() ...])."""
        packed = torch.cat([plan.n_unique_dev().to(torch.int64)] + [t.reshape(-1).to(torch.int64) for t in tensors])
        host = packed.tolist()
        out, k = [], 1
        for t in tensors:
            out.append(host[k:k + t.numel()])
            k += t.numel()
        return host[0], out


class ShardedEmbedding:
    """One row-range shard per rank + all-to-all pull/push.  `table` is this rank's shard."""

    def __init__(self, rows, width, device, group=None, engine=None, table=None):
        self.group = group
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

    # -- exchange plumbing ---------------------------------------------------------------------------
    def _a2a(self, out, inp, out_splits, in_splits):
        if self.world == 1:
            out.copy_(inp)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        return out

    def _route(self, plan):
        """Counts all-to-all.  -> (U, send_counts, recv_counts, local_keys) with host-side counts.
        The routing of a plan is cached on the plan object: pull and push of the same batch (the
        training step) pay for one counts exchange and one host synchronisation."""
        cached = getattr(plan, "_route_cache", None)
        if cached is not None and cached[0] is self:
            return cached[1]
        r = self._route_uncached(plan)
        try:
            plan._route_cache = (self, r)
        except AttributeError:
            pass
        return r

    def _route_uncached(self, plan):
        eng = self.engine
        offsets, local = eng.bucket(plan, self.starts)
        send_cnt_dev = (offsets[1:] - offsets[:-1]).to(torch.int64)
        recv_cnt_dev = torch.empty_like(send_cnt_dev)
        self._a2a(recv_cnt_dev, send_cnt_dev, None, None)
        u, (send_cnt, recv_cnt) = eng.n_unique_and(plan, send_cnt_dev, recv_cnt_dev)
        # the shard-local keys every owner will be asked for (pull) / handed rows for (push)
        keys_send = local[:u].contiguous()
        keys_recv = torch.empty(sum(recv_cnt), dtype=keys_send.dtype, device=self.device)
        self._a2a(keys_recv, keys_send, recv_cnt, send_cnt)
        return u, send_cnt, recv_cnt, keys_recv

    def _account(self, send_cnt, recv_cnt, bytes_per_key_out, bytes_per_key_in):
        r = self.rank
        out_keys = sum(c for g, c in enumerate(send_cnt) if g != r)
        in_keys = sum(c for g, c in enumerate(recv_cnt) if g != r)
        self.stats["xgmi_bytes_out"] += out_keys * bytes_per_key_out + in_keys * bytes_per_key_in
        self.stats["xgmi_bytes_in"] += in_keys * bytes_per_key_out + out_keys * bytes_per_key_in

    # -- SparsePull -------------------------------------------------------------------------------------
    def pull(self, ids, plan=None, return_plan=False):
        """out[i,:] = table_global[ids[i],:] for this rank's batch of ids."""
        eng = self.engine
        if plan is None:
            plan = eng.plan(ids)
        u, send_cnt, recv_cnt, keys_recv = self._route(plan)
        rows_send = eng.gather_keys(self.table, keys_recv)
        rows_recv = torch.empty((u, self.width), dtype=torch.float32, device=self.device)
        self._a2a(rows_recv, rows_send, send_cnt, recv_cnt)
        out = eng.expand(rows_recv, plan)
        self._account(send_cnt, recv_cnt, 4, 4 * self.width)
        out = out.reshape(tuple(ids.shape) + (self.width,))
        return (out, plan) if return_plan else out

    # -- SparsePush -------------------------------------------------------------------------------------
    def push(self, ids, values, lr=None, plan=None):
        """table_global[id,:] += sum over this rank's positions of (-lr * values)  (scale 1 if lr is None),
        pushes of different ranks applied in rank order."""
        eng = self.engine
        if plan is None:
            plan = eng.plan(ids)
        scale = 1.0 if lr is None else -float(lr)
        reduced = eng.reduce_scaled(plan, values.reshape(-1, self.width), scale)
        u, send_cnt, recv_cnt, keys_recv = self._route(plan)
        nrecv = sum(recv_cnt)
        rows_send = reduced[:u].contiguous()
        rows_recv = torch.empty((nrecv, self.width), dtype=torch.float32, device=self.device)
        self._a2a(rows_recv, rows_send, recv_cnt, send_cnt)
        eng.acc_apply(self.table, keys_recv, rows_recv)
        self._account(send_cnt, recv_cnt, 4 * self.width, 0)

    # -- SSPushPull (push this batch, pull the next one): ParameterServerCommunicate.py:74-76 ------------
    def push_pull(self, push_ids, values, lr, pull_ids):
        self.push(push_ids, values, lr)
        return self.pull(pull_ids)

    # -- checkpoint format of the reference: raw fp32 `<name>_<part>.dat` per shard ------------------------
    def save(self, path_prefix):
        """PSAgent ParamSave (PSAgent.h:447-476, PSFHandle.h:401-439): raw little-endian fp32 rows."""
        self.table.detach().cpu().numpy().tofile("%s_%d.dat" % (path_prefix, self.rank))

    def load(self, path_prefix):
        import numpy as np
        a = np.fromfile("%s_%d.dat" % (path_prefix, self.rank), dtype=np.float32)
        self.table.copy_(torch.from_numpy(a.reshape(self.local_rows, self.width)).to(self.device))
