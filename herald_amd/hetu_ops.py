"""Operator-level mirror of the reference's embedding ops: which native call is made,
when, and on which buffers -- without the graph executor around them.

Mirrors (paths relative to /root/reference):
  EmbeddingLookUp / EmbeddingLookUp_Gradient      python/hetu/gpu_ops/EmbeddingLookUp.py:10-125
  ParameterServerCommunicateOp                    python/hetu/gpu_ops/ParameterServerCommunicate.py:12-250
  ParameterServerSparsePullOp                     python/hetu/gpu_ops/ParameterServerCommunicate.py:254-306
  SGD sparse dispatch of OptimizerOp              python/hetu/gpu_links/OptimizerLink.py:23-33

`Config` carries the HetuConfig fields those ops read (executor.py:162-182): comm_mode, bsp, prefetch,
cstable_policy, cache_bound, cache_limit, use_sparse_pull.  The data loader contract is the reference's
`get_arr` / `get_next_arr` (python/hetu/dataloader.py:63-98): `next_ids()` returns the ids of the batch
after the current one.  Everything computed goes through libherald_amd.so.
"""
import ctypes

import torch

from . import _lib, cache as hcache, ops
from ._lib import check
from .sharded import ShardedEmbedding


class Config:
    def __init__(self, comm_mode=None, bsp=0, prefetch=True, cstable_policy=None, cache_bound=100, cache_limit=0,
                 use_sparse_pull=True, cache_perf_enable=False, cache_plan_ahead=False):
        self.comm_mode, self.bsp, self.prefetch = comm_mode, bsp, prefetch
        # not a HetuConfig field: the cache's PLANNED flow (csrc/cache_block.hip) for the bsp-prefetch schedule -- the bookkeeping
        # of batch k + 1 runs on a side stream beside the model's step on batch k.  Needs ids one batch further ahead than
        # get_next_arr gives them: ParameterServerCommunicateOp(..., peek_ids=...); the reference's loader ring is three batches
        # deep (python/hetu/dataloader.py:63-98), so it has them.
        self.cache_plan_ahead = cache_plan_ahead
        self.cstable_policy, self.cache_bound, self.cache_limit = cstable_policy, cache_bound, cache_limit
        self.use_sparse_pull = use_sparse_pull
        self.cache_perf_enable = cache_perf_enable        # executor.py: cache_perf_enable (run_hetu.py:508-515 dumps the dicts)
        self.ps_map = {}


class EmbeddingParameter:
    """The embedding Variable node: a device table (comm None/AllReduce) or a PS-resident table reached
    through a ShardedEmbedding and, optionally, a cache."""
    _next_id = 0

    def __init__(self, table=None, store=None):
        self.id = EmbeddingParameter._next_id
        EmbeddingParameter._next_id += 1
        self.table, self.store = table, store
        self.is_embed = True
        self.cache = None
        self.shape = tuple(table.shape) if table is not None else (store.rows, store.width)


def _same_tensor(a, b):
    return a is not None and a.data_ptr() == b.data_ptr() and a.numel() == b.numel() and a.dtype == b.dtype


def scale_(values, factor, stream=None):
    """values *= factor on the device (one rounding per element)."""
    check(_lib.load().ha_scale_f32(ctypes.c_void_p(values.data_ptr()), values.numel(), ctypes.c_float(factor),
                                   ops._stream_ptr(stream)), "ha_scale_f32")
    return values


class EmbeddingLookUp:
    """embedding_lookup_op(embedding, index, enable_push_index)."""

    def __init__(self, embedding, enable_push_index=False):
        self.embedding, self.enable_push_index = embedding, enable_push_index

    def forward_hook(self, config):                                   # EmbeddingLookUp.py:56-75
        self.config = config
        if config.use_sparse_pull and config.comm_mode in ("PS", "Hybrid") or config.cstable_policy:
            if config.prefetch:
                self.compute = self._compute_prefetched
            elif config.cstable_policy:
                self.compute = self._compute_sparsepull_from_cache
            else:
                self.compute = self._compute_sparsepull_from_ps
        else:
            self.compute = self._compute_gpu

    def _compute_gpu(self, ids, output_val, stream=None):              # :24-26
        return ops.embedding_lookup(self.embedding.table, ids, out=output_val, stream=stream)

    def _compute_sparsepull_from_ps(self, ids, output_val, stream=None):   # :28-35
        output_val.copy_(self.embedding.store.pull(ids))
        return output_val

    def _compute_sparsepull_from_cache(self, ids, output_val, stream=None):  # :37-42
        self.embedding.cache.embedding_lookup(ids, output_val).wait()
        return output_val

    def _compute_prefetched(self, ids, output_val, stream=None):
        """With prefetch the op is not computed: its output is the buffer the communicate op filled for
        this batch during the previous step (executor.py:625,883-885)."""
        ev, buf = self.config.ps_map[self.embedding]
        if ev is not None:
            ev.wait()
        output_val.copy_(buf)
        return output_val


class EmbeddingLookUp_Gradient:
    def __init__(self, embed_shape, enable_push_index=False):
        self.embed_shape, self.enable_push_index = embed_shape, enable_push_index

    def compute(self, vectors, index):                                 # EmbeddingLookUp.py:95-111
        if isinstance(index, tuple):
            push = index[1] if self.enable_push_index else None
            return ops.IndexedSlices(indices=index[0], values=vectors, dense_shape=self.embed_shape,
                                     push_indices=push)
        if self.enable_push_index:
            raise TypeError
        return ops.IndexedSlices(indices=index, values=vectors, dense_shape=self.embed_shape)


def sgd_update_sparse(param, grad, lr, stream=None):
    """OptimizerOp's sparse SGD branch on a device table (OptimizerLink.py:23-33): no dedup, duplicates in
    occurrence order."""
    ops.dl_call("SGDOptimizerSparseUpdate",
                [param.table, grad.indices.contiguous(), grad.values.reshape(-1, param.table.shape[1]).contiguous()],
                scalars=[ctypes.c_float(lr)], stream=stream)


class ParameterServerCommunicateOp:
    def __init__(self, parameter, learning_rate, next_ids, peek_ids=None):
        """peek_ids(j) (optional, Config.cache_plan_ahead): the ids of the batch j batches after the one next_ids() returns
        (peek_ids(0) = that batch itself), without advancing the loader; None when there is none."""
        self.parameter = parameter
        self.learning_rate = -learning_rate                           # :24
        self.next_ids = next_ids
        self.peek_ids = peek_ids
        self._peek_offset = 1
        self._planned = None          # the planned flow: ids tensors of the planned batches, oldest first

    def forward_hook(self, config, first_ids=None, barrier=lambda: None):   # :130-242
        self.config, self.barrier = config, barrier
        p = self.parameter
        self.use_cache_table = config.cstable_policy is not None and p.is_embed
        if self.use_cache_table:
            width = p.shape[1]
            store = p.store
            if store.world > 1:
                # the table is sharded over the ranks: the cache talks to the owners through the inbox /
                # outbox exchange (remote_store.ShardedStore) -- only misses, stale lines and pushed lines
                # cross the fabric (PSAgent.h:537-627 / PSFhandle_embedding.cc:5-79)
                from . import remote_store
                versions = torch.zeros(store.local_rows, dtype=torch.int64, device=store.table.device)
                torch.cuda.current_stream(versions.device).synchronize()
                rstore = remote_store.ShardedStore(p.shape[0], width, store.table.device,
                                                   remote_store.LocalStore(store.table, versions),
                                                   group=store.group, a2a=store._a2a_fn)
                cls = {"lru": hcache.LRUCache, "lfu": hcache.LFUCache, "lfuopt": hcache.LFUOptCache}[
                    config.cstable_policy.lower()]
                raw = cls(config.cache_limit, p.shape[0], width, node_id=-1 - p.id, device=store.table.device)
                raw.pull_bound = raw.push_bound = config.cache_bound
                raw.bind_remote(rstore)
                self.cache = hcache.CacheSparseTable.wrap(raw)
                self.remote_store = rstore
            else:
                hcache.register_table(p.id, store.table, row_start=store.starts[store.rank])
                self.cache = hcache.CacheSparseTable(config.cache_limit, p.shape[0], width, p.id,
                                                     config.cstable_policy, config.cache_bound,
                                                     device=store.table.device)
            p.cache = self.cache
            if getattr(config, "cache_perf_enable", False):
                self.cache.perf_enabled(True)
            self._push, self._pull, self._push_pull = self._push_cache, self._pull_cache, self._push_pull_cache
            if config.bsp == 0 and config.prefetch:
                self.compute = self._compute_bsp_prefetch
                if getattr(config, "cache_plan_ahead", False) and store.world == 1 and self.peek_ids is not None:
                    self._planned = []        # pull(k + 1) follows push(k) of the same ids batch after batch: the planned pairs
            elif config.prefetch:
                self.compute = self._compute_asp_prefetch
            else:
                self.compute = self._compute_no_prefetch
        else:
            self._push, self._pull, self._push_pull = self._push_sparse, self._pull_sparse, self._push_pull_sparse
            # :235-242: bsp >= 0 -> ssp (push, ssp_sync(version), pull), else asp (push_pull) when prefetching
            if config.prefetch and config.bsp >= 0:
                self.compute = self._compute_ssp_prefetch
                self.ssp_version = 0
            elif config.prefetch:
                self.compute = self._compute_asp_prefetch
            else:
                self.compute = self._compute_no_prefetch
        if config.prefetch:                                            # first prefetch (:168-176, 196-205)
            ids = first_ids if first_ids is not None else self.next_ids()
            if isinstance(ids, tuple):          # a laia data loader hands over (ids, push plan): cstable.py:49
                ids = ids[0]
            self.sparse_pull_val = torch.empty(tuple(ids.shape) + (p.shape[1],), dtype=torch.float32,
                                               device=ids.device)
            # (peek_ids counts from the batch next_ids() returns: an explicit first batch is the one before it)
            self._peek_offset = 0 if first_ids is not None else 1
            config.ps_map[p] = (self._pull(ids), self.sparse_pull_val)
            self._peek_offset = 1

    # -- compute variants (:37-56)
    def _mult_lr(self, grad):
        scale_(grad.values, self.learning_rate)

    def _compute_asp_prefetch(self, grad):
        self._mult_lr(grad)
        self.config.ps_map[self.parameter] = (self._push_pull(grad), self.sparse_pull_val)

    def _compute_ssp_prefetch(self, grad):
        """:41-46.  ssp_sync(version) lets a worker run ahead of the slowest one by at most `bsp` versions
        (ps-lite/include/ps/server/ssp_handler.h:41-67).  The sparse push / pull of a sharded store are
        collectives over all ranks, so no rank can run ahead at all: every tolerance is served by the
        lock step of the exchange itself, which satisfies the bound; only the version counter is kept."""
        self._mult_lr(grad)
        w = self._push(grad)
        if w is not None:
            w.wait()
        self.barrier()
        self.config.ps_map[self.parameter] = (self._pull(self.next_ids()), self.sparse_pull_val)
        self.ssp_version += 1

    def _compute_bsp_prefetch(self, grad):
        self._mult_lr(grad)
        w = self._push(grad)
        if w is not None:
            w.wait()
        self.barrier()
        self.config.ps_map[self.parameter] = (self._pull(self.next_ids()), self.sparse_pull_val)

    def _compute_no_prefetch(self, grad):
        self._mult_lr(grad)
        w = self._push(grad)
        if w is not None:
            w.wait()

    # -- cache flavour (:68-72, 88-92, 104-105)
    def _push_cache(self, grad):
        vals = grad.values.reshape(-1, self.parameter.shape[1])
        if self._planned is not None:
            idx = grad.indices.reshape(-1)
            if grad.push_indices is not None or not self._planned or not _same_tensor(self._planned[0], idx):
                raise RuntimeError("ParameterServerCommunicateOp (cache_plan_ahead): the gradients pushed are not those of the "
                                   "batch pulled last")
            self._planned.pop(0)
            return self.cache.embedding_update_planned(vals)
        if grad.push_indices is None:
            # The executor pushes the gradients of the batch it looked up last (bsp / ssp: push(k) follows pull(k) as the
            # cache's next operation, ParameterServerCommunicate.py:41-56) and does not write the ids in between: when the
            # indices ARE that lookup's tensor the update reuses its index plan (and takes the two-launch path).
            idx = grad.indices.reshape(-1)
            return self.cache.embedding_update(idx, vals, same_as_lookup=self.cache.looked_up_last(idx))
        return self.cache.embedding_update_with_push_keys(grad.indices.reshape(-1), grad.push_indices.reshape(-1), vals)

    def _pull_cache(self, ids):
        if isinstance(ids, tuple):              # (ids, push plan) of a laia-scheduled batch (cstable.py:49)
            ids = ids[0]
        dest = self.sparse_pull_val.reshape(-1, self.parameter.shape[1])
        if self._planned is not None:
            flat = ids.reshape(-1)
            if not self._planned:                                  # the first pull: nothing planned yet
                self.cache.plan_block([flat])
                self._planned.append(flat)
            if len(self._planned) != 1 or not _same_tensor(self._planned[0], flat):
                raise RuntimeError("ParameterServerCommunicateOp (cache_plan_ahead): pulls and pushes must alternate, batch "
                                   "after batch, on the tensors the loader handed out")
            nxt = self.peek_ids(self._peek_offset)                 # the batch after this one: its bookkeeping runs from now on,
            if nxt is not None:                                    # beside this batch's rows and the model's step
                nxt = (nxt[0] if isinstance(nxt, tuple) else nxt).reshape(-1)
                self.cache.plan_block([nxt])
                self._planned.append(nxt)
            return self.cache.embedding_lookup_planned(dest)      # (no next batch: the next pull plans for itself)
        return self.cache.embedding_lookup(ids.reshape(-1), dest)

    def _push_pull_cache(self, grad):
        nxt = self.next_ids()
        return self.cache.embedding_push_pull((nxt[0] if isinstance(nxt, tuple) else nxt).reshape(-1),
                                              self.sparse_pull_val.reshape(-1, self.parameter.shape[1]),
                                              grad.indices.reshape(-1),
                                              grad.values.reshape(-1, self.parameter.shape[1]))

    # -- plain PS flavour (SparsePush / SparsePull / SSPushPull, :74-111); values are already scaled
    def _push_sparse(self, grad):
        self.parameter.store.push(grad.indices, grad.values)
        return None

    def _pull_sparse(self, ids):
        if isinstance(ids, tuple):
            ids = ids[0]
        self.sparse_pull_val.copy_(self.parameter.store.pull(ids))
        return None

    def _push_pull_sparse(self, grad):
        self._push_sparse(grad)
        return self._pull_sparse(self.next_ids())


class ParameterServerSparsePullOp:
    """Inference-time pull of the next validation batch (:254-306)."""

    def __init__(self, parameter, next_ids):
        self.parameter, self.next_ids = parameter, next_ids

    def forward_hook(self, config):
        self.use_cache_table = config.cstable_policy is not None
        ids = self.next_ids()
        self.sparse_pull_val = torch.empty(tuple(ids.shape) + (self.parameter.shape[1],), dtype=torch.float32,
                                           device=ids.device)
        self.compute()

    def compute(self):
        ids = self.next_ids()
        if self.use_cache_table:
            w = self.parameter.cache.embedding_lookup(ids.reshape(-1),
                                                      self.sparse_pull_val.reshape(-1, self.parameter.shape[1]))
            w.wait()
        else:
            self.sparse_pull_val.copy_(self.parameter.store.pull(ids))
        return self.sparse_pull_val
