#!/usr/bin/env python3
"""Wide & Deep on Criteo-shaped data with the embedding path on herald_amd (MI355X).

The model is the reference's examples/ctr/models/wdl_criteo.py:8-46 (33,762,577 x d embedding table,
26 sparse fields, 13 dense features, a 13-256-256-256 tower, one 256+26d -> 1 output layer, sigmoid +
binary cross entropy, SGD on every parameter).  The dense tower runs on PyTorch-ROCm; everything that
touches the embedding table goes through the operator mirrors of herald_amd.hetu_ops, i.e. through
libherald_amd.so, in the three placements the reference's run_hetu.py offers:

  --embedding hbm    table in this GPU's HBM: EmbeddingLookUp -> DLGpuEmbeddingLookUp,
                     sparse SGD -> SGDOptimizerSparseUpdate           (comm_mode None)
  --embedding step   as hbm, but ONE launch per training step: ha_sgd_push_pull applies the sparse SGD of batch
                     k and looks batch k+1 up (rows both batches touch handed over inside the launch)
  --embedding queue  the work-queue step (ops.QueueStepPipeline / ha_qapply, what bench.py times): plans and queues a block
                     of steps ahead beside the model, long runs as fixed-order tree sums (1e-5 tolerance)
  --embedding step3  the same step through ops.StepPipeline (ha_step_*): ids three batches ahead, updated rows
                     forwarded to the next batch's output by the applying waves, nothing waits inside the launch
  --embedding ps     row-range sharded store (one shard per rank): SparsePull / SparsePush through
                     ParameterServerCommunicateOp                     (comm_mode PS; torchrun for N > 1)
  --embedding cache  HET cache (LRU / LFU / LFUOpt, bounded staleness) in front of the store
                     (comm_mode Hybrid + --cache POLICY --bound B, run_hetu.py:178-190)

  --laia             the reference's run_laia.py loop (examples/ctr/run_laia.py:214-236): every rank runs the laia
                     scheduler over the whole sample set (herald_amd.laia.LAIAScheduler: LaiaScheduler, or with
                     --local-shared the TopkScheduler whose local rank 0 feeds the others through shared-memory rings),
                     three LAIADataloaders (sparse ids, labels, dense features) hand it ITS share of each global batch,
                     the sparse one as the tuple (ids, push plan); the rows come through the HET cache over the table
                     row-sharded across the ranks, EmbeddingLookUp_Gradient(enable_push_index=True) turns the plan into
                     IndexedSlices.push_indices and the communicate op pushes with embedding_update_with_push_keys;
                     the dense tower's gradients are all-reduced (Hybrid).  torchrun for N > 1.
  --model dcn        Deep & Cross (examples/ctr/models/dcn_criteo.py:8-74; the reference hard-codes d = 128 there, the
                     embedding width is a parameter here) instead of Wide & Deep.

Data is synthetic (herald_amd.synth: per-field zipf over the public Criteo cardinalities); labels come
from a fixed random linear rule so that the loss has something to learn.

    python examples/ctr/run_wdl.py --embedding hbm --rows 2000000 --width 128 --steps 200
    torchrun --nproc-per-node 4 --master-addr 127.0.0.1 examples/ctr/run_wdl.py --laia --model dcn --width 128
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from herald_amd import hetu_ops, synth
from herald_amd.sharded import ShardedEmbedding

NFIELD, NDENSE = 26, 13


class Tower(torch.nn.Module):
    """The dense part of wdl_criteo (models/wdl_criteo.py:19-36); weights ~ N(0, 0.01) like init.random_normal."""

    def __init__(self, width, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)

        def w(*shape):
            return torch.nn.Parameter(torch.randn(*shape, generator=g) * 0.01)
        self.W1, self.W2, self.W3 = w(NDENSE, 256), w(256, 256), w(256, 256)
        self.W4 = w(256 + NFIELD * width, 1)

    def forward(self, dense, emb_flat):
        y3 = torch.relu(torch.relu(dense @ self.W1) @ self.W2) @ self.W3
        return torch.sigmoid(torch.cat([emb_flat, y3], dim=1) @ self.W4)


class CrossTower(torch.nn.Module):
    """The dense part of dcn_criteo (models/dcn_criteo.py:8-74): x0 = [embeddings | dense features], three cross
    layers x_{l+1} = x0 * (x_l w_l) + x_l + b_l, a 3-layer DNN on x0, one output layer over [cross | dnn]; every
    weight ~ N(0, 0.01)."""

    def __init__(self, width, seed=0, layers=3):
        super().__init__()
        g = torch.Generator().manual_seed(seed)

        def w(*shape):
            return torch.nn.Parameter(torch.randn(*shape, generator=g) * 0.01)
        n = NFIELD * width + NDENSE
        self.cw = torch.nn.ParameterList([w(n, 1) for _ in range(layers)])
        self.cb = torch.nn.ParameterList([w(n) for _ in range(layers)])
        self.W1, self.W2, self.W3 = w(n, 256), w(256, 256), w(256, 256)
        self.W4 = w(256 + n, 1)

    def forward(self, dense, emb_flat):
        x0 = torch.cat([emb_flat, dense], dim=1)
        x1 = x0
        for cw, cb in zip(self.cw, self.cb):
            x1 = x0 * (x1 @ cw) + x1 + cb
        y3 = torch.relu(torch.relu(x0 @ self.W1) @ self.W2) @ self.W3
        return torch.sigmoid(torch.cat([x1, y3], dim=1) @ self.W4)


def make_tower(model, width, seed=0):
    return {"wdl": Tower, "dcn": CrossTower}[model](width, seed)


def make_samples(nsamples, rows, seed=0):
    """One sample set for every rank (the laia scheduler distributes it): ids float32 [S, 26], dense [S, 13],
    labels [S, 1]."""
    rng = np.random.default_rng(seed + 77)
    wd = rng.standard_normal(NDENSE).astype(np.float32)
    chunk = 256
    raw = np.concatenate([synth.criteo_batch(chunk, step=5000 + seed * 131 + c, rows=rows)
                          for c in range(-(-nsamples // chunk))], axis=0)[:nsamples]
    ids = np.minimum(synth.as_f32_ids(raw), np.float32(rows - 1))
    dense = rng.standard_normal((nsamples, NDENSE)).astype(np.float32)
    score = dense @ wd + ((raw[:, :4].sum(axis=1) % 7) - 3).astype(np.float32)
    return ids, dense, (score > 0).astype(np.float32).reshape(-1, 1)


def train_laia(model="wdl", rows=200000, width=32, batch=64, steps=20, lr=0.01, cache="LRU", bound=0, cache_limit=None,
               seed=0, device="cuda:0", table_init=None, a2a=None, allreduce=None, local_shared=False, nsamples=None,
               perf=False, log_every=0):
    """The laia-driven loop (run_laia.py:214-236).  `batch` = samples per worker and step; a2a / allreduce: optional
    replacements of the collectives (several ranks on one GPU under gloo in the tests).  Returns (losses, embedding
    parameter, tower, communicate op)."""
    import torch.distributed as dist
    from herald_amd import laia as hlaia
    dev = torch.device(device)
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if nsamples is None:
        nsamples = world * batch * max(steps + 2, hlaia.LAIAScheduler.WINDOW + 1)
    ids_all, dense_all, label_all = make_samples(nsamples, rows, seed)
    limit = cache_limit if cache_limit is not None else max(rows // 10, batch * NFIELD)
    # the scheduler walks the whole sample set (every rank the same: laia_dataloader.py:29-95) ...
    sched = hlaia.LAIAScheduler(ids_all, batch, dataset="criteo", local_shared=local_shared)
    epochs = -(-steps * batch * world // nsamples) + 1
    sched.start(nrank=world, rank=rank, cache_limit=limit, dataset_num=3, epoch_num=epochs, key_limit=rows,
                local_rank=rank, local_size=world)
    # ... and three loaders hand this worker its share of every global batch (run_laia.py:188-199)
    sparse_dl = hlaia.LAIADataloader(sched, 0, True, ids_all, batch, name="train", device=dev)
    label_dl = hlaia.LAIADataloader(sched, 1, False, label_all, batch, name="train", device=dev)
    dense_dl = hlaia.LAIADataloader(sched, 2, False, dense_all, batch, name="train", device=dev)
    for dl in (sparse_dl, label_dl, dense_dl):
        dl.init_states(rank, world)
    batch = sched.batch_size
    tower = make_tower(model, width, seed).to(dev)
    opt = torch.optim.SGD(tower.parameters(), lr=lr)
    if table_init is None:
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        table_init = torch.randn((rows, width), generator=g, device=dev) * 0.01
    store = ShardedEmbedding(rows, width, dev, a2a=a2a)
    store.table.copy_(table_init[store.starts[store.rank]:store.starts[store.rank + 1]])
    param = hetu_ops.EmbeddingParameter(store=store)
    config = hetu_ops.Config(comm_mode="Hybrid", bsp=0, prefetch=True, cstable_policy=cache, cache_bound=bound,
                             cache_limit=limit, cache_perf_enable=perf)
    comm = hetu_ops.ParameterServerCommunicateOp(param, lr, next_ids=sparse_dl.get_next_arr)
    barrier = dist.barrier if world > 1 else (lambda: None)
    comm.forward_hook(config, first_ids=sparse_dl.get_next_arr(), barrier=barrier)
    lookup = hetu_ops.EmbeddingLookUp(param, enable_push_index=True)
    lookup.forward_hook(config)
    lookup_grad = hetu_ops.EmbeddingLookUp_Gradient(param.shape, enable_push_index=True)
    if allreduce is None:
        def allreduce(t):
            dist.all_reduce(t)

    losses = []
    t0 = time.perf_counter()
    for k in range(steps):
        ids_plan = sparse_dl.get_arr()                             # (ids [batch, 26], push plan): this worker's samples
        label = label_dl.get_arr()
        dense = dense_dl.get_arr()
        emb = torch.empty((batch, NFIELD, width), dtype=torch.float32, device=dev)
        lookup.compute(ids_plan, emb)                              # the rows the communicate op pulled for this batch
        emb.requires_grad_(True)
        pred = tower(dense, emb.reshape(batch, NFIELD * width))
        loss = torch.nn.functional.binary_cross_entropy(pred, label)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if world > 1:                                              # Hybrid: dense parameters by all-reduce, averaged
            for p_ in tower.parameters():
                allreduce(p_.grad)
                p_.grad.div_(world)
        opt.step()
        grad = lookup_grad.compute(emb.grad, ids_plan)             # IndexedSlices(indices, values, push_indices = plan)
        comm.compute(grad)              # -lr scale, embedding_update_with_push_keys, barrier, lookup of the next batch
        losses.append(float(loss.detach()))
        if log_every and (k + 1) % log_every == 0 and rank == 0:
            print("step %d loss %.5f (%.1f ms/step)" % (k + 1, np.mean(losses[-log_every:]),
                                                        1e3 * (time.perf_counter() - t0) / (k + 1)))
    torch.cuda.synchronize()
    sched.sched.close()
    return losses, param, tower, comm


def make_batches(nbatch, batch, rows, seed=0, rank=0, world=1):
    """-> list of (ids float32 [batch, 26], dense float32 [batch, 13], label float32 [batch, 1])."""
    rng = np.random.default_rng(seed + 1000 * rank)
    wd = rng.standard_normal(NDENSE).astype(np.float32)
    out = []
    for b in range(nbatch):
        raw = synth.criteo_batch(batch, step=b * world + rank, rows=rows)
        ids = np.minimum(synth.as_f32_ids(raw), np.float32(rows - 1))
        dense = rng.standard_normal((batch, NDENSE)).astype(np.float32)
        score = dense @ wd + ((raw[:, :4].sum(axis=1) % 7) - 3).astype(np.float32)
        out.append((ids, dense, (score > 0).astype(np.float32).reshape(-1, 1)))
    return out


def train(embedding="hbm", rows=200000, width=32, batch=256, steps=50, lr=0.01, cache="LRU", bound=0,
          cache_limit=None, seed=0, device="cuda:0", table_init=None, log_every=0, model="wdl", a2a=None, allreduce=None,
          bsp=0, cache_perf=False, perf_csv_dir=None, cache_planned=False):
    """Runs `steps` training steps; returns (losses, embedding parameter, tower).  cache_planned (--cache-planned, with
    --embedding cache at bsp 0 on one rank): the cache's planned flow -- the loader's ring hands the communicate op the ids one
    batch further ahead (`peek_ids`), the bookkeeping of batch k + 1 runs beside the step on batch k.  a2a / allreduce: optional
    replacements of the collectives at world size > 1 (several ranks on one GPU under gloo in the tests)."""
    dev = torch.device(device)
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    tower = make_tower(model, width, seed).to(dev)
    if allreduce is None:
        def allreduce(t):
            dist.all_reduce(t)
    opt = torch.optim.SGD(tower.parameters(), lr=lr)
    batches = make_batches(min(steps + 1, 64), batch, rows, seed, rank, world)
    dev_batches = [tuple(torch.from_numpy(a).to(dev) for a in b) for b in batches]
    state = {"k": 0}

    def ids_of(k):
        return dev_batches[k % len(dev_batches)][0]

    if table_init is None:
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        table_init = torch.randn((rows, width), generator=g, device=dev) * 0.01    # init.random_normal(stddev=0.01)
    if embedding in ("hbm", "step", "step3", "queue"):
        param = hetu_ops.EmbeddingParameter(table=table_init.clone())
        config = hetu_ops.Config(comm_mode=None)
        comm = None
    else:
        store = ShardedEmbedding(rows, width, dev, a2a=a2a)
        store.table.copy_(table_init[store.starts[store.rank]:store.starts[store.rank + 1]])
        param = hetu_ops.EmbeddingParameter(store=store)
        config = hetu_ops.Config(comm_mode="PS" if embedding == "ps" else "Hybrid", bsp=bsp, prefetch=True,
                                 cstable_policy=cache if embedding == "cache" else None, cache_bound=bound,
                                 cache_limit=cache_limit if cache_limit is not None else max(rows // 10, batch * NFIELD),
                                 cache_perf_enable=cache_perf, cache_plan_ahead=cache_planned)
        comm = hetu_ops.ParameterServerCommunicateOp(param, lr, next_ids=lambda: ids_of(state["k"] + 1),
                                                     peek_ids=lambda j: ids_of(state["k"] + 1 + j))
        barrier = dist.barrier if world > 1 else (lambda: None)
        comm.forward_hook(config, first_ids=ids_of(0), barrier=barrier)
    lookup = hetu_ops.EmbeddingLookUp(param)
    lookup.forward_hook(config)
    lookup_grad = hetu_ops.EmbeddingLookUp_Gradient(param.shape)

    fused = None
    if embedding == "step":
        from herald_amd import ops
        n = batch * NFIELD
        fused = {"plans": [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)],
                 "pends": [ops.PendingTable(dev), ops.PendingTable(dev)],
                 "outs": [torch.empty((batch, NFIELD, width), dtype=torch.float32, device=dev) for _ in range(2)]}
        ops.lookup_sort_pend(param.table, ids_of(0).reshape(-1), fused["plans"][0], fused["pends"][0],
                             out=fused["outs"][0])                  # the lookup of the first batch

    pipe = None
    if embedding == "step3":
        from herald_amd import ops
        pipe = ops.StepPipeline(param.table, batch * NFIELD, lr)
        pipe_out = pipe.start(ids_of(0), ids_of(1), ids_of(2))     # rows of the first batch
    qpipe = None
    if embedding == "queue":
        # the work-queue step (ha_qapply; what bench.py times): plans and queues a block of steps ahead on a side stream,
        # one launch per step between the model's backward and the next forward
        from herald_amd import ops
        qpipe = pipe = ops.QueueStepPipeline(param.table, batch * NFIELD, lr, block=4)
        pipe_out = pipe.start([ids_of(j) for j in range(pipe.LOOKAHEAD)])

    losses = []
    t0 = time.perf_counter()
    for k in range(steps):
        state["k"] = k
        ids, dense, label = dev_batches[k % len(dev_batches)]
        if pipe is not None:
            emb = pipe_out.detach().clone()                        # written by the previous step's launch
        elif fused is not None:
            emb = fused["outs"][k % 2].detach().clone()            # looked up by the previous step's launch
        else:
            emb = torch.empty((batch, NFIELD, width), dtype=torch.float32, device=dev)
            lookup.compute(ids, emb)                               # embedding_lookup_op
        emb.requires_grad_(True)
        pred = tower(dense, emb.reshape(batch, NFIELD * width))
        loss = torch.nn.functional.binary_cross_entropy(pred, label)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if world > 1:                                              # data parallel: dense parameters by all-reduce, averaged
            for p_ in tower.parameters():
                allreduce(p_.grad)
                p_.grad.div_(world)
        opt.step()
        grad = lookup_grad.compute(emb.grad, ids)                  # IndexedSlices(indices, values)
        if pipe is not None:
            # sparse SGD of this batch, rows of the next one, plan finish of batch k+2, sort of batch k+3: one launch
            pipe_out = pipe.step(grad.values.reshape(-1, width).contiguous(),
                                 ids_of(k + (qpipe.LOOKAHEAD if qpipe is not None else 3)))
        elif fused is not None:
            # sparse SGD of this batch + lookup of the next one, one launch (plans / pending tables alternate)
            ops.sgd_push_pull(param.table, fused["plans"][k % 2], grad.values.reshape(-1, width).contiguous(), lr,
                              fused["pends"][k % 2], ids_of(k + 1).reshape(-1), fused["plans"][(k + 1) % 2],
                              fused["pends"][(k + 1) % 2], next_out=fused["outs"][(k + 1) % 2])
        elif comm is None:
            hetu_ops.sgd_update_sparse(param, grad, lr)            # OptimizerOp, sparse SGD branch
        else:
            comm.compute(grad)                                     # -lr scale, push, (barrier), pull of batch k+1
        losses.append(float(loss.detach()))       # synchronises: a cheap place to poll the hand-off flag
        if fused is not None and log_every and (k + 1) % log_every == 0:
            ops.check_handoff(fused["plans"])
        if log_every and (k + 1) % log_every == 0 and rank == 0:
            print("step %d loss %.5f (%.1f ms/step)" % (k + 1, np.mean(losses[-log_every:]),
                                                        1e3 * (time.perf_counter() - t0) / (k + 1)))
    torch.cuda.synchronize()
    if fused is not None:
        ops.check_handoff(fused["plans"])
    if cache_perf and comm is not None and getattr(comm, "cache", None) is not None:
        dump_cache_perf([comm.cache], rank, perf_csv_dir)
    return losses, param, tower


def dump_cache_perf(caches, rank, csv_dir=None):
    """What /root/reference/examples/ctr/run_hetu.py:508-515 does after training with --cache-perf: one CSV per cache
    table, `csv/hetu_cache<idx>_<rank>.csv` beside the script, one row per cache call (the perf dicts: counts -- num_all,
    num_unique, num_miss, num_evict, num_transfered, ... -- and stage times)."""
    import csv
    csv_dir = csv_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csv")
    os.makedirs(csv_dir, exist_ok=True)
    paths = []
    for idx, c in enumerate(caches):
        if c is None:
            print("Cache perf is None")
            continue
        rows = list(c.get_perf())
        path = os.path.join(csv_dir, "hetu_cache%d_%d.csv" % (idx, rank))
        cols = sorted({k for r in rows for k in r})
        with open(path, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow([""] + cols)                      # pandas' to_csv layout: an index column first
            for i, r in enumerate(rows):
                w.writerow([i] + [r.get(k, "") for k in cols])
        paths.append(path)
    return paths


def main():
    # The reference's launch lines work as they are (/root/reference/examples/ctr/run_hetu.py:546-586, run_laia.py):
    #   python run_wdl.py --model wdl_criteo --comm Hybrid --cache lru --bound 3 --bsp 0 -b 256 -e 512 -r 0.1 --cache-perf
    # --comm / --cache choose the embedding engine unless --embedding names one (this build's own engines: the HBM-resident
    # table with one launch per step, "queue" = the work-queue step with lookahead, ...).
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="wdl", help="wdl_criteo | dcn_criteo (the reference's names) or wdl | dcn")
    ap.add_argument("-b", "--batch-size", "--batch", dest="batch", type=int, default=256)
    ap.add_argument("-e", "--embedding-size", "--width", dest="width", type=int, default=128)
    ap.add_argument("-r", "--cache-limit-ratio", type=float, default=0.1,
                    help="ratio of cache limit to total embedding sizes")
    ap.add_argument("--cache-perf", action="store_true",
                    help="record the cache's per-call counters and stage times; csv/hetu_cache<idx>_<rank>.csv after training")
    ap.add_argument("--val", action="store_true", help="accepted for the reference's launch lines (the data is synthetic)")
    ap.add_argument("--all", action="store_true", help="accepted for the reference's launch lines (the data is synthetic)")
    ap.add_argument("--comm", default=None, help="None, PS or Hybrid (sparse pull / push over the row-range sharded table)")
    ap.add_argument("--bsp", type=int, default=-1, help="bsp 0, asp -1, ssp > 0")
    ap.add_argument("--cache", default=None, help="cache policy: lru | lfu | lfuopt (with --comm PS / Hybrid)")
    ap.add_argument("--bound", type=int, default=100, help="cache bound")
    ap.add_argument("--cache-planned", action="store_true",
                    help="--embedding cache at --bsp 0: the cache's planned flow (bookkeeping of the next batch beside this step)")
    ap.add_argument("--nepoch", type=int, default=-1, help="epochs of `--steps` steps each (default: one)")
    ap.add_argument("--embedding", choices=["hbm", "step", "step3", "queue", "ps", "cache"], default=None,
                    help="this build's engine names; default: from --comm / --cache")
    ap.add_argument("--rows", type=int, default=33762577)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--laia", action="store_true", help="the laia-scheduled loop of run_laia.py (cache over the sharded table)")
    ap.add_argument("--local-shared", action="store_true", help="--laia with the TopkScheduler + shared-memory rings")
    args = ap.parse_args()
    model = args.model.split("_")[0]
    if model not in ("wdl", "dcn"):
        ap.error("--model must be wdl_criteo, dcn_criteo, wdl or dcn")
    args.model = model
    comm = None if args.comm in (None, "None") else args.comm
    if comm not in (None, "PS", "Hybrid"):
        ap.error("--comm must be None, PS or Hybrid (dense AllReduce-only runs have no sparse path to replace)")
    policy = {"lru": "LRU", "lfu": "LFU", "lfuopt": "LFUOpt"}.get((args.cache or "lru").lower())
    if policy is None:
        ap.error("--cache must be lru, lfu or lfuopt")
    if args.embedding is None:
        args.embedding = "hbm" if comm is None else ("cache" if args.cache else "ps")
    args.cache = policy
    if args.nepoch > 0:
        args.steps *= args.nepoch
    cache_limit = max(int(args.cache_limit_ratio * args.rows), args.batch * NFIELD)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.laia:
        losses = train_laia(args.model, args.rows, args.width, args.batch, args.steps, args.lr, args.cache, args.bound,
                            device="cuda:%d" % local_rank, local_shared=args.local_shared,
                            log_every=max(1, args.steps // 10))[0]
    else:
        losses = train(args.embedding, args.rows, args.width, args.batch, args.steps, args.lr, args.cache,
                       args.bound, cache_limit=cache_limit, device="cuda:%d" % local_rank,
                       log_every=max(1, args.steps // 10), model=args.model, bsp=args.bsp if comm is not None else 0,
                       cache_perf=args.cache_perf, cache_planned=args.cache_planned)[0]
    if local_rank == 0:
        print("first 10 steps: loss %.5f   last 10 steps: loss %.5f" % (np.mean(losses[:10]), np.mean(losses[-10:])))


if __name__ == "__main__":
    main()
