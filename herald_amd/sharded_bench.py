"""bench.py's N>1 leg: the same per-GPU workload as N=1 (weak scaling), with the table row-range
sharded over the ranks and the sparse pull / push carried by RCCL all-to-all (herald_amd.sharded)."""
import json
import time

import numpy as np
import torch
import torch.distributed as dist

XGMI_LINK_GBS = 153.0   # per link, 7 links per GPU (MI355X_MICROARCH / task statement)
HBM_PEAK_GBS = 8000.0


def run(args, rank, world, dev, cpu_baseline_fn=None):
    from herald_amd import synth
    from herald_amd.sharded import ShardedEmbedding

    n = args.batch * args.fields
    nb = min(args.distinct_batches, 256)
    ids_host = np.empty((nb, n), dtype=np.float32)
    for b in range(nb):
        f = synth.as_f32_ids(synth.criteo_batch(args.batch, step=b * world + rank, rows=args.rows,
                                                nfields=args.fields)).reshape(-1)
        np.minimum(f, np.float32(args.rows - 1), out=f)
        ids_host[b] = f
    ids_dev = torch.from_numpy(ids_host).to(dev)
    import os
    emb = ShardedEmbedding(args.rows, args.width, dev, side_group=os.environ.get("HA_SHARD_SIDE_GROUP") == "1")
    g = torch.Generator(device=dev)
    g.manual_seed(123 + rank)
    chunk = 1 << 20
    for s in range(0, emb.local_rows, chunk):
        emb.table[s:s + chunk].normal_(0.0, 0.01, generator=g)
    gen = torch.Generator(device=dev)
    gen.manual_seed(456 + rank)
    grads = [torch.randn((n, args.width), dtype=torch.float32, device=dev, generator=gen) for _ in range(2)]
    lr = 1e-6

    # The routing of batch k+1 (plan, counts and keys exchange, the one host read-back) is prefetched on
    # the side stream while the rows of batch k are pulled and pushed: its ids are resident one step
    # ahead (the reference's dataloader / PS prefetch does the same).
    state = {"route": emb.prefetch(ids_dev[0], after_current=False)}

    def step(k):
        cur = state["route"]
        nxt = emb.prefetch(ids_dev[(k + 1) % nb], after_current=False)
        out = emb.pull(route=cur)                            # forward lookup
        emb.push(None, grads[k % 2], lr, route=cur)          # backward: -lr scale, dedup-reduce, exchange, apply
        emb.complete(nxt)                                    # host counts + keys exchange of the next batch
        state["route"] = nxt
        return out

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    dist.barrier()
    emb.stats = {"xgmi_bytes_out": 0, "xgmi_bytes_in": 0}
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    xg = torch.tensor([emb.stats["xgmi_bytes_out"], emb.stats["xgmi_bytes_in"]], dtype=torch.float64, device=dev)
    dist.all_reduce(xg, op=dist.ReduceOp.MAX)
    # per-GPU unique counts of the batches this rank ran (for the local kernels' algorithmic bytes)
    used = [(args.warmup + k) % nb for k in range(args.steps)]
    u_mean = float(np.mean([np.unique(ids_host[b]).size for b in sorted(set(used))]))
    ranks_seen = dist.get_world_size()
    if rank == 0:
        rows_per_s = world * n * args.steps / el
        # HBM side of one rank's step: the rows it serves and the gradients it applies are, summed over
        # the ranks, the same N*(8d+4) + N*(4d+4) + U*8d bytes as at N=1 (weak scaling: every rank brings
        # its own batch); per GPU that is the N=1 figure, moved in `el / steps`.
        d = args.width
        step_bytes = n * (8 * d + 4) + n * (4 * d + 4) + u_mean * 8 * d
        hbm_gbs = step_bytes / (el / args.steps) / 1e9
        cpu_base = None
        if cpu_baseline_fn is not None and not args.no_cpu_baseline:
            cpu_base = cpu_baseline_fn(args, ids_host)
        xgmi_gbs = float(xg[0].item()) / el / 1e9
        links = min(world - 1, 7)
        print(json.dumps({
            "metric": "embedding rows/s (lookup+grad)", "value": rows_per_s, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "wdl_criteo bs=%d d=%d per GPU, %d fields, %d-row fp32 table row-range "
                                   "sharded over %d GPUs (AveragePartitioner), sparse pull/push by RCCL "
                                   "all-to-all" % (args.batch, args.width, args.fields, args.rows, world),
                       "ids_per_step_per_gpu": n, "parallelism": "row-sharded x%d" % world},
            "xgmi": {"egress_GBps_per_gpu_max": xgmi_gbs, "peak_GBps_per_gpu": links * XGMI_LINK_GBS,
                     "frac": xgmi_gbs / (max(links, 1) * XGMI_LINK_GBS)},
            "ranks_seen": ranks_seen,
            "roofline": {"bound": "hbm", "kernel": "per-GPU local kernels of one sharded step (gather of served "
                                                   "rows, dedup-reduce, apply of received gradients)",
                         "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_step_per_gpu": step_bytes,
                         "note": "whole-step time (exchanges and host work included), not a kernel duration"},
            "cpu_baseline": cpu_base,
        }), flush=True)
    dist.barrier()          # rank 0 may still be timing the CPU baseline: tear the group down together
    dist.destroy_process_group()
