"""bench.py's N>1 leg: the same per-GPU workload as N=1 (weak scaling), with the table row-range
sharded over the ranks and the sparse pull / push carried by RCCL all-to-all (herald_amd.sharded)."""
import json
import time

import numpy as np
import torch
import torch.distributed as dist

XGMI_LINK_GBS = 153.0   # per link, 7 links per GPU (MI355X_MICROARCH / task statement)
HBM_PEAK_GBS = 8000.0


def _measure(args, rank, world, dev, batch, width, steps, warmup, per_kernel):
    """One sharded workload: `steps` timed steps of bs=`batch`, d=`width` per GPU -> dict of what rank 0 reports."""
    import os
    from herald_amd import ops, synth
    from herald_amd.sharded import FramedStep, ShardedEmbedding

    # runs of 64+ occurrences of a key as a fixed-order tree sum (within BASELINE.json's 1e-5 on accumulated gradients;
    # the N=1 step's work-queue engine has the same classes); HA_EXACT_SGD=1: the serial chain everywhere
    tolerance = os.environ.get("HA_EXACT_SGD") != "1"
    ops.set_tolerance_mode(tolerance)
    n = batch * args.fields
    nb = min(args.distinct_batches, 256)
    ids_host = np.empty((nb, n), dtype=np.float32)
    for b in range(nb):
        f = synth.as_f32_ids(synth.criteo_batch(batch, step=b * world + rank, rows=args.rows,
                                                nfields=args.fields)).reshape(-1)
        np.minimum(f, np.float32(args.rows - 1), out=f)
        ids_host[b] = f
    ids_dev = torch.from_numpy(ids_host).to(dev)
    emb = ShardedEmbedding(args.rows, width, dev, side_group=os.environ.get("HA_SHARD_SIDE_GROUP") == "1")
    g = torch.Generator(device=dev)
    g.manual_seed(123 + rank)
    chunk = 1 << 20
    for s in range(0, emb.local_rows, chunk):
        emb.table[s:s + chunk].normal_(0.0, 0.01, generator=g)
    gen = torch.Generator(device=dev)
    gen.manual_seed(456 + rank)
    grads = [torch.randn((n, width), dtype=torch.float32, device=dev, generator=gen) for _ in range(2)]
    lr = 1e-6
    framed = os.environ.get("HA_SHARD_SIZED") != "1"
    fs = None
    if framed:
        # Fixed frames (herald_amd.sharded.FramedStep): no launch or exchange size depends on a device-side count, so
        # a step needs no host read-back (and can replay from hipGraphs: HA_SHARD_GRAPHS=1 -- measured slower than
        # plain launches for graphs of three kernels, so it is not the default); the routing runs a block of batches at
        # a time, one block ahead, beside the steps (four launches and one key exchange per block).  A batch that
        # overflows its frames on any rank takes the sized exchange (counted below).
        fs = FramedStep(emb, n, row_cap=int(os.environ["HA_SHARD_ROW_CAP"]) if "HA_SHARD_ROW_CAP" in os.environ else None,
                        block=int(os.environ.get("HA_SHARD_BLOCK", "8")), graphs=os.environ.get("HA_SHARD_GRAPHS") == "1")
        outs = [torch.empty((n, width), dtype=torch.float32, device=dev) for _ in range(2)]
        LA = fs.LOOKAHEAD
        fs.start([ids_dev[j % nb] for j in range(LA)])

        def step(k):
            out = fs.pull(ids_dev[(k + LA) % nb], out=outs[k % 2])    # forward lookup; batch k + LA enters the routing
            fs.push(grads[k % 2], lr)                                 # backward: reduce, exchange, rank-ordered apply
            return out
    else:
        # Sized exchanges: the routing of batch k+1 (plan, counts and keys exchange, the one host read-back) is
        # prefetched while the rows of batch k are pulled and pushed: its ids are resident one step ahead (the
        # reference's dataloader / PS prefetch does the same).
        state = {"route": emb.prefetch(ids_dev[0], after_current=False)}

        def step(k):
            cur = state["route"]
            nxt = emb.prefetch(ids_dev[(k + 1) % nb], after_current=False)
            out = emb.pull(route=cur)                            # forward lookup
            emb.push(None, grads[k % 2], lr, route=cur)          # backward: -lr scale, dedup-reduce, exchange, apply
            emb.complete(nxt)                                    # host counts + keys exchange of the next batch
            state["route"] = nxt
            return out

    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    dist.barrier()
    emb.stats = {"xgmi_bytes_out": 0, "xgmi_bytes_in": 0}
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    xg = torch.tensor([emb.stats["xgmi_bytes_out"], emb.stats["xgmi_bytes_in"]], dtype=torch.float64, device=dev)
    dist.all_reduce(xg, op=dist.ReduceOp.MAX)
    used = [(warmup + k) % nb for k in range(steps)]
    u_mean = float(np.mean([np.unique(ids_host[b]).size for b in sorted(set(used))]))
    d = width
    # HBM side of one rank's step: the rows it serves and the gradients it applies are, summed over the ranks, the
    # same N*(8d+4) + N*(4d+4) + U*8d bytes as at N=1 (weak scaling: every rank brings its own batch)
    step_bytes = n * (8 * d + 4) + n * (4 * d + 4) + u_mean * 8 * d
    kernels = None
    if per_kernel and fs is not None and rank == 0 and world == 1:
        # per launch: duration by HIP events, algorithmic bytes (U = unique keys of the batch, all owned here at W = 1)
        fs.pull(ids_dev[(warmup + steps + LA) % nb], out=outs[0])
        times = fs.kernel_times(grads[0], lr)
        fs.push(grads[0], lr)
        torch.cuda.synchronize()
        alg = {"serve_pull": u_mean * (8 * d + 4), "expand": n * (4 * d + 4) + u_mean * 4 * d,
               "reduce": n * (4 * d + 4) + u_mean * 4 * d, "serve_push": u_mean * (12 * d + 4)}
        traffic = _pmc_traffic()
        kernels = {}
        for name, us in times.items():
            short = name.split(" ")[0]
            kernels[name] = {"us": us, "algorithmic_bytes": alg[short], "GBps": alg[short] / us / 1e3,
                             "frac_of_hbm_peak": alg[short] / us / 1e3 / HBM_PEAK_GBS,
                             "traffic": traffic.get(short)}
    links = min(world - 1, 7)
    xgmi_gbs = float(xg[0].item()) / el / 1e9
    return {
        "value": world * n * steps / el, "ms_per_step": 1e3 * el / steps, "n": n, "steps": steps, "warmup": warmup,
        "ids_host": ids_host,
        "workload": "wdl_criteo bs=%d d=%d per GPU, %d fields, %d-row fp32 table row-range sharded over %d GPUs "
                    "(AveragePartitioner), sparse pull/push by RCCL all-to-all" % (batch, width, args.fields, args.rows, world),
        "sparse_update": "runs >= 64 occurrences: fixed-order tree sum (1e-5 tolerance); shorter: the reference's serial chain"
                         if tolerance else "the reference's serial chain, bit-exact",
        "exchange": ("fixed frames of %d rows per owner, routing in blocks of %d batches, %s; %d of %d steps took the "
                     "sized exchange" % (fs.rcap, fs.block, "hipGraph replay" if fs.graphs else "plain launches, no host "
                                         "read-back", fs.fallbacks, steps + warmup)) if framed else "sized (host read-back)",
        "xgmi": {"egress_GBps_per_gpu_max": xgmi_gbs, "peak_GBps_per_gpu": links * XGMI_LINK_GBS,
                 "frac": xgmi_gbs / (max(links, 1) * XGMI_LINK_GBS)},
        "roofline": {"bound": "hbm", "kernel": "per-GPU local kernels of one sharded step (gather of served rows, "
                                               "dedup-reduce, apply of received gradients)",
                     "achieved": step_bytes / (el / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": step_bytes / (el / steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_step_per_gpu": step_bytes,
                     "note": "whole-step time (exchanges and host work included), not a kernel duration",
                     "kernels": kernels},
    }


def _pmc_traffic():
    """HBM bytes per launch of the sharded step's kernels from the newest profiles/r*/pmc_traffic_sharded.json
    (rocprofv3 --pmc passes of the world-size-1 run, tools/profile_sharded.sh); {} if there is none."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*", "pmc_traffic_sharded.json")))
    if not files:
        return {}
    try:
        k = json.load(open(files[-1]))["kernels"]
    except Exception:
        return {}
    names = {"serve_pull": "ha::shard_serve_pull_frames_kernel", "expand": "ha::gather_vec4_kernel",
             "reduce": "ha::apply_mapped_kernel", "serve_push": "ha::shard_frames_apply_kernel"}
    out = {}
    for short, kn in names.items():
        hit = [v for name, v in k.items() if name.startswith(kn)]
        if hit:
            out[short] = hit[0].get("hbm_bytes_per_launch")
    return out


def run(args, rank, world, dev, cpu_baseline_fn=None):
    main = _measure(args, rank, world, dev, args.batch, args.width, args.steps, args.warmup, per_kernel=True)
    # BASELINE configs[2] (wdl_criteo bs=4096 d=128 per GPU: the shape the reference's scaling table is quoted on) as a
    # second, shorter measurement in the same line -- never part of `value`
    second = None
    if not (args.batch == 4096 and args.width == 128) and not getattr(args, "no_config_c", False):
        try:
            torch.cuda.empty_cache()
            second = _measure(args, rank, world, dev, 4096, 128, max(20, min(args.steps, 100)), 60, per_kernel=True)
        except Exception as e:      # a failure here must not lose the headline line
            second = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        cpu_base = None
        if cpu_baseline_fn is not None and not args.no_cpu_baseline:
            cpu_base = cpu_baseline_fn(args, main["ids_host"])
        line = {
            "metric": "embedding rows/s (lookup+grad)", "value": main["value"], "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": main["workload"], "ids_per_step_per_gpu": main["n"],
                       "parallelism": "row-sharded x%d" % world, "exchange": main["exchange"],
                       "sparse_update": main["sparse_update"]},
            "xgmi": main["xgmi"], "ranks_seen": dist.get_world_size(), "roofline": main["roofline"],
            "cpu_baseline": cpu_base,
        }
        if second is not None:
            if "error" in second:
                line["config_c"] = second
            else:
                line["config_c"] = {"value": second["value"], "unit": "rows/s", "ms_per_step": second["ms_per_step"],
                                    "steps": second["steps"], "warmup": second["warmup"],
                                    "config": {"workload": second["workload"], "ids_per_step_per_gpu": second["n"],
                                               "exchange": second["exchange"], "sparse_update": second["sparse_update"]},
                                    "xgmi": second["xgmi"], "roofline": second["roofline"],
                                    "note": "BASELINE configs[2]'s per-GPU shape; not part of `value`"}
        print(json.dumps(line), flush=True)
    dist.barrier()          # rank 0 may still be timing the CPU baseline: tear the group down together
    dist.destroy_process_group()
