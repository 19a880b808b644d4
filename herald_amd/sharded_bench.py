"""bench.py's N>1 leg: the same per-GPU workload as N=1 (weak scaling), with the table row-range
sharded over the ranks and the sparse pull / push carried by RCCL all-to-all (herald_amd.sharded)."""
import json
import time

import numpy as np
import torch
import torch.distributed as dist

XGMI_LINK_GBS = 153.0   # per link, 7 links per GPU (MI355X_MICROARCH / task statement)
HBM_PEAK_GBS = 8000.0
# tests: a replacement of all_to_all_single for process groups whose backend cannot move device tensors (several ranks on
# ONE GPU under gloo: tests/test_gpu_bench_contract.py); None = the group's own collective (RCCL)
A2A_HOOK = None


def _world1_same_path(args, rank, dev, emb, ids_dev, grads, n, lr, steps=300, warmup=40):
    """The SAME engine (FramedStep, sized exchanges) at world size 1 on this rank's own shard: what a step of the N>1
    path costs before a byte crosses a link -- the N=1 line of bench.py comes from another engine (the work-queue
    step), so the step from N=1 to N=2 in a scaling curve is this number + communication, not communication alone.
    Every rank runs it (a GPU each: they stay in lockstep); ids are folded into the shard's row range."""
    from herald_amd.sharded import FramedStep, ShardedEmbedding
    # a store that sees itself alone: no process group is involved (ShardedEmbedding reads world / rank from one)
    one = ShardedEmbedding.__new__(ShardedEmbedding)
    one.group, one._a2a_fn, one.max_ids, one.side_stream = None, None, None, False
    one.world, one.rank = 1, 0
    one.rows, one.width, one.device = emb.local_rows, emb.width, emb.device
    one.starts = [0, emb.local_rows]
    one.local_rows, one.engine, one.table = emb.local_rows, emb.engine, emb.table
    one.stats = {"xgmi_bytes_out": 0, "xgmi_bytes_in": 0}
    one._slot, one._live, one.side_group = 0, {}, None
    local = [torch.remainder(t, float(emb.local_rows)) for t in ids_dev[:64]]
    fs = FramedStep(one, n, block=int(__import__("os").environ.get("HA_SHARD_BLOCK", "16")), graphs=False)
    nb, LA = len(local), fs.LOOKAHEAD
    outs = [torch.empty((n, emb.width), dtype=torch.float32, device=dev) for _ in range(2)]
    fs.start([local[j % nb] for j in range(LA)])
    native = fs.native_ok() and __import__("os").environ.get("HA_SHARD_NATIVE", "1") != "0"
    B = fs.block
    warmup -= warmup % B
    k = 0
    while k < warmup + steps:
        if k == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if native:          # a routing block's steps by one library call (ha_shard_steps)
            cnt = min(B - k % B, warmup + steps - k, (warmup - k) if k < warmup else 1 << 30)
            fs.steps([local[(k + i + LA) % nb] for i in range(cnt)], [grads[(k + i) % 2] for i in range(cnt)], lr,
                     outs=[outs[(k + i) % 2] for i in range(cnt)])
            k += cnt
        else:
            fs.pull(local[(k + LA) % nb], out=outs[k % 2])
            fs.push(grads[k % 2], lr)
            k += 1
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


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
    emb = ShardedEmbedding(args.rows, width, dev, side_group=os.environ.get("HA_SHARD_SIDE_GROUP") == "1", a2a=A2A_HOOK)
    g = torch.Generator(device=dev)
    g.manual_seed(123 + rank)
    chunk = 1 << 20
    for s in range(0, emb.local_rows, chunk):
        emb.table[s:s + chunk].normal_(0.0, 0.01, generator=g)
    gen = torch.Generator(device=dev)
    gen.manual_seed(456 + rank)
    grads = [torch.randn((n, width), dtype=torch.float32, device=dev, generator=gen) for _ in range(2)]
    lr = 1e-6
    framed = os.environ.get("HA_SHARD_READBACK", os.environ.get("HA_SHARD_SIZED")) != "1"
    fs = None
    same_path = None
    if framed and world > 1 and per_kernel and os.environ.get("HA_SHARD_NO_WORLD1") != "1":
        try:        # a diagnostic: it must never cost the line (every rank takes the same branch: no collective inside)
            same_path = _world1_same_path(args, rank, dev, emb, ids_dev, grads, n, lr)
        except Exception as e:      # noqa: BLE001
            same_path = None
            if rank == 0:
                import sys
                sys.stderr.write("same-path world-1 measurement failed: %s: %s\n" % (type(e).__name__, e))
        torch.cuda.synchronize()
    if framed:
        # herald_amd.sharded.FramedStep: the routing (plans, key frames, ONE key exchange) runs a block of batches at a
        # time, one block ahead, beside the steps; it leaves the per-owner counts of every batch in pinned host memory, so
        # the two row exchanges of a step are SIZED by the real counts without a read-back in the step -- exactly the rows
        # the batch names cross the fabric, and the keys a rank owns itself are read from / applied to its shard
        # directly.  HA_SHARD_FIXED=1 (or HA_SHARD_GRAPHS=1: hipGraph replay, measured slower than plain launches for
        # graphs of three kernels): the fixed row frames of round 3, equal-split exchanges that carry their padding.  A
        # batch that overflows its KEY frames on any rank takes the exchange with a read-back (counted below).
        graphs = os.environ.get("HA_SHARD_GRAPHS") == "1"
        fs = FramedStep(emb, n, row_cap=int(os.environ["HA_SHARD_ROW_CAP"]) if "HA_SHARD_ROW_CAP" in os.environ else None,
                        block=int(os.environ.get("HA_SHARD_BLOCK", "16")), graphs=graphs,
                        sized=not (graphs or os.environ.get("HA_SHARD_FIXED") == "1"))
        outs = [torch.empty((n, width), dtype=torch.float32, device=dev) for _ in range(2)]
        LA = fs.LOOKAHEAD
        ids_rows = [ids_dev[i] for i in range(nb)]          # the batches as tensors of their own, sliced once
        fs.start([ids_rows[j % nb] for j in range(LA)])

        def step(k):
            out = fs.pull(ids_rows[(k + LA) % nb], out=outs[k % 2])    # forward lookup; batch k + LA enters the routing
            fs.push(grads[k % 2], lr)                                 # backward: reduce, exchange, rank-ordered apply
            return out

        # ONE native call per run of steps inside a routing block (ha_shard_steps, csrc/shard.hip: the launches and the two row
        # exchanges of every step enqueued by the library from the pinned counts -- no Python between them; PSAgent.h:124-237
        # does a pull / a push inside one C++ call as well).  Needs the library's own exchange at world > 1.
        native_steps = fs.native_ok() and os.environ.get("HA_SHARD_NATIVE", "1") != "0"

        def run_steps(k0, count):
            k, end = k0, k0 + count
            while k < end:
                if native_steps:
                    cnt = min(fs.block - k % fs.block, end - k)
                    fs.steps([ids_rows[(k + i + LA) % nb] for i in range(cnt)], [grads[(k + i) % 2] for i in range(cnt)], lr,
                             outs=[outs[(k + i) % 2] for i in range(cnt)])
                    k += cnt
                else:
                    step(k)
                    k += 1
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

        native_steps = False

        def run_steps(k0, count):
            for k in range(k0, k0 + count):
                step(k)

    run_steps(0, warmup)
    torch.cuda.synchronize()
    dist.barrier()
    emb.stats = {"xgmi_bytes_out": 0, "xgmi_bytes_in": 0}
    fb0 = fs.fallbacks if fs is not None else 0
    t0 = time.perf_counter()
    run_steps(warmup, steps)
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    st = emb.stats
    xg = torch.tensor([st["xgmi_bytes_out"], st["xgmi_bytes_in"], st.get("xgmi_row_bytes_out", -1.0),
                       st.get("xgmi_useful_bytes_out", -1.0), st.get("owner_rows_max", 0)], dtype=torch.float64, device=dev)
    dist.all_reduce(xg, op=dist.ReduceOp.MAX)
    named = torch.tensor([st.get("owner_rows_sum", 0), st.get("owner_steps", 0)], dtype=torch.float64, device=dev)
    dist.all_reduce(named, op=dist.ReduceOp.SUM)
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
               "reduce": n * (4 * d + 4) + u_mean * 4 * d, "serve_push": u_mean * (12 * d + 4),
               # sized frames at world size 1: the lookup itself (id map, row read, row write per position) and the
               # reduce + server add in one launch (gradient row per position, one read-modify-write per unique row)
               "push_alone": n * (4 * d + 4) + u_mean * 8 * d}
        if fs.sized:
            # (the lookup reads a table row per UNIQUE key from HBM -- the other occurrences hit a cache -- and writes one per
            # position; the per-occurrence count n * 8d overstated it: 1.3 "of peak" at d = 128 in round 4's line)
            alg["expand"] = n * (4 * d + 4) + u_mean * 4 * d
        traffic = _pmc_traffic(batch, width)
        kernels = {}
        for name, us in times.items():
            short = name.split(" ")[0]
            tr = traffic.get(short)
            kernels[name] = {"us": us, "algorithmic_bytes": alg[short], "GBps": alg[short] / us / 1e3,
                             "frac_of_hbm_peak": alg[short] / us / 1e3 / HBM_PEAK_GBS,
                             "traffic": tr, "frac_by_traffic": (tr / us / 1e3 / HBM_PEAK_GBS) if tr else None}
    links = min(world - 1, 7)
    xgmi_gbs = float(xg[0].item()) / el / 1e9
    return {
        "value": world * n * steps / el, "ms_per_step": 1e3 * el / steps, "n": n, "steps": steps, "warmup": warmup,
        "ids_host": ids_host,
        "workload": "wdl_criteo bs=%d d=%d per GPU, %d fields, %d-row fp32 table row-range sharded over %d GPUs "
                    "(AveragePartitioner), sparse pull/push by RCCL all-to-all" % (batch, width, args.fields, args.rows, world),
        "sparse_update": "runs >= 64 occurrences: fixed-order tree sum (1e-5 tolerance); shorter: the reference's serial chain"
                         if tolerance else "the reference's serial chain, bit-exact",
        "exchange": (("row exchanges sized by the real per-owner counts (known a routing block ahead, no read-back in the "
                      "step), own keys served from the shard; key frames of %d keys per owner, routing in blocks of %d "
                      "batches; %d of %d timed steps overflowed their key frames and took the exchange with a read-back"
                      % (fs.rcap, fs.block, fs.fallbacks - fb0, steps)) if fs.sized else
                     ("fixed frames of %d rows per owner (padding travels), routing in blocks of %d batches, %s; %d of %d "
                      "timed steps took the sized exchange" % (fs.rcap, fs.block, "hipGraph replay" if fs.graphs else
                                                               "plain launches, no host read-back", fs.fallbacks - fb0, steps)))
                    if framed else "sized per step (host read-back of the counts)",
        "step_calls": ("one library call per run of steps inside a routing block (ha_shard_steps)" if native_steps else
                       "two Python calls per step (pull, push)"),
        "collectives": ("RCCL send / recv groups issued by the library on the step's stream (ha_xchg_*, checked against "
                        "torch.distributed's all-to-all at start-up)" if getattr(emb, "native", None) is not None else
                        "torch.distributed.all_to_all_single" if dist.get_world_size() > 1 else "none (world size 1)"),
        # egress of the busiest GPU.  bytes_carried: everything handed to the all-to-alls (rows + key frames);
        # bytes_useful: the rows some batch names + the key words in use -- with sized exchanges every row carried is one
        # (fixed frames: null, the frames do not say how full they are).  frac is priced on what was CARRIED.
        "xgmi": {"egress_GBps_per_gpu_max": xgmi_gbs, "peak_GBps_per_gpu": links * XGMI_LINK_GBS,
                 "frac": xgmi_gbs / (max(links, 1) * XGMI_LINK_GBS),
                 "bytes_carried_per_step": float(xg[0].item()) / steps,
                 "row_bytes_carried_per_step": float(xg[2].item()) / steps if xg[2].item() >= 0 else None,
                 "bytes_useful_per_step": float(xg[3].item()) / steps if xg[3].item() >= 0 else None,
                 # unique rows an owner is named per step by all ranks together: row-range shards of the Criteo key space
                 # are unevenly loaded (the small tables sit in one range) -- inherent to the partitioning north_star names
                 "owner_rows_mean": float(named[0].item() / named[1].item()) if named[1].item() > 0 else None,
                 "owner_rows_max": float(xg[4].item()) if named[1].item() > 0 else None},
        "same_path_world1_ms_per_step": same_path,
        "roofline": {"bound": "hbm", "kernel": "per-GPU local kernels of one sharded step (gather of served rows, "
                                               "dedup-reduce, apply of received gradients)",
                     "achieved": step_bytes / (el / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": step_bytes / (el / steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_step_per_gpu": step_bytes,
                     "note": "whole-step time (exchanges and host work included), not a kernel duration",
                     "kernels": kernels},
    }


def _pmc_traffic(batch, width):
    """HBM bytes per launch of the sharded step's kernels AT THIS SHAPE from the newest
    profiles/r*/pmc_traffic_sharded_bs<batch>_d<width>.json (rocprofv3 --pmc passes of the world-size-1 run at that
    shape, tools/profile_sharded.sh); {} -- every `traffic` null -- if no file was made for the shape."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*", "pmc_traffic_sharded_bs%d_d%d.json" % (batch, width))))
    if not files:
        return {}
    try:
        k = json.load(open(files[-1]))["kernels"]
    except Exception:
        return {}
    names = {"serve_pull": "ha::shard_serve_pull", "expand": "ha::gather",
             "reduce": "ha::apply_mapped_kernel", "serve_push": "ha::shard_frames_apply_kernel",
             "push_alone": "ha::apply_"}
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
                       "collectives": main["collectives"], "step_calls": main["step_calls"],
                       "sparse_update": main["sparse_update"]},
            "xgmi": main["xgmi"], "ranks_seen": dist.get_world_size(), "roofline": main["roofline"],
            # the N>1 engine at world size 1 on one shard (ms per step; null at N = 1, where this line IS that number):
            # bench.py's N=1 line runs the work-queue engine instead, so N=1 -> N=2 = an engine change + communication
            "same_path_world1_ms_per_step": main["same_path_world1_ms_per_step"],
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
                                    "same_path_world1_ms_per_step": second["same_path_world1_ms_per_step"],
                                    "note": "BASELINE configs[2]'s per-GPU shape; not part of `value`"}
        # (RCCL writes its version banner to C stdout when NCCL_DEBUG=VERSION -- the GPU boxes set it -- and C stdio holds it back
        # until the process exits: flushed HERE, so that the JSON line is the last line of the output, not the first of six)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:      # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)
    dist.barrier()          # rank 0 may still be timing the CPU baseline: tear the group down together
    dist.destroy_process_group()
