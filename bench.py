#!/usr/bin/env python3
"""bench.py -- embedding rows/s (lookup + grad) of the hot path on MI355X.

One "step" = one pass of the hot path over one synthetic Criteo-shaped batch:
    forward : out[i,:] = table[ids[i],:]                         (ha_gather_f32ids)
    plan    : sorted-unique / inverse / counts of the batch ids  (ha_plan_build_f32ids)
    backward: table[id,:] -= lr * grad[i,:] per occurrence, occurrence order per row,
              every unique row read and written once             (ha_sgd_apply)
Inputs (ids of >= 1024 distinct batches, gradient rows, the table) are resident in HBM before the
timed region.  N=1 workload: BASELINE.json configs[1], wdl_criteo bs=256 d=512, full 33,762,577-row
table in HBM.  N>1: the table is row-range sharded over the ranks (AveragePartitioner ranges) and
ids / rows / reduced gradients cross ranks with RCCL all-to-all (herald_amd.sharded).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
LR = 1e-6              # examples/ctr/models/wdl_criteo.py:12


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2048)
    p.add_argument("--warmup", type=int, default=256)
    p.add_argument("--batch", type=int, default=256)
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--rows", type=int, default=33762577)
    p.add_argument("--fields", type=int, default=26)
    p.add_argument("--distinct-batches", type=int, default=1024)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-pass", action="store_true")
    p.add_argument("--no-cpu-full-table", action="store_true",
                   help="cpu_baseline: skip the second figure on a host copy of the whole table (needs rows x width x 4 bytes "
                        "of host DRAM and ~30 s)")
    p.add_argument("--no-cache-prefill", action="store_true",
                   help="cache_tier: time the pairs on the cache as 64 warm-up batches leave it (far from full: no "
                        "evictions) instead of filling it to its limit first")
    p.add_argument("--no-cache-tier", action="store_true",
                   help="skip the secondary measurement of the LRU cache tier (limit 0.1 x rows)")
    p.add_argument("--no-laia", action="store_true",
                   help="skip the secondary measurement of the laia scheduler (configs[3] shape)")
    p.add_argument("--no-config-c", action="store_true",
                   help="N>1 (sharded) leg: skip the second measurement at BASELINE configs[2]'s shape (bs=4096 d=128)")
    p.add_argument("--no-wide", action="store_true",
                   help="skip the secondary measurements of the work-queue step at BASELINE configs[2] / configs[3]'s per-GPU "
                        "shapes (bs=4096 d=128 and bs=1024 d=512 on one GPU: the wide path, batches beyond 7,168 ids)")
    p.add_argument("--no-sweeps", action="store_true",
                   help="skip `bit_exact_engine` (the bit-exact one-launch engine beside the headline) and `lookahead_sweep` (the "
                        "headline's step at blocks of 1, 2, 4 steps = 3, 6, 12 batches of id lookahead)")
    p.add_argument("--no-cold-tier", action="store_true",
                   help="skip the secondary measurement of the host-DRAM cold tier (BASELINE configs[4] shape)")
    p.add_argument("--cold-rows", type=int, default=33554432,
                   help="rows of the pinned host table of the cold-tier line (x 64 floats = 8 GiB)")
    p.add_argument("--graph-steps", type=int, default=None,
                   help="steps captured per hipGraph (1 = plain launches).  Default: plain launches for the queue engine "
                        "(one ~14 us launch per step: the host enqueues a block of them by one library call, ~9 us each, and "
                        "a launch-to-launch boundary on a stream is shorter than one inside a hipGraph: 13.5 vs 14.0 us per "
                        "step), 32 otherwise")
    p.add_argument("--lookahead", type=int, default=None, choices=(1, 3),
                   help="older spelling of --engine: 1 = handoff, 3 = forward")
    p.add_argument("--pre-roll", type=int, default=2,
                   help="how many of the W warm-up steps run right in front of the timed region (no synchronisation in "
                        "between); 0 = all W before the synchronisation")
    p.add_argument("--clock-warm", type=int, default=8,
                   help="untimed 512 MiB device copies enqueued right before the pre-roll (0 = none)")
    p.add_argument("--cache-warm", type=int, default=256,
                   help="untimed read-only lookups of the batches that PRECEDE the timed region in the stream of batches, "
                        "enqueued after the clock-warm copies: a long run reaches the timed steps with the rows of its recent "
                        "batches in the last-level cache; a 20-step run behind 1 GiB copies does not (0 = none)")
    p.add_argument("--no-gate", action="store_true",
                   help="do not hold the stream until the host has enqueued the timed region")
    p.add_argument("--engine", default=None, choices=("handoff", "forward", "queue"),
                   help="the one-launch step: queue (default) = ha_qapply, every step driven by a work queue; plans and "
                        "queues are prepared a block of steps at a time on a side stream inside the timed region; keys "
                        "with 16+ occurrences in a batch applied as row - tree_sum(lr*g), within BASELINE.json's 1e-5 "
                        "(everything else bit-exact).  handoff = ha_sgd_push_pull_* (ids one batch ahead, in-launch "
                        "hand-off through pending tables, bit-exact throughout).  forward = ha_step_* (ids three batches "
                        "ahead, rows forwarded from the applying waves, bit-exact throughout)")
    p.add_argument("--queue-block", type=int, default=16,
                   help="--engine queue: steps per block (the plans / queues of a block are prepared by two launches on a "
                        "side stream beside the steps of the block before; ids are needed 3 blocks ahead)")
    p.add_argument("--queue-serial", action="store_true",
                   help="--engine queue with the preparation launch in front of every step on the timed stream instead of "
                        "beside the steps on side streams")
    p.add_argument("--launches", type=int, default=1, choices=(1, 2),
                   help="launches per step: 1 = ha_sgd_push_pull (apply(k) beside lookup(k+1), default), "
                        "2 = ha_lookup_sort + ha_sgd_apply_finish")
    p.add_argument("--grad-buffers", type=int, default=24,
                   help="distinct gradient / output buffers cycled through (24 x 13.6 MB = 327 MB > the "
                        "256 MiB Infinity Cache, so gradient reads and output writes are HBM traffic)")
    return p.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes (one rank per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and wait for them.  This parent never touches the GPU
    and never re-execs itself."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    return rc


def algorithmic_bytes(n, u, width):
    """fwd n*(8d+4); bwd n*(4d+4) + U*8d."""
    fwd = n * (8 * width + 4)
    bwd = n * (4 * width + 4) + u * 8 * width
    return fwd, bwd


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the newest committed PMC summary (profiles/rNN/
    pmc_traffic.json, produced by tools/profile_round.sh in separate --pmc passes), or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                ks = json.load(fh)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for name, v in ks.items():
            if name.startswith(kernel_prefix):
                return v["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, None


def init_table(rows, width, dev, seed=123):
    t = torch.empty((rows, width), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    chunk = 1 << 20
    for s in range(0, rows, chunk):
        t[s:s + chunk].normal_(0.0, 0.01, generator=g)   # init.random_normal(stddev=0.01)
    return t


def make_batches(args, rank, world):
    from herald_amd import synth
    nb = args.distinct_batches
    ids = np.empty((nb, args.batch * args.fields), dtype=np.float32)
    uniq = np.empty(nb, dtype=np.int64)
    for b in range(nb):
        # (HA_BENCH_SAME_BATCH=1, tools/l2_affinity_bound.sh only: every step names batch 0 -- an experiment, not the workload;
        # the line says so in config.same_batch)
        same = os.environ.get("HA_BENCH_SAME_BATCH") == "1"
        raw = synth.criteo_batch(args.batch, step=0 if same else b * world + rank, rows=args.rows, nfields=args.fields)
        f = synth.as_f32_ids(raw).reshape(-1)
        # float32 rounding above 2^24 can land on `rows` itself; the reference would read out of
        # bounds there, the synthetic stream keeps ids inside the table
        np.minimum(f, np.float32(args.rows - 1), out=f)
        ids[b] = f
        uniq[b] = np.unique(f).size
    return ids, uniq


def cpu_baseline(args, ids_host):
    """Reference CPU path restated (oracle/oracle.c), timed on this box's host cores on a bounded
    sample: a 1M-row slice of the table, ids folded into it, gather (OpenMP) + serial sparse SGD."""
    from oracle import cpu, ref
    kind = "port"
    lookup, update = cpu.embedding_lookup, cpu.sgd_sparse_update
    if ref.dnnl_available():
        # the reference's own cpu_EmbeddingLookup / cpu_SGDOptimizerSparseUpdate (oracle/_ref/libref_dnnl.so,
        # src/dnnl_ops/*.cpp compiled unchanged by oracle/build_ref.sh; prebuilt, travels with the snapshot)
        kind = "reference"
        lookup, update = ref.dnnl_embedding_lookup, ref.dnnl_sgd_sparse_update
    rows_cpu = min(args.rows, 1_000_000)
    rng = np.random.default_rng(7)
    table = (rng.standard_normal((rows_cpu, args.width), dtype=np.float32) * np.float32(0.01))
    n = ids_host.shape[1]
    grads = rng.standard_normal((n, args.width), dtype=np.float32)
    steps = 0
    t_budget = 12.0
    # warm-up
    idf = np.mod(ids_host[0], rows_cpu).astype(np.float32)
    lookup(table, idf)
    update(table, idf, grads, LR)
    t0 = time.perf_counter()
    while True:
        idf = np.mod(ids_host[steps % ids_host.shape[0]], rows_cpu).astype(np.float32)
        lookup(table, idf)
        update(table, idf, grads, LR)
        steps += 1
        el = time.perf_counter() - t0
        if el > t_budget or steps >= 400:
            break
    # The same on the table the GPU line uses (all `--rows` rows, host DRAM): a 1M-row slice (2 GB) sits partly in the host's
    # last-level caches and flatters the CPU.  Only where the host has the memory; ids as they are (no folding).
    full = None
    try:
        import psutil
        need = args.rows * args.width * 4
        if rows_cpu < args.rows and psutil.virtual_memory().available > need + (32 << 30) and not args.no_cpu_full_table:
            del table
            big = np.empty((args.rows, args.width), dtype=np.float32)
            for s0 in range(0, args.rows, 1 << 20):      # (touch every page once: the timed steps must not fault them in)
                big[s0:s0 + (1 << 20)] = np.float32(0.01)
            idf = ids_host[0].astype(np.float32)
            lookup(big, idf)
            update(big, idf, grads, LR)
            fs, t1 = 0, time.perf_counter()
            while True:
                idf = ids_host[fs % ids_host.shape[0]].astype(np.float32)
                lookup(big, idf)
                update(big, idf, grads, LR)
                fs += 1
                fel = time.perf_counter() - t1
                if fel > 6.0 or fs >= 200:
                    break
            full = {"value": n * fs / fel, "unit": "rows/s", "rows": args.rows, "steps": fs, "ms_per_step": 1e3 * fel / fs,
                    "table_GB": need / 1e9}
            del big
    except Exception as ex:      # noqa: BLE001 -- the slice figure stands
        full = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return {
        "value": n * steps / el, "unit": "rows/s", "cores": cpu.num_threads(), "kind": kind,
        "full_table": full,
        "sample": "%d steps of bs=%d d=%d on a %d-row table slice: %s OpenMP gather (%d threads) + serial "
                  "sparse SGD (1 thread, as cpu_SGDOptimizerSparseUpdate mandates), %.1f ms/step"
                  % (steps, args.batch, args.width, rows_cpu,
                     "the reference's compiled cpu_EmbeddingLookup / cpu_SGDOptimizerSparseUpdate:"
                     if kind == "reference" else "oracle.c:",
                     cpu.num_threads(), 1e3 * el / steps),
    }


def cache_tier(args, table, ids_dev, out, grad, dev):
    """Secondary line: the same batches through the HET cache tier of configs[1] (LRU, limit 0.1 x rows,
    bound 100; cache.cc:60-257): one embedding_lookup + one embedding_update per batch, keys and
    gradients resident in HBM.  Runs last (it takes the table over as its server store).  The other two policies of the
    reference (lfu_cache.cc, lfuopt_cache.cc) are timed the same way beside it (`lfu`, `lfuopt`)."""
    res = _cache_tier_policy(args, table, ids_dev, out, grad, dev, "LRU")
    if not args.no_cache_prefill and os.environ.get("HA_CACHE_BENCH_OTHER_POLICIES", "1") == "1":
        for pol in os.environ.get("HA_CACHE_BENCH_POLICIES", "LFU,LFUOpt").split(","):
            try:
                r = _cache_tier_policy(args, table, ids_dev, out, grad, dev, pol)
                res[pol.lower()] = {k: r[k] for k in ("us_per_step", "value", "flow", "unique_miss_rate", "evicted_lines_per_step",
                                                      "cache_full")}
                res[pol.lower()]["call_by_call_us_per_step"] = r["call_by_call"]["us_per_step"]
                if r.get("planned"):
                    res[pol.lower()]["enqueue_us_per_step"] = r["planned"]["enqueue_us_per_step"]
            except Exception as e:      # a secondary figure never takes the line down
                res[pol.lower()] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


_CACHE_ROW_STREAM = None


def _planned_timed(c, blocks, outs16, grads16, side, nwarm, ntimed, first_block, accept_us):
    """Blocks of 16 planned lookup + update pairs of cache `c` on the row stream `side`: nwarm untimed, ntimed timed by HIP events
    on the row stream (plain launches, the host enqueues ahead; block b + 1 is planned when block b starts).  -> (us per pair,
    host enqueue us per pair, tries).  The planning stream: the library's (cache.py: one per device and row stream, picked by
    herald_amd.streams.pick_side_stream); if the pairs then take longer than `accept_us` -- an UNLUCKY pair of hardware queues,
    which one order of stream creation in four produces and the library's probe does not always see: the row launches take three
    times as long, profiles/r06/stream_pairs.txt -- the measurement is repeated beside another planning stream (at most three
    more: high, normal, high priority).  Every try goes on from the cache's state and the block where the one before stopped."""
    import gc
    GS = len(outs16)
    tries, best = [], None
    cands = [None, -1, 0, -1]
    at = first_block
    for prio in cands:
        if os.environ.get("HA_BENCH_TRACE") == "1":
            print("[bench]   planned try prio=%s at=%d" % (prio, at), file=sys.stderr, flush=True)
        plan_side = None if prio is None else torch.cuda.Stream(device=side.device, priority=prio)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            c.plan_block(blocks[at % len(blocks)], side=plan_side)
            gc.collect()
            gc.disable()        # (a collector pass in the middle of 6 ms of enqueueing is a host stall the device then waits out)
            t0 = None
            for b in range(nwarm + ntimed):
                if b == nwarm:
                    e0.record(side)
                    t0 = time.perf_counter()
                if b + 1 < nwarm + ntimed:
                    c.plan_block(blocks[(at + b + 1) % len(blocks)], side=plan_side)
                c.run_planned_pairs(outs16, grads16)
            e1.record(side)
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            gc.enable()
        at += nwarm + ntimed
        if os.environ.get("HA_BENCH_TRACE") == "1":
            try:
                st = (c.cache if hasattr(c, "cache") else c).state()
                print("[bench]   state after the try: size %d free %d" % (st["size"], st["free_slots"]), file=sys.stderr, flush=True)
            except Exception as ex:      # noqa: BLE001
                print("[bench]   state after the try RAISED: %s" % ex, file=sys.stderr, flush=True)
        us = 1e3 * e0.elapsed_time(e1) / (ntimed * GS)
        tries.append(round(us, 2))
        if best is None or us < best[0]:
            best = (us, 1e6 * t_enq / (ntimed * GS))
        if us <= accept_us:
            break
    return best[0], best[1], tries


def _cache_tier_policy(args, table, ids_dev, out, grad, dev, policy):
    from herald_amd import cache as hcache
    if os.environ.get("HA_BENCH_TRACE") == "1":
        print("[bench]  cache tier policy %s" % policy, file=sys.stderr, flush=True)
    n = ids_dev.shape[1]
    versions = torch.zeros(args.rows, dtype=torch.int64, device=dev)
    hcache.register_table(0, table, versions)
    limit = int(0.1 * args.rows)
    c = hcache.CacheSparseTable(limit, args.rows, args.width, 0, policy, bound=100, max_batch=n, device=dev)
    nb = min(ids_dev.shape[0], 512)
    prefilled = 0
    if not args.no_cache_prefill:
        # Fill the cache to its limit with looked-up AND updated lines (dirty, as every line of a training run is), so
        # that the timed pairs run in the steady state: each lookup evicts as many lines as it misses and each update
        # pushes those to the table.  (Left as the warm-up leaves it, the cache holds ~2 % of its limit and evicts nothing.)
        base = torch.arange(n, device=dev)
        for lo in range(0, limit + n, n):
            kk = ((base + lo) % args.rows).to(ids_dev.dtype)
            c.embedding_lookup(kk, out)
            c.embedding_update(kk, grad, same_as_lookup=True)
            prefilled += n
        torch.cuda.synchronize()

    use_ahead = os.environ.get("HA_CACHE_BENCH_AHEAD", "0") == "1"
    # the ids of a block of 16 batches are known at its start (the lookahead the headline's work-queue step requires as
    # well): their sorts are ONE launch at the head of the block (HA_CACHE_BENCH_BLOCK=0: every lookup sorts its own batch)
    use_block = os.environ.get("HA_CACHE_BENCH_BLOCK", "1") == "1" and not use_ahead

    def step(k, ahead=True):
        c.embedding_lookup(ids_dev[k % nb], out)
        if ahead and use_ahead:       # the loader has the next batch's ids: their sort runs beside this batch's update
            c.prefetch_keys(ids_dev[(k + 1) % nb])
        c.embedding_update(ids_dev[k % nb], grad, same_as_lookup=True)

    # the ten launches of a lookup + update pair are replayed from hipGraphs of 16 pairs (the calls enqueue
    # kernels only, nothing is read back while the counters are off)
    GS = 16
    # (ONE row stream for the three policies' measurements: a stream created later may share a hardware queue with other
    # streams of the process -- the third policy's row launches then took 3x as long, profiles/r06/cache_tier_third_instance.txt)
    global _CACHE_ROW_STREAM
    if _CACHE_ROW_STREAM is None:
        _CACHE_ROW_STREAM = torch.cuda.Stream(device=dev)
    side = _CACHE_ROW_STREAM
    c.cache.stream = side
    for k in range(64):
        if use_block and k % GS == 0:
            c.prefetch_keys_batch([ids_dev[j % nb] for j in range(k, k + GS)])
        step(k)
    torch.cuda.synchronize()
    graphs = []
    for g0 in range(0, nb, GS):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            if use_block:
                c.prefetch_keys_batch([ids_dev[j % nb] for j in range(g0, g0 + GS)])
            for k in range(g0, g0 + GS):
                step(k, ahead=k + 1 < g0 + GS)      # (a fork must join inside its graph)
        graphs.append(g)
    torch.cuda.synchronize()
    steps = 256
    with torch.cuda.stream(side):
        for i in range(4):
            graphs[i % len(graphs)].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps // GS):
            graphs[(4 + i) % len(graphs)].replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    # THE figure (all three policies): the planned flow (csrc/cache_block.hip).  The ids of a block of 16 batches are known at its start (as
    # for the headline's work-queue step): the bookkeeping of block b + 1 -- hits, misses, slots, evictions, update counters, the
    # bounded push -- runs on a side stream beside the rows of block b; every lookup and every update is ONE launch.  Plain
    # launches, a block's 16 pairs enqueued by one library call; same cache, same steady state, continuing where the
    # call-by-call measurement above stopped.
    planned = None
    if os.environ.get("HA_CACHE_BENCH_PLANNED", "1") == "1":
        blocks = [[ids_dev[j % nb] for j in range(g0, g0 + GS)] for g0 in range(0, nb, GS)]
        outs16, grads16 = [out] * GS, [grad] * GS
        nwarm, ntimed = 4, steps // GS
        pus, enq_us, tries = _planned_timed(c, blocks, outs16, grads16, side, nwarm, ntimed, 0, 0.8 * 1e6 * el / steps)
        planned = {"us_per_step": pus, "value": n / (pus * 1e-6), "steps": ntimed * GS,
                   "launches_per_pair": 2, "enqueue_us_per_step": enq_us, "us_per_step_by_planning_stream_tried": tries,
                   "timed": "HIP events on the row stream around %d blocks of 16 pairs (plain launches, the host enqueues ahead "
                            "of the device; each block's bookkeeping -- 4 launches -- on a side stream beside the rows of the "
                            "block before, inside the timed region)" % ntimed,
                   "bookkeeping": "a block of 16 batches ahead, on a side stream (4 launches per block)"}
    c.perf_enabled(True)       # the counters are read back per call: outside the timed loop
    for k in range(32):        # batches the cache has not seen yet
        b = (nb + k) % ids_dev.shape[0]
        c.embedding_lookup(ids_dev[b], out)
        c.embedding_update(ids_dev[b], grad, same_as_lookup=True)
    torch.cuda.synchronize()
    pulls = [r for r in c.perf if r["type"] == "Pull"]
    pushes = [r for r in c.perf if r["type"] == "Push"]
    miss = float(np.mean([r["num_miss"] / max(r["num_unique"], 1) for r in pulls])) if pulls else None
    evict = float(np.mean([r["num_evict"] for r in pushes])) if pushes else None
    head = {"value": n * steps / el, "us_per_step": 1e6 * el / steps}
    if planned is not None:
        head = {"value": planned["value"], "us_per_step": planned["us_per_step"]}
    return {"value": head["value"], "unit": "rows/s", "us_per_step": head["us_per_step"], "policy": policy,
            "flow": "planned (ha_cache_plan_block: bookkeeping a block ahead, one launch per lookup / update)" if planned is not None
                    else "call by call",
            "planned": planned,
            "call_by_call": {"us_per_step": 1e6 * el / steps, "value": n * steps / el, "steps": steps},
            "limit_rows": limit, "bound": 100, "steps": steps, "unique_miss_rate": miss,
            "evicted_lines_per_step": evict, "cache_full": bool(pulls and pulls[-1]["is_full"]),
            "prefilled_keys": prefilled,
            "note": "HET cache tier in front of the same HBM-resident table: lookup + update per batch; cache filled to its limit "
                    "before the timed pairs; `call_by_call`: %s, replayed from hipGraphs of 16 pairs; not part of `value`"
                    % ("4 launches per pair + one launch per block of 16 batches that sorts their keys (the ids are known a "
                       "block early, as for the headline's work-queue step)" if use_block else
                       "5 launches per pair" + (", the next batch's sort forked beside the update" if use_ahead else "")),
            "keys_sorted": "per block of 16 batches" if use_block else "per lookup"}


def laia_scheduler(args):
    """Secondary measurement (scheduler us per 4096-sample global batch, configs[3] shape): the laia
    LaiaScheduler for 4 workers x mini batch 1024, 26 tables over the full key space, cache_size = 0.1 x rows.  With a
    cache of at least one global batch of rows the scheduler state lives on the GPU (MiniLRU snapshots, assignment,
    sorted-unique key lists: csrc/laia.hip, laia_next_device) and the host parts are zero; HA_LAIA_HOST=1 measures the
    host-snapshot mode."""
    from herald_amd import laia as hlaia, synth
    W, mini_bs, T, batch_num = 4, 1024, args.fields, 96
    per = 256
    need = W * mini_bs * batch_num + 1000
    parts = [synth.criteo_batch(per, step=5000 + s, rows=args.rows, nfields=T) for s in range((need + per - 1) // per)]
    samples = np.concatenate(parts, axis=0)[:need].astype(np.uint64)
    ahead = os.environ.get("HA_LAIA_AHEAD", "1") == "1"
    plug = hlaia.native_plugin() if os.environ.get("HA_LAIA_PYTHON_THREAD") != "1" else None
    if plug is not None:
        # the reference's surface: the pybind11 module `laia_cache`, its scheduler loop in a C++ thread (launch(),
        # laia_scheduler.cc:115-169)
        s = plug.LaiaScheduler()
        s.start(samples, samples.shape[0], T, 1, mini_bs, batch_num, W, 0, int(0.1 * args.rows), 16, 24)
    else:
        s = hlaia.LaiaScheduler()
        s.start(samples, samples.shape[0], T, 1, mini_bs, batch_num, W, 0, int(0.1 * args.rows), 16, 24, key_limit=args.rows,
                ahead=ahead)
    while True:
        item = s.pop_arrays()
        if len(item) == 1 and int(item[0]) == 0:
            break
    tm = s.timing()
    s.close()
    # THE figure: the scheduler thread's period -- one (plan, dist) pair per this many microseconds, the reference's launch()
    # loop as a whole (laia_scheduler.cc:115-169: the library call + handing the pair to the queue).  `in_call_us` is the part
    # spent inside the library call (with `one_batch_ahead` the device works on batch k+1 behind it).
    # (steady state: without the first 8 batches -- the first call allocates and clears the scheduler's device state, ~1.5 ms;
    # the figure over all batches is beside it)
    steady = tm.get("steady_thread_wall_us_per_batch")
    return {"us_per_global_batch": steady if steady is not None else tm["thread_wall_us_per_batch"],
            "in_call_us_per_global_batch": tm.get("steady_us_per_batch", tm["us_per_batch"]),
            "steady_from_batch": tm.get("steady_from_batch"),
            "us_per_global_batch_incl_first_batches": tm["thread_wall_us_per_batch"],
            "in_call_us_per_global_batch_incl_first_batches": tm["us_per_batch"],
            "global_batch_samples": W * mini_bs, "workers": W,
            "thread_wall_us_per_global_batch": steady if steady is not None else tm["thread_wall_us_per_batch"],
            "one_batch_ahead": ahead,
            "hand_off": "arrays (pop_arrays); pop() converts to the reference's Python lists in the caller's thread",
            "scheduler_thread": "C++ thread of the laia_cache plugin" if plug is not None else "Python thread (herald_amd.laia)",
            "tables": T, "cache_size": int(0.1 * args.rows), "batches": tm["batches"],
            "host_assign_us": tm["host_assign_us"], "host_snapshot_us": tm["host_snapshot_us"],
            "gpu_and_transfer_us": tm["gpu_and_transfer_us"],
            # inside the call, by what the host does: enqueueing the NEXT batch's launches, waiting for this batch's sequence
            # word (the device), copying dist and plan out of the pinned mirror
            "in_call_issue_us": tm.get("steady_issue_us", tm.get("issue_us")),
            "in_call_wait_us": tm.get("steady_wait_us", tm.get("wait_us")),
            "in_call_unpack_us": tm.get("steady_unpack_us", tm.get("unpack_us")),
            "mode": "host snapshots" if os.environ.get("HA_LAIA_HOST") == "1" else "device-resident state",
            "note": "the scheduler runs ahead of training in its own thread; not part of `value`"}


def queue_leg(table, ids_rows, grads, outs, n, block, steps, warm, dev, sync="flags"):
    """The work-queue step of the headline at another BLOCK size (= id lookahead 3 x block), long-run form: plain launches, a
    block's steps enqueued by one library call, the preparation beside them, HIP events around `steps` steps after `warm`.
    -> (device ms per step, enqueue ms per step)."""
    from herald_amd import ops
    pipe = ops.QueueStepPipeline(table, n, LR, block=block, overlap=True, sync=sync)
    s = torch.cuda.Stream(device=dev)
    LA, Bk, nb, nbuf = pipe.LOOKAHEAD, pipe.block, len(ids_rows), len(grads)
    ids_of = lambda j: ids_rows[j % nb] if j >= 0 else None
    packs = {}

    def chunks(k0, count):
        k, end = k0, k0 + count
        while k < end:
            ln = min(end, (k // Bk + 1) * Bk) - k
            yield k, ln
            k += ln

    for k, ln in list(chunks(0, warm)) + list(chunks(warm, steps)):      # arguments converted once, outside the timed region
        key = (k % pipe.ROTATION, k % nb, ln)
        if key not in packs:
            bs = [(k + i) % nb for i in range(ln)]
            packs[key] = pipe.apply_steps_call(k, [grads[x % nbuf] for x in bs], [outs[(x + 1) % nb % nbuf] for x in bs], s, n)

    def run(k0, count):
        for k, ln in chunks(k0, count):
            if k % Bk == 0:
                pipe.prepare_block(k // Bk, ids_of, stream=s)
            packs[(k % pipe.ROTATION, k % nb, ln)](k)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        for c in range(-LA, 0):
            if c % Bk == 0:
                pipe.prepare_block(c // Bk, ids_of, stream=s)
        pipe.apply(-1, None, outs[0], stream=s, n_cur=0, n_next=n)
        run(0, warm)
        e0.record(s)
        t0 = time.perf_counter()
        run(warm, steps)
        t_enq = time.perf_counter() - t0
        e1.record(s)
    torch.cuda.synchronize()
    if pipe.overflowed():
        raise RuntimeError("the work-queue engine raised its sticky error word")
    ms = e0.elapsed_time(e1) / steps
    pipe.close()
    return ms, 1e3 * t_enq / steps


def lookahead_sweep(table, ids_rows, grads, outs, n, dev, headline_block, headline_ms):
    """config A's step at the id lookahead a caller can actually supply: block 1 / 2 / 4 = 3 / 6 / 12 batches ahead (the
    reference's loader keeps a 3-deep ring one batch ahead, python/hetu/dataloader.py:63-98; laia's queue is 5 deep,
    python/hetu/laia/laia_dataloader.py:108-114 -- block 1), beside the headline's block."""
    res = {}
    for blk in (1, 2, 4):
        # (blocks of fewer than 8 steps are ordered by events: QueueStepPipeline's rule -- a launch that polls for its queue
        # keeps the queue's builder off the chip, csrc/qstep.hip)
        ms, enq = queue_leg(table, ids_rows, grads, outs, n, blk, steps=768, warm=96, dev=dev, sync="flags")
        best = {"lookahead_batches": 3 * blk, "device_ms_per_step": ms, "enqueue_ms_per_step": enq,
                "ms_per_step": max(ms, enq), "host_bound": enq > ms, "stream_sync": "events"}
        res["block_%d" % blk] = best
    res["block_%d" % headline_block] = {"lookahead_batches": 3 * headline_block, "ms_per_step": headline_ms,
                                        "note": "the headline line itself"}
    res["note"] = ("long-run form (768 steps behind 96, no gate): the larger of the device's time (HIP events) and the host's "
                   "enqueue time per step; below blocks of 4 the preparation's own latency (~40 us per block: plans, then queues, "
                   "two dependent launches of single workgroups) is the bound, not the steps.  A "
                   "caller with 1 / 3 batches of lookahead is better served by the bit-exact one-launch engines "
                   "(`bit_exact_engine`: ha_sgd_push_pull_*, 1 batch ahead; ha_step_*, 3 ahead)")
    return res


def bit_exact_leg(table, ids_dev, grads, outs, n, dev, steps=256, warm=64):
    """The bit-exact one-launch engine (ha_sgd_push_pull_*: the reference's serial chain for every run length, ids one
    batch ahead) on the same table and batches: what the headline's tolerance class (keys with 16+ occurrences) buys."""
    from herald_amd import ops
    nb, nbuf = ids_dev.shape[0], len(grads)
    s = torch.cuda.Stream(device=dev)
    plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
    pends = [ops.PendingTable(dev), ops.PendingTable(dev)]

    def step(k):
        b = k % nb
        bn = (b + 1) % nb
        ops.sgd_push_pull(table, plans[k % 2], grads[b % nbuf], LR, pends[k % 2], ids_dev[bn], plans[(k + 1) % 2],
                          pends[(k + 1) % 2], next_out=outs[bn % nbuf], stream=s)

    G = 32
    graphs = []
    with torch.cuda.stream(s):
        ops.lookup_sort_pend(table, ids_dev[0], plans[0], pends[0], out=outs[0], stream=s)
        for k in range(2):
            step(k)
    torch.cuda.synchronize()
    for g0 in range(2, 2 + warm + steps, G):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for k in range(g0, g0 + G):
                step(k)
        graphs.append(g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        for i in range(warm // G):
            graphs[i].replay()
        e0.record(s)
        for i in range(warm // G, (warm + steps) // G):
            graphs[i].replay()
        e1.record(s)
    torch.cuda.synchronize()
    if plans[0].handoff_timed_out() or plans[1].handoff_timed_out():
        raise RuntimeError("the in-launch hand-off timed out")
    return e0.elapsed_time(e1) / steps


def cold_tier(args, dev):
    """Secondary line, BASELINE configs[4] shape on one GPU: avazu-like power-law ids (22 fields, uint64),
    d=64, the table in PINNED HOST DRAM (HostStore), an LRU hot tier of 0.1 x rows lines in HBM in front of
    it (HET cache in remote-store mode): misses / stale lines are staged from the host over PCIe by the
    owner-side kernels on a copy stream, pushed lines are written back the same way."""
    from herald_amd import cache as hcache, remote_store, synth
    rows, width, fields, bs = args.cold_rows, 64, 22, args.batch
    n = bs * fields
    store = remote_store.HostStore(rows, width, dev)
    limit = int(0.1 * rows)
    c = hcache.LRUCache(limit, rows, width, node_id=-7, max_batch=n, device=dev)
    c.bind_remote(store)
    c.pull_bound = c.push_bound = 100
    nb = 320     # 20 graphs of 16 batches: four replayed as warm-up, sixteen (not seen before) timed
    ids = [torch.from_numpy(synth.criteo_batch(bs, 9000 + b, rows=rows, nfields=fields).reshape(-1)).to(dev)
           for b in range(nb)]
    out = torch.empty((n, width), dtype=torch.float32, device=dev)
    grad = torch.randn((n, width), dtype=torch.float32, device=dev) * 1e-3

    def step(k):
        c.embedding_lookup(ids[k % nb], out)
        c.embedding_update(ids[k % nb], grad, same_as_lookup=True)

    # Neither call reads anything back (the store is on this device: request and outbox are handed over padded),
    # so the pairs are replayed from hipGraphs of 16 like the HBM cache tier's
    GS = 16
    side = torch.cuda.Stream(device=dev)
    c.stream = side
    for k in range(128):
        step(k)
    torch.cuda.synchronize()
    graphs = []
    for g0 in range(0, nb, GS):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for k in range(g0, g0 + GS):
                step(128 + k)
        graphs.append(g)
    torch.cuda.synchronize()
    steps = 256
    with torch.cuda.stream(side):
        for i in range(4):
            graphs[i % len(graphs)].replay()
        torch.cuda.synchronize()
        store.traffic(reset=True)
        t0 = time.perf_counter()
        for i in range(steps // GS):
            graphs[(4 + i) % len(graphs)].replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    tr = store.traffic()
    # THE figure: the same store addressed DIRECTLY -- the cache bound to the pinned host table (device-visible) and the HBM
    # versions, lookup + update through the planned flow (csrc/cache_block.hip): the staleness-bounded pull reads the store's row
    # over PCIe in the lookup's one launch, a pushed / evicted line's read-modify-write of its store row rides in the update's.
    planned = None
    if os.environ.get("HA_COLD_PLANNED", "1") == "1":
        try:
            c2 = hcache.LRUCache(limit, rows, width, node_id=-8, max_batch=n, device=dev)
            c2.bind_store(store.table, store.versions)
            c2.pull_bound = c2.push_bound = 100
            c2.stream = side
            blocks = [[ids[j % nb] for j in range(g0, g0 + GS)] for g0 in range(0, nb, GS)]
            outs16, grads16 = [out] * GS, [grad] * GS
            nwarm, ntimed = 12, steps // GS       # (the warm-up: the 128 + 64 batches the remote-protocol measurement had seen)
            pus, _, tries = _planned_timed(c2, blocks, outs16, grads16, side, nwarm, ntimed, 0, 0.5 * 1e6 * el / steps)
            planned = {"us_per_step": pus, "value": n / (pus * 1e-6), "steps": ntimed * GS,
                       "us_per_step_by_planning_stream_tried": tries,
                       "flow": "planned (ha_cache_plan_block), the cache bound to the pinned host table: pulls and pushes cross PCIe "
                               "inside the lookup's / the update's one launch",
                       "key_sequence": "the remote-protocol measurement's, batch for batch (128 + 64 warm-up batches, then the same "
                                       "256: the same rows pulled and lines pushed per step) for the first planning stream tried; a "
                                       "further try goes on where the one before stopped"}
        except Exception as e:      # a secondary figure never takes the line down
            planned = {"error": "%s: %s" % (type(e).__name__, e)}
    head_us, head_val = 1e6 * el / steps, n * steps / el
    if planned and "us_per_step" in planned:
        head_us, head_val = planned["us_per_step"], planned["value"]
    return {"value": head_val, "unit": "rows/s", "us_per_step": head_us, "planned": planned,
            "remote_protocol": {"us_per_step": 1e6 * el / steps, "value": n * steps / el},
            "workload": "power-law ids, %d fields, bs=%d, d=%d; %d-row fp32 table (%.1f GiB) in pinned host DRAM; "
                        "LRU hot tier of %d lines in HBM (bound 100)" % (fields, bs, width, rows,
                                                                        rows * width * 4 / 2 ** 30, limit),
            "hot_tier_hit_rate": 1.0 - tr["rows_pulled"] / max(tr["keys_synced"], 1),
            "rows_pulled_per_step": tr["rows_pulled"] / steps, "lines_pushed_per_step": tr["lines_pushed"] / steps,
            "pcie_GBps": tr["pcie_bytes"] / el / 1e9, "steps": steps,
            "note": "`remote_protocol`: lookup + update per batch through the remote-store protocol (request / inbox / outbox, "
                    "handed over padded: no host read-back), replayed from hipGraphs of 16 pairs; `planned`: see its `flow`; "
                    "hit rate / rows pulled / lines pushed / PCIe rate are the remote-protocol run's counters; not part of `value`"}


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))          # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from herald_amd import ops

    if world > 1 or os.environ.get("HA_FORCE_SHARDED") == "1":
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        from herald_amd import sharded_bench
        return sharded_bench.run(args, rank, world, dev, cpu_baseline_fn=cpu_baseline)

    n = args.batch * args.fields
    one = args.launches == 1
    if args.engine is None:
        args.engine = {None: "queue", 1: "handoff", 3: "forward"}[args.lookahead]
    queue = one and args.engine == "queue"
    if args.graph_steps is None:
        args.graph_steps = 1 if queue else 32
    G = max(1, args.graph_steps)
    ahead2 = one and args.engine in ("forward", "queue")
    # graphs never straddle the wrap-around of the batch list; ha_step_* rotates four plans and four key tables,
    # so the list is also cut to a multiple of 4 (a graph then depends on k % nb only)
    # buffers (and side streams) of step k and step k + rot_ring are the same
    rot_ring = (4 * (1 if args.queue_serial else args.queue_block)) if queue else 4
    # Step numbers start at K0, chosen so that step K0 + 16 starts a block whatever the block size: a window of
    # `--steps 20 --warmup 5` then holds exactly one block boundary (one preparation of a block on the side stream, one
    # wait of the timed stream for it), as it did when blocks were 16 steps -- a short run times its share of the pipeline's
    # preparation work, not a stretch between two boundaries.
    K0 = (args.queue_block - 16) if (queue and not args.queue_serial and args.queue_block > 16) else 0
    period = (G * rot_ring // math.gcd(G, rot_ring)) if ahead2 else G
    if args.distinct_batches >= period:
        args.distinct_batches -= args.distinct_batches % period
    elif ahead2:
        raise SystemExit("this engine needs --distinct-batches >= lcm(graph steps, %d) = %d" % (rot_ring, period))
    ids_host, uniq_counts = make_batches(args, rank, world)
    ids_dev = torch.from_numpy(ids_host).to(dev)
    table = init_table(args.rows, args.width, dev)
    nbuf = max(1, args.grad_buffers)
    gen = torch.Generator(device=dev)
    gen.manual_seed(456)
    grads = [torch.randn((n, args.width), dtype=torch.float32, device=dev, generator=gen) for _ in range(nbuf)]
    outs = [torch.empty((n, args.width), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    nb = ids_dev.shape[0]
    main_prio = 0
    if os.environ.get("HA_BENCH_MAIN_PRIO") == "high":
        try:
            main_prio = min(torch.cuda.Stream.priority_range())
        except Exception:      # noqa: BLE001
            main_prio = 0
    main_s = torch.cuda.Stream(device=dev, priority=main_prio)
    before_chunk = None
    if ahead2:
        # One step = ONE launch (ha_step_f32ids): SGD apply of batch k, the rows of batch k+1 (forwarded from
        # the applying waves where both batches name a row, copied from the table otherwise), plan finish of
        # batch k+2, stable sort of batch k+3.  Every step applies one batch, looks one batch up, finishes one
        # plan and sorts one batch, as before; sorting the first three batches and looking the first one up is
        # the prologue (untimed).
        if queue:
            # One step = ONE launch on the timed stream (ha_qapply: the items of step k -- SGD apply of batch k, rows of
            # batch k+1 -- from the queue prepared for it).  The preparation runs a BLOCK of steps at a time on a side
            # stream, beside the steps of the block before: at the start of block b, one launch plans the batches of
            # block b+2 (one workgroup each) and one builds the queues of block b+1 (two workgroups each).  All of it
            # happens inside the timed region, every block; --queue-serial: block = 1 on the timed stream.
            # sync="flags": nothing but apply launches on the timed stream -- every queue carries the epoch of its step (the
            # apply checks it before its first item) and the last launch of a block completes the event the side stream waits
            # for (no event record / wait between two steps); hipGraph replays (--graph-steps > 1) keep the event pair
            qsync = "events" if (args.graph_steps > 1 or os.environ.get("HA_QSYNC") == "events") else "flags"
            pipe = ops.QueueStepPipeline(table, n, LR, block=args.queue_block, overlap=not args.queue_serial, sync=qsync)
            LA, Bk = pipe.LOOKAHEAD, pipe.block
            G = Bk                       # graphs are cut at block starts (the side work is enqueued between them)
            ids_rows = [ids_dev[i] for i in range(nb)]        # the batches as tensors of their own, sliced once
            ids_of = lambda j: ids_rows[j % nb] if j >= K0 else None          # the stream of batches starts at step K0
            with torch.cuda.stream(main_s):
                for c in range(K0 - LA, K0):
                    if c % Bk == 0:
                        pipe.prepare_block(c // Bk, ids_of, stream=main_s)
                pipe.apply(K0 - 1, None, outs[0], stream=main_s, n_cur=0, n_next=n)

            def before_chunk(k):
                if k % Bk == 0:
                    pipe.prepare_block(k // Bk, ids_of, stream=main_s)

            calls = {}

            def step(k):
                b = k % nb
                key = (k % rot_ring, b % nbuf, (b + 1) % nb % nbuf)
                fn = calls.get(key)
                if fn is None:      # arguments converted once per (rotation phase, buffers)
                    fn = calls[key] = pipe.apply_call(k, grads[b % nbuf], outs[(b + 1) % nb % nbuf], main_s, n, n,
                                                     sized=args.graph_steps <= 1)
                fn(k)
        else:
            pipe = ops.StepPipeline(table, n, LR)
            with torch.cuda.stream(main_s):
                pipe.reset(stream=main_s)
                pipe.launch(-3, 0, None, 0, None, 0, ids_dev[0], stream=main_s)
                pipe.launch(-2, 0, None, 0, None, n, ids_dev[1 % nb], stream=main_s)
                pipe.launch(-1, 0, None, n, outs[0], n, ids_dev[2 % nb], stream=main_s)

            def step(k):
                b = k % nb
                pipe.launch(k, n, grads[b % nbuf], n, outs[(b + 1) % nb % nbuf], n, ids_dev[(b + 3) % nb], stream=main_s)
        plans = pipe.plans
    elif one:
        # One step = ONE launch (ha_sgd_push_pull_f32ids): the backward of batch k (fused SGD apply + plan
        # finish) beside the forward of batch k+1 (gather + stable sort), rows both batches touch handed
        # over inside the launch.  Every step applies one batch and looks one batch up, as before; the
        # lookup of the very first batch is the prologue below (untimed, like the table fill).
        plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
        pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
        with torch.cuda.stream(main_s):
            ops.lookup_sort_pend(table, ids_dev[0], plans[0], pends[0], out=outs[0], stream=main_s)

        def step(k):
            b = k % nb
            bn = (b + 1) % nb
            ops.sgd_push_pull(table, plans[k % 2], grads[b % nbuf], LR, pends[k % 2], ids_dev[bn],
                              plans[(k + 1) % 2], pends[(k + 1) % 2], next_out=outs[bn % nbuf], stream=main_s)
    else:
        plan = ops.IndexPlan(n, dev)

        # One step = two launches on one stream:
        #   forward : gather(ids) + stable sort of ids            (ha_lookup_sort_f32ids)
        #   backward: fused SGD apply + plan finish (uniq/counts) (ha_sgd_apply_finish)
        # The table dependency gather(k) -> apply(k) -> gather(k+1) is the stream order.
        # The backward launch is handed the ids of the NEXT batch (resident one step ahead, as the
        # reference's prefetching data loader provides them): its idle waves touch the rows the next
        # lookup will gather.
        def step(k):
            b = k % nb
            ops.lookup_sort(table, ids_dev[b], plan, out=outs[b % nbuf], stream=main_s)
            ops.sgd_apply_finish(table, plan, grads[b % nbuf], LR, stream=main_s, next_ids=ids_dev[(b + 1) % nb])

    # Batch b always uses gradient / output buffer b % nbuf, so a captured graph depends on b only
    # (nb is even, so the plan / pending-table parity of step k is that of batch k % nb as well).
    use_graph = G > 1 and args.graph_steps > 1      # --graph-steps 1: plain launches (the PMC passes)
    graphs = {}
    step_chunk = None
    if queue and not use_graph:
        packs = {}

        def make_pack(k, ln):
            bs = [(k + i) % nb for i in range(ln)]
            return pipe.apply_steps_call(k, [grads[x % nbuf] for x in bs], [outs[(x + 1) % nb % nbuf] for x in bs], main_s, n)

        def step_chunk(k, ln):
            key = (k % rot_ring, k % nb, ln)
            fn = packs.get(key)
            if fn is None:      # arguments converted once per (rotation phase, first batch, length)
                fn = packs[key] = make_pack(k, ln)
            fn(k)

    def chunks(k0, count):
        """[k0, k0+count) cut at the multiples of G: (first step, length) pieces of at most G steps."""
        k, end = k0, k0 + count
        while k < end:
            ln = min(end, (k // G + 1) * G) - k
            yield k, ln
            k += ln

    # plans / pending tables (one-launch step) and plans / key tables (lookahead step) rotate with the STEP
    # index, so a captured graph is only valid for steps with the same rotation phase
    rot = rot_ring if ahead2 else (2 if one else 1)

    def graph_for(k, ln):
        key = (k % nb, k % rot, ln)
        if key not in graphs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_s):
                for j in range(k, k + ln):
                    step(j)
            graphs[key] = g
        return graphs[key]

    def run(k0, count):
        """Exactly `count` steps starting at step index k0; returns the number of graph replays."""
        replays = 0
        with torch.cuda.stream(main_s):
            for k, ln in chunks(k0, count):
                _t = time.perf_counter()
                if before_chunk is not None:
                    before_chunk(k)
                if os.environ.get("HA_BENCH_TRACE"):
                    sys.stderr.write("before_chunk(%d) %.1f us\n" % (k, (time.perf_counter() - _t) * 1e6))
                    _t = time.perf_counter()
                if use_graph:
                    graph_for(k, ln).replay()
                    replays += 1
                elif step_chunk is not None:
                    step_chunk(k, ln)          # the steps of a chunk enqueued by one library call
                    if os.environ.get("HA_BENCH_TRACE"):
                        sys.stderr.write("step_chunk(%d, %d) %.1f us\n" % (k, ln, (time.perf_counter() - _t) * 1e6))
                else:
                    for j in range(k, k + ln):
                        step(j)
        return replays

    # The W warm-up steps are split: W - P of them, a synchronisation, and the last P (`--pre-roll`, default 2) right in
    # front of the timed region on the same stream.  The timed region is exactly K steps between two events, with a
    # synchronisation before the sequence and after it.  What the split and the gate below remove is not work but idle
    # time that a short run otherwise measures: (1) the first launch after an idle period takes 95-135 us and the second
    # 19-31 us (profiles/r02/timelines/first_steps_after_idle.txt) -- they are now warm-up launches; (2) the device
    # reaching the first event before the host has enqueued the K steps behind it -- the stream is held by a gate kernel
    # (ha_stream_gate) that the host opens once pre-roll, events and timed steps are all queued.
    wu = args.warmup
    pre = min(max(args.pre_roll, 0), wu)
    if step_chunk is not None:      # convert the arguments of everything the warm-up and the timed region enqueue
        for k, ln in list(chunks(K0, wu - pre)) + list(chunks(K0 + wu - pre, pre)) + list(chunks(K0 + wu, args.steps)):
            if (k % rot_ring, k % nb, ln) not in packs:
                packs[(k % rot_ring, k % nb, ln)] = make_pack(k, ln)
    if use_graph:     # capture everything the warm-up and the timed region replay, before either runs
        torch.cuda.synchronize()
        for k, ln in list(chunks(K0, wu - pre)) + list(chunks(K0 + wu - pre, pre)) + list(chunks(K0 + wu, args.steps)):
            graph_for(k, ln)
        torch.cuda.synchronize()
    run(K0, wu - pre)
    torch.cuda.synchronize()
    # plain launches allocate a few Python objects per step; a generation-2 pass of the cyclic garbage collector over
    # everything torch has imported takes 35-65 ms -- two thousand steps -- and would land in the timed region of every
    # other short run: collect now, keep the collector off until the steps are enqueued
    import gc
    gc.collect()
    gc.disable()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    warm = [torch.zeros(1 << 27, dtype=torch.float32, device=dev) for _ in range(2)] if args.clock_warm > 0 else None
    torch.cuda.synchronize()
    gate = None
    # the gate holds the stream until everything behind it is enqueued: with plain launches that is one queue packet per
    # step, so only short runs are gated (a full hardware queue behind a closed gate would never drain)
    if not args.no_gate and (use_graph or args.steps + pre + args.cache_warm <= 512):
        from herald_amd import _lib as _hl
        gate = torch.zeros(1, dtype=torch.int32).pin_memory()
        with torch.cuda.stream(main_s):
            _hl.check(_hl.load().ha_stream_gate(gate.data_ptr(), main_s.cuda_stream), "ha_stream_gate")
    if args.clock_warm > 0:
        # untimed, behind the gate: device-to-device copies that keep the memory system busy for ~170 us each, so that
        # the pre-roll and the timed steps do not start on clocks that an idle device has dropped
        with torch.cuda.stream(main_s):
            for _ in range(args.clock_warm):
                warm[0].copy_(warm[1])
    if args.cache_warm > 0:
        # untimed, behind the gate, read-only: the rows of the `cache_warm` batches that precede the first pre-roll step in
        # the stream of batches are looked up into a scratch buffer (not the batches of the timed steps: what a long run
        # would have touched on its way here).  The table is not modified; these are not steps.
        scratch = (warm[0] if warm is not None else torch.empty(1 << 27, dtype=torch.float32, device=dev))[:n * args.width] \
            .view(n, args.width)
        first = K0 + wu - pre
        with torch.cuda.stream(main_s):
            for k in range(first - args.cache_warm, first):
                ops.embedding_lookup(table, ids_dev[k % nb], out=scratch, stream=main_s)
    run(K0 + wu - pre, pre)
    t0 = time.perf_counter()
    e0.record(main_s)
    replays = run(K0 + wu, args.steps)
    e1.record(main_s)
    t_enq = time.perf_counter() - t0
    if gate is not None:
        gate[0] = 1             # everything is queued: open the gate
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gc.enable()
    dev_ms = e0.elapsed_time(e1)
    # the K steps' time on the stream (HIP events).  Without a gate the host enqueues while the device runs: if it needed
    # longer than the device, the run is host-bound and the host time counts.  Behind a gate every step is enqueued BEFORE
    # the device starts on the region, so the enqueue time is not part of it by construction (it is reported: `enqueue_ms`;
    # a 20-step window holds one block's preparation calls, ~130 us of host time next to 245 us of device time, and flipped
    # this rule at random; the ungated default run shows what the host sustains: ~6 us per step against 12)
    total_ms = dev_ms if gate is not None else max(dev_ms, t_enq * 1e3)
    ms_per_step = total_ms / args.steps
    rows_per_s = n * args.steps / (total_ms * 1e-3)

    used = [(K0 + wu + k) % nb for k in range(args.steps)]
    u_mean = float(np.mean(uniq_counts[used]))
    fwd_b, bwd_b = algorithmic_bytes(n, u_mean, args.width)

    # ---- per-kernel pass: for every kernel a graph of KL back-to-back launches over KL distinct
    # batches, bracketed by HIP events on the launch stream -> average launch duration (boundary
    # to the next launch included)
    kernels = {}
    roofline = None
    if one:
        # the step IS one launch of ha::step_kernel: its average duration is the HIP-event time of the
        # timed region / K, measured on the launch stream
        kname = "ha::qapply_kernel" if queue else "ha::step_fwd_kernel" if ahead2 else "ha::step_kernel"
        traffic, traffic_src = pmc_traffic(kname)
        n_launch = args.steps           # one launch = one step
        per_launch = 1.0
        dom_bytes = fwd_b + bwd_b
        ach = dom_bytes / (dev_ms / n_launch * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kname + (" (the items of a step: SGD apply of batch k + rows of batch k+1 "
                                                       "from its work queue)" if queue else
                                                       " (SGD apply of batch k, rows of batch k+1, plan finish of "
                                                       "batch k+2, sort of batch k+3)" if ahead2 else
                                                       " (SGD apply + finish of batch k, gather + sort of batch k+1)"),
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": dev_ms / n_launch * 1e3,
                    "avg_launch_source": "HIP events around the timed region on the launch stream / its %d launches "
                                         "(%.2f steps per launch)" % (n_launch, per_launch),
                    "steps_per_launch": per_launch,
                    "algorithmic_bytes_per_launch": dom_bytes}
    elif not args.no_kernel_pass:
        KL = 64
        kplans = [ops.IndexPlan(n, dev).sort(ids_dev[(wu + i) % nb], stream=main_s) for i in range(KL)]
        main_s.synchronize()

        def timed_graph(fn, reps=5):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_s):
                for i in range(KL):
                    fn(i)
            with torch.cuda.stream(main_s):
                g.replay()
                main_s.synchronize()
                a, bq = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(main_s)
                for _ in range(reps):
                    g.replay()
                bq.record(main_s)
                main_s.synchronize()
            return a.elapsed_time(bq) / (reps * KL)

        g_ms = timed_graph(lambda i: ops.lookup_sort(table, ids_dev[(wu + i) % nb], kplans[i],
                                                     out=outs[i % nbuf], stream=main_s))
        a_ms = timed_graph(lambda i: ops.sgd_apply_finish(table, kplans[i], grads[i % nbuf], LR, stream=main_s,
                                                          next_ids=ids_dev[(wu + i + 1) % nb]))
        # `measured_us`: HIP events around a graph of 64 back-to-back launches of this kernel over 64
        # distinct batches (boundary to the next launch included).  `in_step_us` scales both so that
        # they sum to the step time of the timed region (the two launches alternate there and every
        # boundary also pays for what the predecessor left behind); it is derived, not measured.
        share = ms_per_step / (g_ms + a_ms)
        kernels = {
            "fwd_fused_kernel(gather+rank)": {"measured_us": g_ms * 1e3, "in_step_us": g_ms * share * 1e3,
                                              "algorithmic_bytes": fwd_b,
                                              "GBps": fwd_b / (g_ms * 1e-3) / 1e9},
            "bwd_fused_kernel(sgd apply+finish)": {"measured_us": a_ms * 1e3, "in_step_us": a_ms * share * 1e3,
                                                   "algorithmic_bytes": bwd_b,
                                                   "GBps": bwd_b / (a_ms * 1e-3) / 1e9},
        }
        bwd_dom = a_ms >= g_ms
        dom = "bwd_fused_kernel(sgd apply+finish)" if bwd_dom else "fwd_fused_kernel(gather+rank)"
        dom_bytes = bwd_b if bwd_dom else fwd_b
        dom_ms = max(a_ms, g_ms)
        traffic, traffic_src = pmc_traffic("ha::bwd_fused_kernel" if bwd_dom else "ha::fwd_fused_kernel")
        roofline = {"bound": "hbm", "kernel": dom, "achieved": dom_bytes / (dom_ms * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_src,
                    "avg_launch_us": dom_ms * 1e3,
                    "avg_launch_source": "HIP events around a hipGraph of 64 back-to-back launches, 5 replays",
                    "algorithmic_bytes_per_launch": dom_bytes}

    step_gbs = (fwd_b + bwd_b) / (ms_per_step * 1e-3) / 1e9
    result = {
        "metric": "embedding rows/s (lookup+grad)", "value": rows_per_s, "unit": "rows/s",
        "n_gpus": 1, "steps": args.steps, "warmup": wu, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "wdl_criteo bs=%d d=%d, %d fields, full %d-row fp32 table in HBM (%.1f GB), "
                               "gather + dedup plan + fused SGD scatter-apply per step (%s); the LRU "
                               "cache-limit-0.1 tier is not part of this line"
                               % (args.batch, args.width, args.fields, args.rows,
                                  args.rows * args.width * 4 / 1e9,
                                  ("one launch per %s from the steps' work queues: apply(k) + rows of k+1 per unique key; plans "
                                   "and queues prepared a block of %d steps at a time by two launches on a side stream, "
                                   "inside the timed region; keys with 16+ occurrences as row - tree_sum(lr*g)"
                                   % ("step", args.queue_block)) if queue and not args.queue_serial else
                                  "three launches per step: plan of batch k+2, queue of step k+1, then the items of step "
                                  "k from its queue" if queue else
                                  "one launch: apply(k), rows of k+1 forwarded / copied, finish(k+2), sort(k+3)" if ahead2 else
                                  "one launch: apply(k) beside lookup(k+1)" if one else "two launches"),
                   "ids_per_step": n, "unique_per_step": u_mean, "distinct_batches": nb,
                   "same_batch": os.environ.get("HA_BENCH_SAME_BATCH") == "1",
                   "grad_and_out_buffers": nbuf,
                   "launches_per_step": args.launches,
                   "lookahead_batches": (pipe.LOOKAHEAD if queue else 3 if ahead2 else 1),
                   "engine": args.engine if one else "two launches",
                   # keys with 16+ occurrences in a batch: row - tree_sum(lr*g) in a fixed order (within BASELINE.json's 1e-5
                   # on accumulated gradients; everything else the reference's serial chain bit for bit).  The other engines
                   # (--engine handoff / forward) are the serial chain throughout: numbers are like for like only per engine.
                   "numerics": ("tolerance>=16: keys with 16+ occurrences in a batch as row - tree_sum(lr*g) in a fixed order, "
                                "|result - serial chain| <= 1e-5 * (lr * sum|g| + |row|) per element (the bound the tests enforce: "
                                "relative to sum|g|, not |sum g|, plus one |row| term; observed maximum 3.1e-7 of it, "
                                "tests/test_gpu_fullscale.py prints it); below 16 occurrences the reference's serial chain bit for "
                                "bit; `bit_exact_ms_per_step` = the engine that is the serial chain throughout") if queue
                               else "bit-exact",
                   "stream_sync": (pipe.sync if queue and not args.queue_serial else None),
                   "launch": ("%d hipGraph replays of at most %d steps each" % (replays, G)) if use_graph
                             else "plain launches, one per step, enqueued ahead of the device",
                   "parallelism": "1 GPU"},
        "step_algorithmic_bytes": fwd_b + bwd_b,
        "step_hbm_GBps": step_gbs, "step_hbm_frac_of_peak": step_gbs / HBM_PEAK_GBS,
        "device_ms": dev_ms, "enqueue_ms": t_enq * 1e3, "wall_ms": wall * 1e3,
        # method_version 3 (rounds 4-5): behind the gate `ms_per_step` is the device's time over the region (HIP events);
        # versions 1-2 (rounds 1-3) reported the larger of device and enqueue time.  That figure is kept beside it:
        "method_version": 3,
        "ms_per_step_incl_enqueue": max(dev_ms, t_enq * 1e3) / max(args.steps, 1),
        "timed_region": "exactly %d steps between two HIP events on the launch stream; %d of the %d warm-up steps run "
                        "right in front of it%s%s" % (args.steps, pre, wu, ", behind a gate the host opens once everything "
                                                      "is enqueued (the enqueue time of the steps, `enqueue_ms`, is therefore "
                                                      "not part of the region)" if gate is not None else "",
                                                      "; untimed prologue in front of those: %d device copies of 512 MiB (clocks) and "
                                                      "%d read-only lookups of the batches that precede them in the stream (the "
                                                      "last-level cache as a long run leaves it); the table is not modified by either"
                                                      % (args.clock_warm, args.cache_warm)),
        "host_bound": bool(gate is None and t_enq * 1e3 > dev_ms),
        "block_boundaries_in_timed_region": (len([k for k in range(K0 + wu, K0 + wu + args.steps) if k % args.queue_block == 0])
                                             if queue and not args.queue_serial else None),
        "roofline": roofline, "kernels": kernels,
    }
    if one and not ahead2:
        torch.cuda.synchronize()
        result["handoff_timeouts"] = int(plans[0].handoff_timed_out()) + int(plans[1].handoff_timed_out())
    if queue:
        torch.cuda.synchronize()
        if pipe.overflowed():        # a queue overflowed / was read before it was complete: the number means nothing
            raise SystemExit("bench.py: the work-queue engine raised its sticky error word")
    def secondary(name, fn):
        """Secondary lines must never cost the headline: a failure is recorded, not raised."""
        if os.environ.get("HA_BENCH_TRACE") == "1":
            print("[bench] leg %s" % name, file=sys.stderr, flush=True)
        try:
            result[name] = fn()
        except Exception as ex:      # noqa: BLE001 -- anything (allocation, capture, a stream error)
            result[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            try:
                torch.cuda.synchronize()
            except Exception:        # noqa: BLE001
                pass

    if queue and not args.queue_serial and not args.no_sweeps:
        # (both legs take more SGD steps on the headline's table: after the headline's window, before the tiers that use the table
        # as their store)
        def _bit_exact():
            ms = bit_exact_leg(table, ids_dev, grads, outs, n, dev)
            result["bit_exact_ms_per_step"] = ms
            return {"ms_per_step": ms, "engine": "handoff (ha_sgd_push_pull_f32ids, ids one batch ahead, hipGraphs of 32 steps)",
                    "steps": 256, "headline_over_bit_exact": ms_per_step / ms}
        secondary("bit_exact_engine", _bit_exact)
        secondary("lookahead_sweep", lambda: lookahead_sweep(table, [ids_dev[i] for i in range(nb)], grads, outs, n, dev,
                                                             args.queue_block, ms_per_step))
    if not args.no_cpu_baseline:
        secondary("cpu_baseline", lambda: cpu_baseline(args, ids_host))
    if not args.no_cache_tier:
        secondary("cache_tier", lambda: cache_tier(args, table, ids_dev, outs[0], grads[0], dev))
    if not args.no_laia:
        secondary("laia_scheduler", lambda: laia_scheduler(args))
    if not args.no_wide and queue and args.width == 512 and args.rows >= (1 << 20):
        from herald_amd import wide_bench
        # configs[3]'s per-GPU shape on the headline's own table (bs=1024 d=512), configs[2]'s on a d=128 table of the same rows
        secondary("wide_bs1024_d512", lambda: wide_bench.measure(table, args.rows, 1024, args.width, fields=args.fields))

        def _wide_c():
            t128 = init_table(args.rows, 128, dev)
            try:
                return wide_bench.measure(t128, args.rows, 4096, 128, fields=args.fields)
            finally:
                del t128
                torch.cuda.empty_cache()
        secondary("wide_bs4096_d128", _wide_c)
    if not args.no_cold_tier:
        del table
        torch.cuda.empty_cache()
        secondary("cold_tier", lambda: cold_tier(args, dev))
    try:        # (anything a native library left in C stdio's buffer goes out in front of the line, not behind it)
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
