#!/usr/bin/env python3
"""Per-GPU kernels of BASELINE configs[2] shape: bs=4096 d=128 (n = 106,496 ids, radix-sort path) (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
if os.environ.get("TOL") == "1":
    ops.set_tolerance_mode(True)       # runs of 64+ occurrences as a fixed-order tree sum (BASELINE.json's 1e-5)
print("tolerance mode:", bool(__import__("herald_amd._lib", fromlist=["x"]).load().ha_get_tolerance_mode()))
rows, width, bs = 33762577, int(os.environ.get("WIDTH", "128")), int(os.environ.get("BATCH", "4096"))
n = bs * 26
table = torch.empty((rows, width), device=dev)
for s in range(0, rows, 1 << 21):
    table[s:s + (1 << 21)].normal_(0, 0.01)
NB = 16
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(bs, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(NB)]
out = torch.empty((n, width), device=dev)
grads = torch.randn((n, width), device=dev)
plan = ops.IndexPlan(n, dev)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
k = [0]
def nxt():
    k[0] += 1
    return ids[k[0] % NB]
print("n=%d unique~%d" % (n, np.unique(ids[0].cpu().numpy()).size))
print("gather       %.1f us" % timeit(lambda: ops.embedding_lookup(table, nxt(), out=out)))
print("sort         %.1f us" % timeit(lambda: plan.sort(nxt())))
print("sort (limit) %.1f us" % timeit(lambda: plan.sort(nxt(), key_limit=rows)))
print("lookup_sort  %.1f us" % timeit(lambda: ops.lookup_sort(table, nxt(), plan, out=out)))
print("apply_finish %.1f us" % timeit(lambda: ops.sgd_apply_finish(table, plan, grads, 1e-6)))
plan.sort(ids[0])
print("finish       %.1f us" % timeit(lambda: plan.finish()))
print("apply        %.1f us" % timeit(lambda: ops.sgd_apply(table, plan, grads, 1e-6)))
def step():
    i = nxt()
    ops.lookup_sort(table, i, plan, out=out)
    ops.sgd_apply_finish(table, plan, grads, 1e-6)
t = timeit(step)
print("step         %.1f us  -> %.1f M rows/s" % (t, n / t))
plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
ops.lookup_sort_pend(table, ids[0], plans[0], pends[0], out=out)
kk = [0]
def step1():
    j = kk[0]
    ops.sgd_push_pull(table, plans[j % 2], grads, 1e-6, pends[j % 2], ids[(j + 1) % NB], plans[(j + 1) % 2],
                      pends[(j + 1) % 2], next_out=out)
    kk[0] += 1
t = timeit(step1, reps=50)
print("one-launch step %.1f us  -> %.1f M rows/s   (time-outs: %s %s)" % (t, n / t, plans[0].handoff_timed_out(),
                                                                         plans[1].handoff_timed_out()))

# ---- the sort of batch k+1 on a second stream beside gather(k) + apply(k): a graph of NB steps --------------
main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream()
pl = [ops.IndexPlan(n, dev) for _ in range(2)]
NG = 16
def pipelined_graph():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main_s):
        pl[0].sort(ids[0], key_limit=rows, stream=main_s)          # prologue (eager): plan of batch 0
        main_s.synchronize()
        with torch.cuda.graph(g, stream=main_s):
            for k in range(NG):
                # side stream: sort batch k+1 into the other plan (free since apply(k-1) finished on main)
                side_s.wait_stream(main_s)
                with torch.cuda.stream(side_s):
                    pl[(k + 1) % 2].sort(ids[(k + 1) % NB], key_limit=rows, stream=side_s)
                ops.embedding_lookup(table, ids[k % NB], out=out, stream=main_s)
                ops.sgd_apply_finish(table, pl[k % 2], grads, 1e-6, stream=main_s)
                main_s.wait_stream(side_s)
    return g
def time_graph(g, reps=10):
    with torch.cuda.stream(main_s):
        g.replay(); main_s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main_s)
        for _ in range(reps):
            g.replay()
        b.record(main_s); main_s.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * NG)
try:
    t = time_graph(pipelined_graph())
    print("two streams (sort of batch k+1 beside gather + apply of batch k) %.1f us  -> %.1f M rows/s" % (t, n / t))
except Exception as e:   # noqa
    print("two-stream schedule failed:", e)

# ---- the product schedule for these shapes: ops.SortAheadPipeline (plans sorted a BLOCK ahead on the side stream, the
# streams meet once per block) ----------------------------------------------------------------------------------------
def sort_ahead(block, graph):
    """48 steps = 48 / block blocks (a multiple of the pipeline's three plan slots and of the period of the 16 id buffers, so
    that the captured sequence ends in the state it starts from and replays are exact)."""
    pipe = ops.SortAheadPipeline(table, n, 1e-6, block=block, key_limit=rows)
    nblk = 48 // block
    blocks = [[ids[(b * block + i) % NB] for i in range(block)] for b in range(nblk + 1)]
    def body():
        k = 0
        for b in range(nblk):
            pipe.prepare_block(blocks[b + 1])
            for i in range(block):
                pipe.lookup(b * block + i + body.base, blocks[b][i], out=out, stream=main_s)
                pipe.apply(b * block + i + body.base, grads, stream=main_s)
        main_s.wait_stream(pipe.side)
        body.base += nblk * block
    body.base = 0
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(main_s):
        pipe.prepare_block(blocks[0])
        pipe._plan(0, main_s)          # the wait for the prologue's block: outside the capture
        main_s.wait_stream(pipe.side)
        if graph:
            main_s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_s):
                body()
            run = g.replay
        else:
            run = body
        run(); main_s.synchronize()
        import gc
        gc.collect(); gc.disable()          # a collection inside the eager loop is 5-60 ms of host time (bench.py does the same)
        a.record(main_s)
        for _ in range(4):
            run()
        b_.record(main_s)
        main_s.synchronize()
        gc.enable()
    return a.elapsed_time(b_) * 1e3 / (4 * nblk * block)
for blk in (4, 8, 16):
    for graph in (True, False):
        try:
            t = sort_ahead(blk, graph)
            print("SortAheadPipeline, blocks of %2d, %s %.1f us  -> %.1f M rows/s" % (
                blk, "one hipGraph of 48 steps" if graph else "eager launches        ", t, n / t))
        except Exception as e:   # noqa
            print("SortAheadPipeline (block %d, graph %s) failed: %s" % (blk, graph, e))

# the same schedule launched eagerly (are the graph's parallel branches really concurrent?)
def eager_two_streams(steps=200):
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    with torch.cuda.stream(main_s):
        pl[0].sort(ids[0], key_limit=rows, stream=main_s)
    main_s.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main_s)
    for k in range(steps):
        side_s.wait_stream(main_s)
        pl[(k + 1) % 2].sort(ids[(k + 1) % NB], key_limit=rows, stream=side_s)
        ops.embedding_lookup(table, ids[k % NB], out=out, stream=main_s)
        ops.sgd_apply_finish(table, pl[k % 2], grads, 1e-6, stream=main_s)
        main_s.wait_stream(side_s)
    b.record(main_s)
    main_s.synchronize()
    return a.elapsed_time(b) * 1e3 / steps
eager_two_streams(20)
t = eager_two_streams()
print("two streams, eager launches %.1f us  -> %.1f M rows/s" % (t, n / t))
