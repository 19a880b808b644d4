#!/usr/bin/env python3
"""Per-GPU kernels of BASELINE configs[2] shape: bs=4096 d=128 (n = 106,496 ids, radix-sort path) (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
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
