#!/usr/bin/env python3
"""Times the parts of the one-launch step in isolation (development aid): gather+rank role only, apply+finish
role only, the merged launch, and the two fused launches of round 1 -- each as a graph of 64 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import ops, synth

dev = torch.device("cuda:0")
rows, width, n = int(os.environ.get("ROWS", "33762577")), 512, 6656
table = torch.empty((rows, width), device=dev)
for _s in range(0, rows, 1 << 20):
    table[_s:_s + (1 << 20)].normal_(0, 0.01)
KL = 64
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev)
       for b in range(KL + 1)]
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
s = torch.cuda.Stream(device=dev)
plans = [ops.IndexPlan(n, dev).sort(ids[i], stream=s) for i in range(KL + 1)]
pends = [ops.PendingTable(dev) for _ in range(KL + 1)]
scratch_plan = ops.IndexPlan(n, dev)
s.synchronize()

def timed(fn, reps=5):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(KL):
            fn(i)
    with torch.cuda.stream(s):
        g.replay(); s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s); s.synchronize()
    return a.elapsed_time(b) / (reps * KL) * 1e3

print("fwd_fused (lookup_sort)          %.2f us" % timed(lambda i: ops.lookup_sort(table, ids[i], scratch_plan, out=outs[i % 24], stream=s)))
print("bwd_fused (sgd_apply_finish)     %.2f us" % timed(lambda i: ops.sgd_apply_finish(table, plans[i], grads[i % 24], 1e-6, stream=s, next_ids=ids[i + 1])))
print("bwd_fused no prefetch            %.2f us" % timed(lambda i: ops.sgd_apply_finish(table, plans[i], grads[i % 24], 1e-6, stream=s)))
print("step: gather+rank roles only     %.2f us" % timed(lambda i: ops.lookup_sort_pend(table, ids[i], scratch_plan, pends[KL], out=outs[i % 24], stream=s)))
print("step: apply+finish roles only    %.2f us" % timed(lambda i: ops.sgd_push_pull(table, plans[i], grads[i % 24], 1e-6, pends[i], stream=s)))
