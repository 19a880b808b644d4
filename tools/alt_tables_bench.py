#!/usr/bin/env python3
"""Where does the alternation penalty of the step come from? (development aid)
Graphs of 32 fwd + 32 bwd launches in different orders / with different data sharing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows = int(os.environ.get("ROWS", "33762577")); width, n = 512, 6656
def mk():
    t = torch.empty((rows, width), device=dev)
    for s in range(0, rows, 1 << 20):
        t[s:s + (1 << 20)].normal_(0, 0.01)
    return t
table = mk()
table2 = table
out = torch.empty((n, width), device=dev)
grads = [torch.randn((n, width), device=dev) for _ in range(4)]
NB, G = 256, 32
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(NB)]
s = torch.cuda.Stream()
plans = [ops.IndexPlan(n, dev) for _ in range(NB)]
for b in range(NB):
    plans[b].sort(ids[b], stream=s)
s.synchronize()

def build(order, k0, ft, bt):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        if order == "alt_pf":
            for k in range(k0, k0 + G):
                ops.lookup_sort(ft, ids[k], plans[k], out=out, stream=s)
                ops.sgd_apply_finish(bt, plans[k], grads[k % 4], 1e-6, stream=s, next_ids=ids[(k + 1) % NB])
        elif order == "bwd_pf":
            for k in range(k0, k0 + G):
                ops.sgd_apply_finish(bt, plans[k], grads[k % 4], 1e-6, stream=s, next_ids=ids[(k + 1) % NB])
        elif order == "alt":
            for k in range(k0, k0 + G):
                ops.lookup_sort(ft, ids[k], plans[k], out=out, stream=s)
                ops.sgd_apply_finish(bt, plans[k], grads[k % 4], 1e-6, stream=s)
        elif order == "blocks":
            for k in range(k0, k0 + G):
                ops.lookup_sort(ft, ids[k], plans[k], out=out, stream=s)
            for k in range(k0, k0 + G):
                ops.sgd_apply_finish(bt, plans[k], grads[k % 4], 1e-6, stream=s)
        elif order == "fwd":
            for k in range(k0, k0 + G):
                ops.lookup_sort(ft, ids[k], plans[k], out=out, stream=s)
        elif order == "bwd":
            for k in range(k0, k0 + G):
                ops.sgd_apply_finish(bt, plans[k], grads[k % 4], 1e-6, stream=s)
    return g

def timeit(graphs, reps=10):
    with torch.cuda.stream(s):
        for g in graphs: g.replay()
        s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            for g in graphs: g.replay()
        b.record(s); s.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * len(graphs) * G)

for name, order, ft, bt in (("fwd only", "fwd", table, table), ("bwd only", "bwd", table, table),
                            ("alternating, same table", "alt", table, table),
                            ("blocks of 32, same table", "blocks", table, table),
                            ("alternating + next-batch row prefetch", "alt_pf", table, table),
                            ("bwd only + prefetch", "bwd_pf", table, table)):
    gs = [build(order, k0, ft, bt) for k0 in range(0, NB, G)]
    print("%-50s %.2f us per step-equivalent" % (name, timeit(gs)))
