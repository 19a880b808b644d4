#!/usr/bin/env python3
"""Do two independent kernels on two streams overlap inside a hipGraph on this system? (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows, width, n = 2000000, 512, 6656
table = torch.randn((rows, width), device=dev) * 0.01
table2 = torch.randn((rows, width), device=dev) * 0.01
out = torch.empty((n, width), device=dev)
grads = torch.randn((n, width), device=dev)
NB = 32
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(NB)]
plansA = [ops.IndexPlan(n, dev).sort(ids[b]) for b in range(NB)]
plansB = [ops.IndexPlan(n, dev) for b in range(NB)]
main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()

def build(mode):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main_s):
        for b in range(NB):
            if mode in ("both", "apply"):
                ops.sgd_apply_finish(table, plansA[b], grads, 1e-6, stream=main_s)
            if mode == "both":
                side_s.wait_stream(main_s) if b == 0 else None
                ops.lookup_sort(table2, ids[b], plansB[b], out=out, stream=side_s)
            if mode == "fwd":
                ops.lookup_sort(table2, ids[b], plansB[b], out=out, stream=main_s)
        if mode == "both":
            main_s.wait_stream(side_s)
    return g

def timeit(g, reps=10):
    with torch.cuda.stream(main_s):
        g.replay(); main_s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main_s)
        for _ in range(reps):
            g.replay()
        b.record(main_s); main_s.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * NB)

for mode in ("apply", "fwd", "both"):
    print(mode, "%.2f us per iteration" % timeit(build(mode)))
