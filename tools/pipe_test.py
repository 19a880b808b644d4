#!/usr/bin/env python3
"""Serial vs software-pipelined step schedule inside a hipGraph (development aid).
serial   : [gather(k)+sort(k)] -> [apply(k)+finish(k)]                       (two fused launches, one stream)
pipelined: main  gather(k) -> apply+finish(k) -> gather(k+1) ...
           side  sort(k+1) while gather(k) / apply(k) run                    (ids are known one batch ahead)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows = int(os.environ.get("ROWS", "33762577"))
width, n = 512, 6656
table = torch.empty((rows, width), device=dev)
for s in range(0, rows, 1 << 20):
    table[s:s + (1 << 20)].normal_(0, 0.01)
out = torch.empty((n, width), device=dev)
grads = [torch.randn((n, width), device=dev) for _ in range(4)]
NB, G = 256, 32
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(NB)]
main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream()
plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
torch.cuda.synchronize()

def build_serial(k0):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main_s):
        for k in range(k0, k0 + G):
            ops.lookup_sort(table, ids[k % NB], plans[0], out=out, stream=main_s)
            ops.sgd_apply_finish(table, plans[0], grads[k % 4], 1e-6, stream=main_s)
    return g

def build_pipe(k0, where):
    """Entering the graph, plans[k0 % 2] holds the sort of batch k0."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main_s):
        for k in range(k0, k0 + G):
            if where == "with_gather":
                side_s.wait_stream(main_s)      # fork: sort(k+1) may start with gather(k)
            ops.embedding_lookup(table, ids[k % NB], out=out, stream=main_s)
            if where == "with_apply":
                side_s.wait_stream(main_s)      # fork: sort(k+1) starts with apply(k)
            plans[(k + 1) % 2].sort(ids[(k + 1) % NB], stream=side_s)
            ops.sgd_apply_finish(table, plans[k % 2], grads[k % 4], 1e-6, stream=main_s)
            main_s.wait_stream(side_s)          # join before the next step (plans[(k+1)%2] ready)
    return g

def timeit(graphs, reps=20):
    with torch.cuda.stream(main_s):
        for g in graphs: g.replay()
        main_s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main_s)
        for _ in range(reps):
            for g in graphs: g.replay()
        b.record(main_s); main_s.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * len(graphs) * G)

ser = [build_serial(k0) for k0 in range(0, NB, G)]
print("serial    %.2f us/step" % timeit(ser))
def build_fused(k0):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main_s):
        for k in range(k0, k0 + G):
            ops.embedding_lookup(table, ids[k % NB], out=out, stream=main_s)
            ops.sgd_apply_finish_sort_next(table, plans[k % 2], grads[k % 4], 1e-6, ids[(k + 1) % NB],
                                           plans[(k + 1) % 2], stream=main_s)
    return g

plans[0].sort(ids[0], stream=main_s); main_s.synchronize()
fg = [build_fused(k0) for k0 in range(0, NB, G)]
print("pipelined (one launch: apply+finish(k)+sort(k+1)) %.2f us/step" % timeit(fg))
