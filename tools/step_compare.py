#!/usr/bin/env python3
"""ha_step_* (forwarding) against ha_sgd_push_pull_* (pending tables) over row widths (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows, bs = 4_000_000, int(os.environ.get("BATCH", "256"))
n = bs * 26
NB = 24
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(bs, b, rows=rows)).reshape(-1), rows - 1)).to(dev)
       for b in range(NB)]
s = torch.cuda.Stream()
def timed(graph, reps=20):
    with torch.cuda.stream(s):
        graph.replay(); s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            graph.replay()
        b.record(s); s.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * NB)
for width in (8, 16, 64, 128, 512):
    table = torch.randn((rows, width), device=dev) * 0.01
    grads = [torch.randn((n, width), device=dev) for _ in range(NB)]
    outs = [torch.empty((n, width), device=dev) for _ in range(NB)]
    with torch.cuda.stream(s):
        pipe = ops.StepPipeline(table, n, 1e-6)
        pipe.reset(stream=s)
        pipe.launch(-3, 0, None, 0, None, 0, ids[0], stream=s)
        pipe.launch(-2, 0, None, 0, None, n, ids[1], stream=s)
        pipe.launch(-1, 0, None, n, outs[0], n, ids[2], stream=s)
        pipe.launch(0, n, grads[0], n, outs[1], n, ids[3], stream=s)
        s.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s):
            for k in range(1, NB + 1):
                pipe.launch(k, n, grads[k % NB], n, outs[(k + 1) % NB], n, ids[(k + 3) % NB], stream=s)
        t_fwd = timed(g1)
        plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
        pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
        ops.lookup_sort_pend(table, ids[0], plans[0], pends[0], out=outs[0], stream=s)
        ops.sgd_push_pull(table, plans[0], grads[0], 1e-6, pends[0], ids[1], plans[1], pends[1], next_out=outs[1], stream=s)
        s.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            for k in range(1, NB + 1):
                ops.sgd_push_pull(table, plans[k % 2], grads[k % NB], 1e-6, pends[k % 2], ids[(k + 1) % NB],
                                  plans[(k + 1) % 2], pends[(k + 1) % 2], next_out=outs[(k + 1) % NB], stream=s)
        t_pp = timed(g2)
    print("width %4d: ha_step (forwarding) %.2f us   ha_sgd_push_pull %.2f us%s"
          % (width, t_fwd, t_pp, "   (separate launches: width %% 32 != 0)" if width % 32 else ""))
    del table, grads, outs
