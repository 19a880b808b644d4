#!/usr/bin/env python3
"""Per-step device time of the first steps after an idle period (development aid): is the short run's overhead a
fixed start-up cost or a gradual warm-up?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows, width, n = 33762577, 512, 6656
table = torch.empty((rows, width), device=dev)
for s0 in range(0, rows, 1 << 21):
    table[s0:s0 + (1 << 21)].normal_(0, 0.01)
NB = 128
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(NB)]
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ops.lookup_sort_pend(table, ids[0], plans[0], pends[0], out=outs[0], stream=s)
    s.synchronize()
    import time
    for trial in range(3):
        time.sleep(0.05)
        K = 64
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        ev[0].record(s)
        for k in range(K):
            kk = trial * K + k
            ops.sgd_push_pull(table, plans[kk % 2], grads[kk % 24], 1e-6, pends[kk % 2], ids[(kk + 1) % NB],
                              plans[(kk + 1) % 2], pends[(kk + 1) % 2], next_out=outs[(kk + 1) % 24], stream=s)
            ev[k + 1].record(s)
        s.synchronize()
        t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
        print("trial %d: first 12 steps %s | mean 12..31 %.2f | mean 32..63 %.2f" %
              (trial, " ".join("%.1f" % x for x in t[:12]), np.mean(t[12:32]), np.mean(t[32:])))
