#!/usr/bin/env python3
"""100 sorts + apply_finish of a large Criteo batch, for rocprofv3 --kernel-trace --stats (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows, width, bs = 33762577, int(os.environ.get("WIDTH", "128")), int(os.environ.get("BATCH", "4096"))
n = bs * 26
table = torch.zeros((4_000_000, width), device=dev)
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(bs, b, rows=rows)).reshape(-1), 3_999_999)).to(dev) for b in range(8)]
grads = torch.randn((n, width), device=dev)
plan = ops.IndexPlan(n, dev)
for k in range(100):
    plan.sort(ids[k % 8], key_limit=rows)
    ops.sgd_apply_finish(table, plan, grads, 1e-6)
torch.cuda.synchronize()
