#!/usr/bin/env python3
"""The work-queue step at the per-GPU shapes of BASELINE configs[2] / configs[3] on the full 33,762,577-row table (the
measurement bench.py reports as `wide_*`; herald_amd/wide_bench.py).  env: BATCH, WIDTH, BLOCK, STEPS, SYNC."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from herald_amd import wide_bench
dev = torch.device("cuda:0")
rows, width, bs = 33762577, int(os.environ.get("WIDTH", "128")), int(os.environ.get("BATCH", "4096"))
table = torch.empty((rows, width), device=dev)
for s in range(0, rows, 1 << 21):
    table[s:s + (1 << 21)].normal_(0, 0.01)
r = wide_bench.measure(table, rows, bs, width, block=int(os.environ.get("BLOCK", "16")), steps=int(os.environ.get("STEPS", "96")),
                       sync=os.environ.get("SYNC", "flags"), alone=os.environ.get("ALONE") == "1")
print("bs=%d d=%d: %.2f us/step  %.1f M rows/s  frac %.3f of 8 TB/s  (%s)" % (bs, width, r["us_per_step"], r["rows_per_s"] / 1e6,
                                                                          r["roofline"]["frac"], r["stream_sync"]))
print(json.dumps(r))
if r.get("apply_alone_us"):
    print("apply launch alone (no preparation beside it): %.2f us" % r["apply_alone_us"])
