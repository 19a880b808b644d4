#!/usr/bin/env python3
"""Throughput of the HET cache tier on the N=1 workload (development aid): LRU, limit = 0.1 x rows,
one embedding_lookup + one embedding_update per batch, device-resident keys / gradients."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import cache as hcache, synth
dev = torch.device("cuda:0")
rows = int(os.environ.get("ROWS", "33762577")); width = 512; n = 6656
table = torch.empty((rows, width), device=dev)
for s in range(0, rows, 1 << 20):
    table[s:s + (1 << 20)].normal_(0, 0.01)
versions = torch.zeros(rows, dtype=torch.int64, device=dev)
hcache.register_table(0, table, versions)
c = hcache.CacheSparseTable(int(0.1 * rows), rows, width, 0, "LRU", bound=100, max_batch=n, device=dev)
NB = 256
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1).astype(np.int64)).to(dev) for b in range(NB)]
dest = torch.empty((n, width), device=dev)
grads = torch.randn((n, width), device=dev)
def step(k):
    c.embedding_lookup(ids[k % NB], dest)
    c.embedding_update(ids[k % NB], grads)
for k in range(64): step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 400
for k in range(K): step(64 + k)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("cache tier: %.1f us/step, %.1f M rows/s" % (el / K * 1e6, n * K / el / 1e6))
print(c.perf if hasattr(c, "perf") else "")
