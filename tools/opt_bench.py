#!/usr/bin/env python3
"""Fused deduplicate + Adam step vs the reference's two-step sequence (dedup-reduce, then the optimizer kernel),
wdl_criteo bs=256 d=512 shape on a 4M-row table (development aid)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows, width, n = 4_000_000, 512, 6656
param = torch.randn((rows, width), device=dev) * 0.01
m = torch.zeros_like(param); v = torch.zeros_like(param)
ids = [torch.from_numpy((synth.criteo_batch(256, b).reshape(-1) % rows).astype(np.float32)).to(dev) for b in range(32)]
grads = [torch.randn((n, width), device=dev) for _ in range(8)]
plan = ops.IndexPlan(n, dev)
def timeit(fn, reps=200):
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
def fused(i):
    ops.sparse_opt_fused("adam", param, ids[i % 32], grads[i % 8], m, v, plan=plan)
sc = [ctypes.c_float(x) for x in (0.01, 0.9, 0.999, 0.9, 0.999, 1e-7)]
def two_step(i):
    sl = ops.IndexedSlices(ids[i % 32], grads[i % 8], (rows, width)).deduplicate()
    ops.dl_call("AdamOptimizerSparseUpdate", [param, sl.indices.contiguous(), sl.values.contiguous(), m, v], scalars=sc)
print("fused dedup + Adam        %.1f us per batch" % timeit(fused))
print("deduplicate() then Adam   %.1f us per batch" % timeit(two_step))
