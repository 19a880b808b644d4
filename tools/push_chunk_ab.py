#!/usr/bin/env python3
"""ha_push_apply_scaled_finished at BASELINE configs[2]'s per-GPU shape (106,496 ids, d = 128): exact chain / tolerance tree with
one workgroup per (key, slice) (mode 1) / tolerance tree with runs beyond 256 occurrences in chunks (mode 2)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth, _lib
dev = torch.device("cuda:0")
rows, width, bs = 4_000_000, int(os.environ.get("WIDTH", "128")), 4096
L = _lib.load()
table = torch.zeros((rows, width), device=dev)
plans, grads = [], []
NBUF = int(os.environ.get("NBUF", "8"))      # batches in rotation (8 x 54.5 MB of gradients: beyond the 256 MB Infinity Cache)
for b in range(NBUF):
    if os.environ.get("UNIFORM") == "1":      # no long runs at all: every key by a wave of the keys' role
        ids = torch.from_numpy(np.random.default_rng(b).integers(0, rows, size=bs * 26).astype(np.float32)).to(dev)
    else:
        ids = torch.from_numpy((synth.criteo_batch(bs, 100 + b).reshape(-1) % rows).astype(np.float32)).to(dev)
    plans.append(ops.IndexPlan(ids.numel(), dev).build(ids))
    grads.append(torch.randn((ids.numel(), width), device=dev))
n = plans[0].n
def run(reps):
    for r in range(reps):
        p, g = plans[r % NBUF], grads[r % NBUF]
        _lib.check(L.ha_push_apply_scaled_finished(ctypes.c_void_p(table.data_ptr()), rows, width, ctypes.c_void_p(p.ws.data_ptr()), n,
                                                   ctypes.c_void_p(g.data_ptr()), ctypes.c_float(-0.01), None), "push")
for mode in (0, 1, 2):
    ops.set_tolerance_mode(mode)
    run(8); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(64); e1.record(); torch.cuda.synchronize()
    print("d=%d tolerance mode %d: %.2f us per call" % (width, mode, e0.elapsed_time(e1) / 64 * 1e3))
