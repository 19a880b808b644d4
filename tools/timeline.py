#!/usr/bin/env python3
"""Per-wave timeline of the SGD apply kernel (development aid)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import _lib, ops, synth

dev = torch.device("cuda:0")
L = _lib.load()
L.ha_debug_apply_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p,
                                      ctypes.c_void_p]
BATCH = int(os.environ.get("BATCH", "256"))
rows, width, n = int(os.environ.get("ROWS", "2000000")), int(os.environ.get("WIDTH", "512")), BATCH * 26
table = torch.empty((rows, width), device=dev)
for _s in range(0, rows, 1 << 20):
    table[_s:_s + (1 << 20)].normal_(0, 0.01)
grads = torch.randn((n, width), device=dev)
for case in ("criteo", "distinct"):
    for rep in range(3):
        if case == "criteo":
            ids = np.minimum(synth.as_f32_ids(synth.criteo_batch(BATCH, rep, rows=rows)).reshape(-1), rows - 1)
        else:
            ids = np.random.default_rng(rep).choice(rows, size=n, replace=False).astype(np.float32)
        d_ids = torch.from_numpy(ids).to(dev)
        plan = ops.IndexPlan(n, dev).sort(d_ids)
        dbg = torch.zeros(n * 4, dtype=torch.int64, device=dev)
        ops.embedding_lookup(table, d_ids)     # as in a training step: the lookup of the batch precedes its apply
        torch.cuda.synchronize()
        rc = L.ha_debug_apply_timeline(ctypes.c_void_p(table.data_ptr()), rows, width,
                                       ctypes.c_void_p(plan.ws.data_ptr()), n,
                                       ctypes.c_void_p(grads.data_ptr()), ctypes.c_float(1e-6),
                                       ctypes.c_void_p(dbg.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(n, 4)
    t0 = d[:, 0].astype(np.int64); t1 = d[:, 1].astype(np.int64)
    live = t0 > 0
    base = t0[live].min()
    s = (t0 - base) * 10e-3; e = (t1 - base) * 10e-3          # us
    info = d[:, 2]; o = info >> 16; ln = info & 0x7FFF
    dur = e - s
    print("== %s: waves stamped %d, kernel span %.2f us" % (case, live.sum(), e[live].max()))
    print("   start times: p50 %.2f p90 %.2f max %.2f us" % tuple(np.percentile(s[live], [50, 90, 100])))
    worker = live & (dur > 0.3)
    for lo, hi in ((1, 1), (2, 3), (4, 15), (16, 47), (48, 10000)):
        m = live & (ln >= lo) & (ln <= hi) & (o == 0)
        if m.any():
            print("   run len %4d-%-5d heads %4d: dur p50 %.2f max %.2f us, end max %.2f" %
                  (lo, hi, m.sum(), np.median(dur[m]), dur[m].max(), e[m].max()))


    top = np.argsort(-e * live)[:6]
    for p in top:
        print("   late wave p=%d o=%d len=%d start %.2f end %.2f cycles %d" % (p, o[p], ln[p], s[p], e[p], d[p, 3]))
