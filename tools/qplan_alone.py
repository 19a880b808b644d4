#!/usr/bin/env python3
"""Phase stamps of the plan (A) and queue (B) workgroups of ha_qstep_* when nothing else runs beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import ops, synth

dev = torch.device("cuda:0")
rows, width, n = 2_000_000, 512, 6656
table = torch.zeros((rows, width), device=dev)
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev)
       for b in range(8)]
pipe = ops.QueueStepPipeline(table, n, 1e-6)
dbg = torch.zeros(4 * 64 + 64, dtype=torch.int64, device=dev)


def phases(tag, nblk):
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy()
    ph = raw[nblk * 64:nblk * 64 + 32]
    for name, off in (("plan A", 0), ("queue B0", 16), ("queue B1", 24)):
        pts = [(i, int(v)) for i, v in enumerate(ph[off:off + 8]) if v > 0]
        if pts:
            print(tag, name, "phases (us):", ", ".join("%d:%.2f" % (i, (v - pts[0][1]) * 0.01) for i, v in pts))
    dbg.zero_()


for rep in range(3):
    pipe.launch(-3, 0, None, 0, None, 0, ids[0], dbg=dbg)          # A only
    phases("A alone:", 1)
    pipe.launch(-3, 0, None, 0, None, 0, ids[1], dbg=dbg)
    phases("A alone:", 1)
    pipe.launch(-2, 0, None, 0, None, n, ids[2], dbg=dbg)          # A + B(nothing, batch 0)
    phases("A + B(copies only):", 3)
    out = torch.empty((n, width), device=dev)
    big = torch.zeros(600 * 64 + 64, dtype=torch.int64, device=dev)
    pipe.launch(-1, 0, None, n, out, n, ids[3], dbg=big)           # A + B(batch 0, batch 1) + the copies of batch 0
    torch.cuda.synchronize()
    raw = big.cpu().numpy()
    nblk = 3 + 0 + min(448, n // 16 + 1)
    for name, off in (("plan A", 0), ("queue B0", 16), ("queue B1", 24)):
        pts = [(i, int(v)) for i, v in enumerate(raw[nblk * 64 + off:nblk * 64 + off + 8]) if v > 0]
        print("A + B(join) + copies:", name, "phases (us):", ", ".join("%d:%.2f" % (i, (v - pts[0][1]) * 0.01) for i, v in pts))
    pipe.reset()
