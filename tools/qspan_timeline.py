#!/usr/bin/env python3
"""Per-wave timeline of ONE spanning launch (ha_qapply_span: the items of a block of consecutive steps, development aid):
per step when its waves start and end, how long they waited for the items they depend on, resident waves over time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import ops, synth

dev = torch.device("cuda:0")
rows, width, n = int(os.environ.get("ROWS", "33762577")), int(os.environ.get("WIDTH", "512")), 6656
table = torch.empty((rows, width), device=dev)
for _s in range(0, rows, 1 << 20):
    table[_s:_s + (1 << 20)].normal_(0, 0.01)
BK = int(os.environ.get("BLOCK", "16"))
SPAN = int(os.environ.get("SPAN", str(BK)))
nblocks = 4
nsteps = nblocks * BK
ids = []
for b in range(nsteps + 4 * BK):
    f = np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)
    ids.append(torch.from_numpy(f).to(dev))
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
pipe = ops.QueueStepPipeline(table, n, 1e-6, block=BK, span=True)
LA = pipe.LOOKAHEAD
WPW = int(os.environ.get("WPW", "16"))
NBLK = 520 * (16 // WPW) * SPAN
dbg = torch.zeros(NBLK * WPW * 8, dtype=torch.int64, device=dev)
ids_of = lambda j: ids[j] if 0 <= j < len(ids) else None
stamp = 2 * BK               # the third block's launch is the stamped one
for c in range(-LA, 0):
    if c % BK == 0:
        pipe.prepare_block(c // BK, ids_of)
pipe.apply(-1, None, outs[0], n_cur=0, n_next=n)
c = 0
while c < nsteps:
    if c % BK == 0:
        pipe.prepare_block(c // BK, ids_of)
    m = min(SPAN, BK - c % BK)
    if c == stamp:
        torch.cuda.synchronize()
    pipe.apply_span(c, [grads[(c + i) % 24] for i in range(m)], [outs[(c + i + 1) % 24] for i in range(m)],
                    dbg=dbg if c == stamp else None)
    c += m
torch.cuda.synchronize()
assert not pipe.overflowed()
d = dbg.cpu().numpy().reshape(-1, 8)
live = d[:, 0] > 0
base = d[live, 0].min()
t0 = (d[:, 0] - base) * 0.01
t1 = (d[:, 1] - base) * 0.01
role = d[:, 2] & 0xFF
wait = ((d[:, 2] >> 16) & 0xFFFFFFFF) * 0.01
kind = (d[:, 3] & 0xFF).astype(np.int64)
kind = np.where(kind > 100, -1, kind)
step = ((d[:, 3] >> 8) & 0xFFFF).astype(np.int64)
iscopy = ((d[:, 3] >> 24) & 1).astype(bool)
marks = np.stack([((d[:, 4] >> (16 * k)) & 0xFFFF) * 0.01 for k in range(4)] + [(d[:, 5] & 0xFFFF) * 0.01], axis=1)
nst = int(step[live].max()) + 1
print("launch of %d steps: span %.2f us = %.2f us per step, stamped waves %d" % (nst, t1[live].max(), t1[live].max() / nst, live.sum()))
names = {4: "G", 2: "L", 1: "M", 0: "S", 3: "Z", -1: "idle"}
for s in range(nst):
    m = live & (step == s)
    w = m & (kind >= 0)
    print("step %2d: waves %5d  start p1 %.2f p50 %.2f p99 %.2f | end p50 %.2f p99 %.2f max %.2f | dur p50 %.2f p90 %.2f | waited: %4d waves, "
          "p50 %.2f p90 %.2f max %.2f us, sum %.0f wave-us"
          % ((s, m.sum()) + tuple(np.percentile(t0[m], [1, 50, 99])) + tuple(np.percentile(t1[m], [50, 99, 100]))
             + tuple(np.percentile((t1 - t0)[w], [50, 90])) + (int((wait[w] > 0).sum()),)
             + (tuple(np.percentile(wait[w][wait[w] > 0], [50, 90, 100])) if (wait[w] > 0).any() else (0, 0, 0)) + (wait[w].sum(),)))
print("by class over the launch (steps >= 2):")
for k, nm in names.items():
    m = live & (kind == k) & (step >= 2)
    if m.any():
        print("  %-4s waves %6d  dur p50 %.2f p90 %.2f p99 %.2f | waited %5d waves p50 %.2f p90 %.2f"
              % ((nm, m.sum()) + tuple(np.percentile((t1 - t0)[m], [50, 90, 99])) + (int((wait[m] > 0).sum()),)
                 + (tuple(np.percentile(wait[m][wait[m] > 0], [50, 90])) if (wait[m] > 0).any() else (0, 0))))
T = t1[live].max()
for lo in np.arange(0, T, max(1.0, round(T / 60))):
    mm = live & (t0 < lo + 1.0) & (t1 > lo)
    act = mm & (kind >= 0)
    sts = np.unique(step[act]) if act.any() else []
    print("t=%5.1f us resident waves %5d (with an item %5d) of steps %s" % (lo, mm.sum(), act.sum(), list(map(int, sts))))

print("phases of S items (us since the wave started; marks: 1 item arrived, 2 row may be touched, 3 row + gradient rows here, "
      "4 stores issued, 5 row's stores in memory; then the wave's end):")
for nm, sel in (("S apply, step 0", live & (kind == 0) & ~iscopy & (step == 0)), ("S copy,  step 0", live & (kind == 0) & iscopy & (step == 0)),
                ("S apply, steps >= 2", live & (kind == 0) & ~iscopy & (step >= 2)), ("S copy,  steps >= 2", live & (kind == 0) & iscopy & (step >= 2))):
    if not sel.any():
        continue
    mk = marks[sel]
    end = (t1 - t0)[sel]
    row = []
    for k in range(5):
        v = mk[:, k][mk[:, k] > 0]
        row.append("m%d p50 %.2f p90 %.2f" % (k + 1, np.percentile(v, 50), np.percentile(v, 90)) if v.size else "m%d -" % (k + 1))
    print("  %-20s n %6d | %s | end p50 %.2f p90 %.2f" % (nm, sel.sum(), " | ".join(row), np.percentile(end, 50), np.percentile(end, 90)))
