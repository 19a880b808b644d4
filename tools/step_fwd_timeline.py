#!/usr/bin/env python3
"""Per-wave timeline of ha_step_* (three batches of lookahead; development aid): roles, start / end."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import _lib, ops, synth

dev = torch.device("cuda:0")
L = _lib.load()
vp = ctypes.c_void_p
rows, width, n = int(os.environ.get("ROWS", "33762577")), 512, 6656
table = torch.empty((rows, width), device=dev)
for _s in range(0, rows, 1 << 20):
    table[_s:_s + (1 << 20)].normal_(0, 0.01)
nb = 40
ids = []
for b in range(nb + 3):
    f = np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)
    ids.append(torch.from_numpy(f).to(dev))
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
pipe = ops.StepPipeline(table, n, 1e-6)
NBLK = 2400
dbg = torch.zeros(NBLK * 16 * 4, dtype=torch.int64, device=dev)
pipe.start(ids[0], ids[1], ids[2], out=outs[0])
P = lambda t: vp(t.data_ptr())
for k in range(nb - 1):
    if k == nb - 2:
        dbg.zero_()
        torch.cuda.synchronize()
        rc = L.ha_debug_step_fwd_timeline(P(table), rows, width, P(pipe.plan_of(k).ws), n, P(grads[k % 24]), 1e-6,
                                          pipe._tab(k), P(pipe.plan_of(k + 1).ws), n, P(outs[(k + 1) % 24]),
                                          pipe._tab(k + 1), P(pipe.plan_of(k + 2).ws), n, pipe._tab(k + 2),
                                          P(ids[k + 3]), n, P(pipe.plan_of(k + 3).ws), pipe._tab(k + 3), P(dbg),
                                          None)
        assert rc == 0
    else:
        pipe.step(grads[k % 24], ids[k + 3], out=outs[(k + 1) % 24])
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 4)
live = d[:, 0] > 0
base = d[live, 0].min()
t0 = (d[:, 0] - base) * 0.01
tm = np.where(d[:, 1] > 0, (d[:, 1] - base) * 0.01, np.nan)
t1 = (d[:, 2] - base) * 0.01
role = d[:, 3] & 0xFF
print("span %.2f us, waves %d" % (t1[live].max(), live.sum()))
names = ("finish", "apply", "rank", "gather", "clear")
for r, name in enumerate(names):
    m = live & (role == r)
    if not m.any():
        continue
    print("%-7s waves %5d  start p10 %.2f p50 %.2f p90 %.2f max %.2f | end p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f | dur p50 %.2f p90 %.2f"
          % ((name, m.sum()) + tuple(np.percentile(t0[m], [10, 50, 90, 100])) + tuple(np.percentile(t1[m], [10, 50, 90, 99, 100]))
             + tuple(np.percentile((t1 - t0)[m], [50, 90]))))
m = live & (role == 3)
copied = m & ((t1 - tm) > 0.3)
print("gather: %d of %d waves copied a row (probe p50 %.2f us; copy p50 %.2f p90 %.2f us)"
      % (copied.sum(), m.sum(), np.nanpercentile(tm[m] - t0[m], 50), np.nanpercentile((t1 - tm)[copied], 50),
         np.nanpercentile((t1 - tm)[copied], 90)))
for lo in np.arange(0, t1[live].max(), 1.0):
    row = []
    for r in range(5):
        mm = live & (role == r) & (t0 < lo + 1.0) & (t1 > lo)
        row.append(int(mm.sum()))
    print("t=%4.1f us resident waves finish/apply/rank/gather/clear: %s" % (lo, row))
m = live & (role == 1)
idx = np.nonzero(m)[0]
order = idx[np.argsort(-t1[idx])][:12]
print("latest apply waves (block, wave, start, end):", [(int(i // 16), int(i % 16), round(float(t0[i]), 2), round(float(t1[i]), 2)) for i in order])
