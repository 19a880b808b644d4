#!/usr/bin/env python3
"""Per-wave timeline of the one-launch step (development aid): roles, start/end, hand-off waits."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import _lib, ops, synth

dev = torch.device("cuda:0")
L = _lib.load()
vp, i64 = ctypes.c_void_p, ctypes.c_int64
L.ha_debug_step_timeline.argtypes = [vp, i64, i64, vp, i64, vp, ctypes.c_float, vp, vp, i64, vp, vp, vp, vp, vp]
rows, width, n = int(os.environ.get("ROWS", "33762577")), 512, 6656
table = torch.empty((rows, width), device=dev)
for _s in range(0, rows, 1 << 20):
    table[_s:_s + (1 << 20)].normal_(0, 0.01)
nb = 40
ids = []
for b in range(nb):
    f = np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)
    ids.append(torch.from_numpy(f).to(dev))
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
plans = [ops.IndexPlan(n, dev), ops.IndexPlan(n, dev)]
pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
NBLK = 2400
dbg = torch.zeros(NBLK * 16 * 4, dtype=torch.int64, device=dev)
ops.lookup_sort_pend(table, ids[0], plans[0], pends[0], out=outs[0])
P = lambda t: vp(t.data_ptr())
for k in range(nb - 1):
    last = k == nb - 2
    if last:
        dbg.zero_()
        torch.cuda.synchronize()
        rc = L.ha_debug_step_timeline(P(table), rows, width, P(plans[k % 2].ws), n, P(grads[k % 24]), 1e-6,
                                      P(pends[k % 2].buf), P(ids[k + 1]), n, P(outs[(k + 1) % 24]),
                                      P(plans[(k + 1) % 2].ws), P(pends[(k + 1) % 2].buf), P(dbg), None)
        assert rc == 0
    else:
        ops.sgd_push_pull(table, plans[k % 2], grads[k % 24], 1e-6, pends[k % 2], ids[k + 1], plans[(k + 1) % 2],
                          pends[(k + 1) % 2], next_out=outs[(k + 1) % 24])
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 4)
live = d[:, 0] > 0
base = d[live, 0].min()
t0 = (d[:, 0] - base) * 0.01
tm = np.where(d[:, 1] > 0, (d[:, 1] - base) * 0.01, np.nan)
t1 = (d[:, 2] - base) * 0.01
role = d[:, 3] & 0xFF
print("span %.2f us, waves %d" % (t1[live].max(), live.sum()))
for r, name in enumerate(("finish", "apply/short", "rank", "gather", "long", "medium")):
    m = live & (role == r)
    if not m.any():
        continue
    print("%-7s waves %5d  start p10 %.2f p50 %.2f p90 %.2f max %.2f | end p10 %.2f p50 %.2f p90 %.2f max %.2f | dur p50 %.2f p90 %.2f"
          % ((name, m.sum()) + tuple(np.percentile(t0[m], [10, 50, 90, 100])) + tuple(np.percentile(t1[m], [10, 50, 90, 100]))
             + tuple(np.percentile((t1 - t0)[m], [50, 90]))))
m = live & (role == 3)
w = tm[m] - t0[m]
print("gather wait: p50 %.2f p90 %.2f max %.2f us; waves waiting > 0.5us: %d of %d; after-wait copy p50 %.2f p90 %.2f"
      % (np.nanpercentile(w, 50), np.nanpercentile(w, 90), np.nanmax(w), (w > 0.5).sum(), m.sum(),
         np.nanpercentile((t1[m] - tm[m]), 50), np.nanpercentile((t1[m] - tm[m]), 90)))
# occupancy over time: resident waves per role in 1 us bins
for lo in np.arange(0, t1[live].max(), 1.0):
    row = []
    for r in range(6):
        mm = live & (role == r) & (t0 < lo + 1.0) & (t1 > lo)
        row.append(int(mm.sum()))
    print("t=%4.1f us resident waves finish/short/rank/gather/long/medium: %s" % (lo, row))
