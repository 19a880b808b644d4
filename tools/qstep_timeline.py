#!/usr/bin/env python3
"""Per-wave timeline of ha_qstep_* (work-queue step; development aid): roles, item kinds, start / end."""
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
nb = 40
ids = []
for b in range(nb + 3):
    f = np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)
    ids.append(torch.from_numpy(f).to(dev))
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
OVERLAP = os.environ.get("SERIAL", "0") != "1"
pipe = ops.QueueStepPipeline(table, n, 1e-6, overlap=OVERLAP)
LA = pipe.LOOKAHEAD
NBLK = 1024
dbg = torch.zeros(NBLK * 16 * 4, dtype=torch.int64, device=dev)
dbgp = torch.zeros(4 * 64 + 64, dtype=torch.int64, device=dev)
n_of = lambda b: n if b >= 0 else 0
for c in range(-LA, 0):
    pipe.launch(c, n_of, None, outs[0], ids[c + LA])
for k in range(nb - LA - 1):
    if k == nb - LA - 2:
        torch.cuda.synchronize()
        print("queue of the stamped launch:", pipe.queue_header(k))
        pipe.launch(k, n_of, grads[k % 24], outs[(k + 1) % 24], ids[k + LA], dbg_prep=dbgp, dbg_apply=dbg)
    else:
        pipe.launch(k, n_of, grads[k % 24], outs[(k + 1) % 24], ids[k + LA])
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
rawp = dbgp.cpu().numpy()
nblk = 48 + min(448, (2 * n) // 16 + 1)
ph = rawp[3 * 64:3 * 64 + 32]
t_apply0 = int(raw[raw > 0].min()) if (raw > 0).any() else 0
for name, off in (("plan A (0 ids 1 claim 2 number 3 label 4 rank 5 scan 6 out 7)", 0),
                  ("queue B0 = keys of the batch to apply (0 load 1 table 2 count 3 scans 4 emit 5)", 16),
                  ("queue B1 = copies", 24)):
    st = ph[off:off + 8]
    pts = [(i, int(v)) for i, v in enumerate(st) if v > 0]
    if pts:
        print(name, "phases (us):", ", ".join("%d:%.2f" % (i, (v - pts[0][1]) * 0.01) for i, v in pts),
              "| first stamp %.2f us after the step's first wave" % ((pts[0][1] - t_apply0) * 0.01))
d = raw[:nblk * 64].reshape(-1, 4)
live = d[:, 0] > 0
base = d[live, 0].min()
t0 = (d[:, 0] - base) * 0.01
t1 = (d[:, 1] - base) * 0.01
role = d[:, 2] & 0xFF
kind = d[:, 3].astype(np.int64)
kind = np.where(kind > 100, -1, kind)      # 0xFFFFFFFF = the wave had no item
print("span %.2f us, stamped waves %d" % (t1[live].max(), live.sum()))
groups = [("coop G", (role == 0) & (kind == 4)), ("coop idle", (role == 0) & (kind < 0)), ("plan A", role == 1),
          ("queue B", role == 2), ("long L", (role == 3) & (kind == 2)), ("medium M", (role == 3) & (kind == 1)),
          ("small S", (role == 3) & (kind == 0)), ("zero Z", (role == 3) & (kind == 3)),
          ("worker idle", (role == 3) & (kind < 0))]
for name, sel in groups:
    m = live & sel
    if not m.any():
        continue
    print("%-11s waves %5d  start p10 %.2f p50 %.2f p90 %.2f max %.2f | end p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f | dur p50 %.2f p90 %.2f max %.2f"
          % ((name, m.sum()) + tuple(np.percentile(t0[m], [10, 50, 90, 100])) + tuple(np.percentile(t1[m], [10, 50, 90, 99, 100]))
             + tuple(np.percentile((t1 - t0)[m], [50, 90, 100]))))
for lo in np.arange(0, t1[live].max(), 1.0):
    row = []
    for name, sel in groups:
        mm = live & sel & (t0 < lo + 1.0) & (t1 > lo)
        row.append(int(mm.sum()))
    print("t=%4.1f us resident waves %s: %s" % (lo, "/".join(g[0].split()[-1] for g in groups), row))
idx = np.nonzero(live)[0]
order = idx[np.argsort(-t1[idx])][:12]
print("latest waves (block, wave, role, kind, start, end):",
      [(int(i // 16), int(i % 16), int(role[i]), int(kind[i]), round(float(t0[i]), 2), round(float(t1[i]), 2)) for i in order])
