#!/usr/bin/env python3
"""Per-wave timeline of ha_qapply (the items of one work-queue step) and the phase stamps of a plan / a queue workgroup
(development aid): roles, item kinds, start / end."""
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
pipe_probe_la = 3 * BK
nsteps = 3 * BK + 2
ids = []
for b in range(nsteps + pipe_probe_la + 2 * BK):
    f = np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)
    ids.append(torch.from_numpy(f).to(dev))
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
pipe = ops.QueueStepPipeline(table, n, 1e-6, block=BK)
LA = pipe.LOOKAHEAD
NBLK = 2304
WPW = int(os.environ.get("WPW", "16"))      # waves per workgroup of the apply launch (16 = the product's geometry)
dbg = torch.zeros(NBLK * WPW * 4, dtype=torch.int64, device=dev)
dbgp = torch.zeros(64, dtype=torch.int64, device=dev)
ids_of = lambda j: ids[j] if 0 <= j < len(ids) else None
stamp = 2 * BK + BK // 2          # a step in the middle of a block (the side work of that block is long under way)
for c in range(-LA, nsteps):
    if c % BK == 0:
        pipe.prepare_block(c // BK, ids_of, ph=dbgp if c == 2 * BK else None)
    if c < -1:
        continue
    if c == stamp:
        torch.cuda.synchronize()
        print("queue of the stamped step:", pipe.queue_header(c))
    pipe.apply(c, grads[c % 24] if c >= 0 else None, outs[(c + 1) % 24], dbg=dbg if c == stamp else None)
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
ph = dbgp.cpu().numpy()
nblk = (64 if WPW == 16 else 256) + min(448 * 16 // WPW, (2 * n) // WPW + 1)
for name, off in (("plan workgroup (0 ids 1 claim 2 number 3 label 4 rank 5 scan 6 out 7)", 0),
                  ("queue workgroup 0 = keys of the batch to apply (0 load 1 table 2 count 3 scans 4 emit 5)", 16),
                  ("queue workgroup 1 = copies", 24)):
    st = ph[off:off + 8]
    pts = [(i, int(v)) for i, v in enumerate(st) if v > 0]
    if pts:
        print(name, "phases (us):", ", ".join("%d:%.2f" % (i, (v - pts[0][1]) * 0.01) for i, v in pts))
d = raw[:nblk * WPW * 4].reshape(-1, 4)
live = d[:, 0] > 0
base = d[live, 0].min()
t0 = (d[:, 0] - base) * 0.01
t1 = (d[:, 1] - base) * 0.01
role = d[:, 2] & 0xFF
kind = (d[:, 3] & 0xFF).astype(np.int64)
kind = np.where(kind > 100, -1, kind)      # 0xFFFFFFFF = the wave had no item
print("span %.2f us, stamped waves %d" % (t1[live].max(), live.sum()))
groups = [("coop G", (role == 0) & (kind == 4)), ("coop idle", (role == 0) & (kind < 0)),
          ("long L", (role == 3) & (kind == 2)), ("medium M", (role == 3) & (kind == 1)),
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
# ---- tail census: what the waves that are still alive late in the launch are (item class, occurrences c, destinations m)
cc = ((d[:, 3] >> 8) & 0xFFFFF).astype(np.int64)
mm_ = ((d[:, 3] >> 28) & 0xFFFFF).astype(np.int64)
col0 = ((d[:, 3] >> 48) & 0xFFFF).astype(np.int64) * 4
names = {0: "S", 1: "M", 2: "L", 3: "Z", 4: "G", -1: "-"}
span = t1[live].max()
for cut in (float(os.environ.get("CUT", "7.0")), 8.0, 9.0):
    late = live & (t1 > cut)
    print("== waves alive after %.1f us: %d of %d (span %.2f)" % (cut, late.sum(), live.sum(), span))
    for k in (0, 1, 2, 4):
        sel = late & (kind == k)
        allk = live & (kind == k)
        if not allk.any():
            continue
        is_copy = sel & (cc == 0)
        print("   %s: %5d late of %5d | apply c p50 %s max %s, m p50 %s max %s | copies (c = 0) %d, their m p50 %s max %s | start p50 %.2f dur p50 %.2f"
              % (names[k], sel.sum(), allk.sum(),
                 (int(np.median(cc[sel & (cc > 0)])) if (sel & (cc > 0)).any() else "-"), (int(cc[sel].max()) if sel.any() else "-"),
                 (int(np.median(mm_[sel & (cc > 0)])) if (sel & (cc > 0)).any() else "-"),
                 (int(mm_[sel & (cc > 0)].max()) if (sel & (cc > 0)).any() else "-"),
                 is_copy.sum(), (int(np.median(mm_[is_copy])) if is_copy.any() else "-"), (int(mm_[is_copy].max()) if is_copy.any() else "-"),
                 float(np.median(t0[sel])) if sel.any() else 0.0, float(np.median((t1 - t0)[sel])) if sel.any() else 0.0))
# duration by (class, c) and by m, all waves
for k in (0, 1, 2):
    sel = live & (kind == k)
    if not sel.any():
        continue
    rows_ = []
    for cv in sorted(set(cc[sel].tolist()))[:20]:
        s2 = sel & (cc == cv)
        rows_.append("c=%d:%d waves dur %.2f end %.2f" % (cv, s2.sum(), np.median((t1 - t0)[s2]), np.median(t1[s2])))
    print("   %s by c: %s" % (names[k], "; ".join(rows_)))
idx = np.nonzero(live)[0]
order = idx[np.argsort(-t1[idx])][:12]
print("latest waves (block, wave, role, kind, start, end):",
      [(int(i // WPW), int(i % WPW), int(role[i]), int(kind[i]), round(float(t0[i]), 2), round(float(t1[i]), 2)) for i in order])
