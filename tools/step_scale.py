"""ha_sgd_push_pull timing / time-out probe at large table sizes (no parity: the table lives on the GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import ops, synth

dev = torch.device("cuda:0")
rows = int(sys.argv[1]); steps = int(sys.argv[2]); width = 512
clampmode = sys.argv[3] if len(sys.argv) > 3 else "min"
table = torch.empty((rows, width), dtype=torch.float32, device=dev)
table.normal_(0, 0.01)
nb = 64
ids = []
for b in range(nb):
    f = synth.as_f32_ids(synth.criteo_batch(256, step=b, rows=rows, nfields=26)).reshape(-1)
    if clampmode == "min":
        np.minimum(f, np.float32(rows - 1), out=f)
    ids.append(torch.from_numpy(f).to(dev))
g = [torch.randn((6656, width), device=dev) for _ in range(4)]
outs = [torch.empty((6656, width), dtype=torch.float32, device=dev) for _ in range(4)]
plans = [ops.IndexPlan(6656, dev), ops.IndexPlan(6656, dev)]
pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    ops.lookup_sort_pend(table, ids[0], plans[0], pends[0], out=outs[0], stream=s)
    torch.cuda.synchronize()
    for k in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        ops.sgd_push_pull(table, plans[k % 2], g[k % 4], 1e-6, pends[k % 2], ids[(k + 1) % nb], plans[(k + 1) % 2],
                          pends[(k + 1) % 2], next_out=outs[(k + 1) % 4], stream=s)
        e1.record(s)
        torch.cuda.synchronize()
        w = pends[k % 2].buf.view(torch.int32)
        nz = torch.nonzero(w).reshape(-1)
        print("step", k, "ms", round(e0.elapsed_time(e1), 4), "timeouts", plans[0].handoff_timed_out(), plans[1].handoff_timed_out(),
              "drained-table nonzero", nz.numel(), w[nz][:6].tolist())
