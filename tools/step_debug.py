"""Debug driver for ha_sgd_push_pull: back-to-back launches (eager / graph), checks time-outs, pending
tables and parity against the oracle at the end."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from herald_amd import ops, synth
from oracle import cpu

dev = torch.device("cuda:0")
rows, width, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, 512, int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = sys.argv[3] if len(sys.argv) > 3 else "eager"
rng = np.random.default_rng(1)
table0 = (rng.standard_normal((rows, width), dtype=np.float32) * np.float32(0.01))
batches = [(synth.criteo_batch(256, step=s).reshape(-1) % rows).astype(np.float32) for s in range(steps + 1)]
grads = [rng.standard_normal((6656, width), dtype=np.float32) for _ in range(3)]
table = torch.from_numpy(table0.copy()).to(dev)
d_ids = [torch.from_numpy(b).to(dev) for b in batches]
d_g = [torch.from_numpy(g).to(dev) for g in grads]
outs = [torch.empty((6656, width), dtype=torch.float32, device=dev) for _ in range(steps + 1)]
plans = [ops.IndexPlan(6656, dev), ops.IndexPlan(6656, dev)]
pends = [ops.PendingTable(dev), ops.PendingTable(dev)]
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    ops.lookup_sort_pend(table, d_ids[0], plans[0], pends[0], out=outs[0], stream=s)

def step(k):
    ops.sgd_push_pull(table, plans[k % 2], d_g[k % 3], 0.01, pends[k % 2], d_ids[k + 1], plans[(k + 1) % 2],
                      pends[(k + 1) % 2], next_out=outs[k + 1], stream=s)

torch.cuda.synchronize()
t0 = time.perf_counter()
if mode == "graph":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for k in range(steps):
            step(k)
    with torch.cuda.stream(s):
        g.replay()
else:
    with torch.cuda.stream(s):
        for k in range(steps):
            step(k)
            if mode == "sync":
                torch.cuda.synchronize()
torch.cuda.synchronize()
print("mode", mode, "elapsed ms/step", 1e3 * (time.perf_counter() - t0) / steps)
print("timeouts", plans[0].handoff_timed_out(), plans[1].handoff_timed_out())
for i, p in enumerate(pends):
    w = p.buf.view(torch.int32)
    nz = torch.nonzero(w).reshape(-1)
    print("pend", i, "nonzero words", nz.numel(), w[nz][:8].tolist())
want = table0.copy()
bad = 0
for k in range(steps + 1):
    wo = cpu.embedding_lookup(want, batches[k])
    got = outs[k].cpu().numpy()
    if not np.array_equal(got, wo):
        rowsbad = np.nonzero((got != wo).any(axis=1))[0]
        print("step", k, "bad out rows", rowsbad.size, rowsbad[:5], "keys", batches[k][rowsbad[:5]])
        bad += 1
    if k < steps:
        cpu.sgd_sparse_update(want, batches[k], grads[k % 3], 0.01)
print("bad steps", bad, "table equal", np.array_equal(table.cpu().numpy(), want))
