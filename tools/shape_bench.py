#!/usr/bin/env python3
"""The work-queue step at the per-GPU shapes of BASELINE configs[2] / configs[3] (bs=4096 d=128: 106,496 ids per step;
bs=1024 d=512: 26,624) on the full 33,762,577-row table: the WIDE path of ops.QueueStepPipeline (hash buckets, no sort),
blocks prepared on a side stream inside the timed region.  Prints us per step (HIP events over the timed steps),
algorithmic bytes N*(12d+8) + U*8d and the fraction of 8 TB/s.  Development aid + the source of profiles/r04/shape_*.txt.
env: BATCH, WIDTH, BLOCK (steps per block, default 4), STEPS, SYNC (flags | events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows, width, bs = 33762577, int(os.environ.get("WIDTH", "128")), int(os.environ.get("BATCH", "4096"))
block, steps, sync = int(os.environ.get("BLOCK", "4")), int(os.environ.get("STEPS", "96")), os.environ.get("SYNC", "flags")
n = bs * 26
table = torch.empty((rows, width), device=dev)
for s in range(0, rows, 1 << 21):
    table[s:s + (1 << 21)].normal_(0, 0.01)
NB = 32
host = [np.minimum(synth.as_f32_ids(synth.criteo_batch(bs, b, rows=rows)).reshape(-1), rows - 1) for b in range(NB)]
ids = [torch.from_numpy(h).to(dev) for h in host]
u_mean = float(np.mean([np.unique(h).size for h in host]))
nbuf = max(2, min(24, (400 << 20) // (n * width * 4)))
grads = [torch.randn((n, width), device=dev) for _ in range(nbuf)]
outs = [torch.empty((n, width), device=dev) for _ in range(nbuf)]
pipe = ops.QueueStepPipeline(table, n, 1e-6, block=block, sync=sync)
print("n=%d unique~%.0f width=%d block=%d sync=%s wide=%s buckets=%s buffers=%d" % (
    n, u_mean, width, block, pipe.sync, pipe.wide, getattr(pipe.plans[0], "buckets", 1), nbuf))
LA = pipe.LOOKAHEAD
s = torch.cuda.Stream(device=dev)
ids_of = lambda j: ids[j % NB] if j >= 0 else None
warm = 2 * block * 3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    for c in range(-LA, warm + steps):
        if c % block == 0:
            pipe.prepare_block(c // block, ids_of, stream=s)
        if c < -1:
            continue
        if c == warm:
            e0.record(s)
        pipe.apply(c, grads[c % nbuf] if c >= 0 else None, outs[(c + 1) % nbuf], stream=s, n_cur=n if c >= 0 else 0, n_next=n)
    e1.record(s)
torch.cuda.synchronize()
assert not pipe.overflowed()
us = e0.elapsed_time(e1) * 1e3 / steps
alg = n * (12 * width + 8) + u_mean * 8 * width
print("work-queue step (wide path): %.2f us/step  %.1f M rows/s  algorithmic %.1f MB  %.2f TB/s  frac %.3f of 8 TB/s"
      % (us, n / us, alg / 1e6, alg / us / 1e6, alg / us / 1e6 / 8.0))
print("queue of a step:", pipe.queue_header(warm + steps - 1))
