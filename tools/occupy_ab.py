#!/usr/bin/env python3
"""What does a kernel that only HOLDS wave slots take from the wide path's apply launch?  The apply launch of one step again
and again (queue stays built), alone and beside an occupant on a second stream: W workgroups of T threads with L bytes of LDS
that sleep for as long as an apply launch takes and touch no memory (ha_debug_occupy).  The preparation kernels of the wide
path are 128-256 workgroups of 1,024 threads resident ~90 % of the time (profiles/r05/wide_streams_timeline.txt)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth, _lib
dev = torch.device("cuda:0")
rows, width, bs = 33762577, int(os.environ.get("WIDTH", "128")), int(os.environ.get("BATCH", "4096"))
L = _lib.load()
table = torch.empty((rows, width), device=dev)
for s0 in range(0, rows, 1 << 21):
    table[s0:s0 + (1 << 21)].normal_(0, 0.01)
n, block, distinct = bs * 26, 4, 16
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(bs, b, rows=rows)).reshape(-1), np.float32(rows - 1))).to(dev)
       for b in range(distinct)]
nbuf = max(2, min(24, (400 << 20) // (n * width * 4)))
grads = [torch.randn((n, width), device=dev) for _ in range(nbuf)]
outs = [torch.empty((n, width), device=dev) for _ in range(nbuf)]
pipe = ops.QueueStepPipeline(table, n, 1e-6, block=block, sync="events")
LA = pipe.LOOKAHEAD
s, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ids_of = lambda j: ids[j % distinct] if j >= 0 else None
with torch.cuda.stream(s):
    for c in range(-LA, 3 * block):
        if c % block == 0:
            pipe.prepare_block(c // block, ids_of, stream=s)
        if c >= -1:
            pipe.apply(c, grads[c % nbuf] if c >= 0 else None, outs[(c + 1) % nbuf], stream=s, n_cur=n if c >= 0 else 0, n_next=n)
torch.cuda.synchronize()
c = 3 * block - 2
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def run(occ):
    reps = 40
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for i in range(reps + 4):
            if i == 4:
                e0.record(s)
            if occ is not None:      # one occupant launch per apply launch, as long as the apply takes (they queue up on `side`)
                _lib.check(L.ha_debug_occupy(occ[0], occ[1], occ[2], int(occ[3] * 100), ctypes.c_void_p(side.cuda_stream)), "occupy")
            pipe.apply(c, grads[i % nbuf], outs[(i + 1) % nbuf], stream=s, n_cur=n, n_next=n)
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
base = run(None)
print("bs=%d d=%d apply launch alone: %.2f us" % (bs, width, base))
for occ in ((256, 1024, 32 << 10, 40), (128, 1024, 110 << 10, 40), (128, 1024, 32 << 10, 40), (256, 256, 32 << 10, 40), (128, 256, 110 << 10, 40),
            (128, 256, 28 << 10, 40), (512, 64, 28 << 10, 40)):
    t = run(occ)
    print("  beside %4d workgroups x %4d threads, %3d KB LDS, asleep %d us each: %.2f us  (+%.1f %%)   [%5d waves = %4.1f %% of the slots]"
          % (occ[0], occ[1], occ[2] >> 10, occ[3], t, 100 * (t / base - 1), occ[0] * occ[1] // 64, 100 * occ[0] * occ[1] / 64 / 8192))
