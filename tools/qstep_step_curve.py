"""Per-step durations of the work-queue step right after a short warm-up (development aid): which steps of a 20-step run
cost more than the steady state -- the first ones, or the ones beside a block start?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import ops, synth
dev = torch.device("cuda:0")
rows, width, n, Bk = 33762577, 512, 6656, 16
table = torch.empty((rows, width), device=dev)
for s in range(0, rows, 1 << 21):
    table[s:s + (1 << 21)].normal_(0, 0.01)
nb = 256
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(nb)]
grads = [torch.randn((n, width), device=dev) for _ in range(24)]
outs = [torch.empty((n, width), device=dev) for _ in range(24)]
for rep in range(3):
    pipe = ops.QueueStepPipeline(table, n, 1e-6, block=Bk)
    LA = pipe.LOOKAHEAD
    ids_of = lambda j: ids[j % nb] if j >= 0 else None
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        for c in range(-LA, 0):
            if c % Bk == 0: pipe.prepare_block(c // Bk, ids_of, stream=main)
        pipe.apply(-1, None, outs[0], stream=main, n_cur=0, n_next=n)
        torch.cuda.synchronize()
        W, K = 5, 40
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(W + K + 1)]
        for k in range(W + K):
            if k % Bk == 0: pipe.prepare_block(k // Bk, ids_of, stream=main)
            ev[k].record(main)
            pipe.apply(k, grads[k % 24], outs[(k + 1) % 24], stream=main, n_cur=n, n_next=n)
        ev[W + K].record(main)
        torch.cuda.synchronize()
    d = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(W + K)]
    print("rep %d:" % rep, " ".join("%d:%.1f" % (k, d[k]) for k in range(W + K)))
