"""Development aid: the Criteo stream of tests/test_gpu_qstep.py::test_qspan_criteo_stream through spanning launches; on a
mismatch, which keys / columns / item classes are wrong."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from herald_amd import ops, synth
from oracle import qstep_model

dev = torch.device("cuda:0")
rows, width, bs = 400_000, 512, 256
block = int(sys.argv[1]) if len(sys.argv) > 1 else 16
span = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sync = sys.argv[3] if len(sys.argv) > 3 else "flags"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.default_rng(31)
table0 = (rng.standard_normal((rows, width), dtype=np.float32) * np.float32(0.01))
steps = 40
batches = [synth.criteo_batch(bs, step=k, rows=rows).reshape(-1) for k in range(steps)]
grads = [rng.standard_normal((b.size, width), dtype=np.float32) for b in batches]
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for rep in range(reps):
    table = d(table0)
    model_t = table0.copy()
    pipe = ops.QueueStepPipeline(table, batches[0].size, 0.01, overlap=True, block=block, sync=sync, span=True)
    L = pipe.LOOKAHEAD
    d_ids = [d(b.astype(np.float32)) for b in batches]
    d_g = [d(g) for g in grads]
    out = pipe.start(d_ids[:L])
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(-1, width), model_t[batches[0]])
    k = 0
    bad = 0
    while k < steps:
        cnt = min(span, steps - k, block - k % block)
        outs = pipe.step_span(d_g[k:k + cnt], [d_ids[k + i + L] if k + i + L < steps else None for i in range(cnt)])
        torch.cuda.synchronize()
        for i in range(cnt):
            qstep_model.sgd_sparse_update(model_t, batches[k + i].astype(np.int64), grads[k + i], 0.01)
            if k + i + 1 < steps:
                ids = batches[k + i + 1].astype(np.int64)
                got = outs[i].cpu().numpy().reshape(-1, width)
                want = model_t[ids]
                neq = got != want
                if neq.any():
                    bad += 1
                    pos = np.nonzero(neq.any(axis=1))[0]
                    keys = np.unique(ids[pos])
                    print("rep %d: lookup of batch %d (step %d, %d-th of its span): %d positions, %d keys" %
                          (rep, k + i + 1, k + i, i, pos.size, keys.size))
                    for key in keys[:6]:
                        p0 = pos[ids[pos] == key][0]
                        cols = np.nonzero(neq[p0])[0]
                        hist = [int((batches[j] == key).sum()) for j in range(max(0, k + i - 4), min(steps, k + i + 3))]
                        # is the wrong value the row BEFORE this step's update (stale) ?
                        print("   key %d: cols %d..%d (%d), occurrences in batches %d..: %s, m=%d; got-want max %.3g"
                              % (key, cols.min(), cols.max(), cols.size, max(0, k + i - 4), hist, int((ids == key).sum()),
                                 np.abs(got[p0] - want[p0]).max()))
        k += cnt
    torch.cuda.synchronize()
    tb = table.cpu().numpy()
    print("rep %d: %d bad lookups; table equal: %s; overflowed %s" % (rep, bad, np.array_equal(tb, model_t), pipe.overflowed()))
