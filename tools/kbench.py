#!/usr/bin/env python3
"""Per-kernel micro-timings on the GPU (development aid): every case is `reps` back-to-back launches
bracketed by one pair of events, so the number is launch-amortised device time per launch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from herald_amd import ops, synth


def timeit(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=33762577)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--case", default="all")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n = args.batch * args.fields
    table = torch.empty((args.rows, args.width), dtype=torch.float32, device=dev)
    table.normal_(0, 0.01) if args.rows <= (1 << 22) else [table[s:s + (1 << 20)].normal_(0, 0.01)
                                                            for s in range(0, args.rows, 1 << 20)]
    out = torch.empty((n, args.width), dtype=torch.float32, device=dev)
    grads = torch.randn((n, args.width), dtype=torch.float32, device=dev)
    nb = 64
    cr = np.stack([np.minimum(synth.as_f32_ids(synth.criteo_batch(args.batch, b, rows=args.rows,
                                                                   nfields=args.fields)).reshape(-1),
                              np.float32(args.rows - 1)) for b in range(nb)])
    rng = np.random.default_rng(0)
    distinct = np.stack([rng.choice(min(args.rows, 1 << 24), size=n, replace=False).astype(np.float32)
                         for _ in range(nb)])
    hot = np.stack([(rng.integers(0, 3, size=n) * 1000).astype(np.float32) for _ in range(nb)])
    mid = np.stack([(rng.integers(0, n // 8, size=n) * 7).astype(np.float32) for _ in range(nb)])
    cases = {"criteo": torch.from_numpy(cr).to(dev), "distinct": torch.from_numpy(distinct).to(dev),
             "hot3": torch.from_numpy(hot).to(dev), "mid8": torch.from_numpy(mid).to(dev)}
    if args.case != "all":
        cases = {args.case: cases[args.case]}
    plan = ops.IndexPlan(n, dev)
    print("rows=%d width=%d n=%d" % (args.rows, args.width, n))
    for name, ids in cases.items():
        k = [0]

        def nxt():
            k[0] += 1
            return ids[k[0] % nb]
        t_g = timeit(lambda: ops.embedding_lookup(table, nxt(), out=out))
        t_s = timeit(lambda: plan.sort(nxt()))
        plan.sort(ids[0])
        t_f = timeit(lambda: plan.finish())
        # apply against a fixed plan (same rows every launch: L2/MALL warm) and a fresh plan each time
        t_a_fixed = timeit(lambda: ops.sgd_apply(table, plan, grads, 1e-6))

        def sort_apply():
            plan.sort(nxt())
            ops.sgd_apply(table, plan, grads, 1e-6)
        t_sa = timeit(sort_apply)
        plan.build(ids[0])
        red = torch.empty((n, args.width), dtype=torch.float32, device=dev)
        t_r = timeit(lambda: ops.dedup_reduce(plan, grads, out=red))
        print("%-9s gather %.2f us | sort %.2f | finish %.2f | apply(fixed plan) %.2f | sort+apply %.2f "
              "| dedup_reduce %.2f" % (name, t_g, t_s, t_f, t_a_fixed, t_sa, t_r))


if __name__ == "__main__":
    main()
