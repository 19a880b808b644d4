#!/usr/bin/env python3
"""The HET cache tier of BASELINE configs[1] through the PLANNED flow (csrc/cache_block.hip), development aid: LRU,
limit = 0.1 x rows, bound 100, wdl_criteo bs=256 d=512 batches; cache filled to its limit first (planned pairs as well), then
timed blocks of 16 pairs.  POLICY (LRU / LFU / LFUOpt) / ROWS / BLOCKS / CLASSIC=1 (the call-by-call flow beside it) / KTRACE=<dir of a rocprofv3
--kernel-trace --output-format csv run>: per-kernel averages and the row stream's idle time from the trace."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from herald_amd import cache as hcache, synth


def main():
    dev = torch.device("cuda:0")
    rows = int(os.environ.get("ROWS", "33762577"))
    width, bs = int(os.environ.get("WIDTH", "512")), 256
    n = bs * 26
    table = torch.empty((rows, width), device=dev)
    for s in range(0, rows, 1 << 20):
        table[s:s + (1 << 20)].normal_(0, 0.01)
    versions = torch.zeros(rows, dtype=torch.int64, device=dev)
    hcache.register_table(0, table, versions)
    limit = int(0.1 * rows)
    dummies = [hcache.CacheSparseTable(limit, rows, width, 0, "LRU", bound=100, max_batch=n, device=dev)
               for _ in range(int(os.environ.get("DUMMIES", "0")))]       # (earlier instances that stay allocated, as in bench.py)
    c = hcache.CacheSparseTable(limit, rows, width, 0, os.environ.get("POLICY", "LRU"), bound=100, max_batch=n, device=dev)
    NB = 512
    ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(bs, b, rows=rows)).reshape(-1), rows - 1)).to(dev)
           for b in range(NB)]
    out = torch.empty((n, width), device=dev)
    grad = torch.randn((n, width), device=dev) * 1e-3
    _skip = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("SKIP_STREAMS", "0")))]     # (shifts the pool)
    _skip_hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(int(os.environ.get("SKIP_STREAMS_HI", "0")))]
    for st in _skip + _skip_hi:                 # (a stream gets its hardware queue when it is first used)
        with torch.cuda.stream(st):
            torch.zeros(8, device=dev).add_(1)
    torch.cuda.synchronize()
    main_s = torch.cuda.Stream(device=dev, priority=-1 if os.environ.get("ROW_PRIO") == "high" else 0)
    c.cache.stream = main_s
    GS = 16
    outs, grads = [out] * GS, [grad] * GS
    # fill: keys lo .. lo + n - 1 per batch, dirty lines (as bench.py's prefill), through the planned flow
    base = torch.arange(n, device=dev)
    fill = [((base + lo) % rows).to(torch.float32) for lo in range(0, limit + n, n)]
    t0 = time.perf_counter()
    with torch.cuda.stream(main_s):
        if os.environ.get("FILL") == "classic":      # as bench.py: call-by-call pairs, then real batches call by call
            for kk in fill:
                c.embedding_lookup(kk, out)
                c.embedding_update(kk, grad, same_as_lookup=True)
            for k in range(int(os.environ.get("CLASSIC_PAIRS", "448"))):
                c.embedding_lookup(ids[k % NB], out)
                c.embedding_update(ids[k % NB], grad, same_as_lookup=True)
        else:
            blocks = [fill[i:i + GS] for i in range(0, len(fill), GS)]
            c.plan_block(blocks[0])
            for b, blk in enumerate(blocks):
                if b + 1 < len(blocks):
                    c.plan_block(blocks[b + 1])
                c.run_planned_pairs(outs[:len(blk)], grads[:len(blk)])
    torch.cuda.synchronize()
    print("fill: %d pairs in %.2f s, size %d / %d" % (len(fill), time.perf_counter() - t0, c.cache.size(), limit))
    blocks = [[ids[j % NB] for j in range(g0, g0 + GS)] for g0 in range(0, NB, GS)]
    nwarm, ntimed = 4, int(os.environ.get("BLOCKS", "16"))
    with torch.cuda.stream(main_s):
        c.plan_block(blocks[0])
        for b in range(nwarm + ntimed):
            if b == nwarm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if b + 1 < nwarm + ntimed:
                c.plan_block(blocks[(b + 1) % len(blocks)])
            c.run_planned_pairs(outs, grads)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print("planned: %.2f us per pair (%d pairs), %.1f M rows/s" % (1e6 * el / (ntimed * GS), ntimed * GS, n * ntimed * GS / el / 1e6))
    if os.environ.get("CLASSIC") == "1":
        with torch.cuda.stream(main_s):
            for k in range(32):
                c.embedding_lookup(ids[k], out)
                c.embedding_update(ids[k], grad, same_as_lookup=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(32, 32 + 128):
                c.embedding_lookup(ids[k % NB], out)
                c.embedding_update(ids[k % NB], grad, same_as_lookup=True)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print("call by call (plain launches): %.2f us per pair" % (1e6 * el / 128))
    c.perf_enabled(True)
    with torch.cuda.stream(main_s):
        c.plan_block(blocks[5][:4])
        for k in range(4):
            c.embedding_lookup_planned(out)
            c.embedding_update_planned(grad)
    torch.cuda.synchronize()
    for r in c.perf[-2:]:
        print({k: v for k, v in r.items() if not k.endswith("time")})


def ktrace(d):
    """Per-kernel averages + what the row stream does, from a rocprofv3 kernel trace (csv)."""
    import csv
    import glob
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rowsx = list(csv.DictReader(open(f)))
    ks = {}
    for r in rowsx:
        nm = r["Kernel_Name"].split("(")[0]
        ks.setdefault(nm, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")))
    for nm, v in sorted(ks.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        if "cache" in nm or "plan" in nm or "finish" in nm or "rank" in nm:
            tail = v[len(v) // 2:]
            print("%-60s calls %6d  avg %8.2f us  (second half: %8.2f us)" % (
                nm[:60], len(v), sum(e - s for s, e, _ in v) / len(v) / 1e3, sum(e - s for s, e, _ in tail) / len(tail) / 1e3))
    # the row stream: lookup / update launches in time order, the gaps between them
    seq = sorted([(s, e, "L") for s, e, _ in ks.get("void ha::cache_lookup_planned_kernel<4>", [])] +
                 [(s, e, "U") for s, e, _ in ks.get("void ha::cache_update_planned_kernel<4>", [])])
    seq = seq[len(seq) // 2:]
    gaps = [seq[i + 1][0] - seq[i][1] for i in range(len(seq) - 1)]
    if gaps:
        g = np.array(gaps) / 1e3
        print("row stream, second half: %d launches, gaps mean %.2f us, p50 %.2f, p90 %.2f, max %.1f; period per pair %.2f us" % (
            len(seq), g.mean(), np.percentile(g, 50), np.percentile(g, 90), g.max(),
            2 * (seq[-1][1] - seq[0][0]) / 1e3 / len(seq)))


if __name__ == "__main__":
    if os.environ.get("KTRACE"):
        ktrace(os.environ["KTRACE"])
    else:
        main()
