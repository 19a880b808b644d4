#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the REFERENCE's own code
(oracle/_ref/libherald_ref.so, built from /root/reference by oracle/build_ref.sh).  Run in the build
container only; the JSON files are committed and travel, the reference does not.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref  # noqa: E402


def policy_trace(kind, limit, nkeys, nops, seed):
    rng = np.random.default_rng(seed)
    p = ref.Policy(kind, limit)
    ops, res = [], []
    for _ in range(nops):
        k = int(min(rng.zipf(1.4) - 1, nkeys - 1)) if rng.random() < 0.7 else int(rng.integers(0, nkeys))
        if rng.random() < 0.5:
            hit = p.lookup(k)
            ops.append(["l", k])
            res.append({"hit": hit, "size": p.size(), "evicted": p.take_evicted()})
        else:
            upd = int(rng.integers(0, 3))
            p.insert(k, upd)
            ops.append(["i", k, upd])
            res.append({"size": p.size(), "evicted": p.take_evicted()})
    final = [k for k in range(nkeys) if p.count(k)]
    return {"kind": kind, "limit": limit, "nkeys": nkeys, "ops": ops, "results": res, "final_keys": final}


def minilru_trace(cap, nkeys, nops, seed):
    rng = np.random.default_rng(seed)
    c = ref.MiniLRU(cap)
    ops, res = [], []
    for _ in range(nops):
        k = int(rng.integers(0, nkeys))
        r = rng.random()
        if r < 0.5:
            ops.append(["get", k]); res.append(c.get(k))
        elif r < 0.8:
            ops.append(["check", k]); res.append(c.check(k))
        else:
            ops.append(["outdate", k]); c.outdate(k); res.append(None)
    return {"capacity": cap, "nkeys": nkeys, "ops": ops, "results": res, "final_valid_keys": c.keys()}


def unique_vectors(seed):
    rng = np.random.default_rng(seed)
    out = []
    for n, hi in ((0, 5), (1, 5), (17, 4), (200, 50), (1000, 33762577)):
        keys = rng.integers(0, hi, size=n).astype(np.uint64)
        u, inv = ref.unique(keys)
        out.append({"keys": keys.tolist(), "uniq": u.tolist(), "inverse": inv.tolist()})
    return out


def main():
    assert ref.available(), "build oracle/_ref first (oracle/build_ref.sh)"
    for kind in ("lru", "lfu", "lfuopt"):
        traces = [policy_trace(kind, 8, 40, 400, 1), policy_trace(kind, 3, 12, 300, 2),
                  policy_trace(kind, 25, 30, 300, 3)]
        json.dump(traces, open(os.path.join(HERE, "cache_policy_%s.json" % kind), "w"))
    json.dump([minilru_trace(6, 30, 500, 4), minilru_trace(50, 60, 500, 5)],
              open(os.path.join(HERE, "minilru.json"), "w"))
    json.dump(unique_vectors(6), open(os.path.join(HERE, "unique.json"), "w"))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
