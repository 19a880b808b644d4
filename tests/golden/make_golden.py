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
sys.path.insert(0, HERE)
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


def dnnl_cases(seed):
    """cpu_EmbeddingLookup / cpu_SGDOptimizerSparseUpdate of the REFERENCE (oracle/_ref/libref_dnnl.so =
    src/dnnl_ops/EmbeddingLookup.cpp + Optimizers.cpp compiled unchanged) on closed-form tables
    (formula.py): SURVEY 8(c)'s list -- d in {4, 64, 128, 512}, duplicates, a hot id, ids > 2^24,
    empty and single-id batches, 2-D id arrays."""
    import formula
    rng = np.random.default_rng(seed)
    specs = [
        # name, rows, width, ids (float32 array, any rank), lr
        ("d4_dups", 37, 4, rng.integers(0, 37, size=(6, 5)), 0.01),
        ("d4_hot", 11, 4, np.concatenate([np.full(40, 7), rng.integers(0, 11, size=24)])[rng.permutation(64)], 0.5),
        ("d64_dups", 101, 64, rng.integers(0, 101, size=(4, 8)), 0.01),
        ("d64_hot", 9, 64, np.concatenate([np.full(50, 3), rng.integers(0, 9, size=14)])[rng.permutation(64)], 1e-3),
        ("d128_dups", 53, 128, rng.integers(0, 53, size=24), 1e-6),
        ("d512_dups", 29, 512, np.array([5, 28, 5, 0, 17, 5, 28, 11, 0, 5, 3, 28]), 1e-6),
        ("empty", 5, 4, np.zeros((0,), dtype=np.int64), 0.01),
        ("single", 5, 64, np.array([4]), 0.01),
        # float32 ids above 2^24: only even numbers are representable; the reference truncates the float
        ("big_ids_d4", 33762577, 4, np.array([16777216, 16777218, 33762576, 20000002, 16777218, 25000000,
                                               33762576, 1, 0, 16777220], dtype=np.float32), 0.01),
    ]
    out = []
    for name, rows, width, ids, lr in specs:
        ids = np.asarray(ids).astype(np.float32)
        tab = formula.table(rows, width)
        n = ids.size
        grads = rng.standard_normal((n, width)).astype(np.float32)
        got = ref.dnnl_embedding_lookup(tab, ids)
        assert got.shape == ids.shape + (width,)
        upd = tab.copy()
        if n:
            ref.dnnl_sgd_sparse_update(upd, ids.reshape(-1), grads, lr)
        keys = sorted(set(int(x) for x in ids.reshape(-1)))
        changed = np.nonzero((upd != tab).any(axis=1))[0].tolist()
        assert set(changed) <= set(keys)
        out.append({"name": name, "rows": rows, "width": width, "ids_shape": list(ids.shape),
                    "ids_bits": formula.bits(ids), "lr_bits": formula.bits(np.float32(lr))[0],
                    "grads_bits": formula.bits(grads), "out_bits": formula.bits(got),
                    "touched_rows": keys, "touched_bits": formula.bits(upd[keys]) if keys else []})
    return out


def _reference_cpu_deduplicate():
    """IndexedSlices.cpu_deduplicate as the REFERENCE wrote it: the method's text is read from
    /root/reference/python/hetu/ndarray.py at generation time (never copied into the repo) and executed on
    a bare object.  The `hetu` package itself cannot be imported here (it dlopens libc_runtime_api.so), so the
    three names the method takes from its module are bound to array CONTAINERS only -- `array(a, ctx)` wraps a
    numpy array, `cpu(i)` is a tag, `is_gpu_ctx` says no; every arithmetic step (np.unique, the occurrence-order
    `new_values[ind] += flatten[i]` loop in float32) is the reference's own statement."""
    import ast
    import textwrap
    path = "/root/reference/python/hetu/ndarray.py"
    src = open(path).read()
    tree = ast.parse(src)
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "IndexedSlices":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "cpu_deduplicate":
                    fn = item
    assert fn is not None, "IndexedSlices.cpu_deduplicate not found in %s" % path
    text = textwrap.dedent("\n".join(src.splitlines()[fn.lineno - 1:fn.end_lineno]))

    class Box:                      # the NDArray container: data + ctx, nothing else
        def __init__(self, a, ctx=None):
            self.a, self.ctx, self.shape = np.asarray(a), ctx, np.asarray(a).shape

        def asnumpy(self):
            return self.a

    ns = {"np": np, "array": lambda a, ctx=None: Box(a, ctx), "cpu": lambda i: ("cpu", i),
          "is_gpu_ctx": lambda ctx: False}
    exec(compile(text, path, "exec"), ns)
    method = ns["cpu_deduplicate"]

    def run(ids, values, push=None):
        class Slices:
            pass
        sl = Slices()
        sl.indices, sl.values = Box(ids, ("cpu", 0)), Box(values, ("cpu", 0))
        sl.push_indices = None if push is None else Box(push, ("cpu", 0))
        method(sl)
        return sl.indices.a, sl.values.a, (None if push is None else sl.push_indices.a)
    return run


def dedup_cases(seed):
    """IndexedSlices.cpu_deduplicate (python/hetu/ndarray.py:556-576): float32 ids of any rank with duplicates,
    a hot id, ids above 2^24, empty and single-id batches, push_indices; d in {4, 64, 512}."""
    import formula
    from oracle import cpu
    run = _reference_cpu_deduplicate()
    rng = np.random.default_rng(seed)
    specs = [
        ("d4_dups", rng.integers(0, 9, size=(5, 4)), 4, None),
        ("d64_hot", np.concatenate([np.full(70, 3), rng.integers(0, 40, size=58)])[rng.permutation(128)], 64, None),
        ("d512_dups", rng.integers(0, 12, size=20), 512, None),
        ("d4_push", rng.integers(0, 30, size=40), 4, rng.integers(0, 30, size=17)),
        ("empty", np.zeros((0,), dtype=np.int64), 4, None),
        ("single", np.array([6]), 64, None),
        ("big_ids", np.array([16777216, 16777218, 33762576, 20000002, 16777218, 33762576, 1, 0, 16777218],
                             dtype=np.float32), 4, None),
    ]
    out = []
    for name, ids, width, push in specs:
        ids = np.asarray(ids).astype(np.float32)
        vals = rng.standard_normal(ids.shape + (width,)).astype(np.float32)
        pf = None if push is None else np.asarray(push).astype(np.float32)
        # The reference pins numpy 1.20.3 (environment.yml:43), whose np.unique returns a FLAT inverse for ids of
        # any rank; numpy >= 2.0 (this container: 2.2) returns it in the ids' shape, which would turn the method's
        # `for i, ind in enumerate(inverse)` into a loop over rows.  Ids are therefore handed over flattened -- the
        # method flattens the values itself (`reshape((-1, last_dim))`), so under numpy 1.20 the result is the same.
        u, red, pu = run(ids.reshape(-1), vals, pf)
        # the repo's restatement of the same loop (oracle/cpu.py) must agree before the vector is written
        ru, rred = cpu.np_cpu_deduplicate(ids, vals)
        assert np.array_equal(u, ru) and np.array_equal(red, rred), name
        assert red.dtype == np.float32
        out.append({"name": name, "width": width, "ids_shape": list(ids.shape), "ids_bits": formula.bits(ids),
                    "values_bits": formula.bits(vals), "uniq_bits": formula.bits(np.asarray(u, dtype=np.float32)),
                    "reduced_bits": formula.bits(red),
                    "push_bits": None if pf is None else formula.bits(pf),
                    "push_uniq_bits": None if pu is None else formula.bits(np.asarray(pu, dtype=np.float32))})
    return out


def main():
    assert ref.available(), "build oracle/_ref first (oracle/build_ref.sh)"
    assert ref.dnnl_available(), "build oracle/_ref/libref_dnnl.so first (oracle/build_ref.sh)"
    json.dump(dnnl_cases(7), open(os.path.join(HERE, "dnnl_ops.json"), "w"))
    for kind in ("lru", "lfu", "lfuopt"):
        traces = [policy_trace(kind, 8, 40, 400, 1), policy_trace(kind, 3, 12, 300, 2),
                  policy_trace(kind, 25, 30, 300, 3)]
        json.dump(traces, open(os.path.join(HERE, "cache_policy_%s.json" % kind), "w"))
    json.dump([minilru_trace(6, 30, 500, 4), minilru_trace(50, 60, 500, 5)],
              open(os.path.join(HERE, "minilru.json"), "w"))
    json.dump(unique_vectors(6), open(os.path.join(HERE, "unique.json"), "w"))
    json.dump(dedup_cases(8), open(os.path.join(HERE, "dedup.json"), "w"))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
