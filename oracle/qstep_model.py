"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/, never by herald_amd/): numpy model of the
floating-point order of ha_qstep_* (herald_amd/csrc/qstep.hip).

The reference applies a sparse SGD step occurrence by occurrence, `row[j] -= lr * g[i][j]` in the order the ids
appear (cpu_SGDOptimizerSparseUpdate, /root/reference/src/dnnl_ops/Optimizers.cpp:65-72).  ha_qstep_* keeps that
serial chain for keys with fewer than 16 occurrences in a batch and subtracts a fixed TREE SUM of the lr * g_i for
keys with more (BASELINE.json's north star allows 1e-5 relative on accumulated gradients).  This file restates the
tree so that the tolerance mode can ALSO be held bit for bit -- it is deterministic --, next to the tolerance check
against the serial chain (oracle/cpu.py).  Every operation is a separate float32 rounding, as in the kernel."""
import numpy as np

F = np.float32
LONG_MIN, COOP_MIN = 16, 64      # kQMediumC + 1, kQLongC of qstep.hip


def _acc(p, lr, g):
    return (p + (F(lr) * g).astype(F)).astype(F)


def tree_long(rows_g, lr):
    """16 <= c < 64: lane group r sums occurrences r, r+8, ... in order; the eight sums meet as
    ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7))."""
    c, width = rows_g.shape
    p = [np.zeros(width, F) for _ in range(8)]
    for t in range(8):
        for r in range(8):
            occ = 8 * t + r
            if occ < c:
                p[r] = _acc(p[r], lr, rows_g[occ])
    s1 = [(p[0] + p[1]).astype(F), (p[2] + p[3]).astype(F), (p[4] + p[5]).astype(F), (p[6] + p[7]).astype(F)]
    s2 = [(s1[0] + s1[1]).astype(F), (s1[2] + s1[3]).astype(F)]
    return (s2[0] + s2[1]).astype(F)


def tree_coop4(rows_g, lr):
    """The workgroup items of the 256-thread launch geometry (csrc/qstep.hip built with -DQV_GOLD=0 -DQV_WG=256: a timing
    variant, not the product -- the product's workgroup items are tree_coop below): FOUR waves x lane group r (8): wave w sums the
    occurrences base + 64 w + 8 t + r, t = 0..7, over the blocks of 256 in order; per wave the eight sums meet as
    ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)) (an L item's tree); the four waves as (w0+w1) + (w2+w3)."""
    c, width = rows_g.shape
    part = []
    for w in range(4):
        p = [np.zeros(width, F) for _ in range(8)]
        for base in range(0, c, 256):
            mine = base + 64 * w
            if mine >= c:
                break
            for t in range(8):
                for r in range(8):
                    occ = mine + 8 * t + r
                    if occ < c:
                        p[r] = _acc(p[r], lr, rows_g[occ])
        s1 = [(p[0] + p[1]).astype(F), (p[2] + p[3]).astype(F), (p[4] + p[5]).astype(F), (p[6] + p[7]).astype(F)]
        part.append((((s1[0] + s1[1]).astype(F)) + ((s1[2] + s1[3]).astype(F))).astype(F))
    return (((part[0] + part[1]).astype(F)) + ((part[2] + part[3]).astype(F))).astype(F)


def tree_coop(rows_g, lr):
    """ha_qstep's workgroup items (c >= 64; csrc/qstep.hip q_coop_r3) and the tolerance mode of the plan-driven applies
    (csrc/scatter_dev.h coop_slice_tree), the same tree: wave w (16) x lane
    group r (4): occurrences base + 16 w + 4 t + r over blocks of 256 in order; per wave
    (p0+p1)+(p2+p3); waves 4q..4q+3 as (a+b)+(c+d); the four quads as (q0+q1)+(q2+q3)."""
    c, width = rows_g.shape
    part = []
    for w in range(16):
        p = [np.zeros(width, F) for _ in range(4)]
        for base in range(0, c, 256):
            mine = base + 16 * w
            if mine >= c:
                break
            for t in range(4):
                for r in range(4):
                    occ = mine + 4 * t + r
                    if occ < c:
                        p[r] = _acc(p[r], lr, rows_g[occ])
        part.append((((p[0] + p[1]).astype(F)) + ((p[2] + p[3]).astype(F))).astype(F))
    quad = [(((part[4 * q] + part[4 * q + 1]).astype(F)) + ((part[4 * q + 2] + part[4 * q + 3]).astype(F))).astype(F)
            for q in range(4)]
    return (((quad[0] + quad[1]).astype(F)) + ((quad[2] + quad[3]).astype(F))).astype(F)


CHUNK = 256      # kQChunk of qstep.hip


def tree_coop_chunked(rows_g, lr):
    """ha_qstep's workgroup items: a key with more than 256 occurrences is cut into chunks of 256 (csrc/qstep.hip q_emit_words /
    q_coop_r3); every chunk is summed by a workgroup of its own with the tree above, and the chunk sums are added up in chunk
    order by whichever workgroup finishes last: ((p0 + p1) + p2) + ..."""
    c = rows_g.shape[0]
    if c <= CHUNK:
        return tree_coop(rows_g, lr)
    tot = None
    for lo in range(0, c, CHUNK):
        p = tree_coop(rows_g[lo:lo + CHUNK], lr)
        tot = p if tot is None else (tot + p).astype(F)
    return tot


def listed_chunking(ids, width):
    """Whether the plan-driven applies of a FINISHED plan cut runs beyond 256 occurrences into chunks in tolerance mode 2
    (ha_set_tolerance_mode(2); csrc/scatter.hip apply_listed_kernel / apply_chunk_plan_kernel; the conditions are functions of the batch alone): a batch of
    36,865 .. 2^20 ids, rows of at most 256 floats (a multiple of 4), at most 2,040 keys with 48+ occurrences, at least one run
    beyond 256, and the chunk sums fit the plan's 2 n scratch words."""
    ids = np.asarray(ids).reshape(-1)
    n = ids.size
    if not (36864 < n <= (1 << 20)) or width % 4 or width > 256:
        return False
    _, cnt = np.unique(ids, return_counts=True)
    listed = cnt[cnt >= 48]
    if listed.size == 0 or listed.size > 2040:
        return False
    total = int(np.where((listed >= 64) & (listed > CHUNK), (listed + CHUNK - 1) // CHUNK, 1).sum())
    nslice = (width + 63) // 64
    return total > listed.size and total * nslice * 64 <= 2 * n


def sgd_sparse_update(table, ids, grads, lr, long_min=LONG_MIN, coop_min=COOP_MIN, mode="sgd", chunked=False):
    """In place: table after one ha_qstep apply of (ids, grads).  ids: integer array (keys beyond the table are
    ignored, as by the library).

    long_min=None, coop_min=64 restates the library's TOLERANCE MODE (ha_set_tolerance_mode, csrc/scatter_dev.h
    coop_slice_tree: the same sixteen-wave tree as ha_qstep's workgroup items, from 64 occurrences -- over the WHOLE run,
    where ha_qstep cuts runs beyond 256 occurrences into chunks; chunked=True: so do the applies of a finished plan when
    listed_chunking(ids, width) holds --; everything shorter is the serial chain).  mode: "sgd" row - sum(lr*g) / chain row -= lr*g;  "push" row + sum(g) (lr is ignored: the
    library reduces from 0 in order, then adds once -- ha_push_apply);  the reduced rows of ha_dedup_reduce_scaled
    are mode="push" on a zero table with grads pre-scaled."""
    ids = np.asarray(ids).reshape(-1).astype(np.int64)
    grads = np.asarray(grads, dtype=F).reshape(ids.size, -1)
    order = np.argsort(ids, kind="stable")
    sk = ids[order]
    starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]]) if ids.size else np.zeros(0, np.int64)
    ends = np.r_[starts[1:], ids.size]
    one = F(1.0) if mode == "push" else F(lr)
    for s, e in zip(starts, ends):
        key = int(sk[s])
        if key < 0 or key >= table.shape[0]:
            continue
        occ = order[s:e]                      # occurrence order (stable sort)
        c = e - s
        if c >= coop_min:
            # (the tolerance mode of the plan-driven applies -- long_min None -- keeps one tree for the whole run)
            t = tree_coop(grads[occ], one) if long_min is None and not chunked else tree_coop_chunked(grads[occ], one)
        elif long_min is not None and c >= long_min:
            t = tree_long(grads[occ], one)
        else:
            t = None
        if t is not None:
            table[key] = (table[key] - t).astype(F) if mode == "sgd" else (table[key] + t).astype(F)
        elif mode == "sgd":
            row = table[key].copy()
            for i in occ:
                row = (row - (F(lr) * grads[i]).astype(F)).astype(F)
            table[key] = row
        else:
            acc = np.zeros(table.shape[1], F)
            for i in occ:
                acc = (acc + grads[i]).astype(F)
            table[key] = (table[key] + acc).astype(F)
    return table


def tolerance(ids, grads, lr, rows, rel=1e-5, tree_min=LONG_MIN):
    """Per-row absolute bound of the difference between the tree mode and the serial chain: `rel` x the accumulated
    gradient magnitude lr * sum_i |g_i| of the row's key (zero for keys below `tree_min` occurrences: exact)."""
    ids = np.asarray(ids).reshape(-1).astype(np.int64)
    g = np.abs(np.asarray(grads, dtype=F).reshape(ids.size, -1)).astype(np.float64)
    ok = (ids >= 0) & (ids < rows)
    tol = {}
    for k in np.unique(ids[ok]):
        sel = ids == k
        cnt = int(sel.sum())
        tol[int(k)] = (rel * lr * g[sel].sum(axis=0)) if cnt >= tree_min else np.zeros(g.shape[1])
    return tol
