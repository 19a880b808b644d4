"""CPU restatement of the laia embedding scheduler (reference laia/).  TEST INFRASTRUCTURE ONLY.

Follows laia/include/mini_lru_cache.h:14-137 (MiniLRUCache, hash mode) and
laia/src/laia_scheduler.cc:115-271 (LaiaScheduler::launch / get_dist).  Pure Python: the traces the
tests use are a few thousand samples.

Parity pin: MiniLRU is pinned against the reference's own header compiled from /root/reference
(oracle/_ref, tests/golden/minilru.json).  LaiaScheduler itself needs Boost (flat_set,
laia/include/utils.h) which this image lacks, so it cannot be built here without a stand-in header:
its restatement below is "parity unpinned" beyond MiniLRU (the reference has no asserting test for
it either: laia/test/test_laia_scheduler.py only prints lengths).
"""
from collections import OrderedDict

import numpy as np


def _as_int(key):
    """emb_key_t (uint64) passed to `int key` parameters (mini_lru_cache.h:54,66): truncation to int32."""
    k = int(key) & 0xFFFFFFFF
    return k - (1 << 32) if k >= (1 << 31) else k


class MiniLRU:
    def __init__(self, capacity):
        self.cap = capacity
        self.od = OrderedDict()            # key -> valid flag ; last item = list front

    def check(self, key):                                                    # :54-63
        return self.od.get(_as_int(key), False) is True

    def get(self, key):                                                      # :69-87
        key = _as_int(key)
        if key not in self.od:
            return self.insert(key)
        res = -1 if self.od[key] else -2
        del self.od[key]
        self.od[key] = True
        return res

    def insert(self, key):                                                   # :90-107
        self.od[key] = True
        if len(self.od) > self.cap:
            _, flag = self.od.popitem(last=False)
            return 1 if flag else 0
        return 0

    def outdate(self, key):                                                  # :120-128
        key = _as_int(key)
        if key in self.od:
            self.od[key] = False

    def keys(self):                                                          # :130-139 (valid keys, sorted)
        return sorted(k for k, v in self.od.items() if v)


class LaiaSchedulerModel:
    """LaiaScheduler (laia/src/laia_scheduler.cc).  emit() returns what the queue would hold for `rank`:
    [plan_0, dist_0, plan_1, dist_1, ..., [0]]."""

    def __init__(self, samples, epoch_num, mini_batch_size, batch_num, nrank, rank, cache_size):
        self.samples = np.asarray(samples, dtype=np.uint64)
        self.S, self.T = self.samples.shape
        self.epoch_num, self.mini_bs, self.batch_num = epoch_num, mini_batch_size, batch_num
        self.W, self.rank = nrank, rank
        self.B = mini_batch_size * nrank                                     # :47
        self.snaps = [MiniLRU(cache_size) for _ in range(nrank)]

    def get_dist(self, batch_id):                                            # :171-271
        B, W, T, S = self.B, self.W, self.T, self.S
        start = (batch_id * B) % S
        batch = [self.samples[(start + i) % S] for i in range(B)]
        scores = np.zeros((B, W), dtype=np.int64)
        dep = [[[] for _ in range(W)] for _ in range(B)]
        for i in range(B):                                                   # score :194-223
            for j in range(T):
                emb = int(batch[i][j])
                for z in range(W):
                    if self.snaps[z].check(emb):
                        scores[i][z] += 1
                        dep[i][z].append(emb)
        workload = [0] * W
        dist = [[0] * self.mini_bs for _ in range(W)]
        assigned = [set() for _ in range(W)]
        owner = [0] * B
        for i in range(B):                                                   # assign :231-249
            max_score, max_worker = -1, -1
            for j in range(W):
                w = (j + batch_id) % W
                if workload[w] < self.mini_bs and max_score < scores[i][w]:
                    max_score, max_worker = int(scores[i][w]), w
            pos = (i + start) % S
            dist[max_worker][workload[max_worker]] = pos
            assigned[max_worker].add(pos)
            workload[max_worker] += 1
            owner[i] = max_worker
        cplan = []
        for w in range(W):                                                   # plan :252-270
            plan = set()
            for s in range(B):
                pos = (s + start) % S
                if pos not in assigned[w]:
                    plan.update(dep[s][w])
            cplan.append(sorted(plan))
        return cplan, dist

    def emit(self):                                                          # launch :115-169
        out = []
        epoch_id = 0
        batch_num = self.batch_num
        while epoch_id < self.epoch_num:
            batch_id = 0
            epoch_id += 1
            if epoch_id == self.epoch_num:
                batch_num += 1
            while batch_id < batch_num:
                cplan, dist = self.get_dist(batch_id)
                out.append([int(k) for k in cplan[self.rank]])
                out.append([int(p) for p in dist[self.rank]])
                for w in range(self.W):                                      # snapshot update :146-162
                    for key in cplan[w]:
                        self.snaps[w].outdate(key)
                    uk = set()
                    for p in dist[w]:
                        uk.update(int(e) for e in self.samples[p])
                    for key in sorted(uk):
                        self.snaps[w].get(key)
                batch_id += 1
        out.append([0])
        return out


# ---- TopkScheduler (laia/src/topk_scheduler.cc) -----------------------------------------------------
# pre-profiled table orders, topk_scheduler.cc:151-165
TOPK_TABLE_ORDER = {
    "criteo": [9, 13, 22, 20, 12, 21, 17, 14, 24, 3, 5, 10, 16, 15, 19, 2, 4, 11, 7, 25, 23, 18, 8, 1, 0, 6],
    "avazu": [1, 2, 4, 5, 15, 7, 6, 16, 12, 0, 17, 8, 14, 10, 9, 11, 13, 3],
    "movie": [0, 1],
    "criteosearch": [0, 11, 3, 4, 5, 14, 1, 6, 2, 13, 16, 9, 8, 10, 12, 7, 15],
}


def topk_thread_slices(total, num_threads):
    """[start, end) of every pool thread over `total` items: thread 0 takes the remainder
    (topk_scheduler.cc:398-407, used for both the samples and the per-worker quota)."""
    x, y = divmod(total, num_threads)
    out = []
    for t in range(num_threads):
        start = 0 if t == 0 else y + t * x
        end = x + y if t == 0 else start + x
        out.append((start, end))
    return out


class TopkSchedulerModel:
    """TopkScheduler::get_dist / launch (topk_scheduler.cc:362-502, 277-354), standalone queue mode.

    Differences from LaiaScheduler: only the first `top_k_table` tables of the dataset's fixed order are
    scored (:411-429); the batch and every worker's quota are cut into per-thread slices and each
    thread assigns its samples independently (:393-455); a sample is offered to the workers in the
    order (j + candidate) % W where candidate is the worker that first reached the sample's top score
    (:421-443); plan[w] = the keys (all tables) of w's OWN samples that w's snapshot holds valid
    (:468-500).  The shipped loop erases from the flat_set it is range-iterating (:485-489, undefined
    behaviour); this is the intended filter.  A thread whose samples outnumber W x its quota writes
    dist[-1] in the reference (no worker passes the capacity test, :436-443): such parameters are
    rejected here.  Parity unpinned: nothing of the reference asserts on this scheduler and it does
    not build in this image (boost::interprocess)."""

    def __init__(self, samples, epoch_num, mini_batch_size, batch_num, nrank, rank, cache_size,
                 num_threads, dataset, top_k_table):
        self.samples = np.asarray(samples, dtype=np.uint64)
        self.S, self.T = self.samples.shape
        self.epoch_num, self.mini_bs, self.batch_num = epoch_num, mini_batch_size, batch_num
        self.W, self.rank = nrank, rank
        self.B = mini_batch_size * nrank
        self.nt = num_threads
        order = TOPK_TABLE_ORDER[dataset]
        k = top_k_table if top_k_table else self.T                            # :131-133
        self.order = order[:min(k, len(order))]                              # :167-168
        self.snaps = [MiniLRU(cache_size) for _ in range(nrank)]
        self.miss_pull = [0] * nrank
        self.miss_push = [0] * nrank
        self.update_pull = [0] * nrank
        self.update_push = [0] * nrank
        for (s0, s1), (q0, q1) in zip(topk_thread_slices(self.B, self.nt),
                                      topk_thread_slices(self.mini_bs, self.nt)):
            if s1 - s0 > self.W * (q1 - q0):
                raise ValueError("thread slice of %d samples exceeds %d workers x quota %d" %
                                 (s1 - s0, self.W, q1 - q0))

    def get_dist(self, batch_id):
        B, W, S = self.B, self.W, self.S
        start = (batch_id * B) % S                                           # :371
        batch = [self.samples[(start + i) % S] for i in range(B)]
        dist = [[0] * self.mini_bs for _ in range(W)]                        # dist.reset(0), :382
        owner = [0] * B
        sl = topk_thread_slices(B, self.nt)
        ql = topk_thread_slices(self.mini_bs, self.nt)
        for t in range(self.nt):
            (s0, s1), (wstart, wend) = sl[t], ql[t]
            quota = wend - wstart
            workload = [0] * W
            for i in range(s0, s1):
                scores = [0] * W
                top, cand = 0, 0                                             # max_score / mscore_worker, :377-378
                for j in self.order:                                         # :411-429
                    emb = int(batch[i][j])
                    for z in range(W):
                        if self.snaps[z].check(emb):
                            scores[z] += 1
                            if scores[z] > top:
                                top, cand = scores[z], z
                best, best_w = -1, -1                                        # :430-443
                for j in range(W):
                    w = (j + cand) % W
                    if best < scores[w] and workload[w] < quota:
                        best, best_w = scores[w], w
                        if best_w == cand:
                            break
                dist[best_w][wstart + workload[best_w]] = (i + start) % S    # :448-450
                workload[best_w] += 1
                owner[i] = best_w
        cplan = []
        for w in range(W):                                                   # :468-500
            mine = set(dist[w])
            keys = set()
            for s in range(B):
                if (s + start) % S in mine:
                    keys.update(int(e) for e in batch[s])
            cplan.append(sorted(k for k in keys if self.snaps[w].check(k)))
        return cplan, dist

    def emit(self, ranks=None):
        """Streams for the given ranks (default: [self.rank]): {rank: [plan, dist, ..., [0]]}."""
        ranks = [self.rank] if ranks is None else list(ranks)
        out = {r: [] for r in ranks}
        epoch_id = 0
        batch_num = self.batch_num
        while epoch_id < self.epoch_num:
            batch_id = 0
            epoch_id += 1
            if epoch_id == self.epoch_num:
                batch_num += 1                                               # :293-295
            while batch_id < batch_num:
                cplan, dist = self.get_dist(batch_id)
                for r in ranks:                                              # :304-318
                    out[r].append([int(k) for k in cplan[r]])
                    out[r].append([int(p) for p in dist[r]])
                for w in range(self.W):                                      # :325-345
                    for key in cplan[w]:
                        self.snaps[w].outdate(key)
                    uk = set()
                    for p in dist[w]:
                        uk.update(int(e) for e in self.samples[p])
                    for key in sorted(uk):
                        res = self.snaps[w].get(key)
                        if res < 0:
                            if res == -2:
                                self.update_pull[w] += 1
                        else:
                            self.miss_pull[w] += 1
                            if res > 0:
                                self.miss_push[w] += 1
                    self.update_push[w] += len(cplan[w])
                batch_id += 1
        for r in ranks:
            out[r].append([0])
        return out
