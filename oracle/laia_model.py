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
