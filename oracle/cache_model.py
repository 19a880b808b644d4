"""CPU restatement of the HET embedding cache (reference src/hetu_cache) and of the PS handlers it
talks to.  TEST INFRASTRUCTURE ONLY -- never imported by herald_amd.

Pure Python / numpy on purpose: this is bookkeeping over a few hundred keys per step; every
function cites the reference code it follows (paths relative to /root/reference).

Parity pin: the full cache flow needs ps-lite's worker/server (ZeroMQ + protobuf, absent here), so it
cannot be built from the reference without stand-ins: parity of the FLOWS is therefore unpinned
("parity unpinned").  The eviction POLICIES are pinned against the reference's own
LRUCache/LFUCache/LFUOptCache objects compiled from /root/reference where they lie
(oracle/build_ref.sh -> oracle/_ref/hetu_cache*.so, tests/golden/cache_policy_*.json).
"""
from collections import OrderedDict

import numpy as np


class Line:
    """Line<T>, src/hetu_cache/include/embedding.h:18-149."""

    def __init__(self, key, width, init_data=True):
        self.key = int(key)
        self.width = width
        self.data = np.zeros(width, dtype=np.float32) if init_data else None   # :37-43
        self.grad = None                                                       # lazily created :131-134
        self.updates = 0
        self.version = -1

    def accumulate(self, g):                                                   # :78-91
        if self.grad is None:
            self.grad = np.zeros(self.width, dtype=np.float32)
        self.grad = (self.grad + g).astype(np.float32)
        if self.data is not None:
            self.data = (self.data + g).astype(np.float32)
        self.updates += 1

    def addup(self):                                                           # :92-96
        if self.grad is not None:
            self.data = (self.data + self.grad).astype(np.float32)

    def zero_grad(self):                                                       # :112-118
        self.grad = np.zeros(self.width, dtype=np.float32)
        self.updates = 0


class Server:
    """CacheTable = Param2D + version_t ver[len] (ps-lite/include/ps/server/param.h:119-138) with the
    handlers of ps-lite/src/PSFhandle_embedding.cc."""

    def __init__(self, table):
        self.table = np.array(table, dtype=np.float32, copy=True)
        self.ver = np.zeros(self.table.shape[0], dtype=np.int64)

    def push_embedding(self, lines):                                           # :5-28
        for ln in lines:
            self.ver[ln.key] += ln.updates
            self.table[ln.key] = (self.table[ln.key] + ln.grad).astype(np.float32)

    def sync_embedding(self, lines, bound):                                    # :30-64 + hetu_client.cc:6-39
        pulled = 0
        for ln in lines:
            if ln.version == -1 or self.ver[ln.key] - ln.version > bound:
                ln.version = int(self.ver[ln.key])
                ln.data = self.table[ln.key].copy()
                ln.addup()
                pulled += 1
        return pulled


class LRUPolicy:
    """LRUCache, src/hetu_cache/src/lru_cache.cc:5-39 (list front = most recent)."""

    def __init__(self, limit):
        self.limit = limit
        self.od = OrderedDict()            # last item = list front

    def size(self):
        return len(self.od)

    def count(self, k):
        return 1 if k in self.od else 0

    def lookup(self, k):                                                       # :27-39
        ln = self.od.get(k)
        if ln is None:
            return None
        self.od.move_to_end(k)
        return ln

    def insert(self, ln, evict_out):                                           # :9-25
        if ln.key in self.od:
            del self.od[ln.key]
        self.od[ln.key] = ln
        if len(self.od) > self.limit:
            k, victim = self.od.popitem(last=False)
            if victim.updates != 0:
                evict_out.append(victim)

    def keys(self):
        return sorted(self.od.keys())


class LFUPolicy:
    """LFUCache, src/hetu_cache/src/lfu_cache.cc:9-70: frequency buckets, each a list with the most
    recently arrived line at the front; evict the back of the lowest bucket when size == limit."""

    def __init__(self, limit):
        self.limit = limit
        self.freq = {}                     # key -> use count
        self.lines = {}
        self.buckets = {}                  # use -> OrderedDict (last = front)

    def size(self):
        return len(self.lines)

    def count(self, k):
        return 1 if k in self.lines else 0

    def _put(self, k, use):
        self.buckets.setdefault(use, OrderedDict())[k] = True
        self.freq[k] = use

    def _increase(self, k):                                                    # :51-68
        use = self.freq[k]
        del self.buckets[use][k]
        if not self.buckets[use]:
            del self.buckets[use]
        self._put(k, use + 1)

    def _evict(self, evict_out):                                               # :31-42
        use = min(self.buckets)
        k, _ = self.buckets[use].popitem(last=False)
        if not self.buckets[use]:
            del self.buckets[use]
        victim = self.lines.pop(k)
        del self.freq[k]
        if victim.updates != 0:
            evict_out.append(victim)

    def lookup(self, k):                                                       # :22-29
        if k not in self.lines:
            return None
        self._increase(k)
        return self.lines[k]

    def insert(self, ln, evict_out):                                           # :9-20
        if ln.key not in self.lines:
            if len(self.lines) == self.limit:
                self._evict(evict_out)
            self.lines[ln.key] = ln
            self._put(ln.key, 1)
        else:
            self.lines[ln.key] = ln
            self._increase(ln.key)

    def keys(self):
        return sorted(self.lines.keys())


class LFUOptPolicy:
    """LFUOptCache, src/hetu_cache/src/lfuopt_cache.cc:9-71: ten use buckets 0..9; a line reaching use 9
    moves to a never-evicted store; when only the store is left new lines are dropped."""
    K = 10

    def __init__(self, limit):
        self.limit = limit
        self.store = {}
        self.lines = {}
        self.use = {}
        self.clist = [OrderedDict() for _ in range(self.K)]   # last = front

    def size(self):
        return len(self.store) + len(self.lines)

    def count(self, k):
        return (1 if k in self.store else 0) + (1 if k in self.lines else 0)

    def lookup(self, k):                                                       # :26-41
        if k in self.store:
            return self.store[k]
        if k not in self.lines:
            return None
        ln = self.lines[k]
        u = self.use[k]
        if u + 1 < self.K:
            del self.clist[u][k]
            self.clist[u + 1][k] = True
            self.use[k] = u + 1
        else:
            self.store[k] = ln
            del self.clist[self.K - 1][k]
            del self.lines[k]
            del self.use[k]
        return ln

    def _evict(self, evict_out):                                               # :48-60
        for i in range(self.K):
            if self.clist[i]:
                k, _ = self.clist[i].popitem(last=False)
                victim = self.lines.pop(k)
                del self.use[k]
                if victim.updates:
                    evict_out.append(victim)
                break

    def insert(self, ln, evict_out):                                           # :9-24
        if ln.key in self.store:
            self.store[ln.key] = ln
            return
        if ln.key in self.lines:
            self.lines[ln.key] = ln
            return
        if self.size() == self.limit:
            if self.lines:
                self._evict(evict_out)
            else:
                return
        self.lines[ln.key] = ln
        self.use[ln.key] = 0
        self.clist[0][ln.key] = True

    def keys(self):
        return sorted(list(self.store.keys()) + list(self.lines.keys()))


def unique_sorted(keys):
    """hetu::Unique<T> (include/unqiue_tools.h:27-48): sorted unique + map."""
    u, inv = np.unique(np.asarray(keys, dtype=np.uint64), return_inverse=True)
    return u, inv


class CacheModel:
    """CacheBase flows, src/hetu_cache/src/cache.cc."""

    def __init__(self, policy, limit, width, server, pull_bound=5, push_bound=5):
        self.policy = {"lru": LRUPolicy, "lfu": LFUPolicy, "lfuopt": LFUOptPolicy}[policy.lower()](limit)
        self.limit, self.width, self.server = limit, width, server
        self.pull_bound, self.push_bound = pull_bound, push_bound
        self.evict = []
        self.bypass = False
        self.perf = []

    def _batched_lookup(self, ukeys):                                          # :15-26
        if self.bypass:
            return [None] * len(ukeys)
        return [self.policy.lookup(int(k)) for k in ukeys]

    def _batched_insert(self, lines):                                          # :28-35
        if self.bypass:
            return
        for ln in lines:
            self.policy.insert(ln, self.evict)

    def lookup(self, keys, width=None):                                        # _embeddingLookup :60-107
        keys = np.asarray(keys, dtype=np.uint64).reshape(-1)
        ukeys, inv = unique_sorted(keys)
        embeds = self._batched_lookup(ukeys)
        should_insert = []
        for i, k in enumerate(ukeys):
            if embeds[i] is None:
                embeds[i] = Line(k, self.width)
                should_insert.append(embeds[i])
        pulled = self.server.sync_embedding(embeds, self.pull_bound)
        dest = np.empty((keys.size, self.width), dtype=np.float32)
        for j in range(keys.size):
            dest[j] = embeds[inv[j]].data
        self._batched_insert(should_insert)
        self.perf.append({"type": "Pull", "is_full": self.policy.size() == self.limit, "num_all": keys.size,
                          "num_unique": len(ukeys), "num_miss": len(should_insert), "num_transfered": pulled})
        return dest

    def _accumulate(self, keys, grads):
        ukeys, inv = unique_sorted(keys)
        embeds = self._batched_lookup(ukeys)
        miss = 0
        evict = self.evict
        self.evict = []
        for j in range(keys.size):
            i = inv[j]
            if embeds[i] is None:
                embeds[i] = Line(ukeys[i], self.width, init_data=False)
                miss += 1
            embeds[i].accumulate(grads[j])
        return ukeys, embeds, miss, evict

    def update(self, keys, grads):                                             # _embeddingUpdate :132-197
        keys = np.asarray(keys, dtype=np.uint64).reshape(-1)
        grads = np.asarray(grads, dtype=np.float32).reshape(keys.size, self.width)
        ukeys, embeds, miss, evict = self._accumulate(keys, grads)
        should_push = []
        it = 0
        for i in range(len(ukeys)):
            if embeds[i].updates > self.push_bound or embeds[i].data is None:
                while it < len(evict) and evict[it].key < embeds[i].key:
                    should_push.append(evict[it])
                    it += 1
                should_push.append(embeds[i])
        should_push.extend(evict[it:])
        self.server.push_embedding(should_push)
        for i in range(len(ukeys)):
            if embeds[i].updates > self.push_bound and embeds[i].data is not None:
                embeds[i].version += embeds[i].updates
                embeds[i].zero_grad()
        self.perf.append({"type": "Push", "is_full": self.policy.size() == self.limit, "num_all": keys.size,
                          "num_unique": len(ukeys), "num_evict": len(evict), "num_miss": miss,
                          "num_transfered": len(should_push)})

    def update_with_push_keys(self, keys, push_keys, grads):                   # :248-335
        keys = np.asarray(keys, dtype=np.uint64).reshape(-1)
        push_keys = [int(k) for k in np.asarray(push_keys, dtype=np.uint64).reshape(-1)]
        grads = np.asarray(grads, dtype=np.float32).reshape(keys.size, self.width)
        ukeys, embeds, miss, evict = self._accumulate(keys, grads)
        should_push, idxs = [], []
        it, pit = 0, 0
        for i in range(len(ukeys)):
            while it < len(evict) and evict[it].key < embeds[i].key:
                should_push.append(evict[it])
                it += 1
            while pit < len(push_keys) and push_keys[pit] < embeds[i].key:
                pit += 1
            if pit < len(push_keys) and push_keys[pit] == embeds[i].key and embeds[i].data is not None:
                should_push.append(embeds[i])
                idxs.append(i)
        should_push.extend(evict[it:])
        self.server.push_embedding(should_push)
        j = 0
        for i in range(len(ukeys)):
            embeds[i].version += embeds[i].updates
            if j < len(idxs) and idxs[j] == i:
                embeds[i].zero_grad()
                j += 1
        self.perf.append({"type": "Push", "is_full": self.policy.size() == self.limit, "num_all": keys.size,
                          "num_unique": len(ukeys), "num_evict": len(evict), "num_miss": miss,
                          "num_transfered": len(should_push)})

    def push_pull(self, pull_keys, push_keys, grads):                          # _embeddingPushPull :356-422
        self.push_pull_begin(pull_keys, push_keys, grads)
        return self.push_pull_finish()

    def push_pull_begin(self, pull_keys, push_keys, grads):
        """Everything up to and including the server's push half of kPushSyncEmbedding.  (Split so that
        the tests can interleave several workers: all pushes of a round reach the server before any sync.)"""
        pull_keys = np.asarray(pull_keys, dtype=np.uint64).reshape(-1)
        push_keys = np.asarray(push_keys, dtype=np.uint64).reshape(-1)
        grads = np.asarray(grads, dtype=np.float32).reshape(push_keys.size, self.width)
        ukeys, inv = unique_sorted(pull_keys)
        embeds = self._batched_lookup(ukeys)
        should_insert = []
        for i, k in enumerate(ukeys):
            if embeds[i] is None:
                embeds[i] = Line(k, self.width)
                should_insert.append(embeds[i])
        pkeys, pembeds, miss, evict = self._accumulate(push_keys, grads)
        should_push = []
        it = 0
        for i in range(len(pkeys)):
            if pembeds[i].updates > self.push_bound or pembeds[i].data is None:
                while it < len(evict) and evict[it].key < pembeds[i].key:
                    should_push.append(evict[it])
                    it += 1
                should_push.append(pembeds[i])
        should_push.extend(evict[it:])
        self.server.push_embedding(should_push)                                 # server: push, then sync
        self._pp = (pull_keys, inv, embeds, should_insert, pkeys, pembeds)

    def push_pull_finish(self):
        pull_keys, inv, embeds, should_insert, pkeys, pembeds = self._pp
        self._pp = None
        self.server.sync_embedding(embeds, self.pull_bound)                     # (PSFhandle_embedding.cc:66-79)
        dest = np.empty((pull_keys.size, self.width), dtype=np.float32)
        for j in range(pull_keys.size):
            dest[j] = embeds[inv[j]].data
        self._batched_insert(should_insert)
        for i in range(len(pkeys)):
            if pembeds[i].updates > self.push_bound and pembeds[i].data is not None:
                pembeds[i].version += pembeds[i].updates
                pembeds[i].zero_grad()
        return dest

    # -- inspection helpers for the tests
    def resident(self):
        pol = self.policy
        if isinstance(pol, LRUPolicy):
            return dict(pol.od)
        if isinstance(pol, LFUPolicy):
            return dict(pol.lines)
        d = dict(pol.lines)
        d.update(pol.store)
        return d
