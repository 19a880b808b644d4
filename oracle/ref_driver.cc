// Driver that exposes pieces of the REFERENCE's own code through a C ABI, for pinning the oracle.
// TEST INFRASTRUCTURE ONLY.  This file is ours; everything it drives is #included / compiled from
// /root/reference where it lies (see oracle/build_ref.sh).  No reference source is copied and no
// stand-in header or library is provided: the four ps:: functions that src/hetu_cache/src/
// hetu_client.cc calls stay UNRESOLVED in the shared object (lazy PLT binding); they are only reached
// from CacheBase::_embedding* which this driver never calls.
//
//   ref_unique_*   hetu::Unique<T>            src/hetu_cache/include/unqiue_tools.h:27-48
//   ref_policy_*   hetu::LRUCache / LFUCache / LFUOptCache   src/hetu_cache/src/{lru,lfu,lfuopt}_cache.cc
//   ref_minilru_*  laia_cache::MiniLRUCache   laia/include/mini_lru_cache.h:14-137
#include <cstdint>
#include <cstring>
#include <vector>

#include "unqiue_tools.h"
#include "lru_cache.h"
#include "lfu_cache.h"
#include "lfuopt_cache.h"
#include "mini_lru_cache.h"

namespace {

template <class Base>
struct Probe : Base {
    using Base::Base;
    std::vector<hetu::EmbeddingPT> &evicted() { return this->evict_; }
};

struct Policy {
    int kind;
    size_t width;
    Probe<hetu::LRUCache> *lru = nullptr;
    Probe<hetu::LFUCache> *lfu = nullptr;
    Probe<hetu::LFUOptCache> *opt = nullptr;
    hetu::CacheBase *base() {
        return kind == 0 ? static_cast<hetu::CacheBase *>(lru)
                         : kind == 1 ? static_cast<hetu::CacheBase *>(lfu) : static_cast<hetu::CacheBase *>(opt);
    }
    std::vector<hetu::EmbeddingPT> &evicted() {
        return kind == 0 ? lru->evicted() : kind == 1 ? lfu->evicted() : opt->evicted();
    }
};

}  // namespace

extern "C" {

// ---- Unique<uint64_t> ---------------------------------------------------------------------------
size_t ref_unique_u64(const uint64_t *keys, size_t n, uint64_t *uniq, int64_t *inverse) {
    hetu::Unique<uint64_t> u(keys, n);
    for (size_t i = 0; i < u.size(); ++i)
        uniq[i] = u[i];
    for (size_t i = 0; i < n; ++i)
        inverse[i] = static_cast<int64_t>(u.map(i));
    return u.size();
}

// ---- cache policies -------------------------------------------------------------------------------
void *ref_policy_new(int kind, size_t limit, size_t width) {
    Policy *p = new Policy();
    p->kind = kind;
    p->width = width;
    if (kind == 0)
        p->lru = new Probe<hetu::LRUCache>(limit, 0, width, 0);
    else if (kind == 1)
        p->lfu = new Probe<hetu::LFUCache>(limit, 0, width, 0);
    else
        p->opt = new Probe<hetu::LFUOptCache>(limit, 0, width, 0);
    return p;
}
void ref_policy_free(void *h) {
    Policy *p = static_cast<Policy *>(h);
    delete p->lru;
    delete p->lfu;
    delete p->opt;
    delete p;
}
// insert a fresh line for `key` whose `updates` counter is `updates` (accumulate() called that often)
void ref_policy_insert(void *h, uint64_t key, int updates) {
    Policy *p = static_cast<Policy *>(h);
    auto e = std::make_shared<hetu::Embedding>(key, p->width);
    std::vector<float> g(p->width, 0.f);
    for (int i = 0; i < updates; ++i)
        e->accumulate(g.data());
    p->base()->insert(e);
}
// returns 1 on hit (and performs the policy's touch), 0 on miss
int ref_policy_lookup(void *h, uint64_t key) {
    Policy *p = static_cast<Policy *>(h);
    return p->base()->lookup(key) ? 1 : 0;
}
int ref_policy_count(void *h, uint64_t key) { return static_cast<Policy *>(h)->base()->count(key); }
size_t ref_policy_size(void *h) { return static_cast<Policy *>(h)->base()->size(); }
// keys of the lines queued in evict_ (dirty evictions), in queue order; clears the queue
size_t ref_policy_take_evicted(void *h, uint64_t *out, size_t cap) {
    auto &ev = static_cast<Policy *>(h)->evicted();
    size_t n = 0;
    for (auto &e : ev)
        if (n < cap)
            out[n++] = e->key();
    ev.clear();
    return n;
}

// ---- MiniLRUCache -----------------------------------------------------------------------------------
void *ref_minilru_new(int capacity) {
    auto *c = new laia_cache::MiniLRUCache();
    c->set_cap(capacity);
    return c;
}
void ref_minilru_free(void *h) { delete static_cast<laia_cache::MiniLRUCache *>(h); }
int ref_minilru_check(void *h, int key) { return static_cast<laia_cache::MiniLRUCache *>(h)->check(key) ? 1 : 0; }
int ref_minilru_get(void *h, int key) { return static_cast<laia_cache::MiniLRUCache *>(h)->get(key); }
void ref_minilru_outdate(void *h, int key) { static_cast<laia_cache::MiniLRUCache *>(h)->outdate(key); }
size_t ref_minilru_keys(void *h, int *out, size_t cap) {
    auto keys = static_cast<laia_cache::MiniLRUCache *>(h)->get_keys();
    size_t n = 0;
    for (int k : keys)
        if (n < cap)
            out[n++] = k;
    return keys.size();
}

}  // extern "C"
