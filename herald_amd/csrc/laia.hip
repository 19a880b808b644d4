// laia embedding scheduler (reference laia/): per global batch, score every
// sample against per-worker cache snapshots, assign samples to workers, and emit each worker's
// communication plan (rows it holds that another worker will touch).
//
// Reference: LaiaScheduler::get_dist / launch (laia/src/laia_scheduler.cc:115-271) with
// MiniLRUCache snapshots (laia/include/mini_lru_cache.h:14-137).
//
// Split between the GPU and the host thread that drives it:
//   GPU  * probing: for every (sample, table) of the batch, the set of workers whose snapshot holds
//          the row VALID (one byte per (worker,row) in HBM, B*T*W probes per batch) -> per-sample
//          scores and per-(sample,table) worker masks;
//        * plan extraction: rows valid at w that occur in samples NOT assigned to w, and each
//          worker's touched rows, as composite keys w*R+row, sorted-unique by the index-plan sort;
//        * applying the snapshot deltas to the validity bytes.
//   host * the greedy capacity-bounded assignment (sequential by definition, laia_scheduler.cc:231-249);
//        * the exact MiniLRUCache bookkeeping (get / outdate in sorted-key order with interleaved
//          evictions, mini_lru_cache.h:69-128) -- a pointer-chasing recurrence with no parallel form
//          that keeps the reference's eviction order.
// The emitted (plan, dist) sequence is bit-identical to the reference's.
#include "plan_dev.h"

#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace ha {

// ---- host-side snapshot: the semantics of MiniLRUCache (hash mode) ---------------------------------
// The reference keeps a linked list in use order (mini_lru_cache.h: splice to the front on every get, evict
// from the back).  Every touch of a list costs four to five dependent random host-memory accesses (the
// node, both neighbours, the head).  Here -- like the GPU cache's LRU (cache.hip) -- a touch stamps the node with a
// monotone counter and appends (node, stamp) to a ring log; the list order IS the stamp order, an entry is
// stale once its node was touched again or left the snapshot, and the eviction victim is the first live
// entry from the log's head.  One random access per touch (the node) besides the key map, the log is
// sequential; each entry is looked at once more when the head passes it.
struct alignas(128) Snapshot {   // one worker thread per snapshot: keep their hot fields in different cache lines
    struct Node {
        uint32_t stamp;   // 0 = not resident
        int32_t key;
        uint8_t valid;
    };
    struct Entry {
        int32_t node;
        uint32_t stamp;
    };
    int cap = 0;
    // key -> node: a direct map over the row range where that fits in memory (one load per probe, like the
    // device-side validity bytes), a hash map for very large key spaces
    bool direct = false;
    int32_t *dmap = nullptr;      // mmap'ed, transparent huge pages requested: 135 MB of random probes per worker
    size_t dmap_len = 0;
    std::unordered_map<int32_t, int> hmap;
    size_t live = 0;
    std::vector<Node> node;
    std::vector<int> free_nodes;
    std::vector<Entry> log;       // ring
    size_t log_head = 0, log_size = 0;
    uint32_t counter = 0;

    void init(int capacity, long long key_range, bool use_direct) {
        cap = capacity;
        const int n = capacity + 2;
        node.assign(n, Node{0u, 0, 0});
        free_nodes.clear();
        for (int i = n - 1; i >= 0; --i)
            free_nodes.push_back(i);
        size_t lg = 1;
        while (lg < static_cast<size_t>(n) * 4)
            lg <<= 1;
        log.assign(lg, Entry{0, 0u});
        log_head = log_size = 0;
        counter = 0;
        direct = use_direct;
        hmap.clear();
        release();
        if (direct) {
            dmap_len = static_cast<size_t>(key_range);
            const size_t bytes = (dmap_len * sizeof(int32_t) + (2u << 20) - 1) & ~((size_t(2) << 20) - 1);
            void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) {
                direct = false;     // fall back to the hash map
                dmap_len = 0;
            } else {
                madvise(m, bytes, MADV_HUGEPAGE);
                dmap = static_cast<int32_t *>(m);
                memset(dmap, 0xFF, dmap_len * sizeof(int32_t));   // -1 = absent
            }
        }
        if (!direct)
            hmap.reserve(static_cast<size_t>(capacity) * 3);  // set_cap, mini_lru_cache.h:49-52
        live = 0;
    }
    void release() {
        if (dmap != nullptr) {
            munmap(dmap, (dmap_len * sizeof(int32_t) + (2u << 20) - 1) & ~((size_t(2) << 20) - 1));
            dmap = nullptr;
            dmap_len = 0;
        }
    }
    Snapshot() = default;
    Snapshot(const Snapshot &) = delete;
    Snapshot &operator=(const Snapshot &) = delete;
    Snapshot(Snapshot &&o) noexcept { *this = std::move(o); }
    Snapshot &operator=(Snapshot &&o) noexcept {
        if (this != &o) {
            release();
            cap = o.cap; direct = o.direct; dmap = o.dmap; dmap_len = o.dmap_len; hmap = std::move(o.hmap);
            live = o.live; node = std::move(o.node); free_nodes = std::move(o.free_nodes); log = std::move(o.log);
            log_head = o.log_head; log_size = o.log_size; counter = o.counter;
            o.dmap = nullptr; o.dmap_len = 0;
        }
        return *this;
    }
    ~Snapshot() { release(); }
    int find(int32_t k) const {
        if (direct)
            return dmap[static_cast<size_t>(k)];
        auto it = hmap.find(k);
        return it == hmap.end() ? -1 : it->second;
    }
    void bind(int32_t k, int x) {
        if (direct)
            dmap[static_cast<size_t>(k)] = x;
        else
            hmap[k] = x;
        ++live;
    }
    void unbind(int32_t k) {
        if (direct)
            dmap[static_cast<size_t>(k)] = -1;
        else
            hmap.erase(k);
        --live;
    }
    // validity of key k as the device mirror must show it
    uint8_t state(int32_t k) const {
        const int x = find(k);
        return x >= 0 && node[x].valid ? 1 : 0;
    }
    // drop the stale entries (in place, order kept) when the ring is full
    void compact() {
        const size_t mask = log.size() - 1;
        size_t w = 0;
        for (size_t i = 0; i < log_size; ++i) {
            const Entry e = log[(log_head + i) & mask];
            if (node[e.node].stamp == e.stamp) {
                log[(log_head + w) & mask] = e;
                ++w;
            }
        }
        log_size = w;
    }
    void renumber() {   // the 32-bit counter wrapped: restamp the live entries 1, 2, ... in order
        compact();
        const size_t mask = log.size() - 1;
        for (size_t i = 0; i < log_size; ++i) {
            Entry &e = log[(log_head + i) & mask];
            e.stamp = static_cast<uint32_t>(i + 1);
            node[e.node].stamp = e.stamp;
        }
        counter = static_cast<uint32_t>(log_size);
    }
    void touch(int x) {   // move to the front of the use order
        if (counter == 0xFFFFFFFFu)
            renumber();
        if (log_size == log.size())
            compact();      // live entries <= cap + 1 < log.size() / 4: always room afterwards
        node[x].stamp = ++counter;
        log[(log_head + log_size) & (log.size() - 1)] = Entry{x, counter};
        ++log_size;
    }
    int pop_back() {      // the least recently used resident node
        const size_t mask = log.size() - 1;
        for (;;) {
            const Entry e = log[log_head];
            log_head = (log_head + 1) & mask;
            --log_size;
            if (node[e.node].stamp == e.stamp)
                return e.node;
        }
    }
    // touched: keys whose validity byte may have changed (the caller mirrors state(k) on the device;
    // a key listed twice is written twice with the same final value)
    void outdate(int32_t k, std::vector<int32_t> &touched) {
        const int x = find(k);
        if (x >= 0 && node[x].valid) {
            node[x].valid = 0;
            touched.push_back(k);
        }
    }
    int get(int32_t k, std::vector<int32_t> &touched) {
        int x = find(k);
        if (x >= 0) {
            const int res = node[x].valid ? -1 : -2;
            touch(x);
            if (!node[x].valid) {
                node[x].valid = 1;
                touched.push_back(k);
            }
            return res;
        }
        x = free_nodes.back();
        free_nodes.pop_back();
        node[x].key = k;
        node[x].valid = 1;
        touch(x);
        bind(k, x);
        touched.push_back(k);
        if (static_cast<long long>(live) > cap) {
            const int e = pop_back();
            const bool flag = node[e].valid != 0;
            node[e].stamp = 0;
            unbind(node[e].key);
            if (flag)
                touched.push_back(node[e].key);
            free_nodes.push_back(e);
            return flag ? 1 : 0;
        }
        return 0;
    }
    // valid resident keys (any order)
    void valid_keys(std::vector<int32_t> &out) const {
        for (const Node &nd : node)
            if (nd.stamp != 0 && nd.valid)
                out.push_back(nd.key);
    }
};

// ---- kernels -------------------------------------------------------------------------------------------
// mask[i*T+j] = set of workers whose snapshot holds row samples[(start+i)%S][j] valid
__global__ __launch_bounds__(256) void laia_probe_kernel(
    const uint32_t *__restrict__ samples, long long S, int T, long long start, int B, int W,
    const uint8_t *__restrict__ valid, long long R, unsigned long long *__restrict__ mask) {
    const long long total = static_cast<long long>(B) * T;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        const long long i = e / T;
        const int j = static_cast<int>(e - i * T);
        const uint32_t emb = samples[((start + i) % S) * T + j];
        unsigned long long m = 0;
        if (emb < R) {
            for (int w = 0; w < W; ++w)
                if (valid[static_cast<long long>(w) * R + emb])
                    m |= 1ull << w;
        }
        mask[e] = m;
    }
}

// scores[i*W+w] = number of tables j with bit w set
__global__ __launch_bounds__(256) void laia_score_kernel(
    const unsigned long long *__restrict__ mask, int B, int T, int W, int32_t *__restrict__ scores) {
    const int total = B * W;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e / W, w = e - i * W;
        int s = 0;
        for (int j = 0; j < T; ++j)
            s += static_cast<int>((mask[static_cast<long long>(i) * T + j] >> w) & 1ull);
        scores[e] = s;
    }
}

// plan pairs: (w, row) for rows valid at w in samples not assigned to w (LaiaScheduler), or -- own_plan,
// TopkScheduler -- (owner, row) for rows valid at the sample's own worker; touched pairs: (owner, row)
__global__ __launch_bounds__(256) void laia_pairs_kernel(
    const uint32_t *__restrict__ samples, long long S, int T, long long start, int B, int W,
    const unsigned long long *__restrict__ mask, const int32_t *__restrict__ owner, long long R,
    uint32_t *__restrict__ plan_pairs, unsigned long long *__restrict__ plan_count,
    uint32_t *__restrict__ touch_pairs, int own_plan) {
    const long long total = static_cast<long long>(B) * T;
    const int lane = threadIdx.x & 63;
    // every wave makes the same number of trips (the ballots / shuffles below need all lanes)
    const long long trips = (total + gridDim.x * 256ll - 1) / (gridDim.x * 256ll);
    for (long long t = 0; t < trips; ++t) {
        const long long e = t * gridDim.x * 256ll + blockIdx.x * 256ll + threadIdx.x;
        unsigned long long m = 0;
        uint32_t emb = 0;
        if (e < total) {
            const long long i = e / T;
            const int j = static_cast<int>(e - i * T);
            emb = samples[((start + i) % S) * T + j];
            const int ow = owner[i];
            touch_pairs[e] = static_cast<uint32_t>(ow * R + emb);
            m = own_plan ? (mask[e] & (1ull << ow)) : (mask[e] & ~(1ull << ow));
        }
        // one counter bump per wave instead of one per pair (a quarter of a million on one address per batch)
        const int cnt = __builtin_popcountll(m);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += y;
        }
        const int wave_total = __shfl(incl, 63, 64);
        unsigned long long base = 0;
        if (lane == 63 && wave_total > 0)
            base = atomicAdd(plan_count, static_cast<unsigned long long>(wave_total));
        base = __shfl(base, 63, 64);
        unsigned long long pos = base + static_cast<unsigned long long>(incl - cnt);
        while (m) {
            const int w = __builtin_ctzll(m);
            m &= m - 1;
            plan_pairs[pos++] = static_cast<uint32_t>(w * R + emb);
        }
    }
}

// TopkScheduler scoring (topk_scheduler.cc:411-429): only the tables order[0..top_k) count, visited in
// that order; cand[i] = the worker whose running score first reached the sample's final maximum
// (worker 0 when no table hits).  One thread per sample.
struct TableOrder {
    int32_t t[64];
};
__global__ __launch_bounds__(256) void laia_topk_score_kernel(
    const unsigned long long *__restrict__ mask, int B, int T, int W, TableOrder order, int top_k,
    int32_t *__restrict__ scores, int32_t *__restrict__ cand) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B)
        return;
    int32_t *sc = scores + static_cast<long long>(i) * W;
    for (int z = 0; z < W; ++z)
        sc[z] = 0;
    int top = 0, c = 0;
    for (int k = 0; k < top_k; ++k) {
        unsigned long long m = mask[static_cast<long long>(i) * T + order.t[k]];
        while (m) {  // ascending worker index, like the z loop
            const int z = __builtin_ctzll(m);
            m &= m - 1;
            const int v = ++sc[z];
            if (v > top) {
                top = v;
                c = z;
            }
        }
    }
    cand[i] = c;
}

__global__ __launch_bounds__(256) void laia_delta_kernel(const uint32_t *__restrict__ dkeys,
                                                         const uint8_t *__restrict__ dvals,
                                                         long long n, uint8_t *__restrict__ valid) {
    // deltas of one batch are applied in list order per key; duplicates of a key are rare (a row
    // revalidated and then evicted) -- the host keeps only the LAST value per (worker,row)
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n; e += gridDim.x * 256ll)
        valid[dkeys[e]] = dvals[e];
}

// A few persistent host threads for the per-worker snapshot updates (creating std::threads per batch costs
// ~200 us, the update itself ~150 us per worker).  run(n, f): f(1) .. f(n-1) on the pool, f(0) on the caller.
struct SnapshotPool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> job;
    unsigned long long gen = 0;
    int njobs = 0, pending = 0;
    bool stop = false;

    void ensure(int n) {
        while (static_cast<int>(threads.size()) < n) {
            const int id = static_cast<int>(threads.size()) + 1;
            threads.emplace_back([this, id] {
                unsigned long long seen = 0;
                for (;;) {
                    std::function<void(int)> f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return stop || gen != seen; });
                        if (stop)
                            return;
                        seen = gen;
                        if (id >= njobs)
                            continue;
                        f = job;
                    }
                    f(id);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0)
                            cv_done.notify_one();
                    }
                }
            });
        }
    }
    void run(int n, const std::function<void(int)> &f) {
        if (n <= 1) {
            if (n == 1)
                f(0);
            return;
        }
        ensure(n - 1);
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f;
            njobs = n;
            pending = n - 1;
            ++gen;
        }
        cv_go.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~SnapshotPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_go.notify_all();
        for (auto &t : threads)
            t.join();
    }
};

struct Laia {
    SnapshotPool pool;
    long long S = 0, R = 0;
    int T = 0, W = 0, cache_size = 0;
    std::vector<uint64_t> samples_host;
    std::vector<Snapshot> snaps;
    hipStream_t stream = nullptr;
    uint32_t *d_samples = nullptr;
    uint8_t *d_valid = nullptr;
    unsigned long long *d_mask = nullptr, *d_count = nullptr;
    int32_t *d_scores = nullptr, *d_owner = nullptr, *d_cand = nullptr;
    // TopkScheduler traffic counters per worker (topk_scheduler.cc:319-331)
    std::vector<long long> miss_pull, miss_push, update_pull, update_push;
    uint32_t *d_plan_pairs = nullptr, *d_touch_pairs = nullptr, *d_dkeys = nullptr;
    uint8_t *d_dvals = nullptr;
    void *d_plan_ws = nullptr, *d_plan_ws2 = nullptr;
    uint32_t *h_touch = nullptr, *h_plan = nullptr;   // pinned: sorted-unique results (+ 2 words: their counts)
    size_t plan_cap = 0, delta_cap = 0;
    int Bcap = 0;
    std::vector<void *> allocs;
    // wall time per phase, summed over the calls (ha_laia_timing): the whole call, the host's greedy
    // assignment, the host's snapshot (MiniLRU) bookkeeping; the rest is GPU work, transfers and waits
    double t_total_us = 0, t_assign_us = 0, t_snap_us = 0;
    long long t_calls = 0;
};

static inline double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace ha

using namespace ha;

struct ha_laia {
    Laia l;
};

static int laia_alloc(Laia &l, void **p, size_t bytes) {
    HA_CHECK_HIP(hipMalloc(p, bytes + 256));
    l.allocs.push_back(*p);
    return 0;
}

extern "C" ha_laia *ha_laia_create(const uint64_t *samples_host, int64_t num_sample, int64_t num_table,
                                   int64_t nrank, int64_t cache_size, int64_t key_limit,
                                   int64_t max_batch) {
    if (!samples_host || num_sample <= 0 || num_table <= 0 || nrank <= 0 || nrank > 64 || cache_size < 0 ||
        key_limit <= 0 || max_batch <= 0 || static_cast<unsigned long long>(nrank) * key_limit > 0xFFFFFFFEull) {
        set_error("ha_laia_create: bad arguments (need nrank <= 64 and nrank*key_limit < 2^32)");
        return nullptr;
    }
    ha_laia *h = new ha_laia();
    Laia &l = h->l;
    l.S = num_sample;
    l.T = static_cast<int>(num_table);
    l.W = static_cast<int>(nrank);
    l.R = key_limit;
    l.cache_size = static_cast<int>(cache_size);
    l.Bcap = static_cast<int>(max_batch);
    l.samples_host.assign(samples_host, samples_host + num_sample * num_table);
    l.snaps.resize(l.W);
    // direct maps while all W of them stay below 2 GiB of host memory, hash maps beyond
    const bool use_direct = static_cast<unsigned long long>(l.W) * static_cast<unsigned long long>(l.R) * 4ull <=
                            (2ull << 30);
    for (auto &s : l.snaps)
        s.init(l.cache_size, l.R, use_direct);
    l.miss_pull.assign(l.W, 0);
    l.miss_push.assign(l.W, 0);
    l.update_pull.assign(l.W, 0);
    l.update_push.assign(l.W, 0);
    const size_t BT = static_cast<size_t>(l.Bcap) * l.T;
    l.plan_cap = BT * (l.W > 1 ? l.W - 1 : 1);
    if (l.plan_cap < BT)
        l.plan_cap = BT;
    l.delta_cap = l.plan_cap + BT * 2 + static_cast<size_t>(l.W) * 16;   // outdated plan keys + two per get()
    bool ok = hipStreamCreate(&l.stream) == hipSuccess;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_samples), num_sample * num_table * 4) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_valid), static_cast<size_t>(l.W) * l.R) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_mask), BT * 8) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_count), 8) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_scores), static_cast<size_t>(l.Bcap) * l.W * 4) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_owner), static_cast<size_t>(l.Bcap) * 4) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_cand), static_cast<size_t>(l.Bcap) * 4) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_plan_pairs), l.plan_cap * 4) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_touch_pairs), BT * 4) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_dkeys), l.delta_cap * 4) == 0;
    ok = ok && laia_alloc(l, reinterpret_cast<void **>(&l.d_dvals), l.delta_cap) == 0;
    ok = ok && laia_alloc(l, &l.d_plan_ws, ha_plan_bytes(static_cast<int64_t>(l.plan_cap))) == 0;
    ok = ok && laia_alloc(l, &l.d_plan_ws2, ha_plan_bytes(static_cast<int64_t>(BT))) == 0;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&l.h_touch), (BT + 4) * 4, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&l.h_plan), (l.plan_cap + 4) * 4, hipHostMallocDefault) == hipSuccess;
    if (ok) {
        std::vector<uint32_t> s32(l.samples_host.size());
        for (size_t i = 0; i < s32.size(); ++i)
            s32[i] = l.samples_host[i] > 0xFFFFFFFEull ? 0xFFFFFFFEu : static_cast<uint32_t>(l.samples_host[i]);
        ok = hipMemcpy(l.d_samples, s32.data(), s32.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemset(l.d_valid, 0, static_cast<size_t>(l.W) * l.R) == hipSuccess;
    }
    if (!ok) {
        if (g_err_is_empty())
            set_error("ha_laia_create: device allocation failed");
        for (void *p : l.allocs)
            (void)hipFree(p);
        if (l.h_touch)
            (void)hipHostFree(l.h_touch);
        if (l.h_plan)
            (void)hipHostFree(l.h_plan);
        if (l.stream)
            (void)hipStreamDestroy(l.stream);
        delete h;
        return nullptr;
    }
    return h;
}

extern "C" void ha_laia_destroy(ha_laia *h) {
    if (!h)
        return;
    (void)hipStreamSynchronize(h->l.stream);
    for (void *p : h->l.allocs)
        (void)hipFree(p);
    if (h->l.h_touch)
        (void)hipHostFree(h->l.h_touch);
    if (h->l.h_plan)
        (void)hipHostFree(h->l.h_plan);
    (void)hipStreamDestroy(h->l.stream);
    delete h;
}

// One global batch: get_dist(batch_id) followed by the snapshot update of launch().
//   dist_out   [W * mini_bs]   global sample indices per worker (laia_scheduler.cc:245)
//   plan_out   concatenated sorted plans of workers 0..W-1, plan_off[W+1] their offsets;
//              plan_cap_elems bounds plan_out.
struct TopkParams {
    const int32_t *order;  // table order, top_k entries used
    int top_k, num_threads;
};

// thread slice of `total` items: thread 0 takes the remainder (topk_scheduler.cc:398-407)
static void topk_slice(long long total, int nt, int t, long long *s0, long long *s1) {
    const long long x = total / nt, y = total % nt;
    *s0 = t == 0 ? 0 : y + t * x;
    *s1 = t == 0 ? x + y : *s0 + x;
}

static int laia_next_impl(ha_laia *h, int64_t batch_id, int64_t mini_bs, int64_t *dist_out,
                          uint64_t *plan_out, int64_t plan_cap_elems, int64_t *plan_off,
                          const TopkParams *topk) {
    HA_REQUIRE(h && dist_out && plan_out && plan_off && mini_bs > 0, "laia_next: bad arguments");
    Laia &l = h->l;
    const int W = l.W, T = l.T;
    const long long B = mini_bs * W;
    HA_REQUIRE(B <= l.Bcap, "laia_next: global batch %lld exceeds max_batch %d", B, l.Bcap);
    const double t_begin = now_us();
    const long long start = (batch_id * B) % l.S;  // laia_scheduler.cc:182
    const long long BT = B * T;
    int blocks = static_cast<int>((BT + 255) / 256);
    if (blocks > 4096)
        blocks = 4096;
    // ---- score
    hipLaunchKernelGGL(laia_probe_kernel, dim3(blocks), dim3(256), 0, l.stream, l.d_samples, l.S, T,
                       start, (int)B, W, l.d_valid, l.R, l.d_mask);
    std::vector<int32_t> cand;
    if (topk) {
        HA_REQUIRE(topk->top_k >= 1 && topk->top_k <= T && T <= 64 && topk->num_threads >= 1,
                   "laia_next_topk: need 1 <= top_k <= num_table <= 64 and num_threads >= 1");
        TableOrder order;
        for (int k = 0; k < topk->top_k; ++k) {
            HA_REQUIRE(topk->order[k] >= 0 && topk->order[k] < T, "laia_next_topk: table index out of range");
            order.t[k] = topk->order[k];
        }
        hipLaunchKernelGGL(laia_topk_score_kernel, dim3((int)((B + 255) / 256)), dim3(256), 0, l.stream,
                           l.d_mask, (int)B, T, W, order, topk->top_k, l.d_scores, l.d_cand);
        cand.resize(static_cast<size_t>(B));
    } else {
        hipLaunchKernelGGL(laia_score_kernel, dim3((int)((B * W + 255) / 256)), dim3(256), 0, l.stream,
                           l.d_mask, (int)B, T, W, l.d_scores);
    }
    HA_LAUNCH_CHECK();
    std::vector<int32_t> scores(static_cast<size_t>(B) * W);
    HA_CHECK_HIP(hipMemcpyAsync(scores.data(), l.d_scores, scores.size() * 4, hipMemcpyDeviceToHost, l.stream));
    if (topk)
        HA_CHECK_HIP(hipMemcpyAsync(cand.data(), l.d_cand, cand.size() * 4, hipMemcpyDeviceToHost, l.stream));
    HA_CHECK_HIP(hipStreamSynchronize(l.stream));
    const double t_assign0 = now_us();
    std::vector<int32_t> owner(static_cast<size_t>(B));
    if (topk) {
        // ---- assign (topk_scheduler.cc:393-455): the batch and every worker's quota are cut into
        // per-thread slices; each slice is assigned sequentially, offering a sample to the workers in the
        // order (j + candidate) % W, strictly greater score wins, stop at the candidate itself
        const int nt = topk->num_threads;
        for (int t = 0; t < nt; ++t) {
            long long s0, s1, q0, q1;
            topk_slice(B, nt, t, &s0, &s1);
            topk_slice(mini_bs, nt, t, &q0, &q1);
            HA_REQUIRE(s1 - s0 <= W * (q1 - q0),
                       "laia_next_topk: thread %d has %lld samples for %d workers x quota %lld (the reference "
                       "writes dist[-1] here); pick num_threads dividing mini_batch_size", t, s1 - s0, W, q1 - q0);
        }
        for (long long k = 0; k < static_cast<long long>(W) * mini_bs; ++k)
            dist_out[k] = 0;  // dist.reset(0), topk_scheduler.cc:382
        std::vector<long long> wl(W);
        for (int t = 0; t < nt; ++t) {
            long long s0, s1, q0, q1;
            topk_slice(B, nt, t, &s0, &s1);
            topk_slice(mini_bs, nt, t, &q0, &q1);
            std::fill(wl.begin(), wl.end(), 0);
            for (long long i = s0; i < s1; ++i) {
                const int c = cand[static_cast<size_t>(i)];
                int best = -1, best_w = -1;
                for (int j = 0; j < W; ++j) {
                    const int w = (j + c) % W;
                    const int sc = scores[static_cast<size_t>(i) * W + w];
                    if (best < sc && wl[w] < q1 - q0) {
                        best = sc;
                        best_w = w;
                        if (best_w == c)
                            break;
                    }
                }
                dist_out[static_cast<size_t>(best_w) * mini_bs + q0 + wl[best_w]] = (i + start) % l.S;
                wl[best_w] += 1;
                owner[static_cast<size_t>(i)] = best_w;
            }
        }
    }
    // ---- assign (laia_scheduler.cc:226-249): sequential, capacity mini_bs per worker, workers visited
    // in the order (j + batch_id) % W, strictly greater score wins
    std::vector<long long> workload(W, 0);
    for (long long i = 0; !topk && i < B; ++i) {
        int max_score = -1, max_worker = -1;
        for (int j = 0; j < W; ++j) {
            const int w = static_cast<int>((j + batch_id) % W);
            const int sc = scores[static_cast<size_t>(i) * W + w];
            if (workload[w] < mini_bs && max_score < sc) {
                max_score = sc;
                max_worker = w;
            }
        }
        dist_out[static_cast<size_t>(max_worker) * mini_bs + workload[max_worker]] = (i + start) % l.S;
        workload[max_worker] += 1;
        owner[static_cast<size_t>(i)] = max_worker;
    }
    l.t_assign_us += now_us() - t_assign0;
    // ---- plan + touched rows
    HA_CHECK_HIP(hipMemcpyAsync(l.d_owner, owner.data(), owner.size() * 4, hipMemcpyHostToDevice, l.stream));
    HA_CHECK_HIP(hipMemsetAsync(l.d_count, 0, 8, l.stream));
    hipLaunchKernelGGL(laia_pairs_kernel, dim3(blocks), dim3(256), 0, l.stream, l.d_samples, l.S, T, start,
                       (int)B, W, l.d_mask, l.d_owner, l.R, l.d_plan_pairs, l.d_count, l.d_touch_pairs,
                       topk ? 1 : 0);
    HA_LAUNCH_CHECK();
    // Two synchronisations for the rest of the GPU part (five before): the sort of the touched pairs does not need
    // the number of plan pairs, so it is enqueued first and its result -- count and all BT candidate entries --
    // comes back together with that number; the plan pairs' sort follows and returns the same way.
    std::vector<uint32_t> plan_keys, touch_keys;
    int key_bits = 1;
    while ((static_cast<unsigned long long>(l.W) * l.R) >> key_bits)
        ++key_bits;
    unsigned long long npairs = 0;
    long long u_touch = 0, u_plan = 0;
    if (ha_plan_build_u32keys(l.d_touch_pairs, static_cast<int64_t>(BT), l.d_plan_ws2, key_bits, l.stream))
        return -1;
    {
        PlanPtrs pt = plan_layout(l.d_plan_ws2, static_cast<int64_t>(BT));
        HA_CHECK_HIP(hipMemcpyAsync(&npairs, l.d_count, 8, hipMemcpyDeviceToHost, l.stream));
        HA_CHECK_HIP(hipMemcpyAsync(&u_touch, &pt.hdr->n_unique, 8, hipMemcpyDeviceToHost, l.stream));
        HA_CHECK_HIP(hipMemcpyAsync(l.h_touch, pt.uniq, static_cast<size_t>(BT) * 4, hipMemcpyDeviceToHost, l.stream));
        HA_CHECK_HIP(hipStreamSynchronize(l.stream));
        touch_keys.assign(l.h_touch, l.h_touch + u_touch);
    }
    if (npairs > 0) {
        if (ha_plan_build_u32keys(l.d_plan_pairs, static_cast<int64_t>(npairs), l.d_plan_ws, key_bits, l.stream))
            return -1;
        PlanPtrs pp = plan_layout(l.d_plan_ws, static_cast<int64_t>(npairs));
        HA_CHECK_HIP(hipMemcpyAsync(&u_plan, &pp.hdr->n_unique, 8, hipMemcpyDeviceToHost, l.stream));
        HA_CHECK_HIP(hipMemcpyAsync(l.h_plan, pp.uniq, static_cast<size_t>(npairs) * 4, hipMemcpyDeviceToHost, l.stream));
        HA_CHECK_HIP(hipStreamSynchronize(l.stream));
        plan_keys.assign(l.h_plan, l.h_plan + u_plan);
    }
    // ---- emit plans (composite keys are sorted by worker, then row)
    HA_REQUIRE(static_cast<long long>(plan_keys.size()) <= plan_cap_elems, "laia_next: plan buffer too small");
    {
        size_t k = 0;
        for (int w = 0; w < W; ++w) {
            plan_off[w] = static_cast<int64_t>(k);
            while (k < plan_keys.size() && plan_keys[k] / l.R == static_cast<unsigned long long>(w)) {
                plan_out[k] = plan_keys[k] - static_cast<unsigned long long>(w) * l.R;
                ++k;
            }
        }
        plan_off[W] = static_cast<int64_t>(plan_keys.size());
    }
    // ---- snapshot update (laia_scheduler.cc:146-162): outdate the plan keys, then get() every unique
    // touched key in ascending order
    std::vector<uint32_t> dkeys;
    std::vector<uint8_t> dvals;
    const double t_snap0 = now_us();
    {
        // the W snapshots are independent (the reference keeps one per worker and walks them one after
        // the other, laia_scheduler.cc:146-162): one host thread per worker
        std::vector<size_t> pk0(W + 1), tk0(W + 1);
        {
            size_t pk = 0, tk = 0;
            for (int w = 0; w < W; ++w) {
                pk0[w] = pk;
                tk0[w] = tk;
                while (pk < plan_keys.size() && plan_keys[pk] / l.R == static_cast<unsigned long long>(w))
                    ++pk;
                while (tk < touch_keys.size() && touch_keys[tk] / l.R == static_cast<unsigned long long>(w))
                    ++tk;
            }
            pk0[W] = pk;
            tk0[W] = tk;
        }
        std::vector<std::vector<int32_t>> touched(W);
        std::vector<std::vector<uint8_t>> tstate(W);
        auto work = [&](int w) {
            const unsigned long long base = static_cast<unsigned long long>(w) * l.R;
            std::vector<int32_t> td;     // thread-local (the shared vectors' headers are neighbours in memory)
            td.reserve((pk0[w + 1] - pk0[w]) + 2 * (tk0[w + 1] - tk0[w]));
            for (size_t pk = pk0[w]; pk < pk0[w + 1]; ++pk)
                l.snaps[w].outdate(static_cast<int32_t>(plan_keys[pk] - base), td);

            l.update_push[w] += static_cast<long long>(pk0[w + 1] - pk0[w]);
            // per-thread counters, written back once: the workers' slots of the shared arrays sit in one cache
            // line, and bumping them per key made the four threads fight over it (300+ ns per key)
            long long c_update_pull = 0, c_miss_pull = 0, c_miss_push = 0;
            for (size_t tk = tk0[w]; tk < tk0[w + 1]; ++tk) {
                // the direct map and the list nodes are random host-memory accesses: look a few keys ahead
                if (l.snaps[w].direct && tk + 16 < tk0[w + 1])
                    __builtin_prefetch(&l.snaps[w].dmap[touch_keys[tk + 16] - base]);
                if (l.snaps[w].direct && tk + 4 < tk0[w + 1]) {
                    const int x = l.snaps[w].dmap[touch_keys[tk + 4] - base];
                    if (x >= 0)
                        __builtin_prefetch(&l.snaps[w].node[x]);
                }
                const int res = l.snaps[w].get(static_cast<int32_t>(touch_keys[tk] - base), td);
                if (res < 0) {  // traffic counters, topk_scheduler.cc:319-331
                    if (res == -2)
                        c_update_pull += 1;
                } else {
                    c_miss_pull += 1;
                    if (res > 0)
                        c_miss_push += 1;
                }
            }
            l.update_pull[w] += c_update_pull;
            l.miss_pull[w] += c_miss_pull;
            l.miss_push[w] += c_miss_push;
            // the final validity of every key whose byte may have changed, looked up here (in the worker's own
            // thread, while its nodes are warm) rather than by the merging thread
            std::vector<uint8_t> ts(td.size());
            for (size_t i = 0; i < td.size(); ++i)
                ts[i] = l.snaps[w].state(td[i]);
            touched[w] = std::move(td);
            tstate[w] = std::move(ts);
        };
        if (W > 1 && touch_keys.size() > 4096) {
            l.pool.run(W, work);
        } else {
            for (int w = 0; w < W; ++w)
                work(w);
        }
        for (int w = 0; w < W; ++w) {
            const unsigned long long base = static_cast<unsigned long long>(w) * l.R;
            for (size_t i = 0; i < touched[w].size(); ++i) {
                dkeys.push_back(static_cast<uint32_t>(base + static_cast<uint32_t>(touched[w][i])));
                dvals.push_back(tstate[w][i]);
            }
        }
    }
    l.t_snap_us += now_us() - t_snap0;
    if (!dkeys.empty()) {
        HA_REQUIRE(dkeys.size() <= l.delta_cap, "laia_next: delta buffer too small");
        HA_CHECK_HIP(hipMemcpyAsync(l.d_dkeys, dkeys.data(), dkeys.size() * 4, hipMemcpyHostToDevice, l.stream));
        HA_CHECK_HIP(hipMemcpyAsync(l.d_dvals, dvals.data(), dvals.size(), hipMemcpyHostToDevice, l.stream));
        hipLaunchKernelGGL(laia_delta_kernel, dim3((int)((dkeys.size() + 255) / 256)), dim3(256), 0, l.stream,
                           l.d_dkeys, l.d_dvals, (long long)dkeys.size(), l.d_valid);
        HA_LAUNCH_CHECK();
        HA_CHECK_HIP(hipStreamSynchronize(l.stream));
    }
    l.t_total_us += now_us() - t_begin;
    l.t_calls += 1;
    return 0;
}

// out[4] = {calls, total us, host assignment us, host snapshot us} summed since creation
extern "C" int ha_laia_timing(ha_laia *h, double *out) {
    HA_REQUIRE(h && out, "ha_laia_timing: null pointer");
    out[0] = static_cast<double>(h->l.t_calls);
    out[1] = h->l.t_total_us;
    out[2] = h->l.t_assign_us;
    out[3] = h->l.t_snap_us;
    return 0;
}

extern "C" int ha_laia_next(ha_laia *h, int64_t batch_id, int64_t mini_bs, int64_t *dist_out,
                            uint64_t *plan_out, int64_t plan_cap_elems, int64_t *plan_off) {
    return laia_next_impl(h, batch_id, mini_bs, dist_out, plan_out, plan_cap_elems, plan_off, nullptr);
}

extern "C" int ha_laia_next_topk(ha_laia *h, int64_t batch_id, int64_t mini_bs,
                                 const int32_t *table_order, int64_t top_k, int64_t num_threads,
                                 int64_t *dist_out, uint64_t *plan_out, int64_t plan_cap_elems,
                                 int64_t *plan_off) {
    HA_REQUIRE(table_order && top_k >= 1 && top_k <= 64 && num_threads >= 1 && num_threads <= (1 << 20),
               "laia_next_topk: bad arguments");
    const TopkParams tp{table_order, static_cast<int>(top_k), static_cast<int>(num_threads)};
    return laia_next_impl(h, batch_id, mini_bs, dist_out, plan_out, plan_cap_elems, plan_off, &tp);
}

// out[4*W] = miss_pull[W], miss_push[W], update_pull[W], update_push[W] accumulated so far
extern "C" int ha_laia_counters(ha_laia *h, int64_t *out) {
    HA_REQUIRE(h && out, "laia_counters: bad arguments");
    const Laia &l = h->l;
    for (int w = 0; w < l.W; ++w) {
        out[w] = l.miss_pull[w];
        out[l.W + w] = l.miss_push[w];
        out[2 * l.W + w] = l.update_pull[w];
        out[3 * l.W + w] = l.update_push[w];
    }
    return 0;
}

// valid keys of worker w's snapshot, ascending (MiniLRUCache::get_keys); returns the count
extern "C" int64_t ha_laia_snapshot_keys(ha_laia *h, int64_t w, int32_t *out, int64_t cap) {
    if (!h || w < 0 || w >= h->l.W)
        return -1;
    Snapshot &s = h->l.snaps[static_cast<size_t>(w)];
    std::vector<int32_t> keys;
    s.valid_keys(keys);
    std::sort(keys.begin(), keys.end());
    for (size_t i = 0; i < keys.size() && static_cast<int64_t>(i) < cap; ++i)
        out[i] = keys[i];
    return static_cast<int64_t>(keys.size());
}

// ---- local-shared plan distribution -----------------------------------------------------------------
// The reference's TopkScheduler runs on local rank 0 only and hands every local worker its
// [plan, dist] stream through a boost::interprocess shared-memory ring named "laia_cache_<i>"
// (laia/include/share_mem.h:40-193, ring_buffer.h:13-125).  Same topology here (one process per GPU):
// a single-producer / single-consumer ring of uint64 words in POSIX shared memory, one message =
// [length, payload...]; send / recv never block (the callers poll, topk_scheduler.cc:204-247).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <string>

struct ha_shm_ring {
    struct Header {
        std::atomic<uint64_t> head;  // words consumed
        std::atomic<uint64_t> tail;  // words produced
        uint64_t capacity;           // payload words (power of two)
        uint64_t pad[5];
    };
    Header *hdr = nullptr;
    uint64_t *data = nullptr;
    size_t bytes = 0;
    std::string name;
    bool owner = false;
};

extern "C" ha_shm_ring *ha_shm_ring_open(const char *name, int create, int64_t capacity_words) {
    if (!name || (create && capacity_words < 16)) {
        ha::set_error("ha_shm_ring_open: bad arguments");
        return nullptr;
    }
    std::string nm = name[0] == '/' ? std::string(name) : "/" + std::string(name);
    uint64_t cap = 16;
    if (create) {
        while (cap < static_cast<uint64_t>(capacity_words))
            cap <<= 1;
        (void)shm_unlink(nm.c_str());
    }
    const int fd = shm_open(nm.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) {
        ha::set_error("ha_shm_ring_open: shm_open(%s) failed", nm.c_str());
        return nullptr;
    }
    size_t bytes = 0;
    if (create) {
        bytes = sizeof(ha_shm_ring::Header) + cap * 8;
        if (ftruncate(fd, static_cast<off_t>(bytes)) != 0) {
            close(fd);
            (void)shm_unlink(nm.c_str());
            ha::set_error("ha_shm_ring_open: ftruncate failed");
            return nullptr;
        }
    } else {
        struct stat st;
        if (fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < sizeof(ha_shm_ring::Header) + 16 * 8) {
            close(fd);
            ha::set_error("ha_shm_ring_open: %s is not an initialised ring", nm.c_str());
            return nullptr;
        }
        bytes = static_cast<size_t>(st.st_size);
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        ha::set_error("ha_shm_ring_open: mmap failed");
        return nullptr;
    }
    ha_shm_ring *r = new ha_shm_ring();
    r->hdr = static_cast<ha_shm_ring::Header *>(p);
    r->data = reinterpret_cast<uint64_t *>(r->hdr + 1);
    r->bytes = bytes;
    r->name = nm;
    r->owner = create != 0;
    if (create) {
        r->hdr->head.store(0);
        r->hdr->tail.store(0);
        r->hdr->capacity = cap;
    }
    return r;
}

extern "C" void ha_shm_ring_close(ha_shm_ring *r) {
    if (!r)
        return;
    munmap(r->hdr, r->bytes);
    if (r->owner)
        (void)shm_unlink(r->name.c_str());
    delete r;
}

// 1 = sent, 0 = not enough room right now, -1 = message can never fit
extern "C" int ha_shm_ring_send(ha_shm_ring *r, const uint64_t *words, int64_t n) {
    if (!r || n < 0 || (n > 0 && !words))
        return -1;
    const uint64_t cap = r->hdr->capacity, need = static_cast<uint64_t>(n) + 1;
    if (need > cap)
        return -1;
    const uint64_t tail = r->hdr->tail.load(std::memory_order_relaxed);
    const uint64_t head = r->hdr->head.load(std::memory_order_acquire);
    if (cap - (tail - head) < need)
        return 0;
    r->data[tail & (cap - 1)] = static_cast<uint64_t>(n);
    for (int64_t i = 0; i < n; ++i)
        r->data[(tail + 1 + static_cast<uint64_t>(i)) & (cap - 1)] = words[i];
    r->hdr->tail.store(tail + need, std::memory_order_release);
    return 1;
}

// >= 0: length of the received message (copied to out); -1: nothing to read; -2: out too small
// (the message stays queued; *needed = its length)
extern "C" int64_t ha_shm_ring_recv(ha_shm_ring *r, uint64_t *out, int64_t cap_words, int64_t *needed) {
    if (!r)
        return -1;
    const uint64_t cap = r->hdr->capacity;
    const uint64_t head = r->hdr->head.load(std::memory_order_relaxed);
    const uint64_t tail = r->hdr->tail.load(std::memory_order_acquire);
    if (tail == head)
        return -1;
    const uint64_t n = r->data[head & (cap - 1)];
    if (needed)
        *needed = static_cast<int64_t>(n);
    if (static_cast<int64_t>(n) > cap_words)
        return -2;
    for (uint64_t i = 0; i < n; ++i)
        out[i] = r->data[(head + 1 + i) & (cap - 1)];
    r->hdr->head.store(head + 1 + n, std::memory_order_release);
    return static_cast<int64_t>(n);
}

// number of queued messages is not tracked; queued words > 0 <=> something to read
extern "C" int64_t ha_shm_ring_pending_words(ha_shm_ring *r) {
    if (!r)
        return 0;
    return static_cast<int64_t>(r->hdr->tail.load(std::memory_order_acquire) -
                                r->hdr->head.load(std::memory_order_acquire));
}
