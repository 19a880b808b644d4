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

// probe + score in one launch: a wave takes floor(64 / T) whole samples, lane = (sample, table); mask as laia_probe_kernel,
// scores[i][w] = number of tables of sample i valid at w by ballots over the sample's lanes (T <= 64)
__global__ __launch_bounds__(256) void laia_probe_score_kernel(
    const uint32_t *__restrict__ samples, long long S, int T, long long start, int B, int W,
    const uint8_t *__restrict__ valid, long long R, unsigned long long *__restrict__ mask, int32_t *__restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int spw = 64 / T;                                     // samples per wave
    const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int sl = lane / T, j = lane - sl * T;                 // sample slot of the lane, table
    const long long i = wave * spw + sl;
    const bool on = sl < spw && i < B;
    unsigned long long m = 0;
    if (on) {
        const uint32_t emb = samples[((start + i) % S) * T + j];
        if (emb < R)
            for (int w = 0; w < W; ++w)
                if (valid[static_cast<long long>(w) * R + emb])
                    m |= 1ull << w;
        mask[i * T + j] = m;
    }
    const unsigned long long seg = (T >= 64 ? ~0ull : ((1ull << T) - 1ull)) << (sl * T);   // the lanes of this lane's sample
    for (int w = 0; w < W; ++w) {
        const unsigned long long b = __ballot(on && ((m >> w) & 1ull));
        if (on && j == 0)
            scores[i * W + w] = __builtin_popcountll(b & seg);
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

// =====================================================================================================
// Device-resident scheduler state (round 3): the MiniLRU snapshots, the greedy assignment and the sorted-unique
// key lists on the GPU -- per global batch the host only launches kernels and receives `dist` and the plans.
// Semantics: LaiaScheduler::get_dist + the snapshot update of launch() (laia/src/laia_scheduler.cc:146-271),
// MiniLRUCache get / insert / outdate (laia/include/mini_lru_cache.h:54-128).
//
//  * assignment (laia_scheduler.cc:226-249: samples in order, strictly greater score wins among the workers that
//    still have quota, visited in the order (j + batch_id) % W): until a worker's quota fills every sample's choice
//    is independent of the others, so ONE workgroup runs rounds of { argmax over the available workers, prefix
//    counts per worker, first sample at which a quota fills }: at most W + 1 rounds instead of B dependent steps.
//  * sorted-unique (worker, row) lists without a sort: laia_bits_kernel sets one flag byte per touched / plan pair (and per
//    group of 64 rows, per chunk of 4,096) with plain stores, laia_bits_pack_kernel turns the flags of the chunks that have
//    any into bit words (one per 64 rows) + a summary word per chunk and counts them, and an ordered compaction (emit --
//    which also clears what it read) yields the rows ascending per worker.
//  * MiniLRU per worker as a stamp log (as the host Snapshot above and cache.hip): stamp[w][row] (0 = absent), a
//    ring log of (row, stamp), valid bytes = the device mirror the probe reads.  A batch's get()s come in ascending
//    row order and the rows are distinct, so row i of n ends with stamp counter + i + 1 whatever happens; what has
//    to be worked out is the eviction victims.  With live0 lines resident, A rows of the batch absent at its start
//    and capacity cap (>= n, checked at creation): exactly N = max(0, live0 + A - cap) lines that are NOT in the
//    batch are evicted -- the first N such live log entries from the head -- independent of the order of events.
//    A line that IS in the batch can be evicted before its turn and re-inserted (one more miss, one more eviction);
//    going through the batch lines among the first entries in log order: line j at batch position p is evicted
//    early iff  live0 + A(p-1) + EB(p-1) - cap >= 1 + (non-batch lines before j) + (early-evicted lines before j),
//    EB(i) = early-evicted lines found so far with position <= i.  /tmp prototype and tests/test_gpu_laia.py hold
//    this against the sequential MiniLRU model.
constexpr int kLaiaAssignPer = 16;                 // samples per thread of the assignment workgroup: B <= 16,384
constexpr int kLaiaBitsChunk = 4096;               // rows per summary bit-word (64 data words)

struct LruState {   // per worker, device memory
    unsigned long long log_head, log_size;
    unsigned long long miss_pull, miss_push, update_pull, update_push;
    uint32_t counter;
    int32_t live;
    int32_t err;      // sticky: 1 = log window ran out (never expected), 2 = candidate list overflow
    uint32_t apply_counter0;           // for laia_lru_apply_kernel: stamp of the batch's first row - 1 ...
    unsigned long long apply_tail;     // ... and the log position of its entry
};

// ---- assignment ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long laia_block_scan_u64(unsigned long long v, unsigned long long *s_w,
                                                                  unsigned long long *total) {
    // exclusive scan over the 1024 threads of a workgroup; every 16-bit field of v stays below 2^16 in total
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(incl, o, 64);
        if (lane >= o)
            incl += y;
    }
    if (lane == 63)
        s_w[wv] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
        const unsigned long long x = s_w[k];
        if (k < wv)
            base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(1024) void laia_assign_kernel(const int32_t *__restrict__ scores, int B, int W, int mini_bs,
                                                           long long batch_id, long long start, long long S,
                                                           int32_t *__restrict__ owner, long long *__restrict__ dist,
                                                           int lds_scores) {
    __shared__ unsigned long long s_w[16];
    __shared__ int s_base[64], s_add[64];
    __shared__ int s_istar;
    __shared__ signed char s_pref[1024 * kLaiaAssignPer];   // choice of sample i this round, -1 = already final
    extern __shared__ unsigned char s_sc[];                 // the scores as bytes (<= tables <= 255), when B * W fits
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int i0 = t * per, i1 = min(B, i0 + per);
    const int rot = static_cast<int>(batch_id % W);
    if (t < 64) {
        s_base[t] = 0;
        s_add[t] = 0;
    }
    unsigned long long avail = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
    int i_start = 0;
    if (lds_scores)
        for (int e = t; e < B * W; e += 1024)
            s_sc[e] = static_cast<unsigned char>(scores[e]);
    __syncthreads();
    while (i_start < B) {
        // choice of every open sample among the workers that still have quota (a thread reads only its own entries)
        for (int i = i0; i < i1; ++i) {
            int bw = -1;
            if (i >= i_start) {
                int best = -1;
                for (int j = 0; j < W; ++j) {
                    int w = j + rot;
                    w = w >= W ? w - W : w;
                    if (!((avail >> w) & 1ull))
                        continue;
                    const int sc = lds_scores ? static_cast<int>(s_sc[i * W + w]) : scores[static_cast<long long>(i) * W + w];
                    if (best < sc) {
                        best = sc;
                        bw = w;
                    }
                }
            }
            s_pref[i] = static_cast<signed char>(bw);
        }
        if (t == 0)
            s_istar = B;      // = nobody fills: the rest of the batch is final
        __syncthreads();
        // groups of four workers: 16-bit packed counts, ONE scan per group and round -- the number of earlier open
        // samples with the same choice gives both the sample that fills a quota and every sample's slot
        unsigned long long exs[16];       // W <= 64: up to 16 groups
        for (int g = 0; g * 4 < W; ++g) {
            unsigned long long mine = 0;
            for (int i = i0; i < i1; ++i) {
                const int pf = s_pref[i];
                if (pf >= 0 && (pf >> 2) == g)
                    mine += 1ull << (16 * (pf & 3));
            }
            unsigned long long tot;
            const unsigned long long ex = laia_block_scan_u64(mine, s_w, &tot);
            exs[g] = ex;
            for (int f = 0; f < 4 && 4 * g + f < W; ++f) {
                const int w = 4 * g + f;
                const int need = mini_bs - s_base[w];            // occurrences until w is full (>= 1 while available)
                const int before = static_cast<int>((ex >> (16 * f)) & 0xFFFFull);
                const int local = static_cast<int>((mine >> (16 * f)) & 0xFFFFull);
                if (((avail >> w) & 1ull) && before < need && need <= before + local) {
                    int seen = before;
                    for (int i = i0; i < i1; ++i)
                        if (s_pref[i] == w && ++seen == need)
                            atomicMin(&s_istar, i);
                }
            }
        }
        __syncthreads();
        const int istar = s_istar;      // last sample of this round (B: all remaining)
        // finalise [i_start, istar]: slots in sample order
        for (int i = i0; i < i1 && i <= istar; ++i) {
            const int pf = s_pref[i];
            if (pf >= 0) {
                const int g = pf >> 2, f = pf & 3;
                const int slot = s_base[pf] + static_cast<int>((exs[g] >> (16 * f)) & 0xFFFFull);
                exs[g] += 1ull << (16 * f);
                owner[i] = pf;
                dist[static_cast<long long>(pf) * mini_bs + slot] = (i + start) % S;
                atomicAdd(&s_add[pf], 1);
            }
        }
        __syncthreads();
        if (t < W) {
            s_base[t] += s_add[t];
            s_add[t] = 0;
        }
        __syncthreads();
        for (int w = 0; w < W; ++w)
            if (s_base[w] >= mini_bs)
                avail &= ~(1ull << w);
        i_start = istar + 1;
        __syncthreads();
    }
}

// TopkScheduler's assignment (topk_scheduler.cc:393-455): the batch and every worker's quota are cut into `nt` thread
// slices (thread 0 takes the remainders, :398-407); a slice is assigned sequentially -- a sample is offered to the
// workers in the order (j + candidate) % W, strictly greater score wins among the workers whose SLICE quota is not
// used up, the walk stops at the candidate itself --, slices are independent of each other: one lane per slice.
__global__ __launch_bounds__(128) void laia_topk_assign_kernel(const int32_t *__restrict__ scores, const int32_t *__restrict__ cand,
                                                               int B, int W, int mini_bs, int nt, long long start, long long S,
                                                               int32_t *__restrict__ owner, long long *__restrict__ dist) {
    const int t = blockIdx.x * 128 + threadIdx.x;
    if (t >= nt)
        return;
    const long long sx = B / nt, sy = B % nt, qx = mini_bs / nt, qy = mini_bs % nt;
    const long long s0 = t == 0 ? 0 : sy + t * sx, s1 = t == 0 ? sx + sy : s0 + sx;
    const long long q0 = t == 0 ? 0 : qy + t * qx, q1 = t == 0 ? qx + qy : q0 + qx;
    int wl[64];
    for (int w = 0; w < W; ++w)
        wl[w] = 0;
    for (long long i = s0; i < s1; ++i) {
        const int c = cand[i];
        int best = -1, best_w = -1;
        for (int j = 0; j < W; ++j) {
            const int w = (j + c) % W;
            const int sc = scores[i * W + w];
            if (best < sc && wl[w] < q1 - q0) {
                best = sc;
                best_w = w;
                if (best_w == c)
                    break;
            }
        }
        dist[static_cast<long long>(best_w) * mini_bs + q0 + wl[best_w]] = (i + start) % S;
        wl[best_w] += 1;
        owner[i] = best_w;
    }
}

// ---- (worker, row) bitmaps ----------------------------------------------------------------------------------
// touch: (owner of the sample, row); plan: (w, row) for the rows valid at w in samples not assigned to w.
// Thread e takes (table j, sample i) = (e / B, e % B): the lanes of a wave hold consecutive samples of ONE table, so a hot
// row -- named by thousands of samples of a batch -- shows up many times per wave.  Every distinct bit of a wave is set
// by one lane (leader loop over the distinct values), and only if a look at the word says it is not set yet: a few
// atomics per hot row and batch instead of thousands on one address.
// One flag BYTE per (worker, row) pair named and one per group of 64 rows -- plain stores, no
// atomics, no look, no duplicate elimination (a flag is set to 1 by whoever names it).  Until round 6 this launch set BITS
// with atomicOr (and counted the new ones per chunk with atomicAdd): 36-46 us of a 120 us global batch, all of it the ~130 k
// scattered device-scope atomics of a batch -- they execute at the memory side, one 64-byte request each, and a batch's rows
// share no lines (stage exits: loads 4.7 us, + duplicate elimination 7.3, + looks 7.7, + atomics 45.8; one atomic per wave and
// address instead of one per lane: 43 -- docs/EXPERIMENTS.md round 6 section 10).  laia_bits_pack_kernel turns the flags of the
// chunks that have any into the bit words the compaction reads.
struct LaiaFlags {
    uint8_t *row[2];        // [W * Rpad]        0 = touch, 1 = plan
    uint8_t *g64[2];        // [W * Rpad / 64]   (a chunk of 4,096 rows = 64 of these = one 64-byte line)
};
__global__ __launch_bounds__(256) void laia_bits_kernel(const uint32_t *__restrict__ samples, long long S, int T,
                                                        long long start, int B, int W,
                                                        const unsigned long long *__restrict__ mask,
                                                        const int32_t *__restrict__ owner, long long R, long long Rpad,
                                                        LaiaFlags f, int own_plan) {
    const long long total = static_cast<long long>(B) * T;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        const int j = static_cast<int>(e / B);
        const long long i = e - static_cast<long long>(j) * B;
        const uint32_t emb = samples[((start + i) % S) * T + j];
        if (emb >= R)
            continue;
        const int ow = owner[i];
        // LaiaScheduler: rows valid at w in samples NOT assigned to w (laia_scheduler.cc:252-270); TopkScheduler: rows of
        // w's OWN samples that w holds valid (topk_scheduler.cc:468-500)
        unsigned long long m = own_plan ? (mask[i * T + j] & (1ull << ow)) : (mask[i * T + j] & ~(1ull << ow));
        const unsigned long long b = static_cast<unsigned long long>(ow) * Rpad + emb;
        f.row[0][b] = 1;
        f.g64[0][b >> 6] = 1;
        while (m) {
            const int w = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned long long p = static_cast<unsigned long long>(w) * Rpad + emb;
            f.row[1][p] = 1;
            f.g64[1][p >> 6] = 1;
        }
    }
}

struct LaiaBits {
    unsigned long long *bits[2], *sum[2];   // 0 = touch, 1 = plan
    uint32_t *rows[2];                      // output: rows ascending per worker
    int32_t *off[2];                        // [W + 1]
    uint32_t *cnt[2];                       // exclusive offsets inside a block  [nsum]
    uint32_t *blk[2];                       // per-block totals, then exclusive offsets   [nblk + 1]
};

__device__ __forceinline__ uint32_t laia_block_scan_u32(uint32_t v, uint32_t *s_w, uint32_t *total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o, 64);
        if (lane >= o)
            incl += y;
    }
    if (lane == 63)
        s_w[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int k = 0; k < nw; ++k) {
        const uint32_t x = s_w[k];
        if (k < wv)
            base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// The flags of a chunk (4,096 rows = one summary word) as the compaction wants them: the summary word (bit k = group k of 64
// rows holds a flag), per group its 64-bit data word, the number of rows -- and the flags cleared.  ONE WAVE = the 16 chunks of
// one offset group, in TWO trips to memory whatever the chunks hold: the sixteen lines of group flags at once, then the flagged
// groups -- a handful per wave for the sparse tables -- dealt out one per LANE (64 per round), whichever chunk they belong to.
// (A lane per group of its own chunk, four chunks a wave: 16.5 k waves, 27 us; the same with sixteen chunks a wave in four
// dependent rounds: 30 us -- stage exits: the flag lines 6 us, + the row flags 21.)  It also leaves the exclusive offsets of its
// chunks inside the group and the group's total (what the scan and the emit go by).
__global__ __launch_bounds__(64) void laia_bits_pack_kernel(LaiaBits a, LaiaFlags f, long long nsum) {
    __shared__ unsigned long long s_sw[16];
    __shared__ uint32_t s_cnt[16], s_base[17];
    const int which = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long long x0 = blockIdx.x * 16ll;
    uint8_t gf[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
        gf[c] = x0 + c < nsum ? f.g64[which][(x0 + c) * 64 + lane] : 0;
    uint32_t total = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const unsigned long long sw = __ballot(gf[c] != 0);
        if (lane == 0) {
            s_sw[c] = sw;
            s_base[c] = total;
            s_cnt[c] = 0;
            if (sw)
                a.sum[which][x0 + c] = sw;
        }
        total += static_cast<uint32_t>(__builtin_popcountll(sw));      // (wave-uniform)
    }
    if (lane == 0)
        s_base[16] = total;
    __builtin_amdgcn_s_barrier();          // (one wave: orders the LDS words)
    // flagged group t of the wave (chunks in order, groups in order) -> lane t % 64, round t / 64
    for (uint32_t t0 = 0; t0 < total; t0 += 64) {
        const uint32_t t = t0 + lane;
        int c = -1, k = 0;
        if (t < total) {
            c = 0;
#pragma unroll
            for (int q = 1; q < 16; ++q)
                c += t >= s_base[q] ? 1 : 0;
            // the (t - base)-th set bit of the chunk's summary word
            unsigned long long w = s_sw[c];
            for (uint32_t r = t - s_base[c]; r > 0; --r)
                w &= w - 1;
            k = __builtin_ctzll(w);
        }
        unsigned long long dw = 0;
        if (c >= 0) {
            const long long grp = (x0 + c) * 64 + k;
            uint4 *p = reinterpret_cast<uint4 *>(f.row[which] + grp * 64);
            uint4 rb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                rb[q] = p[q];
            // 64 flag bytes (0 / 1) -> 64 bits: four bytes of a word to a nibble by one multiply
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t v[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    dw |= static_cast<unsigned long long>((v[d] * 0x01020408u) >> 24 & 0xFu) << (16 * q + 4 * d);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                p[q] = uint4{0u, 0u, 0u, 0u};
            f.g64[which][grp] = 0;
            a.bits[which][grp] = dw;
            atomicAdd(&s_cnt[c], static_cast<uint32_t>(__builtin_popcountll(dw)));
        }
    }
    __builtin_amdgcn_s_barrier();
    const uint32_t myc = lane < 16 ? s_cnt[lane] : 0u;
    uint32_t incl = myc;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= o)
            incl += v;
    }
    const long long y = x0 + lane;
    if (lane < 16 && y < nsum) {
        a.cnt[which][y] = incl - myc;
        if (lane == 15 || y == nsum - 1)
            a.blk[which][blockIdx.x] = incl;
    }
}

// emits the rows in order (one wave per summary word, lane k its data word k), writes the per-worker offsets and
// clears the words it read
__global__ __launch_bounds__(256) void laia_bits_emit_kernel(LaiaBits a, long long nsum, long long sum_per_worker, int W,
                                                             long long Rpad) {
    // a block = the 16 summary words of one offset group, four per wave (most are empty: fewer, longer-lived waves)
    __shared__ uint32_t s_red[4];
    const int which = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the rows in front of this offset group = the totals of the groups before it, summed here (a scan launch of two
    // workgroups in between was 5 us of a 105 us global batch for 2 x 2,060 words)
    uint32_t part = 0;
    for (int b = threadIdx.x; b < static_cast<int>(blockIdx.x); b += 256)
        part += a.blk[which][b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        part += __shfl_xor(part, o, 64);
    if (lane == 0)
        s_red[wv] = part;
    __syncthreads();
    const uint32_t blk0 = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    // the wave's four summary words, their offsets and then their data words: each a batch of loads (word by word it was
    // three dependent trips to memory per summary word, four words per wave)
    const long long x0 = blockIdx.x * 16ll + wv * 4;
    uint32_t cn[4];
    unsigned long long sw4[4], dw4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long x = x0 + i < nsum ? x0 + i : nsum - 1;
        cn[i] = a.cnt[which][x];
        sw4[i] = x0 + i < nsum ? a.sum[which][x] : 0ull;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        dw4[i] = ((sw4[i] >> lane) & 1ull) ? a.bits[which][(x0 + i) * 64 + lane] : 0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long x = x0 + i;
        if (x >= nsum)
            return;
        const uint32_t at0 = blk0 + cn[i];
        if (lane == 0) {
            if (x % sum_per_worker == 0)
                a.off[which][x / sum_per_worker] = static_cast<int32_t>(at0);
            if (x == nsum - 1)
                a.off[which][W] = static_cast<int32_t>(blk0 + a.blk[which][blockIdx.x]);      // (the last group: everything)
        }
        const unsigned long long sw = sw4[i];      // wave-uniform
        if (sw == 0)
            continue;
        unsigned long long dw = dw4[i];
        if ((sw >> lane) & 1ull)
            a.bits[which][x * 64 + lane] = 0;
        if (lane == 0)
            a.sum[which][x] = 0;
        const uint32_t c = __builtin_popcountll(dw);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += y;
        }
        uint32_t at = at0 + incl - c;
        const long long w = x / sum_per_worker;
        const unsigned long long base = static_cast<unsigned long long>(x * 64 + lane) * 64ull - static_cast<unsigned long long>(w) * Rpad;
        while (dw) {
            const int b = __builtin_ctzll(dw);
            dw &= dw - 1;
            a.rows[which][at++] = static_cast<uint32_t>(base + b);
        }
    }
}

// ---- the MiniLRU update of one worker, one workgroup ---------------------------------------------------------
constexpr int kLaiaLruBlocks = 32;   // workgroups per worker in the parallel phases of the MiniLRU update

struct LaiaLru {
    uint32_t *stamp;        // [W * R]
    uint8_t *valid;         // [W * R]
    uint32_t *log_key, *log_stamp;   // [W * L]
    LruState *state;        // [W]
    const uint32_t *touch_rows, *plan_rows;
    const int32_t *tw_off, *pl_off;
    uint32_t *flag;         // [W * BT]  bit 0 resident at batch start, bit 1 valid (after the outdates), bit 2 evicted early
    uint32_t *newcnt;       // [W * (BT + 1)]  rows of the batch absent at its start before row i, within its block's chunk
    uint32_t *blocktot;     // [W * kLaiaLruBlocks]  such rows per chunk
    uint32_t *cand;         // [W * 3 * cand_cap]  (position, non-batch lines before, A(position - 1))
    long long R, L;
    int BT, cap, cand_cap;
};

__device__ __forceinline__ int laia_find(const uint32_t *__restrict__ a, int n, uint32_t key) {   // index or -1
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < n && a[lo] == key) ? lo : -1;
}

// drop the stale entries of the ring in place (order kept); renumber: restamp the live entries 1, 2, ...
__device__ void laia_log_compact(const LaiaLru &a, int w, bool renumber, uint32_t *s_w, unsigned long long *s_u64) {
    LruState &st = a.state[w];
    uint32_t *lk = a.log_key + static_cast<long long>(w) * a.L;
    uint32_t *ls = a.log_stamp + static_cast<long long>(w) * a.L;
    uint32_t *stamp = a.stamp + static_cast<long long>(w) * a.R;
    const unsigned long long mask = static_cast<unsigned long long>(a.L) - 1ull;
    const unsigned long long head = st.log_head, size = st.log_size;
    unsigned long long wr = 0;
    for (unsigned long long r0 = 0; r0 < size; r0 += 1024) {
        const unsigned long long r = r0 + threadIdx.x;
        uint32_t key = 0, sp = 0;
        bool live = false;
        if (r < size) {
            key = lk[(head + r) & mask];
            sp = ls[(head + r) & mask];
            live = stamp[key] == sp && sp != 0;
        }
        uint32_t tot;
        const uint32_t ex = laia_block_scan_u32(live ? 1u : 0u, s_w, &tot);   // barriers: reads before writes
        if (live) {
            const unsigned long long d = wr + ex;
            const uint32_t ns = renumber ? static_cast<uint32_t>(d + 1) : sp;
            lk[(head + d) & mask] = key;
            ls[(head + d) & mask] = ns;
            if (renumber)
                stamp[key] = ns;
        }
        wr += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st.log_size = wr;
        if (renumber)
            st.counter = static_cast<uint32_t>(wr);
    }
    __syncthreads();
    (void)s_u64;
}

__device__ __forceinline__ int laia_chunk(int n) {   // batch rows per workgroup of the parallel phases
    return (n + kLaiaLruBlocks - 1) / kLaiaLruBlocks;
}

// Phase 1 (W x kLaiaLruBlocks workgroups): outdate the plan keys (mini_lru_cache.h:120-128); per row of the batch:
// resident? valid after the outdates? and the number of absent rows before it inside its chunk.
__global__ __launch_bounds__(256) void laia_lru_classify_kernel(const LaiaLru a) {
    __shared__ uint32_t s_w[4];
    const int w = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
    uint32_t *stamp = a.stamp + static_cast<long long>(w) * a.R;
    uint8_t *valid = a.valid + static_cast<long long>(w) * a.R;
    const uint32_t *T = a.touch_rows + a.tw_off[w];
    const uint32_t *P = a.plan_rows + a.pl_off[w];
    const int n = a.tw_off[w + 1] - a.tw_off[w], m = a.pl_off[w + 1] - a.pl_off[w];
    uint32_t *flag = a.flag + static_cast<long long>(w) * a.BT;
    uint32_t *newcnt = a.newcnt + static_cast<long long>(w) * (a.BT + 1);
    for (int i = g * 256 + t; i < m; i += kLaiaLruBlocks * 256) {
        const uint32_t r = P[i];
        if (stamp[r] != 0 && valid[r])
            valid[r] = 0;
    }
    const int chunk = laia_chunk(n), c0 = g * chunk, c1 = min(n, c0 + chunk);
    const int per = (chunk + 255) / 256;
    uint32_t mine = 0;
    for (int q = 0; q < per; ++q) {
        const int i = c0 + t * per + q;
        if (i < c1) {
            const uint32_t r = T[i];
            const bool old = stamp[r] != 0;
            // a row that is also a plan key is outdated by another thread of this launch: decide that from the list
            const bool val = valid[r] && laia_find(P, m, r) < 0;
            flag[i] = (old ? 1u : 0u) | (val ? 2u : 0u);
            mine += old ? 0u : 1u;
        }
    }
    uint32_t tot;
    uint32_t run = laia_block_scan_u32(mine, s_w, &tot);
    for (int q = 0; q < per; ++q) {
        const int i = c0 + t * per + q;
        if (i < c1) {
            newcnt[i] = run;
            run += (flag[i] & 1u) ? 0u : 1u;
        }
    }
    if (t == 0)
        a.blocktot[w * kLaiaLruBlocks + g] = tot;
}

// Phase 2 (one workgroup per worker): the victims and the early evictions
__global__ __launch_bounds__(1024) void laia_lru_window_kernel(const LaiaLru a, uint32_t debug_wrap_at) {
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_u64[4];
    __shared__ uint32_t s_eb[1024];       // positions of the early-evicted lines (more: sticky error)
    __shared__ uint32_t s_misc[8];
    __shared__ uint32_t s_boff[kLaiaLruBlocks + 1];
    const int w = blockIdx.x, t = threadIdx.x;
    LruState &st = a.state[w];
    uint32_t *stamp = a.stamp + static_cast<long long>(w) * a.R;
    uint8_t *valid = a.valid + static_cast<long long>(w) * a.R;
    uint32_t *lk = a.log_key + static_cast<long long>(w) * a.L;
    uint32_t *ls = a.log_stamp + static_cast<long long>(w) * a.L;
    const uint32_t *T = a.touch_rows + a.tw_off[w];
    const int n = a.tw_off[w + 1] - a.tw_off[w], m = a.pl_off[w + 1] - a.pl_off[w];
    uint32_t *flag = a.flag + static_cast<long long>(w) * a.BT;
    const uint32_t *newcnt = a.newcnt + static_cast<long long>(w) * (a.BT + 1);
    uint32_t *cand = a.cand + static_cast<long long>(w) * 3 * a.cand_cap;
    const unsigned long long mask = static_cast<unsigned long long>(a.L) - 1ull;
    if (t < 8)
        s_misc[t] = 0;
    if (t == 0) {
        uint32_t run = 0;
        for (int g = 0; g < kLaiaLruBlocks; ++g) {
            s_boff[g] = run;
            run += a.blocktot[w * kLaiaLruBlocks + g];
        }
        s_boff[kLaiaLruBlocks] = run;
    }
    __syncthreads();
    const uint32_t new_total = s_boff[kLaiaLruBlocks];
    const int chunk = laia_chunk(n);

    // room in the log for n appends, and the 32-bit stamp counter
    const bool wrap = static_cast<unsigned long long>(st.counter) + static_cast<unsigned long long>(n) >=
                      static_cast<unsigned long long>(debug_wrap_at);
    if (wrap || st.log_size + static_cast<unsigned long long>(n) > static_cast<unsigned long long>(a.L))
        laia_log_compact(a, w, wrap, s_w, s_u64);

    const int live0 = st.live;
    const long long Nll = static_cast<long long>(live0) + new_total - a.cap;
    const uint32_t N = Nll > 0 ? static_cast<uint32_t>(Nll) : 0u;
    // the first N live log entries that are not rows of the batch are the victims; batch rows met on the way are
    // candidates for an early eviction
    unsigned long long consumed = 0;      // log entries passed (up to and including the N-th victim)
    uint32_t nb_base = 0, ncand = 0;
    const unsigned long long head = st.log_head, size = st.log_size;
    while (nb_base < N && consumed < size) {
        const unsigned long long r = consumed + t;
        uint32_t key = 0;
        int pos = -1;
        bool live = false;
        if (r < size) {
            key = lk[(head + r) & mask];
            live = stamp[key] == ls[(head + r) & mask];
            if (live)
                pos = laia_find(T, n, key);
        }
        const bool nonbatch = live && pos < 0;
        uint32_t tot_nb, tot_c;
        const uint32_t ex_nb = laia_block_scan_u32(nonbatch ? 1u : 0u, s_w, &tot_nb);
        const bool victim = nonbatch && nb_base + ex_nb < N;
        const bool is_cand = live && pos >= 0 && nb_base + ex_nb < N;     // before the last victim
        const uint32_t ex_c = laia_block_scan_u32(is_cand ? 1u : 0u, s_w, &tot_c);
        if (victim) {
            if (valid[key])
                atomicAdd(&s_misc[0], 1u);
            valid[key] = 0;
            stamp[key] = 0;
        }
        if (is_cand) {
            const uint32_t c = ncand + ex_c;
            if (c < static_cast<uint32_t>(a.cand_cap)) {
                cand[3 * c + 0] = static_cast<uint32_t>(pos);            // 0-based position
                cand[3 * c + 1] = nb_base + ex_nb;                         // non-batch lines before it
                cand[3 * c + 2] = s_boff[pos / chunk] + newcnt[pos];       // A(position - 1) for the 1-based position
            }
        }
        if (victim && nb_base + ex_nb + 1 == N)
            s_misc[1] = t + 1;         // the N-th victim ends the window: entries behind it stay in the log
        __syncthreads();
        if (nb_base + tot_nb >= N)
            consumed += s_misc[1];
        else
            consumed += (size - consumed < 1024ull) ? (size - consumed) : 1024ull;
        nb_base += tot_nb;
        ncand += tot_c;
        __syncthreads();
    }
    if (t == 0) {
        if (nb_base < N)
            st.err = 1;
        if (ncand > static_cast<uint32_t>(a.cand_cap))
            st.err = 2;
    }
    if (ncand > static_cast<uint32_t>(a.cand_cap))
        ncand = static_cast<uint32_t>(a.cand_cap);
    __syncthreads();
    // early evictions, in log order (one thread; the candidates are few: lines near the LRU end that this very batch names)
    if (t == 0) {
        uint32_t eb = 0, push_valid = 0;
        for (uint32_t c = 0; c < ncand; ++c) {
            const uint32_t pos = cand[3 * c], nbb = cand[3 * c + 1], A = cand[3 * c + 2];
            uint32_t ebc = 0;
            for (uint32_t k = 0; k < eb && k < 1024u; ++k)
                ebc += s_eb[k] < pos ? 1u : 0u;          // early-evicted lines with a (0-based) position < pos
            const long long lhs = static_cast<long long>(live0) + A + ebc - a.cap;
            if (lhs >= static_cast<long long>(1u + nbb + eb)) {
                if (eb < 1024u)
                    s_eb[eb] = pos;
                else
                    st.err = 2;
                eb += 1;
                if (flag[pos] & 2u)
                    push_valid += 1;
                flag[pos] |= 4u;
            }
        }
        st.apply_counter0 = st.counter;
        st.apply_tail = head + size;
        st.counter = st.counter + static_cast<uint32_t>(n);
        st.live = live0 + static_cast<int>(new_total) - static_cast<int>(nb_base < N ? nb_base : N);
        st.log_head = (head + consumed) & mask;
        st.log_size = size - consumed + static_cast<unsigned long long>(n);
        st.miss_pull += new_total + eb;
        st.miss_push += s_misc[0] + push_valid;
        st.update_push += static_cast<unsigned long long>(m);
    }
}

// Phase 3 (W x kLaiaLruBlocks workgroups): every row of the batch ends resident, valid, stamped in batch order; log appends
__global__ __launch_bounds__(256) void laia_lru_apply_kernel(const LaiaLru a) {
    __shared__ uint32_t s_w[4];
    const int w = blockIdx.y, t = threadIdx.x;
    LruState &st = a.state[w];
    uint32_t *stamp = a.stamp + static_cast<long long>(w) * a.R;
    uint8_t *valid = a.valid + static_cast<long long>(w) * a.R;
    uint32_t *lk = a.log_key + static_cast<long long>(w) * a.L;
    uint32_t *ls = a.log_stamp + static_cast<long long>(w) * a.L;
    const uint32_t *T = a.touch_rows + a.tw_off[w];
    const int n = a.tw_off[w + 1] - a.tw_off[w];
    const uint32_t *flag = a.flag + static_cast<long long>(w) * a.BT;
    const unsigned long long mask = static_cast<unsigned long long>(a.L) - 1ull;
    const uint32_t counter0 = st.apply_counter0;
    const unsigned long long tail = st.apply_tail;
    uint32_t upd = 0;
    for (int i = blockIdx.x * 256 + t; i < n; i += kLaiaLruBlocks * 256) {
        const uint32_t r = T[i];
        const uint32_t f = flag[i];
        if ((f & 1u) && !(f & 2u) && !(f & 4u))
            upd += 1;                                  // get() of a resident, outdated line: -2
        stamp[r] = counter0 + static_cast<uint32_t>(i) + 1u;
        valid[r] = 1;
        lk[(tail + i) & mask] = r;
        ls[(tail + i) & mask] = counter0 + static_cast<uint32_t>(i) + 1u;
    }
    uint32_t tot;
    (void)laia_block_scan_u32(upd, s_w, &tot);
    if (t == 0 && tot)
        atomicAdd(&st.update_pull, static_cast<unsigned long long>(tot));
}

// ---- host side of the device-resident mode ------------------------------------------------------------------------
struct LaiaDev {
    bool on = false;
    long long Rpad = 0, L = 0, nsum = 0, sum_per_worker = 0;
    int nblk = 0, cand_cap = 0;
    uint32_t debug_wrap_at = 0xFFFFFF00u;
    LaiaBits bits{};
    LaiaFlags flags{};
    LaiaLru lru{};
    long long *d_dist = nullptr;
    // what the host reads per batch sits in ONE device block -- dist [W * Bcap], offsets [2 * (W + 1)], states [W] --
    // mirrored in pinned memory by a single copy; the plan rows follow in a second one (their number is only known then)
    char *d_out = nullptr, *h_out = nullptr;
    size_t out_bytes = 0, off_at = 0, state_at = 0;
    long long *h_dist = nullptr;
    int32_t *h_off = nullptr;
    uint32_t *h_plan_rows = nullptr;
    LruState *h_state = nullptr;
    // One batch AHEAD (ha_laia_hint_next): the caller names the batch of its next call, and the call that returns batch k
    // enqueues the launches and copies of that batch before it hands k's results over -- the device works on k+1 while the
    // host (the reference's launch() loop, laia_scheduler.cc:115-169: queueing the plan and the dist) deals with k.  The
    // pinned mirrors exist twice for it.  Round 5: (1) batch k+1 is enqueued BEFORE the host waits for batch k; (2) the results
    // reach the host WITHOUT copy commands: the batch's last kernel (laia_export_kernel) writes the block and exactly the plan
    // rows the caller wants into the pinned mirror and then a sequence word the host polls.  (Two hipMemcpyAsync per batch on
    // the scheduler's stream cost ~38 us of idle stream around them -- 13 us in front of and behind each copy -- where the
    // kernels of a batch take 115; profiles/r05/laia_gaps_*.txt.)
    unsigned *d_export_ctr = nullptr;            // blocks of the export kernel that have finished
    unsigned long long *h_seq = nullptr;         // pinned: [2] the sequence number of the batch last exported into a mirror
    unsigned long long seq_issued[2] = {0, 0}, seq_next = 0;
    char *h_out_base = nullptr;
    uint32_t *h_plan_base = nullptr;
    size_t plan_words = 0;          // words per pinned plan mirror
    int set = 0;                    // the mirror the pointers above are set to
    long long next_hint = -1;       // batch the caller announced for its next call (-1: none)
    long long inflight = -1;        // batch whose launches and copies are enqueued already (-1: none)
    int inflight_set = 0, inflight_rank = -1, inflight_topk = 0;
    long long inflight_mini_bs = 0;
};
static inline void laia_host_set(LaiaDev &d, int set) {
    d.set = set;
    d.h_out = d.h_out_base + static_cast<size_t>(set) * ((d.out_bytes + 255) / 256 * 256);
    d.h_plan_rows = d.h_plan_base + static_cast<size_t>(set) * d.plan_words;
    d.h_dist = reinterpret_cast<long long *>(d.h_out);
    d.h_off = reinterpret_cast<int32_t *>(d.h_out + d.off_at);
    d.h_state = reinterpret_cast<LruState *>(d.h_out + d.state_at);
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
    // device-resident mode, the same calls by what the host does in them (ha_laia_timing_device): enqueueing a batch's launches,
    // waiting for the batch's sequence word, copying dist and plan out of the pinned mirror
    double t_issue_us = 0, t_wait_us = 0, t_unpack_us = 0;
    long long t_calls = 0;
    LaiaDev dev;              // device-resident mode (LaiaScheduler with cache_size >= max_batch * tables)
    bool host_ready = false;  // the host snapshots are initialised at the first call that needs them
    bool decided = false;     // device or host mode: fixed by the first ha_laia_next* call
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
    l.snaps.resize(l.W);      // initialised by laia_host_snaps() when the host mode is first used
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
        // (a launch on the null stream the host does not wait for; the scheduler's stream is a non-blocking one)
        ok = ok && hipMemsetAsync(l.d_valid, 0, static_cast<size_t>(l.W) * l.R, nullptr) == hipSuccess &&
             hipStreamSynchronize(nullptr) == hipSuccess;
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
    for (void *p : {static_cast<void *>(h->l.dev.h_out_base), static_cast<void *>(h->l.dev.h_plan_base)})
        if (p)
            (void)hipHostFree(p);
    if (h->l.dev.h_seq)
        (void)hipHostFree(h->l.dev.h_seq);
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

static void laia_host_snaps(Laia &l) {
    if (l.host_ready)
        return;
    // direct maps while all W of them stay below 2 GiB of host memory, hash maps beyond
    const bool use_direct = static_cast<unsigned long long>(l.W) * static_cast<unsigned long long>(l.R) * 4ull <=
                            (2ull << 30);
    for (auto &s : l.snaps)
        s.init(l.cache_size, l.R, use_direct);
    l.host_ready = true;
}

// Device-resident mode is possible when a batch can never evict its own rows (cache_size >= max_batch * tables) and the
// batch fits the assignment workgroup; HA_LAIA_HOST=1 keeps the host snapshots.
static bool laia_dev_eligible(const Laia &l) {
    const char *e = getenv("HA_LAIA_HOST");
    if (e && e[0] == '1')
        return false;
    return static_cast<long long>(l.cache_size) >= static_cast<long long>(l.Bcap) * l.T && l.Bcap <= 1024 * kLaiaAssignPer;
}

static int laia_dev_init(Laia &l) {
    LaiaDev &d = l.dev;
    const long long W = l.W, R = l.R;
    const size_t BT = static_cast<size_t>(l.Bcap) * l.T;
    d.Rpad = (R + kLaiaBitsChunk - 1) / kLaiaBitsChunk * kLaiaBitsChunk;
    d.sum_per_worker = d.Rpad / kLaiaBitsChunk;
    d.nsum = W * d.sum_per_worker;
    d.nblk = static_cast<int>((d.nsum + 15) / 16);
    long long L = 1024;
    while (L < 4ll * (static_cast<long long>(l.cache_size) + static_cast<long long>(BT) + 2))
        L <<= 1;
    if (const char *e = getenv("HA_LAIA_DEBUG_LOG")) {      // tests: a short log forces compactions
        long long v = atoll(e);
        L = 1024;
        while (L < v)
            L <<= 1;
    }
    if (const char *e = getenv("HA_LAIA_DEBUG_WRAP"))       // tests: renumber the stamps early
        d.debug_wrap_at = static_cast<uint32_t>(strtoul(e, nullptr, 10));
    HA_REQUIRE(L >= static_cast<long long>(l.cache_size) + 2 * static_cast<long long>(BT) + 2,
               "laia: the log must hold the resident lines and two batches");
    d.L = L;
    d.cand_cap = static_cast<int>(BT < 65536 ? BT : 65536);
    const size_t nwords = static_cast<size_t>(W) * d.Rpad / 64;
    bool ok = true;
    auto alloc = [&](void **p, size_t bytes, bool zero) {
        ok = ok && laia_alloc(l, p, bytes) == 0;
        if (ok && zero)
            ok = hipMemsetAsync(*p, 0, bytes, l.stream) == hipSuccess;
    };
    alloc(reinterpret_cast<void **>(&d.lru.stamp), static_cast<size_t>(W) * R * 4, true);
    alloc(reinterpret_cast<void **>(&d.lru.log_key), static_cast<size_t>(W) * L * 4, false);
    alloc(reinterpret_cast<void **>(&d.lru.log_stamp), static_cast<size_t>(W) * L * 4, false);
    alloc(reinterpret_cast<void **>(&d.lru.flag), static_cast<size_t>(W) * BT * 4, false);
    alloc(reinterpret_cast<void **>(&d.lru.newcnt), static_cast<size_t>(W) * (BT + 1) * 4, false);
    alloc(reinterpret_cast<void **>(&d.lru.cand), static_cast<size_t>(W) * 3 * d.cand_cap * 4, false);
    alloc(reinterpret_cast<void **>(&d.lru.blocktot), static_cast<size_t>(W) * kLaiaLruBlocks * 4, true);
    for (int k = 0; k < 2; ++k) {
        alloc(reinterpret_cast<void **>(&d.bits.bits[k]), nwords * 8, true);
        alloc(reinterpret_cast<void **>(&d.bits.sum[k]), static_cast<size_t>(d.nsum) * 8, true);
        alloc(reinterpret_cast<void **>(&d.bits.rows[k]), (k == 0 ? BT : l.plan_cap) * 4, false);
        alloc(reinterpret_cast<void **>(&d.bits.cnt[k]), static_cast<size_t>(d.nsum) * 4, false);
        alloc(reinterpret_cast<void **>(&d.flags.row[k]), static_cast<size_t>(W) * d.Rpad, true);
        alloc(reinterpret_cast<void **>(&d.flags.g64[k]), nwords, true);
        alloc(reinterpret_cast<void **>(&d.bits.blk[k]), static_cast<size_t>(d.nblk + 1) * 4, false);
    }
    d.off_at = static_cast<size_t>(l.Bcap) * 8;                                   // dist holds one entry per sample
    d.state_at = (d.off_at + static_cast<size_t>(2 * (W + 1)) * 4 + 15) / 16 * 16;
    d.out_bytes = d.state_at + static_cast<size_t>(W) * sizeof(LruState);
    alloc(reinterpret_cast<void **>(&d.d_out), d.out_bytes, true);
    alloc(reinterpret_cast<void **>(&d.d_export_ctr), 256, true);
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&d.h_seq), 256, hipHostMallocDefault) == hipSuccess;
    if (ok)
        d.h_seq[0] = d.h_seq[1] = 0;
    d.plan_words = (static_cast<size_t>(l.plan_cap) + 4 + 63) / 64 * 64;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&d.h_out_base), 2 * ((d.out_bytes + 255) / 256 * 256),
                             hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&d.h_plan_base), 2 * d.plan_words * 4, hipHostMallocDefault) == hipSuccess;
    if (ok) {
        d.d_dist = reinterpret_cast<long long *>(d.d_out);
        d.bits.off[0] = reinterpret_cast<int32_t *>(d.d_out + d.off_at);
        d.bits.off[1] = d.bits.off[0] + (W + 1);
        d.lru.state = reinterpret_cast<LruState *>(d.d_out + d.state_at);
        laia_host_set(d, 0);
    }
    HA_REQUIRE(ok, "laia: device allocation of the resident scheduler state failed");
    d.lru.valid = l.d_valid;
    d.lru.touch_rows = d.bits.rows[0];
    d.lru.plan_rows = d.bits.rows[1];
    d.lru.tw_off = d.bits.off[0];
    d.lru.pl_off = d.bits.off[1];
    d.lru.R = R;
    d.lru.L = L;
    d.lru.BT = static_cast<int>(BT);
    d.lru.cap = l.cache_size;
    d.lru.cand_cap = d.cand_cap;
    HA_CHECK_HIP(hipStreamSynchronize(l.stream));
    d.on = true;
    return 0;
}

// The last kernel of a batch: the block the host reads (dist, offsets, states: `out_words` words at d_out) and the plan rows
// of the worker the caller asked for (only_rank >= 0) or of all workers go straight into the pinned mirror -- plain stores to
// host memory, drained by every wave --, and the block that finishes last publishes the batch's sequence number behind a
// system-scope fence.  The host polls that word: no copy command, no event, nothing between this kernel and the next batch's
// first one on the stream.
__global__ __launch_bounds__(256) void laia_export_kernel(const uint32_t *__restrict__ d_out, long long out_words,
                                                          const uint32_t *__restrict__ rows, const int32_t *__restrict__ off,
                                                          int W, int only_rank, long long cap, uint32_t *__restrict__ h_out,
                                                          uint32_t *__restrict__ h_plan, unsigned *__restrict__ ctr,
                                                          unsigned long long *__restrict__ h_seq, unsigned long long seq) {
    const long long stride = static_cast<long long>(gridDim.x) * 256, t0 = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    for (long long i = t0; i < out_words; i += stride)
        h_out[i] = d_out[i];
    long long a0 = only_rank >= 0 ? off[only_rank] : 0, a1 = only_rank >= 0 ? off[only_rank + 1] : off[W];
    a1 = a1 < cap ? a1 : cap;
    for (long long i = a0 + t0; i < a1; i += stride)
        h_plan[i] = rows[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static int laia_dev_states(Laia &l) {   // -> l.dev.h_state (synchronises the scheduler's stream)
    HA_CHECK_HIP(hipMemcpyAsync(l.dev.h_out, l.dev.d_out, l.dev.out_bytes, hipMemcpyDeviceToHost, l.stream));
    HA_CHECK_HIP(hipStreamSynchronize(l.stream));
    return 0;
}

// One global batch with the scheduler state on the device, in two halves.  ISSUE: nine launches, then `dist` + offsets +
// states to the pinned mirror `set` in one copy and -- speculatively -- the plan rows in a second (their number is only known
// afterwards).  COLLECT (laia_next_device): wait, fetch what the speculation missed, hand the results over.  A call issues its
// own batch unless the call before it did (ha_laia_hint_next).
static int laia_dev_issue(ha_laia *h, int64_t batch_id, int64_t mini_bs, int only_rank, const TopkParams *topk, int set) {
    Laia &l = h->l;
    LaiaDev &d = l.dev;
    const int W = l.W, T = l.T;
    const long long B = mini_bs * W;
    HA_REQUIRE(B <= l.Bcap, "laia_next: global batch %lld exceeds max_batch %d", B, l.Bcap);
    laia_host_set(d, set);
    const long long start = (batch_id * B) % l.S;  // laia_scheduler.cc:182
    const long long BT = B * T;
    int blocks = static_cast<int>((BT + 255) / 256);
    if (blocks > 4096)
        blocks = 4096;
    if (topk) {
        HA_REQUIRE(topk->top_k >= 1 && topk->top_k <= T && T <= 64 && topk->num_threads >= 1 && W <= 64,
                   "laia_next_topk: need 1 <= top_k <= num_table <= 64, num_threads >= 1 and at most 64 workers");
        const int nt = topk->num_threads;
        for (int t = 0; t < nt; ++t) {
            long long s0, s1, q0, q1;
            topk_slice(B, nt, t, &s0, &s1);
            topk_slice(mini_bs, nt, t, &q0, &q1);
            HA_REQUIRE(s1 - s0 <= W * (q1 - q0),
                       "laia_next_topk: thread %d has %lld samples for %d workers x quota %lld (the reference "
                       "writes dist[-1] here); pick num_threads dividing mini_batch_size", t, s1 - s0, W, q1 - q0);
        }
        TableOrder order;
        for (int k = 0; k < topk->top_k; ++k) {
            HA_REQUIRE(topk->order[k] >= 0 && topk->order[k] < T, "laia_next_topk: table index out of range");
            order.t[k] = topk->order[k];
        }
        hipLaunchKernelGGL(laia_probe_kernel, dim3(blocks), dim3(256), 0, l.stream, l.d_samples, l.S, T, start, (int)B, W,
                           l.d_valid, l.R, l.d_mask);
        hipLaunchKernelGGL(laia_topk_score_kernel, dim3((int)((B + 255) / 256)), dim3(256), 0, l.stream, l.d_mask, (int)B, T, W,
                           order, topk->top_k, l.d_scores, l.d_cand);
        HA_CHECK_HIP(hipMemsetAsync(d.d_dist, 0, static_cast<size_t>(W) * mini_bs * sizeof(long long), l.stream));   // dist.reset(0), :382
        hipLaunchKernelGGL(laia_topk_assign_kernel, dim3((nt + 127) / 128), dim3(128), 0, l.stream, l.d_scores, l.d_cand, (int)B,
                           W, (int)mini_bs, nt, start, l.S, l.d_owner, d.d_dist);
    } else if (T <= 64) {
        const long long waves = (B + (64 / T) - 1) / (64 / T);
        hipLaunchKernelGGL(laia_probe_score_kernel, dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0, l.stream,
                           l.d_samples, l.S, T, start, (int)B, W, l.d_valid, l.R, l.d_mask, l.d_scores);
    } else {
        hipLaunchKernelGGL(laia_probe_kernel, dim3(blocks), dim3(256), 0, l.stream, l.d_samples, l.S, T, start, (int)B, W,
                           l.d_valid, l.R, l.d_mask);
        hipLaunchKernelGGL(laia_score_kernel, dim3((int)((B * W + 255) / 256)), dim3(256), 0, l.stream, l.d_mask, (int)B, T,
                           W, l.d_scores);
    }
    const size_t sc_bytes = static_cast<size_t>(B) * W;
    const int lds_scores = sc_bytes <= (size_t(32) << 10) && T <= 255 ? 1 : 0;
    if (!topk)
        hipLaunchKernelGGL(laia_assign_kernel, dim3(1), dim3(1024), lds_scores ? sc_bytes : 0, l.stream, l.d_scores, (int)B, W,
                           (int)mini_bs, (long long)batch_id, start, l.S, l.d_owner, d.d_dist, lds_scores);
    hipLaunchKernelGGL(laia_bits_kernel, dim3(blocks), dim3(256), 0, l.stream, l.d_samples, l.S, T, start, (int)B, W,
                       l.d_mask, l.d_owner, l.R, d.Rpad, d.flags, topk ? 1 : 0);
    hipLaunchKernelGGL(laia_bits_pack_kernel, dim3(d.nblk, 2), dim3(64), 0, l.stream, d.bits, d.flags, d.nsum);
    hipLaunchKernelGGL(laia_bits_emit_kernel, dim3(d.nblk, 2), dim3(256), 0, l.stream, d.bits, d.nsum, d.sum_per_worker, W,
                       d.Rpad);
    hipLaunchKernelGGL(laia_lru_classify_kernel, dim3(kLaiaLruBlocks, W), dim3(256), 0, l.stream, d.lru);
    hipLaunchKernelGGL(laia_lru_window_kernel, dim3(W), dim3(1024), 0, l.stream, d.lru, d.debug_wrap_at);
    hipLaunchKernelGGL(laia_lru_apply_kernel, dim3(kLaiaLruBlocks, W), dim3(256), 0, l.stream, d.lru);
    d.seq_issued[set] = ++d.seq_next;
    hipLaunchKernelGGL(laia_export_kernel, dim3(32), dim3(256), 0, l.stream, reinterpret_cast<const uint32_t *>(d.d_out),
                       static_cast<long long>(d.out_bytes / 4), d.bits.rows[1], d.bits.off[1], W, only_rank,
                       static_cast<long long>(l.plan_cap), reinterpret_cast<uint32_t *>(d.h_out), d.h_plan_rows, d.d_export_ctr,
                       d.h_seq + set, d.seq_issued[set]);
    HA_LAUNCH_CHECK();
    return 0;
}

static int laia_next_device(ha_laia *h, int64_t batch_id, int64_t mini_bs, int64_t *dist_out, uint64_t *plan_out,
                            int64_t plan_cap_elems, int64_t *plan_off, int only_rank, const TopkParams *topk = nullptr) {
    Laia &l = h->l;
    LaiaDev &d = l.dev;
    const int W = l.W;
    const long long B = mini_bs * W;
    const double t_begin = now_us();
    int set = d.set ^ 1;
    if (d.inflight >= 0) {
        // the call before this one enqueued a batch on the caller's word: it has to be this one (the device state is past it)
        HA_REQUIRE(d.inflight == batch_id && d.inflight_mini_bs == mini_bs && d.inflight_rank == only_rank &&
                       d.inflight_topk == (topk ? 1 : 0),
                   "laia_next: batch %lld was announced (ha_laia_hint_next) and is enqueued; this call asks for batch %lld",
                   d.inflight, (long long)batch_id);
        set = d.inflight_set;
        d.inflight = -1;
    } else if (laia_dev_issue(h, batch_id, mini_bs, only_rank, topk, set)) {
        return -1;
    }
    const double t_issue0 = now_us();
    if (d.next_hint >= 0) {     // the caller's next batch: its launches and copies go out BEFORE this one is waited for
        const long long nb = d.next_hint;
        d.next_hint = -1;
        if (laia_dev_issue(h, nb, mini_bs, only_rank, topk, set ^ 1))
            return -1;
        d.inflight = nb;
        d.inflight_set = set ^ 1;
        d.inflight_mini_bs = mini_bs;
        d.inflight_rank = only_rank;
        d.inflight_topk = topk ? 1 : 0;
    }
    laia_host_set(d, set);
    const double t_wait_begin = now_us();
    l.t_issue_us += t_wait_begin - t_issue0;
    {   // the batch's sequence word (written by its export kernel); a stream that died shows up in the synchronise
        const unsigned long long want = d.seq_issued[set];
        volatile unsigned long long *word = d.h_seq + set;
        const double t_wait0 = now_us();
        unsigned spins = 0;
        while (__atomic_load_n(const_cast<unsigned long long *>(word), __ATOMIC_ACQUIRE) != want) {
            if ((++spins & 1023u) == 0 && now_us() - t_wait0 > 5e6) {
                HA_CHECK_HIP(hipStreamSynchronize(l.stream));
                HA_REQUIRE(__atomic_load_n(const_cast<unsigned long long *>(word), __ATOMIC_ACQUIRE) == want,
                           "laia: the device never published batch %lld", (long long)batch_id);
                break;
            }
            __builtin_ia32_pause();
        }
    }
    l.t_wait_us += now_us() - t_wait_begin;
    for (int w = 0; w < W; ++w)
        HA_REQUIRE(d.h_state[w].err == 0, "laia: the device snapshot of worker %d is inconsistent (code %d)", w,
                   d.h_state[w].err);
    const int32_t *pl_off = d.h_off + (W + 1);
    const long long nplan = pl_off[W];
    const long long a0 = only_rank >= 0 ? pl_off[only_rank] : 0, a1 = only_rank >= 0 ? pl_off[only_rank + 1] : nplan;
    HA_REQUIRE(a1 - a0 <= plan_cap_elems && nplan <= static_cast<long long>(l.plan_cap), "laia_next: plan buffer too small");
    const double t_out0 = now_us();
    for (long long k = 0; k < B; ++k)
        dist_out[k] = d.h_dist[k];
    if (only_rank >= 0) {       // only that worker's plan, at the front: offsets 0 .. count around it
        for (int w = 0; w <= W; ++w)
            plan_off[w] = w <= only_rank ? 0 : a1 - a0;
    } else {
        for (int w = 0; w <= W; ++w)
            plan_off[w] = pl_off[w];
    }
    for (long long k = a0; k < a1; ++k)
        plan_out[k - a0] = d.h_plan_rows[k];
    l.t_assign_us += now_us() - t_out0;      // device mode: the host's share is copying dist and the plans out
    l.t_unpack_us += now_us() - t_out0;
    l.t_total_us += now_us() - t_begin;
    l.t_calls += 1;
    return 0;
}

static int laia_next_impl(ha_laia *h, int64_t batch_id, int64_t mini_bs, int64_t *dist_out,
                          uint64_t *plan_out, int64_t plan_cap_elems, int64_t *plan_off,
                          const TopkParams *topk, int only_rank = -1) {
    HA_REQUIRE(h && dist_out && plan_out && plan_off && mini_bs > 0, "laia_next: bad arguments");
    Laia &l = h->l;
    if (!l.decided) {      // a cache that holds at least one global batch of rows: the snapshots live on the device (both
        l.decided = true;  // schedulers); smaller caches and HA_LAIA_HOST=1 keep them on the host
        if (laia_dev_eligible(l)) {
            if (laia_dev_init(l))
                return -1;
        }
    }
    if (l.dev.on)
        return laia_next_device(h, batch_id, mini_bs, dist_out, plan_out, plan_cap_elems, plan_off, only_rank, topk);
    laia_host_snaps(l);
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

// 1 = the scheduler keeps its MiniLRU snapshots, assignment and key lists on the device (decided at its first batch: a
// cache that holds at least one global batch of rows, both schedulers), 0 = host snapshots, -1 = not decided yet
extern "C" int ha_laia_on_device(ha_laia *h) {
    if (!h)
        return -1;
    return h->l.decided ? (h->l.dev.on ? 1 : 0) : -1;
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

// device-resident mode: out[4] = {calls, us enqueueing the next batch's launches, us waiting for the batch's results, us copying
// dist and plan out of the pinned mirror}, summed since creation (zeros while the snapshots live on the host)
extern "C" int ha_laia_timing_device(ha_laia *h, double *out) {
    HA_REQUIRE(h && out, "ha_laia_timing_device: null pointer");
    out[0] = static_cast<double>(h->l.t_calls);
    out[1] = h->l.t_issue_us;
    out[2] = h->l.t_wait_us;
    out[3] = h->l.t_unpack_us;
    return 0;
}

// The caller's word that its NEXT ha_laia_next* call will ask for batch `next_batch_id` with the arguments of the call that
// follows this hint (the reference's launch() walks the batches in order, laia_scheduler.cc:115-169): that call then enqueues
// the announced batch's launches and copies before it returns its own results, and the device works on it while the host
// queues plan and dist.  The scheduler's state on the device is one batch ahead from then on -- the announced call MUST
// follow (anything else is an error), and counters / snapshots read in between include the announced batch.  Ignored while the
// snapshots live on the host (small caches, HA_LAIA_HOST=1).  next_batch_id < 0 withdraws a hint that was not used yet.
extern "C" int ha_laia_hint_next(ha_laia *h, int64_t next_batch_id) {
    HA_REQUIRE(h, "laia_hint_next: null handle");
    h->l.dev.next_hint = next_batch_id >= 0 ? next_batch_id : -1;
    return 0;
}

extern "C" int ha_laia_next(ha_laia *h, int64_t batch_id, int64_t mini_bs, int64_t *dist_out,
                            uint64_t *plan_out, int64_t plan_cap_elems, int64_t *plan_off) {
    return laia_next_impl(h, batch_id, mini_bs, dist_out, plan_out, plan_cap_elems, plan_off, nullptr);
}

// ha_laia_next when the caller wants the plan of ONE worker only (what LaiaScheduler::launch queues for its rank,
// laia_scheduler.cc:140-168): plan_out = that worker's plan, plan_off[w] = 0 up to `rank` and its length behind it; dist_out
// as ha_laia_next.  With the scheduler state on the device only that worker's plan rows cross PCIe.
extern "C" int ha_laia_next_for_rank(ha_laia *h, int64_t batch_id, int64_t mini_bs, int64_t rank, int64_t *dist_out,
                                     uint64_t *plan_out, int64_t plan_cap_elems, int64_t *plan_off) {
    HA_REQUIRE(h && rank >= 0 && rank < h->l.W, "laia_next_for_rank: bad rank");
    const int W = h->l.W;
    const bool dev_before = h->l.dev.on || (!h->l.decided && laia_dev_eligible(h->l));
    if (laia_next_impl(h, batch_id, mini_bs, dist_out, plan_out, plan_cap_elems, plan_off, nullptr, static_cast<int>(rank)))
        return -1;
    if (!dev_before) {      // host mode returned all plans: keep the rank's
        const int64_t a0 = plan_off[rank], a1 = plan_off[rank + 1];
        for (int64_t k = a0; k < a1; ++k)
            plan_out[k - a0] = plan_out[k];
        for (int w = 0; w <= W; ++w)
            plan_off[w] = w <= rank ? 0 : a1 - a0;
    }
    return 0;
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
    Laia &l = h->l;
    if (l.dev.on) {
        if (laia_dev_states(l))
            return -1;
        for (int w = 0; w < l.W; ++w) {
            out[w] = static_cast<int64_t>(l.dev.h_state[w].miss_pull);
            out[l.W + w] = static_cast<int64_t>(l.dev.h_state[w].miss_push);
            out[2 * l.W + w] = static_cast<int64_t>(l.dev.h_state[w].update_pull);
            out[3 * l.W + w] = static_cast<int64_t>(l.dev.h_state[w].update_push);
        }
        return 0;
    }
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
    if (h->l.dev.on) {     // the valid bytes of worker w ARE its valid resident keys
        Laia &l = h->l;
        std::vector<uint8_t> v(static_cast<size_t>(l.R));
        if (hipStreamSynchronize(l.stream) != hipSuccess ||
            hipMemcpy(v.data(), l.d_valid + static_cast<size_t>(w) * l.R, v.size(), hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        int64_t cnt = 0;
        for (size_t k = 0; k < v.size(); ++k)
            if (v[k]) {
                if (cnt < cap)
                    out[cnt] = static_cast<int32_t>(k);
                ++cnt;
            }
        return cnt;
    }
    if (!h->l.host_ready)
        return 0;
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
