// The PLANNED flow of the HET embedding cache: LRU / LFU / LFUOpt, local store, the ids of a block of batches known a block early.
//
// Reference flows (src/hetu_cache/src/cache.cc): _embeddingLookup :60-107 and _embeddingUpdate :132-197 of the SAME keys,
// batch after batch -- what a training loop does (python/hetu/cstable.py:38-56: embedding_lookup, the model, embedding_update).
// Policies: LRUCache (src/hetu_cache/src/lru_cache.cc:5-39), LFUCache (src/lfu_cache.cc:9-70), LFUOptCache
// (src/lfuopt_cache.cc:9-71; their bookkeeping: cache_book_lfu_kernel below).  Line::accumulate (include/embedding.h:78-91), Line::addup (:92-96),
// the server's handlers (ps-lite/src/PSFhandle_embedding.cc:5-64).  Results: those of ha_cache_lookup + ha_cache_update_same_keys
// call by call (rows, versions, update counters, resident set, server table and versions; tests/test_gpu_cache_planned.py holds
// both to oracle/cache_model.py).
//
// What is different is WHEN the bookkeeping happens.  Which lines a batch hits, which it misses, which slots the misses get,
// which lines are evicted for them, how many updates a line has collected and whether it is pushed -- all of it follows from the
// IDS alone (the stamps of an LRU list, update counters, a free-slot stack), none of it from a row.  So, as the work-queue step
// does for the headline (csrc/qstep.hip), the bookkeeping of a BLOCK of up to 16 batches runs ahead, on a side stream, beside the
// rows of the block before:
//
//   ha_cache_plan_block     side stream: the index plans of the block's batches (two launches: stable sort, finish) and ONE
//                           bookkeeping launch, cache_book_block_kernel: 32 workgroups walk the batches in order and exchange
//                           their counts inside the launch (two to three exchanges per batch); per batch they leave ITEMS:
//                           per unique key {slot, miss?, update count after the batch, push?}, per evicted dirty line
//                           {slot, key, update count}, and a record of counts (the perf dict's numbers).
//   ha_cache_lookup_planned ONE launch, a wave per sorted position (cache_lookup_planned_kernel): the staleness-bounded pull
//                           decision (cache.cc:84-93: version -1 or lagging by more than pull_bound -- taken HERE, from the
//                           versions as they are when the rows are read: other workers may have pushed meanwhile), row to dest,
//                           refreshed line + Line::addup for pulled lines.
//   ha_cache_update_planned ONE launch (cache_update_planned_kernel): the ordered accumulate into gradient buffer and data row
//                           (scatter_dev.h's bit-exact chains), the pushed lines' server side FUSED into the wave that holds the
//                           line's new gradient (store row += grad, grad = 0), a wave per evicted dirty line (store row += its
//                           gradient), a thread per line for versions.
//
// Field ownership while a block is planned: the bookkeeping launch owns slot_of, a line's key / state / stamp / updates, the
// stamp log, the free stack and the control block (all accessed device-coherently inside the launch: `sc1` loads / stores,
// MI355X_MICROARCH.md "inter-workgroup visibility"); the row launches own data, grad, hasgrad, a line's version, the store's
// rows and versions, and read nothing the bookkeeping writes except the items.  The call-by-call entry points are refused
// until the planned batches are consumed.
#include "cache_dev.h"
#include "scatter_dev.h"

extern "C" int ha_plan_build_batch_f32ids_lim(const float *const *ids, const int64_t *n, void *const *ws, int count,
                                              uint64_t key_limit, ha_stream_t stream);
extern "C" int ha_plan_build_batch_u64ids_lim(const uint64_t *const *ids, const int64_t *n, void *const *ws, int count,
                                              uint64_t key_limit, ha_stream_t stream);

namespace ha {

constexpr int kBookWg = 32;          // workgroups of the bookkeeping launch (all resident: 32 x 256 threads)
constexpr int kBookThreads = 256;
constexpr int kBookKeysPerThread = (kSmallMax + kBookWg * kBookThreads - 1) / (kBookWg * kBookThreads);   // 5
constexpr long long kVerKeep = -2;   // pver: the lookup did not pull this line

// ---- device-coherent accessors (agent scope = `sc1`: the load bypasses the CU's L1, the store is written through the L2) ----
template <typename T>
__device__ __forceinline__ T ldc(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void stc(T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a line record as four 8-byte words: {stamp, version, key | updates << 32, freq | state << 32}
__device__ __forceinline__ unsigned long long *line_word(LineMeta *line, long long s, int w) {
    return reinterpret_cast<unsigned long long *>(line + s) + w;
}
static_assert(offsetof(LineMeta, stamp) == 0 && offsetof(LineMeta, version) == 8 && offsetof(LineMeta, key) == 16 &&
              offsetof(LineMeta, updates) == 20 && offsetof(LineMeta, freq) == 24 && offsetof(LineMeta, state) == 28,
              "the bookkeeping launch addresses a line record by 8-byte words");

struct BookArgs {
    int count;
    int n[kPlanBlockMax];
    const PlanHeader *hdr[kPlanBlockMax];
    const uint32_t *uniq[kPlanBlockMax];
    const int32_t *counts[kPlanBlockMax];
    int32_t *it_slot;        // [count][nmax] slot of unique key u (-1: a key beyond the cache's key range)
    uint8_t *it_flag;        // [count][nmax] kPosMiss | kPosInit (the line has a gradient buffer) | kPosPush
    int32_t *it_upd;         // [count][nmax] update counter of the line after this batch (what a push carries)
    int32_t *ev_slot;        // [count][nmax] evicted dirty lines of the batch's lookup, pushed by its update
    uint32_t *ev_key;
    int32_t *ev_upd;
    PlanRec *rec;            // [count]
    unsigned long long *xw;  // [4][kBookWg] exchange words
    long long nmax;
};
// a line record's word 3 = freq | state << 32 | hg << 40
__device__ __forceinline__ unsigned long long line_w3(uint8_t state, bool hg) {
    return (static_cast<unsigned long long>(state) << 32) | (hg ? 1ull << 40 : 0ull);
}

// One exchange between the workgroups of the bookkeeping launch: every workgroup publishes (a, b) -- both < 2^20 -- and
// learns everybody's.  Word = seq << 40 | b << 20 | a; four word arrays in turn (a workgroup is at most one exchange ahead
// of the slowest one, so a word is never overwritten before everybody has read it).  Every wave drains its stores first:
// what it stored (`sc1`) is in memory before its workgroup's word is.  Returns false after ~2 s without an answer (the
// sticky word of the control block is set; the caller leaves the launch).
__device__ __forceinline__ bool book_exchange(CacheCtl *ctl, unsigned long long *xw, unsigned long long seq, uint32_t a,
                                              uint32_t b, uint32_t *s_a, uint32_t *s_b, int *s_abort) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long *words = xw + (seq & 3ull) * kBookWg;
    const int tid = threadIdx.x;
    if (tid == 0)
        stc(words + blockIdx.x, ((seq & 0xFFFFFFull) << 40) | (static_cast<unsigned long long>(b) << 20) | a);
    if (tid < kBookWg) {
        unsigned long long w = ldc(words + tid);
        if ((w >> 40) != (seq & 0xFFFFFFull)) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
            do {
                __builtin_amdgcn_s_sleep(1);
                w = ldc(words + tid);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                    *s_abort = 1;
                    ctl->fb_timeout = 1;
                    break;
                }
            } while ((w >> 40) != (seq & 0xFFFFFFull));
        }
        s_a[tid] = static_cast<uint32_t>(w & 0xFFFFFull);
        s_b[tid] = static_cast<uint32_t>((w >> 20) & 0xFFFFFull);
    }
    __syncthreads();
    return *s_abort == 0;
}

// rank of a set flag among the set flags of the workgroup's lower threads + total (256 threads = 4 waves)
__device__ __forceinline__ uint32_t book_rank(bool f, uint32_t *s_w4, uint32_t *total) {
    const unsigned long long m = __ballot(f);
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t below = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0)
        s_w4[w] = __builtin_popcountll(m);
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kBookThreads / 64; ++k) {
        const uint32_t c = s_w4[k];
        off += k < w ? c : 0u;
        tot += c;
    }
    *total = tot;
    return off + below;
}

// ---- the bookkeeping of a block of batches ---------------------------------------------------------------------------------
// Per batch i (its unique keys u = 0 .. U-1 in key order, as `Unique<T>` hands them to batchedLookup, cache.cc:15-26,66-68):
//   every line of the batch ends the pair lookup + update as the cache's newest, in key order: stamp = clock + u, log entry
//   tail + u (lookup and update both touch every line in key order, lru_cache.cc:27-39; what the lookup's touch leaves is
//   overwritten by the update's before anything reads it: the eviction in between takes lines OUTSIDE the batch, limit >= batch);
//   misses take slots from the free stack (by miss rank), become resident;
//   LRUCache::insert (lru_cache.cc:9-25) evicts while size > limit: the first size + M - limit VALID entries from the log head
//   (valid: the line is resident and still carries the entry's stamp); their slots go back on the stack, the dirty ones
//   (updates != 0) are listed for the batch's update to push (cache.cc:160-166: the pending evictions join every push);
//   updates += occurrences; a line whose counter exceeds push_bound is pushed and starts again at 0 (cache.cc:159,171-177).
__global__ __launch_bounds__(kBookThreads) void cache_book_block_kernel(Cache c, BookArgs a) {
    __shared__ uint32_t s_a[kBookWg], s_b[kBookWg], s_w4[kBookThreads / 64];
    __shared__ int s_abort;
    CacheCtl *ctl = c.ctl;
    const int tid = threadIdx.x, g = blockIdx.x;
    if (tid == 0)
        s_abort = 0;
    // the control block as the last launch that owned it left it (every workgroup carries it forward by itself: what
    // changes it are the exchanged counts)
    long long clock = ctl->clock, tail = ctl->log_tail, head = ctl->log_head, ftop = ctl->free_top, size = ctl->size;
    unsigned long long seq = static_cast<unsigned long long>(ctl->book_seq);
    __syncthreads();
    for (int i = 0; i < a.count; ++i) {
        const int n = a.n[i];
        const long long at = static_cast<long long>(i) * a.nmax;
        if (n == 0) {
            if (g == 0 && tid == 0) {
                PlanRec r{};
                r.size = size;
                r.full = size == c.limit;
                r.vh_slot = -1;
                a.rec[i] = r;
            }
            continue;
        }
        const int U = static_cast<int>(a.hdr[i]->n_unique);
        const uint32_t *uniq = a.uniq[i];
        const int32_t *counts = a.counts[i];
        // ---- log nearly full: compact it first (valid entries keep their order), a tile of kBookWg x 256 entries at a time
        if (tail - head > c.Lcap - 4 * c.nmax - 2048 - kBookWg * kBookThreads) {
            long long wr = head;
            for (long long t0 = head; t0 < tail; t0 += kBookWg * kBookThreads) {
                const long long e = t0 + g * kBookThreads + tid;
                int ls = -1;
                unsigned long long lst = 0;
                bool valid = false;
                if (e < tail) {
                    ls = static_cast<int>(ldc(c.log_slot + e % c.Lcap));
                    lst = ldc(c.log_stamp + e % c.Lcap);
                    const unsigned long long st = ldc(line_word(c.line, ls, 0));
                    const unsigned long long w3 = ldc(line_word(c.line, ls, 3));
                    valid = static_cast<uint8_t>(w3 >> 32) == kResident && st == lst;
                }
                uint32_t tot;
                const uint32_t r = book_rank(valid, s_w4, &tot);
                if (!book_exchange(ctl, a.xw, ++seq, tot, 0, s_a, s_b, &s_abort))
                    return;
                uint32_t before = 0, all = 0;
                for (int k = 0; k < kBookWg; ++k) {
                    before += k < g ? s_a[k] : 0u;
                    all += s_a[k];
                }
                if (valid) {      // (positions below this tile's first entry: read by everybody before the exchange)
                    stc(c.log_slot + (wr + before + r) % c.Lcap, static_cast<uint32_t>(ls));
                    stc(c.log_stamp + (wr + before + r) % c.Lcap, lst);
                }
                wr += all;
            }
            tail = wr;
            if (!book_exchange(ctl, a.xw, ++seq, 0, 0, s_a, s_b, &s_abort))
                return;
        }
        // ---- phase 1: probe; hits are touched and counted at once -----------------------------------------------------------
        const int per = (U + kBookWg - 1) / kBookWg;         // this workgroup's keys: [u0, u1)
        const int u0 = min(g * per, U), u1 = min(u0 + per, U);
        int sl[kBookKeysPerThread];
        uint32_t kk[kBookKeysPerThread], rk[kBookKeysPerThread];
        bool miss[kBookKeysPerThread];
        uint32_t wg_miss = 0;
#pragma unroll
        for (int j = 0; j < kBookKeysPerThread; ++j) {
            const int u = u0 + j * kBookThreads + tid;
            const bool on = u < u1;
            kk[j] = on ? uniq[u] : 0u;
            const bool known = on && kk[j] < static_cast<unsigned long long>(c.length);
            sl[j] = known ? ldc(c.slot_of + kk[j]) : -1;
            miss[j] = known && sl[j] < 0;
            if (on && !known) {        // a key the cache has no line for: zeros on lookup, ignored by the update
                a.it_slot[at + u] = -1;
                a.it_flag[at + u] = 0;
                a.it_upd[at + u] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < kBookKeysPerThread; ++j) {
            const int u = u0 + j * kBookThreads + tid;
            if (u < u1 && sl[j] >= 0) {
                const int s = sl[j];
                const unsigned long long w2 = ldc(line_word(c.line, s, 2));
                const unsigned long long w3 = ldc(line_word(c.line, s, 3));
                const bool hg = ((w3 >> 40) & 1ull) != 0ull;
                const int upd = static_cast<int>(w2 >> 32) + counts[u];
                const bool push = upd > c.push_bound;
                const unsigned long long st = static_cast<unsigned long long>(clock + u);
                stc(line_word(c.line, s, 0), st);
                stc(line_word(c.line, s, 2), static_cast<unsigned long long>(kk[j]) |
                                                 (static_cast<unsigned long long>(static_cast<uint32_t>(push ? 0 : upd)) << 32));
                const long long pos = (tail + u) % c.Lcap;
                stc(c.log_slot + pos, static_cast<uint32_t>(s));
                stc(c.log_stamp + pos, st);
                if (!hg)           // the batch's update gives the line its gradient buffer (Line::_maybeInitGrad)
                    stc(line_word(c.line, s, 3), line_w3(kResident, true));
                a.it_slot[at + u] = s;
                a.it_flag[at + u] = static_cast<uint8_t>((hg ? kPosInit : 0) | (push ? kPosPush : 0));
                a.it_upd[at + u] = upd;
            }
            uint32_t tot;
            rk[j] = wg_miss + book_rank(miss[j], s_w4, &tot);
            wg_miss += tot;
        }
        if (!book_exchange(ctl, a.xw, ++seq, wg_miss, 0, s_a, s_b, &s_abort))
            return;
        uint32_t mb = 0, M = 0;
        for (int k = 0; k < kBookWg; ++k) {
            mb += k < g ? s_a[k] : 0u;
            M += s_a[k];
        }
        // ---- phase 2: the misses become lines -------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < kBookKeysPerThread; ++j) {
            const int u = u0 + j * kBookThreads + tid;
            if (u < u1 && miss[j]) {
                const long long fi = ftop - 1 - static_cast<long long>(mb + rk[j]);
                const int s = fi >= 0 ? ldc(c.free_list + fi) : 0;     // (running out of slots: sizing, checked on the host)
                const int upd = counts[u];
                const bool push = upd > c.push_bound;
                const unsigned long long st = static_cast<unsigned long long>(clock + u);
                stc(line_word(c.line, s, 0), st);
                stc(line_word(c.line, s, 2), static_cast<unsigned long long>(kk[j]) |
                                                 (static_cast<unsigned long long>(static_cast<uint32_t>(push ? 0 : upd)) << 32));
                stc(line_word(c.line, s, 3), line_w3(kResident, true));
                stc(c.slot_of + kk[j], s);
                const long long pos = (tail + u) % c.Lcap;
                stc(c.log_slot + pos, static_cast<uint32_t>(s));
                stc(c.log_stamp + pos, st);
                a.it_slot[at + u] = s;
                a.it_flag[at + u] = static_cast<uint8_t>(kPosMiss | (push ? kPosPush : 0));
                a.it_upd[at + u] = upd;
            }
        }
        // ---- phase 3: LRUCache::insert's evictions --------------------------------------------------------------------------
        long long need = size + M > c.limit ? size + M - c.limit : 0;
        const long long need0 = need;
        uint32_t taken_before = 0, dirty_before = 0;      // victims / dirty victims of the rounds before this one
        long long new_head = head;
        while (need > 0 && new_head < tail) {
            const long long e = new_head + g * kBookThreads + tid;
            int ls = -1;
            unsigned long long lst = 0, w2 = 0;
            bool valid = false;
            if (e < tail) {
                ls = static_cast<int>(ldc(c.log_slot + e % c.Lcap));
                lst = ldc(c.log_stamp + e % c.Lcap);
                const unsigned long long st = ldc(line_word(c.line, ls, 0));
                w2 = ldc(line_word(c.line, ls, 2));
                const unsigned long long w3 = ldc(line_word(c.line, ls, 3));
                valid = static_cast<uint8_t>(w3 >> 32) == kResident && st == lst;
            }
            const bool dirty = valid && static_cast<uint32_t>(w2 >> 32) != 0u;
            uint32_t tv, td;
            const uint32_t rv = book_rank(valid, s_w4, &tv);
            const uint32_t rd = book_rank(dirty, s_w4, &td);
            if (!book_exchange(ctl, a.xw, ++seq, tv, td, s_a, s_b, &s_abort))
                return;
            uint32_t vb = 0, db = 0, vall = 0;
            for (int k = 0; k < kBookWg; ++k) {
                vb += k < g ? s_a[k] : 0u;
                db += k < g ? s_b[k] : 0u;
                vall += s_a[k];
            }
            const bool take = valid && static_cast<long long>(vb + rv) < need;
            if (take) {
                const uint32_t key = static_cast<uint32_t>(w2);
                stc(c.slot_of + key, -1);
                stc(line_word(c.line, ls, 3), line_w3(kFree, false));
                stc(c.free_list + (ftop - M + taken_before + vb + rv), ls);
                if (dirty) {      // every valid entry in front of a victim is a victim: its rank among the dirty victims
                    const long long at_e = at + dirty_before + db + rd;
                    a.ev_slot[at_e] = ls;
                    a.ev_key[at_e] = key;
                    a.ev_upd[at_e] = static_cast<int32_t>(w2 >> 32);
                }
            }
            // the round's last victim tells everybody where the log's head is now, and how many dirty lines were taken
            const bool last = take && static_cast<long long>(vb + rv) == need - 1;
            const bool all_taken = static_cast<long long>(vall) <= need;
            uint32_t adv = 0, dcut = 0;
            if (last) {
                adv = static_cast<uint32_t>(e + 1 - new_head);
                dcut = db + rd + (dirty ? 1u : 0u);
            }
            if (!all_taken || static_cast<long long>(vall) == need) {
                // the cut is inside this round: one thread holds it
                const unsigned long long mm = __ballot(last);
                __syncthreads();
                if (tid < 2)
                    s_w4[tid] = 0;
                __syncthreads();
                if (mm != 0ull && lane_id() == __builtin_ctzll(mm)) {
                    s_w4[0] = adv;
                    s_w4[1] = dcut;
                }
                __syncthreads();
                if (!book_exchange(ctl, a.xw, ++seq, s_w4[0], s_w4[1], s_a, s_b, &s_abort))
                    return;
                uint32_t A = 0, D = 0;
                for (int k = 0; k < kBookWg; ++k) {
                    A += s_a[k];
                    D += s_b[k];
                }
                new_head += A;
                dirty_before += D;
                taken_before += static_cast<uint32_t>(need);
                need = 0;
            } else {
                uint32_t dall = 0;
                for (int k = 0; k < kBookWg; ++k)
                    dall += s_b[k];
                new_head = min(new_head + static_cast<long long>(kBookWg) * kBookThreads, tail);
                taken_before += vall;
                dirty_before += dall;
                need -= vall;
            }
        }
        const long long evicted = need0 - need;
        // ---- the batch is booked -------------------------------------------------------------------------------------------
        head = new_head;
        ftop = ftop - M + evicted;
        size = size + M - evicted;
        tail += U;
        clock += U;
        if (g == 0 && tid == 0) {
            PlanRec r{};
            r.n = n;
            r.U = U;
            r.M = M;
            r.E = dirty_before;
            r.evicted = evicted;
            r.size = size;
            r.full = size == c.limit;
            r.npush = -1;         // (counted when the perf dict asks: ha_cache_perf)
            r.erep = dirty_before;
            r.vh_slot = -1;
            a.rec[i] = r;
        }
        // the next batch probes what this one inserted and evicted
        if (!book_exchange(ctl, a.xw, ++seq, 0, 0, s_a, s_b, &s_abort))
            return;
    }
    if (g == 0 && tid == 0) {
        ctl->clock = clock;
        ctl->log_tail = tail;
        ctl->log_head = head;
        ctl->free_top = ftop;
        ctl->size = size;
        ctl->evict_n = 0;
        ctl->book_seq = static_cast<long long>(seq);
        ctl->U = 0;
        ctl->M = 0;
    }
}


// ---- LFU / LFUOpt: the bookkeeping of a block of batches ---------------------------------------------------------------------
// What the two policies do to a lookup + update pair of the SAME keys (lfu_cache.cc / lfuopt_cache.cc; every line found is
// touched twice, by the lookup and by the update: use + 2, or -- LFUOpt -- into the never-evicted store once use reaches 10),
// given that the lowest use bucket is EMPTY when the batch starts (it is after every pair: the update's touch lifts every line
// the lookup inserted; ha_cache_plan_block checks it when the planned flow takes over from call-by-call calls):
//   cache not full (free0 = limit - size > 0): the first free0 misses are inserted; every further insert evicts the back of
//       the lowest bucket = the batch's own oldest insert (lfu_cache.cc:31-42): of M misses the first M - free0 never stay;
//   cache full: the FIRST insert evicts the line with the least (use, arrival) of all lines outside LFUOpt's store -- lines
//       of this batch included, as the lookup's touch left them -- and every further insert evicts the insert before it: one
//       old line leaves, the batch's LAST miss stays (LFUOpt with nothing outside the store: every insert is dropped,
//       lfuopt_cache.cc:18-24);
//   the update does not find the keys whose inserts did not stay (nor the evicted line's, when the batch holds it): their
//       gradients go to a line without data that is pushed at once (cache.cc:147-152,159).
// "The least (use, arrival) of all lines" is the one global question; the call-by-call flow answers it by a scan over every
// line (cache_scan_victim_*), 3.4 M records per batch at configs[1]'s cache.  Here a two-level minimum is kept instead: lkey[s]
// = use << 48 | stamp of the line in slot s (all ones: none / stored), bmin[b] = the minimum of a block of 32 slots.  A batch's
// touches rewrite their lines' keys and re-reduce those blocks; the question is one pass over bmin (106 K words at that cache).
constexpr int kLfuBlk = 32;
constexpr unsigned long long kKeyNone = ~0ull;
struct LfuTree {
    unsigned long long *lkey;    // [nblk * kLfuBlk]
    unsigned long long *bmin;    // [nblk]
    long long nblk;
    unsigned long long *xk;      // [kBookWg] a query's candidate per workgroup: key, block
    long long *xb;
};
__device__ __forceinline__ unsigned long long lfu_key(uint32_t use, unsigned long long stamp) {
    // (a use count beyond 65,535 orders by arrival alone among its like: never the minimum in practice)
    return (static_cast<unsigned long long>(use < 0xFFFFu ? use : 0xFFFFu) << 48) | (stamp & 0xFFFFFFFFFFFFull);
}
__global__ __launch_bounds__(256) void cache_lfu_keys_kernel(Cache c, LfuTree t) {
    const long long s = blockIdx.x * 256ll + threadIdx.x;
    if (s >= t.nblk * kLfuBlk)
        return;
    unsigned long long k = kKeyNone;
    if (s < c.S) {
        const LineMeta m = c.line[s];
        if (m.state == kResident)
            k = lfu_key(static_cast<uint32_t>(m.freq), m.stamp);
    }
    t.lkey[s] = k;
}
__global__ __launch_bounds__(256) void cache_lfu_mins_kernel(LfuTree t) {
    const long long b = blockIdx.x * 256ll + threadIdx.x;
    if (b >= t.nblk)
        return;
    unsigned long long m = kKeyNone;
    for (int k = 0; k < kLfuBlk; ++k) {
        const unsigned long long v = t.lkey[b * kLfuBlk + k];
        m = v < m ? v : m;
    }
    t.bmin[b] = m;
}

// book_exchange with a 40-bit payload per workgroup
__device__ __forceinline__ bool book_exchange_p(CacheCtl *ctl, unsigned long long *xw, unsigned long long seq,
                                                unsigned long long payload, unsigned long long *s_p, int *s_abort) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long *words = xw + (seq & 3ull) * kBookWg;
    const int tid = threadIdx.x;
    if (tid == 0)
        stc(words + blockIdx.x, ((seq & 0xFFFFFFull) << 40) | (payload & 0xFFFFFFFFFFull));
    if (tid < kBookWg) {
        unsigned long long w = ldc(words + tid);
        if ((w >> 40) != (seq & 0xFFFFFFull)) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
            do {
                __builtin_amdgcn_s_sleep(1);
                w = ldc(words + tid);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                    *s_abort = 1;
                    ctl->fb_timeout = 1;
                    break;
                }
            } while ((w >> 40) != (seq & 0xFFFFFFull));
        }
        s_p[tid] = w & 0xFFFFFFFFFFull;
    }
    __syncthreads();
    return *s_abort == 0;
}
// the block minimum of slot s's block from its 32 keys (an exchange separates this from the keys' stores)
__device__ __forceinline__ void lfu_block_refresh(const LfuTree &t, long long s) {
    const long long b = s / kLfuBlk;
    const unsigned long long *p = t.lkey + b * kLfuBlk;
    unsigned long long v[kLfuBlk];
#pragma unroll
    for (int k = 0; k < kLfuBlk; ++k)
        v[k] = ldc(p + k);
    unsigned long long m = kKeyNone;
#pragma unroll
    for (int k = 0; k < kLfuBlk; ++k)
        m = v[k] < m ? v[k] : m;
    stc(t.bmin + b, m);
}
// the slot with the least key of all (-1: there is none) and that key; one exchange.  Every workgroup settles the SLOT of
// its candidate before the exchange (nobody rewrites keys between the exchange in front of a query and the query's own);
// behind it the first workgroups through are already rewriting lines -- the answer must not be looked up again there.
__device__ __forceinline__ bool lfu_query(CacheCtl *ctl, unsigned long long *xw, unsigned long long seq, const LfuTree &t,
                                          unsigned long long *s_p, unsigned long long *s_k, long long *s_i, int *s_abort,
                                          long long *slot_out, unsigned long long *key_out) {
    const int tid = threadIdx.x, g = blockIdx.x, lane = lane_id(), w = tid >> 6;
    unsigned long long best = kKeyNone;
    long long bi = -1;
    for (long long b = g * kBookThreads + tid; b < t.nblk; b += static_cast<long long>(kBookWg) * kBookThreads) {
        const unsigned long long v = ldc(t.bmin + b);
        if (v < best) {
            best = v;
            bi = b;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o, 64);
        const long long oi = __shfl_xor(bi, o, 64);
        if (ob < best) {
            best = ob;
            bi = oi;
        }
    }
    __syncthreads();
    if (lane == 0) {
        s_k[w] = best;
        s_i[w] = bi;
    }
    __syncthreads();
    if (w == 0) {
        best = s_k[0];
        bi = s_i[0];
        for (int k = 1; k < kBookThreads / 64; ++k)
            if (s_k[k] < best) {
                best = s_k[k];
                bi = s_i[k];
            }
        // (keys are unique: a stamp is given once) the slot of the block that carries the key
        long long slot = -1;
        if (best != kKeyNone) {
            const unsigned long long v = lane < kLfuBlk ? ldc(t.lkey + bi * kLfuBlk + lane) : kKeyNone;
            const unsigned long long m = __ballot(v == best);
            slot = m ? bi * kLfuBlk + __builtin_ctzll(m) : -1;
        }
        if (lane == 0) {
            stc(t.xk + g, slot >= 0 ? best : kKeyNone);
            stc(t.xb + g, slot);
        }
    }
    if (!book_exchange_p(ctl, xw, seq, 0ull, s_p, s_abort))
        return false;
    if (w == 0) {
        best = lane < kBookWg ? ldc(t.xk + lane) : kKeyNone;
        bi = lane < kBookWg ? ldc(t.xb + lane) : -1;
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ob = __shfl_xor(best, o, 64);
            const long long oi = __shfl_xor(bi, o, 64);
            if (ob < best) {
                best = ob;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_i[0] = best != kKeyNone ? bi : -1;
            s_k[0] = best;
        }
    }
    __syncthreads();
    *slot_out = s_i[0];
    *key_out = s_k[0];
    __syncthreads();
    return true;
}
// is `key` one of the batch's sorted unique keys?  (every thread of the workgroup asks; two probes)
__device__ __forceinline__ bool lfu_in_batch(const uint32_t *uniq, int U, uint32_t key) {
    const int tid = threadIdx.x;
    const int stride = (U + kBookThreads - 1) / kBookThreads;
    const int i0 = tid * stride;
    const int cnt = __syncthreads_count(i0 < U && uniq[i0] <= key);
    const int seg0 = (cnt > 0 ? cnt - 1 : 0) * stride;
    return __syncthreads_or(cnt > 0 && tid < stride && seg0 + tid < U && uniq[seg0 + tid] == key) != 0;
}

__global__ __launch_bounds__(kBookThreads) void cache_book_lfu_kernel(Cache c, BookArgs a, LfuTree t) {
    __shared__ unsigned long long s_p[kBookWg], s_k[kBookThreads / 64];
    __shared__ long long s_i[kBookThreads / 64];
    __shared__ uint32_t s_w4[kBookThreads / 64];
    __shared__ int s_abort;
    CacheCtl *ctl = c.ctl;
    const int tid = threadIdx.x, g = blockIdx.x;
    if (tid == 0)
        s_abort = 0;
    long long clock = ctl->clock, ftop = ctl->free_top, size = ctl->size, n_hash = ctl->n_hash;
    unsigned long long seq = static_cast<unsigned long long>(ctl->book_seq);
    const bool opt = c.policy == kLFUOpt;
    const uint32_t new_use = opt ? 1u : 2u;       // a line the lookup inserted, after the update's touch
    __syncthreads();
    for (int i = 0; i < a.count; ++i) {
        const int n = a.n[i];
        const long long at = static_cast<long long>(i) * a.nmax;
        if (n == 0) {
            if (g == 0 && tid == 0) {
                PlanRec r{};
                r.size = size;
                r.full = size == c.limit;
                r.vh_slot = -1;
                a.rec[i] = r;
            }
            continue;
        }
        const int U = static_cast<int>(a.hdr[i]->n_unique);
        const uint32_t *uniq = a.uniq[i];
        const int32_t *counts = a.counts[i];
        const int per = (U + kBookWg - 1) / kBookWg;
        const int u0 = min(g * per, U), u1 = min(u0 + per, U);
        // ---- phase 1: probe; what the two touches will make of every line found ---------------------------------------------
        int sl[kBookKeysPerThread];
        uint32_t kk[kBookKeysPerThread], rk[kBookKeysPerThread];
        unsigned long long w2[kBookKeysPerThread], w3[kBookKeysPerThread];
        bool miss[kBookKeysPerThread];
        uint32_t wg_miss = 0, wg_st1 = 0, wg_st2 = 0;
#pragma unroll
        for (int j = 0; j < kBookKeysPerThread; ++j) {
            const int u = u0 + j * kBookThreads + tid;
            const bool on = u < u1;
            kk[j] = on ? uniq[u] : 0u;
            const bool known = on && kk[j] < static_cast<unsigned long long>(c.length);
            sl[j] = known ? ldc(c.slot_of + kk[j]) : -1;
            miss[j] = known && sl[j] < 0;
            if (on && !known) {
                a.it_slot[at + u] = -1;
                a.it_flag[at + u] = 0;
                a.it_upd[at + u] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < kBookKeysPerThread; ++j) {
            bool st1 = false, st2 = false;
            w2[j] = w3[j] = 0ull;
            if (sl[j] >= 0) {
                w2[j] = ldc(line_word(c.line, sl[j], 2));
                w3[j] = ldc(line_word(c.line, sl[j], 3));
                const uint32_t f = static_cast<uint32_t>(w3[j]);
                const bool res = static_cast<uint8_t>(w3[j] >> 32) == kResident;
                st1 = opt && res && f + 1 >= static_cast<uint32_t>(kUseCntMax);     // lfuopt_cache.cc:33-39, by the lookup's touch
                st2 = opt && res && !st1 && f + 2 >= static_cast<uint32_t>(kUseCntMax);   // by the update's
            }
            uint32_t tot;
            rk[j] = wg_miss + book_rank(miss[j], s_w4, &tot);
            wg_miss += tot;
            wg_st1 += static_cast<uint32_t>(__syncthreads_count(st1));
            wg_st2 += static_cast<uint32_t>(__syncthreads_count(st2));
        }
        if (!book_exchange_p(ctl, a.xw, ++seq, static_cast<unsigned long long>(wg_miss) | (static_cast<unsigned long long>(wg_st1) << 13) |
                                                 (static_cast<unsigned long long>(wg_st2) << 26), s_p, &s_abort))
            return;
        long long mb = 0, M = 0, ST1 = 0, ST2 = 0;
        for (int k = 0; k < kBookWg; ++k) {
            const long long m_k = static_cast<long long>(s_p[k] & 0x1FFFull);
            mb += k < g ? m_k : 0;
            M += m_k;
            ST1 += static_cast<long long>((s_p[k] >> 13) & 0x1FFFull);
            ST2 += static_cast<long long>((s_p[k] >> 26) & 0x1FFFull);
        }
        const long long free0 = c.limit > size ? c.limit - size : 0;
        const long long ev = M > free0 ? M - free0 : 0;
        const bool drop_all = opt && ev > 0 && free0 == 0 && n_hash - ST1 == 0;
        const bool scan = ev > 0 && free0 == 0 && !drop_all;
        const long long v_new = drop_all ? M : (scan ? ev - 1 : ev);     // the batch's first v_new inserts do not stay
        // ---- the line that leaves for the batch's first insert ----------------------------------------------------------------
        long long vslot = -1;
        unsigned long long vw2 = 0, vkey_lfu = kKeyNone;
        bool vin = false;
        if (scan) {
            // as the tree stands (the lines as the batch found them): the lookup's touch only raises keys, so a minimum that
            // is NOT a line of this batch is the minimum after the touch as well
            // (of the line's record only the word {key, updates} is read: nobody writes it before the batch is booked; its
            // state word is the first thing the fastest workgroup rewrites)
            if (!lfu_query(ctl, a.xw, ++seq, t, s_p, s_k, s_i, &s_abort, &vslot, &vkey_lfu))
                return;
            if (vslot >= 0) {
                vw2 = ldc(line_word(c.line, vslot, 2));
                vin = lfu_in_batch(uniq, U, static_cast<uint32_t>(vw2));
            }
            if (vin) {
                // it is one of the batch's lines: the keys of the batch's lines as the lookup's touch leaves them (use + 1,
                // arrival in key order; LFUOpt: a line that reaches the store drops out), then the question again
#pragma unroll
                for (int j = 0; j < kBookKeysPerThread; ++j) {
                    const int u = u0 + j * kBookThreads + tid;
                    if (sl[j] >= 0 && static_cast<uint8_t>(w3[j] >> 32) == kResident) {
                        const uint32_t f = static_cast<uint32_t>(w3[j]);
                        const bool st1 = opt && f + 1 >= static_cast<uint32_t>(kUseCntMax);
                        stc(t.lkey + sl[j], st1 ? kKeyNone : lfu_key(f + 1, static_cast<unsigned long long>(clock + u)));
                    }
                }
                if (!book_exchange_p(ctl, a.xw, ++seq, 0ull, s_p, &s_abort))
                    return;
#pragma unroll
                for (int j = 0; j < kBookKeysPerThread; ++j)
                    if (sl[j] >= 0 && static_cast<uint8_t>(w3[j] >> 32) == kResident)
                        lfu_block_refresh(t, sl[j]);
                if (!book_exchange_p(ctl, a.xw, ++seq, 0ull, s_p, &s_abort))
                    return;
                if (!lfu_query(ctl, a.xw, ++seq, t, s_p, s_k, s_i, &s_abort, &vslot, &vkey_lfu))
                    return;
                if (vslot >= 0) {
                    vw2 = ldc(line_word(c.line, vslot, 2));
                    vin = lfu_in_batch(uniq, U, static_cast<uint32_t>(vw2));
                }
            }
            if (vslot < 0) {       // (the counters say a line exists: the tree does not hold what they count)
                if (tid == 0) {
                    ctl->fb_timeout = 2;
                    ctl->ph[8] = static_cast<unsigned long long>(i);
                    ctl->ph[9] = static_cast<unsigned long long>(size);
                    ctl->ph[10] = static_cast<unsigned long long>(M);
                    ctl->ph[11] = vkey_lfu;
                    ctl->ph[12] = static_cast<unsigned long long>(n_hash);
                }
                return;
            }
        }
        const uint32_t vkey = static_cast<uint32_t>(vw2);
        const int vupd = static_cast<int>(vw2 >> 32);
        const bool vdirty = scan && vupd != 0;
        // ---- phase 2: the state after the pair; the items --------------------------------------------------------------------
        const long long clock2 = clock + U;
        long long rf[kBookKeysPerThread];          // slots whose block minimum this thread re-reduces
#pragma unroll
        for (int j = 0; j < kBookKeysPerThread; ++j) {
            const int u = u0 + j * kBookThreads + tid;
            rf[j] = -1;
            if (u >= u1)
                continue;
            if (sl[j] >= 0) {
                const int s = sl[j];
                const uint32_t f = static_cast<uint32_t>(w3[j]);
                const uint8_t state = static_cast<uint8_t>(w3[j] >> 32);
                const bool hg = ((w3[j] >> 40) & 1ull) != 0ull;
                if (scan && s == vslot) {
                    // this line is the one the batch's first insert evicts: the lookup still reads it (slot vh_slot of the
                    // record), the update does not find it -- a line without data in a spare slot, pushed at once
                    a.it_slot[at + u] = ldc(c.free_list + (ftop - M - 1));
                    a.it_flag[at + u] = static_cast<uint8_t>(kPosTemp | kPosPush | kPosVictim | (hg ? kPosVictimHg : 0) |
                                                             (vdirty ? kPosVictimPush : 0));
                    a.it_upd[at + u] = counts[u];
                    continue;
                }
                const int upd = static_cast<int>(w2[j] >> 32) + counts[u];
                const bool push = upd > c.push_bound;
                uint8_t ns = state;
                uint32_t f2 = f;
                if (state == kResident) {
                    if (opt && f + 2 >= static_cast<uint32_t>(kUseCntMax))
                        ns = kStored;
                    else
                        f2 = f + 2;
                }
                const unsigned long long st = static_cast<unsigned long long>(clock2 + u);
                stc(line_word(c.line, s, 0), st);
                stc(line_word(c.line, s, 2), static_cast<unsigned long long>(kk[j]) |
                                                 (static_cast<unsigned long long>(static_cast<uint32_t>(push ? 0 : upd)) << 32));
                stc(line_word(c.line, s, 3), static_cast<unsigned long long>(f2) | line_w3(ns, true));
                if (state == kResident) {
                    stc(t.lkey + s, ns == kStored ? kKeyNone : lfu_key(f2, st));
                    rf[j] = s;
                }
                a.it_slot[at + u] = s;
                a.it_flag[at + u] = static_cast<uint8_t>((hg ? kPosInit : 0) | (push ? kPosPush : 0));
                a.it_upd[at + u] = upd;
            } else if (miss[j]) {
                // the batch's misses take the top M stack entries in rank order from below: the ones that stay are on top
                const long long q = mb + rk[j];
                const int s = ldc(c.free_list + (ftop - M + q));
                const int upd = counts[u];
                if (q < v_new) {          // inserted and evicted again by a later insert (or dropped): see the header
                    a.it_slot[at + u] = s;
                    a.it_flag[at + u] = static_cast<uint8_t>(kPosMiss | kPosTemp | kPosPush);
                    a.it_upd[at + u] = upd;
                    continue;
                }
                const bool push = upd > c.push_bound;
                const unsigned long long st = static_cast<unsigned long long>(clock2 + u);
                stc(line_word(c.line, s, 0), st);
                stc(line_word(c.line, s, 2), static_cast<unsigned long long>(kk[j]) |
                                                 (static_cast<unsigned long long>(static_cast<uint32_t>(push ? 0 : upd)) << 32));
                stc(line_word(c.line, s, 3), static_cast<unsigned long long>(new_use) | line_w3(kResident, true));
                stc(c.slot_of + kk[j], s);
                stc(t.lkey + s, lfu_key(new_use, st));
                rf[j] = s;
                a.it_slot[at + u] = s;
                a.it_flag[at + u] = static_cast<uint8_t>(kPosMiss | (push ? kPosPush : 0));
                a.it_upd[at + u] = upd;
            }
        }
        const long long stay = M - v_new;
        // (the evicted line of the batch itself was counted among the lines the update's touch stores when its use was 8: the
        // key it left with is use + 1)
        const bool v_st2 = scan && vin && opt && static_cast<uint32_t>(vkey_lfu >> 48) + 1 >= static_cast<uint32_t>(kUseCntMax);
        if (g == 0 && tid == 0) {
            long long E = 0;
            if (scan) {
                stc(c.slot_of + vkey, -1);
                stc(line_word(c.line, vslot, 3), line_w3(kFree, false));
                stc(t.lkey + vslot, kKeyNone);
                if (vdirty && !vin) {
                    a.ev_slot[at] = static_cast<int32_t>(vslot);
                    a.ev_key[at] = vkey;
                    a.ev_upd[at] = vupd;
                    E = 1;
                }
            }
            PlanRec r{};
            r.n = n;
            r.U = U;
            r.M = M;
            r.E = E;
            r.evicted = scan ? 1 : 0;
            r.size = size + stay - (scan ? 1 : 0);
            r.full = r.size == c.limit;
            r.npush = -1;
            r.erep = vdirty ? 1 : 0;
            r.umiss = v_new + (scan && vin ? 1 : 0);
            r.vh_slot = scan && vin ? vslot : -1;
            r.vh_upd = scan && vin ? vupd : 0;
            a.rec[i] = r;
        }
        // ---- the batch is booked: the next one probes what this one left; the blocks of the rewritten keys ---------------------
        if (!book_exchange_p(ctl, a.xw, ++seq, 0ull, s_p, &s_abort))
            return;
#pragma unroll
        for (int j = 0; j < kBookKeysPerThread; ++j)
            if (rf[j] >= 0)
                lfu_block_refresh(t, rf[j]);
        if (scan && g == 0 && tid == 0) {
            lfu_block_refresh(t, vslot);
            stc(c.free_list + (ftop - stay), static_cast<int32_t>(vslot));    // (everybody has read the stack's old top)
        }
        ftop = ftop - stay + (scan ? 1 : 0);
        size = size + stay - (scan ? 1 : 0);
        n_hash = n_hash - ST1 - ST2 + (v_st2 ? 1 : 0) + stay - (scan ? 1 : 0);
        clock += 2ll * U;
    }
    if (g == 0 && tid == 0) {
        ctl->clock = clock;
        ctl->free_top = ftop;
        ctl->size = size;
        ctl->n_hash = n_hash;
        ctl->n_base = 0;
        ctl->evict_n = 0;
        ctl->book_seq = static_cast<long long>(seq);
        ctl->U = 0;
        ctl->M = 0;
    }
}

// ---- the items per sorted position (side stream, behind the bookkeeping launch) --------------------------------------------
// pos_item[p] = {slot, key, kPos* flags (| kPosHead at the first position of a key), occurrence index perm[p]}: what a row
// wave of position p needs, in ONE 16-byte record instead of three dependent lookups (upos[p] -> item of the key -> rows).
struct PlanExpandPtrs {
    const int32_t *upos[kPlanBlockMax];
    const int32_t *perm[kPlanBlockMax];
};
__global__ __launch_bounds__(256) void cache_plan_expand_kernel(BookArgs a, PlanExpandPtrs e, int4 *pos_item, int32_t *it_upd_pos) {
    const int i = blockIdx.y;
    const int n = a.n[i];
    const long long at = static_cast<long long>(i) * a.nmax;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const int u = e.upos[i][p];
        const bool head = p == 0 || e.upos[i][p - 1] != u;
        pos_item[at + p] = int4{a.it_slot[at + u], static_cast<int>(a.uniq[i][u]),
                                static_cast<int>(a.it_flag[at + u]) | (head ? kPosHead : 0), e.perm[i][p]};
        it_upd_pos[at + p] = a.it_upd[at + u];
    }
}

// ---- the lookup of a planned batch: ONE launch, a wave per sorted position -------------------------------------------------
// dest[perm[p],:] = the line of the position's key after syncEmbedding (cache.cc:84-97).  The pull decision is taken by every
// wave of a key from words nothing in this launch writes (the line's version, the store's version); the wave of the key's
// first position refreshes the line and STAGES its new version (pver[p]; the update's launch commits it -- a store into the
// record here would race with the other waves' reads).  Two round trips: the position's record; then versions and row together.
// (Measured, docs/EXPERIMENTS.md round 6: the forward gather's shape -- the output as flat 16-byte vectors, four vectors of
// different rows per lane, 256-thread workgroups -- 11.3 us against 9.1 for a wave per position; without the two version reads
// 9.07 against 9.29 us; without the row stores 7.65: the launch is the part's "copy of 6,656 random 2 KB rows" (DESIGN.md
// section 6, yardstick: 8.0 us), the staleness check is almost free beside it.)
template <int VEC>
__global__ __launch_bounds__(1024) void cache_lookup_planned_kernel(
    Cache c, const int4 *__restrict__ pos_item, long long n, float *__restrict__ dest, long long *__restrict__ pver,
    const PlanRec *__restrict__ rec) {
    const int lane = lane_id();
    const long long p = static_cast<long long>(blockIdx.x) * 16ll + uniform(static_cast<int>(threadIdx.x >> 6));
    if (p >= n)
        return;
    const int4 it = pos_item[p];
    int s = uniform(it.x);
    const long long lk = static_cast<long long>(uniform(static_cast<uint32_t>(it.y)));
    int fl = uniform(it.z);
    if (fl & kPosVictim) {      // (LFU policies) the line this batch's own lookup evicts: still in its old slot when the rows are read
        s = uniform(static_cast<int>(rec->vh_slot));
        fl = (fl & ~kPosInit) | ((fl & kPosVictimHg) ? kPosInit : 0);
    }
    const bool head = (fl & kPosHead) != 0;
    float *out = dest + static_cast<long long>(uniform(it.w)) * c.width;
    if (s >= c.S || lk >= c.store_rows || it.w < 0 || it.w >= n) {
        // an item no bookkeeping launch wrote (it gave up, or the row launch ran in front of it): nothing is touched, the sticky
        // word says so (ha_cache_state / ha_cache_perf raise)
        if (lane == 0 && c.ctl->fb_timeout == 0) {
            c.ctl->fb_timeout = 3;
            c.ctl->ph[8] = static_cast<unsigned long long>(static_cast<uint32_t>(it.x));
            c.ctl->ph[9] = static_cast<unsigned long long>(static_cast<uint32_t>(it.y));
            c.ctl->ph[10] = static_cast<unsigned long long>(static_cast<uint32_t>(it.z));
            c.ctl->ph[11] = static_cast<unsigned long long>(static_cast<uint32_t>(it.w));
            c.ctl->ph[12] = static_cast<unsigned long long>(p) | (static_cast<unsigned long long>(rec->M) << 32);
            c.ctl->ph[13] = static_cast<unsigned long long>(rec->n) | (static_cast<unsigned long long>(rec->U) << 32);
            c.ctl->ph[14] = static_cast<unsigned long long>(rec->size);
            c.ctl->ph[15] = static_cast<unsigned long long>(n);
        }
        return;
    }
    if (s < 0) {
        for (long long j = lane; j < c.width; j += kWave)
            out[j] = 0.f;
        return;
    }
    float *line = c.data + static_cast<long long>(s) * c.width;
    const bool is_miss = (fl & kPosMiss) != 0;
    const long long sv = c.srv_ver[lk];
    const long long v = is_miss ? -1 : c.line[s].version;
    // the cached row, requested beside the two versions (a hit that is not stale -- the usual case -- has it on the way)
    float4v x0{0.f, 0.f, 0.f, 0.f}, x1 = x0;
    const long long j0 = lane * 4, j1 = j0 + kWave * 4;
    if (VEC == 4 && !is_miss) {
        if (j0 < c.width)
            x0 = ld4(line + j0);
        if (j1 < c.width)
            x1 = ld4(line + j1);
    }
    const bool pull = is_miss || v == -1 || sv - v > c.pull_bound;
    if (!pull) {
        if (VEC == 4) {
            if (j0 < c.width)
                st4_nt(out + j0, x0);
            if (j1 < c.width)
                st4_nt(out + j1, x1);
            for (long long j = j1 + kWave * 4; j < c.width; j += kWave * 4)
                st4_nt(out + j, ld4(line + j));
        } else {
            for (long long j = lane; j < c.width; j += kWave)
                out[j] = line[j];
        }
        if (head && lane == 0)
            pver[p] = kVerKeep;
        return;
    }
    const bool hg = (fl & kPosInit) != 0;      // the line has a gradient buffer: Line::addup() re-adds it to the pulled row
    const float *src = c.table + lk * c.width;
    const float *gr = c.grad + static_cast<long long>(s) * c.width;
    if (VEC == 4) {
        for (long long j = lane * 4; j < c.width; j += kWave * 4) {
            float4v x = ld4(src + j);
            if (hg) {
                const float4v gv = ld4(gr + j);
                x = float4v{__fadd_rn(x[0], gv[0]), __fadd_rn(x[1], gv[1]), __fadd_rn(x[2], gv[2]), __fadd_rn(x[3], gv[3])};
            }
            st4_nt(out + j, x);
            if (head)
                st4(line + j, x);
        }
    } else {
        for (long long j = lane; j < c.width; j += kWave) {
            float x = src[j];
            if (hg)
                x = __fadd_rn(x, gr[j]);  // Line::addup(): data += grad
            out[j] = x;
            if (head)
                line[j] = x;
        }
    }
    if (head && lane == 0)
        pver[p] = sv;
}

// ---- the update of a planned batch: ONE launch -----------------------------------------------------------------------------
// the first kPlanMetaBlocks workgroups: a thread per sorted position, the heads work: the line's version (staged by the
//     lookup, + updates for a pushed line: cache.cc:171-177), the store's version of a pushed line, hasgrad;
// the next kPlanEvictBlocks workgroups: a wave per evicted dirty line: store row += its gradient, store version += its updates
//     (PSFhandle_embedding.cc:23-27); the slot is free already (the bookkeeping freed it, nothing reuses it before the next
//     batch's lookup);
// the rest: the ordered accumulate (apply_body, DUAL == 2: gradient buffer and data row; pushed lines take the push epilogue --
//     store row += the line's new gradient, gradient buffer = 0).
// (The independent short roles come first in the grid: they start with the launch, not behind 400 accumulate workgroups.)
constexpr int kPlanEvictBlocks = 64, kPlanMetaBlocks = 8;
template <int VEC>
__global__ __launch_bounds__(1024, 8) void cache_update_planned_kernel(
    Cache c, const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm, int n, const float *__restrict__ grads,
    ApplyMaps maps, const int32_t *__restrict__ it_upd_pos, const long long *__restrict__ pver,
    const int32_t *__restrict__ ev_slot, const uint32_t *__restrict__ ev_key, const int32_t *__restrict__ ev_upd,
    const PlanRec *__restrict__ rec) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    const int b = blockIdx.x;
    if (b >= kPlanMetaBlocks + kPlanEvictBlocks) {
        apply_body<kModeSgd, VEC, 2>(c.grad, static_cast<uint64_t>(c.S), static_cast<int>(c.width), sorted, perm, nullptr, n,
                                     grads, -1.0f, b - kPlanMetaBlocks - kPlanEvictBlocks, s_apply, nullptr, maps);
        return;
    }
    const int lane = lane_id();
    if (b >= kPlanMetaBlocks) {
        const int E = static_cast<int>(min(rec->E, static_cast<long long>(n)));
        for (int j = (b - kPlanMetaBlocks) * 16 + static_cast<int>(threadIdx.x >> 6); j < E; j += kPlanEvictBlocks * 16) {
            const int s = uniform(ev_slot[j]);
            const long long lk = static_cast<long long>(uniform(ev_key[j]));
            if (s < 0 || s >= c.S || lk >= c.store_rows)
                continue;
            float *row = c.table + lk * c.width;
            const float *g = c.grad + static_cast<long long>(s) * c.width;
            if (VEC == 4) {
                for (long long q0 = 0; q0 < c.width; q0 += kWave * 8) {
                    const long long q = q0 + lane * 4, q2 = q + kWave * 4;
                    const bool x = q < c.width, y = q2 < c.width;
                    float4v r0{0.f, 0.f, 0.f, 0.f}, r1 = r0, g0 = r0, g1 = r0;
                    if (x) {
                        r0 = ld4(row + q);
                        g0 = ld4(g + q);
                    }
                    if (y) {
                        r1 = ld4(row + q2);
                        g1 = ld4(g + q2);
                    }
                    if (x)
                        st4(row + q, float4v{__fadd_rn(r0[0], g0[0]), __fadd_rn(r0[1], g0[1]), __fadd_rn(r0[2], g0[2]),
                                             __fadd_rn(r0[3], g0[3])});
                    if (y)
                        st4(row + q2, float4v{__fadd_rn(r1[0], g1[0]), __fadd_rn(r1[1], g1[1]), __fadd_rn(r1[2], g1[2]),
                                              __fadd_rn(r1[3], g1[3])});
                }
            } else {
                for (long long q = lane; q < c.width; q += kWave)
                    row[q] = __fadd_rn(row[q], g[q]);
            }
            if (lane == 0)
                c.srv_ver[lk] += ev_upd[j];
        }
        return;
    }
    for (int p = b * 1024 + static_cast<int>(threadIdx.x); p < n; p += kPlanMetaBlocks * 1024) {
        const int4 it = maps.pos_item[p];
        if (it.x < 0 || !(it.z & kPosHead) || it.x >= c.S || static_cast<long long>(static_cast<uint32_t>(it.y)) >= c.store_rows)
            continue;
        if (it.z & kPosTemp) {     // (LFU policies) a line that is not in the cache: pushed, nothing of it stays
            c.srv_ver[static_cast<uint32_t>(it.y)] += it_upd_pos[p] + ((it.z & kPosVictimPush) ? static_cast<int>(rec->vh_upd) : 0);
            continue;
        }
        const int s = it.x;
        const long long pv = pver[p];
        long long v = pv != kVerKeep ? pv : c.line[s].version;
        if (it.z & kPosPush) {
            const int upd = it_upd_pos[p];
            v += upd;
            c.srv_ver[static_cast<uint32_t>(it.y)] += upd;
        }
        c.line[s].version = v;
        c.hasgrad[s] = 1;
    }
}

// the perf dict's data-dependent counts of a planned batch, on demand: lines the lookup pulled, lines the update pushed
__global__ __launch_bounds__(1024) void cache_plan_count_kernel(PlanRec *rec, const long long *pver, const int4 *pos_item) {
    __shared__ unsigned long long s_p[16], s_q[16];
    const int n = static_cast<int>(rec->n);
    unsigned long long pulled = 0, pushed = 0;
    for (int p = threadIdx.x; p < n; p += 1024) {
        const int4 it = pos_item[p];
        if (it.x < 0 || !(it.z & kPosHead))
            continue;
        pulled += pver[p] != kVerKeep ? 1 : 0;
        pushed += (it.z & kPosPush) ? 1 : 0;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        pulled += __shfl_xor(pulled, o, 64);
        pushed += __shfl_xor(pushed, o, 64);
    }
    if (lane_id() == 0) {
        s_p[threadIdx.x >> 6] = pulled;
        s_q[threadIdx.x >> 6] = pushed;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int k = 0; k < 16; ++k) {
            a += s_p[k];
            b += s_q[k];
        }
        rec->pulled = static_cast<long long>(a);
        rec->npush = static_cast<long long>(b);
    }
}

}  // namespace ha

using namespace ha;

// ---- host side -------------------------------------------------------------------------------------------------------------
static int plan_slot_alloc(ha_cache *h, PlanSlot &sl) {
    if (sl.it_slot)
        return 0;
    const size_t per = static_cast<size_t>(h->c.nmax), all = per * kPlanBlockMax;
    bool ok = true;
#define PLAN_ALLOC(field, count)                                                \
    do {                                                                        \
        if (ok && dmalloc(&sl.field, static_cast<size_t>(count)) != 0)          \
            ok = false;                                                         \
        else if (ok)                                                            \
            h->allocs.push_back(sl.field);                                      \
    } while (0)
    PLAN_ALLOC(it_slot, all);
    PLAN_ALLOC(it_flag, all);
    PLAN_ALLOC(it_upd, all);
    PLAN_ALLOC(pos_item, all);
    PLAN_ALLOC(it_upd_pos, all);
    PLAN_ALLOC(pver, all);
    PLAN_ALLOC(ev_slot, all);
    PLAN_ALLOC(ev_key, all);
    PLAN_ALLOC(ev_upd, all);
    PLAN_ALLOC(rec, kPlanBlockMax);
#undef PLAN_ALLOC
    for (int i = 0; ok && i < kPlanBlockMax; ++i) {
        char *p = nullptr;
        if (dmalloc(&p, h->plan_bytes) != 0) {
            ok = false;
            break;
        }
        ok = dzero(p, 256) == 0;      // the plan header (sticky flags)
        sl.ws[i] = p;
        h->allocs.push_back(p);
    }
    HA_REQUIRE(ok, "cache_plan_block: out of device memory");
    HA_CHECK_HIP(hipEventCreateWithFlags(&sl.booked, hipEventDisableTiming));
    HA_CHECK_HIP(hipEventCreateWithFlags(&sl.rows_done, hipEventDisableTiming));
    return 0;
}

extern "C" int ha_cache_plan_pending(ha_cache *h) {
    if (!h)
        return 0;
    int pending = 0;
    for (const PlanSlot &sl : h->plan)
        pending += sl.count > 0 ? 2 * sl.count - sl.next_call : 0;
    return pending;
}

// The bookkeeping of the next `count` (1..16) batches: their lookups and updates follow as ha_cache_lookup_planned /
// ha_cache_update_planned, in this order, lookup and update alternating.  `side`: the stream the plans and the bookkeeping
// launch run on; `main`: the stream of the row launches (everything enqueued on it so far is ordered in front of the
// bookkeeping: the row launches of the block that used this block's buffers before, or call-by-call entry points).  At most TWO
// blocks are outstanding (the one being consumed and the next).  The key buffers must stay unchanged until the bookkeeping
// has run (event `booked`; the row launches wait for it).
extern "C" int ha_cache_plan_block(ha_cache *h, const void *const *keys, int key_kind, const int64_t *n, int count,
                                   ha_stream_t side, ha_stream_t main) {
    HA_REQUIRE(h && keys && n && (key_kind == 0 || key_kind == 1) && count >= 1 && count <= kPlanBlockMax,
               "cache_plan_block: bad arguments (1..%d batches)", kPlanBlockMax);
    Cache &c = h->c;
    HA_REQUIRE(c.table && !c.remote && !c.bypass, "cache_plan_block: a cache over a local store, not bypassed");
    HA_REQUIRE(c.row_start == 0 && c.store_rows >= c.length, "cache_plan_block: the store must hold every key of the cache's range");

    HA_REQUIRE(c.limit >= 1, "cache_plan_block: an empty cache");
    HA_REQUIRE(h->evict_empty || ha_cache_plan_pending(h) > 0, "cache_plan_block: evicted lines are pending (an update must follow "
               "the last lookup first)");
    HA_REQUIRE(h->ahead_n < 0, "cache_plan_block: a ha_cache_sort_ahead is pending");
    for (int i = 0; i < count; ++i) {
        HA_REQUIRE(n[i] >= 0 && n[i] <= c.nmax && n[i] <= kSmallMax && (n[i] == 0 || keys[i]),
                   "cache_plan_block: batch %d of %ld keys (at most min(max_batch, %d))", i, (long)n[i], kSmallMax);
        HA_REQUIRE(c.policy != kLRU || n[i] <= c.limit, "cache_plan_block: limit (%ld) must be at least the batch (%ld keys): the "
                   "lines of an LRU batch are never evicted by its own lookup", (long)c.limit, (long)n[i]);
    }
    PlanSlot &sl = h->plan[h->plan_next % ha_cache::kPlanSlots];
    int blocks_out = 0;
    for (const PlanSlot &q : h->plan)
        blocks_out += q.count > 0 && q.next_call < 2 * q.count ? 1 : 0;
    HA_REQUIRE(blocks_out < 2 && (sl.count == 0 || sl.next_call >= 2 * sl.count),
               "cache_plan_block: two planned blocks are outstanding already");
    if (plan_slot_alloc(h, sl))
        return -1;
    if (!h->plan_xw) {
        HA_REQUIRE(dmalloc(&h->plan_xw, static_cast<size_t>(4 * kBookWg)) == 0, "cache_plan_block: out of device memory");
        h->allocs.push_back(h->plan_xw);
        if (dzero(h->plan_xw, 4 * kBookWg * 8))
            return -1;
        HA_CHECK_HIP(hipEventCreateWithFlags(&h->plan_fork, hipEventDisableTiming));
    }
    hipStream_t ss = as_stream(side), ms = as_stream(main);
    LfuTree tree{};
    if (c.policy != kLRU) {
        if (!h->lfu_lkey) {
            h->lfu_nblk = (c.S + kLfuBlk - 1) / kLfuBlk;
            HA_REQUIRE(dmalloc(&h->lfu_lkey, static_cast<size_t>(h->lfu_nblk * kLfuBlk)) == 0 &&
                       dmalloc(&h->lfu_bmin, static_cast<size_t>(h->lfu_nblk)) == 0 &&
                       dmalloc(&h->lfu_xk, static_cast<size_t>(kBookWg)) == 0 && dmalloc(&h->lfu_xb, static_cast<size_t>(kBookWg)) == 0,
                       "cache_plan_block: out of device memory");
            h->allocs.push_back(h->lfu_lkey);
            h->allocs.push_back(h->lfu_bmin);
            h->allocs.push_back(h->lfu_xk);
            h->allocs.push_back(h->lfu_xb);
        }
        tree = LfuTree{h->lfu_lkey, h->lfu_bmin, h->lfu_nblk, h->lfu_xk, h->lfu_xb};
        if (!h->lfu_tree_ok) {
            // the planned flow takes over from call-by-call calls (or starts): the lowest use bucket must be empty -- it is
            // after every lookup + update pair, not after a lookup without its update (see cache_book_lfu_kernel)
            HA_CHECK_HIP(hipStreamSynchronize(ms));
            CacheCtl ctl_h;
            HA_CHECK_HIP(hipMemcpy(&ctl_h, c.ctl, sizeof(ctl_h), hipMemcpyDeviceToHost));
            HA_REQUIRE(ctl_h.n_base == 0, "cache_plan_block: %ld lines are in the lowest use bucket (a lookup without its update "
                       "put them there): the planned flow of the LFU policies starts from a cache whose lines were all updated",
                       (long)ctl_h.n_base);
        }
    }
    if (ss != ms) {
        if (sl.rows_recorded && h->last_planned_type >= 0) {
            // the planned flow goes on: this slot's buffers were last read by the rows of the block two before the one being
            // consumed -- nothing else of the row stream concerns the bookkeeping (it owns the control fields, the rows the
            // data fields), and this event is long complete: no barrier parked on the planning stream's queue
            HA_CHECK_HIP(hipStreamWaitEvent(ss, sl.rows_done, 0));
        } else {
            // the first blocks, or call-by-call entry points ran in between: everything enqueued on `main` so far comes first
            HA_CHECK_HIP(hipEventRecord(h->plan_fork, ms));
            HA_CHECK_HIP(hipStreamWaitEvent(ss, h->plan_fork, 0));
        }
    }
    // the block before this one was booked on whatever side stream its call named: order behind it
    PlanSlot &other = h->plan[(h->plan_next + ha_cache::kPlanSlots - 1) % ha_cache::kPlanSlots];
    if (other.count > 0 && other.booked_on != ss)
        HA_CHECK_HIP(hipStreamWaitEvent(ss, other.booked, 0));
    const uint64_t lim = static_cast<uint64_t>(c.length);
    if (key_kind == 0 ? ha_plan_build_batch_f32ids_lim(reinterpret_cast<const float *const *>(keys), n, sl.ws, count, lim, side)
                      : ha_plan_build_batch_u64ids_lim(reinterpret_cast<const uint64_t *const *>(keys), n, sl.ws, count, lim, side))
        return -1;
    BookArgs a;
    memset(&a, 0, sizeof(a));
    a.count = count;
    for (int i = 0; i < count; ++i) {
        PlanPtrs p = plan_layout(sl.ws[i], n[i]);
        a.n[i] = static_cast<int>(n[i]);
        a.hdr[i] = p.hdr;
        a.uniq[i] = p.uniq;
        a.counts[i] = p.counts;
        sl.n[i] = n[i];
    }
    a.it_slot = sl.it_slot; a.it_flag = sl.it_flag; a.it_upd = sl.it_upd;
    a.ev_slot = sl.ev_slot; a.ev_key = sl.ev_key; a.ev_upd = sl.ev_upd;
    a.rec = sl.rec;
    a.xw = h->plan_xw;
    a.nmax = c.nmax;
    if (c.policy == kLRU) {
        hipLaunchKernelGGL(cache_book_block_kernel, dim3(kBookWg), dim3(kBookThreads), 0, ss, c, a);
    } else {
        if (!h->lfu_tree_ok) {
            hipLaunchKernelGGL(cache_lfu_keys_kernel, dim3(static_cast<unsigned>((tree.nblk * kLfuBlk + 255) / 256)), dim3(256), 0, ss,
                               c, tree);
            hipLaunchKernelGGL(cache_lfu_mins_kernel, dim3(static_cast<unsigned>((tree.nblk + 255) / 256)), dim3(256), 0, ss, tree);
            h->lfu_tree_ok = true;
        }
        hipLaunchKernelGGL(cache_book_lfu_kernel, dim3(kBookWg), dim3(kBookThreads), 0, ss, c, a, tree);
    }
    {   // the items per sorted position
        PlanExpandPtrs ep;
        int nmx = 1;
        for (int i = 0; i < count; ++i) {
            PlanPtrs p = plan_layout(sl.ws[i], n[i]);
            ep.upos[i] = p.upos;
            ep.perm[i] = p.perm;
            nmx = n[i] > nmx ? static_cast<int>(n[i]) : nmx;
        }
        hipLaunchKernelGGL(cache_plan_expand_kernel, dim3((nmx + 255) / 256, count), dim3(256), 0, ss, a, ep, sl.pos_item,
                           sl.it_upd_pos);
    }
    HA_LAUNCH_CHECK();
    HA_CHECK_HIP(hipEventRecord(sl.booked, ss));
    sl.booked_on = ss;
    sl.count = count;
    sl.next_call = 0;
    sl.waited = false;
    h->plan_next += 1;
    h->plan_n = -1;
    h->same_fast = false;
    h->ring_count = h->ring_head = 0;
    return 0;
}

// the slot and batch index of the next planned call of `type` (0 lookup, 1 update)
static PlanSlot *plan_current(ha_cache *h, int type, int *idx) {
    for (int k = 0; k < ha_cache::kPlanSlots; ++k) {         // the older block first
        PlanSlot &sl = h->plan[(h->plan_next + k) % ha_cache::kPlanSlots];
        if (sl.count > 0 && sl.next_call < 2 * sl.count) {
            if ((sl.next_call & 1) != type)
                return nullptr;
            *idx = sl.next_call >> 1;
            return &sl;
        }
    }
    return nullptr;
}

extern "C" int ha_cache_lookup_planned(ha_cache *h, int64_t n, float *dest, ha_stream_t stream) {
    HA_REQUIRE(h, "cache_lookup_planned: null handle");
    int i = 0;
    PlanSlot *sl = plan_current(h, 0, &i);
    HA_REQUIRE(sl != nullptr, "cache_lookup_planned: no planned batch is due for its lookup (ha_cache_plan_block; lookup and "
               "update alternate)");
    HA_REQUIRE(sl->n[i] == n && (n == 0 || dest), "cache_lookup_planned: the planned batch has %ld keys (got %ld)", (long)sl->n[i],
               (long)n);
    Cache &c = h->c;
    hipStream_t s = as_stream(stream);
    if (!sl->waited) {
        HA_CHECK_HIP(hipStreamWaitEvent(s, sl->booked, 0));
        sl->waited = true;
    }
    cache_mark(h, kTStart, s, true);
    if (n > 0) {
        PlanPtrs p = plan_layout(sl->ws[i], n);
        const long long at = static_cast<long long>(i) * c.nmax;
        const unsigned blocks = static_cast<unsigned>((n + 15) / 16);
        const bool vec_ok = (c.width % 4 == 0) && (reinterpret_cast<uintptr_t>(dest) % 16 == 0) &&
                            (reinterpret_cast<uintptr_t>(c.table) % 16 == 0);
        if (vec_ok)
            hipLaunchKernelGGL(cache_lookup_planned_kernel<4>, dim3(blocks), dim3(1024), 0, s, c, sl->pos_item + at, (long long)n,
                               dest, sl->pver + at, sl->rec + i);
        else
            hipLaunchKernelGGL(cache_lookup_planned_kernel<1>, dim3(blocks), dim3(1024), 0, s, c, sl->pos_item + at, (long long)n,
                               dest, sl->pver + at, sl->rec + i);
        HA_LAUNCH_CHECK();
    }
    cache_mark(h, kTEnd, s);
    sl->next_call += 1;
    h->last_planned = sl;
    h->last_planned_idx = i;
    h->last_planned_type = 0;
    return 0;
}

extern "C" int ha_cache_update_planned(ha_cache *h, int64_t n, const float *grads, ha_stream_t stream) {
    HA_REQUIRE(h, "cache_update_planned: null handle");
    int i = 0;
    PlanSlot *sl = plan_current(h, 1, &i);
    HA_REQUIRE(sl != nullptr, "cache_update_planned: no planned batch is due for its update (its lookup comes first)");
    HA_REQUIRE(sl->n[i] == n && (n == 0 || grads), "cache_update_planned: the planned batch has %ld keys (got %ld)", (long)sl->n[i],
               (long)n);
    Cache &c = h->c;
    hipStream_t s = as_stream(stream);
    cache_mark(h, kTStart, s, true);
    if (n > 0) {
        PlanPtrs p = plan_layout(sl->ws[i], n);
        const long long at = static_cast<long long>(i) * c.nmax;
        const int apply_blocks = static_cast<int>((n + kPosPerBlock - 1) / kPosPerBlock);
        ApplyMaps maps{};
        maps.dst2 = c.data;
        maps.push_tab = c.table;
        maps.push_rows = static_cast<uint64_t>(c.store_rows);
        maps.pos_item = sl->pos_item + at;
        maps.victim_row = reinterpret_cast<const int *>(&(sl->rec + i)->vh_slot);
        const bool vec_ok = (c.width % 4 == 0) && (reinterpret_cast<uintptr_t>(grads) % 16 == 0) &&
                            (reinterpret_cast<uintptr_t>(c.table) % 16 == 0);
        const dim3 grid(static_cast<unsigned>(apply_blocks + kPlanEvictBlocks + kPlanMetaBlocks));
        if (vec_ok)
            hipLaunchKernelGGL(cache_update_planned_kernel<4>, grid, dim3(1024), kApplyLdsBytes, s, c, p.sorted, p.perm, (int)n,
                               grads, maps, sl->it_upd_pos + at, sl->pver + at, sl->ev_slot + at, sl->ev_key + at, sl->ev_upd + at,
                               sl->rec + i);
        else
            hipLaunchKernelGGL(cache_update_planned_kernel<1>, grid, dim3(1024), kApplyLdsBytes, s, c, p.sorted, p.perm, (int)n,
                               grads, maps, sl->it_upd_pos + at, sl->pver + at, sl->ev_slot + at, sl->ev_key + at, sl->ev_upd + at,
                               sl->rec + i);
        HA_LAUNCH_CHECK();
    }
    cache_mark(h, kTEnd, s);
    sl->next_call += 1;
    if (sl->next_call == 2 * sl->count) {       // the block's last row launch is enqueued
        HA_CHECK_HIP(hipEventRecord(sl->rows_done, s));
        sl->rows_recorded = true;
    }
    h->last_planned = sl;
    h->last_planned_idx = i;
    h->last_planned_type = 1;
    h->evict_empty = true;
    return 0;
}

// `count` planned pairs by ONE call: lookup of the next planned batch into dests[k], its update with grads[k], ... (a caller
// that has the gradient buffers of the pairs at hand: a benchmark loop, a pipeline whose model runs elsewhere)
extern "C" int ha_cache_run_planned_pairs(ha_cache *h, int count, const int64_t *n, float *const *dests,
                                          const float *const *grads, ha_stream_t stream) {
    HA_REQUIRE(h && count >= 0 && (count == 0 || (n && dests && grads)), "cache_run_planned_pairs: bad arguments");
    for (int k = 0; k < count; ++k) {
        if (ha_cache_lookup_planned(h, n[k], dests[k], stream))
            return -1;
        if (ha_cache_update_planned(h, n[k], grads[k], stream))
            return -1;
    }
    return 0;
}

// out[8]: the report of the last planned call, as ha_cache_perf's (synchronises the stream)
int ha::cache_perf_planned(ha_cache *h, int64_t *out_host, hipStream_t s) {
    PlanSlot *sl = h->last_planned;
    const int i = h->last_planned_idx;
    const long long at = static_cast<long long>(i) * h->c.nmax;
    hipLaunchKernelGGL(cache_plan_count_kernel, dim3(1), dim3(1024), 0, s, sl->rec + i, sl->pver + at, sl->pos_item + at);
    PlanRec r;
    long long sticky = 0;
    HA_CHECK_HIP(hipMemcpyAsync(&r, sl->rec + i, sizeof(r), hipMemcpyDeviceToHost, s));
    HA_CHECK_HIP(hipMemcpyAsync(&sticky, &h->c.ctl->fb_timeout, sizeof(sticky), hipMemcpyDeviceToHost, s));
    HA_CHECK_HIP(hipStreamSynchronize(s));
    HA_REQUIRE(sticky == 0, "cache: a bookkeeping launch gave up (code %ld, see ha_cache_state; the cache's state is not to be "
               "trusted)", (long)sticky);
    const int type = h->last_planned_type;
    out_host[0] = type;
    out_host[1] = r.n;
    out_host[2] = r.U;
    out_host[3] = type == 0 ? r.M : r.umiss;
    out_host[4] = type == 0 ? r.pulled : r.npush + r.erep;
    out_host[5] = type == 0 ? 0 : r.erep;
    out_host[6] = r.full;
    out_host[7] = r.size;
    return 0;
}
