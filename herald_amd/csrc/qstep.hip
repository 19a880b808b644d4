// ha_qapply / ha_qplan_batch_* / ha_qqueue_batch: ONE launch per training step, driven by a WORK QUEUE that
// preparation launches -- on a stream of their own, a block of steps at a time -- have built.
//
//   step c (caller's stream):     ha_qapply        [coop items of queue c] [wave items of queue c]
//   every B steps (side stream):  ha_qplan_batch   plans of the B batches two blocks ahead, one workgroup each
//                                 ha_qqueue_batch  queues of the B steps of the next block, two workgroups each
//
// Same contract as ha_step_* (step.hip): batch c is applied in the reference's occurrence order
// (cpu_SGDOptimizerSparseUpdate, src/dnnl_ops/Optimizers.cpp:51-74), the rows of batch c+1 are those AFTER that
// update (cpu_EmbeddingLookup, src/dnnl_ops/EmbeddingLookup.cpp:16-35, run behind it); the ids are known ahead (the
// reference's data loader holds the epoch, its laia scheduler walks it in a thread of its own: dataloader.py:63-98,
// laia/src/laia_scheduler.cc:115-169).  What is different is who does the bookkeeping and when:
//
//   plan   one 1024-thread workgroup groups a whole batch by key in LDS (hash table, ranks in occurrence order) and
//          finishes its plan (unique keys, segment starts, counts, inverse, occurrence lists) -- instead of the 208
//          rank-by-counting workgroups + 7 finish workgroups of step.hip inside every step's launch;
//   queue  two workgroups join the unique keys of two consecutive batches (hash table in LDS), classify every key of
//          their union by its occurrences in the batch to apply (c) and in the batch to look up (m), and write the
//          QUEUE of a step: one 32-byte item per unit of work, heaviest classes first.
//          A plan takes one workgroup ~15 us and a queue ~13-20 us whatever else runs -- longer than the items of a
//          step (11 us) -- so they cannot ride in the step's launch (26 us per step, measured) nor follow the steps one
//          by one on a side stream (a captured graph runs its branches level by level: 24 us).  They are BATCHED instead:
//          the plans of a block of batches side by side in one launch, the queues of a block of steps in another,
//          beside the steps of the block before (~2.6 us of side-stream time per step);
//   workers  one wave per item, no searching, no probing, no waiting: the item names the key, the column slice,
//      where the occurrence indices and the destinations are.  A wave applies its key's gradient rows to the table
//      row it holds in registers, writes the row back and writes it to every output row of the next batch that
//      names the key (so rows both batches touch are never re-read); keys only the next batch names are plain
//      copies (c = 0).  ~2,700 + ~2,000 one-wave items per Criteo step instead of 6,656 + 6,656 + 3,328 waves.
//      The waves of a launch take apply items and copy items ALTERNATELY (qapply_kernel): a compute unit that holds only
//      copies is bound by its store path, one that holds only applies by its load path (13.5 -> 12.3 us per step).
//
// Item classes (c = occurrences to apply, m = destinations to write):
//   S  c <= 3 and m <= 16      one wave, <= 512 columns of the row, 16-byte vectors, ordered chain       (bit-exact)
//   M  c <= 15                 one wave per 128-column slice, 8-byte vectors, ordered chain               (bit-exact)
//   L  16 <= c < 64            one wave per 32-column slice: lane = (occurrence group r of 8, column quad); every lane
//                              sums lr*g over its occurrences r, r+8, ... in order, the eight partial sums are added as
//                              a fixed tree, one subtract                                                  (tolerance)
//   G  c >= 64                 one WORKGROUP per 64-column slice: wave w takes occurrences 16w..16w+15 of every block
//                              of 256, partial sums meet in LDS, fixed tree over the waves, one subtract; all 16 waves
//                              write the destinations                                                      (tolerance)
//                              (QV_GOLD = 0, smaller workgroups: the workgroup's waves on a 32-column slice, each an L item
//                              on its share of every block of occurrences)
//   Z  key beyond the table    zeros to its destinations (the library's definition of such ids)
// "tolerance": the reference subtracts lr*g occurrence by occurrence (two roundings each); L and G subtract a
// deterministic tree sum instead -- within the 1e-5 relative the north star allows for accumulated gradients
// (tests/test_gpu_qstep.py holds both: bit-exact below 16 occurrences, 1e-5 above).  Callers that need the serial
// chain for every run length use ha_step_* / ha_sgd_push_pull_*.
// NOTE (round 6): the launches that overlapped consecutive steps (round 5: a spanning launch, a gated two-stream form, a
// persistent form -- all slower than one launch per step) are gone from the tree; docs/EXPERIMENTS.md round 5 keeps the write-up.
#include <hip/hip_ext.h>

#include "plan_dev.h"
#include "gather_dev.h"

namespace ha {

constexpr int kQMax = 7168;          // ids per batch: keys + two index buffers + counters = 75 KiB of LDS, two workgroups per CU
// Launch geometry.  QV_GOLD = 1 (the product): 1024-thread workgroups, sixteen wave items each; a G item is one WORKGROUP per
// 64-column slice (sixteen waves x four lane groups).  QV_GOLD = 0 (-DQV_WG=256|512): smaller workgroups, a G item = the
// workgroup's 4 / 8 waves on a 32-column slice -- a kernel of the step's traffic WITHOUT bookkeeping prefers those (10.6 vs
// 12.0 us per launch, tools/floor_bench.hip: the items spread evenly over the compute units), the real launch does not: its
// 2,000 small workgroups take 1.3 us to start instead of 0.6, and same-box A/B has 12.5 us per step for 1024 threads against
// 12.75 for 512 and 12.9 for 256 (profiles/r04/ab_workgroup_size.txt).
#ifndef QV_GOLD
#define QV_GOLD 1
#endif
#ifndef QV_WG
#define QV_WG (QV_GOLD ? 1024 : 256)
#endif
#ifndef QV_COOPSLOTS
#define QV_COOPSLOTS (QV_GOLD ? 64 : 256)
#endif
constexpr int kQWg = QV_WG;                    // threads per workgroup of the apply launch
constexpr int kQWpw = kQWg / 64;               // waves (= wave items) per workgroup
constexpr int kQCoopSlots = QV_COOPSLOTS;      // workgroups reserved for G items (they loop if there are more)
constexpr int kQWorkerMax = 448 * 16 / kQWpw;  // worker workgroups: with the coop slots the launch stays below the
                                               // chip's 8,192 resident waves; the waves loop beyond that
constexpr int kQSmallC = 3, kQSmallM = 16, kQMediumC = 15, kQLongC = 64;
enum QKind { kQS = 0, kQM = 1, kQL = 2, kQZ = 3, kQG = 4, kQNone = 15 };

struct QHeader {
    uint32_t n_wave, n_coop;                       // items of the keys the batch to apply names (one workgroup writes them)
    uint32_t n_long, n_medium, n_small;            //   per class (diagnostics, tests)
    uint32_t n_copy, n_copy_medium, n_copy_small;  // items of the keys only the lookup names (another workgroup)
    uint32_t overflow_wave, overflow_copy;         // 1 = the builder counted more items than the queue holds (never, by the
                                                   // layout's bounds; checked by the host so that it could only fail loudly)
    // "this queue is complete": the step's EPOCH (a non-zero tag of the step it was built for), written by the builder's
    // two parts behind a device-scope release once their items are in place.  An apply launch that is given the epoch
    // of its step checks both words before it reads an item (ha_qapply_steps_sync): callers that order the two streams
    // without a wait on the apply's stream rest on it.  `done`: wide path, bucket workgroups that have finished.
    uint32_t epoch_wave, epoch_copy, done;
    // the apply launch's ONE decision about this queue (sync = "flags"): workgroup 0 resolves the epoch words -- there at the
    // first look, or polled for up to 2 s -- and publishes the step's epoch (go) or its complement (the launch gave up:
    // NOTHING of the step is applied); a workgroup that does not find the tag at its first look follows this word, never
    // the tag itself (ADVICE round 4 / verdict round 5: a timed-out step must not be partly applied)
    uint32_t verdict;
    uint32_t reserved[50];
};
static_assert(sizeof(QHeader) == 256, "queue header is one 256-byte line");
struct QEntry {
    uint32_t w[8];   // kind | col0/4 << 4, key, c, first position of the key's occurrence list (apply batch), m, first
                     // position of its destination list (lookup batch), the first three occurrence indices as 21-bit
                     // fields of the last two words (batches of up to 2^21 ids; the slice's width follows from kind / col0)
};
constexpr uint32_t kQOccMask = 0x1FFFFFu;
__host__ __device__ __forceinline__ uint2 q_occ_pack(uint32_t o0, uint32_t o1, uint32_t o2) {
    const unsigned long long v = static_cast<unsigned long long>(o0 & kQOccMask) |
                                 (static_cast<unsigned long long>(o1 & kQOccMask) << 21) |
                                 (static_cast<unsigned long long>(o2 & kQOccMask) << 42);
    return uint2{static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32)};
}
static_assert(sizeof(QEntry) == 32, "queue items are 32 bytes");

struct QLayout {
    QHeader *hdr;
    QEntry *coop, *wave, *copy;
    float *part;           // [cap_coop * 64] partial sums of chunked workgroup items (one 64-column slice each)
    uint32_t *pcnt;        // [cap_coop] chunks of a (key, slice) that have delivered theirs (at the index of its chunk 0)
    uint32_t cap_coop, cap_wave, cap_copy;
    size_t bytes;
};
constexpr uint32_t kQChunk = 256;      // occurrences per workgroup item: a key with more is cut into CHUNKS (QV_GOLD)
__host__ __device__ __forceinline__ uint32_t q_nchunk(uint32_t c) {
    return (QV_GOLD && c > kQChunk) ? (c + kQChunk - 1u) / kQChunk : 1u;
}
static inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }
// Bounds.  A key with c occurrences in the batch to apply and m in the batch to look up has per512 items (S), per128 <=
// 4 per512 (M: c >= 4 or m >= 17) or per32 <= 16 per512 (L: c >= 16): never more than per512 * (c + m) -- but NOT
// per512 * c (a key with c = 1, m = 17 is an M item: 4 items for one position of the batch to apply).  Hence
// wave items <= ceil(width/512) * (n_a + n_g), copy items <= ceil(width/512) * n_g, coop items <= ceil(width/32) * n_a / 64.
static inline QLayout queue_layout(void *ws, int64_t n_cap, int64_t width) {
    QLayout q;
    char *b = static_cast<char *>(ws);
    q.cap_coop = static_cast<uint32_t>(ceil_div(width, 32) * (ceil_div(n_cap, kQLongC) + 1));     // (QV_GOLD needs half)
    q.cap_wave = static_cast<uint32_t>(ceil_div(width, 512) * 2 * n_cap + 64);
    q.cap_copy = static_cast<uint32_t>(ceil_div(width, 512) * n_cap + 64);
    q.hdr = reinterpret_cast<QHeader *>(b);
    q.coop = reinterpret_cast<QEntry *>(b ? b + sizeof(QHeader) : nullptr);
    q.wave = q.coop ? q.coop + q.cap_coop : nullptr;
    q.copy = q.wave ? q.wave + q.cap_wave : nullptr;
    const size_t items = sizeof(QHeader) + (static_cast<size_t>(q.cap_coop) + q.cap_wave + q.cap_copy) * sizeof(QEntry);
    q.part = reinterpret_cast<float *>(b ? b + items : nullptr);
    q.pcnt = reinterpret_cast<uint32_t *>(b ? b + items + static_cast<size_t>(q.cap_coop) * 256 : nullptr);
    q.bytes = items + static_cast<size_t>(q.cap_coop) * (256 + 4);
    return q;
}

struct QPlan {   // what the roles read / write of a plan workspace
    PlanHeader *hdr;
    uint32_t *keys, *sorted, *uniq;
    int32_t *perm, *inverse, *counts, *seg, *upos;
    uint32_t *occ;     // [2 * n]: per group, the first three occurrence indices packed by q_occ_pack (A writes, B reads)
    int n;
};
static inline QPlan qplan(void *ws, int64_t n) {
    QPlan q;
    memset(&q, 0, sizeof(q));
    if (ws == nullptr || n <= 0)
        return q;
    PlanPtrs p = plan_layout(ws, n);
    q.hdr = p.hdr; q.keys = p.keys; q.sorted = p.sorted; q.uniq = p.uniq;
    q.perm = p.perm; q.inverse = p.inverse; q.counts = p.counts; q.seg = p.seg; q.upos = p.upos;
    q.occ = p.keys_alt;     // keys_alt and perm_alt are adjacent scratch arrays of n words each
    q.n = static_cast<int>(n);
    return q;
}

struct QArgs {
    float *table;
    uint64_t rows;
    int width;
    // workers: queue of this launch, occurrence indices of the batch to apply, destinations of the batch to look up
    const QHeader *qh;
    const QEntry *qcoop, *qwave, *qcopy;
    float *qpart;              // partial sums of chunked workgroup items (queue_layout)
    uint32_t *qpcnt;
    uint32_t cap_coop, cap_wave, cap_copy;
    const int32_t *perm_a;
    int n_a;
    const float *grads;
    float lr;
    const int32_t *perm_g;
    int n_g;
    float *out;
    int ncoop, nworker;
    uint32_t epoch;            // 0: no check; else the tag queue `qh` must carry before an item is read
    uint32_t *err;             // pinned host word raised (8) when the queue never became ready (may be NULL)
    unsigned long long *dbg;   // tools/qstep_timeline.py: {start, end, role | xcc << 8, item kind} per wave
};

// development aid: phase time stamps of the single-workgroup roles (thread 0; ph = nullptr in production)
__device__ __forceinline__ void q_phase(unsigned long long *ph, int k) {
    if (ph != nullptr && threadIdx.x == 0)
        ph[k] = __builtin_amdgcn_s_memrealtime();
}

// ---- exclusive scan over the NW waves of a group of threads (two workgroup barriers); s_w = NW words of the group, w = the
// calling wave's index inside it.  A 1024-thread workgroup is one group of 16 waves, or -- the wide path's small buckets --
// four groups of 4 waves that walk the same phases side by side (every group calls with its own s_w).
template <int NW>
__device__ __forceinline__ uint32_t qscan_n(uint32_t v, uint32_t *s_w, uint32_t *total, int w) {
    const int lane = lane_id();
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o)
            x += y;
    }
    __syncthreads();
    if (lane == 63)
        s_w[w] = x;
    __syncthreads();
    const uint32_t mine = s_w[lane & (NW - 1)];
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t t = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine), k));
        woff += k < w ? t : 0u;
        tot += t;
    }
    *total = tot;
    return woff + x - v;
}

// =====================================================================================================
// Role A: the plan of one batch by ONE workgroup, everything in LDS -- by HASHING, not by sorting.
//
// A stable LSD radix sort of 7,168 keys by one workgroup costs 20+ us (three passes of 4-8 us each: measured with
// ballot ranking and with LDS-atomic ranking alike; the LDS unit of one CU is the bound), twice the time the workers
// of a step need.  Nothing on this path needs KEY ORDER: the workers need, per unique key, its occurrences in
// OCCURRENCE order (the reference's serial chain, Optimizers.cpp:65-72) and the join with the next batch.  So:
//   1. every thread claims a slot of an open-addressing table (8,192 words) for each of its keys (ds_cmpst);
//   2. occupied slots are numbered 0..U-1 by a scan: the GROUP of a key (groups are in slot order, not key order);
//   3. ONE wave walks the positions in order, row by row, and takes rank = atomic_add(count[group], 1): the rank of a
//      position among the occurrences of its key.  One wave's LDS operations execute in program order and the lanes of
//      one atomic instruction that hit the same counter are served in ascending lane order (verified per device by
//      lds_atomics_lane_ordered; otherwise the wave ranks each row with a leader loop), so ranks follow positions;
//   4. counts -> segment starts by a scan; perm[seg[group] + rank] = position.
// The workspace is laid out like an index plan (plan_dev.h) and holds the same relations -- uniq / counts / seg /
// inverse / perm / upos / sorted, perm ascending inside every segment -- except that the unique keys are in slot order
// (header word kGroupedFlagWord = 1).  ha_plan_build_* is the entry point for key-ordered plans.
//   s_tab[8192] u32 | s_lab[npad] u16 | s_cnt[npad + 4] u16 | s_rank[npad] u16 | s_w[32] u32
// =====================================================================================================
constexpr int kQTabBits = 13, kQTabSize = 1 << kQTabBits;
constexpr uint32_t kQTabEmpty = 0xFFFFFFFFu;      // keys are <= 0xFFFFFFFE (to_key)
constexpr int kGroupedFlagWord = 12;             // plan header word: 1 = unique keys in slot order, not key order
constexpr int kOrderFlagWord = 13;               // plan header word: 1 = an occurrence list is NOT in position order (the plan
                                                 // workgroup checks every list it writes; the queue builder hands the word on
                                                 // to the host: the serial chain's order is what bit-exactness rests on)
__device__ __forceinline__ uint32_t q_hash(uint32_t key, int bits = kQTabBits) {
    return (key * 0x9E3779B1u) >> (32 - bits);
}
static inline size_t qsort_lds_bytes(int n) {
    const size_t npad = (static_cast<size_t>(n) + 1023) & ~static_cast<size_t>(1023);
    return kQTabSize * 4 + 3 * npad * 2 + 8 + 32 * 4;
}
// the wide path's small buckets: four of them side by side in one workgroup, 256 threads and a 4,096-slot table each
constexpr int kQSubThreads = 256, kQSubTabBits = 12, kQSubMax = 7 * kQSubThreads;      // ids per small bucket
constexpr size_t kQSubPlanLds = (size_t(1) << kQSubTabBits) * 4 + 3 * size_t(kQSubMax) * 2 + 8 + 32 * 4 + 120;   // 27,392
static_assert(kQSubPlanLds % 128 == 0, "the groups' LDS regions stay aligned");

// BUCKET: the batch is one hash bucket of a larger batch (the wide path below): `ids` are its keys in position order,
// pos_map[i] the position of its i-th id in the whole batch -- occurrence lists and the groups' first occurrences leave
// as positions of the whole batch, and only what the queue builder and the apply read is written (unique keys, counts,
// segment starts, occurrence lists, first occurrences).
// NT / TBITS: the threads that work on the batch and the bits of its table -- the whole workgroup and 8,192 slots, or (wide path,
// buckets of at most 7 * 256 ids) a QUARTER of it and 4,096 slots: four buckets are then grouped side by side by one
// workgroup, every barrier being one all four pass (`lds` = the group's own region, kQSubPlanLds apart).
template <typename IdT, bool RANK_ATOMIC, bool BUCKET = false, int NT = 1024, int TBITS = kQTabBits>
__device__ __forceinline__ void qsort_finish_body(const IdT *__restrict__ ids, const QPlan &p, uint32_t *lds,
                                                  unsigned long long *ph = nullptr,
                                                  const uint32_t *__restrict__ pos_map = nullptr) {
    constexpr int TS = 1 << TBITS, SPT = TS / NT;     // table slots, slots numbered per thread
    const int n = p.n;
    const int tid = static_cast<int>(threadIdx.x) & (NT - 1), lane = lane_id(), w = tid >> 6;
    const int npad = (n + NT - 1) & ~(NT - 1);
    const int P = npad / NT;                      // positions per thread, <= 7
    uint32_t *s_tab = lds;
    uint16_t *s_lab = reinterpret_cast<uint16_t *>(lds + TS);
    uint16_t *s_cnt = s_lab + npad;
    uint16_t *s_rank = s_cnt + npad + 4;      // (a spare counter behind the groups' for positions beyond the batch)
    uint32_t *s_w = reinterpret_cast<uint32_t *>(s_rank + npad);
    q_phase(ph, 0);
    // keys of positions tid, tid + NT, ... stay in registers; the table and the counters are cleared meanwhile
    uint32_t key[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)
        key[k] = to_key<IdT>(ids[max(min(tid + k * NT, n - 1), 0)]);
    for (int i = tid; i < TS; i += NT)
        s_tab[i] = kQTabEmpty;
    for (int i = tid; i < npad / 2 + 2; i += NT)
        reinterpret_cast<uint32_t *>(s_cnt)[i] = 0;
    __syncthreads();
    q_phase(ph, 1);
    // 1. claim a slot per key: the first attempts of a thread's keys go out back to back, collisions are walked after
    {
        uint32_t h[7], seen[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            h[k] = q_hash(key[k], TBITS);
            seen[k] = key[k];
            if (k < P && tid + k * NT < n) {
                // (a bucket of the wide path may hold one key thousands of times: same-address atomics are served one
                // lane at a time, a plain read of a slot that already holds the key is not)
                if (BUCKET && k > 0 && s_tab[h[k]] == key[k])
                    continue;
                seen[k] = kQTabEmpty;
                if (__hip_atomic_compare_exchange_strong(s_tab + h[k], &seen[k], key[k], __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    seen[k] = key[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = tid + k * NT;
            if (k < P && i < n) {
                while (seen[k] != key[k]) {       // somebody else's key sits here: next slot
                    h[k] = (h[k] + 1) & (TS - 1);
                    seen[k] = kQTabEmpty;
                    if (__hip_atomic_compare_exchange_strong(s_tab + h[k], &seen[k], key[k], __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                        seen[k] = key[k];
                }
                s_lab[i] = static_cast<uint16_t>(h[k]);
            }
        }
    }
    __syncthreads();
    q_phase(ph, 2);
    // 2. number the occupied slots (thread t: slots SPT t .. SPT t + SPT - 1); the table then maps slot -> group
    uint32_t U;
    {
        uint32_t kk[SPT], occ = 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            kk[j] = s_tab[tid * SPT + j];
            occ += kk[j] != kQTabEmpty;
        }
        uint32_t g = qscan_n<NT / 64>(occ, s_w, &U, w);
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            if (kk[j] != kQTabEmpty) {
                p.uniq[g] = kk[j];
                s_tab[tid * SPT + j] = g;
                ++g;
            }
        }
    }
    __syncthreads();
    q_phase(ph, 3);
    uint32_t lab[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int i = tid + k * NT;
        lab[k] = 0;
        if (k < P && i < n) {
            lab[k] = s_tab[s_lab[i]];
            s_lab[i] = static_cast<uint16_t>(lab[k]);
        }
    }
    __syncthreads();
    q_phase(ph, 4);
    // 3. ranks in position order: wave 0, eight rows per trip -- all reads, then all atomics, then the extraction and
    // the stores (branch-free: a position beyond the batch counts on a spare counter, so no wait sits between atomics)
    if (w == 0) {
        uint32_t *cnt32 = reinterpret_cast<uint32_t *>(s_cnt);
        const uint32_t spare = static_cast<uint32_t>(npad);      // s_cnt has npad + 2 entries
        uint32_t nx[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            nx[r] = s_lab[min(r * 64 + lane, npad - 1)];
        for (int r0 = 0; r0 < npad; r0 += 8 * 64) {
            uint32_t lb[8], rk[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = r0 + r * 64 + lane;
                lb[r] = i < n ? nx[r] : spare;
            }
            // the labels of the next trip are on their way while this trip's atomics run
#pragma unroll
            for (int r = 0; r < 8; ++r)
                nx[r] = s_lab[min(r0 + 8 * 64 + r * 64 + lane, npad - 1)];
            if (RANK_ATOMIC) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    rk[r] = __hip_atomic_fetch_add(cnt32 + (lb[r] >> 1), 1u << ((lb[r] & 1u) * 16u), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    rk[r] = (rk[r] >> ((lb[r] & 1u) * 16u)) & 0xFFFFu;
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    // leader loop: the lanes of one group at a time, lowest lane first
                    rk[r] = 0;
                    unsigned long long todo = ~0ull;
                    while (todo) {
                        const int lead = __builtin_ctzll(todo);
                        const uint32_t gl = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lb[r]), lead));
                        const unsigned long long same = __ballot(lb[r] == gl);
                        const uint32_t base = s_cnt[gl];
                        if (lb[r] == gl)
                            rk[r] = base + __builtin_popcountll(same & ((1ull << lane) - 1ull));
                        if (lane == lead)
                            s_cnt[gl] = static_cast<uint16_t>(base + __builtin_popcountll(same));
                        todo &= ~same;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = r0 + r * 64 + lane;
                if (i < n)
                    s_rank[i] = static_cast<uint16_t>(rk[r]);
            }
        }
    } else if (!BUCKET) {
        // the other waves write what is known already
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = tid + k * NT;
            if (k < P && i < n) {
                p.keys[i] = key[k];
                p.inverse[i] = static_cast<int32_t>(lab[k]);
            }
        }
    }
    __syncthreads();
    if (w == 0 && !BUCKET) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = tid + k * NT;
            if (k < P && i < n) {
                p.keys[i] = key[k];
                p.inverse[i] = static_cast<int32_t>(lab[k]);
            }
        }
    }
    q_phase(ph, 5);
    // 4. counts -> segment starts (thread t: groups G t .. G t + G - 1, G = npad / NT); s_cnt then holds the starts
    {
        uint32_t c[7], sum = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint32_t g = tid * P + j;
            c[j] = (j < P && g < U) ? s_cnt[g] : 0u;
            sum += c[j];
        }
        uint32_t all;
        uint32_t at = qscan_n<NT / 64>(sum, s_w, &all, w);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint32_t g = tid * P + j;
            if (j < P && g < U) {
                s_cnt[g] = static_cast<uint16_t>(at);
                p.seg[g] = static_cast<int32_t>(at);
                p.counts[g] = static_cast<int32_t>(c[j]);
                at += c[j];
            }
        }
    }
    if (tid == 0) {
        p.hdr->n_unique = U;
        p.hdr->reserved[kGroupedFlagWord] = 1;
        p.hdr->reserved[kOrderFlagWord] = 0;
        p.seg[U] = n;
    }
    __syncthreads();
    q_phase(ph, 6);
    // Outputs.  Every array leaves through LDS so that the global stores are contiguous (scattered 4-byte stores of one
    // CU -- perm, sorted keys, group of a position -- drained for 5-8 us behind the role): keys by position go to the
    // table's space, perm and the group of every grouped position to the label / rank arrays.
    {
        uint32_t q[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = tid + k * NT;
            q[k] = (k < P && i < n) ? static_cast<uint32_t>(s_cnt[lab[k]]) + s_rank[i] : 0u;
        }
        __syncthreads();      // labels, ranks and the table have been read by everyone
        uint16_t *s_perm = s_lab, *s_gq = s_rank;
        uint32_t *s_key = s_tab;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = tid + k * NT;
            if (k < P && i < n) {
                s_perm[q[k]] = static_cast<uint16_t>(i);
                s_gq[q[k]] = static_cast<uint16_t>(lab[k]);
                s_key[i] = key[k];
            }
        }
        __syncthreads();
        bool disorder = false;
        for (int qq = tid; qq < n; qq += NT) {
            const uint32_t i = s_perm[qq];
            if (BUCKET) {
                p.perm[qq] = static_cast<int32_t>(pos_map[i]);
            } else {
                p.perm[qq] = static_cast<int32_t>(i);
                p.sorted[qq] = s_key[i];
                p.upos[qq] = static_cast<int32_t>(s_gq[qq]);
            }
            // the ranking rests on one wave-instruction's same-address LDS atomics being served in lane order (probed per
            // device, lds_atomics_lane_ordered); every list is checked as it leaves, so a pattern the probe did not
            // sample cannot silently reorder a chain
            disorder |= qq > 0 && s_gq[qq] == s_gq[qq - 1] && i <= s_perm[qq - 1];
        }
        if (disorder)
            p.hdr->reserved[kOrderFlagWord] = 1;
        // the first three occurrences of every key also go to the group's own words (the queue builder embeds them in
        // small items; entries beyond the group's count are never used)
        for (uint32_t g = tid; g < U; g += NT) {
            const int st = s_cnt[g];
            uint32_t o0 = s_perm[min(st, n - 1)], o1 = s_perm[min(st + 1, n - 1)], o2 = s_perm[min(st + 2, n - 1)];
            if (BUCKET) {
                o0 = pos_map[o0];
                o1 = pos_map[o1];
                o2 = pos_map[o2];
            }
            reinterpret_cast<uint2 *>(p.occ)[g] = q_occ_pack(o0, o1, o2);
        }
    }
    q_phase(ph, 7);
}

// =====================================================================================================
// Role B: the queue of the NEXT launch from the finished plans of its two batches.
// =====================================================================================================
// the table + the keys in it (+, where it fits, the lookup groups' packed count / start)
static inline size_t qjoin_lds_bytes(int n_a, int n_g) {
    return kQTabSize * 4 + static_cast<size_t>(n_a > n_g ? n_a : n_g) * 4 + 32 * 4;
}
static inline size_t qjoin_lds_resident_bytes(int n_g) {
    return kQTabSize * 4 + static_cast<size_t>(n_g) * 8 + 32 * 4;
}

__device__ __forceinline__ int q_kind(uint32_t c, uint32_t m, bool in_table) {
    if (!in_table)
        return m > 0 ? kQZ : kQNone;
    if (c >= static_cast<uint32_t>(kQLongC))
        return kQG;
    if (c > static_cast<uint32_t>(kQMediumC))
        return kQL;
    if (c > static_cast<uint32_t>(kQSmallC) || m > static_cast<uint32_t>(kQSmallM))
        return kQM;
    return kQS;
}
__device__ __forceinline__ int q_slice(int kind) {   // columns per item
    return kind == kQL ? 32 : kind == kQG ? (QV_GOLD ? 64 : 32) : kind == kQM ? 128 : 512;
}

__device__ __forceinline__ void q_st_item(QEntry *dst, uint4 lo, uint4 hi) {
    float4v a, b;
    __builtin_memcpy(&a, &lo, 16);
    __builtin_memcpy(&b, &hi, 16);
    st4_sc1(reinterpret_cast<float *>(dst), a);
    st4_sc1(reinterpret_cast<float *>(dst) + 4, b);
}
__device__ __forceinline__ void q_emit_words(QEntry *dst, int kind, uint32_t key, uint32_t c, uint32_t st, uint32_t m,
                                             uint32_t fs, int width, uint32_t o01, uint32_t o23, uint32_t *pcnt = nullptr) {
    const int slice = q_slice(kind);
    if (kind == kQZ)
        c = 0;    // ids beyond the table are never applied; their destinations get zeros
    int j = 0;
    const uint32_t nch = kind == kQG ? q_nchunk(c) : 1u;
    if (nch > 1u) {
        // a key with more than kQChunk occurrences: per slice `nch` consecutive items, chunk ch = occurrences [256 ch, ..);
        // whichever of them finishes last adds the partial sums up (in chunk order) and writes the row and the destinations
        for (int col0 = 0; col0 < width; col0 += slice) {
            __hip_atomic_store(pcnt + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t ch = 0; ch < nch; ++ch, ++j) {
                uint4 lo, hi;
                lo.x = static_cast<uint32_t>(kind) | (static_cast<uint32_t>(col0 / 4) << 4);
                lo.y = key;
                lo.z = c;
                lo.w = st;
                hi.x = m;
                hi.y = fs;
                hi.z = ch | (nch << 16);
                hi.w = 0;
                q_st_item(dst + j, lo, hi);
            }
        }
        return;
    }
    if (kind == kQG)
        o01 = 0;     // (one chunk)
    for (int col0 = 0; col0 < width; col0 += slice, ++j) {
        const int cols = min(slice, width - col0);
        uint4 lo, hi;
        lo.x = static_cast<uint32_t>(kind) | (static_cast<uint32_t>(col0 / 4) << 4);
        lo.y = key;
        lo.z = c;
        (void)cols;
        lo.w = st;
        hi.x = m;
        hi.y = fs;
        hi.z = o01;
        hi.w = o23;
        q_st_item(dst + j, lo, hi);      // (through the L2, like everything an apply launch may read before this one ends)
    }
}
// (helper below) an item's two 16-byte halves, written THROUGH the L2 (`sc1`): once the writing thread's `s_waitcnt
// vmcnt(0)` has passed they are in memory, so the builder can publish "this queue is complete" with a relaxed device-scope
// atomic instead of a release fence -- an agent-scope release writes back all the dirty lines of the XCD's L2, per
// workgroup, and the wide path has 2 x 128 builder workgroups per step
struct QCount {   // items per class: coop (G), long, medium, small (S and Z)
    uint32_t g, l, m, s;
};
// (selects on VALUES: an if / else chain over four variables is merged into `*select(&x..) += n`, which puts them --
// and, through the by-reference captures of a lambda, the whole argument block -- into scratch memory)
__device__ __forceinline__ QCount q_count(QCount t, int kind, uint32_t per512, uint32_t per128, uint32_t per32, uint32_t c) {
    t.g += kind == kQG ? (QV_GOLD ? (per32 + 1u) / 2u : per32) * q_nchunk(c) : 0u;
    t.l += kind == kQL ? per32 : 0u;
    t.m += kind == kQM ? per128 : 0u;
    t.s += (kind == kQS || kind == kQZ) ? per512 : 0u;
    return t;
}

// Hash join of the two batches' groups by TWO workgroups, everything after the first trips to memory in LDS:
//   part 0  the groups of the batch to LOOK UP (g) enter an open-addressing table (slot -> group index; their keys and --
//           where it fits -- their packed count | start << 16 sit beside it); every group of the batch to APPLY (a)
//           probes it: a hit yields the destinations (m, fs).  Items of every key the batch to apply names.
//   part 1  the keys of the batch to apply enter the table; every group of the lookup batch probes it: a MISS is a key
//           only the lookup names -- a pure copy (or zeros, beyond the table).  Items of the copies, in their own
//           region of the queue.
// Each part: classes, scans, items (thread by thread in the order they were counted).  Two passes over the probing side,
// each one batch of independent loads (branch-free: 32-bit unsigned clamped indices, scalar base + one offset register
// per round -- signed or predicated indices cost a 64-bit address pair per load) and a probe per group: pass 1 counts
// the items per class, pass 2 -- behind the scans -- reads the groups again with everything an item needs and probes
// again.  Nothing but the running item numbers lives across the scans (holding five words per group and round did:
// the role spilled at the kernel's 64 registers).
struct QJoin {      // what every part of the join needs (passed by reference to inlined code only)
    QPlan pa, pg;
    uint64_t rows;
    int width, Ua, Ug, Ub;      // Ub: keys in the table
    bool cs_res;
    uint32_t *s_tab, *s_bk, *s_cs, *s_w;
    QHeader *bqh;
    QEntry *bcoop, *bwave, *bcopy;
    uint32_t *bpcnt;                // the queue's chunk counters (they follow the item regions: queue_layout)
    uint32_t bcap_coop, bcap_wave, bcap_copy, per512, per128, per32;
    uint32_t st_base, fs_base;      // wide path: where this bucket's occurrence / destination lists start in the batch's
    uint32_t tid, nt;               // the thread's index in its group of nt threads (the workgroup, or a quarter of it)
    int tbits;                      // the table has 1 << tbits slots
};

// group index of `key` in the table, or 0xFFFFFFFF
__device__ __forceinline__ uint32_t qjoin_probe(const QJoin &j, uint32_t key) {
    uint32_t h = q_hash(key, j.tbits);
    const uint32_t mask = (1u << j.tbits) - 1u;
    for (;;) {
        const uint32_t e = j.s_tab[h];
        if (e == kQTabEmpty || j.s_bk[e] == key)
            return e;
        h = (h + 1) & mask;
    }
}
__device__ __forceinline__ uint32_t qjoin_dest(const QJoin &j, uint32_t e) {   // m | fs << 16 of lookup group e
    if (e == kQTabEmpty)
        return 0u;
    return j.cs_res ? j.s_cs[e] : (static_cast<uint32_t>(j.pg.counts[e]) | (static_cast<uint32_t>(j.pg.seg[e]) << 16));
}
__device__ __forceinline__ QCount qjoin_emit_one(const QJoin &j, QCount b, QEntry *wave, uint32_t cap_wave, uint32_t key,
                                                 uint32_t c, uint32_t st, uint32_t m, uint32_t fs, uint32_t o01, uint32_t o2) {
    const int kind = q_kind(c, m, key < j.rows);
    if (kind == kQNone)
        return b;
    const uint32_t at = kind == kQG ? b.g : kind == kQL ? b.l : kind == kQM ? b.m : b.s;
    const QCount nb = q_count(b, kind, j.per512, j.per128, j.per32, c);
    const uint32_t cnt = (nb.g - b.g) + (nb.l - b.l) + (nb.m - b.m) + (nb.s - b.s);
    // the layout's bound makes the test always true; never write beyond the queue
    if (at + cnt <= (kind == kQG ? j.bcap_coop : cap_wave))
        q_emit_words((kind == kQG ? j.bcoop : wave) + at, kind, key, c, st + j.st_base, m, fs + j.fs_base, j.width, o01, o2,
                     kind == kQG ? j.bpcnt + at : nullptr);
    return nb;
}

// ---- part 0: groups of the batch to apply, thread t of the group: groups t, t + nt, ... --------------------------------------
template <int R0, int R1>
__device__ __forceinline__ QCount qjoin_count_groups(const QJoin &j, QCount t) {
    const uint32_t tid = j.tid, last = static_cast<uint32_t>(max(j.Ua, 1) - 1);
    uint32_t ka[R1 - R0], c[R1 - R0];
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const uint32_t x = min(static_cast<uint32_t>(r) * j.nt + tid, last);
        ka[r - R0] = j.pa.uniq[x];
        c[r - R0] = static_cast<uint32_t>(j.pa.counts[x]);
    }
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        if (static_cast<int>(static_cast<uint32_t>(r) * j.nt + tid) < j.Ua) {
            const uint32_t mf = qjoin_dest(j, qjoin_probe(j, ka[r - R0]));
            t = q_count(t, q_kind(c[r - R0], mf & 0xFFFFu, ka[r - R0] < j.rows), j.per512, j.per128, j.per32, c[r - R0]);
        }
    }
    return t;
}
template <int R0, int R1>
__device__ __forceinline__ QCount qjoin_emit_groups(const QJoin &j, QCount b) {
    const uint32_t tid = j.tid, last = static_cast<uint32_t>(max(j.Ua, 1) - 1);
    uint32_t ka[R1 - R0], cst[R1 - R0], o01[R1 - R0], o2[R1 - R0];
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const uint32_t x = min(static_cast<uint32_t>(r) * j.nt + tid, last);
        const uint2 oc = reinterpret_cast<const uint2 *>(j.pa.occ)[x];
        ka[r - R0] = j.pa.uniq[x];
        cst[r - R0] = static_cast<uint32_t>(j.pa.counts[x]) | (static_cast<uint32_t>(j.pa.seg[x]) << 16);
        o01[r - R0] = oc.x;
        o2[r - R0] = oc.y;
    }
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        if (static_cast<int>(static_cast<uint32_t>(r) * j.nt + tid) < j.Ua) {
            const uint32_t mf = qjoin_dest(j, qjoin_probe(j, ka[r - R0]));
            b = qjoin_emit_one(j, b, j.bwave, j.bcap_wave, ka[r - R0], cst[r - R0] & 0xFFFFu, cst[r - R0] >> 16,
                               mf & 0xFFFFu, mf >> 16, o01[r - R0], o2[r - R0]);
        }
    }
    return b;
}

// ---- part 1: groups of the lookup batch, the ones the table (keys of the batch to apply) does not hold ------------
template <int R0, int R1>
__device__ __forceinline__ QCount qjoin_count_copies(const QJoin &j, QCount t) {
    const uint32_t tid = j.tid, last = static_cast<uint32_t>(max(j.Ug, 1) - 1);
    uint32_t kg[R1 - R0], m[R1 - R0];
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const uint32_t y = min(static_cast<uint32_t>(r) * j.nt + tid, last);
        kg[r - R0] = j.pg.uniq[y];
        m[r - R0] = static_cast<uint32_t>(j.pg.counts[y]);
    }
#pragma unroll
    for (int r = R0; r < R1; ++r)
        if (static_cast<int>(static_cast<uint32_t>(r) * j.nt + tid) < j.Ug && qjoin_probe(j, kg[r - R0]) == kQTabEmpty)
            t = q_count(t, q_kind(0u, m[r - R0], kg[r - R0] < j.rows), j.per512, j.per128, j.per32, 0u);
    return t;
}
template <int R0, int R1>
__device__ __forceinline__ QCount qjoin_emit_copies(const QJoin &j, QCount b) {
    const uint32_t tid = j.tid, last = static_cast<uint32_t>(max(j.Ug, 1) - 1);
    uint32_t kg[R1 - R0], mfs[R1 - R0];
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const uint32_t y = min(static_cast<uint32_t>(r) * j.nt + tid, last);
        kg[r - R0] = j.pg.uniq[y];
        mfs[r - R0] = static_cast<uint32_t>(j.pg.counts[y]) | (static_cast<uint32_t>(j.pg.seg[y]) << 16);
    }
#pragma unroll
    for (int r = R0; r < R1; ++r)
        if (static_cast<int>(static_cast<uint32_t>(r) * j.nt + tid) < j.Ug && qjoin_probe(j, kg[r - R0]) == kQTabEmpty)
            b = qjoin_emit_one(j, b, j.bcopy, j.bcap_copy, kg[r - R0], 0u, 0u, mfs[r - R0] & 0xFFFFu, mfs[r - R0] >> 16,
                               0u, 0u);
    return b;
}

// WIDE: pa / pg are ONE hash bucket of two larger batches (the wide path below); the workgroup appends its items to the
// step's queue behind those of the other buckets (one atomic add per region on the header's counters, zeroed before
// the launch), st_base / fs_base = where the bucket's occurrence / destination lists start in the batches' lists.
// NT / TBITS as in qsort_finish_body: the whole workgroup joins one pair of plans, or (WIDE, buckets of at most 2,048 unique
// keys each side) every quarter of it joins a pair of its own, side by side through the same barriers.  signal: this pair
// counts towards the step's `wide_parts` (false: a quarter without a bucket).
template <bool WIDE = false, int NT = 1024, int TBITS = kQTabBits>
__device__ __forceinline__ void qjoin_body(const QPlan pa, const QPlan pg, const uint64_t rows, const int width,
                                           QHeader *bqh, QEntry *bcoop, QEntry *bwave, QEntry *bcopy,
                                           const uint32_t bcap_coop, const uint32_t bcap_wave, const uint32_t bcap_copy,
                                           uint32_t *lds, uint32_t lds_bytes, const int part,
                                           unsigned long long *ph = nullptr, uint32_t *mirror = nullptr,
                                           const uint32_t st_base = 0, const uint32_t fs_base = 0,
                                           const uint32_t epoch = 0, const uint32_t wide_parts = 0, const bool signal = true) {
    constexpr int TS = 1 << TBITS;
    const int tid = static_cast<int>(threadIdx.x) & (NT - 1), w = tid >> 6;
    q_phase(ph, 0);
    QJoin j;
    j.tid = static_cast<uint32_t>(tid);
    j.nt = NT;
    j.tbits = TBITS;
    j.pa = pa;
    j.pg = pg;
    j.st_base = st_base;
    j.fs_base = fs_base;
    j.rows = rows;
    j.width = width;
    // (wave-uniform values read through the vector memory path: back to scalar registers, or every pointer and
    // bound derived from them lives in vector registers and every branch on them becomes a divergent one)
    j.Ua = uniform(pa.n > 0 ? static_cast<int>(pa.hdr->n_unique) : 0);
    j.Ug = uniform(pg.n > 0 ? static_cast<int>(pg.hdr->n_unique) : 0);
    j.Ub = part == 0 ? j.Ug : j.Ua;
    j.cs_res = part == 0 && static_cast<uint32_t>(TS * 4 + j.Ug * 8 + 32 * 4) <= lds_bytes;   // uniform over the group
    j.s_tab = lds;
    j.s_bk = lds + TS;
    j.s_cs = j.s_bk + j.Ub;
    j.s_w = j.s_bk + (j.cs_res ? 2 * j.Ub : j.Ub);
    j.bqh = bqh;
    j.bcoop = bcoop;
    j.bwave = bwave;
    j.bcopy = bcopy;
    j.bcap_coop = bcap_coop;
    // (the regions are contiguous: coop | wave | copy | partial sums | chunk counters)
    j.bpcnt = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(bcoop + (static_cast<size_t>(bcap_coop) + bcap_wave + bcap_copy)) +
                                           static_cast<size_t>(bcap_coop) * 256);
    j.bcap_wave = bcap_wave;
    j.bcap_copy = bcap_copy;
    j.per512 = (width + 511) / 512;
    j.per128 = (width + 127) / 128;
    j.per32 = (width + 31) / 32;
    // the table: keys (+ packed count / start of a lookup group) to LDS, then every key claims a slot
    const uint32_t *bkeys = part == 0 ? pg.uniq : pa.uniq;
    for (int i = tid; i < TS; i += NT)
        j.s_tab[i] = kQTabEmpty;
    for (int y = tid; y < j.Ub; y += NT) {
        j.s_bk[y] = bkeys[y];
        if (j.cs_res)
            j.s_cs[y] = static_cast<uint32_t>(pg.counts[y]) | (static_cast<uint32_t>(pg.seg[y]) << 16);
    }
    __syncthreads();
    q_phase(ph, 1);
    for (int y = tid; y < j.Ub; y += NT) {
        uint32_t h = q_hash(j.s_bk[y], TBITS);
        for (;;) {
            uint32_t seen = kQTabEmpty;
            if (__hip_atomic_compare_exchange_strong(j.s_tab + h, &seen, static_cast<uint32_t>(y), __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                break;
            h = (h + 1) & (TS - 1);
        }
    }
    __syncthreads();      // the table is complete
    q_phase(ph, 2);
    QCount t{0u, 0u, 0u, 0u};
    if (part == 0) {      // workgroup-uniform everywhere below (a batch that is absent has null plan pointers)
        if (j.Ua > 0)
            t = qjoin_count_groups<0, 4>(j, t);
        if (j.Ua > 4 * NT)
            t = qjoin_count_groups<4, 8>(j, t);
        q_phase(ph, 3);
        uint32_t nL, nM, nS, nG;
        QCount b;
        b.l = qscan_n<NT / 64>(t.l, j.s_w, &nL, w);
        b.m = qscan_n<NT / 64>(t.m, j.s_w, &nM, w) + nL;            // queue order: long, medium, small
        b.s = qscan_n<NT / 64>(t.s, j.s_w, &nS, w) + nL + nM;
        b.g = qscan_n<NT / 64>(t.g, j.s_w, &nG, w);
        if (WIDE) {
            __syncthreads();
            if (tid == 0) {
                const uint32_t tot = nL + nM + nS;
                const uint32_t bw = tot ? atomicAdd(&bqh->n_wave, tot) : 0u;
                const uint32_t bc = nG ? atomicAdd(&bqh->n_coop, nG) : 0u;
                if (nL) atomicAdd(&bqh->n_long, nL);
                if (nM) atomicAdd(&bqh->n_medium, nM);
                if (nS) atomicAdd(&bqh->n_small, nS);
                uint32_t over = (bw + tot > bcap_wave || bc + nG > bcap_coop) ? 1u : 0u;
                if ((pa.n > 0 && pa.hdr->reserved[kOrderFlagWord] != 0) || (pg.n > 0 && pg.hdr->reserved[kOrderFlagWord] != 0))
                    over |= 2u;
                if (over) {
                    atomicOr(&bqh->overflow_wave, over);
                    if (mirror)
                        mirror[3] = over;
                }
                j.s_w[20] = bw;
                j.s_w[21] = bc;
            }
            __syncthreads();
            const uint32_t bw = j.s_w[20], bc = j.s_w[21];
            b.l += bw;
            b.m += bw;
            b.s += bw;
            b.g += bc;
        }
        if (!WIDE && tid == 0) {
            bqh->n_wave = min(nL + nM + nS, bcap_wave);
            bqh->n_coop = min(nG, bcap_coop);
            bqh->n_long = nL;
            bqh->n_medium = nM;
            bqh->n_small = nS;
            uint32_t over = (nL + nM + nS > bcap_wave || nG > bcap_coop) ? 1u : 0u;
            if ((pa.n > 0 && pa.hdr->reserved[kOrderFlagWord] != 0) || (pg.n > 0 && pg.hdr->reserved[kOrderFlagWord] != 0))
                over |= 2u;
            bqh->overflow_wave = over;
            if (mirror) {       // pinned host words: the host sizes the step's launch by them
                mirror[0] = min(nL + nM + nS, bcap_wave) + 1u;      // + 1: 0 = not written yet
                mirror[1] = min(nG, bcap_coop) + 1u;
                if (over)
                    mirror[3] = over;    // sticky: the host never clears it (1 = queue overflow, 2 = occurrence order)
            }
        }
        q_phase(ph, 4);
        if (j.Ua > 0)
            b = qjoin_emit_groups<0, 3>(j, b);
        if (j.Ua > 3 * NT)
            b = qjoin_emit_groups<3, 5>(j, b);
        if (j.Ua > 5 * NT)
            b = qjoin_emit_groups<5, 8>(j, b);
    } else {
        if (j.Ug > 0)
            t = qjoin_count_copies<0, 4>(j, t);
        if (j.Ug > 4 * NT)
            t = qjoin_count_copies<4, 8>(j, t);
        q_phase(ph, 3);
        uint32_t nM, nS;
        QCount b{0u, 0u, 0u, 0u};
        b.m = qscan_n<NT / 64>(t.m, j.s_w, &nM, w);                 // copies: medium (many destinations), then small / zero
        b.s = qscan_n<NT / 64>(t.s, j.s_w, &nS, w) + nM;
        if (WIDE) {
            __syncthreads();
            if (tid == 0) {
                const uint32_t tot = nM + nS;
                const uint32_t bw = tot ? atomicAdd(&bqh->n_copy, tot) : 0u;
                if (nM) atomicAdd(&bqh->n_copy_medium, nM);
                if (nS) atomicAdd(&bqh->n_copy_small, nS);
                if (bw + tot > bcap_copy) {
                    atomicOr(&bqh->overflow_copy, 1u);
                    if (mirror)
                        mirror[3] = 1u;
                }
                j.s_w[20] = bw;
            }
            __syncthreads();
            const uint32_t bw = j.s_w[20];
            b.m += bw;
            b.s += bw;
        }
        if (!WIDE && tid == 0) {
            bqh->n_copy = min(nM + nS, bcap_copy);
            bqh->n_copy_medium = nM;
            bqh->n_copy_small = nS;
            bqh->overflow_copy = nM + nS > bcap_copy ? 1u : 0u;
            if (mirror) {
                mirror[2] = min(nM + nS, bcap_copy) + 1u;
                if (nM + nS > bcap_copy)
                    mirror[3] = 1u;
            }
        }
        q_phase(ph, 4);
        if (j.Ug > 0)
            b = qjoin_emit_copies<0, 4>(j, b);
        if (j.Ug > 4 * NT)
            b = qjoin_emit_copies<4, 8>(j, b);
    }
    q_phase(ph, 5);
    // the queue is complete once every builder workgroup has passed this point: items visible device-wide (release), then
    // the epoch words -- by this part for its own region (narrow path: two workgroups per step), or by whichever of the
    // step's `wide_parts` bucket workgroups finishes last
    // (inline asm: the item stores are asm too, invisible to the compiler's scoreboard -- it deletes a builtin wait it
    // believes has nothing to wait for, and the tag then overtakes the items)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this thread's items (write-through stores) are in memory
    __syncthreads();
    if (tid == 0) {
        if (!WIDE) {
            // the counts this part wrote above, through the L2 as well (device-scope atomic stores), then the tag
            if (part == 0) {
                __hip_atomic_store(&bqh->n_wave, bqh->n_wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&bqh->n_coop, bqh->n_coop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(&bqh->n_copy, bqh->n_copy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __hip_atomic_store(part == 0 ? &bqh->epoch_wave : &bqh->epoch_copy, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (signal &&
                   __hip_atomic_fetch_add(&bqh->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == wide_parts) {
            __hip_atomic_store(&bqh->epoch_wave, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&bqh->epoch_copy, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mirror) {       // every bucket has added its items: the counts are final (+ 1: 0 = not built yet)
                const uint32_t nw = __hip_atomic_load(&bqh->n_wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t nc = __hip_atomic_load(&bqh->n_coop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t np = __hip_atomic_load(&bqh->n_copy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mirror[1] = min(nc, bcap_coop) + 1u;
                mirror[2] = min(np, bcap_copy) + 1u;
                __threadfence_system();
                mirror[0] = min(nw, bcap_wave) + 1u;     // the word the host waits for, last
            }
        }
    }
}

// =====================================================================================================
// Workers
// =====================================================================================================
typedef float float2v_ __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4v sgd4(float4v acc, float4v g, float lr) {
    float4v r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __fsub_rn(acc[k], __fmul_rn(lr, g[k]));
    return r;
}
__device__ __forceinline__ float4v acc4(float4v p, float4v g, float lr) {   // p + lr * g, two roundings
    float4v r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __fadd_rn(p[k], __fmul_rn(lr, g[k]));
    return r;
}
__device__ __forceinline__ float4v add4(float4v x, float4v y) {
    float4v r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __fadd_rn(x[k], y[k]);
    return r;
}
__device__ __forceinline__ float4v sub4(float4v x, float4v y) {
    float4v r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __fsub_rn(x[k], y[k]);
    return r;
}
__device__ __forceinline__ float4v shfl_xor4(float4v v, int mask) {
    float4v r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __shfl_xor(v[k], mask, 64);
    return r;
}

// Cache policy of the three heavy streams (A/B knobs of tools/ab_variants.sh; the defaults are what measured best)
#ifndef QV_ROW_NT
#define QV_ROW_NT 1
#endif
#ifndef QV_OUT_NT
#define QV_OUT_NT 1
#endif
#ifndef QV_GRAD_NT
#define QV_GRAD_NT 0
#endif
#ifndef QV_INTERLEAVE
#define QV_INTERLEAVE 1
#endif
#ifndef QV_ROWLD_NT
#define QV_ROWLD_NT 0
#endif
template <typename V>
__device__ __forceinline__ V q_ld_row(const float *p) {     // the table row an item updates / copies: read once per launch
#if QV_ROWLD_NT
    return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
#else
    return *reinterpret_cast<const V *>(p);
#endif
}
template <typename V>
__device__ __forceinline__ void q_st_row(float *p, V v) {
#if QV_ROW_NT
    __builtin_nontemporal_store(v, reinterpret_cast<V *>(p));
#else
    *reinterpret_cast<V *>(p) = v;
#endif
}
template <typename V>
__device__ __forceinline__ void q_st_out(float *p, V v) {
#if QV_OUT_NT
    __builtin_nontemporal_store(v, reinterpret_cast<V *>(p));
#else
    *reinterpret_cast<V *>(p) = v;
#endif
}
template <typename V>
__device__ __forceinline__ V q_ld_grad(const float *p) {
#if QV_GRAD_NT
    return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
#else
    return *reinterpret_cast<const V *>(p);
#endif
}

struct QItem {
    int kind, col0, cols;
    uint32_t key, c, st, m, fs, o01, o23;
};
__device__ __forceinline__ QItem q_load(const QEntry *e, int width) {
    // the address is wave-uniform: scalar loads
    const uint4 lo = reinterpret_cast<const uint4 *>(e)[0], hi = reinterpret_cast<const uint4 *>(e)[1];
    QItem it;
    it.kind = static_cast<int>(uniform(lo.x & 15u));
    it.col0 = static_cast<int>(uniform(lo.x >> 4)) * 4;
    it.key = uniform(lo.y);
    it.c = uniform(lo.z);
    it.cols = min(q_slice(it.kind), width - it.col0);
    it.st = uniform(lo.w);
    it.m = uniform(hi.x);
    it.fs = uniform(hi.y);
    it.o01 = uniform(hi.z);
    it.o23 = uniform(hi.w);
    return it;
}

// S / Z: <= 512 columns by one wave, c <= 3 occurrences, two 16-byte vectors per lane
__device__ __forceinline__ void q_small(const QArgs &a, const QItem &it) {
    const int lane = lane_id();
    const int width = a.width;
    const int ca = it.col0 + 4 * lane, cb = ca + 256;
    const bool a0 = 4 * lane < it.cols, a1 = 256 + 4 * lane < it.cols;
    const int la = a0 ? ca : it.col0, lb = a1 ? cb : it.col0;   // loads stay branch-free: clamped columns
    float *row = a.table + static_cast<uint64_t>(it.kind == kQZ ? 0u : it.key) * static_cast<uint64_t>(width);
    float4v r0{0.f, 0.f, 0.f, 0.f}, r1{0.f, 0.f, 0.f, 0.f};
    if (it.kind != kQZ) {   // every branch on the item is wave-uniform
        r0 = q_ld_row<float4v>(row + la);
        r1 = q_ld_row<float4v>(row + lb);
    }
    int dv = 0;
    if (it.m > 0)
        dv = a.perm_g[it.fs + min(static_cast<uint32_t>(lane), it.m - 1u)];
    float4v g0[kQSmallC], g1[kQSmallC];
#pragma unroll
    for (int t = 0; t < kQSmallC; ++t) {
        if (static_cast<uint32_t>(t) < it.c) {
            const uint32_t o = t == 0 ? (it.o01 & kQOccMask)
                                      : t == 1 ? (((it.o01 >> 21) | (it.o23 << 11)) & kQOccMask) : ((it.o23 >> 10) & kQOccMask);
            const float *src = a.grads + static_cast<uint64_t>(o) * static_cast<uint64_t>(width);
            g0[t] = q_ld_grad<float4v>(src + la);
            g1[t] = q_ld_grad<float4v>(src + lb);
        }
    }
#pragma unroll
    for (int t = 0; t < kQSmallC; ++t) {
        if (static_cast<uint32_t>(t) < it.c) {
            r0 = sgd4(r0, g0[t], a.lr);
            r1 = sgd4(r1, g1[t], a.lr);
        }
    }
    if (it.c > 0) {
        if (a0)
            q_st_row(row + ca, r0);
        if (a1)
            q_st_row(row + cb, r1);
    }
    for (uint32_t j0 = 0; j0 < it.m; j0 += 64) {
        if (j0 > 0)
            dv = a.perm_g[it.fs + min(j0 + static_cast<uint32_t>(lane), it.m - 1u)];
        const int cnt = static_cast<int>(min(64u, it.m - j0));
        for (int j = 0; j < cnt; ++j) {
            float *o = a.out + static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(dv, j))) *
                                   static_cast<uint64_t>(width);
            if (a0)
                q_st_out(o + ca, r0);
            if (a1)
                q_st_out(o + cb, r1);
        }
    }
}

// Two S items by ONE wave, rows of at most 128 columns: lanes 0-31 take item A, lanes 32-63 item B (a 512-byte row is
// 32 lanes of 16 bytes: a wave per item leaves half its lanes idle, and a batch of 106,496 ids of 128-wide rows is ~60,000
// such items against 8,192 resident waves).  Everything is per lane: the item's words are read through the vector path
// (one address per half-wave), loads are branch-free (clamped occurrences), stores are predicated.  Same arithmetic as
// q_small: the serial chain of at most three occurrences.
__device__ __forceinline__ void q_small_pair(const QArgs &a, const QEntry *ea, const QEntry *eb) {
    const int lane = lane_id(), hl = lane & 31;
    const int width = a.width;
    const QEntry *e = lane < 32 ? ea : eb;
    const uint4 lo = reinterpret_cast<const uint4 *>(e)[0], hi = reinterpret_cast<const uint4 *>(e)[1];
    const bool zero = false;       // (S items only)
    const uint32_t key = lo.y, c = lo.z, m = hi.x, fs = hi.y;
    const int col = 4 * hl;
    const bool act = col < width;
    const int lc = act ? col : 0;
    float *row = a.table + static_cast<uint64_t>(zero ? 0u : key) * static_cast<uint64_t>(width);
    float4v r = q_ld_row<float4v>(row + lc);
    if (zero)
        r = float4v{0.f, 0.f, 0.f, 0.f};
    int dv = 0;
    if (m > 0)
        dv = a.perm_g[fs + min(static_cast<uint32_t>(hl), m - 1u)];
    const uint32_t o0 = hi.z & kQOccMask, o1 = ((hi.z >> 21) | (hi.w << 11)) & kQOccMask, o2 = (hi.w >> 10) & kQOccMask;
    // c == 0 (a copy): nothing is applied; the three loads then re-read the row itself instead of a gradient row
    const float *gb = c > 0 ? a.grads : a.table;
    const uint64_t w64 = static_cast<uint64_t>(width);
    const uint64_t i0 = c > 0 ? o0 : (zero ? 0u : key), i1 = c > 1 ? o1 : i0, i2 = c > 2 ? o2 : i0;
    const float4v g0 = q_ld_grad<float4v>(gb + i0 * w64 + lc);
    const float4v g1 = q_ld_grad<float4v>(gb + i1 * w64 + lc);
    const float4v g2 = q_ld_grad<float4v>(gb + i2 * w64 + lc);
    const float4v x0 = sgd4(r, g0, a.lr);
    r = c > 0 ? x0 : r;
    const float4v x1 = sgd4(r, g1, a.lr);
    r = c > 1 ? x1 : r;
    const float4v x2 = sgd4(r, g2, a.lr);
    r = c > 2 ? x2 : r;
    if (c > 0 && act)
        q_st_row(row + col, r);
#pragma unroll 4
    for (int j = 0; j < kQSmallM; ++j) {
        const uint32_t d = static_cast<uint32_t>(__shfl(dv, (lane & 32) + j, 64));
        if (static_cast<uint32_t>(j) < m && act)
            q_st_out(a.out + static_cast<uint64_t>(d) * w64 + col, r);
    }
}

// M: one 128-column slice, c <= 15 occurrences, 8 bytes per lane, ordered chain
__device__ __forceinline__ void q_medium(const QArgs &a, const QItem &it) {
    const int lane = lane_id();
    const int width = a.width;
    const int col = it.col0 + 2 * lane;
    const bool act = 2 * lane < it.cols;
    const int lc = act ? col : it.col0;
    float *row = a.table + static_cast<uint64_t>(it.key) * static_cast<uint64_t>(width);
    int pidx = 0, dv = 0;
    if (it.c > 0)
        pidx = a.perm_a[it.st + min(static_cast<uint32_t>(lane), it.c - 1u)];
    if (it.m > 0)
        dv = a.perm_g[it.fs + min(static_cast<uint32_t>(lane), it.m - 1u)];
    float2v_ r = q_ld_row<float2v_>(row + lc);
    // branch-free: lanes >= c hold the index of the last occurrence, so the loads beyond c repeat a line the wave
    // has just asked for and the chain skips them by select (uniform branches around 15 loads make the compiler
    // spill; clamped loads are what scatter_dev.h does as well)
    float2v_ g[kQMediumC];
    const float *gbase = it.c > 0 ? a.grads : a.table;   // c == 0: a pure copy, nothing is applied
#pragma unroll
    for (int t = 0; t < kQMediumC; ++t) {
        const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_readlane(pidx, t));
        g[t] = q_ld_grad<float2v_>(gbase + static_cast<uint64_t>(o) * static_cast<uint64_t>(width) + lc);
    }
#pragma unroll
    for (int t = 0; t < kQMediumC; ++t) {
        const float x0 = __fsub_rn(r[0], __fmul_rn(a.lr, g[t][0]));
        const float x1 = __fsub_rn(r[1], __fmul_rn(a.lr, g[t][1]));
        const bool on = static_cast<uint32_t>(t) < it.c;
        r[0] = on ? x0 : r[0];
        r[1] = on ? x1 : r[1];
    }
    if (it.c > 0 && act)
        q_st_row(row + col, r);
    for (uint32_t j0 = 0; j0 < it.m; j0 += 64) {
        if (j0 > 0)
            dv = a.perm_g[it.fs + min(j0 + static_cast<uint32_t>(lane), it.m - 1u)];
        const int cnt = static_cast<int>(min(64u, it.m - j0));
        for (int j = 0; j < cnt; ++j) {
            float *o = a.out + static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(dv, j))) *
                                   static_cast<uint64_t>(width);
            if (act)
                q_st_out(o + col, r);
        }
    }
}

// L: one 32-column slice, 16 <= c < 64: lane = (occurrence group r of 8, column quad c4); fixed-order tree
__device__ __forceinline__ void q_long(const QArgs &a, const QItem &it) {
    const int lane = lane_id();
    const int width = a.width;
    const int r = lane >> 3, c4 = lane & 7;
    const bool act = 4 * c4 < it.cols;
    const int col = it.col0 + (act ? 4 * c4 : 0);
    float *row = a.table + static_cast<uint64_t>(it.key) * static_cast<uint64_t>(width);
    const int pidx = a.perm_a[it.st + min(static_cast<uint32_t>(lane), it.c - 1u)];
    int dv = 0;
    if (it.m > 0)
        dv = a.perm_g[it.fs + min(static_cast<uint32_t>(lane), it.m - 1u)];
    const float4v cur = q_ld_row<float4v>(row + col);
    float4v g[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint32_t occ = static_cast<uint32_t>(8 * t + r);
        const uint32_t o = static_cast<uint32_t>(__shfl(pidx, static_cast<int>(min(occ, it.c - 1u)), 64));
        g[t] = q_ld_grad<float4v>(a.grads + static_cast<uint64_t>(o) * static_cast<uint64_t>(width) + col);   // branch-free (clamped)
    }
    float4v p{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float4v q = acc4(p, g[t], a.lr);
        const bool valid = static_cast<uint32_t>(8 * t + r) < it.c;
        p = valid ? q : p;
    }
    p = add4(p, shfl_xor4(p, 8));
    p = add4(p, shfl_xor4(p, 16));
    p = add4(p, shfl_xor4(p, 32));
    const float4v nv = sub4(cur, p);
    if (r == 0 && act)
        q_st_row(row + col, nv);
    for (uint32_t j0 = 0; j0 < it.m; j0 += 64) {
        if (j0 > 0)
            dv = a.perm_g[it.fs + min(j0 + static_cast<uint32_t>(lane), it.m - 1u)];
        const uint32_t cnt = min(64u, it.m - j0);
        for (uint32_t j = 0; j < cnt; j += 8) {
            const uint32_t d = static_cast<uint32_t>(__shfl(dv, static_cast<int>(min(j + r, cnt - 1u)), 64));
            if (j + r < cnt && act)
                q_st_out(a.out + static_cast<uint64_t>(d) * static_cast<uint64_t>(width) + col, nv);
        }
    }
}

#if QV_GOLD   // the product's G item: sixteen waves per 64-column slice
// G: one 64-column slice by a whole workgroup, c >= 64.  lane = (row r of 4, column quad c4 of 16); wave w takes
// occurrences 16w .. 16w+15 of every block of 256 (four 16-byte loads per lane and block).  s_part = 16 x 64 floats.
__device__ __forceinline__ void q_coop_r3(const QArgs &a, const QItem &it, float *s_part, uint32_t e) {
    const int lane = lane_id(), w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int width = a.width;
    const int r = lane >> 4, c4 = lane & 15;
    const bool act = 4 * c4 < it.cols;
    const int col = it.col0 + (act ? 4 * c4 : 0);
    float *row = a.table + static_cast<uint64_t>(it.key) * static_cast<uint64_t>(width);
    // a key with more than kQChunk occurrences arrives as `nch` items per slice: this one sums chunk `ch`
    const uint32_t ch = it.o01 & 0xFFFFu, nch = it.o01 >> 16;
    const uint32_t c_lo = kQChunk * ch, c_hi = nch > 1u ? min(it.c, c_lo + kQChunk) : it.c;
    float4v cur = q_ld_row<float4v>(row + col);
    // destinations of this wave: j = 64 * k + 4 * w + r in round k.  Lane l fetches the one of (k, r) = (l >> 2, l & 3):
    // one register covers the first 16 rounds (1,024 destinations)
    int dv = 0;
    if (it.m > 0)
        dv = a.perm_g[it.fs + min(static_cast<uint32_t>(64 * (lane >> 2) + 4 * w + (lane & 3)), it.m - 1u)];
    float4v p{0.f, 0.f, 0.f, 0.f};
    for (uint32_t base = c_lo; base < c_hi; base += 256) {
        const uint32_t mine = base + 16u * static_cast<uint32_t>(w);   // this wave's first occurrence of the block
        if (mine >= c_hi)
            break;   // wave-uniform; no barrier inside the loop
        const int pidx = a.perm_a[it.st + min(mine + static_cast<uint32_t>(lane & 15), c_hi - 1u)];
        float4v g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t o = static_cast<uint32_t>(__shfl(pidx, 4 * t + r, 64));
            g[t] = q_ld_grad<float4v>(a.grads + static_cast<uint64_t>(o) * static_cast<uint64_t>(width) + col);   // branch-free (clamped)
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4v q = acc4(p, g[t], a.lr);
            const bool valid = mine + static_cast<uint32_t>(4 * t + r) < c_hi;
            p = valid ? q : p;
        }
    }
    p = add4(p, shfl_xor4(p, 16));
    p = add4(p, shfl_xor4(p, 32));
    if (lane < 16)
        *reinterpret_cast<float4v *>(s_part + w * 64 + 4 * c4) = p;
    __syncthreads();
    // fixed tree over the 16 wave partials: lane group r adds the partials of waves 4r .. 4r+3 as (a + b) + (c + d),
    // the four group sums meet by two butterfly steps -- ((q0 + q1) + (q2 + q3)) in every lane (IEEE addition
    // commutes, so both partners of a step hold the same bits)
    const float *sp = s_part + (4 * r) * 64 + 4 * c4;
    float4v total = add4(add4(*reinterpret_cast<const float4v *>(sp), *reinterpret_cast<const float4v *>(sp + 64)),
                         add4(*reinterpret_cast<const float4v *>(sp + 128), *reinterpret_cast<const float4v *>(sp + 192)));
    total = add4(total, shfl_xor4(total, 16));
    total = add4(total, shfl_xor4(total, 32));
    if (nch > 1u) {
        // deliver this chunk's sum (device-coherent stores, drained, then the counter of the slice -- the hand-off rules of
        // step.hip); whoever delivers last adds all of them up in chunk order.  Nobody waits for anybody.
        const uint32_t first = e - ch;
        if (w == 0 && r == 0)
            st4_sc1(a.qpart + static_cast<size_t>(e) * 64 + 4 * c4, total);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (asm: see qjoin_body)
        __syncthreads();
        uint32_t *s_flag = reinterpret_cast<uint32_t *>(s_part);     // (the partial sums in LDS have been consumed)
        if (threadIdx.x == 0)      // (relaxed: the sum went through the L2 and was drained; a release would write the whole L2 back)
            *s_flag = __hip_atomic_fetch_add(a.qpcnt + first, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const bool last = *s_flag + 1u == nch;
        __syncthreads();
        if (!last)
            return;
        // the chunk sums, four loads in flight at a time, added in chunk order
        const float *pp = a.qpart + static_cast<size_t>(first) * 64 + 4 * c4;
        float4v tot{0.f, 0.f, 0.f, 0.f};
        for (uint32_t j0 = 0; j0 < nch; j0 += 4) {
            float4v q0 = ld4_sc1_async(pp + static_cast<size_t>(min(j0, nch - 1u)) * 64);
            float4v q1 = ld4_sc1_async(pp + static_cast<size_t>(min(j0 + 1u, nch - 1u)) * 64);
            float4v q2 = ld4_sc1_async(pp + static_cast<size_t>(min(j0 + 2u, nch - 1u)) * 64);
            float4v q3 = ld4_sc1_async(pp + static_cast<size_t>(min(j0 + 3u, nch - 1u)) * 64);
            wait_loads(q0, q1, q2, q3);
            tot = j0 == 0 ? q0 : add4(tot, q0);
            if (j0 + 1u < nch) tot = add4(tot, q1);
            if (j0 + 2u < nch) tot = add4(tot, q2);
            if (j0 + 3u < nch) tot = add4(tot, q3);
        }
        total = tot;
    }
    const float4v nv = sub4(cur, total);
    if (w == 0 && r == 0 && act)
        q_st_row(row + col, nv);
    for (uint32_t k0 = 0; k0 * 64u < it.m; k0 += 16) {
        if (k0 > 0)
            dv = a.perm_g[it.fs + min(64u * (k0 + static_cast<uint32_t>(lane >> 2)) + static_cast<uint32_t>(4 * w + (lane & 3)),
                                      it.m - 1u)];
        const uint32_t rounds = min(16u, (it.m - 64u * k0 + 63u) / 64u);
        for (uint32_t k = 0; k < rounds; ++k) {
            const uint32_t j = 64u * (k0 + k) + static_cast<uint32_t>(4 * w + r);
            const uint32_t d = static_cast<uint32_t>(__shfl(dv, static_cast<int>(4 * k) + r, 64));
            if (j < it.m && act)
                q_st_out(a.out + static_cast<uint64_t>(d) * static_cast<uint64_t>(width) + col, nv);
        }
    }
    __syncthreads();   // s_part is reused by the next item of this workgroup
}

#endif

// G: one 32-column slice by a whole workgroup of FOUR waves, c >= 64.  Every wave works like an L item on its share of
// the run -- lane = (occurrence group r of 8, column quad c4 of 8), wave w takes occurrences 64w .. 64w+63 of every
// block of 256 (eight 16-byte loads per lane and block) --, the four wave sums meet in LDS as (w0 + w1) + (w2 + w3).
// s_part = 4 x 32 floats.
__device__ __forceinline__ void q_coop(const QArgs &a, const QItem &it, float *s_part) {
    const int lane = lane_id(), w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int width = a.width;
    const int r = lane >> 3, c4 = lane & 7;
    const bool act = 4 * c4 < it.cols;
    const int col = it.col0 + (act ? 4 * c4 : 0);
    float *row = a.table + static_cast<uint64_t>(it.key) * static_cast<uint64_t>(width);
    const float4v cur = q_ld_row<float4v>(row + col);
    // destinations of this wave: j = 8 W * k + 8 * w + r in round k (W waves).  Lane l fetches the one of (k, r) =
    // (l >> 3, l & 7): one register covers eight rounds
    constexpr uint32_t kDW = 8u * kQWpw, kBlk = 64u * kQWpw;
    int dv = 0;
    if (it.m > 0)
        dv = a.perm_g[it.fs + min(kDW * static_cast<uint32_t>(lane >> 3) + static_cast<uint32_t>(8 * w + (lane & 7)), it.m - 1u)];
    float4v p{0.f, 0.f, 0.f, 0.f};
    for (uint32_t base = 0; base < it.c; base += kBlk) {
        const uint32_t mine = base + 64u * static_cast<uint32_t>(w);   // this wave's first occurrence of the block
        if (mine >= it.c)
            break;   // wave-uniform; no barrier inside the loop
        const int pidx = a.perm_a[it.st + min(mine + static_cast<uint32_t>(lane), it.c - 1u)];
        float4v g[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t o = static_cast<uint32_t>(__shfl(pidx, 8 * t + r, 64));
            g[t] = q_ld_grad<float4v>(a.grads + static_cast<uint64_t>(o) * static_cast<uint64_t>(width) + col);   // branch-free (clamped)
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4v q = acc4(p, g[t], a.lr);
            const bool valid = mine + static_cast<uint32_t>(8 * t + r) < it.c;
            p = valid ? q : p;
        }
    }
    p = add4(p, shfl_xor4(p, 8));
    p = add4(p, shfl_xor4(p, 16));
    p = add4(p, shfl_xor4(p, 32));
    if (lane < 8)
        *reinterpret_cast<float4v *>(s_part + w * 32 + 4 * c4) = p;
    __syncthreads();
    const float *sp = s_part + 4 * c4;
    float4v total = add4(add4(*reinterpret_cast<const float4v *>(sp), *reinterpret_cast<const float4v *>(sp + 32)),
                         add4(*reinterpret_cast<const float4v *>(sp + 64), *reinterpret_cast<const float4v *>(sp + 96)));
#pragma unroll
    for (int q = 1; q < kQWpw / 4; ++q)        // (timing variants with larger workgroups; the product has four waves)
        total = add4(total, add4(add4(*reinterpret_cast<const float4v *>(sp + 128 * q), *reinterpret_cast<const float4v *>(sp + 128 * q + 32)),
                                 add4(*reinterpret_cast<const float4v *>(sp + 128 * q + 64), *reinterpret_cast<const float4v *>(sp + 128 * q + 96))));
    const float4v nv = sub4(cur, total);
    if (w == 0 && r == 0 && act)
        q_st_row(row + col, nv);
    for (uint32_t k0 = 0; k0 * kDW < it.m; k0 += 8) {
        if (k0 > 0)
            dv = a.perm_g[it.fs + min(kDW * (k0 + static_cast<uint32_t>(lane >> 3)) + static_cast<uint32_t>(8 * w + (lane & 7)),
                                      it.m - 1u)];
        const uint32_t rounds = min(8u, (it.m - kDW * k0 + kDW - 1u) / kDW);
        for (uint32_t k = 0; k < rounds; ++k) {
            const uint32_t j = kDW * (k0 + k) + static_cast<uint32_t>(8 * w + r);
            const uint32_t d = static_cast<uint32_t>(__shfl(dv, static_cast<int>(8 * k) + r, 64));
            if (j < it.m && act)
                q_st_out(a.out + static_cast<uint64_t>(d) * static_cast<uint64_t>(width) + col, nv);
        }
    }
    __syncthreads();   // s_part is reused by the next item of this workgroup
}

// ---- the three launches ----------------------------------------------------------------------------------------
// The items of one step: workgroups [0, ncoop) take the G items, the others one wave item per wave.
static_assert(kQWg == 256 || kQWg == 512 || kQWg == 1024, "workgroups of 4, 8 or 16 waves");
static_assert(!QV_GOLD || kQWg == 1024, "the sixteen-wave G item needs 1024-thread workgroups");
__global__ __launch_bounds__(kQWg, 8) void qapply_kernel(const QArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const unsigned long long t0 = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int role, kind = -1;
    int b = blockIdx.x;
    if (a.epoch != 0u) {
        // The caller orders this launch behind the queue's builder without a wait on this stream (the builder ran a block of
        // steps ago): the epoch words say so.  They are there on the first look; if not, poll (device-coherent loads, bounded:
        // 2 s) -- the builder never waits for this launch -- and give up loudly rather than read a half-built queue.
        // One decision per launch: see QHeader::verdict.
        const uint32_t fail = ~a.epoch;
        bool ready = a.qh->epoch_wave == a.epoch && a.qh->epoch_copy == a.epoch &&      // (read with the header's counts)
                     a.qh->verdict != fail;
        if (blockIdx.x == 0) {
            // (bounded by the 100 MHz clock: 2 s.  A count of spins alone -- 2^20, round 4 -- came to ~40 s on a loaded chip)
            const unsigned long long t_poll = __builtin_amdgcn_s_memrealtime();
            for (int spin = 0; !ready && spin < (1 << 22); ++spin) {
                const uint32_t e0 = __hip_atomic_load(&a.qh->epoch_wave, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t e1 = __hip_atomic_load(&a.qh->epoch_copy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                ready = uniform(static_cast<uint32_t>(e0 == a.epoch && e1 == a.epoch)) != 0u;
                if (ready || __builtin_amdgcn_s_memrealtime() - t_poll > 200000000ull)
                    break;
                __builtin_amdgcn_s_sleep(64);
            }
            if (threadIdx.x == 0) {
                __hip_atomic_store(const_cast<uint32_t *>(&a.qh->verdict), ready ? a.epoch : fail, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);      // (the one word an apply launch writes into its queue)
                if (!ready && a.err != nullptr)
                    __hip_atomic_store(a.err, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else if (!ready) {
            // not there at the first look: workgroup 0 decides (a little beyond its 2 s: it may have started later)
            const unsigned long long t_poll = __builtin_amdgcn_s_memrealtime();
            uint32_t vd = 0u;
            for (int spin = 0; spin < (1 << 23); ++spin) {
                vd = uniform(__hip_atomic_load(&a.qh->verdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
                if (vd == a.epoch || vd == fail || __builtin_amdgcn_s_memrealtime() - t_poll > 400000000ull)
                    break;
                __builtin_amdgcn_s_sleep(64);
            }
            ready = vd == a.epoch;
            if (!ready && vd != fail && a.err != nullptr && threadIdx.x == 0)      // (workgroup 0 never spoke: say so as well)
                __hip_atomic_store(a.err, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (!ready)
            return;
    }
    const bool coop = b < a.ncoop;
    if (coop) {
        role = 0;
        const uint32_t n = min(a.qh->n_coop, a.cap_coop);     // (a count beyond its region: the builder raised its overflow word)
        for (uint32_t e = static_cast<uint32_t>(b); e < n; e += static_cast<uint32_t>(a.ncoop)) {
            const QItem it = q_load(a.qcoop + e, a.width);
#if QV_GOLD
            q_coop_r3(a, it, reinterpret_cast<float *>(s_dyn), e);
#else
            q_coop(a, it, reinterpret_cast<float *>(s_dyn));
#endif
            kind = it.kind;
        }
    } else {
        b -= a.ncoop;
        role = 3;
        const uint32_t n0 = min(a.qh->n_wave, a.cap_wave), n = n0 + min(a.qh->n_copy, a.cap_copy);
        const uint32_t stride = static_cast<uint32_t>(a.nworker) * static_cast<uint32_t>(kQWpw);
        const uint32_t wv = uniform(static_cast<uint32_t>(threadIdx.x >> 6));
        if (a.width <= 128) {
            // narrow rows: every wave takes a PAIR of consecutive items; two S / Z items share the wave (q_small_pair),
            // anything else is done one after the other
            const uint32_t n1 = n - n0, both = 2u * (n0 < n1 ? n0 : n1), npair = (n + 1u) / 2u;
            for (uint32_t pe = static_cast<uint32_t>(b) * static_cast<uint32_t>(kQWpw) + wv; pe < npair; pe += stride) {
                const QEntry *src[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t e = 2u * pe + static_cast<uint32_t>(h);
                    src[h] = e >= n ? nullptr
                                    : e < both ? ((e & 1u) ? a.qcopy + (e >> 1) : a.qwave + (e >> 1))
                                               : (n0 > n1 ? a.qwave + (e - n1) : a.qcopy + (e - n0));
                }
                const uint32_t ka = uniform(src[0]->w[0] & 15u), kb = src[1] ? uniform(src[1]->w[0] & 15u) : 15u;
                // (Z items -- ids beyond the table -- may have any number of destinations: they keep a wave of their own)
                const bool sa = ka == static_cast<uint32_t>(kQS), sb = kb == static_cast<uint32_t>(kQS);
                if (sa && sb) {
                    q_small_pair(a, src[0], src[1]);
                    kind = kQS;
                    continue;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (src[h] == nullptr)
                        continue;
                    const QItem it = q_load(src[h], a.width);
                    kind = it.kind;
                    if (it.kind == kQL)
                        q_long(a, it);
                    else if (it.kind == kQM)
                        q_medium(a, it);
                    else
                        q_small(a, it);
                }
            }
        } else
        for (uint32_t e = static_cast<uint32_t>(b) * static_cast<uint32_t>(kQWpw) + wv; e < n; e += stride) {
#if QV_INTERLEAVE
            // Wave items and copy items ALTERNATE over the launch while both last (then the rest of the longer list):
            // a copy writes m rows for one it reads, an apply item reads more than it writes, and a compute unit that
            // holds only one kind is bound by its load or its store path while the other idles.  With all copies at the
            // end of the launch a step takes 13.5 us, alternating 12.4 (same box; copies first 13.5, two copies per wave
            // item 13.2, copies spread evenly over all wave items 13.2, small wave items first 13.1-13.2).
            const uint32_t n1 = n - n0, both = 2u * (n0 < n1 ? n0 : n1);
            const QEntry *src;
            if (e < both)
                src = (e & 1u) ? a.qcopy + (e >> 1) : a.qwave + (e >> 1);
            else
                src = n0 > n1 ? a.qwave + (e - n1) : a.qcopy + (e - n0);
            const QItem it = q_load(src, a.width);
#else
            const QItem it = q_load(e < n0 ? a.qwave + e : a.qcopy + (e - n0), a.width);
#endif
            kind = it.kind;
            if (it.kind == kQL)
                q_long(a, it);
            else if (it.kind == kQM)
                q_medium(a, it);
            else
                q_small(a, it);
        }
    }
    if (a.dbg) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane_id() == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long *d = a.dbg + (static_cast<size_t>(blockIdx.x) * kQWpw + (threadIdx.x >> 6)) * 4;
            d[0] = t0;
            d[1] = t1;
            d[2] = static_cast<unsigned long long>(role) | (static_cast<unsigned long long>(xcc & 0xF) << 8);
            // the wave's FIRST item again (its only one in a launch that fits the chip): kind | c << 8 | m << 28 | col0 / 4 << 48
            // -- read here, behind the stamps, so that the census costs the measured path nothing (tools/qstep_timeline.py)
            unsigned long long rec = static_cast<unsigned long long>(static_cast<unsigned>(kind)) & 0xFFull;
            if (kind >= 0) {
                const QEntry *src = nullptr;
                if (role == 0) {
                    src = a.qcoop + blockIdx.x;
                } else if (a.width > 128) {
                    const uint32_t n0 = min(a.qh->n_wave, a.cap_wave), n = n0 + min(a.qh->n_copy, a.cap_copy);
                    const uint32_t e = static_cast<uint32_t>(b) * static_cast<uint32_t>(kQWpw) + static_cast<uint32_t>(threadIdx.x >> 6);
                    const uint32_t n1 = n - n0, both = 2u * (n0 < n1 ? n0 : n1);
                    if (e < n)
                        src = e < both ? ((e & 1u) ? a.qcopy + (e >> 1) : a.qwave + (e >> 1))
                                       : (n0 > n1 ? a.qwave + (e - n1) : a.qcopy + (e - n0));
                }
                if (src != nullptr)
                    rec = (src->w[0] & 15u) | (static_cast<unsigned long long>(src->w[2] & 0xFFFFFu) << 8) |
                          (static_cast<unsigned long long>(src->w[4] & 0xFFFFFu) << 28) |
                          (static_cast<unsigned long long>((src->w[0] >> 4) & 0xFFFFu) << 48);
            }
            d[3] = rec;
        }
    }
}

// The plans of up to kQBatch batches, one workgroup each (a plan keeps ONE workgroup busy for ~15 us whatever else
// runs: batches of a block of steps are planned side by side in one launch instead of one after the other).
constexpr int kQBatch = 16;
struct QPlanBatch {
    int count;
    const void *ids[kQBatch];
    QPlan plan[kQBatch];
    unsigned long long *ph;     // development aid: phase stamps of workgroup 0
};
template <typename IdT, bool RANK_ATOMIC>
__global__ __launch_bounds__(1024, 4) void qplan_kernel(const QPlanBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int i = blockIdx.x;
    qsort_finish_body<IdT, RANK_ATOMIC>(static_cast<const IdT *>(b.ids[i]), b.plan[i], s_dyn, i == 0 ? b.ph : nullptr);
}

// The queues of up to kQJoinBatch steps, two workgroups each.
constexpr int kQJoinBatch = 8;
struct QJoinBatch {
    int count, width;
    uint64_t rows;
    uint32_t lds_bytes, cap_coop, cap_wave, cap_copy;
    QPlan pa[kQJoinBatch], pg[kQJoinBatch];
    QHeader *qh[kQJoinBatch];
    QEntry *coop[kQJoinBatch], *wave[kQJoinBatch], *copy[kQJoinBatch];
    uint32_t *mirror[kQJoinBatch];   // optional pinned host words per step: {wave items, workgroup items, copy items} + 1
    uint32_t epoch[kQJoinBatch];     // the tag the finished queue carries (0: none asked for)
    unsigned long long *ph;     // development aid: phase stamps of workgroups 0 and 1
};
__global__ __launch_bounds__(1024, 4) void qqueue_kernel(const QJoinBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int i = blockIdx.x >> 1, part = blockIdx.x & 1;
    qjoin_body(b.pa[i], b.pg[i], b.rows, b.width, b.qh[i], b.coop[i], b.wave[i], b.copy[i], b.cap_coop, b.cap_wave,
               b.cap_copy, s_dyn, b.lds_bytes, part, (i == 0 && b.ph) ? b.ph + 8 * part : nullptr, b.mirror[i], 0u, 0u,
               b.epoch[i]);
}

// One wave-instruction's LDS atomics on one address: are the lanes served in ascending lane order?  Four collision
// patterns (all lanes one counter; two; sixteen; 512 counters), seven rows, 16 waves.
__global__ __launch_bounds__(1024) void q_lds_order_check_kernel(uint32_t *bad) {
    __shared__ uint32_t cnt[16][256];
    __shared__ uint32_t ref[16][512];
    const int w = threadIdx.x >> 6, lane = lane_id();
    for (int i = threadIdx.x; i < 16 * 256; i += 1024)
        (&cnt[0][0])[i] = 0;
    for (int i = threadIdx.x; i < 16 * 512; i += 1024)
        (&ref[0][0])[i] = 0;
    __syncthreads();
    uint32_t s = 12345u + 977u * threadIdx.x, wrong = 0;
    for (int r = 0; r < 7; ++r) {
        s = s * 1664525u + 1013904223u;
        const uint32_t x = s >> 16;
        const int mode = w & 3;
        const uint32_t d = mode == 0 ? 7u : mode == 1 ? (x & 1u) * 255u + 3u : mode == 2 ? (x & 15u) * 3u : (x & 511u);
        const uint32_t sh = (d & 1u) * 16u;
        const uint32_t old = (__hip_atomic_fetch_add(&cnt[w][d >> 1], 1u << sh, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 0xFFFFu;
        // reference rank: lanes of this wave one after the other
        uint32_t want = 0;
        for (int l = 0; l < 64; ++l) {
            if (lane == l) {
                want = ref[w][d];
                ref[w][d] = want + 1;
            }
        }
        wrong += old != want;
    }
    if (wrong)
        atomicAdd(bad, wrong);
}

// 1 = lane-ordered (use the atomic ranking), 0 = not / could not be checked (ballot ranking).  Checked once per device,
// synchronously: the first ha_qstep_* call on a device must not be made inside a stream capture (as for the LDS
// attribute).  HA_QSTEP_BALLOT=1 forces the ballot form.
static int lds_atomics_lane_ordered() {
    static std::atomic<int> state[256];   // 0 unknown, 1 yes, 2 no
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256)
        return 0;
    int st = state[d].load(std::memory_order_acquire);
    if (st == 0) {
        st = 2;
        const char *env = getenv("HA_QSTEP_BALLOT");
        uint32_t *bad = nullptr, host = 1;
        if (!(env && env[0] == '1') && hipMalloc(reinterpret_cast<void **>(&bad), 4) == hipSuccess) {
            if (hipMemset(bad, 0, 4) == hipSuccess) {
                hipLaunchKernelGGL(q_lds_order_check_kernel, dim3(8), dim3(1024), 0, nullptr, bad);
                if (hipGetLastError() == hipSuccess &&
                    hipMemcpy(&host, bad, 4, hipMemcpyDeviceToHost) == hipSuccess && host == 0)
                    st = 1;
            }
            (void)hipFree(bad);
        }
        state[d].store(st, std::memory_order_release);
    }
    return st == 1;
}

template <typename IdT>
static int qplan_batch(const IdT *const *ids, const int64_t *n, void *const *plans, int64_t count, hipStream_t stream,
                       unsigned long long *ph = nullptr) {
    HA_REQUIRE(count >= 0 && (count == 0 || (ids && n && plans)), "ha_qplan_batch: null pointer");
    static DeviceOnce lds_allowed;   // once per device, and outside any stream capture (the first call is eager)
    if (lds_allowed.run([]() -> int {
            HA_ALLOW_LDS((qplan_kernel<IdT, true>), 160 * 1024);
            HA_ALLOW_LDS((qplan_kernel<IdT, false>), 160 * 1024);
            return 0;
        }))
        return -1;
    const bool ordered = lds_atomics_lane_ordered() != 0;
    for (int64_t k0 = 0; k0 < count; k0 += kQBatch) {
        QPlanBatch b;
        memset(&b, 0, sizeof(b));
        b.ph = k0 == 0 ? ph : nullptr;
        size_t lds = 0;
        for (int64_t k = k0; k < count && b.count < kQBatch; ++k) {
            HA_REQUIRE(n[k] >= 0 && n[k] <= kQMax, "ha_qplan_batch: at most %d ids per batch", kQMax);
            if (n[k] == 0)
                continue;      // an empty batch has no plan
            HA_REQUIRE(ids[k] && plans[k], "ha_qplan_batch: null pointer (batch %lld)", (long long)k);
            b.ids[b.count] = ids[k];
            b.plan[b.count] = qplan(plans[k], n[k]);
            lds = lds > qsort_lds_bytes(static_cast<int>(n[k])) ? lds : qsort_lds_bytes(static_cast<int>(n[k]));
            ++b.count;
        }
        if (b.count == 0)
            continue;
        if (ordered)
            hipLaunchKernelGGL((qplan_kernel<IdT, true>), dim3(b.count), dim3(1024), lds, stream, b);
        else
            hipLaunchKernelGGL((qplan_kernel<IdT, false>), dim3(b.count), dim3(1024), lds, stream, b);
        HA_LAUNCH_CHECK();
    }
    return 0;
}

static int qqueue_batch(int64_t rows, int64_t width, void *const *plans_a, const int64_t *n_a, void *const *plans_g,
                        const int64_t *n_g, void *const *queues, int64_t queue_n_cap, int64_t count,
                        hipStream_t stream, unsigned long long *ph = nullptr, uint32_t *const *counts_host = nullptr,
                        const uint32_t *epochs = nullptr) {
    HA_REQUIRE(rows >= 0 && rows <= 0xFFFFFFFEll && width >= 4 && width % 4 == 0 && width <= (1 << 20),
               "ha_qqueue_batch: rows of a multiple of 4 floats");
    HA_REQUIRE(count >= 0 && (count == 0 || (plans_a && n_a && plans_g && n_g && queues)), "ha_qqueue_batch: null pointer");
    HA_REQUIRE(count == 0 || (queue_n_cap >= 1 && queue_n_cap <= kQMax), "ha_qqueue_batch: bad queue capacity");
    static DeviceOnce lds_allowed;
    if (lds_allowed.run([]() -> int {
            HA_ALLOW_LDS(qqueue_kernel, 160 * 1024);
            return 0;
        }))
        return -1;
    for (int64_t k0 = 0; k0 < count; k0 += kQJoinBatch) {
        QJoinBatch b;
        memset(&b, 0, sizeof(b));
        b.rows = static_cast<uint64_t>(rows);
        b.width = static_cast<int>(width);
        b.ph = k0 == 0 ? ph : nullptr;
        size_t lds = 0;
        for (int64_t k = k0; k < count && b.count < kQJoinBatch; ++k) {
            HA_REQUIRE(n_a[k] >= 0 && n_g[k] >= 0 && n_a[k] <= queue_n_cap && n_g[k] <= queue_n_cap,
                       "ha_qqueue_batch: a batch is larger than the queues were sized for");
            if (n_a[k] == 0 && n_g[k] == 0)
                continue;      // nothing to apply, nothing to look up: no queue
            HA_REQUIRE(queues[k] && (n_a[k] == 0 || plans_a[k]) && (n_g[k] == 0 || plans_g[k]),
                       "ha_qqueue_batch: null pointer (step %lld)", (long long)k);
            const int i = b.count++;
            b.pa[i] = qplan(plans_a[k], n_a[k]);
            b.pg[i] = qplan(plans_g[k], n_g[k]);
            const QLayout q = queue_layout(queues[k], queue_n_cap, width);
            b.qh[i] = q.hdr;
            b.mirror[i] = counts_host ? counts_host[k] : nullptr;
            b.epoch[i] = epochs ? epochs[k] : 0u;
            b.coop[i] = q.coop;
            b.wave[i] = q.wave;
            b.copy[i] = q.copy;
            b.cap_coop = q.cap_coop;
            b.cap_wave = q.cap_wave;
            b.cap_copy = q.cap_copy;
            const size_t need = qjoin_lds_bytes(b.pa[i].n, b.pg[i].n), res = qjoin_lds_resident_bytes(b.pg[i].n);
            lds = lds > need ? lds : need;
            lds = lds > res ? lds : res;      // the lookup batch's counts / starts in LDS too (a launch of its own: no
                                              // second workgroup has to fit on the CU)
        }
        if (b.count == 0)
            continue;
        b.lds_bytes = static_cast<uint32_t>(lds);
        hipLaunchKernelGGL(qqueue_kernel, dim3(2 * b.count), dim3(1024), lds, stream, b);
        HA_LAUNCH_CHECK();
    }
    return 0;
}

static int qapply_lists(float *table, int64_t rows, int64_t width, const int32_t *perm_cur, int64_t n_cur, const float *grads,
                        float lr, const int32_t *perm_next, int64_t n_next, float *next_out, const void *queue_cur,
                        int64_t queue_n_cap, int64_t n_max, hipStream_t stream, unsigned long long *dbg = nullptr,
                        int64_t wave_items = -1, uint32_t epoch = 0, uint32_t *err = nullptr, hipEvent_t done = nullptr,
                        int64_t coop_items = -1);

static int qapply(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads, float lr,
                  void *plan_next, int64_t n_next, float *next_out, const void *queue_cur, int64_t queue_n_cap,
                  hipStream_t stream, unsigned long long *dbg = nullptr, int64_t wave_items = -1, uint32_t epoch = 0,
                  uint32_t *err = nullptr, hipEvent_t done = nullptr) {
    HA_REQUIRE(n_cur >= 0 && n_next >= 0 && (n_cur == 0 || plan_cur) && (n_next == 0 || plan_next),
               "ha_qapply: a batch needs its plan");
    return qapply_lists(table, rows, width, n_cur > 0 ? plan_layout(plan_cur, n_cur).perm : nullptr, n_cur, grads, lr,
                        n_next > 0 ? plan_layout(plan_next, n_next).perm : nullptr, n_next, next_out, queue_cur, queue_n_cap,
                        kQMax, stream, dbg, wave_items, epoch, err, done);
}

// perm_cur / perm_next: the occurrence lists of the batch to apply / the destination lists of the batch to look up (what
// the items' `st` / `fs` index); n_max: the largest batch of the path that built the queue
static int qapply_lists(float *table, int64_t rows, int64_t width, const int32_t *perm_cur, int64_t n_cur, const float *grads,
                        float lr, const int32_t *perm_next, int64_t n_next, float *next_out, const void *queue_cur,
                        int64_t queue_n_cap, int64_t n_max, hipStream_t stream, unsigned long long *dbg,
                        int64_t wave_items, uint32_t epoch, uint32_t *err, hipEvent_t done, int64_t coop_items) {
    HA_REQUIRE(table != nullptr && rows >= 0 && rows <= 0xFFFFFFFEll && width >= 4 && width % 4 == 0 &&
                   width <= (1 << 20) && reinterpret_cast<uintptr_t>(table) % 16 == 0,
               "ha_qapply: the table must be 16-byte aligned with rows of a multiple of 4 floats");
    HA_REQUIRE(n_cur >= 0 && n_next >= 0 && n_cur <= n_max && n_next <= n_max && queue_n_cap >= 1 &&
                   queue_n_cap <= n_max && n_cur <= queue_n_cap && n_next <= queue_n_cap,
               "ha_qapply: at most %lld ids per batch and no more than the queue was sized for", (long long)n_max);
    HA_REQUIRE(n_cur == 0 || (perm_cur && grads && reinterpret_cast<uintptr_t>(grads) % 16 == 0),
               "ha_qapply: current batch needs its plan and 16-byte aligned gradients");
    HA_REQUIRE(n_next == 0 || (perm_next && next_out && reinterpret_cast<uintptr_t>(next_out) % 16 == 0),
               "ha_qapply: next batch needs its plan and a 16-byte aligned output");
    if (n_cur == 0 && n_next == 0) {
        if (done != nullptr)        // nothing to launch: the event still has to mark this point of the stream
            HA_CHECK_HIP(hipEventRecord(done, stream));
        return 0;
    }
    HA_REQUIRE(queue_cur != nullptr, "ha_qapply: the queue of this step is missing");
    QArgs a;
    memset(&a, 0, sizeof(a));
    a.epoch = epoch;
    a.err = err;
    a.table = table;
    a.rows = static_cast<uint64_t>(rows);
    a.width = static_cast<int>(width);
    a.lr = lr;
    a.dbg = dbg;
    const QLayout q = queue_layout(const_cast<void *>(queue_cur), queue_n_cap, width);
    a.qh = q.hdr;
    a.qcoop = q.coop;
    a.qwave = q.wave;
    a.qcopy = q.copy;
    a.qpart = q.part;
    a.qpcnt = q.pcnt;
    a.cap_coop = q.cap_coop;
    a.cap_wave = q.cap_wave;
    a.cap_copy = q.cap_copy;
    a.perm_a = perm_cur;
    a.n_a = static_cast<int>(n_cur);
    a.grads = grads;
    a.perm_g = perm_next;
    a.n_g = static_cast<int>(n_next);
    a.out = next_out;
    a.ncoop = n_cur >= kQLongC ? kQCoopSlots : 0;
    if (coop_items >= 0) {      // the caller knows the number of workgroup items (wide batches have thousands): one workgroup
        const int64_t most = 4096 / kQWpw;      // each, up to half of the chip's resident waves
        a.ncoop = static_cast<int>(coop_items < most ? coop_items : most);
    }
    // wave items <= ceil(width/512) * (n_cur + n_next); beyond kQWorkerMax workgroups the waves loop
    const int64_t bound = static_cast<int64_t>(ceil_div(width, 512)) * (n_cur + n_next);
    a.nworker = static_cast<int>(bound / kQWpw + 1 < kQWorkerMax ? bound / kQWpw + 1 : kQWorkerMax);
    // the caller knows how many wave items the queue holds (ha_qstep_queue_mirror): no workgroups that find nothing --
    // a shorter launch ramp, and free slots for the preparation launches that run beside the steps
    if (wave_items >= 0 && wave_items / kQWpw + 1 < a.nworker)
        a.nworker = static_cast<int>(wave_items / kQWpw + 1);
    // (sync = "flags": a launch whose queue is still being built POLLS -- workgroup 0 the epoch words, the others its verdict --
    // and while it does, no compute unit is EMPTY; the builder's workgroups fit only an empty one: 106 scalar registers a wave,
    // beside a workgroup of this kernel a SIMD's scalar file is short.  A builder that has not started by then starts when the
    // poll gives up, after 2 s, and the step fails (cleanly: nothing applied).  Measured: sporadic time-outs at blocks of 2 steps,
    // where the builder is only a block's worth of time ahead; none from blocks of 8.  Holding the builder kernels to 80 scalar
    // registers lets them in -- and costs the steps 12 % (headline 13.7 us against 12.2, wide shapes 52 / 56 against 36 / 47):
    // the preparation waiting for empty compute units is what keeps it out of the steps' way.  QueueStepPipeline therefore
    // orders blocks of fewer than 8 steps by events; docs/EXPERIMENTS.md round 6.)
    if (done != nullptr)
        // the event completes with THIS launch (the dispatch packet's own completion signal): no packet of its own on the
        // stream -- an event record between two launches of a stream costs what a short kernel costs
        hipExtLaunchKernelGGL(qapply_kernel, dim3(static_cast<unsigned>(a.ncoop + a.nworker)), dim3(kQWg), kQWpw * 64 * 4, stream,
                              nullptr, done, 0, a);
    else
        hipLaunchKernelGGL(qapply_kernel, dim3(static_cast<unsigned>(a.ncoop + a.nworker)), dim3(kQWg), kQWpw * 64 * 4, stream, a);
    HA_LAUNCH_CHECK();
    return 0;
}


// =====================================================================================================
// The WIDE path: batches of more than kQMax ids (BASELINE configs[2] / configs[3] on one GPU: 106,496 and 26,624 ids).
//
// A plan workgroup groups at most kQMax ids (its keys sit in registers, its hash table in LDS).  A larger batch is cut
// into P hash BUCKETS first -- bucket = top bits of key * 0x85EBCA6B, a multiplier of its own so that the table hash
// inside a bucket stays uniform -- by ONE stable multisplit of (key, position): tile histograms, then a scatter that
// ranks its tile with wave ballots (the order of the positions inside a bucket is the order of the batch, which is what
// keeps the occurrence lists in the reference's order).  Every bucket is then a small batch of its own:
//   plans   one workgroup per (batch, bucket), the SAME grouping body (qsort_finish_body<.., BUCKET>); unique keys,
//           counts, segment starts and first occurrences go to the bucket's slice of per-batch arrays, the occurrence
//           lists -- as positions of the whole batch -- to its slice of ONE list per batch (`gperm`);
//   queues  equal keys of two batches fall into equal buckets, so bucket p of the batch to apply joins bucket p of the
//           batch to look up and nothing else: two workgroups per (step, bucket), the SAME join body (qjoin_body<WIDE>),
//           appending to the step's queue with one atomic add per region;
//   apply   the same launch as the narrow path (qapply_kernel), items numbered across the buckets.
// No sort anywhere: the 43 us radix sort of a 106,496-id batch (four launches of 26 workgroups) becomes two partition
// launches and P-fold parallel plan / queue workgroups of ~1,000 ids each.  Limits: at most kQBigMax ids per batch
// (positions are 21-bit fields of an item); a BUCKET holds at most kQMax ids -- with P = n / 1024 rounded up to a power
// of two that leaves room for ~6,000 occurrences of hot keys per bucket (a key names a sample of a field once: c <= the
// batch size, 4,096 at configs[2]; two such keys in one bucket do not fit).  A batch with a bucket beyond that is NOT
// planned here: its queues carry flag 4 (header word and the pinned mirror, written with the counts by the bucket
// workgroup that finishes last), and the caller takes another path for the steps that touch it
// (ops.QueueStepPipeline: the sorted plan + ha_sgd_apply / ha_gather_* for that step -- rare, and correct).
// =====================================================================================================
constexpr int kQBigTile = 4096;
constexpr int kQBigBucketsMax = 128;
constexpr int64_t kQBigMax = 1 << 17;
constexpr int kQBigTilesMax = static_cast<int>(kQBigMax / kQBigTile);      // 32

static inline int qbig_buckets(int64_t n_cap) {     // ~1,024 ids per bucket on average: room for ~6,000 occurrences of hot keys
    int p = 4;      // (a multiple of four: a workgroup takes four buckets)
    while (p < kQBigBucketsMax && static_cast<int64_t>(p) * 1024 < n_cap)
        p *= 2;
    return p;
}
__device__ __forceinline__ uint32_t qbig_bucket(uint32_t key, int logp) {
    return (key * 0x85EBCA6Bu) >> (32 - logp);
}

struct QBigRef {        // one batch's wide-plan workspace (device pointers)
    const void *ids;
    int n;
    uint32_t *meta;     // [0]: a bucket holds more than kQMax ids (sticky until the next partition)
    uint32_t *boff;     // [P + 1] first position of every bucket in the bucket-ordered arrays
    uint32_t *thist;    // [tiles * P] ids per (tile, bucket)
    uint32_t *bkeys, *bpos;         // [n] keys / positions in bucket order (position order inside a bucket)
    int32_t *gperm;     // [n] occurrence lists of all buckets' groups, as positions of the batch
    uint32_t *uniq;     // [n] unique keys, bucket p's groups from boff[p]
    int32_t *counts;    // [n]
    int32_t *seg;       // [n + P + 1] bucket p's segment starts (local to the bucket) from boff[p] + p
    uint32_t *occ;      // [2 n] first occurrences per group (q_occ_pack)
    PlanHeader *bhdr;   // [P]
    PlanHeader *dummy;  // a header + a few words nobody reads (a quarter-workgroup without a bucket of its own writes here)
};
static inline size_t qbig_layout(void *ws, int64_t n_cap, QBigRef *r) {
    char *b = static_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = b ? b + off : nullptr;
        off += align_up(bytes, 256);
        return q;
    };
    const int P = qbig_buckets(n_cap);
    const size_t n4 = static_cast<size_t>(n_cap) * 4;
    QBigRef t;
    memset(&t, 0, sizeof(t));
    t.meta = reinterpret_cast<uint32_t *>(take(256));
    t.boff = reinterpret_cast<uint32_t *>(take((P + 1) * 4));
    t.thist = reinterpret_cast<uint32_t *>(take(static_cast<size_t>(kQBigTilesMax) * P * 4));
    t.bkeys = reinterpret_cast<uint32_t *>(take(n4));
    t.bpos = reinterpret_cast<uint32_t *>(take(n4));
    t.gperm = reinterpret_cast<int32_t *>(take(n4));
    t.uniq = reinterpret_cast<uint32_t *>(take(n4));
    t.counts = reinterpret_cast<int32_t *>(take(n4));
    t.seg = reinterpret_cast<int32_t *>(take(n4 + (P + 1) * 4));
    t.occ = reinterpret_cast<uint32_t *>(take(2 * n4));
    t.bhdr = reinterpret_cast<PlanHeader *>(take(static_cast<size_t>(P) * sizeof(PlanHeader)));
    t.dummy = reinterpret_cast<PlanHeader *>(take(2 * sizeof(PlanHeader)));
    if (r)
        *r = t;
    return off;
}

struct QBigBatch {
    int count, P, logp;
    QBigRef r[kQBatch];
};

template <typename IdT>
__global__ __launch_bounds__(1024) void qbpart_hist_kernel(const QBigBatch b) {
    __shared__ uint32_t s_h[kQBigBucketsMax];
    const QBigRef &r = b.r[blockIdx.y];
    const int n = r.n, base = static_cast<int>(blockIdx.x) * kQBigTile;
    if (base >= n)
        return;
    if (threadIdx.x < kQBigBucketsMax)
        s_h[threadIdx.x] = 0;
    __syncthreads();
    const IdT *ids = static_cast<const IdT *>(r.ids);
#pragma unroll
    for (int k = 0; k < kQBigTile / 1024; ++k) {
        const int i = base + k * 1024 + static_cast<int>(threadIdx.x);
        if (i < n)
            atomicAdd(&s_h[qbig_bucket(to_key<IdT>(ids[i]), b.logp)], 1u);
    }
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < b.P)
        r.thist[blockIdx.x * b.P + threadIdx.x] = s_h[threadIdx.x];
}

template <typename IdT>
__global__ __launch_bounds__(1024) void qbpart_scatter_kernel(const QBigBatch b) {
    __shared__ uint32_t s_base[kQBigBucketsMax], s_tot[kQBigBucketsMax];
    __shared__ uint32_t s_cnt[(kQBigTile / 64) * kQBigBucketsMax];      // [wave-row][bucket]
    const QBigRef &r = b.r[blockIdx.y];
    const int n = r.n, P = b.P, t = static_cast<int>(blockIdx.x), tid = static_cast<int>(threadIdx.x);
    const int T = (n + kQBigTile - 1) / kQBigTile;
    if (n == 0) {
        if (t == 0 && tid <= P) {
            r.boff[tid] = 0;
            if (tid == 0)
                r.meta[0] = 0;
        }
        return;
    }
    if (t >= T)
        return;
    uint32_t before = 0;
    if (tid < P) {
        uint32_t tot = 0;
        for (int q = 0; q < T; ++q) {
            const uint32_t v = r.thist[q * P + tid];
            tot += v;
            before += q < t ? v : 0u;
        }
        s_tot[tid] = tot;
    }
    for (int q = tid; q < (kQBigTile / 64) * kQBigBucketsMax; q += 1024)
        s_cnt[q] = 0;
    __syncthreads();
    if (tid < P) {
        uint32_t off = 0;
        for (int q = 0; q < tid; ++q)
            off += s_tot[q];
        s_base[tid] = off + before;
        if (t == 0) {
            r.boff[tid] = off;
            if (tid == P - 1)
                r.boff[P] = static_cast<uint32_t>(n);
        }
    }
    if (t == 0 && tid == 0) {
        uint32_t over = 0;
        for (int q = 0; q < P; ++q)
            over |= s_tot[q] > static_cast<uint32_t>(kQMax) ? 1u : 0u;
        r.meta[0] = over;
    }
    const IdT *ids = static_cast<const IdT *>(r.ids);
    const int base = t * kQBigTile, lane = lane_id(), w = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t key[kQBigTile / 1024], dig[kQBigTile / 1024], rank[kQBigTile / 1024];
#pragma unroll
    for (int k = 0; k < kQBigTile / 1024; ++k) {
        const int i = base + k * 1024 + tid;
        const bool valid = i < n;
        key[k] = to_key<IdT>(ids[min(i, n - 1)]);
        dig[k] = qbig_bucket(key[k], b.logp);
        unsigned long long m = __ballot(valid);
        for (int bit = 0; bit < b.logp; ++bit) {
            const bool on = (dig[k] >> bit) & 1u;
            const unsigned long long bb = __ballot(on);
            m &= on ? bb : ~bb;
        }
        rank[k] = static_cast<uint32_t>(__builtin_popcountll(m & below));
        if (valid && rank[k] == 0)
            s_cnt[(k * 16 + w) * kQBigBucketsMax + dig[k]] = static_cast<uint32_t>(__builtin_popcountll(m));
    }
    __syncthreads();
    if (tid < P) {       // exclusive prefix over the tile's 64 wave-rows, bucket by bucket
        uint32_t run = 0;
        for (int q = 0; q < kQBigTile / 64; ++q) {
            const uint32_t c = s_cnt[q * kQBigBucketsMax + tid];
            s_cnt[q * kQBigBucketsMax + tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kQBigTile / 1024; ++k) {
        const int i = base + k * 1024 + tid;
        if (i < n) {
            const uint32_t pos = s_base[dig[k]] + s_cnt[(k * 16 + w) * kQBigBucketsMax + dig[k]] + rank[k];
            r.bkeys[pos] = key[k];
            r.bpos[pos] = static_cast<uint32_t>(i);
        }
    }
}

__device__ __forceinline__ QPlan qbig_slice(const QBigRef &r, int p) {
    QPlan q;
    const uint32_t off = uniform(r.boff[p]), nb = uniform(r.boff[p + 1]) - off;
    q.hdr = r.bhdr + p;
    q.keys = r.bkeys + off;
    q.sorted = nullptr;
    q.uniq = r.uniq + off;
    q.perm = r.gperm + off;
    q.inverse = nullptr;
    q.counts = r.counts + off;
    q.seg = r.seg + off + p;
    q.upos = nullptr;
    q.occ = r.occ + 2 * static_cast<size_t>(off);
    q.n = static_cast<int>(nb);
    return q;
}

// Buckets of at most kQSubMax ids (all but those of hot keys: ~830 ids on average) are grouped FOUR to a workgroup, side by
// side, a quarter of its threads each -- a bucket's ~800 ids keep 256 threads busy, not 1024, and the preparation runs beside
// the steps.  The larger ones (a key with thousands of occurrences) get a workgroup each from a second launch.
// (Two instantiations, launched one after the other over the same grid: QUARTERS takes the workgroups whose four buckets are
// small, the other one the rest -- one kernel with both bodies needs more registers than leave room for the steps' waves.)
template <bool RANK_ATOMIC, bool QUARTERS>
__global__ __launch_bounds__(1024, 8) void qbplan_kernel(const QBigBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const QBigRef &r = b.r[blockIdx.y];
    if (QUARTERS) {     // grid: P / 4 workgroups per batch; a quarter whose bucket is a large one plans an empty batch instead
        const int sub = static_cast<int>(threadIdx.x) / kQSubThreads;
        QPlan q = qbig_slice(r, static_cast<int>(blockIdx.x) * 4 + sub);
        if (q.n > kQSubMax) {
            q.n = 0;
            q.hdr = r.dummy;
            q.seg = reinterpret_cast<int32_t *>(r.dummy + 1);
        }
        qsort_finish_body<uint32_t, RANK_ATOMIC, true, kQSubThreads, kQSubTabBits>(
            q.keys, q, s_dyn + sub * (kQSubPlanLds / 4), nullptr, r.bpos + (q.keys - r.bkeys));
        return;
    }
    // grid: P workgroups per batch, the ones with a small bucket leave at once
    const QPlan q = qbig_slice(r, static_cast<int>(blockIdx.x));
    if (q.n <= kQSubMax)
        return;
    if (q.n > kQMax) {      // (an oversized bucket: the partition raised meta[0]; the queue builder hands it on)
        if (threadIdx.x == 0) {
            q.hdr->n_unique = 0;
            q.hdr->reserved[kGroupedFlagWord] = 1;
            q.hdr->reserved[kOrderFlagWord] = 0;
            q.seg[0] = 0;
        }
        return;
    }
    qsort_finish_body<uint32_t, RANK_ATOMIC, true>(q.keys, q, s_dyn, nullptr, r.bpos + (q.keys - r.bkeys));
}

struct QBigZero {
    int count;
    QHeader *qh[kQBatch];
};
__global__ __launch_bounds__(64) void qbzero_kernel(const QBigZero z) {
    reinterpret_cast<uint32_t *>(z.qh[blockIdx.x])[threadIdx.x] = 0;       // the 64 words of a queue header
}

struct QBigJoinBatch {
    int count, width, P;
    uint64_t rows;
    uint32_t lds_bytes, cap_coop, cap_wave, cap_copy;
    QBigRef a[kQJoinBatch], g[kQJoinBatch];
    QHeader *qh[kQJoinBatch];
    QEntry *coop[kQJoinBatch], *wave[kQJoinBatch], *copy[kQJoinBatch];
    uint32_t *mirror[kQJoinBatch];
    uint32_t epoch[kQJoinBatch];
};
constexpr int kQSubJoinKeys = 8 * kQSubThreads;        // unique keys per side a quarter-workgroup joins (2,048)
constexpr size_t kQSubJoinLds = (size_t(1) << kQSubTabBits) * 4 + size_t(kQSubJoinKeys) * 8 + 32 * 4;      // 32,896
static_assert(kQSubJoinLds % 128 == 0, "the groups' LDS regions stay aligned");
// One workgroup per (step, part, FOUR buckets): quarters side by side where every bucket of the four has at most
// kQSubJoinKeys unique keys on either side, else one after the other (see qbplan_kernel).
template <bool QUARTERS>
__global__ __launch_bounds__(1024, QUARTERS ? 8 : 4) void qbqueue_kernel(const QBigJoinBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int i = static_cast<int>(blockIdx.y), p0 = static_cast<int>(blockIdx.x >> 1) * 4, part = static_cast<int>(blockIdx.x & 1);
    uint32_t bad = 0, most = 0;
    if (b.a[i].n > 0) {
        bad |= b.a[i].meta[0];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            most = max(most, uniform(static_cast<uint32_t>(b.a[i].bhdr[p0 + s].n_unique)));
    }
    if (b.g[i].n > 0) {
        bad |= b.g[i].meta[0];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            most = max(most, uniform(static_cast<uint32_t>(b.g[i].bhdr[p0 + s].n_unique)));
    }
    if (bad && p0 == 0 && part == 0 && threadIdx.x == 0) {       // a bucket beyond kQMax ids: its keys are in no queue
        atomicOr(&b.qh[i]->overflow_wave, 4u);
        if (b.mirror[i])
            b.mirror[i][3] = 4u;
    }
    constexpr bool quarters = QUARTERS;
    if ((most <= static_cast<uint32_t>(kQSubJoinKeys)) != QUARTERS)
        return;
    const int sub = quarters ? static_cast<int>(threadIdx.x) / kQSubThreads : 0;
    for (int s = 0; s < (quarters ? 1 : 4); ++s) {
        const int p = p0 + (quarters ? sub : s);
        QPlan pa, pg;
        memset(&pa, 0, sizeof(pa));
        memset(&pg, 0, sizeof(pg));
        uint32_t st_base = 0, fs_base = 0;
        if (b.a[i].n > 0) {
            pa = qbig_slice(b.a[i], p);
            st_base = uniform(b.a[i].boff[p]);
        }
        if (b.g[i].n > 0) {
            pg = qbig_slice(b.g[i], p);
            fs_base = uniform(b.g[i].boff[p]);
        }
        if (quarters) {
            qjoin_body<true, kQSubThreads, kQSubTabBits>(pa, pg, b.rows, b.width, b.qh[i], b.coop[i], b.wave[i], b.copy[i],
                                                         b.cap_coop, b.cap_wave, b.cap_copy, s_dyn + sub * (kQSubJoinLds / 4),
                                                         static_cast<uint32_t>(kQSubJoinLds), part, nullptr, b.mirror[i], st_base,
                                                         fs_base, b.epoch[i], 2u * static_cast<uint32_t>(b.P));
        } else {
            __syncthreads();      // the LDS of the bucket before
            qjoin_body<true>(pa, pg, b.rows, b.width, b.qh[i], b.coop[i], b.wave[i], b.copy[i], b.cap_coop, b.cap_wave,
                             b.cap_copy, s_dyn, b.lds_bytes, part, nullptr, b.mirror[i], st_base, fs_base, b.epoch[i],
                             2u * static_cast<uint32_t>(b.P));
        }
    }
}

template <typename IdT>
static int qbig_plan_batch(const IdT *const *ids, const int64_t *n, void *const *ws, int64_t n_cap, int64_t count,
                           hipStream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (ids && n && ws)), "ha_qbig_plan_batch: null pointer");
    HA_REQUIRE(n_cap >= 1 && n_cap <= kQBigMax, "ha_qbig_plan_batch: at most %lld ids per batch", (long long)kQBigMax);
    static DeviceOnce lds_allowed;
    if (lds_allowed.run([]() -> int {
            HA_ALLOW_LDS((qbplan_kernel<true, true>), 160 * 1024);
            HA_ALLOW_LDS((qbplan_kernel<false, true>), 160 * 1024);
            HA_ALLOW_LDS((qbplan_kernel<true, false>), 160 * 1024);
            HA_ALLOW_LDS((qbplan_kernel<false, false>), 160 * 1024);
            return 0;
        }))
        return -1;
    const bool ordered = lds_atomics_lane_ordered() != 0;
    const int P = qbig_buckets(n_cap);
    int logp = 0;
    while ((1 << logp) < P)
        ++logp;
    for (int64_t k0 = 0; k0 < count; k0 += kQBatch) {
        QBigBatch b;
        memset(&b, 0, sizeof(b));
        b.P = P;
        b.logp = logp;
        int64_t nmax = 0;
        for (int64_t k = k0; k < count && b.count < kQBatch; ++k) {
            HA_REQUIRE(n[k] >= 0 && n[k] <= n_cap, "ha_qbig_plan_batch: a batch is larger than its workspace was sized for");
            HA_REQUIRE(ws[k] && (n[k] == 0 || ids[k]), "ha_qbig_plan_batch: null pointer (batch %lld)", (long long)k);
            QBigRef &r = b.r[b.count++];
            qbig_layout(ws[k], n_cap, &r);
            r.ids = ids[k];
            r.n = static_cast<int>(n[k]);
            nmax = nmax > n[k] ? nmax : n[k];
        }
        if (b.count == 0)
            continue;
        const unsigned tiles = static_cast<unsigned>(nmax > 0 ? (nmax + kQBigTile - 1) / kQBigTile : 1);
        hipLaunchKernelGGL((qbpart_hist_kernel<IdT>), dim3(tiles, b.count), dim3(1024), 0, stream, b);
        HA_LAUNCH_CHECK();
        hipLaunchKernelGGL((qbpart_scatter_kernel<IdT>), dim3(tiles, b.count), dim3(1024), 0, stream, b);
        HA_LAUNCH_CHECK();
        if (ordered) {
            hipLaunchKernelGGL((qbplan_kernel<true, true>), dim3(P / 4, b.count), dim3(1024), 4 * kQSubPlanLds, stream, b);
            hipLaunchKernelGGL((qbplan_kernel<true, false>), dim3(P, b.count), dim3(1024), qsort_lds_bytes(kQMax), stream, b);
        } else {
            hipLaunchKernelGGL((qbplan_kernel<false, true>), dim3(P / 4, b.count), dim3(1024), 4 * kQSubPlanLds, stream, b);
            hipLaunchKernelGGL((qbplan_kernel<false, false>), dim3(P, b.count), dim3(1024), qsort_lds_bytes(kQMax), stream, b);
        }
        HA_LAUNCH_CHECK();
    }
    return 0;
}

static int qbig_queue_batch(int64_t rows, int64_t width, void *const *ws_a, const int64_t *n_a, void *const *ws_g,
                            const int64_t *n_g, void *const *queues, int64_t n_cap, int64_t count,
                            uint32_t *const *counts_host, const uint32_t *epochs, hipStream_t stream) {
    HA_REQUIRE(rows >= 0 && rows <= 0xFFFFFFFEll && width >= 4 && width % 4 == 0 && width <= (1 << 20),
               "ha_qbig_queue_batch: rows of a multiple of 4 floats");
    HA_REQUIRE(count >= 0 && (count == 0 || (ws_a && n_a && ws_g && n_g && queues)), "ha_qbig_queue_batch: null pointer");
    HA_REQUIRE(n_cap >= 1 && n_cap <= kQBigMax, "ha_qbig_queue_batch: bad capacity");
    static DeviceOnce lds_allowed;
    if (lds_allowed.run([]() -> int {
            HA_ALLOW_LDS(qbqueue_kernel<true>, 160 * 1024);
            HA_ALLOW_LDS(qbqueue_kernel<false>, 160 * 1024);
            return 0;
        }))
        return -1;
    const int P = qbig_buckets(n_cap);
    for (int64_t k0 = 0; k0 < count; k0 += kQJoinBatch) {
        QBigJoinBatch b;
        QBigZero z;
        memset(&b, 0, sizeof(b));
        memset(&z, 0, sizeof(z));
        b.rows = static_cast<uint64_t>(rows);
        b.width = static_cast<int>(width);
        b.P = P;
        for (int64_t k = k0; k < count && b.count < kQJoinBatch; ++k) {
            HA_REQUIRE(n_a[k] >= 0 && n_g[k] >= 0 && n_a[k] <= n_cap && n_g[k] <= n_cap,
                       "ha_qbig_queue_batch: a batch is larger than the queues were sized for");
            if (n_a[k] == 0 && n_g[k] == 0)
                continue;
            HA_REQUIRE(queues[k] && (n_a[k] == 0 || ws_a[k]) && (n_g[k] == 0 || ws_g[k]),
                       "ha_qbig_queue_batch: null pointer (step %lld)", (long long)k);
            const int i = b.count++;
            if (n_a[k] > 0)
                qbig_layout(ws_a[k], n_cap, &b.a[i]);
            if (n_g[k] > 0)
                qbig_layout(ws_g[k], n_cap, &b.g[i]);
            b.a[i].n = static_cast<int>(n_a[k]);
            b.g[i].n = static_cast<int>(n_g[k]);
            const QLayout q = queue_layout(queues[k], n_cap, width);
            b.qh[i] = q.hdr;
            z.qh[z.count++] = q.hdr;
            b.mirror[i] = counts_host ? counts_host[k] : nullptr;
            b.epoch[i] = epochs ? epochs[k] : 0u;
            b.coop[i] = q.coop;
            b.wave[i] = q.wave;
            b.copy[i] = q.copy;
            b.cap_coop = q.cap_coop;
            b.cap_wave = q.cap_wave;
            b.cap_copy = q.cap_copy;
        }
        if (b.count == 0)
            continue;
        const size_t need = qjoin_lds_bytes(kQMax, kQMax), res = qjoin_lds_resident_bytes(kQMax);
        const size_t lds1 = need > res ? need : res;
        b.lds_bytes = static_cast<uint32_t>(lds1);      // (what a whole-workgroup join has; the launch also fits four quarters)
        hipLaunchKernelGGL(qbzero_kernel, dim3(z.count), dim3(64), 0, stream, z);
        HA_LAUNCH_CHECK();
        hipLaunchKernelGGL(qbqueue_kernel<true>, dim3(2 * (P / 4), b.count), dim3(1024), 4 * kQSubJoinLds, stream, b);
        hipLaunchKernelGGL(qbqueue_kernel<false>, dim3(2 * (P / 4), b.count), dim3(1024), lds1, stream, b);
        HA_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace ha

using namespace ha;

extern "C" int64_t ha_qstep_max_ids(void) { return kQMax; }

// Per-device set-up of the ha_q* entry points, to be called once OUTSIDE any stream capture (ops.QueueStepPipeline does):
// the LDS attributes of the plan / queue kernels and the lane-order probe of the LDS atomics (a small synchronous launch
// on the null stream).  The entry points still do the same lazily on their first call.  Returns 1 if the atomic ranking
// is used, 0 for the ballot ranking, -1 on error.
extern "C" int ha_qstep_init(void) {
    const float *ids = nullptr;
    const uint64_t *ids64 = nullptr;
    if (qplan_batch<float>(&ids, nullptr, nullptr, 0, nullptr) || qplan_batch<uint64_t>(&ids64, nullptr, nullptr, 0, nullptr) ||
        qqueue_batch(0, 4, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, nullptr))
        return -1;
    return lds_atomics_lane_ordered();
}

extern "C" size_t ha_qstep_queue_bytes(int64_t n_cap, int64_t width) {
    if (n_cap < 1 || width < 4)
        return 0;
    return queue_layout(nullptr, n_cap, width).bytes;
}

extern "C" int ha_qplan_batch_f32ids(const float *const *ids, const int64_t *n, void *const *plans, int64_t count,
                                     ha_stream_t stream) {
    return qplan_batch<float>(ids, n, plans, count, as_stream(stream));
}
extern "C" int ha_qplan_batch_u64ids(const uint64_t *const *ids, const int64_t *n, void *const *plans, int64_t count,
                                     ha_stream_t stream) {
    return qplan_batch<uint64_t>(ids, n, plans, count, as_stream(stream));
}
extern "C" int ha_qqueue_batch(int64_t rows, int64_t width, void *const *plans_a, const int64_t *n_a,
                               void *const *plans_g, const int64_t *n_g, void *const *queues, int64_t queue_n_cap,
                               int64_t count, ha_stream_t stream) {
    return qqueue_batch(rows, width, plans_a, n_a, plans_g, n_g, queues, queue_n_cap, count, as_stream(stream));
}
extern "C" int ha_qapply(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads,
                         float lr, void *plan_next, int64_t n_next, float *next_out, const void *queue_cur,
                         int64_t queue_n_cap, ha_stream_t stream) {
    return qapply(table, rows, width, plan_cur, n_cur, grads, lr, plan_next, n_next, next_out, queue_cur, queue_n_cap,
                  as_stream(stream));
}
extern "C" int ha_qapply_sized(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads,
                               float lr, void *plan_next, int64_t n_next, float *next_out, const void *queue_cur,
                               int64_t queue_n_cap, int64_t wave_items, ha_stream_t stream) {
    return qapply(table, rows, width, plan_cur, n_cur, grads, lr, plan_next, n_next, next_out, queue_cur, queue_n_cap,
                  as_stream(stream), nullptr, wave_items);
}
// `count` consecutive steps enqueued by one call (the host side of a launch through ctypes costs more than the launch):
// per-step arrays of what ha_qapply_sized takes.
extern "C" int ha_qapply_steps(float *table, int64_t rows, int64_t width, float lr, int64_t queue_n_cap, int64_t count,
                               void *const *plan_cur, const int64_t *n_cur, const float *const *grads,
                               void *const *plan_next, const int64_t *n_next, float *const *next_out,
                               const void *const *queue_cur, const int64_t *wave_items, ha_stream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (plan_cur && n_cur && grads && plan_next && n_next && next_out && queue_cur)),
               "ha_qapply_steps: null pointer");
    for (int64_t k = 0; k < count; ++k)
        if (qapply(table, rows, width, plan_cur[k], n_cur[k], grads[k], lr, plan_next[k], n_next[k], next_out[k],
                   queue_cur[k], queue_n_cap, as_stream(stream), nullptr, wave_items ? wave_items[k] : -1))
            return -1;
    return 0;
}

// ---- ordering the two streams without a packet on the apply's stream --------------------------------------------------
// A caller that prepares queues on a side stream has two orders to keep.  (1) The apply of step c must not read queue c
// before it is built: ha_qqueue_batch_epochs tags every finished queue with the caller's epoch of its step, and an apply
// launch that is given the same epoch (ha_qapply_steps_sync / ha_qapply_sync) checks the tag before its first item
// instead of the stream waiting on an event.  (2) The side stream must not rewrite plans / queues that steps still read:
// the LAST launch of a block of steps completes an event of the library's own (ha_event_create) -- the event rides on
// that launch's dispatch packet (hipExtLaunchKernelGGL), nothing is recorded between launches --, and the side stream
// waits for it (ha_stream_wait_event).  The apply's stream then holds apply launches and nothing else: an event record
// + an event wait at every block boundary cost ~1 us per step at blocks of 16 (profiles/r04).
extern "C" void *ha_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
        return nullptr;
    return e;
}
extern "C" int ha_event_destroy(void *event) {
    if (event)
        HA_CHECK_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return 0;
}
extern "C" int ha_event_record(void *event, ha_stream_t stream) {
    HA_REQUIRE(event != nullptr, "ha_event_record: null event");
    HA_CHECK_HIP(hipEventRecord(static_cast<hipEvent_t>(event), as_stream(stream)));
    return 0;
}
extern "C" int ha_stream_wait_event(ha_stream_t stream, void *event) {
    HA_REQUIRE(event != nullptr, "ha_stream_wait_event: null event");
    HA_CHECK_HIP(hipStreamWaitEvent(as_stream(stream), static_cast<hipEvent_t>(event), 0));
    return 0;
}
extern "C" int ha_qqueue_batch_epochs(int64_t rows, int64_t width, void *const *plans_a, const int64_t *n_a,
                                      void *const *plans_g, const int64_t *n_g, void *const *queues, int64_t queue_n_cap,
                                      int64_t count, uint32_t *const *counts_host, const uint32_t *epochs,
                                      ha_stream_t stream) {
    return qqueue_batch(rows, width, plans_a, n_a, plans_g, n_g, queues, queue_n_cap, count, as_stream(stream), nullptr,
                        counts_host, epochs);
}
// ha_qapply_steps with, per step, the epoch its queue must carry (epochs may be NULL / an entry 0: no check), `err` = a
// pinned host word raised to 8 if a queue never became ready, and `done_event` (ha_event_create; NULL: none) completed by
// the last launch of the call.
extern "C" int ha_qapply_steps_sync(float *table, int64_t rows, int64_t width, float lr, int64_t queue_n_cap, int64_t count,
                                    void *const *plan_cur, const int64_t *n_cur, const float *const *grads,
                                    void *const *plan_next, const int64_t *n_next, float *const *next_out,
                                    const void *const *queue_cur, const int64_t *wave_items, const uint32_t *epochs,
                                    uint32_t *err, void *done_event, ha_stream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (plan_cur && n_cur && grads && plan_next && n_next && next_out && queue_cur)),
               "ha_qapply_steps_sync: null pointer");
    for (int64_t k = 0; k < count; ++k)
        if (qapply(table, rows, width, plan_cur[k], n_cur[k], grads[k], lr, plan_next[k], n_next[k], next_out[k],
                   queue_cur[k], queue_n_cap, as_stream(stream), nullptr, wave_items ? wave_items[k] : -1,
                   epochs ? epochs[k] : 0u, err, k + 1 == count ? static_cast<hipEvent_t>(done_event) : nullptr))
            return -1;
    if (count == 0 && done_event)
        HA_CHECK_HIP(hipEventRecord(static_cast<hipEvent_t>(done_event), as_stream(stream)));
    return 0;
}
// ha_qapply_steps_sync with the pinned count words of every step's queue beside the hints (one launch per step sizes its grid
// by `wave_items`)
extern "C" int ha_qapply_steps_counts(float *table, int64_t rows, int64_t width, float lr, int64_t queue_n_cap, int64_t count,
                                      void *const *plan_cur, const int64_t *n_cur, const float *const *grads,
                                      void *const *plan_next, const int64_t *n_next, float *const *next_out,
                                      const void *const *queue_cur, const int64_t *wave_items,
                                      const uint32_t *const *counts_host, const uint32_t *epochs, uint32_t *err,
                                      void *done_event, ha_stream_t stream) {
    (void)counts_host;
    return ha_qapply_steps_sync(table, rows, width, lr, queue_n_cap, count, plan_cur, n_cur, grads, plan_next, n_next, next_out,
                                queue_cur, wave_items, epochs, err, done_event, stream);
}
extern "C" int ha_qapply_sync(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads,
                              float lr, void *plan_next, int64_t n_next, float *next_out, const void *queue_cur,
                              int64_t queue_n_cap, int64_t wave_items, uint32_t epoch, uint32_t *err, void *done_event,
                              ha_stream_t stream) {
    return qapply(table, rows, width, plan_cur, n_cur, grads, lr, plan_next, n_next, next_out, queue_cur, queue_n_cap,
                  as_stream(stream), nullptr, wave_items, epoch, err, static_cast<hipEvent_t>(done_event));
}

// ha_qqueue_batch that also writes {wave items + 1, workgroup items + 1, copy items + 1} of step k's queue to the three
// pinned host words counts_host[k] (device-visible; NULL entries: none).  The caller zeroes the words before the call and
// reads them whenever it likes: 0 = not built yet.  ha_qapply_sized takes the sum as a hint only -- a stale or missing
// value costs time, never correctness (the waves of the apply loop over the items).
extern "C" int ha_qqueue_batch_counts(int64_t rows, int64_t width, void *const *plans_a, const int64_t *n_a,
                                      void *const *plans_g, const int64_t *n_g, void *const *queues, int64_t queue_n_cap,
                                      int64_t count, uint32_t *const *counts_host, ha_stream_t stream) {
    return qqueue_batch(rows, width, plans_a, n_a, plans_g, n_g, queues, queue_n_cap, count, as_stream(stream), nullptr,
                        counts_host);
}

// ha_qprep_*: one plan and / or one queue; ha_qstep_*: that followed by the step -- the serial forms
template <typename IdT>
static int qprep_one(int64_t rows, int64_t width, const IdT *ahead_ids, int64_t n_ahead, void *plan_ahead, void *plan_a,
                     int64_t n_a, void *plan_g, int64_t n_g, void *queue_build, int64_t queue_n_cap, hipStream_t s) {
    if (n_ahead > 0 && qplan_batch<IdT>(&ahead_ids, &n_ahead, &plan_ahead, 1, s))
        return -1;
    if (n_a > 0 || n_g > 0)
        return qqueue_batch(rows, width, &plan_a, &n_a, &plan_g, &n_g, &queue_build, queue_n_cap, 1, s);
    return 0;
}
extern "C" int ha_qprep_f32ids(int64_t rows, int64_t width, const float *ahead_ids, int64_t n_ahead, void *plan_ahead,
                               void *plan_a, int64_t n_a, void *plan_g, int64_t n_g, void *queue_build,
                               int64_t queue_n_cap, ha_stream_t stream) {
    return qprep_one<float>(rows, width, ahead_ids, n_ahead, plan_ahead, plan_a, n_a, plan_g, n_g, queue_build,
                            queue_n_cap, as_stream(stream));
}
extern "C" int ha_qprep_u64ids(int64_t rows, int64_t width, const uint64_t *ahead_ids, int64_t n_ahead, void *plan_ahead,
                               void *plan_a, int64_t n_a, void *plan_g, int64_t n_g, void *queue_build,
                               int64_t queue_n_cap, ha_stream_t stream) {
    return qprep_one<uint64_t>(rows, width, ahead_ids, n_ahead, plan_ahead, plan_a, n_a, plan_g, n_g, queue_build,
                               queue_n_cap, as_stream(stream));
}
extern "C" int ha_qstep_f32ids(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur,
                               const float *grads, float lr, void *plan_next, int64_t n_next, float *next_out,
                               const void *queue_cur, void *plan_b1, int64_t n_b1, void *queue_build,
                               int64_t queue_n_cap, const float *ahead_ids, int64_t n_ahead, void *plan_ahead,
                               ha_stream_t stream) {
    if (qprep_one<float>(rows, width, ahead_ids, n_ahead, plan_ahead, plan_next, n_next, plan_b1, n_b1, queue_build,
                         queue_n_cap, as_stream(stream)))
        return -1;
    return qapply(table, rows, width, plan_cur, n_cur, grads, lr, plan_next, n_next, next_out, queue_cur, queue_n_cap,
                  as_stream(stream));
}
extern "C" int ha_qstep_u64ids(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur,
                               const float *grads, float lr, void *plan_next, int64_t n_next, float *next_out,
                               const void *queue_cur, void *plan_b1, int64_t n_b1, void *queue_build,
                               int64_t queue_n_cap, const uint64_t *ahead_ids, int64_t n_ahead, void *plan_ahead,
                               ha_stream_t stream) {
    if (qprep_one<uint64_t>(rows, width, ahead_ids, n_ahead, plan_ahead, plan_next, n_next, plan_b1, n_b1, queue_build,
                            queue_n_cap, as_stream(stream)))
        return -1;
    return qapply(table, rows, width, plan_cur, n_cur, grads, lr, plan_next, n_next, next_out, queue_cur, queue_n_cap,
                  as_stream(stream));
}

// ---- the wide path (batches of up to ha_qbig_max_ids() ids): see the comment above qbig_buckets -------------------------
extern "C" int64_t ha_qbig_max_ids(void) { return kQBigMax; }
extern "C" size_t ha_qbig_plan_bytes(int64_t n_cap) {
    if (n_cap < 1 || n_cap > kQBigMax)
        return 0;
    return qbig_layout(nullptr, n_cap, nullptr);
}
extern "C" int ha_qbig_buckets(int64_t n_cap) { return n_cap >= 1 && n_cap <= kQBigMax ? qbig_buckets(n_cap) : 0; }
extern "C" int ha_qbig_plan_batch_f32ids(const float *const *ids, const int64_t *n, void *const *ws, int64_t n_cap,
                                         int64_t count, ha_stream_t stream) {
    return qbig_plan_batch<float>(ids, n, ws, n_cap, count, as_stream(stream));
}
extern "C" int ha_qbig_plan_batch_u64ids(const uint64_t *const *ids, const int64_t *n, void *const *ws, int64_t n_cap,
                                         int64_t count, ha_stream_t stream) {
    return qbig_plan_batch<uint64_t>(ids, n, ws, n_cap, count, as_stream(stream));
}
extern "C" int ha_qbig_queue_batch(int64_t rows, int64_t width, void *const *ws_a, const int64_t *n_a, void *const *ws_g,
                                   const int64_t *n_g, void *const *queues, int64_t n_cap, int64_t count,
                                   uint32_t *const *counts_host, const uint32_t *epochs, ha_stream_t stream) {
    return qbig_queue_batch(rows, width, ws_a, n_a, ws_g, n_g, queues, n_cap, count, counts_host, epochs, as_stream(stream));
}
extern "C" int ha_qbig_apply(float *table, int64_t rows, int64_t width, void *ws_cur, int64_t n_cur, const float *grads,
                             float lr, void *ws_next, int64_t n_next, float *next_out, const void *queue_cur, int64_t n_cap,
                             int64_t coop_items, uint32_t epoch, uint32_t *err, void *done_event, ha_stream_t stream) {
    HA_REQUIRE(n_cap >= 1 && n_cap <= kQBigMax && (n_cur == 0 || ws_cur) && (n_next == 0 || ws_next), "ha_qbig_apply: bad arguments");
    QBigRef a, g;
    memset(&a, 0, sizeof(a));
    memset(&g, 0, sizeof(g));
    if (n_cur > 0)
        qbig_layout(ws_cur, n_cap, &a);
    if (n_next > 0)
        qbig_layout(ws_next, n_cap, &g);
    return qapply_lists(table, rows, width, a.gperm, n_cur, grads, lr, g.gperm, n_next, next_out, queue_cur, n_cap, kQBigMax,
                        as_stream(stream), nullptr, -1, epoch, err, static_cast<hipEvent_t>(done_event), coop_items);
}
// a wide plan's per-bucket results for tests: bucket offsets [P + 1], then per bucket its number of unique keys
extern "C" int ha_qbig_plan_view(void *ws, int64_t n_cap, void **boff, void **bhdr, void **uniq, void **counts, void **seg,
                                 void **gperm, void **meta) {
    HA_REQUIRE(ws && n_cap >= 1 && n_cap <= kQBigMax, "ha_qbig_plan_view: bad arguments");
    QBigRef r;
    qbig_layout(ws, n_cap, &r);
    if (boff) *boff = r.boff;
    if (bhdr) *bhdr = r.bhdr;
    if (uniq) *uniq = r.uniq;
    if (counts) *counts = r.counts;
    if (seg) *seg = r.seg;
    if (gperm) *gperm = r.gperm;
    if (meta) *meta = r.meta;
    return 0;
}

// development aids: the items of a step with per-wave time stamps (dbg = device uint64[(workgroups) * 4 * 4], zeroed);
// one plan + one queue with the phase stamps of their workgroups (ph = device uint64[32], zeroed: plan at 0, the
// queue's two workgroups at 16 and 24)
extern "C" int ha_debug_qapply(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur,
                               const float *grads, float lr, void *plan_next, int64_t n_next, float *next_out,
                               const void *queue_cur, int64_t queue_n_cap, unsigned long long *dbg, ha_stream_t stream) {
    HA_REQUIRE(dbg != nullptr, "qapply timeline: null debug buffer");
    return qapply(table, rows, width, plan_cur, n_cur, grads, lr, plan_next, n_next, next_out, queue_cur, queue_n_cap,
                  as_stream(stream), dbg);
}
extern "C" int ha_debug_qprep_f32ids(int64_t rows, int64_t width, const float *ahead_ids, int64_t n_ahead,
                                     void *plan_ahead, void *plan_a, int64_t n_a, void *plan_g, int64_t n_g,
                                     void *queue_build, int64_t queue_n_cap, unsigned long long *ph, ha_stream_t stream) {
    HA_REQUIRE(ph != nullptr, "qprep phases: null debug buffer");
    if (n_ahead > 0 && qplan_batch<float>(&ahead_ids, &n_ahead, &plan_ahead, 1, as_stream(stream), ph))
        return -1;
    if (n_a > 0 || n_g > 0)
        return qqueue_batch(rows, width, &plan_a, &n_a, &plan_g, &n_g, &queue_build, queue_n_cap, 1, as_stream(stream),
                            ph + 16);
    return 0;
}

// queue header of a built queue: {wave items, workgroup items, long, medium, small, copy items} (tests / tools; device
// pointer)
extern "C" const uint32_t *ha_qstep_queue_header(const void *queue) {
    return reinterpret_cast<const uint32_t *>(queue);
}
