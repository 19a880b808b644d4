// ONE launch per training step: the backward of batch k beside the forward of batch k+1.
//
//   ha_sgd_push_pull_*  ==  ha_sgd_apply_finish(plan k, grads k)  then  ha_lookup_sort_*(ids k+1)
//
// which is the reference's embedding_push_pull (src/hetu_cache/src/cache.cc:356-422: push the
// gradients of one batch and pull the rows of the next in one request) applied to the HBM-resident
// table: table rows are updated in occurrence order exactly as cpu_SGDOptimizerSparseUpdate does
// (src/dnnl_ops/Optimizers.cpp:51-74) and the lookup of batch k+1 returns the rows AFTER that update
// (src/dnnl_ops/EmbeddingLookup.cpp:16-35 run behind it).  Results are bit-identical to the two
// separate launches; what changes is the schedule: the two row streams (gradient + table rows of batch
// k, table + output rows of batch k+1) overlap inside one grid and one dependent kernel boundary
// (1.7-1.9 us between streaming kernels on MI355X) disappears per step.
//
// Grid = four roles selected by blockIdx, 1024-thread workgroups, in this order:
//   [finish(k)] [apply(k)] [rank(k+1)] [gather(k+1)]
// Roughly two thirds of the positions of a Criteo batch name a row that the previous batch also
// updates, so the gather cannot simply run beside the apply.  Hand-off, per row:
//   * every batch owns a pending table (2^16 words, hashed by key).  When batch k is sorted (rank role
//     of the launch before), each occurrence of a key adds `nslice` units to its word;
//   * an apply wave that has written s of the row's 64-column slices for a key with L occurrences
//     stores them device-coherently (`sc1`, written through the XCD's L2), waits for the stores
//     (s_waitcnt vmcnt(0)) and gives L*s units back with an agent-scope atomic (signal_done,
//     scatter_dev.h).  The table returns to all-zero by itself at the end of the launch;
//   * a gather wave polls the words of its keys with relaxed agent-scope loads and reads the rows with
//     `sc1` loads only once they are 0 (a hash collision only makes it wait longer).
// Apply workgroups have lower block numbers than gather workgroups and never wait themselves, so
// every wave a gather wave waits for is resident or ahead of it in the dispatch order.  A poll loop
// that exceeds kSpinMax gives up and raises plan_next's header flag (header word 8) instead of hanging.
#include "plan_dev.h"
#include "gather_dev.h"
#include "scatter_dev.h"

namespace ha {

// fused.hip
template <int MODE>
int apply_finish_entry(float *dst, int64_t rows, int64_t width, void *plan_ws, int64_t n,
                       const float *grads, float lr, hipStream_t stream);

constexpr int kGatherLoads = 2;    // sixteen-byte loads a gather lane keeps in flight: one 2 KiB row per wave at
                                   // d = 512 (measured: 8 -> 18.4 us per step, 4 -> 16.2, 2 -> 15.6: the finer the
                                   // waves, the less a released row waits for its wave's other rows)
constexpr int kSpinMax = 1 << 18;   // polls of ~0.3 us each before a gather wave gives up (~0.1 s)

struct StepArgs {
    // table
    float *table;
    uint64_t rows;
    int width;
    // batch k: apply + finish
    const uint32_t *sorted;
    const int32_t *perm;
    int n_cur;
    const float *grads;
    float lr;
    PlanHeader *hdr;
    uint32_t *uniq;
    int32_t *seg, *counts, *inverse, *upos;
    uint32_t *pend_cur;     // words of batch k (nullptr: nothing to wait for)
    int nfin, napply;
    // batch k+1: rank + gather
    const void *next_ids;
    int n_next;
    float *out;
    uint32_t *nkeys, *nsorted;
    int32_t *nperm;
    PlanHeader *nhdr;
    uint32_t *pend_next;    // words batch k+1 registers in
    int nrank, ngather;
    int nv, nv_shift, group;  // vectors per row, log2(nv) or -1, positions per gather wave
    unsigned long long *dbg;  // tools/step_timeline.py only: {start, after-wait, end, role | xcc << 8} per wave
};

// the results of the inline-asm loads may be used only behind this (common.h, ld4_sc1_async)
__device__ __forceinline__ void wait_gather_loads(float4v (&v)[2]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
}
__device__ __forceinline__ void wait_gather_loads(float4v (&v)[4]) {
    wait_loads(v[0], v[1], v[2], v[3]);
}

// One wave copies `group` consecutive positions (group * nv <= kGatherLoads * 64 sixteen-byte vectors, all of
// a lane's loads in flight together), after the rows it needs have been released.
// (Copying the released positions of a wave first and the others as they follow was measured: no gain,
// the extra polls cost what the earlier copies save.)
template <typename IdT>
__device__ __forceinline__ void gather_wait_body(const StepArgs &a, int wave_index,
                                                 unsigned long long *t_mid = nullptr) {
    const int lane = lane_id();
    const int p0 = wave_index * a.group;
    if (p0 >= a.n_next)
        return;
    const int cnt = min(a.group, a.n_next - p0);
    const IdT *ids = static_cast<const IdT *>(a.next_ids);
    uint32_t key = 0;
    bool ok = false;
    if (lane < cnt) {
        const uint64_t r = id_to_row<IdT>(ids[p0 + lane]);
        ok = r < a.rows;
        key = ok ? static_cast<uint32_t>(r) : 0u;
    }
    if (a.pend_cur != nullptr) {
        const uint32_t *word = a.pend_cur + pend_slot(key);
        for (int spins = 0;; ++spins) {
            uint32_t c = 0;
            if (ok)
                c = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(c != 0) == 0ull)
                break;
            if (spins >= kSpinMax) {
                if (lane == 0)
                    a.nhdr->reserved[kHandoffFlagWord] = 1;   // hand-off timed out: the rows below may be stale
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (t_mid)
        *t_mid = __builtin_amdgcn_s_memrealtime();
    const int total = cnt * a.nv;
    float *dst = a.out + static_cast<uint64_t>(p0) * static_cast<uint64_t>(a.nv) * 4u;
    for (int base = 0; base < total; base += kGatherLoads * kWave) {
        float4v v[kGatherLoads];
        bool okv[kGatherLoads];
#pragma unroll
        for (int u = 0; u < kGatherLoads; ++u) {
            const int e = base + u * kWave + lane;
            const int ec = e < total ? e : total - 1;
            const int pos = a.nv_shift >= 0 ? (ec >> a.nv_shift) : (ec / a.nv);
            const int col = ec - pos * a.nv;
            const uint32_t k = static_cast<uint32_t>(__shfl(static_cast<int>(key), pos, kWave));
            okv[u] = __shfl(static_cast<int>(ok), pos, kWave) != 0;
            v[u] = ld4_sc1_async(a.table + (static_cast<uint64_t>(k) * static_cast<uint64_t>(a.nv) +
                                            static_cast<uint64_t>(col)) * 4u);
        }
        wait_gather_loads(v);
#pragma unroll
        for (int u = 0; u < kGatherLoads; ++u) {
            const int e = base + u * kWave + lane;
            if (e < total)
                st4_nt(dst + static_cast<uint64_t>(e) * 4u, okv[u] ? v[u] : float4v{0.f, 0.f, 0.f, 0.f});
        }
    }
}

// Block order: finish | apply | rank | gather.  Apply workgroups never wait and come before every gather
// workgroup, so whatever a gather wave waits for is resident or ahead of it in the dispatch order.
// (Measured and rejected: apply | gather | rank | finish (the rank tiles then form a 4 us tail), gather and
// rank alternating 1:1 or 2:1 (18 us), and a second design in which the launch that sorts a batch also
// finishes its plan behind its rank tiles and the apply maps persistent waves to UNIQUE keys from lists of
// short / medium / long runs (20-24 us: with a few hundred persistent waves each key pays three dependent
// round trips one after the other; one wave per sorted position has every row's loads in flight at t = 0).)
template <typename IdT>
__device__ __forceinline__ int step_roles(const StepArgs &a, uint32_t *s_dyn, unsigned long long *t_mid) {
    int b = blockIdx.x;
    if (b < a.nfin) {
        finish_block_body(a.sorted, a.perm, a.n_cur, a.hdr, a.uniq, a.seg, a.counts, a.inverse, a.upos, b, s_dyn);
        return 0;
    }
    b -= a.nfin;
    if (b < a.napply) {
        apply_body<kModeSgd, 4, false, kHandSignal>(a.table, a.rows, a.width, a.sorted, a.perm, nullptr, a.n_cur,
                                                    a.grads, a.lr, b, s_dyn, nullptr,
                                                    ApplyMaps{nullptr, nullptr, nullptr, nullptr, nullptr},
                                                    Hand{a.pend_cur, nullptr, nullptr, 0, nullptr});
        return 1;
    }
    b -= a.napply;
    if (b < a.nrank) {
        const IdT *ids = static_cast<const IdT *>(a.next_ids);
        rank_tile_body<IdT>(ids, a.n_next, a.nkeys, a.nsorted, a.nperm, b, s_dyn);
        // register the batch: every occurrence adds one unit per 64-column slice of its row
        const int i = b * kRankTile + static_cast<int>(threadIdx.x);
        if (threadIdx.x < kRankTile && i < a.n_next) {
            const uint64_t r = id_to_row<IdT>(ids[i]);
            if (r < a.rows)
                __hip_atomic_fetch_add(a.pend_next + pend_slot(static_cast<uint32_t>(r)),
                                       static_cast<uint32_t>((a.width + kWave - 1) / kWave), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
        return 2;
    }
    b -= a.nrank;
    gather_wait_body<IdT>(a, b * kPosPerBlock + static_cast<int>(threadIdx.x >> 6), t_mid);
    return 3;
}

template <typename IdT>
__global__ __launch_bounds__(1024, 8) void step_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    step_roles<IdT>(a, s_dyn, nullptr);
}

// the same grid with per-wave time stamps (development aid, tools/step_timeline.py)
template <typename IdT>
__global__ __launch_bounds__(1024, 8) void step_timeline_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tm = 0;
    const int role = step_roles<IdT>(a, s_dyn, &tm);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane_id() == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *d = a.dbg + (static_cast<size_t>(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 4;
        d[0] = t0;
        d[1] = tm;
        d[2] = t1;
        d[3] = static_cast<unsigned long long>(role) | (static_cast<unsigned long long>(xcc & 0xF) << 8);
    }
}

// The in-launch hand-off needs every 128-byte line of the table to belong to ONE row (a line is written
// whole by one store instruction and is never in a reader's L2 before its last write of the launch):
// rows of a multiple of 32 floats in a 128-byte aligned table.  Everything else takes the separate launches.
static bool step_fast(int64_t n, int64_t width, const void *table) {
    // above kBucketMin ids the rank-by-counting tiles of the single launch (O(n^2)) lose against the bucket
    // sort of the separate launches
    return n > 0 && n <= kBucketMin && width % 32 == 0 && width < (1 << 30) &&
           reinterpret_cast<uintptr_t>(table) % 128 == 0;
}

template <typename IdT>
static int push_pull(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur,
                     const float *grads, float lr, void *pend_cur, const IdT *next_ids, int64_t n_next,
                     float *next_out, void *plan_next, void *pend_next, hipStream_t stream,
                     int (*lookup_sort_plain)(const float *, int64_t, int64_t, const IdT *, int64_t, float *,
                                              void *, ha_stream_t),
                     unsigned long long *dbg = nullptr) {
    HA_REQUIRE(rows >= 0 && rows < (1ll << 32) && width >= 1 && n_cur >= 0 && n_next >= 0,
               "push_pull: bad sizes");
    HA_REQUIRE(n_cur == 0 || (plan_cur && grads && table), "push_pull: null pointer (current batch)");
    HA_REQUIRE(n_next == 0 || (plan_next && next_ids && next_out && table), "push_pull: null pointer (next batch)");
    const bool fast_cur = step_fast(n_cur, width, table), fast_next = step_fast(n_next, width, table);
    HA_REQUIRE(!(fast_cur && !pend_cur) && !(fast_next && !pend_next), "push_pull: null pending table");
    if (fast_cur || fast_next)
        HA_REQUIRE((!fast_cur || reinterpret_cast<uintptr_t>(grads) % 16 == 0) &&
                       (!fast_next || reinterpret_cast<uintptr_t>(next_out) % 16 == 0),
                   "push_pull: grads and out must be 16-byte aligned");
    // batches outside the single-launch regime run as the separate launches (their plans were never
    // registered in a pending table, so nothing waits for them)
    if (n_cur > 0 && !fast_cur) {
        if (apply_finish_entry<kModeSgd>(table, rows, width, plan_cur, n_cur, grads, lr, stream))
            return -1;
    } else if (n_cur == 0 && plan_cur) {
        if (ha_plan_finish(plan_cur, 0, stream))
            return -1;
    }
    if (fast_cur || fast_next) {
        StepArgs a;
        memset(&a, 0, sizeof(a));
        a.table = table;
        a.rows = static_cast<uint64_t>(rows);
        a.width = static_cast<int>(width);
        size_t lds = kFinishLdsWords * 4;
        if (fast_cur) {
            PlanPtrs p = plan_layout(plan_cur, n_cur);
            a.sorted = p.sorted;
            a.perm = p.perm;
            a.n_cur = static_cast<int>(n_cur);
            a.grads = grads;
            a.lr = lr;
            a.hdr = p.hdr;
            a.uniq = p.uniq;
            a.seg = p.seg;
            a.counts = p.counts;
            a.inverse = p.inverse;
            a.upos = p.upos;
            a.pend_cur = static_cast<uint32_t *>(pend_cur);
            a.nfin = finish_blocks(a.n_cur);
            a.napply = (a.n_cur + kPosPerBlock - 1) / kPosPerBlock;
            lds = kApplyLdsBytes;
        }
        if (fast_next) {
            PlanPtrs q = plan_layout(plan_next, n_next);
            a.next_ids = next_ids;
            a.n_next = static_cast<int>(n_next);
            a.out = next_out;
            a.nkeys = q.keys;
            a.nsorted = q.sorted;
            a.nperm = q.perm;
            a.nhdr = q.hdr;
            a.pend_next = static_cast<uint32_t *>(pend_next);
            a.nrank = (a.n_next + kRankTile - 1) / kRankTile;
            a.nv = static_cast<int>(width / 4);
            a.nv_shift = -1;
            for (int s = 0; s < 30; ++s)
                if (a.nv == (1 << s))
                    a.nv_shift = s;
            {
                const int vecs = kGatherLoads * kWave;   // vectors one wave copies per round trip
                a.group = a.nv >= vecs ? 1 : (vecs / a.nv > kWave ? kWave : vecs / a.nv);
            }
            const int waves = (a.n_next + a.group - 1) / a.group;
            a.ngather = (waves + kPosPerBlock - 1) / kPosPerBlock;
            lds = lds > rank_small_lds_bytes(a.n_next) ? lds : rank_small_lds_bytes(a.n_next);
        }
        const unsigned blocks = static_cast<unsigned>(a.nfin + a.napply + a.nrank + a.ngather);
        static DeviceOnce lds_allowed;   // once per device, and outside any stream capture (the first call is eager)
        if (lds_allowed.run([]() -> int {
                HA_ALLOW_LDS((step_kernel<IdT>), 160 * 1024);
                return 0;
            }))
            return -1;
        if (dbg) {
            a.dbg = dbg;
            HA_ALLOW_LDS((step_timeline_kernel<IdT>), 160 * 1024);
            hipLaunchKernelGGL((step_timeline_kernel<IdT>), dim3(blocks), dim3(1024), lds, stream, a);
        } else {
            hipLaunchKernelGGL((step_kernel<IdT>), dim3(blocks), dim3(1024), lds, stream, a);
        }
        HA_LAUNCH_CHECK();
    }
    if (n_next > 0 && !fast_next)
        return lookup_sort_plain(table, rows, width, next_ids, n_next, next_out, plan_next, stream);
    return 0;
}


// =====================================================================================================
// ha_step_*: the same step with THREE batches of lookahead and no waiting inside the launch.
//
//   launch k:  [clear table k+3] [finish(k+2) + key table k+2] [apply(k) + forward to out(k+1)] [rank(k+3)]
//              [gather(k+1) \ k]
//
// Batch k+1 was sorted by launch k-2 and its plan finished by launch k-1, which also entered its unique
// keys in the batch's KEY TABLE (common.h: key -> first sorted position, occurrences).  So when launch k
// starts, the wave that ends up with the final values of a row of batch k in its registers (scatter_dev.h,
// kHandForward) finds the key's positions in batch k+1 with one probe and writes the row to every output
// row of that batch that names it; the gather role copies only the rows of batch k+1 that batch k does
// NOT touch (membership = the key table of batch k).  Compared with step_kernel above: no pending words,
// no polling, no device-coherent stores, no atomics on shared words (a key table is filled by the finish
// blocks: one claim per unique key), the two thirds of the positions both batches share are not read back
// from HBM, and rows narrower than a 128-byte line need no special case (nothing written in the launch is
// read in it).  Nothing in the launch needs the rank tiles' result.
//
// Measured (MI355X, wdl_criteo bs=256 d=512, same box): 15.9-16.6 us per step against 15.4 us of step_kernel.
// Forwarding costs the applying waves -- the waves the step already waits for -- 2.4 us (with forwarding off:
// 14.3 us), the gather of the remaining third of the rows 1.3 us, and apply + rank + finish alone take 13.1 us:
// the step is bound by wave slots (the chip's 8,192 are full of apply and rank waves that sit on 2-3 us memory
// round trips), not by the bytes the forwarding saves.  Kept because it runs ANY table in one launch (rows
// narrower than a line, unaligned tables) and has no waiting, hence no time-out path.
// =====================================================================================================
constexpr int kStepMax = 12288;        // ids per batch (key table at <= 37.5 % load)
constexpr int kClearBlocks = 4;        // workgroups that clear one key table (512 KiB)

struct FwdArgs {
    float *table;
    uint64_t rows;
    int width;
    // batch k: apply
    const uint32_t *sorted;
    const int32_t *perm;
    int n_cur;
    const float *grads;
    float lr;
    const uint4 *tab_cur;      // keys of batch k (nullptr: the gather copies every row)
    int napply;
    // batch k+1: forward + gather
    const uint32_t *nkeys;
    const int32_t *nperm;
    int n_next;
    float *out;
    const uint4 *tab_next;
    int ngather, nv, nv_shift, group;
    // batch k+2: finish + key table
    const uint32_t *fsorted;
    const int32_t *fperm;
    int n_fin;
    PlanHeader *fhdr;
    uint32_t *funiq;
    int32_t *fseg, *fcounts, *finverse, *fupos;
    uint32_t *tab_fin;
    int nfin;
    // batch k+3: rank
    const void *ahead_ids;
    int n_ahead;
    uint32_t *akeys, *asorted;
    int32_t *aperm;
    int nrank;
    // the table whose batch is done
    uint4 *tab_clear;
    int nclear;
    unsigned long long *dbg;
};

// One wave copies `group` consecutive positions of batch k+1 -- those whose key batch k does not update.
__device__ __forceinline__ void gather_rest_body(const FwdArgs &a, int wave_index, unsigned long long *t_mid) {
    const int lane = lane_id();
    const int p0 = wave_index * a.group;
    if (p0 >= a.n_next)
        return;
    const int cnt = min(a.group, a.n_next - p0);
    uint32_t key = 0;
    bool ok = false, skip = true;
    if (lane < cnt) {
        key = a.nkeys[p0 + lane];
        ok = key < a.rows;
        skip = false;
        if (ok && a.tab_cur != nullptr) {
            uint32_t sl = tab_slot(key);
            for (;;) {
                const uint32_t k = a.tab_cur[sl].x;
                if (k == key) {
                    skip = true;   // forwarded by the wave that applies the key
                    break;
                }
                if (k == kTabEmpty)
                    break;
                sl = (sl + 1) & kTabMask;
            }
        }
    }
    if (t_mid)
        *t_mid = __builtin_amdgcn_s_memrealtime();
    if (__ballot(!skip) == 0ull)
        return;
    const int total = cnt * a.nv;
    float *dst = a.out + static_cast<uint64_t>(p0) * static_cast<uint64_t>(a.nv) * 4u;
    for (int base = 0; base < total; base += kGatherLoads * kWave) {
        float4v v[kGatherLoads];
        bool okv[kGatherLoads], sk[kGatherLoads];
#pragma unroll
        for (int u = 0; u < kGatherLoads; ++u) {
            const int e = base + u * kWave + lane;
            const int ec = e < total ? e : total - 1;
            const int pos = a.nv_shift >= 0 ? (ec >> a.nv_shift) : (ec / a.nv);
            const int col = ec - pos * a.nv;
            const uint32_t k = static_cast<uint32_t>(__shfl(static_cast<int>(key), pos, kWave));
            okv[u] = __shfl(static_cast<int>(ok), pos, kWave) != 0;
            sk[u] = __shfl(static_cast<int>(skip), pos, kWave) != 0;
            v[u] = ld4(a.table + (static_cast<uint64_t>(okv[u] ? k : 0u) * static_cast<uint64_t>(a.nv) +
                                  static_cast<uint64_t>(col)) * 4u);
        }
#pragma unroll
        for (int u = 0; u < kGatherLoads; ++u) {
            const int e = base + u * kWave + lane;
            if (e < total && !sk[u])
                st4_nt(dst + static_cast<uint64_t>(e) * 4u, okv[u] ? v[u] : float4v{0.f, 0.f, 0.f, 0.f});
        }
    }
}

template <typename IdT>
__device__ __forceinline__ int fwd_roles(const FwdArgs &a, uint32_t *s_dyn, unsigned long long *t_mid) {
    int b = blockIdx.x;
    if (b < a.nclear) {
        // all-ones = empty (common.h); 2^kTabBits entries of 16 bytes over nclear workgroups
        const int per = (1 << kTabBits) / a.nclear;
        for (int e = threadIdx.x; e < per; e += 1024)
            a.tab_clear[b * per + e] = uint4{kTabEmpty, kTabEmpty, kTabEmpty, kTabEmpty};
        return 4;
    }
    b -= a.nclear;
    if (b < a.nfin) {
        finish_block_body(a.fsorted, a.fperm, a.n_fin, a.fhdr, a.funiq, a.fseg, a.fcounts, a.finverse, a.fupos, b,
                          s_dyn, nullptr, nullptr, a.tab_fin, a.rows);
        return 0;
    }
    b -= a.nfin;
    // apply | rank | gather.  Measured alternatives (same box, 16.5 us for this order): rank | apply | gather 18.5,
    // apply | gather | rank 18.6, gather | apply | rank 17.8, and every workgroup doing a rank tile, then its 16
    // sorted positions, then its 16 output rows (all 427 workgroups resident from the start) 19.3 -- the roles'
    // latency chains then add up inside each workgroup instead of overlapping across workgroups.
    if (b < a.napply) {
        apply_body<kModeSgd, 4, false, kHandForward>(a.table, a.rows, a.width, a.sorted, a.perm, nullptr, a.n_cur,
                                                     a.grads, a.lr, b, s_dyn, nullptr,
                                                     ApplyMaps{nullptr, nullptr, nullptr, nullptr, nullptr},
                                                     Hand{nullptr, a.tab_next, a.nperm, a.n_next, a.out});
        return 1;
    }
    b -= a.napply;
    if (b < a.nrank) {
        rank_tile_body<IdT>(static_cast<const IdT *>(a.ahead_ids), a.n_ahead, a.akeys, a.asorted, a.aperm, b, s_dyn);
        return 2;
    }
    b -= a.nrank;
    gather_rest_body(a, b * kPosPerBlock + static_cast<int>(threadIdx.x >> 6), t_mid);
    return 3;
}

template <typename IdT>
__global__ __launch_bounds__(1024, 8) void step_fwd_kernel(const FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    fwd_roles<IdT>(a, s_dyn, nullptr);
}

template <typename IdT>
__global__ __launch_bounds__(1024, 8) void step_fwd_timeline_kernel(const FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tm = 0;
    const int role = fwd_roles<IdT>(a, s_dyn, &tm);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane_id() == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *d = a.dbg + (static_cast<size_t>(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 4;
        d[0] = t0;
        d[1] = tm;
        d[2] = t1;
        d[3] = static_cast<unsigned long long>(role) | (static_cast<unsigned long long>(xcc & 0xF) << 8);
    }
}

template <typename IdT>
static int step_fwd(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads,
                    float lr, const void *tab_cur, void *plan_next, int64_t n_next, float *next_out,
                    const void *tab_next, void *plan_fin, int64_t n_fin, void *tab_fin, const IdT *ahead_ids,
                    int64_t n_ahead, void *plan_ahead, void *tab_clear, hipStream_t stream,
                    unsigned long long *dbg = nullptr) {
    HA_REQUIRE(table != nullptr && rows >= 0 && rows <= 0xFFFFFFFEll && width >= 4 && width % 4 == 0 &&
                   width < (1 << 30) && reinterpret_cast<uintptr_t>(table) % 16 == 0,
               "ha_step: the table must be 16-byte aligned with rows of a multiple of 4 floats "
               "(other tables: ha_sgd_push_pull_* or the separate launches)");
    HA_REQUIRE(n_cur >= 0 && n_next >= 0 && n_fin >= 0 && n_ahead >= 0 && n_cur <= kStepMax &&
                   n_next <= kStepMax && n_fin <= kStepMax && n_ahead <= kStepMax,
               "ha_step: at most %d ids per batch (larger batches: ha_lookup_sort_* + ha_sgd_apply_finish)", kStepMax);
    HA_REQUIRE(n_cur == 0 || (plan_cur && grads && reinterpret_cast<uintptr_t>(grads) % 16 == 0),
               "ha_step: current batch needs its plan and 16-byte aligned gradients");
    HA_REQUIRE(n_next == 0 || (plan_next && next_out && reinterpret_cast<uintptr_t>(next_out) % 16 == 0),
               "ha_step: next batch needs its plan and a 16-byte aligned output");
    HA_REQUIRE(n_next == 0 || n_cur == 0 || (tab_cur && tab_next),
               "ha_step: the key tables of the current and the next batch are needed to forward rows");
    HA_REQUIRE(n_fin == 0 || (plan_fin && tab_fin), "ha_step: null pointer (batch to finish)");
    HA_REQUIRE(n_ahead == 0 || (ahead_ids && plan_ahead), "ha_step: null pointer (batch ahead)");
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.table = table;
    a.rows = static_cast<uint64_t>(rows);
    a.width = static_cast<int>(width);
    size_t lds = kFinishLdsWords * 4;
    if (n_cur > 0) {
        PlanPtrs p = plan_layout(plan_cur, n_cur);
        a.sorted = p.sorted;
        a.perm = p.perm;
        a.n_cur = static_cast<int>(n_cur);
        a.grads = grads;
        a.lr = lr;
        a.napply = (a.n_cur + kPosPerBlock - 1) / kPosPerBlock;
        lds = kApplyLdsBytes;
    }
    if (n_next > 0) {
        PlanPtrs q = plan_layout(plan_next, n_next);
        a.nkeys = q.keys;
        a.nperm = q.perm;
        a.n_next = static_cast<int>(n_next);
        a.out = next_out;
        if (n_cur > 0) {
            a.tab_cur = static_cast<const uint4 *>(tab_cur);
            a.tab_next = static_cast<const uint4 *>(tab_next);
        }
        a.nv = static_cast<int>(width / 4);
        a.nv_shift = -1;
        for (int s = 0; s < 30; ++s)
            if (a.nv == (1 << s))
                a.nv_shift = s;
        const int vecs = kGatherLoads * kWave;
        a.group = a.nv >= vecs ? 1 : (vecs / a.nv > kWave ? kWave : vecs / a.nv);
        const int waves = (a.n_next + a.group - 1) / a.group;
        a.ngather = (waves + kPosPerBlock - 1) / kPosPerBlock;
    }
    if (n_fin > 0) {
        PlanPtrs f = plan_layout(plan_fin, n_fin);
        a.fsorted = f.sorted;
        a.fperm = f.perm;
        a.n_fin = static_cast<int>(n_fin);
        a.fhdr = f.hdr;
        a.funiq = f.uniq;
        a.fseg = f.seg;
        a.fcounts = f.counts;
        a.finverse = f.inverse;
        a.fupos = f.upos;
        a.tab_fin = static_cast<uint32_t *>(tab_fin);
        a.nfin = finish_blocks(a.n_fin);
    }
    if (n_ahead > 0) {
        PlanPtrs r = plan_layout(plan_ahead, n_ahead);
        a.ahead_ids = ahead_ids;
        a.n_ahead = static_cast<int>(n_ahead);
        a.akeys = r.keys;
        a.asorted = r.sorted;
        a.aperm = r.perm;
        a.nrank = (a.n_ahead + kRankTile - 1) / kRankTile;
        lds = lds > rank_small_lds_bytes(a.n_ahead) ? lds : rank_small_lds_bytes(a.n_ahead);
    }
    if (tab_clear) {
        a.tab_clear = static_cast<uint4 *>(tab_clear);
        a.nclear = kClearBlocks;
    }
    const unsigned blocks = static_cast<unsigned>(a.nclear + a.nfin + a.napply + a.nrank + a.ngather);
    if (blocks == 0)
        return 0;
    static DeviceOnce lds_allowed;   // once per device, and outside any stream capture (the first call is eager)
    if (lds_allowed.run([]() -> int {
            HA_ALLOW_LDS((step_fwd_kernel<IdT>), 160 * 1024);
            return 0;
        }))
        return -1;
    if (dbg) {
        a.dbg = dbg;
        HA_ALLOW_LDS((step_fwd_timeline_kernel<IdT>), 160 * 1024);
        hipLaunchKernelGGL((step_fwd_timeline_kernel<IdT>), dim3(blocks), dim3(1024), lds, stream, a);
    } else {
        hipLaunchKernelGGL((step_fwd_kernel<IdT>), dim3(blocks), dim3(1024), lds, stream, a);
    }
    HA_LAUNCH_CHECK();
    return 0;
}

}  // namespace ha

using namespace ha;

extern "C" size_t ha_pend_bytes(void) {
    return sizeof(uint32_t) << kPendBits;
}

extern "C" int64_t *ha_plan_handoff_timeout(void *plan_ws) {
    return plan_ws ? &static_cast<PlanHeader *>(plan_ws)->reserved[kHandoffFlagWord] : nullptr;
}

extern "C" int ha_pend_reset(void *pend, ha_stream_t stream) {
    HA_REQUIRE(pend != nullptr, "ha_pend_reset: null pointer");
    HA_CHECK_HIP(hipMemsetAsync(pend, 0, ha_pend_bytes(), as_stream(stream)));
    return 0;
}

extern "C" int ha_sgd_push_pull_f32ids(float *table, int64_t rows, int64_t width, void *plan_cur,
                                       int64_t n_cur, const float *grads, float lr, void *pend_cur,
                                       const float *next_ids, int64_t n_next, float *next_out,
                                       void *plan_next, void *pend_next, ha_stream_t stream) {
    return push_pull<float>(table, rows, width, plan_cur, n_cur, grads, lr, pend_cur, next_ids, n_next,
                            next_out, plan_next, pend_next, as_stream(stream), ha_lookup_sort_f32ids);
}

// development aid: ha_sgd_push_pull_f32ids with per-wave time stamps; dbg = uint64[blocks * 16 * 4]
extern "C" int ha_debug_step_timeline(float *table, int64_t rows, int64_t width, void *plan_cur,
                                      int64_t n_cur, const float *grads, float lr, void *pend_cur,
                                      const float *next_ids, int64_t n_next, float *next_out,
                                      void *plan_next, void *pend_next, unsigned long long *dbg,
                                      ha_stream_t stream) {
    HA_REQUIRE(dbg != nullptr, "step timeline: null debug buffer");
    return push_pull<float>(table, rows, width, plan_cur, n_cur, grads, lr, pend_cur, next_ids, n_next,
                            next_out, plan_next, pend_next, as_stream(stream), ha_lookup_sort_f32ids, dbg);
}

extern "C" int ha_sgd_push_pull_u64ids(float *table, int64_t rows, int64_t width, void *plan_cur,
                                       int64_t n_cur, const float *grads, float lr, void *pend_cur,
                                       const uint64_t *next_ids, int64_t n_next, float *next_out,
                                       void *plan_next, void *pend_next, ha_stream_t stream) {
    return push_pull<uint64_t>(table, rows, width, plan_cur, n_cur, grads, lr, pend_cur, next_ids, n_next,
                               next_out, plan_next, pend_next, as_stream(stream), ha_lookup_sort_u64ids);
}

extern "C" int ha_lookup_sort_pend_f32ids(const float *table, int64_t rows, int64_t width, const float *ids,
                                          int64_t n, float *out, void *plan_ws, void *pend,
                                          ha_stream_t stream) {
    return push_pull<float>(const_cast<float *>(table), rows, width, nullptr, 0, nullptr, 0.f, nullptr, ids, n,
                            out, plan_ws, pend, as_stream(stream), ha_lookup_sort_f32ids);
}

extern "C" int ha_lookup_sort_pend_u64ids(const float *table, int64_t rows, int64_t width,
                                          const uint64_t *ids, int64_t n, float *out, void *plan_ws,
                                          void *pend, ha_stream_t stream) {
    return push_pull<uint64_t>(const_cast<float *>(table), rows, width, nullptr, 0, nullptr, 0.f, nullptr, ids,
                               n, out, plan_ws, pend, as_stream(stream), ha_lookup_sort_u64ids);
}

// ---- two batches of lookahead (see step_fwd above) ----------------------------------------------------
extern "C" size_t ha_step_tab_bytes(void) {
    return sizeof(uint4) << kTabBits;
}

extern "C" int64_t ha_step_max_ids(void) {
    return kStepMax;
}

extern "C" int ha_step_tab_reset(void *tab, ha_stream_t stream) {
    HA_REQUIRE(tab != nullptr, "ha_step_tab_reset: null pointer");
    HA_CHECK_HIP(hipMemsetAsync(tab, 0xFF, ha_step_tab_bytes(), as_stream(stream)));
    return 0;
}

extern "C" int ha_step_f32ids(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur,
                              const float *grads, float lr, const void *tab_cur, void *plan_next, int64_t n_next,
                              float *next_out, const void *tab_next, void *plan_fin, int64_t n_fin, void *tab_fin,
                              const float *ahead_ids, int64_t n_ahead, void *plan_ahead, void *tab_clear,
                              ha_stream_t stream) {
    return step_fwd<float>(table, rows, width, plan_cur, n_cur, grads, lr, tab_cur, plan_next, n_next, next_out,
                           tab_next, plan_fin, n_fin, tab_fin, ahead_ids, n_ahead, plan_ahead, tab_clear,
                           as_stream(stream));
}

extern "C" int ha_step_u64ids(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur,
                              const float *grads, float lr, const void *tab_cur, void *plan_next, int64_t n_next,
                              float *next_out, const void *tab_next, void *plan_fin, int64_t n_fin, void *tab_fin,
                              const uint64_t *ahead_ids, int64_t n_ahead, void *plan_ahead, void *tab_clear,
                              ha_stream_t stream) {
    return step_fwd<uint64_t>(table, rows, width, plan_cur, n_cur, grads, lr, tab_cur, plan_next, n_next, next_out,
                              tab_next, plan_fin, n_fin, tab_fin, ahead_ids, n_ahead, plan_ahead, tab_clear,
                              as_stream(stream));
}

// development aid: ha_step_f32ids with per-wave time stamps; dbg = uint64[blocks * 16 * 4]
extern "C" int ha_debug_step_fwd_timeline(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur,
                                          const float *grads, float lr, const void *tab_cur, void *plan_next,
                                          int64_t n_next, float *next_out, const void *tab_next, void *plan_fin,
                                          int64_t n_fin, void *tab_fin, const float *ahead_ids, int64_t n_ahead,
                                          void *plan_ahead, void *tab_clear, unsigned long long *dbg,
                                          ha_stream_t stream) {
    HA_REQUIRE(dbg != nullptr, "step timeline: null debug buffer");
    return step_fwd<float>(table, rows, width, plan_cur, n_cur, grads, lr, tab_cur, plan_next, n_next, next_out,
                           tab_next, plan_fin, n_fin, tab_fin, ahead_ids, n_ahead, plan_ahead, tab_clear,
                           as_stream(stream), dbg);
}
