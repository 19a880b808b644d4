// Backward of the embedding lookup: dedup-reduce and fused sparse apply, driven by the sorted plan.
//
// Reference semantics (all fp32, all deterministic here; no atomics):
//   ha_sgd_apply    : cpu_SGDOptimizerSparseUpdate, src/dnnl_ops/Optimizers.cpp:51-74 -- serial loop
//                     over occurrences, `param[id,j] -= lr * g[i,j]` (separate multiply and subtract
//                     roundings: the reference is built with -O3 for baseline x86-64, no FMA).
//                     Rows are independent, so applying each row's occurrences in occurrence order
//                     reproduces the serial loop bit for bit.
//   ha_dedup_reduce : IndexedSlices.cpu_deduplicate, python/hetu/ndarray.py:556-576 --
//                     `new[inv[i]] += g[i]` for i ascending, from 0.0f.  Same order as
//                     PSAgent::vecPushSparse (PSAgent.h:124-183) and Line::accumulate (embedding.h:78-91).
//   ha_push_apply   : server `+=` of PSHandler::serve(SparsePush), PSFHandle.h:130-164, on the
//                     worker-reduced rows.
// The CUDA reference does this with one atomicAdd per element (src/ops/OptimizersSparse.cu:53-99,
// 282-295); here every unique row is read once, updated in registers and written once:
// algorithmic bytes per batch = n*(4*width + 4) + U*8*width.
//
// Work mapping.  Input is the stable sort of the batch keys (`sorted`, `perm`).  One wavefront per
// SORTED POSITION p, no workgroup-level synchronisation.  A wave reads the window of sorted keys
// p-16..p+47 and finds its offset o in its run of equal keys:
//   * run of 1..3 occurrences (the common case): the wave at o == 0 loads the table row and all
//     occurrence rows with 16-byte loads in one batch and applies them in order;
//   * longer run (a low-cardinality Criteo field repeats one id hundreds of times per batch): the
//     first min(L,16) waves of the run each take 64-column slices and stream ALL occurrences of
//     the run for their slice in order, 2x16 loads in flight (split_slice in scatter_dev.h).  The
//     two-instruction chain per occurrence and column is the only ordered part;
//   * every other wave of a run exits.
#include "scatter_dev.h"
#include "plan_dev.h"
static_assert(ha::kPlanLongRun == ha::kLongRun, "the finish lists exactly the runs the cooperative path takes");

namespace ha {

template <int MODE, int VEC>
__global__ __launch_bounds__(1024, 8) void apply_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int tree_from) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    ApplyMaps maps{};
    maps.tree_from = tree_from;
    apply_body<MODE, VEC>(dst, dst_rows, width, sorted, perm, upos, n, grads, lr, blockIdx.x, s_apply, nullptr, maps);
}

// cache flavour: destination / source rows through index maps (see ApplyMaps)
template <int VEC>
__global__ __launch_bounds__(1024, 8) void apply_mapped_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ src,
    float lr, ApplyMaps maps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    apply_body<kModeSgd, VEC>(dst, dst_rows, width, sorted, perm, upos, n, src, lr, blockIdx.x, s_apply, nullptr, maps);
}

// two destinations, one pass (the cache's Line::accumulate: gradient buffer and data row)
template <int VEC>
__global__ __launch_bounds__(1024, 8) void apply_mapped2_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ src,
    float lr, ApplyMaps maps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    apply_body<kModeSgd, VEC, true>(dst, dst_rows, width, sorted, perm, upos, n, src, lr, blockIdx.x, s_apply,
                                    nullptr, maps);
}

// ---- larger batches (n > kSmallMax) with a FINISHED plan: waves map to unique keys ------------------
// One wave per sorted position costs ~5 ns of dispatch per workgroup and most positions of a large
// batch sit inside long runs, so here a fixed grid of workgroups loops over the unique keys instead
// (uniq / seg / counts of the plan): short and medium runs are applied by the wave that owns the key,
// keys with >= kLongRun occurrences are collected in a list ...
template <int MODE, int VEC>
__global__ __launch_bounds__(1024, 8) void apply_unique_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width, PlanHeader *__restrict__ hdr,
    const uint32_t *__restrict__ uniq, const int32_t *__restrict__ seg,
    const int32_t *__restrict__ counts, const int32_t *__restrict__ perm, int n,
    const float *__restrict__ grads, float lr, uint32_t *__restrict__ long_list, ApplyMaps maps) {
    const int U = static_cast<int>(hdr->n_unique);
    const int lane = lane_id();
    const int nwaves = gridDim.x * 16;
    for (int u = blockIdx.x * 16 + static_cast<int>(threadIdx.x >> 6); u < U; u += nwaves) {
        const int s = uniform(seg[u]), len = uniform(counts[u]);
        const uint32_t key = uniform(uniq[u]);
        if (len >= kLongRun) {
            if (lane == 0) {
                const unsigned long long i =
                    atomicAdd(reinterpret_cast<unsigned long long *>(&hdr->reserved[0]), 1ull);
                long_list[i] = static_cast<uint32_t>(u);
            }
            continue;
        }
        const uint64_t row = MODE == kModeReduce ? static_cast<uint64_t>(u) : static_cast<uint64_t>(key);
        if (row >= dst_rows)
            continue;  // out-of-range id: ignored
        float *dst_row = dst + row * static_cast<uint64_t>(width);
        const int pv = perm[min(s + lane, n - 1)];   // lanes 0 .. len-1: the run's occurrence indices
        Second d2{nullptr, false};
        if (MODE == kModeOpt)
            opt_rows(d2, maps, row, width);
        if (len <= kShortRun) {
            short_row<MODE, VEC, false>(dst_row, grads, width, pv, 0, len, lr, true, d2);
        } else {
            if (VEC == 4 && MODE != kModeOpt) {
                for (int c0 = 0; c0 < width; c0 += 2 * kWave)
                    medium_pair<MODE>(dst_row, grads, width, c0 + 2 * lane, pv, len, lr, true);
            } else {
                for (int c0 = 0; c0 < width; c0 += kWave)
                    medium_slice<MODE, false>(dst_row, grads, width, c0 + lane, pv, 0, len, lr, true, d2);
            }
        }
    }
}

// ... and served here: work item = (long key, 64-column slice); the 16 waves of a workgroup load the
// occurrence rows of the slice block by block into LDS and one wave runs the ordered chain
// (coop_slices, scatter_dev.h).
template <int MODE>
__global__ __launch_bounds__(1024, 8) void apply_long_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width, const PlanHeader *__restrict__ hdr,
    const uint32_t *__restrict__ uniq, const int32_t *__restrict__ seg,
    const int32_t *__restrict__ counts, const int32_t *__restrict__ perm, int n,
    const float *__restrict__ grads, float lr, const uint32_t *__restrict__ long_list, ApplyMaps maps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    const int nslice = (width + kWave - 1) / kWave;
    const long long items = hdr->reserved[0] * nslice;
    const int w = static_cast<int>(threadIdx.x >> 6);
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int u = static_cast<int>(long_list[it / nslice]);
        const int j = static_cast<int>(it % nslice);
        const uint64_t row = MODE == kModeReduce ? static_cast<uint64_t>(u) : static_cast<uint64_t>(uniq[u]);
        if (row < dst_rows) {
            Second d2{nullptr, false};
            if (MODE == kModeOpt)
                opt_rows(d2, maps, row, width);
            coop_slices<MODE, false>(dst + row * static_cast<uint64_t>(width), true, d2, grads, perm, maps, n, lr,
                                     seg[u], counts[u], width, j, nslice, w, reinterpret_cast<float *>(s_apply));
        }
        __syncthreads();
    }
}

// Both in ONE launch when the finish has already listed the long keys (batches up to kFinishChunkedMax ids):
// a workgroup first serves its share of the (long key, slice) items -- they are the launch's critical path, a
// 2,000-occurrence run is an ordered chain of 2,000 steps whatever else happens -- and then joins the loop over
// the unique keys.  One launch and one dependent boundary less, and the short keys no longer wait for the list.
//
// TOLERANCE MODE, runs beyond 256 occurrences (BASELINE configs[2]'s per-GPU batch has a key with 4,096): one workgroup per
// (key, slice) walks such a run block by block -- 16 dependent rounds of index load + row loads, ~39 us for a launch whose
// bytes take 10.  A one-workgroup launch in front (apply_chunk_plan_kernel) cuts every such run into chunks of 256: the
// item list becomes (key, chunk, slice), every workgroup reads the chunk offsets of the long keys into LDS and finds its
// items there, chunk sums meet as described at coop_chunk_tree.  OPT-IN (ha_set_tolerance_mode(2)): measured at that shape,
// ha_push_apply_scaled_finished takes 35.8 us with the chunks and 34.5 without (tools/push_chunk_ab.py) -- the long run is not
// what the launch waits for; the chunks stay for batches where it is.  Conditions (all functions of the batch alone, so that the
// oracle can restate them: oracle/qstep_model.py `chunked=`): tolerance mode on, at most kChunkKeysMax listed keys, width <= 256
// (the chunk sums live in the 2 n words of the plan's `dep` array), no index maps / second destination.
constexpr int kChunkKeysMax = 2040;
constexpr size_t kListedLdsBytes = kApplyLdsBytes + (kChunkKeysMax + 8) * 4;
struct ChunkPlan {
    uint32_t *meta;     // [0] total chunks (0: unchunked), [1] listed keys; [8 ..] exclusive chunk offsets of the listed keys, [L] = total
    uint32_t *ctr;      // [L * nslice] arrivals per (key, slice)
    float *part;        // [total * nslice * 64] chunk sums
};
__global__ __launch_bounds__(1024) void apply_chunk_plan_kernel(const PlanHeader *__restrict__ hdr,
                                                                const uint32_t *__restrict__ long_list,
                                                                const int32_t *__restrict__ counts, int nslice, int tree_from,
                                                                long long part_cap_floats, ChunkPlan cp) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const long long L = hdr->reserved[0];
    if (tree_from <= 0 || L <= 0 || L > kChunkKeysMax) {
        if (threadIdx.x == 0)
            cp.meta[0] = 0u;
        return;
    }
    if (threadIdx.x == 0)
        s_carry = 0u;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int base = 0; base < L; base += 1024) {
        const int i = base + threadIdx.x;
        uint32_t nc = 0;
        if (i < L) {
            const int len = counts[long_list[i]];
            nc = len >= tree_from && len > kTreeChunk ? static_cast<uint32_t>((len + kTreeChunk - 1) / kTreeChunk) : 1u;
        }
        uint32_t incl = nc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += y;
        }
        if (lane == 63)
            s_w[w] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int k = 0; k < w; ++k)
            before += s_w[k];
        if (i < L)
            cp.meta[8 + i] = before + incl - nc;
        __syncthreads();
        if (threadIdx.x == 1023)
            s_carry = before + incl;
        __syncthreads();
    }
    const uint32_t total = s_carry;
    for (long long k = threadIdx.x; k < L * nslice; k += 1024)
        cp.ctr[k] = 0u;
    if (threadIdx.x == 0) {
        cp.meta[8 + L] = total;
        cp.meta[1] = static_cast<uint32_t>(L);
        // (nothing to chunk, or the chunk sums would not fit: the launch runs unchunked)
        cp.meta[0] = total > L && static_cast<long long>(total) * nslice * 64 <= part_cap_floats ? total : 0u;
    }
}

template <int MODE, int VEC>
__global__ __launch_bounds__(1024, 8) void apply_listed_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width, const PlanHeader *__restrict__ hdr,
    const uint32_t *__restrict__ uniq, const int32_t *__restrict__ seg,
    const int32_t *__restrict__ counts, const int32_t *__restrict__ perm, int n,
    const float *__restrict__ grads, float lr, const uint32_t *__restrict__ long_list, ApplyMaps maps, ChunkPlan cp) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    const int nslice = (width + kWave - 1) / kWave;
    const int w = static_cast<int>(threadIdx.x >> 6);
    const long long L = hdr->reserved[0];
    const uint32_t total = cp.meta != nullptr ? uniform(cp.meta[0]) : 0u;      // chunks over all listed keys (0: unchunked)
    uint32_t *s_off = s_apply + kApplyLdsBytes / 4;
    if (total != 0u) {
        for (int i = threadIdx.x; i <= L; i += 1024)
            s_off[i] = cp.meta[8 + i];
        __syncthreads();
    }
    // Two roles.  A listed item is five dependent trips (header, list, record, occurrence indices, rows) and two barriers, a
    // wave's keys three trips + one per key; a workgroup that did both in turn took the sum of the two (33.4-36.6 us at
    // configs[2]'s per-GPU shape: 280 listed keys x 2 slices, 23 k keys).  Now the FIRST workgroups -- one per listed item, at
    // most half the grid; dispatched first: the long poles -- take the listed items and nothing else, the others the keys, a wave
    // each.  A listed workgroup fetches the records of all its items in one go and asks for the occurrence indices of the next
    // item before the rows of the current one.  28.6-31 us in every split tried (1 / 2 / 3 items per listed workgroup, grids of
    // 512 / 768 / 1,024: docs/EXPERIMENTS.md round 6 section 15) -- what is left is 77 MB as 512-byte reads at random places of
    // a gradient buffer that comes from DRAM.  Which workgroup does what is a function of the plan's header alone.
    const int G = static_cast<int>(gridDim.x);
    const long long items = (total != 0u ? static_cast<long long>(total) : L) * nslice;
    const int NL = static_cast<int>(min(items, static_cast<long long>(G / 2)));
    // (as many workgroups of keys as the chip holds at once, whatever NL is: with few listed items a second round of
    // workgroups would pay the three leading trips twice)
    const int NK = min(G - NL, G / 2);
    const bool keys_role = static_cast<int>(blockIdx.x) >= NL;
    if (static_cast<int>(blockIdx.x) >= NL + NK)
        return;
    if (!keys_role && total != 0u) {
        // runs beyond kTreeChunk occurrences in chunks (ha_set_tolerance_mode(2)): item = (key, chunk, slice)
        for (long long it = blockIdx.x; it < items; it += NL) {
            const int j = static_cast<int>(it % nslice);
            const uint32_t q = static_cast<uint32_t>(it / nslice);
            int lo = 0, hi = static_cast<int>(L);      // the listed key whose chunk range holds chunk q (s_off[L] = total)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_off[mid] <= q)
                    lo = mid;
                else
                    hi = mid;
            }
            const int li = lo, ch = static_cast<int>(q - s_off[lo]), nch = static_cast<int>(s_off[lo + 1] - s_off[lo]);
            const int u = static_cast<int>(long_list[li]);
            const uint64_t row = MODE == kModeReduce ? static_cast<uint64_t>(u) : static_cast<uint64_t>(uniq[u]);
            if (nch > 1) {       // (never with index maps or a second destination: the host leaves cp.meta null then)
                if (row < dst_rows)
                    coop_chunk_tree<MODE>(dst + row * static_cast<uint64_t>(width), true, grads, perm, maps, n, lr, seg[u],
                                          counts[u], width, j, w, reinterpret_cast<float *>(s_apply),
                                          cp.part + (static_cast<size_t>(s_off[li]) * nslice + static_cast<size_t>(j) * nch) * 64,
                                          cp.ctr + static_cast<size_t>(li) * nslice + j, ch, nch);
            } else if (row < dst_rows) {
                Second d2{nullptr, false};
                coop_slices<MODE, false>(dst + row * static_cast<uint64_t>(width), true, d2, grads, perm, maps, n, lr, seg[u],
                                         counts[u], width, j, nslice, w, reinterpret_cast<float *>(s_apply));
            }
            __syncthreads();
        }
        return;
    }
    if (!keys_role) {
        constexpr int kRecs = 32;      // records fetched at once (a workgroup has about three items)
        __shared__ int s_rs[kRecs], s_rl[kRecs], s_ri[kRecs];
        __shared__ unsigned long long s_rr[kRecs];
        const int lane = lane_id();
        for (long long r0 = blockIdx.x; r0 < items; r0 += static_cast<long long>(kRecs) * NL) {
            __syncthreads();       // (the records of the round before have been read)
            if (threadIdx.x < kRecs) {
                const long long it = r0 + static_cast<long long>(threadIdx.x) * NL;
                int rs = 0, rl = 0, ri = 1;
                unsigned long long rr = ~0ull;
                if (it < items) {
                    const int u = static_cast<int>(long_list[it / nslice]);
                    rs = seg[u];
                    rl = counts[u];
                    rr = MODE == kModeReduce ? static_cast<unsigned long long>(u) : static_cast<unsigned long long>(uniq[u]);
                    if (maps.rowmap) {      // destination rows through an index map (ha_apply_mapped): -1 = no destination
                        const int r = maps.rowmap[u];
                        rr = r < 0 ? ~0ull : static_cast<unsigned long long>(r);
                        if (r >= 0 && maps.dst_init)
                            ri = maps.dst_init[r] != 0;
                    }
                }
                s_rs[threadIdx.x] = rs;
                s_rl[threadIdx.x] = rl;
                s_ri[threadIdx.x] = ri;
                s_rr[threadIdx.x] = rr;
            }
            __syncthreads();
            const int cnt = static_cast<int>(min(static_cast<long long>(kRecs), (items - r0 + NL - 1) / NL));
            // this wave's occurrence indices of the first block of 256 of item k (what coop_tree_sum asks for first)
            auto first_indices = [&](int k) {
                const int ks = s_rs[k], kl = s_rl[k];
                return 16 * w < kl ? perm[min(ks + min(16 * w + (lane & 15), kl - 1), n - 1)] : 0;
            };
            int p_next = first_indices(0);
            for (int k = 0; k < cnt; ++k) {
                const int j = static_cast<int>((r0 + static_cast<long long>(k) * NL) % nslice);
                const int ks = s_rs[k], kl = s_rl[k];
                const uint64_t row = s_rr[k];
                const bool init = s_ri[k] != 0;
                const int p_cur = p_next;
                if (k + 1 < cnt)
                    p_next = first_indices(k + 1);
                if (row >= dst_rows)
                    continue;      // (uniform over the workgroup)
                float *dst_row = dst + row * static_cast<uint64_t>(width);
                if (MODE != kModeOpt && maps.tree_from > 0 && kl >= maps.tree_from && (width & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(dst_row) | reinterpret_cast<uintptr_t>(grads)) & 15) == 0) {
                    // (coop_slices' own condition for the tree; ends with a barrier)
                    coop_slice_tree<MODE>(dst_row, init, grads, perm, maps, n, lr, ks, kl, width, j, w,
                                          reinterpret_cast<float *>(s_apply), p_cur, true);
                } else {
                    Second d2{nullptr, false};
                    if (MODE == kModeOpt)
                        opt_rows(d2, maps, row, width);
                    coop_slices<MODE, false>(dst_row, init, d2, grads, perm, maps, n, lr, ks, kl, width, j, nslice, w,
                                             reinterpret_cast<float *>(s_apply));
                    __syncthreads();
                }
            }
        }
        return;
    }
    // The keys below kLongRun, one wave per key, the waves striding over them.  A key costs three dependent trips (its record,
    // its occurrence indices, the rows); a wave has several keys (22 k keys over 8,192 waves at configs[2]'s per-GPU shape), so
    // the records of ALL the wave's keys of a round come in one trip -- lane k holds those of its k-th key -- and the
    // occurrence indices of the next key are asked for before the rows of the current one: from the second key on a key costs
    // the rows' trip alone.  The arithmetic per key is what it was.
    const int U = static_cast<int>(hdr->n_unique);
    const int lane = lane_id();
    const int nwaves = NK * 16;
    for (long long ub = (static_cast<int>(blockIdx.x) - NL) * 16 + w; ub < U; ub += static_cast<long long>(nwaves) * kWave) {
        const long long uk = ub + static_cast<long long>(lane) * nwaves;
        int m_s = 0, m_len = kLongRun, m_r = -1, m_init = 1;
        uint32_t m_key = 0u;
        if (uk < U) {
            m_s = seg[uk];
            m_len = counts[uk];
            m_key = uniq[uk];
            if (maps.rowmap) {
                m_r = maps.rowmap[uk];
                if (m_r >= 0 && maps.dst_init)
                    m_init = maps.dst_init[m_r] != 0;
            }
        }
        const int cnt = static_cast<int>(min(static_cast<long long>(kWave), (U - ub + nwaves - 1) / nwaves));
        int pv_next = perm[min(__builtin_amdgcn_readlane(m_s, 0) + lane, n - 1)];
        for (int k = 0; k < cnt; ++k) {
            const int u = static_cast<int>(ub) + k * nwaves;
            const int len = __builtin_amdgcn_readlane(m_len, k);
            const int pv = pv_next;   // lanes 0 .. len-1: the run's occurrence indices
            if (k + 1 < cnt)
                pv_next = perm[min(__builtin_amdgcn_readlane(m_s, k + 1) + lane, n - 1)];
            if (len >= kLongRun)
                continue;
            const uint32_t key = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m_key), k));
            uint64_t row = MODE == kModeReduce ? static_cast<uint64_t>(u) : static_cast<uint64_t>(key);
            bool init = true;
            if (maps.rowmap) {
                const int r = __builtin_amdgcn_readlane(m_r, k);
                row = r < 0 ? ~0ull : static_cast<uint64_t>(r);
                if (r >= 0 && maps.dst_init)
                    init = __builtin_amdgcn_readlane(m_init, k) != 0;
            }
            if (row >= dst_rows)
                continue;  // out-of-range id / no destination: ignored
            float *dst_row = dst + row * static_cast<uint64_t>(width);
            Second d2{nullptr, false};
            if (MODE == kModeOpt)
                opt_rows(d2, maps, row, width);
            if (len <= kShortRun) {
                short_row<MODE, VEC, false>(dst_row, grads, width, pv, 0, len, lr, init, d2);
            } else {
                if (VEC == 4 && MODE != kModeOpt) {
                    for (int c0 = 0; c0 < width; c0 += 2 * kWave)
                        medium_pair<MODE>(dst_row, grads, width, c0 + 2 * lane, pv, len, lr, init);
                } else {
                    for (int c0 = 0; c0 < width; c0 += kWave)
                        medium_slice<MODE, false>(dst_row, grads, width, c0 + lane, pv, 0, len, lr, init, d2);
                }
            }
        }
    }
}

// diagnostic twin of apply_kernel<kModeSgd,4>: same body plus per-wave time stamps
__global__ __launch_bounds__(1024, 8) void apply_timeline_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    int n, const float *__restrict__ grads, float lr, unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    apply_body<kModeSgd, 4>(dst, dst_rows, width, sorted, perm, nullptr, n, grads, lr, blockIdx.x, s_apply, dbg);
}

}  // namespace ha

// ---- tolerance mode -------------------------------------------------------------------------------------------------
// BASELINE.json's north star asks for 1e-5 relative on accumulated fp32 gradients, not for the reference's serial
// order.  Off (default): every run is the ordered chain of cpu_SGDOptimizerSparseUpdate (Optimizers.cpp:65-72), bit for
// bit.  On: runs of 64 or more occurrences of a key are applied as `row - tree_sum(lr * g)` in a FIXED order
// (coop_slice_tree, scatter_dev.h; deterministic; restated by oracle/qstep_model.py) -- a 2,250-occurrence run of a
// 106,496-id batch stops being a chain of 2,250 dependent subtractions.  Process-wide; read at launch time by
// ha_sgd_apply*, ha_push_apply*, ha_dedup_reduce*, ha_apply_mapped, ha_shard_serve_push and what builds on them.
namespace ha {
static int g_tree_from = 0;
static int g_tree_chunks = 0;      // mode 2: the applies of a finished plan cut runs beyond 256 occurrences into chunks
int tolerance_tree_from() { return g_tree_from; }
}  // namespace ha

extern "C" int ha_set_tolerance_mode(int on) {
    HA_REQUIRE(on >= 0 && on <= 2, "ha_set_tolerance_mode: 0 (off), 1 (trees) or 2 (trees, long runs in chunks)");
    ha::g_tree_from = on ? 64 : 0;
    ha::g_tree_chunks = on == 2 ? 1 : 0;
    return 0;
}
extern "C" int ha_get_tolerance_mode(void) {
    return ha::g_tree_from == 0 ? 0 : 1 + ha::g_tree_chunks;
}

// defined in plan.hip
extern "C" int ha_plan_view_of(void *ws, int64_t n, ha_plan_view *view);

namespace ha {

template <int MODE>
int apply_by_unique(float *dst, int64_t dst_rows, int64_t width, void *plan_ws, int64_t n,
                    const float *grads, float lr, hipStream_t stream, ApplyMaps maps = ApplyMaps{});

template <int MODE>
static int apply_launch(float *dst, int64_t dst_rows, int64_t width,
                        const void *plan_ws, int64_t n, const float *grads,
                        float lr, hipStream_t stream) {
    HA_REQUIRE(n >= 0 && width >= 1 && width < (1 << 30), "apply: bad sizes");
    if (n == 0)
        return 0;
    HA_REQUIRE(dst && plan_ws && grads, "apply: null pointer");
    if (MODE == kModeReduce && n > kSmallMax)   // dedup-reduce reads a FINISHED plan: map waves to unique keys
        return apply_by_unique<MODE>(dst, dst_rows, width, const_cast<void *>(plan_ws), n, grads, lr, stream);
    ha_plan_view v;
    if (ha_plan_view_of(const_cast<void *>(plan_ws), n, &v) != 0)
        return -1;
    const unsigned blocks =
        static_cast<unsigned>((n + kPosPerBlock - 1) / kPosPerBlock);
    const bool vec_ok = (width % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(dst) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(grads) % 16 == 0);
    if (vec_ok) {
        hipLaunchKernelGGL((apply_kernel<MODE, 4>), dim3(blocks), dim3(1024), kApplyLdsBytes,
                           stream, dst, (uint64_t)dst_rows, (int)width,
                           v.sorted, v.perm, v.upos, (int)n, grads, lr, tolerance_tree_from());
    } else {
        hipLaunchKernelGGL((apply_kernel<MODE, 1>), dim3(blocks), dim3(1024), kApplyLdsBytes,
                           stream, dst, (uint64_t)dst_rows, (int)width,
                           v.sorted, v.perm, v.upos, (int)n, grads, lr, tolerance_tree_from());
    }
    HA_LAUNCH_CHECK();
    return 0;
}

}  // namespace ha

// apply of a FINISHED plan by unique key (larger batches); plan scratch keys_alt holds the long list
namespace ha {
template <int MODE>
int apply_by_unique(float *dst, int64_t dst_rows, int64_t width, void *plan_ws, int64_t n,
                    const float *grads, float lr, hipStream_t stream, ApplyMaps maps) {
    HA_REQUIRE(dst && plan_ws && grads && n > 0 && width >= 1 && width < (1 << 30), "apply_by_unique: bad arguments");
    if (maps.tree_from == 0)
        maps.tree_from = tolerance_tree_from();
    PlanPtrs p = plan_layout(plan_ws, n);
    const bool vec_ok = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(dst) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(grads) % 16 == 0);
    const dim3 grid(512), block(1024);   // two workgroups per compute unit, looping over the keys
    if (n > kSmallMax && n <= kFinishChunkedMax) {
        // tolerance mode: runs beyond 256 occurrences in chunks (see apply_listed_kernel); scratch = radix scratch of the plan
        // (perm_alt: offsets and counters) and its `dep` words (chunk sums)
        ChunkPlan cp{nullptr, nullptr, nullptr};
        const int nslice = static_cast<int>((width + kWave - 1) / kWave);
        const bool plain = !maps.rowmap && !maps.valmap && !maps.dst_init && MODE != kModeOpt;
        if (g_tree_chunks && maps.tree_from > 0 && plain && vec_ok && nslice <= 4) {
            static DeviceOnce lds_allowed;
            if (lds_allowed.run([]() -> int {
                    HA_ALLOW_LDS((apply_listed_kernel<MODE, 4>), kListedLdsBytes);
                    return 0;
                }))
                return -1;
            cp.meta = reinterpret_cast<uint32_t *>(p.perm_alt);
            cp.ctr = cp.meta + kChunkKeysMax + 16;
            cp.part = reinterpret_cast<float *>(p.dep);
            hipLaunchKernelGGL(apply_chunk_plan_kernel, dim3(1), dim3(1024), 0, stream, p.hdr, p.keys_alt, p.counts, nslice,
                               maps.tree_from, static_cast<long long>(2 * n), cp);
        }
        // the chunked finish (plan.hip) left the list of long keys in keys_alt / header word 0
        // (grids of 448 / 384 / 256 workgroups, to leave slots to a sort running beside it on another stream:
        // 57.8 / 57.2 / 59.0 us alone against 52.5 us, and no faster together -- tools/cfgc_bench.py)
        const dim3 grid2(1024);      // (apply_listed_kernel: up to 512 workgroups of listed items in front of 512 of keys)
        if (vec_ok)
            hipLaunchKernelGGL((apply_listed_kernel<MODE, 4>), grid2, block, cp.meta ? kListedLdsBytes : kApplyLdsBytes, stream,
                               dst, (uint64_t)dst_rows, (int)width, p.hdr, p.uniq, p.seg, p.counts, p.perm, (int)n, grads,
                               lr, p.keys_alt, maps, cp);
        else
            hipLaunchKernelGGL((apply_listed_kernel<MODE, 1>), grid2, block, kApplyLdsBytes, stream, dst,
                               (uint64_t)dst_rows, (int)width, p.hdr, p.uniq, p.seg, p.counts, p.perm, (int)n, grads,
                               lr, p.keys_alt, maps, cp);
        HA_LAUNCH_CHECK();
        return 0;
    }
    HA_CHECK_HIP(hipMemsetAsync(&p.hdr->reserved[0], 0, sizeof(int64_t), stream));
    if (vec_ok)
        hipLaunchKernelGGL((apply_unique_kernel<MODE, 4>), grid, block, 0, stream, dst, (uint64_t)dst_rows,
                           (int)width, p.hdr, p.uniq, p.seg, p.counts, p.perm, (int)n, grads, lr, p.keys_alt, maps);
    else
        hipLaunchKernelGGL((apply_unique_kernel<MODE, 1>), grid, block, 0, stream, dst, (uint64_t)dst_rows,
                           (int)width, p.hdr, p.uniq, p.seg, p.counts, p.perm, (int)n, grads, lr, p.keys_alt, maps);
    hipLaunchKernelGGL((apply_long_kernel<MODE>), grid, block, kApplyLdsBytes, stream, dst, (uint64_t)dst_rows,
                       (int)width, p.hdr, p.uniq, p.seg, p.counts, p.perm, (int)n, grads, lr, p.keys_alt, maps);
    HA_LAUNCH_CHECK();
    return 0;
}
template int apply_by_unique<kModeSgd>(float *, int64_t, int64_t, void *, int64_t, const float *, float, hipStream_t, ApplyMaps);
template int apply_by_unique<kModePush>(float *, int64_t, int64_t, void *, int64_t, const float *, float, hipStream_t, ApplyMaps);
template int apply_by_unique<kModeReduce>(float *, int64_t, int64_t, void *, int64_t, const float *, float, hipStream_t, ApplyMaps);
template int apply_by_unique<kModeOpt>(float *, int64_t, int64_t, void *, int64_t, const float *, float, hipStream_t, ApplyMaps);
}  // namespace ha

// ---- fused deduplicate + optimizer step ---------------------------------------------------------------
// The reference deduplicates the sparse gradient (host np.unique + DeduplicateIndexedSlices, python/hetu/
// ndarray.py:532-554: the reduced rows are written to HBM) and then runs the optimizer kernel on the reduced
// slices (python/hetu/gpu_links/OptimizerLink.py:52-100).  Here one launch does both: the apply's
// occurrence-ordered sum of a key's gradient rows stays in registers and is consumed by the optimizer step of
// that row (kModeOpt, scatter_dev.h) -- same sums, same expressions, so the result equals
// ha_dedup_reduce + {AdaGrad,Adam,AdamW}OptimizerSparseUpdate bit for bit.
namespace ha {
template <int VEC>
__global__ __launch_bounds__(1024, 4) void apply_opt_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width, const uint32_t *__restrict__ sorted,
    const int32_t *__restrict__ perm, int n, const float *__restrict__ grads, ApplyMaps maps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_apply[];
    apply_body<kModeOpt, VEC>(dst, dst_rows, width, sorted, perm, nullptr, n, grads, 1.f, blockIdx.x, s_apply,
                              nullptr, maps);
}
}  // namespace ha

extern "C" int ha_sparse_opt_fused_f32ids(int kind, float *param, int64_t rows, int64_t width, const float *ids,
                                          int64_t n, const float *grads, float *state1, float *state2,
                                          const float *hyper_host, void *plan_ws, ha_stream_t stream) {
    using namespace ha;
    HA_REQUIRE(kind == kAdaGrad || kind == kAdam || kind == kAdamW, "sparse_opt_fused: kind must be 0 (AdaGrad), "
               "1 (Adam) or 2 (AdamW)");
    HA_REQUIRE(n >= 0 && rows >= 0 && width >= 1 && width < (1 << 30), "sparse_opt_fused: bad sizes");
    if (n == 0)
        return 0;
    HA_REQUIRE(param && ids && grads && state1 && (kind == kAdaGrad || state2) && hyper_host && plan_ws,
               "sparse_opt_fused: null pointer");
    hipStream_t s = as_stream(stream);
    if (ha_plan_sort_f32ids_lim(ids, n, plan_ws, static_cast<uint64_t>(rows), stream))
        return -1;
    ApplyMaps maps{};
    maps.opt_s1 = state1;
    maps.opt_s2 = kind == kAdaGrad ? nullptr : state2;
    maps.opt_kind = kind;
    maps.oa = OptArgs{};
    maps.oa.lr = hyper_host[0];
    maps.oa.eps = hyper_host[1];
    maps.oa.beta1 = hyper_host[2];
    maps.oa.beta2 = hyper_host[3];
    maps.oa.beta1t = hyper_host[4];
    maps.oa.beta2t = hyper_host[5];
    maps.oa.weight_decay = hyper_host[6];
    if (n > kSmallMax) {
        if (ha_plan_finish(plan_ws, n, stream))
            return -1;
        return apply_by_unique<kModeOpt>(param, rows, width, plan_ws, n, grads, 1.f, s, maps);
    }
    PlanPtrs p = plan_layout(plan_ws, n);
    const unsigned blocks = static_cast<unsigned>((n + kPosPerBlock - 1) / kPosPerBlock);
    const bool vec_ok = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(param) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(grads) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(state1) % 16 == 0) &&
                        (state2 == nullptr || reinterpret_cast<uintptr_t>(state2) % 16 == 0);
    if (vec_ok) {
        HA_ALLOW_LDS((apply_opt_kernel<4>), kApplyLdsBytes);
        hipLaunchKernelGGL((apply_opt_kernel<4>), dim3(blocks), dim3(1024), kApplyLdsBytes, s, param, (uint64_t)rows,
                           (int)width, p.sorted, p.perm, (int)n, grads, maps);
    } else {
        HA_ALLOW_LDS((apply_opt_kernel<1>), kApplyLdsBytes);
        hipLaunchKernelGGL((apply_opt_kernel<1>), dim3(blocks), dim3(1024), kApplyLdsBytes, s, param, (uint64_t)rows,
                           (int)width, p.sorted, p.perm, (int)n, grads, maps);
    }
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_sgd_apply(float *table, int64_t rows, int64_t width,
                            const void *plan_ws, int64_t n, const float *grads,
                            float lr, ha_stream_t stream) {
    return ha::apply_launch<ha::kModeSgd>(table, rows, width, plan_ws, n, grads,
                                          lr, ha::as_stream(stream));
}

// ha_sgd_apply for a FINISHED plan (ha_plan_build_*, or ha_plan_sort_* + ha_plan_finish): batches of more than 36,864 ids
// map their waves to unique keys (the finish listed the long keys), smaller ones run as ha_sgd_apply.  Same results.
extern "C" int ha_sgd_apply_finished(float *table, int64_t rows, int64_t width, void *plan_ws, int64_t n,
                                     const float *grads, float lr, ha_stream_t stream) {
    if (n > ha::kSmallMax && n <= ha::kFinishChunkedMax) {
        HA_REQUIRE(table && plan_ws && grads && width >= 1 && width < (1 << 30), "sgd_apply_finished: bad arguments");
        return ha::apply_by_unique<ha::kModeSgd>(table, rows, width, plan_ws, n, grads, lr, ha::as_stream(stream),
                                                 ha::ApplyMaps{nullptr, nullptr, nullptr, nullptr, nullptr});
    }
    return ha::apply_launch<ha::kModeSgd>(table, rows, width, plan_ws, n, grads, lr, ha::as_stream(stream));
}

extern "C" int ha_push_apply(float *table, int64_t rows, int64_t width,
                             const void *plan_ws, int64_t n, const float *grads,
                             ha_stream_t stream) {
    return ha::apply_launch<ha::kModePush>(table, rows, width, plan_ws, n,
                                           grads, 1.f, ha::as_stream(stream));
}

// ha_push_apply with every value multiplied by `scale` before it is summed, for a FINISHED plan: table[key,:] +=
// (0 + scale*g_i0) + scale*g_i1 ... -- worker-side `values *= -lr` + occurrence-ordered reduce + server `+=` of a PS sparse
// push in ONE launch, bit-identical to ha_dedup_reduce_scaled followed by the server add (ParameterServerCommunicate.py:58-59,
// PSAgent.h:146-160, PSFHandle.h:130-164).  What a rank does for the keys of a batch that it owns itself when nobody else
// pushes (herald_amd/sharded.py at world size 1).  Batches beyond 36,864 ids map their waves to unique keys.
extern "C" int ha_push_apply_scaled_finished(float *table, int64_t rows, int64_t width, void *plan_ws, int64_t n,
                                             const float *grads, float scale, ha_stream_t stream) {
    if (n > ha::kSmallMax && n <= ha::kFinishChunkedMax) {
        HA_REQUIRE(table && plan_ws && grads && width >= 1 && width < (1 << 30), "push_apply_scaled_finished: bad arguments");
        return ha::apply_by_unique<ha::kModePush>(table, rows, width, plan_ws, n, grads, scale, ha::as_stream(stream),
                                                  ha::ApplyMaps{nullptr, nullptr, nullptr, nullptr, nullptr});
    }
    return ha::apply_launch<ha::kModePush>(table, rows, width, plan_ws, n, grads, scale, ha::as_stream(stream));
}

extern "C" int ha_dedup_reduce(const void *plan_ws, int64_t n,
                               const float *grads, int64_t width,
                               float *reduced, ha_stream_t stream) {
    return ha::apply_launch<ha::kModeReduce>(reduced, n, width, plan_ws, n,
                                             grads, 1.f, ha::as_stream(stream));
}

extern "C" int ha_dedup_reduce_scaled(const void *plan_ws, int64_t n,
                                      const float *grads, int64_t width,
                                      float scale, float *reduced,
                                      ha_stream_t stream) {
    return ha::apply_launch<ha::kModeReduce>(reduced, n, width, plan_ws, n,
                                             grads, scale, ha::as_stream(stream));
}

// Development aid (tools/timeline.py): SGD apply with per-wave {start, end, role, cycles} stamps
// written to dbg[4*n] (s_memrealtime ticks are 10 ns).  Not part of the product path.
extern "C" int ha_debug_apply_timeline(float *table, int64_t rows, int64_t width,
                                       const void *plan_ws, int64_t n,
                                       const float *grads, float lr,
                                       unsigned long long *dbg, ha_stream_t stream) {
    HA_REQUIRE(table && plan_ws && grads && dbg && n > 0 && width % 4 == 0, "timeline: bad arguments");
    ha_plan_view v;
    if (ha_plan_view_of(const_cast<void *>(plan_ws), n, &v) != 0)
        return -1;
    const unsigned blocks = static_cast<unsigned>((n + ha::kPosPerBlock - 1) / ha::kPosPerBlock);
    hipLaunchKernelGGL(ha::apply_timeline_kernel, dim3(blocks), dim3(1024), ha::kApplyLdsBytes, ha::as_stream(stream),
                       table, (uint64_t)rows, (int)width, v.sorted, v.perm, (int)n, grads, lr, dbg);
    HA_LAUNCH_CHECK();
    return 0;
}

// dst[rowmap[u],:] = (init ? dst[rowmap[u],:] : 0) - lr*src[valmap[i0],:] - lr*src[valmap[i1],:] ...
// over the occurrences i0 < i1 < ... of unique key u of a FINISHED plan.  rowmap / valmap / dst_init
// may each be NULL (identity / identity / always init).  Used by the embedding cache.
namespace ha {
static bool mapped_by_unique() {      // HA_MAPPED_BY_UNIQUE=0: ha_apply_mapped always one wave per sorted position
    static const bool on = [] {
        const char *e = getenv("HA_MAPPED_BY_UNIQUE");
        return !(e && atoi(e) == 0);
    }();
    return on;
}
}  // namespace ha

extern "C" int ha_apply_mapped(float *dst, int64_t dst_rows, int64_t width,
                               const void *plan_ws, int64_t n, const float *src,
                               float lr, const int32_t *rowmap,
                               const int32_t *valmap, const uint8_t *dst_init,
                               ha_stream_t stream) {
    HA_REQUIRE(n >= 0 && width >= 1 && width < (1 << 30), "apply_mapped: bad sizes");
    if (n == 0)
        return 0;
    HA_REQUIRE(dst && plan_ws && src, "apply_mapped: null pointer");
    ha_plan_view v;
    if (ha_plan_view_of(const_cast<void *>(plan_ws), n, &v) != 0)
        return -1;
    const unsigned blocks = static_cast<unsigned>((n + ha::kPosPerBlock - 1) / ha::kPosPerBlock);
    const bool vec_ok = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(dst) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(src) % 16 == 0);
    ha::ApplyMaps maps{rowmap, valmap, dst_init, nullptr, nullptr};
    maps.tree_from = ha::tolerance_tree_from();
    // Larger batches whose chunked finish listed the long keys: waves map to UNIQUE keys (a wave per sorted position
    // spends most of a 106,496-id batch of 128-wide rows on positions that own no row: the framed push's reduce 45.7 us)
    if (valmap == nullptr && rowmap != nullptr && n > ha::kSmallMax && n <= ha::kFinishChunkedMax &&
        ha::mapped_by_unique())
        return ha::apply_by_unique<ha::kModeSgd>(dst, dst_rows, width, const_cast<void *>(plan_ws), n, src, lr,
                                                 ha::as_stream(stream), maps);
    if (vec_ok)
        hipLaunchKernelGGL((ha::apply_mapped_kernel<4>), dim3(blocks), dim3(1024), ha::kApplyLdsBytes, ha::as_stream(stream),
                           dst, (uint64_t)dst_rows, (int)width, v.sorted, v.perm, v.upos, (int)n, src, lr, maps);
    else
        hipLaunchKernelGGL((ha::apply_mapped_kernel<1>), dim3(blocks), dim3(1024), ha::kApplyLdsBytes, ha::as_stream(stream),
                           dst, (uint64_t)dst_rows, (int)width, v.sorted, v.perm, v.upos, (int)n, src, lr, maps);
    HA_LAUNCH_CHECK();
    return 0;
}

// ha_apply_mapped into TWO destinations with one pass over `src`: for every unique key u
//   dst [rowmap[u],:]  = (dst_init[rowmap[u]] ? dst[rowmap[u],:] : 0) - lr*src[i0,:] - lr*src[i1,:] ...
//   dst2[rowmap2[u],:] =  dst2[rowmap2[u],:]                          - lr*src[i0,:] - lr*src[i1,:] ...
// (rowmap2[u] < 0: no second row for u).  Each result is bit-identical to its own ha_apply_mapped call.
extern "C" int ha_apply_mapped2(float *dst, int64_t dst_rows, float *dst2, int64_t width,
                                const void *plan_ws, int64_t n, const float *src, float lr,
                                const int32_t *rowmap, const int32_t *rowmap2,
                                const uint8_t *dst_init, ha_stream_t stream) {
    HA_REQUIRE(n >= 0 && width >= 1 && width < (1 << 30), "apply_mapped2: bad sizes");
    if (n == 0)
        return 0;
    HA_REQUIRE(dst && dst2 && plan_ws && src && rowmap && rowmap2, "apply_mapped2: null pointer");
    ha_plan_view v;
    if (ha_plan_view_of(const_cast<void *>(plan_ws), n, &v) != 0)
        return -1;
    const unsigned blocks = static_cast<unsigned>((n + ha::kPosPerBlock - 1) / ha::kPosPerBlock);
    const bool vec_ok = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(dst) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(dst2) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(src) % 16 == 0);
    const ha::ApplyMaps maps{rowmap, nullptr, dst_init, dst2, rowmap2};
    if (vec_ok)
        hipLaunchKernelGGL((ha::apply_mapped2_kernel<4>), dim3(blocks), dim3(1024), ha::kApplyLdsBytes,
                           ha::as_stream(stream), dst, (uint64_t)dst_rows, (int)width, v.sorted, v.perm, v.upos,
                           (int)n, src, lr, maps);
    else
        hipLaunchKernelGGL((ha::apply_mapped2_kernel<1>), dim3(blocks), dim3(1024), ha::kApplyLdsBytes,
                           ha::as_stream(stream), dst, (uint64_t)dst_rows, (int)width, v.sorted, v.perm, v.upos,
                           (int)n, src, lr, maps);
    HA_LAUNCH_CHECK();
    return 0;
}
