// Fused launches of the hot path: two kernel launches per training step instead of four.
//
//   ha_lookup_sort_*      = forward gather  +  stable sort of the batch ids      (independent work)
//   ha_sgd_apply_finish   = backward fused SGD apply  +  plan finish (uniq/counts/inverse)
//
// Each launch is one grid of 1024-thread workgroups with two roles selected by blockIdx: the index
// work (rank tiles / the finish block) is VALU- and latency-bound and touches a few KiB, the row
// work (gather / apply) is HBM-bound, so they share the chip without competing for the same
// resource, and a dependent kernel boundary (~1.5 us each on MI355X) is saved per pair.  The bodies
// are the same device functions the stand-alone kernels use (plan_dev.h, gather_dev.h,
// scatter_dev.h), so results are bit-identical to the unfused entry points.
#include "plan_dev.h"
#include "gather_dev.h"
#include "scatter_dev.h"

namespace ha {

// scatter.hip
template <int MODE>
int apply_by_unique(float *dst, int64_t dst_rows, int64_t width, void *plan_ws, int64_t n,
                    const float *grads, float lr, hipStream_t stream, ApplyMaps maps = ApplyMaps{});

template <typename IdT, int NV_SHIFT>
__global__ __launch_bounds__(1024) void fwd_fused_kernel(
    const float *__restrict__ table, uint64_t rows, uint32_t nv,
    const IdT *__restrict__ ids, int n, uint64_t total_vec,
    float *__restrict__ out, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ sorted, int32_t *__restrict__ perm,
    int n_rank_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    const int b = blockIdx.x;
    if (b < n_rank_tiles)
        rank_tile_body<IdT>(ids, n, keys, sorted, perm, b, s_mem);
    else
        gather_vec4_body<IdT, 4, NV_SHIFT, 1024>(table, rows, nv, ids, total_vec,
                                                out, b - n_rank_tiles);
}

// pf_ids != nullptr: a wave that left early or applied a short run (more than half of them) ends by
// touching the table row the NEXT batch's position p will gather, so that the forward launch that
// follows finds it in the memory-side cache instead of HBM (the ids of the next batch are known one
// step ahead).  The row is loaded and dropped; waves with medium / long-run work skip this.
template <int MODE, int VEC>
__global__ __launch_bounds__(1024, 8) void bwd_fused_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    int n, const float *__restrict__ grads, float lr,
    PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq,
    int32_t *__restrict__ seg, int32_t *__restrict__ counts,
    int32_t *__restrict__ inverse, int32_t *__restrict__ upos, int n_finish_blocks,
    const float *__restrict__ pf_ids, int pf_n, int tree_from) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int b = blockIdx.x;
    if (b < n_finish_blocks) {
        finish_block_body(sorted, perm, n, hdr, uniq, seg, counts, inverse, upos, b, s_dyn);
        return;
    }
    ApplyMaps maps{};
    maps.tree_from = tree_from;
    const bool heavy = apply_body<MODE, VEC>(dst, dst_rows, width, sorted, perm, nullptr, n,
                                             grads, lr, b - n_finish_blocks, s_dyn, nullptr, maps);
    if (VEC == 4 && pf_ids != nullptr && !heavy) {
        const int p = (b - n_finish_blocks) * kPosPerBlock + static_cast<int>(threadIdx.x >> 6);
        if (p < pf_n) {
            const uint32_t key = f32_to_key(pf_ids[p]);
            if (key < dst_rows) {
                const float *row = dst + static_cast<uint64_t>(key) * static_cast<uint64_t>(width);
                float4v x = {0.f, 0.f, 0.f, 0.f};
                for (int j = lane_id() * 4; j < width; j += kWave * 4)
                    x += ld4(row + j);
                asm volatile("" ::"v"(x));   // keep the loads
            }
        }
    }
}

// Forward of a larger batch (n > kSmallMax): the first launch of the radix sort (id -> key conversion
// and the tile histograms of the first digit, 256-thread blocks) carries the gather blocks, so the
// gather runs beside the latency-bound sort instead of in front of it.
template <typename IdT, int NV_SHIFT>
__global__ __launch_bounds__(256) void fwd_large_kernel(
    const float *__restrict__ table, uint64_t rows, uint32_t nv,
    const IdT *__restrict__ ids, int n, uint64_t total_vec, float *__restrict__ out,
    uint32_t *__restrict__ keys, uint32_t *__restrict__ hist, int nblk, int tile_major, int shift, int msd) {
    __shared__ uint32_t s_h[kRadixBuckets];
    const int b = blockIdx.x;
    if (b < nblk)
        radix_first_tile_body<IdT>(ids, n, nblk, b, keys, hist, tile_major, s_h, shift, msd != 0);
    else
        gather_vec4_body<IdT, 4, NV_SHIFT, 256>(table, rows, nv, ids, total_vec, out, b - nblk);
}

static int nv_shift_of(uint32_t nv) {
    for (int s = 2; s <= 8; ++s)
        if (nv == (1u << s))
            return s;
    return -1;
}

template <typename IdT>
static int lookup_sort(const float *table, int64_t rows, int64_t width,
                       const IdT *ids, int64_t n, float *out, void *plan_ws,
                       hipStream_t stream, bool (*fallback)(const float *, int64_t, int64_t,
                                                            const IdT *, int64_t, float *, void *,
                                                            hipStream_t)) {
    HA_REQUIRE(rows >= 0 && width >= 1 && n >= 0, "lookup_sort: bad sizes");
    HA_REQUIRE(plan_ws != nullptr, "lookup_sort: null plan workspace");
    const bool vec_ok = (width % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(table) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const int shift = vec_ok ? nv_shift_of(static_cast<uint32_t>(width / 4)) : -1;
    // batches above ~12 k ids with a known key range (the table's rows): bucket sort (plan.hip) -- its first
    // launch carries the gather blocks, like the radix sort's
    const bool bucket = bucket_sort_applies(n, static_cast<uint64_t>(rows));
    const int msd_shift = bucket ? bucket_shift(static_cast<uint64_t>(rows)) : 0;
    if ((n > kSmallMax || bucket) && shift >= 0 && n < (1ll << 31)) {
        HA_REQUIRE(table && ids && out, "lookup_sort: null pointer");
        PlanPtrs p = plan_layout(plan_ws, n);
        const int ni = static_cast<int>(n), nblk = radix_tiles(n);
        const uint32_t nv = static_cast<uint32_t>(width / 4);
        const uint64_t total_vec = static_cast<uint64_t>(n) * nv;
        const unsigned gblocks = static_cast<unsigned>((total_vec + 1023) / 1024);
#define HA_FWDL_CASE(S)                                                                          \
    case S:                                                                                      \
        hipLaunchKernelGGL((fwd_large_kernel<IdT, S>), dim3(nblk + gblocks), dim3(256), 0, stream, \
                           table, (uint64_t)rows, nv, ids, ni, total_vec, out, p.keys, p.hist,   \
                           nblk, bucket ? 1 : radix_tile_major(n), msd_shift, bucket ? 1 : 0);   \
        break;
        switch (shift) {
            HA_FWDL_CASE(2) HA_FWDL_CASE(3) HA_FWDL_CASE(4) HA_FWDL_CASE(5)
            HA_FWDL_CASE(6) HA_FWDL_CASE(7) HA_FWDL_CASE(8)
        default: break;
        }
#undef HA_FWDL_CASE
        HA_LAUNCH_CHECK();
        if (bucket)
            return plan_bucket_sort(plan_ws, n, msd_shift, true, stream);
        return plan_radix_sort(plan_ws, n, 32, true, stream);
    }
    if (n == 0 || n > kSmallMax || shift < 0)
        return fallback(table, rows, width, ids, n, out, plan_ws, stream) ? 0 : -1;
    HA_REQUIRE(table && ids && out, "lookup_sort: null pointer");
    PlanPtrs p = plan_layout(plan_ws, n);
    const int ni = static_cast<int>(n);
    const uint32_t nv = static_cast<uint32_t>(width / 4);
    const uint64_t total_vec = static_cast<uint64_t>(n) * nv;
    const int tiles = (ni + kRankTile - 1) / kRankTile;
    const unsigned gblocks = static_cast<unsigned>((total_vec + 4095) / 4096);
    const dim3 grid(tiles + gblocks), block(1024);
    const size_t lds = rank_small_lds_bytes(ni);
#define HA_FWD_CASE(S)                                                          \
    case S:                                                                     \
        HA_ALLOW_LDS((fwd_fused_kernel<IdT, S>), lds);                          \
        hipLaunchKernelGGL((fwd_fused_kernel<IdT, S>), grid, block, lds, stream, \
                           table, (uint64_t)rows, nv, ids, ni, total_vec, out,  \
                           p.keys, p.sorted, p.perm, tiles);                    \
        break;
    switch (shift) {
        HA_FWD_CASE(2) HA_FWD_CASE(3) HA_FWD_CASE(4) HA_FWD_CASE(5)
        HA_FWD_CASE(6) HA_FWD_CASE(7) HA_FWD_CASE(8)
    default: break;
    }
#undef HA_FWD_CASE
    HA_LAUNCH_CHECK();
    return 0;
}

static bool fallback_f32(const float *table, int64_t rows, int64_t width,
                         const float *ids, int64_t n, float *out, void *ws,
                         hipStream_t s) {
    return ha_gather_f32ids(table, rows, width, ids, n, out, s) == 0 &&
           ha_plan_sort_f32ids(ids, n, ws, s) == 0;
}
static bool fallback_u64(const float *table, int64_t rows, int64_t width,
                         const uint64_t *ids, int64_t n, float *out, void *ws,
                         hipStream_t s) {
    return ha_gather_u64ids(table, rows, width, ids, n, out, s) == 0 &&
           ha_plan_sort_u64ids(ids, n, ws, s) == 0;
}

template <int MODE>
static int apply_finish(float *dst, int64_t rows, int64_t width, void *plan_ws,
                        int64_t n, const float *grads, float lr,
                        hipStream_t stream, const float *pf_ids = nullptr, int64_t pf_n = 0) {
    HA_REQUIRE(n >= 0 && width >= 1 && width < (1 << 30), "apply_finish: bad sizes");
    HA_REQUIRE(plan_ws != nullptr, "apply_finish: null plan workspace");
    if (n == 0)
        return ha_plan_finish(plan_ws, n, stream);
    if (n > kSmallMax) {
        // larger batches: finish the plan first, then apply by unique key (scatter.hip)
        if (ha_plan_finish(plan_ws, n, stream))
            return -1;
        return apply_by_unique<MODE>(dst, rows, width, plan_ws, n, grads, lr, stream);
    }
    HA_REQUIRE(dst && grads, "apply_finish: null pointer");
    PlanPtrs p = plan_layout(plan_ws, n);
    const int ni = static_cast<int>(n);
    const int fblocks = finish_blocks(ni);
    const unsigned blocks = static_cast<unsigned>(fblocks) + static_cast<unsigned>((n + kPosPerBlock - 1) / kPosPerBlock);
    const size_t lds = kApplyLdsBytes;
    const bool vec_ok = (width % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(dst) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(grads) % 16 == 0);
    if (vec_ok)
        hipLaunchKernelGGL((bwd_fused_kernel<MODE, 4>), dim3(blocks), dim3(1024), lds,
                           stream, dst, (uint64_t)rows, (int)width, p.sorted, p.perm,
                           ni, grads, lr, p.hdr, p.uniq, p.seg, p.counts, p.inverse, p.upos, fblocks,
                           pf_ids, (int)(pf_n < n ? pf_n : n), tolerance_tree_from());
    else
        hipLaunchKernelGGL((bwd_fused_kernel<MODE, 1>), dim3(blocks), dim3(1024), lds,
                           stream, dst, (uint64_t)rows, (int)width, p.sorted, p.perm,
                           ni, grads, lr, p.hdr, p.uniq, p.seg, p.counts, p.inverse, p.upos, fblocks,
                           pf_ids, (int)(pf_n < n ? pf_n : n), tolerance_tree_from());
    HA_LAUNCH_CHECK();
    return 0;
}

// step.hip: batches outside the single-launch regime
template <int MODE>
int apply_finish_entry(float *dst, int64_t rows, int64_t width, void *plan_ws, int64_t n,
                       const float *grads, float lr, hipStream_t stream) {
    return apply_finish<MODE>(dst, rows, width, plan_ws, n, grads, lr, stream);
}
template int apply_finish_entry<kModeSgd>(float *, int64_t, int64_t, void *, int64_t, const float *, float,
                                          hipStream_t);

}  // namespace ha

using namespace ha;

extern "C" int ha_lookup_sort_f32ids(const float *table, int64_t rows,
                                     int64_t width, const float *ids, int64_t n,
                                     float *out, void *plan_ws,
                                     ha_stream_t stream) {
    return lookup_sort<float>(table, rows, width, ids, n, out, plan_ws,
                              as_stream(stream), fallback_f32);
}

extern "C" int ha_lookup_sort_u64ids(const float *table, int64_t rows,
                                     int64_t width, const uint64_t *ids,
                                     int64_t n, float *out, void *plan_ws,
                                     ha_stream_t stream) {
    return lookup_sort<uint64_t>(table, rows, width, ids, n, out, plan_ws,
                                 as_stream(stream), fallback_u64);
}

extern "C" int ha_sgd_apply_finish(float *table, int64_t rows, int64_t width,
                                   void *plan_ws, int64_t n, const float *grads,
                                   float lr, ha_stream_t stream) {
    return apply_finish<kModeSgd>(table, rows, width, plan_ws, n, grads, lr,
                                  as_stream(stream));
}

extern "C" int ha_push_apply_finish(float *table, int64_t rows, int64_t width,
                                    void *plan_ws, int64_t n, const float *grads,
                                    ha_stream_t stream) {
    return apply_finish<kModePush>(table, rows, width, plan_ws, n, grads, 1.f,
                                   as_stream(stream));
}

extern "C" int ha_sgd_apply_finish_prefetch_f32ids(float *table, int64_t rows, int64_t width,
                                                   void *plan_ws, int64_t n, const float *grads,
                                                   float lr, const float *next_ids, int64_t next_n,
                                                   ha_stream_t stream) {
    HA_REQUIRE(next_n >= 0 && (next_n == 0 || next_ids), "apply_finish_prefetch: bad next batch");
    return apply_finish<kModeSgd>(table, rows, width, plan_ws, n, grads, lr, as_stream(stream),
                                  next_n > 0 ? next_ids : nullptr, next_n);
}
