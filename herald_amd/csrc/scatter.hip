// Backward of the embedding lookup: dedup-reduce and fused sparse apply, driven by an index plan.
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
// Work mapping: one wavefront per SORTED POSITION p.  u = upos[p] is its unique row, o = p - seg[u]
// its offset in the run of `len` equal keys.  A run of length 1 (the common case) is handled by its
// single wave with 16-byte loads over the whole row.  Longer runs are column-split over
// min(len, S) of their own waves so that hot rows (hundreds of occurrences in one batch) do not
// serialise on one wave; every wave keeps kDepth occurrence loads in flight.
#include "common.h"

namespace ha {

struct PlanHeader {
    int64_t n_unique;
    int64_t reserved[31];
};

enum ApplyMode {
    kModeSgd = 0,     // row = row - lr*g  (two roundings per occurrence)
    kModePush = 1,    // row = row + (0 + g0 + g1 ...)   (reduce in order, then one add)
    kModeReduce = 2,  // out[u] = 0 + g0 + g1 ...
};

constexpr int kDepth = 8;       // occurrence rows in flight per wave
constexpr int kFineRunLen = 4;  // runs at least this long use 64-column slices

template <int MODE>
__device__ __forceinline__ float step(float acc, float g, float lr) {
    if (MODE == kModeSgd)
        return __fsub_rn(acc, __fmul_rn(lr, g));
    return __fadd_rn(acc, g);
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float (&o)[VEC]) {
    if constexpr (VEC == 4) {
        const float4v v = ld4(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            o[k] = p[k];
    }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&o)[VEC]) {
    if constexpr (VEC == 4) {
        st4(p, float4v{o[0], o[1], o[2], o[3]});
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            p[k] = o[k];
    }
}

// Process columns [c0, c1) (in floats, both multiples of VEC) of unique row `u`.
template <int MODE, int VEC>
__device__ __forceinline__ void apply_slice(
    float *__restrict__ dst_row, const float *__restrict__ grads,
    const int32_t *__restrict__ occ, int len, int width, int c0, int c1,
    float lr) {
    const int lane = lane_id();
    for (int c = c0 + lane * VEC; c < c1; c += kWave * VEC) {
        float acc[VEC];
        if (MODE == kModeSgd) {
            load_vec<VEC>(dst_row + c, acc);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                acc[k] = 0.f;
        }
        for (int q0 = 0; q0 < len; q0 += kDepth) {
            float g[kDepth][VEC];
#pragma unroll
            for (int t = 0; t < kDepth; ++t) {
                if (q0 + t < len) {
                    const float *src =
                        grads + static_cast<size_t>(occ[q0 + t]) * width + c;
                    load_vec<VEC>(src, g[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < kDepth; ++t) {
                if (q0 + t < len) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        acc[k] = step<MODE>(acc[k], g[t][k], lr);
                }
            }
        }
        if (MODE == kModePush) {
            float cur[VEC];
            load_vec<VEC>(dst_row + c, cur);
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                acc[k] = __fadd_rn(cur[k], acc[k]);
        }
        store_vec<VEC>(dst_row + c, acc);
    }
}

template <int MODE, int VEC>
__global__ __launch_bounds__(256) void apply_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq,
    const int32_t *__restrict__ seg, const int32_t *__restrict__ upos,
    const int32_t *__restrict__ perm, int n, const float *__restrict__ grads,
    float lr) {
    const int p = uniform(static_cast<int>(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (p >= n)
        return;
    const int u = uniform(upos[p]);
    const int start = uniform(seg[u]);
    const int len = uniform(seg[u + 1]) - start;
    const int o = p - start;
    uint64_t row;
    if (MODE == kModeReduce) {
        row = static_cast<uint64_t>(u);
    } else {
        row = uniq[u];
        if (row >= dst_rows)
            return;  // out-of-range id: ignored (the reference has undefined behaviour here)
    }
    float *dst_row = dst + row * static_cast<uint64_t>(width);
    const int32_t *occ = perm + start;
    // Column slices: a run of `len` occurrences owns `len` waves; up to max_slices of them each take
    // a contiguous column range.  Short runs use 256-column (16 B/lane) slices, long runs switch to
    // 64-column (4 B/lane, 256 B per wave access) slices to spread the serial chain over more waves.
    if (VEC == 4 && len < kFineRunLen) {
        const int slice_cols = kWave * 4;
        const int max_slices = (width + slice_cols - 1) / slice_cols;
        const int nslice = min(len, max_slices);
        if (o >= nslice)
            return;
        const int per = (max_slices + nslice - 1) / nslice;
        const int c0 = min(width, o * per * slice_cols);
        const int c1 = min(width, (o + 1) * per * slice_cols);
        apply_slice<MODE, 4>(dst_row, grads, occ, len, width, c0, c1, lr);
    } else {
        const int slice_cols = kWave;
        const int max_slices = (width + slice_cols - 1) / slice_cols;
        const int nslice = min(len, max_slices);
        if (o >= nslice)
            return;
        const int per = (max_slices + nslice - 1) / nslice;
        const int c0 = min(width, o * per * slice_cols);
        const int c1 = min(width, (o + 1) * per * slice_cols);
        apply_slice<MODE, 1>(dst_row, grads, occ, len, width, c0, c1, lr);
    }
}

struct PlanArrays {
    const PlanHeader *hdr;
    const uint32_t *uniq;
    const int32_t *seg, *upos, *perm;
};

}  // namespace ha

// defined in plan.hip
extern "C" int ha_plan_view_of(void *ws, int64_t n, ha_plan_view *view);

namespace ha {

template <int MODE>
static int apply_launch(float *dst, int64_t dst_rows, int64_t width,
                        const void *plan_ws, int64_t n, const float *grads,
                        float lr, hipStream_t stream) {
    HA_REQUIRE(n >= 0 && width >= 1 && width < (1 << 30), "apply: bad sizes");
    if (n == 0)
        return 0;
    HA_REQUIRE(dst && plan_ws && grads, "apply: null pointer");
    ha_plan_view v;
    if (ha_plan_view_of(const_cast<void *>(plan_ws), n, &v) != 0)
        return -1;
    const PlanHeader *hdr = reinterpret_cast<const PlanHeader *>(v.n_unique);
    const unsigned blocks = static_cast<unsigned>((n + 3) / 4);
    const bool vec_ok = (width % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(dst) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(grads) % 16 == 0);
    if (vec_ok) {
        hipLaunchKernelGGL((apply_kernel<MODE, 4>), dim3(blocks), dim3(256), 0,
                           stream, dst, (uint64_t)dst_rows, (int)width, hdr,
                           v.uniq, v.seg, v.upos, v.perm, (int)n, grads, lr);
    } else {
        hipLaunchKernelGGL((apply_kernel<MODE, 1>), dim3(blocks), dim3(256), 0,
                           stream, dst, (uint64_t)dst_rows, (int)width, hdr,
                           v.uniq, v.seg, v.upos, v.perm, (int)n, grads, lr);
    }
    HA_LAUNCH_CHECK();
    return 0;
}

}  // namespace ha

extern "C" int ha_sgd_apply(float *table, int64_t rows, int64_t width,
                            const void *plan_ws, int64_t n, const float *grads,
                            float lr, ha_stream_t stream) {
    return ha::apply_launch<ha::kModeSgd>(table, rows, width, plan_ws, n, grads,
                                          lr, ha::as_stream(stream));
}

extern "C" int ha_push_apply(float *table, int64_t rows, int64_t width,
                             const void *plan_ws, int64_t n, const float *grads,
                             ha_stream_t stream) {
    return ha::apply_launch<ha::kModePush>(table, rows, width, plan_ws, n,
                                           grads, 0.f, ha::as_stream(stream));
}

extern "C" int ha_dedup_reduce(const void *plan_ws, int64_t n,
                               const float *grads, int64_t width,
                               float *reduced, ha_stream_t stream) {
    return ha::apply_launch<ha::kModeReduce>(reduced, n, width, plan_ws, n,
                                             grads, 0.f, ha::as_stream(stream));
}
