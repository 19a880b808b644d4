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
// Work mapping.  Input is the stable sort of the batch keys (`sorted`, `perm`).  A 1024-thread
// workgroup owns 16 consecutive sorted positions, one wave each.  A wave whose position starts a run
// of equal keys ("head") owns that unique row:
//   * short run (< kHotLen occurrences, the common case): the wave alone loads the table row and all
//     occurrence rows with 16-byte loads (up to kDepth occurrence rows in flight) and applies them in
//     order;
//   * long run (hot row: a low-cardinality Criteo field repeats one id hundreds of times per batch):
//     the run is column-split over the waves of the head's workgroup, 64 columns per wave, each
//     wave streaming the occurrence rows of its slice in order with 2x32 loads in flight.  The
//     two-instruction chain per occurrence and column is the only ordered part.
// Waves at non-head positions exit (their occurrence is consumed by the head's wave/workgroup).
#include "common.h"

namespace ha {

enum ApplyMode {
    kModeSgd = 0,     // row = row - lr*g  (two roundings per occurrence)
    kModePush = 1,    // row = row + (0 + g0 + g1 ...)   (reduce in order, then one add)
    kModeReduce = 2,  // out[u] = 0 + g0 + g1 ...
};

constexpr int kPosPerBlock = 16;   // sorted positions (= waves) per workgroup
constexpr int kHotLen = 16;        // runs at least this long take the workgroup-cooperative path
constexpr int kDepth = 8;          // cold path: occurrence rows in flight per wave

template <int MODE>
__device__ __forceinline__ float step(float acc, float g, float lr) {
    if (MODE == kModeSgd)
        return __fsub_rn(acc, __fmul_rn(lr, g));
    return __fadd_rn(acc, g);
}

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    float4v v;
    __device__ __forceinline__ void load(const float *p) { v = ld4(p); }
    __device__ __forceinline__ void store(float *p) const { st4(p, v); }
    __device__ __forceinline__ void zero() { v = float4v{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ float get(int k) const { return v[k]; }
    __device__ __forceinline__ void set(int k, float x) { v[k] = x; }
};
template <>
struct Vec<1> {
    float v;
    __device__ __forceinline__ void load(const float *p) { v = *p; }
    __device__ __forceinline__ void store(float *p) const { *p = v; }
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ float get(int) const { return v; }
    __device__ __forceinline__ void set(int, float x) { v = x; }
};

// ---- cold path: one wave, whole row, columns [cbase, cbase + VB*64*VEC) per call ---------------
template <int MODE, int VEC, int VB>
__device__ __forceinline__ void cold_block(float *__restrict__ dst_row,
                                           const float *__restrict__ grads,
                                           int width, int cbase, int permv,
                                           int len, float lr) {
    const int lane = lane_id();
    Vec<VEC> acc[VB];
    int col[VB];
    int lcol[VB];  // clamped column: loads are branch-free, stores are guarded
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        col[b] = cbase + (b * kWave + lane) * VEC;
        lcol[b] = col[b] < width ? col[b] : 0;
        acc[b].zero();
        if (MODE == kModeSgd)
            acc[b].load(dst_row + lcol[b]);
    }
    for (int q0 = 0; q0 < len; q0 += kDepth) {
        Vec<VEC> g[kDepth][VB];
#pragma unroll
        for (int t = 0; t < kDepth; ++t) {
            if (q0 + t < len) {  // wave-uniform
                const int idx = __builtin_amdgcn_readlane(permv, q0 + t);
                const float *src = grads + static_cast<size_t>(idx) * width;
#pragma unroll
                for (int b = 0; b < VB; ++b)
                    g[t][b].load(src + lcol[b]);
            }
        }
#pragma unroll
        for (int t = 0; t < kDepth; ++t) {
            if (q0 + t < len) {
#pragma unroll
                for (int b = 0; b < VB; ++b)
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        acc[b].set(k, step<MODE>(acc[b].get(k), g[t][b].get(k), lr));
            }
        }
    }
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        if (col[b] < width) {
            if (MODE == kModePush) {
                Vec<VEC> cur;
                cur.load(dst_row + lcol[b]);
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    acc[b].set(k, __fadd_rn(cur.get(k), acc[b].get(k)));
            }
            acc[b].store(dst_row + col[b]);
        }
    }
}

template <int MODE, int VEC>
__device__ __forceinline__ void cold_row(float *__restrict__ dst_row,
                                         const float *__restrict__ grads,
                                         int width, int permv, int len,
                                         float lr) {
    constexpr int kCols1 = kWave * VEC;
    int c = 0;
    for (; width - c > kCols1; c += 2 * kCols1)
        cold_block<MODE, VEC, 2>(dst_row, grads, width, c, permv, len, lr);
    for (; c < width; c += kCols1)
        cold_block<MODE, VEC, 1>(dst_row, grads, width, c, permv, len, lr);
}

// ---- hot path: a long run is column-split over the waves of the head's workgroup ------------------
// Wave s owns the 64 columns [64*s, 64*s+64) (one dword per lane, 256 contiguous bytes per occurrence
// row) and walks the run's occurrences in order with kHotDepth loads in flight in each of two
// register half-rings, so the only serial part is the two-instruction chain per occurrence.  All
// loads are branch-free (occurrence index and column clamped) so that hipcc counts vmcnt across the
// ring instead of draining it.
constexpr int kHotDepth = 32;

template <int MODE>
__device__ __forceinline__ void hot_slice(float *__restrict__ dst_row,
                                          const float *__restrict__ grads,
                                          const int32_t *__restrict__ perm_run,
                                          int len, int width, int col, float lr) {
    const int lane = lane_id();
    const bool live = col < width;
    const int lcol = live ? col : 0;
    float acc = 0.f;
    if (MODE == kModeSgd)
        acc = dst_row[lcol];
    const float *gcol = grads + lcol;

    auto load_chunk = [&](float(&g)[kHotDepth], int q0) {
        // occurrence indices come through the scalar cache (wave-uniform addresses): SMEM loads
        // count on lgkmcnt, so fetching them never drains the vector-memory ring (vmcnt)
#pragma unroll
        for (int t = 0; t < kHotDepth; ++t) {
            const int idx = perm_run[uniform(min(q0 + t, len - 1))];
            g[t] = gcol[static_cast<size_t>(idx) * width];
        }
    };
    auto consume = [&](const float(&g)[kHotDepth], int q0) {
#pragma unroll
        for (int t = 0; t < kHotDepth; ++t) {
            const float nx = step<MODE>(acc, g[t], lr);
            acc = (q0 + t < len) ? nx : acc;
        }
    };

    float ga[kHotDepth], gb[kHotDepth];
    load_chunk(ga, 0);
    for (int q0 = 0; q0 < len; q0 += 2 * kHotDepth) {
        load_chunk(gb, q0 + kHotDepth);
        consume(ga, q0);
        load_chunk(ga, q0 + 2 * kHotDepth);
        consume(gb, q0 + kHotDepth);
    }
    if (live) {
        if (MODE == kModePush)
            acc = __fadd_rn(dst_row[col], acc);
        dst_row[col] = acc;
    }
}

template <int MODE, int VEC>
__global__ __launch_bounds__(1024) void apply_kernel(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr) {
    __shared__ int s_hot_p;
    __shared__ int s_scan[kPosPerBlock];
    const int lane = lane_id();
    const int w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int p = blockIdx.x * kPosPerBlock + w;
    if (threadIdx.x == 0)
        s_hot_p = -1;
    __syncthreads();

    // ---- phase A: classify my sorted position
    const bool in_range = p < n;
    const int pos = p + lane;
    const int cpos = min(pos, n - 1);
    const uint32_t ks = sorted[cpos];
    const int permv = perm[cpos];
    const uint32_t prevk = sorted[max(min(p, n - 1) - 1, 0)];
    const uint32_t key = uniform(ks);
    const unsigned long long same = __ballot(pos < n && ks == key);
    const int len64 = (~same == 0ull) ? 64 : __builtin_ctzll(~same);
    const bool head = in_range && (p == 0 || prevk != key);
    const bool hot = head && len64 >= kHotLen;
    if (hot && lane == 0)
        s_hot_p = p;
    __syncthreads();

    // ---- phase B: the (at most one) hot run whose head lies in this block
    const int hp = s_hot_p;
    if (hp >= 0) {
        const uint32_t hkey = sorted[hp];
        // run length: every wave scans 64 positions per step until a different key shows up
        int len = 0;
        for (int base = hp;; base += kPosPerBlock * kWave) {
            const int q = base + w * kWave + lane;
            const unsigned long long m = __ballot(q < n && sorted[min(q, n - 1)] == hkey);
            const int c = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
            if (lane == 0)
                s_scan[w] = c;
            __syncthreads();
            int add = 0;
            bool full = true;
            for (int k = 0; k < kPosPerBlock; ++k) {
                if (full)
                    add += s_scan[k];
                full = full && s_scan[k] == 64;
            }
            len += add;
            __syncthreads();
            if (!full)
                break;
        }
        uint64_t row;
        bool ok = true;
        if (MODE == kModeReduce) {
            row = static_cast<uint64_t>(upos[hp]);
        } else {
            row = hkey;
            ok = row < dst_rows;
        }
        if (ok) {
            for (int col = w * kWave + lane; col - lane < width; col += kPosPerBlock * kWave)
                hot_slice<MODE>(dst + row * static_cast<uint64_t>(width), grads,
                                perm + hp, len, width, col, lr);
        }
    }

    // ---- phase C: short runs, one wave each
    if (head && !hot) {
        uint64_t row;
        if (MODE == kModeReduce) {
            row = static_cast<uint64_t>(upos[p]);
        } else {
            row = key;
            if (row >= dst_rows)
                return;  // out-of-range id: ignored (undefined behaviour in the reference)
        }
        cold_row<MODE, VEC>(dst + row * static_cast<uint64_t>(width), grads,
                            width, permv, len64, lr);
    }
}

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
    const unsigned blocks =
        static_cast<unsigned>((n + kPosPerBlock - 1) / kPosPerBlock);
    const bool vec_ok = (width % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(dst) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(grads) % 16 == 0);
    if (vec_ok) {
        hipLaunchKernelGGL((apply_kernel<MODE, 4>), dim3(blocks), dim3(1024), 0,
                           stream, dst, (uint64_t)dst_rows, (int)width,
                           v.sorted, v.perm, v.upos, (int)n, grads, lr);
    } else {
        hipLaunchKernelGGL((apply_kernel<MODE, 1>), dim3(blocks), dim3(1024), 0,
                           stream, dst, (uint64_t)dst_rows, (int)width,
                           v.sorted, v.perm, v.upos, (int)n, grads, lr);
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
