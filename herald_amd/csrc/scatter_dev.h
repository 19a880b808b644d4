// Device body of the backward apply / dedup-reduce, shared by scatter.hip and fused.hip
// (semantics and work mapping: scatter.hip).
#pragma once
#include "common.h"

namespace ha {

enum ApplyMode {
    kModeSgd = 0,     // row = row - lr*g  (two roundings per occurrence)
    kModePush = 1,    // row = row + (0 + g0 + g1 ...)   (reduce in order, then one add)
    kModeReduce = 2,  // out[u] = 0 + g0 + g1 ...
};

constexpr int kPosPerBlock = 16;   // sorted positions (= waves) per workgroup
constexpr int kHotLen = 16;        // runs at least this long take the workgroup-cooperative path
constexpr int kDepth = 4;          // cold path: occurrence rows in flight per wave

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
// row) and walks the run's occurrences in order.  One wave can issue roughly one instruction every
// four cycles, so the loop is written for the fewest instructions per occurrence:
//   v_readlane (row byte offset, computed 64 occurrences at a time by one vector multiply)
//   v_add (+ column offset), global_load_dword (SGPR base + 32-bit VGPR offset), v_mul, v_sub
// with two register half-rings of kHotDepth loads in flight.  The index vector of the NEXT 64
// occurrences is requested before the row loads that must stay in flight, so waiting for it
// (in-order vmcnt) never drains the ring; all loads are branch-free (clamped).
constexpr int kHotDepth = 16;

template <int MODE, bool OFF32>
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
    const char *gbase = reinterpret_cast<const char *>(grads);
    const uint32_t col4 = static_cast<uint32_t>(lcol) * 4u;
    const uint32_t rowbytes = static_cast<uint32_t>(width) * 4u;

    // lane l: byte offset of occurrence base+l's gradient row (OFF32) or its row index (!OFF32)
    auto load_idx = [&](int base) -> uint32_t {
        const uint32_t idx = static_cast<uint32_t>(perm_run[min(base + lane, len - 1)]);
        return OFF32 ? idx * rowbytes : idx;
    };
    auto load_chunk = [&](float(&g)[kHotDepth], uint32_t pv, int lane0) {
#pragma unroll
        for (int t = 0; t < kHotDepth; ++t) {
            const uint32_t s = static_cast<uint32_t>(
                __builtin_amdgcn_readlane(static_cast<int>(pv), lane0 + t));
            if (OFF32)
                g[t] = *reinterpret_cast<const float *>(gbase + static_cast<size_t>(s + col4));
            else
                g[t] = *reinterpret_cast<const float *>(
                    gbase + static_cast<size_t>(s) * rowbytes + col4);
        }
    };
    auto consume_full = [&](const float(&g)[kHotDepth]) {
#pragma unroll
        for (int t = 0; t < kHotDepth; ++t)
            acc = step<MODE>(acc, g[t], lr);
    };
    auto consume_tail = [&](const float(&g)[kHotDepth], int cnt) {
#pragma unroll
        for (int t = 0; t < kHotDepth; ++t) {
            const float nx = step<MODE>(acc, g[t], lr);
            acc = (t < cnt) ? nx : acc;
        }
    };
    auto consume = [&](const float(&g)[kHotDepth], int q0) {
        if (q0 + kHotDepth <= len)
            consume_full(g);
        else if (q0 < len)
            consume_tail(g, len - q0);
    };

    float ga[kHotDepth], gb[kHotDepth];
    uint32_t pv_cur = load_idx(0);
    load_chunk(ga, pv_cur, 0);
    for (int q0 = 0; q0 < len; q0 += 64) {
        const uint32_t pv_nxt = load_idx(q0 + 64);
        load_chunk(gb, pv_cur, 16);
        consume(ga, q0);
        load_chunk(ga, pv_cur, 32);
        consume(gb, q0 + 16);
        load_chunk(gb, pv_cur, 48);
        consume(ga, q0 + 32);
        load_chunk(ga, pv_nxt, 0);
        consume(gb, q0 + 48);
        pv_cur = pv_nxt;
    }
    if (live) {
        if (MODE == kModePush)
            acc = __fadd_rn(dst_row[col], acc);
        dst_row[col] = acc;
    }
}

template <int MODE, int VEC>
__device__ __forceinline__ void apply_body(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int vblock) {
    // gradient rows addressable with 32-bit byte offsets (the usual case: n*width*4 < 4 GiB)
    const bool off32 = static_cast<uint64_t>(n) * static_cast<uint64_t>(width) * 4ull < (1ull << 32);
    __shared__ int s_hot_p;
    __shared__ int s_scan[kPosPerBlock];
    const int lane = lane_id();
    const int w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int p = vblock * kPosPerBlock + w;
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
            for (int col = w * kWave + lane; col - lane < width; col += kPosPerBlock * kWave) {
                if (off32)
                    hot_slice<MODE, true>(dst + row * static_cast<uint64_t>(width), grads,
                                          perm + hp, len, width, col, lr);
                else
                    hot_slice<MODE, false>(dst + row * static_cast<uint64_t>(width), grads,
                                           perm + hp, len, width, col, lr);
            }
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
