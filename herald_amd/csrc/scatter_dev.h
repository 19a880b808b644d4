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

// Optional indirections used by the embedding cache (cache.hip): the destination row of unique key u
// is rowmap[u] (-1 = skip), the source row of occurrence i is valmap[i], and a destination row whose
// dst_init flag is 0 starts from 0.0f instead of its stored value (a cache line without a gradient
// buffer yet, Line::_maybeInitGrad, src/hetu_cache/include/embedding.h:131-134).
struct ApplyMaps {
    const int32_t *rowmap;
    const int32_t *valmap;
    const uint8_t *dst_init;
};

constexpr int kPosPerBlock = 16;   // sorted positions (= waves) per 1024-thread workgroup
constexpr int kLookBack = 16;      // positions a wave looks back to find its offset in its run
constexpr int kLongRun = 48;       // runs at least this long are split over waves of different workgroups
constexpr int kShortRun = 3;       // runs up to this long: one wave, whole row, 16-byte accesses
constexpr int kHotDepth = 16;      // split mode: loads per chunk (a 64-block is 4 chunks)

template <int MODE>
__device__ __forceinline__ float step(float acc, float g, float lr) {
    if (MODE == kModeSgd)
        return __fsub_rn(acc, __fmul_rn(lr, g));
    // push / reduce: `lr` is the scale applied to every value before it is summed (the reference
    // multiplies the whole value array by -lr first, ParameterServerCommunicate.py:58-59);
    // scale 1.0f is exact, so the unscaled ops use the same code.
    return __fadd_rn(acc, __fmul_rn(lr, g));
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

// ---- short runs (1..kShortRun occurrences): one wave, whole row --------------------------------
// Columns [cbase, cbase + VB*64*VEC); the table row and every occurrence row are requested in one
// batch (branch-free, clamped), then applied in occurrence order.
template <int MODE, int VEC, int VB>
__device__ __forceinline__ void short_block(float *__restrict__ dst_row,
                                            const float *__restrict__ grads,
                                            int width, int cbase, int pv,
                                            int lane0, int len, float lr,
                                            bool init) {
    const int lane = lane_id();
    Vec<VEC> acc[VB], g[kShortRun][VB];
    int col[VB], lcol[VB];
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        col[b] = cbase + (b * kWave + lane) * VEC;
        lcol[b] = col[b] < width ? col[b] : 0;
        acc[b].zero();
        if (MODE == kModeSgd && init)  // wave-uniform
            acc[b].load(dst_row + lcol[b]);
    }
#pragma unroll
    for (int t = 0; t < kShortRun; ++t) {
        const int idx = __builtin_amdgcn_readlane(pv, lane0 + (t < len ? t : 0));
        const float *src = grads + static_cast<size_t>(idx) * width;
#pragma unroll
        for (int b = 0; b < VB; ++b)
            g[t][b].load(src + lcol[b]);
    }
#pragma unroll
    for (int t = 0; t < kShortRun; ++t) {
        if (t < len) {  // wave-uniform
#pragma unroll
            for (int b = 0; b < VB; ++b)
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    acc[b].set(k, step<MODE>(acc[b].get(k), g[t][b].get(k), lr));
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
__device__ __forceinline__ void short_row(float *__restrict__ dst_row,
                                          const float *__restrict__ grads,
                                          int width, int pv, int lane0, int len,
                                          float lr, bool init) {
    constexpr int kCols1 = kWave * VEC;
    int c = 0;
    for (; width - c > kCols1; c += 2 * kCols1)
        short_block<MODE, VEC, 2>(dst_row, grads, width, c, pv, lane0, len, lr, init);
    for (; c < width; c += kCols1)
        short_block<MODE, VEC, 1>(dst_row, grads, width, c, pv, lane0, len, lr, init);
}

// ---- longer runs: column-split over the run's own waves ------------------------------------------
// The wave at offset o of a run owns the 64-column slice(s) o, o+W, ... (W = min(run length, 16)
// workers) and walks ALL occurrences of the run in order: one dword per lane = 256 contiguous bytes
// per occurrence row.  One wave issues roughly one instruction every four cycles, so the loop is
// written for the fewest instructions per occurrence (v_readlane of a byte offset computed 64
// occurrences at a time, v_add, global_load_dword, v_mul, v_sub), with two register half-rings of
// kHotDepth loads in flight.  The run's end is found on the fly: every block of 64 occurrence
// indices is loaded together with the 64 sorted keys of those positions, one block ahead of the
// row loads that must stay in flight (so the in-order vmcnt wait never drains the ring).
template <int MODE, bool OFF32>
__device__ __forceinline__ void split_slice(float *__restrict__ dst_row,
                                            const float *__restrict__ grads,
                                            const uint32_t *__restrict__ sorted,
                                            const int32_t *__restrict__ perm,
                                            int start, int n, uint32_t key,
                                            int width, int col, float lr,
                                            const int32_t *__restrict__ valmap,
                                            bool init) {
    const int lane = lane_id();
    const bool live = col < width;
    const int lcol = live ? col : 0;
    float acc = 0.f;
    if (MODE == kModeSgd && init)
        acc = dst_row[lcol];
    const char *gbase = reinterpret_cast<const char *>(grads);
    const uint32_t col4 = static_cast<uint32_t>(lcol) * 4u;
    const uint32_t rowbytes = static_cast<uint32_t>(width) * 4u;

    // block b = occurrences [64b, 64b+64) of the run = sorted positions start+64b+lane
    auto load_block = [&](int b, uint32_t &pv, int &cnt) {
        const int q = start + 64 * b + lane;
        const int cq = min(q, n - 1);
        const uint32_t ks = sorted[cq];
        uint32_t idx = static_cast<uint32_t>(perm[cq]);
        if (valmap)  // wave-uniform
            idx = static_cast<uint32_t>(valmap[idx]);
        pv = OFF32 ? idx * rowbytes : idx;
        const unsigned long long m = __ballot(q < n && ks == key);
        cnt = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
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
    auto consume = [&](const float(&g)[kHotDepth], int cnt) {  // cnt = valid entries (may be >= depth)
        if (cnt >= kHotDepth) {
#pragma unroll
            for (int t = 0; t < kHotDepth; ++t)
                acc = step<MODE>(acc, g[t], lr);
        } else {
#pragma unroll
            for (int t = 0; t < kHotDepth; ++t) {
                const float nx = step<MODE>(acc, g[t], lr);
                acc = (t < cnt) ? nx : acc;
            }
        }
    };

    // three register chunks rotate: while chunk c is consumed, chunks c+1 and c+2 are in flight
    // (up to 48 row loads per wave; vmcnt counts 63 at most)
    float ga[kHotDepth], gb[kHotDepth], gc[kHotDepth];
    uint32_t pv_cur, pv_nxt;
    int cnt_cur, cnt_nxt;
    load_block(0, pv_cur, cnt_cur);
    load_chunk(ga, pv_cur, 0);
    if (cnt_cur > 16)
        load_chunk(gb, pv_cur, 16);
    // invariant at loop top: chunks 0 (ga) and 1 (gb) of the current block are in flight
    for (int b = 0;; b += 3) {
        // ---- block b: ga=c0 gb=c1 ; stream c2->gc c3->ga ; next block c0->gb c1->gc
        load_block(b + 1, pv_nxt, cnt_nxt);
        if (cnt_cur < 64) cnt_nxt = 0;
        if (cnt_cur > 32) load_chunk(gc, pv_cur, 32);
        consume(ga, cnt_cur);
        if (cnt_cur > 48) load_chunk(ga, pv_cur, 48);
        if (cnt_cur > 16) consume(gb, cnt_cur - 16);
        if (cnt_nxt > 0) load_chunk(gb, pv_nxt, 0);
        if (cnt_cur > 32) consume(gc, cnt_cur - 32);
        if (cnt_nxt > 16) load_chunk(gc, pv_nxt, 16);
        if (cnt_cur > 48) consume(ga, cnt_cur - 48);
        if (cnt_nxt == 0) break;
        pv_cur = pv_nxt; cnt_cur = cnt_nxt;
        // ---- block b+1: gb=c0 gc=c1 ; c2->ga c3->gb ; next block c0->gc c1->ga
        load_block(b + 2, pv_nxt, cnt_nxt);
        if (cnt_cur < 64) cnt_nxt = 0;
        if (cnt_cur > 32) load_chunk(ga, pv_cur, 32);
        consume(gb, cnt_cur);
        if (cnt_cur > 48) load_chunk(gb, pv_cur, 48);
        if (cnt_cur > 16) consume(gc, cnt_cur - 16);
        if (cnt_nxt > 0) load_chunk(gc, pv_nxt, 0);
        if (cnt_cur > 32) consume(ga, cnt_cur - 32);
        if (cnt_nxt > 16) load_chunk(ga, pv_nxt, 16);
        if (cnt_cur > 48) consume(gb, cnt_cur - 48);
        if (cnt_nxt == 0) break;
        pv_cur = pv_nxt; cnt_cur = cnt_nxt;
        // ---- block b+2: gc=c0 ga=c1 ; c2->gb c3->gc ; next block c0->ga c1->gb
        load_block(b + 3, pv_nxt, cnt_nxt);
        if (cnt_cur < 64) cnt_nxt = 0;
        if (cnt_cur > 32) load_chunk(gb, pv_cur, 32);
        consume(gc, cnt_cur);
        if (cnt_cur > 48) load_chunk(gc, pv_cur, 48);
        if (cnt_cur > 16) consume(ga, cnt_cur - 16);
        if (cnt_nxt > 0) load_chunk(ga, pv_nxt, 0);
        if (cnt_cur > 32) consume(gb, cnt_cur - 32);
        if (cnt_nxt > 16) load_chunk(gb, pv_nxt, 16);
        if (cnt_cur > 48) consume(gc, cnt_cur - 48);
        if (cnt_nxt == 0) break;
        pv_cur = pv_nxt; cnt_cur = cnt_nxt;
    }
    if (live) {
        if (MODE == kModePush)
            acc = __fadd_rn(dst_row[col], acc);
        dst_row[col] = acc;
    }
}

// One wave per sorted position p (wave-uniform p; no workgroup-level synchronisation).
template <int MODE, int VEC>
__device__ __forceinline__ void apply_body_impl(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int p, int *dbg_info, ApplyMaps maps);

// One wave per sorted position p.  `dbg` (tools/timeline.py only) receives
// {realtime start, realtime end, role/len, shader cycles} per position.
template <int MODE, int VEC>
__device__ __forceinline__ void apply_body(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int vblock, unsigned long long *dbg = nullptr,
    ApplyMaps maps = ApplyMaps{nullptr, nullptr, nullptr}) {
    const int w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int p = vblock * kPosPerBlock + w;
    if (p >= n)
        return;
    if (dbg == nullptr) {
        apply_body_impl<MODE, VEC>(dst, dst_rows, width, sorted, perm, upos, n, grads, lr, p, nullptr, maps);
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    int info = 0;
    apply_body_impl<MODE, VEC>(dst, dst_rows, width, sorted, perm, upos, n, grads, lr, p, &info, maps);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane_id() == 0) {
        dbg[p * 4 + 0] = t0;
        dbg[p * 4 + 1] = t1;
        dbg[p * 4 + 2] = static_cast<unsigned long long>(info);
        dbg[p * 4 + 3] = c1 - c0;
    }
}

template <int MODE, int VEC>
__device__ __forceinline__ void apply_body_impl(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int p, int *dbg_info, ApplyMaps maps) {
    const int lane = lane_id();
    // window of sorted positions p-16 .. p+47 (branch-free loads)
    const int q = p - kLookBack + lane;
    const int cq = max(0, min(q, n - 1));
    const uint32_t ks = sorted[cq];
    int pv = perm[cq];
    if (maps.valmap)  // wave-uniform
        pv = maps.valmap[pv];
    const uint32_t key = static_cast<uint32_t>(
        __builtin_amdgcn_readlane(static_cast<int>(ks), kLookBack));
    const unsigned long long eq = __ballot(q >= 0 && q < n && ks == key);
    const uint32_t inv_lo = static_cast<uint32_t>(~eq) & 0xFFFFu;
    const int back = inv_lo == 0 ? kLookBack : (__builtin_clz(inv_lo) - 16);
    const unsigned long long inv_hi = (~eq) >> kLookBack;  // bit t <-> position p+t, 48 valid bits
    const int fwd = __builtin_ctzll(inv_hi | (1ull << 48));  // 1..48
    // A compute unit pulls only ~25 GB/s from HBM (66 GB/s from L2), so a long run (hundreds of
    // occurrence rows) must not be streamed by the waves of ONE workgroup.  Runs of >= kLongRun
    // occurrences are therefore column-split over waves of DIFFERENT workgroups: the designated worker
    // of slice group k is the wave at run offset 16*k (positions 16 apart sit in consecutive
    // workgroups, which the dispatcher spreads over compute units and XCDs).
    int o = back;
    bool long_worker = false;   // designated worker of a long run
    int long_workers = 0, long_k = 0;
    const int nslice = (width + kWave - 1) / kWave;
    if (back >= kLookBack) {
        // deep inside a run: exact offset up to 16*nslice by an extended look-back
        const int maxo = kLookBack * nslice;
        int cnt = kLookBack;
        for (int base = p - kLookBack; cnt < maxo; base -= kWave) {
            const int qq = base - 1 - lane;                 // positions base-1, base-2, ...
            const unsigned long long m = __ballot(qq >= 0 && sorted[max(qq, 0)] == key);
            const int c = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
            cnt += c;
            if (c < 64)
                break;
        }
        o = cnt;
        if (o >= maxo || (o % kLookBack) != 0 || o + fwd < kLongRun)
            return;  // not a designated offset, or the run is not a long one
        long_worker = true;
    } else if (o + fwd >= kLongRun) {
        // a long run (every wave of the run reaches the same verdict: o + fwd is the exact length
        // unless fwd == 48, and then it is >= 48 anyway): only offsets 0, 16, 32, ... work
        if (o != 0)
            return;
        long_worker = true;
    }
    if (long_worker) {
        // run length capped at 16*nslice decides how many designated workers exist
        const int cap = kLookBack * nslice;
        int lcap = o + fwd;
        if (fwd >= 48) {
            for (int base = p + 48; lcap < cap; base += kWave) {
                const int qq = base + lane;
                const unsigned long long m = __ballot(qq < n && sorted[min(qq, n - 1)] == key);
                const int c = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
                lcap += c;
                if (c < 64)
                    break;
            }
        }
        lcap = min(lcap, cap);
        long_workers = min(nslice, lcap / kLookBack);
        long_k = o / kLookBack;
        if (long_k >= long_workers)
            return;
    }
    const bool exact = fwd < 48 && !long_worker;
    const int len_known = o + fwd;  // exact run length when `exact`
    if (dbg_info)
        *dbg_info = (o << 16) | (len_known & 0x7FFF) | (exact ? 0 : 0x8000);

    uint64_t row;
    bool init = true;
    if (maps.rowmap) {
        const int r = maps.rowmap[upos[p]];
        if (r < 0)
            return;  // unique key without a destination
        row = static_cast<uint64_t>(r);
        if (maps.dst_init)
            init = maps.dst_init[r] != 0;
    } else if (MODE == kModeReduce) {
        row = static_cast<uint64_t>(upos[p]);
    } else {
        row = key;
    }
    if (row >= dst_rows)
        return;  // out-of-range id: ignored (undefined behaviour in the reference)
    float *dst_row = dst + row * static_cast<uint64_t>(width);

    if (exact && len_known <= kShortRun) {
        if (o == 0)
            short_row<MODE, VEC>(dst_row, grads, width, pv, kLookBack, len_known, lr, init);
        return;
    }
    // split mode.  Medium runs (4..47): the first min(L,16) waves of the run (same workgroup), worker o
    // takes slices o, o+W, ...  Long runs: designated worker k of long_workers takes slices k, k+W, ...
    const int workers = long_worker ? long_workers : min(len_known, kLookBack);
    const int me = long_worker ? long_k : o;
    if (me >= workers)
        return;
    const bool off32 = static_cast<uint64_t>(n) * static_cast<uint64_t>(width) * 4ull < (1ull << 32);
    const int start = p - o;
    for (int c0 = me * kWave; c0 < width; c0 += workers * kWave) {
        if (off32)
            split_slice<MODE, true>(dst_row, grads, sorted, perm, start, n, key, width, c0 + lane, lr, maps.valmap, init);
        else
            split_slice<MODE, false>(dst_row, grads, sorted, perm, start, n, key, width, c0 + lane, lr, maps.valmap, init);
    }
}

}  // namespace ha
