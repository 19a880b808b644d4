// Device body of the backward apply / dedup-reduce, shared by scatter.hip and fused.hip
// (semantics and work mapping: scatter.hip).
#pragma once
#include "common.h"
#include "optim_dev.h"

namespace ha {

enum ApplyMode {
    kModeSgd = 0,     // row = row - lr*g  (two roundings per occurrence)
    kModePush = 1,    // row = row + (0 + g0 + g1 ...)   (reduce in order, then one add)
    kModeReduce = 2,  // out[u] = 0 + g0 + g1 ...
    kModeOpt = 3,     // g = 0 + g0 + g1 ... (the dedup-reduce), then one AdaGrad / Adam / AdamW step of the row
                      // with it: the reduced gradient never leaves the registers (fused deduplicate + optimizer)
};

// Optional indirections used by the embedding cache (cache.hip): the destination row of unique key u
// is rowmap[u] (-1 = skip), the source row of occurrence i is valmap[i], and a destination row whose
// dst_init flag is 0 starts from 0.0f instead of its stored value (a cache line without a gradient
// buffer yet, Line::_maybeInitGrad, src/hetu_cache/include/embedding.h:131-134).
struct ApplyMaps {
    const int32_t *rowmap;
    const int32_t *valmap;
    const uint8_t *dst_init;
    // optional SECOND destination that receives the same ordered chain (DUAL kernels only): the cache
    // accumulates a batch into a line's gradient buffer and into its data row in one pass
    // (Line::accumulate, src/hetu_cache/include/embedding.h:78-91).  rowmap2[u] < 0 = no second row.
    float *dst2;
    const int32_t *rowmap2;
    // kModeOpt: optimizer state arrays (row-major like dst), kind and hyper-parameters
    float *opt_s1, *opt_s2;
    int opt_kind;
    OptArgs oa;
    // tolerance mode (ha_set_tolerance_mode): runs of at least this many occurrences (0 = never) are applied as
    // `row - tree_sum(lr * g)` in a fixed order instead of the serial chain (coop_slice_tree below)
    int tree_from;
    // DUAL == 2 kernels, the cache's planned update (cache_block.hip): a key whose record carries kPosPush is PUSHED in the
    // same pass -- push_tab[key,:] += the first destination's new value (the line's gradient after the batch,
    // PSFhandle_embedding.cc:23-27), and the first destination is stored as zeros (Line::zeroGrad, embedding.h:112-118)
    float *push_tab;
    uint64_t push_rows;      // rows of push_tab (a record whose key lies beyond is not pushed: defensive, see cache_block.hip)
    // DUAL == 2: one record per SORTED POSITION {destination row (both destinations; -1 = skip), key, kPos* flags, occurrence
    // index}: what the rowmap / init / push lookups through upos[p] give, in the round trip that fetches the window itself
    const int4 *pos_item;
    // DUAL == 2, the LFU policies' planned update: a record with kPosTemp has NO second destination (a line that is not in the
    // cache when the update runs: CacheBase::_embeddingUpdate's `new Line` without data, cache.cc:147-152); a record with
    // kPosVictimPush is the key of the line the batch's own lookup evicted with updates pending (at most one per batch): its
    // push adds that line's old gradient row -- dst row *victim_row -- BEHIND its own (the evicted line follows the batch's
    // line of the same key in should_push, cache.cc:160-170).  kPosVictim / kPosVictimHg: for the lookup (cache_block.hip).
    const int *victim_row;
};
enum : int { kPosMiss = 1, kPosInit = 2, kPosPush = 4, kPosHead = 8, kPosTemp = 16, kPosVictim = 32, kPosVictimHg = 64,
             kPosVictimPush = 128 };

// host: the run length from which the tolerance mode applies (0 = exact everywhere; scatter.hip,
// ha_set_tolerance_mode)
int tolerance_tree_from();

// second destination row of the current key (on == false: none)
struct Second {
    float *row;
    bool on;
    // kModeOpt: the state rows of the current key
    float *s1, *s2;
    int opt_kind;
    OptArgs oa;
    float *push;    // DUAL: the store row that takes the first destination's new value (nullptr: none), see ApplyMaps::push_tab
    const float *push2;   // DUAL == 2: a second row added to the store row behind it (nullptr: none), see ApplyMaps::victim_row
};


// kModeOpt epilogue: p / s1 / s2 at `col` of the current key's rows take one optimizer step with gradient g
template <int VEC>
__device__ __forceinline__ void opt_epilogue(float *__restrict__ dst_row, const Second &d2, int col, const float *g) {
    float p[VEC], x1[VEC], x2[VEC];
    const bool two = d2.opt_kind != kAdaGrad;
    if (VEC == 4) {
        *reinterpret_cast<float4v *>(p) = ld4(dst_row + col);
        *reinterpret_cast<float4v *>(x1) = ld4(d2.s1 + col);
        if (two)
            *reinterpret_cast<float4v *>(x2) = ld4(d2.s2 + col);
    } else {
        p[0] = dst_row[col];
        x1[0] = d2.s1[col];
        if (two)
            x2[0] = d2.s2[col];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        if (!two)
            x2[k] = 0.f;
        opt_step_rt(d2.opt_kind, p[k], g[k], x1[k], x2[k], d2.oa);
    }
    if (VEC == 4) {
        st4(dst_row + col, *reinterpret_cast<float4v *>(p));
        st4(d2.s1 + col, *reinterpret_cast<float4v *>(x1));
        if (two)
            st4(d2.s2 + col, *reinterpret_cast<float4v *>(x2));
    } else {
        dst_row[col] = p[0];
        d2.s1[col] = x1[0];
        if (two)
            d2.s2[col] = x2[0];
    }
}

// kModeOpt: the state rows of destination row `row`
__device__ __forceinline__ void opt_rows(Second &d2, const ApplyMaps &maps, uint64_t row, int width) {
    d2.s1 = maps.opt_s1 + row * static_cast<uint64_t>(width);
    d2.s2 = maps.opt_s2 ? maps.opt_s2 + row * static_cast<uint64_t>(width) : nullptr;
    d2.opt_kind = maps.opt_kind;
    d2.oa = maps.oa;
}

constexpr int kPosPerBlock = 16;   // sorted positions (= waves) per 1024-thread workgroup
constexpr int kLookBack = 16;      // positions a wave looks back to find its offset in its run
constexpr int kLongRun = 48;       // runs at least this long are handled by their full workgroups
constexpr int kShortRun = 3;       // runs up to this long: one wave, whole row, 16-byte accesses
constexpr int kChunk = 16;         // occurrence rows requested per batch (one register each)

// ---- in-launch hand-off of updated rows (step.hip: apply(k) runs beside the lookup of batch k+1) ----
// Every occurrence of a key registered `nslice` units (64-column slices of the row) in the pending table
// of its batch; a wave that has written s slices of a row with L occurrences gives L*s units back, after
// its device-coherent stores have been acknowledged.  A reader may load the row once the word is 0.
__device__ __forceinline__ void signal_done(uint32_t *pend_word, int units) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane_id() == 0)
        __hip_atomic_fetch_add(pend_word, static_cast<uint32_t>(-units), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// ---- forwarding of updated rows (step.hip, ha_step_*: two batches of lookahead) ------------------------
// The next batch was sorted by an EARLIER launch, so its sorted keys, occurrence indices and key table
// (common.h, StepTab) are complete when this launch starts: the wave that holds the final values of a row
// (slice) in registers also writes them to every output row of the next batch that names the key -- the
// lookup of that batch then only copies the rows this batch does not touch, nothing waits inside the
// launch and the updated rows are not read back from HBM.
enum Handoff { kHandNone = 0, kHandSignal = 1, kHandForward = 2 };
struct Hand {
    uint32_t *pend;            // kHandSignal: pending table of this batch
    const uint4 *tab;          // kHandForward: key table of the NEXT batch (nullptr: no next batch)
    const int32_t *nperm;      //   its occurrence indices in sorted order (= output rows)
    int n_next;
    float *out;                //   its output rows [n_next, width]
};
// destinations of the current key in the next batch (wave-uniform except dv)
struct Fwd {
    int start, m;   // sorted positions [start, start + m) of the next batch hold the key; m == 0: none
    int dv;         // lane l: output row of position start + l (l < min(m, 64))
};
__device__ __forceinline__ uint4 fwd_probe(const Hand &hd, uint32_t key) {   // the first probe, issued early
    return hd.tab[tab_slot(key)];
}
__device__ __forceinline__ Fwd fwd_resolve(const Hand &hd, uint32_t key, uint4 e) {
    Fwd f{0, 0, 0};
    uint32_t sl = tab_slot(key);
    while (e.x != key) {          // wave-uniform
        if (e.x == kTabEmpty)
            return f;
        sl = (sl + 1) & kTabMask;
        e = hd.tab[sl];
    }
    f.start = static_cast<int>(e.y);
    f.m = static_cast<int>(e.z + 1u);
    f.dv = hd.nperm[min(f.start + lane_id(), hd.n_next - 1)];
    return f;
}
// occurrence indices of destinations [j0, j0 + 64) of the key (the first 64 come with fwd_resolve)
__device__ __forceinline__ int fwd_dests(const Hand &hd, const Fwd &f, int j0) {
    return j0 == 0 ? f.dv : hd.nperm[min(f.start + j0 + lane_id(), hd.n_next - 1)];
}

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
    // updated rows are written around the L2 (non-temporal): nothing re-reads them inside the launch and
    // dirty lines left in the L2 lengthen the boundary to the next kernel
    __device__ __forceinline__ void store(float *p) const { st4_nt(p, v); }
    // device-coherent write-through store (a row another workgroup of the SAME launch reads, step.hip)
    __device__ __forceinline__ void store_sc1(float *p) const { st4_sc1(p, v); }
    __device__ __forceinline__ void zero() { v = float4v{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ float get(int k) const { return v[k]; }
    __device__ __forceinline__ void set(int k, float x) { v[k] = x; }
};
template <>
struct Vec<1> {
    float v;
    __device__ __forceinline__ void load(const float *p) { v = *p; }
    __device__ __forceinline__ void store(float *p) const { __builtin_nontemporal_store(v, p); }
    __device__ __forceinline__ void store_sc1(float *p) const { st1_sc1(p, v); }
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ float get(int) const { return v; }
    __device__ __forceinline__ void set(int, float x) { v = x; }
};

template <int VEC>
__device__ __forceinline__ void push_epilogue(float *__restrict__ push_row, int col, Vec<VEC> &acc, const float *push2 = nullptr) {
    Vec<VEC> cur;
    cur.load(push_row + col);
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        cur.set(k, __fadd_rn(cur.get(k), acc.get(k)));
    if (push2) {         // (wave-uniform, rare)
        Vec<VEC> old;
        old.load(push2 + col);
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            cur.set(k, __fadd_rn(cur.get(k), old.get(k)));
    }
    cur.store(push_row + col);
    acc.zero();
}

// ---- short runs (1..kShortRun occurrences): one wave, whole row --------------------------------
// Columns [cbase, cbase + VB*64*VEC); the table row and every occurrence row are requested in one
// batch (branch-free, clamped), then applied in occurrence order.
template <int MODE, int VEC, int VB, int DUAL, int HAND = kHandNone>
__device__ __forceinline__ void short_block(float *__restrict__ dst_row,
                                            const float *__restrict__ grads,
                                            int width, int cbase, int pv,
                                            int lane0, int len, float lr,
                                            bool init, Second d2, const Hand &hd = Hand{},
                                            uint32_t key = 0, uint4 pe = uint4{0, 0, 0, 0}, Fwd *fw = nullptr,
                                            bool resolve = false) {
    const int lane = lane_id();
    Vec<VEC> acc[VB], g[kShortRun][VB], acc2[DUAL ? VB : 1];
    Vec<VEC> cur[MODE == kModePush ? VB : 1];   // push: the destination's old value, requested in front of the occurrence rows
    int col[VB], lcol[VB];
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        col[b] = cbase + (b * kWave + lane) * VEC;
        lcol[b] = col[b] < width ? col[b] : 0;
        acc[b].zero();
        if (MODE == kModeSgd && init)  // wave-uniform
            acc[b].load(dst_row + lcol[b]);
        if (MODE == kModePush)
            cur[b].load(dst_row + lcol[b]);
        if (DUAL && d2.on)
            acc2[b].load(d2.row + lcol[b]);
    }
#pragma unroll
    for (int t = 0; t < kShortRun; ++t) {
        const int idx = __builtin_amdgcn_readlane(pv, lane0 + (t < len ? t : 0));
        const float *src = grads + static_cast<size_t>(idx) * width;
#pragma unroll
        for (int b = 0; b < VB; ++b)
            g[t][b].load(src + lcol[b]);
    }
    // the destinations in the next batch: probe issued before the row requests, resolved behind them
    if (HAND == kHandForward && resolve)
        *fw = fwd_resolve(hd, key, pe);
#pragma unroll
    for (int t = 0; t < kShortRun; ++t) {
        if (t < len) {  // wave-uniform
#pragma unroll
            for (int b = 0; b < VB; ++b)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    acc[b].set(k, step<MODE>(acc[b].get(k), g[t][b].get(k), lr));
                    if (DUAL && d2.on)
                        acc2[b].set(k, step<MODE>(acc2[b].get(k), g[t][b].get(k), lr));
                }
        }
    }
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        if (col[b] < width) {
            if (MODE == kModePush) {
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    acc[b].set(k, __fadd_rn(cur[b].get(k), acc[b].get(k)));
            }
            if (DUAL == 2 && d2.push)
                push_epilogue<VEC>(d2.push, col[b], acc[b], d2.push2);
            if (MODE == kModeOpt) {
                float gsum[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    gsum[k] = acc[b].get(k);
                opt_epilogue<VEC>(dst_row, d2, col[b], gsum);
            } else if (HAND == kHandSignal)
                acc[b].store_sc1(dst_row + col[b]);
            else
                acc[b].store(dst_row + col[b]);
            if (DUAL && d2.on)
                acc2[b].store(d2.row + col[b]);
        }
    }
    if (HAND == kHandForward) {
        for (int j0 = 0; j0 < fw->m; j0 += kWave) {
            const int dv = fwd_dests(hd, *fw, j0);
            const int cnt = min(kWave, fw->m - j0);
            for (int j = 0; j < cnt; ++j) {
                float *o = hd.out + static_cast<size_t>(__builtin_amdgcn_readlane(dv, j)) * width;
#pragma unroll
                for (int b = 0; b < VB; ++b)
                    if (col[b] < width)
                        acc[b].store(o + col[b]);
            }
        }
    }
}

template <int MODE, int VEC, int DUAL, int HAND = kHandNone>
__device__ __forceinline__ void short_row(float *__restrict__ dst_row,
                                          const float *__restrict__ grads,
                                          int width, int pv, int lane0, int len,
                                          float lr, bool init, Second d2, const Hand &hd = Hand{},
                                          uint32_t key = 0) {
    constexpr int kCols1 = kWave * VEC;
    Fwd fw{0, 0, 0};
    uint4 pe{0, 0, 0, 0};
    bool resolve = HAND == kHandForward && hd.tab != nullptr;
    if (resolve)
        pe = fwd_probe(hd, key);
    int c = 0;
    for (; width - c > kCols1; c += 2 * kCols1) {
        short_block<MODE, VEC, 2, DUAL, HAND>(dst_row, grads, width, c, pv, lane0, len, lr, init, d2, hd, key, pe, &fw, resolve);
        resolve = false;
    }
    for (; c < width; c += kCols1) {
        short_block<MODE, VEC, 1, DUAL, HAND>(dst_row, grads, width, c, pv, lane0, len, lr, init, d2, hd, key, pe, &fw, resolve);
        resolve = false;
    }
}

// ---- medium runs (kShortRun < L < kLongRun): column-split over the run's own first waves --------
// The wave at offset o of the run owns the 64-column slice(s) o, o+W, ... (W = min(L, 16) workers)
// and applies ALL occurrences of the run to its slice in order: one dword per lane = 256 contiguous
// bytes per occurrence row.  Every occurrence index of such a run is already in the wave's window of
// sorted positions (p-16 .. p+47), so the row loads are issued straight away, up to 32 in flight.
template <int MODE, int DUAL, int HAND = kHandNone>
__device__ __forceinline__ void medium_slice(float *__restrict__ dst_row,
                                             const float *__restrict__ grads,
                                             int width, int col, int pv, int lane_s,
                                             int len, float lr, bool init, Second d2,
                                             const Hand &hd = Hand{}, uint32_t key = 0,
                                             uint4 pe = uint4{0, 0, 0, 0}, Fwd *fw = nullptr, bool resolve = false) {
    const bool live = col < width;
    const int lcol = live ? col : 0;
    float acc = 0.f, acc2 = 0.f, old = 0.f;
    if (MODE == kModeSgd && init)
        acc = dst_row[lcol];
    if (MODE == kModePush)      // (requested in front of the occurrence rows, not behind the last of them)
        old = dst_row[lcol];
    if (DUAL && d2.on)
        acc2 = d2.row[lcol];
    auto load_chunk = [&](float(&g)[kChunk], int t0) {
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            const int idx = __builtin_amdgcn_readlane(pv, lane_s + min(t0 + t, len - 1));
            g[t] = (grads + static_cast<size_t>(idx) * width)[lcol];
        }
    };
    auto consume = [&](const float(&g)[kChunk], int cnt) {  // cnt = valid entries (may exceed kChunk)
        if (cnt >= kChunk) {
#pragma unroll
            for (int t = 0; t < kChunk; ++t) {
                acc = step<MODE>(acc, g[t], lr);
                if (DUAL)
                    acc2 = step<MODE>(acc2, g[t], lr);
            }
        } else {
#pragma unroll
            for (int t = 0; t < kChunk; ++t) {
                const float nx = step<MODE>(acc, g[t], lr);
                acc = (t < cnt) ? nx : acc;
                if (DUAL) {
                    const float nx2 = step<MODE>(acc2, g[t], lr);
                    acc2 = (t < cnt) ? nx2 : acc2;
                }
            }
        }
    };
    float ga[kChunk], gb[kChunk];
    load_chunk(ga, 0);
    if (len > kChunk)
        load_chunk(gb, kChunk);
    if (HAND == kHandForward && resolve)
        *fw = fwd_resolve(hd, key, pe);
    consume(ga, len);
    if (len > 2 * kChunk)
        load_chunk(ga, 2 * kChunk);
    if (len > kChunk)
        consume(gb, len - kChunk);
    if (len > 2 * kChunk)
        consume(ga, len - 2 * kChunk);
    if (live) {
        if (MODE == kModePush)
            acc = __fadd_rn(old, acc);
        if (DUAL == 2 && d2.push) {
            float pr = __fadd_rn(d2.push[col], acc);
            if (d2.push2)
                pr = __fadd_rn(pr, d2.push2[col]);
            d2.push[col] = pr;
            acc = 0.f;
        }
        if (MODE == kModeOpt)
            opt_epilogue<1>(dst_row, d2, col, &acc);
        else if (HAND == kHandSignal)
            st1_sc1(dst_row + col, acc);
        else
            __builtin_nontemporal_store(acc, dst_row + col);
        if (DUAL && d2.on)
            __builtin_nontemporal_store(acc2, d2.row + col);
    }
    if (HAND == kHandForward) {
        for (int j0 = 0; j0 < fw->m; j0 += kWave) {
            const int dv = fwd_dests(hd, *fw, j0);
            const int cnt = min(kWave, fw->m - j0);
            for (int j = 0; j < cnt; ++j) {
                float *o = hd.out + static_cast<size_t>(__builtin_amdgcn_readlane(dv, j)) * width;
                if (live)
                    __builtin_nontemporal_store(acc, o + col);
            }
        }
    }
}

// The same for a slice of 128 columns, TWO per lane (the applies of a finished plan, whose waves take a key each): a
// 128-float row is one pass instead of two dependent ones -- one 512-byte request per occurrence instead of two of 256 --, and
// the destination's old value is asked for in front of the first occurrence rows instead of behind the last.  The chain per
// column is medium_slice's.  `col` = first of the lane's two columns (even; width % 4 == 0, 8-byte aligned rows: the
// vectorised kernels); pv: lanes 0 .. len - 1 hold the occurrence indices.  Not for kModeOpt.
template <int MODE>
__device__ __forceinline__ void medium_pair(float *__restrict__ dst_row, const float *__restrict__ grads, int width, int col,
                                            int pv, int len, float lr, bool init) {
    const bool live = col < width;
    const int lcol = live ? col : 0;
    float2 old{0.f, 0.f};
    if ((MODE == kModeSgd && init) || MODE == kModePush)
        old = *reinterpret_cast<const float2 *>(dst_row + lcol);
    float2 g[kChunk];
    float ax = 0.f, ay = 0.f;
    bool first = true;
    for (int t0 = 0; t0 < len; t0 += kChunk) {
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            const int idx = __builtin_amdgcn_readlane(pv, min(t0 + t, len - 1));
            g[t] = *reinterpret_cast<const float2 *>(grads + static_cast<size_t>(idx) * width + lcol);
        }
        if (first && MODE == kModeSgd) {
            ax = old.x;
            ay = old.y;
        }
        first = false;
        const int cnt = len - t0;
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            const float nx = step<MODE>(ax, g[t].x, lr), ny = step<MODE>(ay, g[t].y, lr);
            ax = (t < cnt) ? nx : ax;
            ay = (t < cnt) ? ny : ay;
        }
    }
    if (live) {
        if (MODE == kModePush) {
            ax = __fadd_rn(old.x, ax);
            ay = __fadd_rn(old.y, ay);
        }
        typedef float float2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(float2v{ax, ay}, reinterpret_cast<float2v *>(dst_row + col));
    }
}

// ---- long runs (L >= kLongRun): the workgroups that lie wholly inside the run ---------------------
// A compute unit pulls only ~25 GB/s from HBM and one wave keeps at most 63 loads in flight, so a run
// of hundreds of occurrence rows is neither streamed by one wave nor by one compute unit.  The run
// [s, e) is handled by its FULL workgroups (16 consecutive sorted positions, all of this key; a run
// of >= 48 has at least two, and consecutive workgroups sit on different XCDs / compute units): the
// first W = min(F, nslice) of its F full workgroups are workers, worker j owns the 64-column slices
// j, j+W, ...  All 16 waves of a worker load occurrence rows of the owned slices (16 dwords per lane
// in one batch), multiply by lr and park the products in LDS in occurrence order; one wave per slice
// then runs the ordered chain `acc = acc - m[t]` out of LDS (one ds_read_b128 per four occurrences)
// and writes the slice of the row.  Blocks of kCoopUnits (slice, occurrence) units repeat until the
// run is exhausted.
constexpr int kCoopUnits = 256;   // units of 64 floats per LDS block (64 KiB)
constexpr int kCoopScan = 1024;   // sorted positions the 16 waves scan each way for the run's ends
constexpr int kCoopPermSpan = 512;   // sorted positions each way whose occurrence indices a full workgroup keeps in LDS
constexpr size_t kApplyLdsBytes = static_cast<size_t>(kCoopUnits) * kWave * 4 + 32 * 4 + 2 * kCoopPermSpan * 4;

template <int MODE>
__device__ __forceinline__ float chain_step(float acc, float m) {
    return MODE == kModeSgd ? __fsub_rn(acc, m) : __fadd_rn(acc, m);
}

// Tolerance mode of a long run (BASELINE.json's north star allows 1e-5 relative on accumulated gradients and names
// wavefront segmented reduce for gradient coalescing): ONE 64-column slice by a whole workgroup, no ordered chain.
// lane = (row r of 4, column quad c4 of 16); wave w takes occurrences 16w .. 16w+15 of every block of 256 (four
// 16-byte loads per lane and block) and sums lr * g over them in order; the four lane groups meet as (p0 + p1) +
// (p2 + p3), the 16 wave partials through LDS as quads (a + b) + (c + d) and (q0 + q1) + (q2 + q3); one subtract /
// add per element.  The same order as q_coop of qstep.hip -- oracle/qstep_model.py tree_coop restates it bit for bit.
// s_part = 16 x 64 floats.  Called by all 16 waves; needs width % 4 == 0 and 16-byte aligned rows.
// the tree's sum over the occurrences [lo, hi) of the run at s: lr * g summed as described above, complete in every lane
// (for its column quad).  Ends with the workgroup's partials consumed (a barrier precedes any reuse of s_part by the caller).
// first / have_first: this wave's occurrence indices of the block at `lo`, fetched by the caller ahead of time (perm values, before
// any valmap).
__device__ __forceinline__ float4v coop_tree_sum(const float *__restrict__ grads, const int32_t *__restrict__ perm,
                                                 const ApplyMaps &maps, int n, float lr, int s, int lo, int hi, int width,
                                                 int col, int w, float *s_part, int first = 0, bool have_first = false) {
    const int lane = lane_id();
    const int r = lane >> 4, c4 = lane & 15;
    float4v p{0.f, 0.f, 0.f, 0.f};
    // (the occurrence indices of the block after the current one are asked for before the current block's rows: a run of
    // thousands costs one trip per block of 256, not two)
    int p_next = 0;
    if (lo + 16 * w < hi)
        p_next = have_first ? first : perm[min(s + min(lo + 16 * w + (lane & 15), hi - 1), n - 1)];
    for (int base = lo; base < hi; base += 256) {
        const int mine = base + 16 * w;   // this wave's first occurrence of the block
        if (mine >= hi)
            break;   // wave-uniform; no barrier inside the loop
        int pidx = p_next;
        if (mine + 256 < hi)
            p_next = perm[min(s + min(mine + 256 + (lane & 15), hi - 1), n - 1)];
        if (maps.valmap)
            pidx = maps.valmap[pidx];
        float4v g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o = __shfl(pidx, 4 * t + r, 64);
            g[t] = *reinterpret_cast<const float4v *>(grads + static_cast<size_t>(o) * width + col);   // clamped: branch-free
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool valid = mine + 4 * t + r < hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float q = __fadd_rn(p[k], __fmul_rn(lr, g[t][k]));
                p[k] = valid ? q : p[k];
            }
        }
    }
    auto add4 = [](float4v x, float4v y) {
        return float4v{__fadd_rn(x[0], y[0]), __fadd_rn(x[1], y[1]), __fadd_rn(x[2], y[2]), __fadd_rn(x[3], y[3])};
    };
    auto xor4 = [](float4v v, int m) {
        return float4v{__shfl_xor(v[0], m, 64), __shfl_xor(v[1], m, 64), __shfl_xor(v[2], m, 64), __shfl_xor(v[3], m, 64)};
    };
    p = add4(p, xor4(p, 16));
    p = add4(p, xor4(p, 32));
    if (lane < 16)
        *reinterpret_cast<float4v *>(s_part + w * 64 + 4 * c4) = p;
    __syncthreads();
    const float *sp = s_part + (4 * r) * 64 + 4 * c4;
    float4v total = add4(add4(*reinterpret_cast<const float4v *>(sp), *reinterpret_cast<const float4v *>(sp + 64)),
                         add4(*reinterpret_cast<const float4v *>(sp + 128), *reinterpret_cast<const float4v *>(sp + 192)));
    total = add4(total, xor4(total, 16));
    total = add4(total, xor4(total, 32));
    return total;
}

template <int MODE>
__device__ __forceinline__ void coop_slice_tree(float *__restrict__ dst_row, bool init, const float *__restrict__ grads,
                                                const int32_t *__restrict__ perm, const ApplyMaps &maps, int n, float lr,
                                                int s, int len, int width, int slice, int w, float *s_part, int first = 0,
                                                bool have_first = false) {
    const int lane = lane_id();
    const int r = lane >> 4, c4 = lane & 15;
    const int col0 = slice * kWave;
    const bool act = col0 + 4 * c4 < width;
    const int col = act ? col0 + 4 * c4 : col0;
    float4v cur{0.f, 0.f, 0.f, 0.f};
    if (MODE != kModeReduce && init)
        cur = *reinterpret_cast<const float4v *>(dst_row + col);
    const float4v total = coop_tree_sum(grads, perm, maps, n, lr, s, 0, len, width, col, w, s_part, first, have_first);
    float4v nv;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        nv[k] = MODE == kModeSgd ? __fsub_rn(cur[k], total[k]) : __fadd_rn(cur[k], total[k]);
    if (w == 0 && r == 0 && act)
        __builtin_nontemporal_store(nv, reinterpret_cast<float4v *>(dst_row + col));
    __syncthreads();   // s_part is reused by the next slice of this workgroup
}

// The same for ONE CHUNK of 256 occurrences of a run that several workgroups share (apply_listed_kernel, runs beyond
// kTreeChunk occurrences in tolerance mode): chunk `ch` of `nch` sums its occurrences with the tree, hands the sum over
// through the L2 (`sc1` store, drained; a relaxed device-scope counter per (key, slice)), and whichever workgroup arrives last
// adds the chunk sums in chunk order -- ((t0 + t1) + t2) + ... -- and applies them: the order of ha_qstep's chunked workgroup
// items (csrc/qstep.hip q_coop_r3), oracle/qstep_model.py tree_coop_chunked.  part = nch x 64 floats of this (key, slice),
// ctr = its counter (zero at the start of the launch).  Nobody waits for anybody.
constexpr int kTreeChunk = 256;
template <int MODE>
__device__ __forceinline__ void coop_chunk_tree(float *__restrict__ dst_row, bool init, const float *__restrict__ grads,
                                                const int32_t *__restrict__ perm, const ApplyMaps &maps, int n, float lr,
                                                int s, int len, int width, int slice, int w, float *s_part,
                                                float *__restrict__ part, uint32_t *__restrict__ ctr, int ch, int nch) {
    const int lane = lane_id();
    const int r = lane >> 4, c4 = lane & 15;
    const int col0 = slice * kWave;
    const bool act = col0 + 4 * c4 < width;
    const int col = act ? col0 + 4 * c4 : col0;
    const int lo = ch * kTreeChunk, hi = min(len, lo + kTreeChunk);
    float4v total = coop_tree_sum(grads, perm, maps, n, lr, s, lo, hi, width, col, w, s_part);
    if (w == 0 && r == 0)
        st4_sc1(part + static_cast<size_t>(ch) * 64 + 4 * c4, total);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();      // (the partials in s_part have been read by every wave)
    uint32_t *s_flag = reinterpret_cast<uint32_t *>(s_part);
    if (threadIdx.x == 0)      // (relaxed: the sum went through the L2 and was drained)
        *s_flag = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *s_flag + 1u == static_cast<uint32_t>(nch);
    __syncthreads();
    if (!last)
        return;
    auto add4 = [](float4v x, float4v y) {
        return float4v{__fadd_rn(x[0], y[0]), __fadd_rn(x[1], y[1]), __fadd_rn(x[2], y[2]), __fadd_rn(x[3], y[3])};
    };
    if (w != 0)
        return;       // (one wave adds the chunk sums up; no barrier follows inside this call)
    float4v cur{0.f, 0.f, 0.f, 0.f};
    if (MODE != kModeReduce && init)
        cur = *reinterpret_cast<const float4v *>(dst_row + col);
    const float *pp = part + 4 * c4;
    float4v tot{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < nch; j0 += 4) {
        float4v q0 = ld4_sc1_async(pp + static_cast<size_t>(min(j0, nch - 1)) * 64);
        float4v q1 = ld4_sc1_async(pp + static_cast<size_t>(min(j0 + 1, nch - 1)) * 64);
        float4v q2 = ld4_sc1_async(pp + static_cast<size_t>(min(j0 + 2, nch - 1)) * 64);
        float4v q3 = ld4_sc1_async(pp + static_cast<size_t>(min(j0 + 3, nch - 1)) * 64);
        wait_loads(q0, q1, q2, q3);
        tot = j0 == 0 ? q0 : add4(tot, q0);
        if (j0 + 1 < nch) tot = add4(tot, q1);
        if (j0 + 2 < nch) tot = add4(tot, q2);
        if (j0 + 3 < nch) tot = add4(tot, q3);
    }
    float4v nv;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        nv[k] = MODE == kModeSgd ? __fsub_rn(cur[k], tot[k]) : __fadd_rn(cur[k], tot[k]);
    if (r == 0 && act)
        __builtin_nontemporal_store(nv, reinterpret_cast<float4v *>(dst_row + col));
}

// The worker part of a long run [s, s+len): this workgroup owns the 64-column slices j, j+workers, ...
// All 16 waves call it (workgroup barriers inside); s_m = kCoopUnits x 64 floats of LDS.
template <int MODE, int DUAL, int HAND = kHandNone>
__device__ __forceinline__ void coop_slices(
    float *__restrict__ dst_row, bool init, Second d2, const float *__restrict__ grads,
    const int32_t *__restrict__ perm, const ApplyMaps &maps, int n, float lr, int s, int len,
    int width, int j, int workers, int w, float *s_m, uint32_t *pend_word = nullptr,
    const int32_t *s_perm = nullptr, int s_perm_first = 0, const Hand &hd = Hand{}, uint32_t key = 0) {
    const int lane = lane_id();
    const int nslice = (width + kWave - 1) / kWave;
    const int my_slices = (nslice - j + workers - 1) / workers;  // slices j, j+workers, ...
    if (!DUAL && HAND == kHandNone && MODE != kModeOpt && maps.tree_from > 0 && len >= maps.tree_from &&
        (width & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst_row) | reinterpret_cast<uintptr_t>(grads)) & 15) == 0) {
        for (int g0 = 0; g0 < my_slices; ++g0)   // every condition above is uniform over the workgroup
            coop_slice_tree<MODE>(dst_row, init, grads, perm, maps, n, lr, s, len, width, j + g0 * workers, w, s_m);
        return;
    }
    // destinations of the key in the next batch, wave w takes w, w + 16, ...: looked up before the chain (two
    // dependent reads that would otherwise follow it)
    Fwd fw{0, 0, 0};
    if (HAND == kHandForward && hd.tab != nullptr) {
        fw = fwd_resolve(hd, key, fwd_probe(hd, key));
        fw.dv = hd.nperm[min(fw.start + w + kPosPerBlock * lane, hd.n_next - 1)];
    }
    for (int g0 = 0; g0 < my_slices; g0 += 8) {
        const int sg = min(8, my_slices - g0);               // slices handled at once
        const int tlen = ((kCoopUnits / sg) / kChunk) * kChunk;  // occurrences per LDS block
        const int upc = tlen / kChunk;                       // 16-occurrence units per slice
        const int cl = w / upc;                              // this wave loads for local slice cl ...
        const int tc = w - cl * upc;                         // ... occurrences tc*16 .. tc*16+15 of the block
        const bool loader = cl < sg;
        const int lslice = j + (g0 + (loader ? cl : 0)) * workers;
        const int lcol_raw = lslice * kWave + lane;
        const int lcol = lcol_raw < width ? lcol_raw : 0;
        // chain role: wave w < sg owns local slice w
        const bool chain = w < sg;
        const int ccol = (j + (g0 + (chain ? w : 0)) * workers) * kWave + lane;
        const bool clive = chain && ccol < width;
        float acc = 0.f, acc2 = 0.f, old = 0.f;
        if (MODE == kModeSgd && init && chain)
            acc = dst_row[ccol < width ? ccol : 0];
        if (MODE == kModePush && chain)      // (in front of the occurrence rows, not behind the chain)
            old = dst_row[ccol < width ? ccol : 0];
        if (DUAL && d2.on && chain)
            acc2 = d2.row[ccol < width ? ccol : 0];
        float4v *s_wr = reinterpret_cast<float4v *>(s_m) +
                        (static_cast<size_t>(cl * tlen + tc * kChunk) / 4) * kWave + lane;
        const float4v *s_rd = reinterpret_cast<const float4v *>(s_m) +
                              (static_cast<size_t>(w * tlen) / 4) * kWave + lane;
        // the rows of block b+1 are requested before the chain of block b runs (registers are free
        // again once block b sits in LDS), so a giant run costs its ordered chain, not chain + loads
        float g[kChunk];
        // s_perm (optional): occurrence indices of the sorted positions [s_perm_first, s_perm_first + 2 * span)
        // fetched in the SAME round trip as the run boundaries, so a run that fits the span starts its row
        // requests one dependent load earlier
        auto load_idx = [&](int base) {   // occurrence index of this wave's 16 positions of a block
            const int t = base + tc * kChunk + (lane & 15);
            const int pos = min(s + t, n - 1);
            const int rel = pos - s_perm_first;
            int idx;
            if (s_perm != nullptr && rel >= 0 && rel < 2 * kCoopPermSpan)
                idx = s_perm[rel];
            else
                idx = perm[pos];
            if (maps.valmap)
                idx = maps.valmap[idx];
            return idx;
        };
        auto request = [&](int idx) {
#pragma unroll
            for (int i = 0; i < kChunk; ++i) {
                const int r = __builtin_amdgcn_readlane(idx, i);
                g[i] = (grads + static_cast<size_t>(r) * width)[lcol];
            }
        };
        int idx_next = 0;
        if (loader) {
            request(load_idx(0));
            idx_next = load_idx(tlen);      // one block ahead of the rows, two ahead of the chain
        }
        // the accumulators' initial loads are older than every row request: resolve them here, or the
        // chain's first use makes the compiler drain the in-order load counter inside the loop and the
        // rows requested for the next block are waited for before the chain instead of behind it
        asm volatile("" ::"v"(acc), "v"(acc2), "v"(old));
        for (int base = 0; base < len; base += tlen) {
            if (loader) {
#pragma unroll
                for (int i = 0; i < kChunk; i += 4)
                    s_wr[(i / 4) * kWave] = float4v{__fmul_rn(lr, g[i]), __fmul_rn(lr, g[i + 1]),
                                                    __fmul_rn(lr, g[i + 2]), __fmul_rn(lr, g[i + 3])};
            }
            __syncthreads();
            if (loader && base + tlen < len) {
                request(idx_next);
                idx_next = load_idx(base + 2 * tlen);
            }
            if (chain) {
                const int cnt = min(tlen, len - base);
                int k = 0;
#pragma unroll 4
                for (; k + 4 <= cnt; k += 4) {
                    const float4v m = s_rd[(k / 4) * kWave];
                    acc = chain_step<MODE>(acc, m[0]);
                    acc = chain_step<MODE>(acc, m[1]);
                    acc = chain_step<MODE>(acc, m[2]);
                    acc = chain_step<MODE>(acc, m[3]);
                    if (DUAL) {
                        acc2 = chain_step<MODE>(acc2, m[0]);
                        acc2 = chain_step<MODE>(acc2, m[1]);
                        acc2 = chain_step<MODE>(acc2, m[2]);
                        acc2 = chain_step<MODE>(acc2, m[3]);
                    }
                }
                if (k < cnt) {
                    const float4v m = s_rd[(k / 4) * kWave];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float nx = chain_step<MODE>(acc, m[i]);
                        acc = (k + i < cnt) ? nx : acc;
                        if (DUAL) {
                            const float nx2 = chain_step<MODE>(acc2, m[i]);
                            acc2 = (k + i < cnt) ? nx2 : acc2;
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (clive) {
            if (MODE == kModePush)
                acc = __fadd_rn(old, acc);
            if (DUAL == 2 && d2.push) {
                float pr = __fadd_rn(d2.push[ccol], acc);
                if (d2.push2)
                    pr = __fadd_rn(pr, d2.push2[ccol]);
                d2.push[ccol] = pr;
                acc = 0.f;
            }
            if (MODE == kModeOpt)
                opt_epilogue<1>(dst_row, d2, ccol, &acc);
            else if (HAND == kHandSignal)
                st1_sc1(dst_row + ccol, acc);
            else
                __builtin_nontemporal_store(acc, dst_row + ccol);
            if (DUAL && d2.on)
                __builtin_nontemporal_store(acc2, d2.row + ccol);
        }
        // this wave's slice of the row is complete: `len` occurrences x 1 slice (see signal_done)
        if (HAND == kHandSignal && chain && (j + (g0 + w) * workers) * kWave < width)
            signal_done(pend_word, len);
        if (HAND == kHandForward && hd.tab != nullptr) {
            // a key of a long run is usually frequent in the next batch as well: the finished slices go through
            // LDS and all 16 waves write them, wave w to destinations w, w + 16, ...
            if (chain)
                s_m[w * kWave + lane] = acc;
            __syncthreads();
            for (int j0 = 0; j0 < fw.m; j0 += kPosPerBlock * kWave) {
                const int dv = j0 == 0 ? fw.dv
                                       : hd.nperm[min(fw.start + j0 + w + kPosPerBlock * lane, hd.n_next - 1)];
                const int cnt = (min(kPosPerBlock * kWave, fw.m - j0) - w + kPosPerBlock - 1) / kPosPerBlock;
                for (int c = 0; c < sg; ++c) {
                    const int col = (j + (g0 + c) * workers) * kWave + lane;
                    const float v = s_m[c * kWave + lane];
                    if (col < width)
                        for (int t = 0; t < cnt; ++t)
                            __builtin_nontemporal_store(
                                v, hd.out + static_cast<size_t>(__builtin_amdgcn_readlane(dv, t)) * width + col);
                }
            }
            __syncthreads();
        }
    }
}

// Returns false when the run is shorter than kLongRun (the caller falls through to the per-wave
// paths).  Called by all 16 waves of a full workgroup; wg0 = its first sorted position.
template <int MODE, int DUAL, int HAND = kHandNone>
__device__ __forceinline__ bool coop_run(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int wg0, int w, uint32_t key, uint32_t bk, uint32_t fk,
    ApplyMaps maps, uint32_t *lds, const Hand &hd = Hand{}, int spv = 0) {
    const int lane = lane_id();
    float *s_m = reinterpret_cast<float *>(lds);
    int *s_cnt = reinterpret_cast<int *>(lds + kCoopUnits * kWave);
    int32_t *s_perm = reinterpret_cast<int32_t *>(lds + kCoopUnits * kWave + 32);
    s_perm[threadIdx.x] = spv;   // perm[wg0 - span + thread], read speculatively with the window (visible behind the barrier below)

    // run start / end from the 2 x 1024 scanned positions (wave w looked at chunk w each way)
    {
        const int qb = wg0 - kWave * (w + 1) + lane;
        const unsigned long long mb = __ballot(qb >= 0 && bk == key);
        const int cb = (~mb == 0ull) ? kWave : __builtin_clzll(~mb);
        const int qf = wg0 + kPosPerBlock + kWave * w + lane;
        const unsigned long long mf = __ballot(qf < n && fk == key);
        const int cf = (~mf == 0ull) ? kWave : __builtin_ctzll(~mf);
        if (lane == 0) {
            s_cnt[w] = cb;
            s_cnt[16 + w] = cf;
        }
    }
    __syncthreads();
    auto combine = [&](int base) {  // contiguous matches over the 16 chunks, nearest chunk first
        const int c = s_cnt[base + (lane & 15)];
        const uint32_t fullm = static_cast<uint32_t>(__ballot(c == kWave)) & 0xFFFFu;
        const int k = uniform(__builtin_ctz(~fullm));  // first chunk that is not all matches (16 = none)
        return k >= 16 ? kCoopScan : kWave * k + __builtin_amdgcn_readlane(c, k);
    };
    const int back_total = uniform(combine(0));
    int fwd_total = uniform(combine(16));
    if (back_total >= kCoopScan)
        return true;  // >= 64 full workgroups precede this one: never a worker
    const int s = wg0 - back_total;
    int e = wg0 + kPosPerBlock + fwd_total;
    while (fwd_total >= kCoopScan) {  // giant run: keep scanning forward, 1024 positions at a time
        const int qf = e + kWave * w + lane;
        const uint32_t ks = sorted[min(qf, n - 1)];
        const unsigned long long mf = __ballot(qf < n && ks == key);
        const int cf = (~mf == 0ull) ? kWave : __builtin_ctzll(~mf);
        __syncthreads();
        if (lane == 0)
            s_cnt[16 + w] = cf;
        __syncthreads();
        fwd_total = uniform(combine(16));
        e += fwd_total;
    }
    const int len = e - s;
    if (len < kLongRun)
        return false;
    const int a = (s + kPosPerBlock - 1) & ~(kPosPerBlock - 1);  // first full workgroup of the run
    const int j = (wg0 - a) / kPosPerBlock;
    const int nfull = (e - a) / kPosPerBlock;
    const int nslice = (width + kWave - 1) / kWave;
    const int workers = min(min(nfull, nslice), kCoopScan / kPosPerBlock);
    if (j >= workers)
        return true;

    uint64_t row;
    bool init = true;
    int4 pit{0, 0, 0, 0};
    if (DUAL == 2) {
        pit = maps.pos_item[wg0];
        if (pit.x < 0)
            return true;
        row = static_cast<uint64_t>(pit.x);
        init = (pit.z & kPosInit) != 0;
    } else if (maps.rowmap) {
        const int r = maps.rowmap[upos[wg0]];
        if (r < 0)
            return true;
        row = static_cast<uint64_t>(r);
        if (maps.dst_init)
            init = maps.dst_init[r] != 0;
    } else if (MODE == kModeReduce) {
        row = static_cast<uint64_t>(upos[wg0]);
    } else {
        row = key;
    }
    if (row >= dst_rows)
        return true;
    float *dst_row = dst + row * static_cast<uint64_t>(width);
    Second d2{nullptr, false};
    if (DUAL == 2) {
        d2.on = (pit.z & kPosTemp) == 0;
        d2.row = maps.dst2 + row * static_cast<uint64_t>(width);
        d2.push = ((pit.z & kPosPush) && static_cast<uint64_t>(static_cast<uint32_t>(pit.y)) < maps.push_rows)
                      ? maps.push_tab + static_cast<uint64_t>(static_cast<uint32_t>(pit.y)) * static_cast<uint64_t>(width)
                      : nullptr;
        d2.push2 = (pit.z & kPosVictimPush) ? dst + static_cast<uint64_t>(*maps.victim_row) * static_cast<uint64_t>(width) : nullptr;
    } else if (DUAL && maps.rowmap2) {
        const int r2 = maps.rowmap2[upos[wg0]];
        d2.on = r2 >= 0;
        d2.row = maps.dst2 + static_cast<uint64_t>(d2.on ? r2 : 0) * static_cast<uint64_t>(width);
    }
    if (MODE == kModeOpt)
        opt_rows(d2, maps, row, width);

    coop_slices<MODE, DUAL, HAND>(dst_row, init, d2, grads, perm, maps, n, lr, s, len, width, j, workers, w, s_m,
                                 HAND == kHandSignal ? hd.pend + pend_slot(key) : nullptr, s_perm,
                                 wg0 - kCoopPermSpan, hd, key);
    return true;
}

// One wave per sorted position p; `lds` = kApplyLdsBytes of workgroup memory (long runs only).
template <int MODE, int VEC, int DUAL, int HAND = kHandNone>
__device__ __forceinline__ bool apply_body_impl(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int p, int w, int *dbg_info, ApplyMaps maps, uint32_t *lds, const Hand &hd = Hand{});

// `dbg` (tools/timeline.py only) receives {realtime start, realtime end, role/len, shader cycles}
// per position.  The early return below is taken by whole waves of the LAST workgroup only, which is
// never a full one, so the barriers of coop_run see all 16 waves.
// Returns true when the wave did medium / long-run work (false: it left early or applied a short run).
template <int MODE, int VEC, int DUAL = false, int HAND = kHandNone>
__device__ __forceinline__ bool apply_body(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int vblock, uint32_t *lds, unsigned long long *dbg = nullptr,
    ApplyMaps maps = ApplyMaps{nullptr, nullptr, nullptr, nullptr, nullptr}, const Hand &hd = Hand{}) {
    const int w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int p = vblock * kPosPerBlock + w;
    if (p >= n)
        return false;
    if (dbg == nullptr)
        return apply_body_impl<MODE, VEC, DUAL, HAND>(dst, dst_rows, width, sorted, perm, upos, n, grads, lr, p, w, nullptr, maps, lds, hd);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    int info = 0;
    const bool heavy = apply_body_impl<MODE, VEC, DUAL, HAND>(dst, dst_rows, width, sorted, perm, upos, n, grads, lr, p, w, &info, maps, lds, hd);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane_id() == 0) {
        dbg[p * 4 + 0] = t0;
        dbg[p * 4 + 1] = t1;
        dbg[p * 4 + 2] = static_cast<unsigned long long>(info);
        dbg[p * 4 + 3] = c1 - c0;
    }
    return heavy;
}

template <int MODE, int VEC, int DUAL, int HAND>
__device__ __forceinline__ bool apply_body_impl(
    float *__restrict__ dst, uint64_t dst_rows, int width,
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ upos, int n, const float *__restrict__ grads,
    float lr, int p, int w, int *dbg_info, ApplyMaps maps, uint32_t *lds, const Hand &hd) {
    const int lane = lane_id();
    const int wg0 = p - w;
    // window of sorted positions p-16 .. p+47, plus -- speculatively, in the same round trip -- this
    // wave's share of the 1024 positions before and after the workgroup (used by long runs only)
    const int q = p - kLookBack + lane;
    const int cq = max(0, min(q, n - 1));
    const uint32_t ks = sorted[cq];
    int pv = perm[cq];
    const uint32_t bk = sorted[max(wg0 - kWave * (w + 1) + lane, 0)];
    const uint32_t fk = sorted[min(wg0 + kPosPerBlock + kWave * w + lane, n - 1)];
    // a full workgroup is a worker of its run only if the run starts less than nslice workgroups
    // before it: one more (uniform) key decides that for workgroups deep inside a long run, which
    // then leave without the scan exchange
    const int deep = wg0 - kPosPerBlock * ((width + kWave - 1) / kWave) - kPosPerBlock;
    const uint32_t dk = sorted[max(deep, 0)];
    // speculative, same round trip: this thread's share of the occurrence indices around the workgroup
    // (used only if the workgroup turns out to be a worker of a long run)
    const int spv = perm[max(0, min(wg0 - kCoopPermSpan + static_cast<int>(threadIdx.x), n - 1))];
    int4 pit{0, 0, 0, 0};      // DUAL == 2: this position's record, in the same round trip
    if (DUAL == 2)
        pit = maps.pos_item[p];
    if (maps.valmap)  // wave-uniform
        pv = maps.valmap[pv];
    const uint32_t key = static_cast<uint32_t>(
        __builtin_amdgcn_readlane(static_cast<int>(ks), kLookBack));
    // a FULL workgroup: its 16 positions hold one key (every wave of it reaches the same verdict)
    const uint32_t key_first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ks), kLookBack - w));
    const uint32_t key_last = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ks), kLookBack - w + kPosPerBlock - 1));
    if (wg0 + kPosPerBlock <= n && key_first == key_last) {
        if (deep >= 0 && dk == key)
            return false;   // >= nslice full workgroups of this run precede: not a worker
        if (dbg_info)
            *dbg_info = (w << 16) | 0x7FFF;
        if (coop_run<MODE, DUAL, HAND>(dst, dst_rows, width, sorted, perm, upos, n, grads, lr, wg0, w, key, bk, fk, maps, lds, hd, spv))
            return true;
    }
    const unsigned long long eq = __ballot(q >= 0 && q < n && ks == key);
    const uint32_t inv_lo = static_cast<uint32_t>(~eq) & 0xFFFFu;
    const int o = inv_lo == 0 ? kLookBack : (__builtin_clz(inv_lo) - 16);   // offset in the run, capped
    const unsigned long long inv_hi = (~eq) >> kLookBack;  // bit t <-> position p+t, 48 valid bits
    const int fwd = __builtin_ctzll(inv_hi | (1ull << 48));  // 1..48
    // per-wave paths: runs shorter than kLongRun; their workers sit at offsets < 16, so both ends of
    // the run are inside the window and the length is exact
    if (o >= kLookBack || fwd >= 48)
        return false;
    const int len = o + fwd;
    if (len >= kLongRun)
        return false;
    if (dbg_info)
        *dbg_info = (o << 16) | len;

    uint64_t row;
    bool init = true;
    if (DUAL == 2) {
        if (pit.x < 0)
            return false;
        row = static_cast<uint64_t>(pit.x);
        init = (pit.z & kPosInit) != 0;
    } else if (maps.rowmap) {
        const int r = maps.rowmap[upos[p]];
        if (r < 0)
            return false;  // unique key without a destination
        row = static_cast<uint64_t>(r);
        if (maps.dst_init)
            init = maps.dst_init[r] != 0;
    } else if (MODE == kModeReduce) {
        row = static_cast<uint64_t>(upos[p]);
    } else {
        row = key;
    }
    if (row >= dst_rows)
        return false;  // out-of-range id: ignored (undefined behaviour in the reference)
    float *dst_row = dst + row * static_cast<uint64_t>(width);
    Second d2{nullptr, false};
    if (DUAL == 2) {
        d2.on = (pit.z & kPosTemp) == 0;
        d2.row = maps.dst2 + row * static_cast<uint64_t>(width);
        d2.push = ((pit.z & kPosPush) && static_cast<uint64_t>(static_cast<uint32_t>(pit.y)) < maps.push_rows)
                      ? maps.push_tab + static_cast<uint64_t>(static_cast<uint32_t>(pit.y)) * static_cast<uint64_t>(width)
                      : nullptr;
        d2.push2 = (pit.z & kPosVictimPush) ? dst + static_cast<uint64_t>(*maps.victim_row) * static_cast<uint64_t>(width) : nullptr;
    } else if (DUAL && maps.rowmap2) {
        const int r2 = maps.rowmap2[upos[p]];
        d2.on = r2 >= 0;
        d2.row = maps.dst2 + static_cast<uint64_t>(d2.on ? r2 : 0) * static_cast<uint64_t>(width);
    }
    if (MODE == kModeOpt)
        opt_rows(d2, maps, row, width);

    const int nslice = (width + kWave - 1) / kWave;
    if (len <= kShortRun) {
        if (o == 0) {
            short_row<MODE, VEC, DUAL, HAND>(dst_row, grads, width, pv, kLookBack, len, lr, init, d2, hd, key);
            if (HAND == kHandSignal)   // the whole row: len occurrences x every slice
                signal_done(hd.pend + pend_slot(key), len * nslice);
        }
        return false;
    }
    const int workers = min(len, kLookBack);
    int mine = 0;
    Fwd fw{0, 0, 0};
    uint4 pe{0, 0, 0, 0};
    bool resolve = HAND == kHandForward && hd.tab != nullptr && o * kWave < width;
    if (resolve)
        pe = fwd_probe(hd, key);
    for (int c0 = o * kWave; c0 < width; c0 += workers * kWave) {
        medium_slice<MODE, DUAL, HAND>(dst_row, grads, width, c0 + lane, pv, kLookBack - o, len, lr, init, d2, hd, key, pe, &fw, resolve);
        resolve = false;
        ++mine;
    }
    if (HAND == kHandSignal && mine > 0)
        signal_done(hd.pend + pend_slot(key), len * mine);
    return o < workers;
}

}  // namespace ha
