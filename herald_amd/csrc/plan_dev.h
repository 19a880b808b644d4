// Device-side pieces of the index plan shared by plan.hip and fused.hip (see plan.hip for the
// reference semantics: np.unique / hetu::Unique<T> / PSAgent's std::map dedup).
#pragma once
#include "common.h"

namespace ha {

constexpr int kSmallMax = 36864;     // rank-by-counting up to here (keys + partial ranks = 148 KiB of the 160 KiB LDS)
constexpr uint32_t kPadKey = 0xFFFFFFFFu;

struct PlanHeader {
    int64_t n_unique;
    int64_t reserved[31];
};
static_assert(sizeof(PlanHeader) == 256, "plan header is one 256-byte line");
// plan header word of the sticky time-out flag (words 0..3 serve plan.hip / scatter.hip): set by a launch whose wait for other
// workgroups of the SAME launch gave up -- the hand-off of ha_sgd_push_pull (step.hip), the histogram exchange of the
// one-launch radix passes (plan.hip); read through ha_plan_handoff_timeout
constexpr int kHandoffFlagWord = 8;

struct PlanPtrs {
    PlanHeader *hdr;
    uint32_t *keys, *sorted, *uniq;
    int32_t *perm, *inverse, *counts, *seg, *upos;
    // radix scratch
    uint32_t *keys_alt;
    int32_t *perm_alt;
    uint32_t *hist;        // [256 * nblocks]
    uint32_t *bucket_start;  // [kRadixBuckets + 1] first sorted position of every most-significant-digit bucket
    uint32_t *pass_flags;    // [kRadixFusedBlocks] one-launch radix passes: tile t has published its histogram of pass k
    uint32_t *block_sums;  // finish scan scratch
    uint32_t *dep;         // [2 n] scratch words of a finished plan's consumers (scatter.hip: the chunk sums of tolerance mode 2)
    size_t bytes;
};

constexpr int kRadixTile = 4096;  // keys per workgroup per radix pass (2048: 48 us, 8192: 56 us, 4096: 45 us for 106,496 keys)
constexpr int kRadixBits = 11;    // digit width: ceil(32/11) = 3 passes sort any key, 2 passes 22 bits
constexpr int kRadixBuckets = 1 << kRadixBits;
constexpr int kRadixFusedBlocks = 64;  // up to this many tiles the scatter scans the histograms itself
constexpr int kFinishTile = 8192; // sorted positions per workgroup in the finish pass

static inline PlanPtrs plan_layout(void *ws, int64_t n) {
    PlanPtrs p;
    char *b = static_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *r = b ? b + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    const size_t n4 = static_cast<size_t>(n) * 4;
    p.hdr = reinterpret_cast<PlanHeader *>(take(sizeof(PlanHeader)));
    p.keys = reinterpret_cast<uint32_t *>(take(n4));
    p.sorted = reinterpret_cast<uint32_t *>(take(n4));
    p.uniq = reinterpret_cast<uint32_t *>(take(n4));
    p.perm = reinterpret_cast<int32_t *>(take(n4));
    p.inverse = reinterpret_cast<int32_t *>(take(n4));
    p.counts = reinterpret_cast<int32_t *>(take(n4));
    p.seg = reinterpret_cast<int32_t *>(take(n4 + 4));
    p.upos = reinterpret_cast<int32_t *>(take(n4));
    p.keys_alt = reinterpret_cast<uint32_t *>(take(n4));
    p.perm_alt = reinterpret_cast<int32_t *>(take(n4));
    const size_t nblk = (static_cast<size_t>(n) + kRadixTile - 1) / kRadixTile;
    p.hist = reinterpret_cast<uint32_t *>(take((nblk * kRadixBuckets + 1) * 4));
    p.bucket_start = reinterpret_cast<uint32_t *>(take((kRadixBuckets + 2) * 4));
    p.pass_flags = reinterpret_cast<uint32_t *>(take(kRadixFusedBlocks * 4));
    const size_t nfin = (static_cast<size_t>(n) + kFinishTile - 1) / kFinishTile;
    p.block_sums = reinterpret_cast<uint32_t *>(take((nfin + 1) * 4));
    p.dep = reinterpret_cast<uint32_t *>(take(2 * n4));
    p.bytes = off;
    return p;
}

template <typename IdT>
__device__ __forceinline__ uint32_t to_key(IdT v);
template <>
__device__ __forceinline__ uint32_t to_key<float>(float v) {
    const uint32_t k = f32_to_key(v);
    return k == kPadKey ? 0xFFFFFFFEu : k;  // keep the pad value out of the key space
}
template <>
__device__ __forceinline__ uint32_t to_key<uint64_t>(uint64_t v) {
    return v > 0xFFFFFFFEull ? 0xFFFFFFFEu : static_cast<uint32_t>(v);
}
template <>
__device__ __forceinline__ uint32_t to_key<uint32_t>(uint32_t v) {
    return v;
}

inline size_t rank_small_lds_bytes(int n) {
    const int npad = (n + 127) & ~127;
    return (static_cast<size_t>(npad) + 32 * 32) * 4;
}

// First radix pass, one tile of kRadixTile ids by a 256-thread block: keys[j] = to_key(ids[j]) and the
// tile's histogram of digit 0 (tile-major or digit-major, see radix_hist_kernel in plan.hip).  Shared by
// plan.hip and by the forward launch of larger batches (fused.hip), where gather blocks ride along.
// Digit of a key for one pass: `msd` = the bucket sort's single most-significant-digit pass
// (plan.hip, bucket_rank_kernel): the top kRadixBits of the table's key range, everything beyond the range in
// the last bucket -- buckets are ordered key ranges, keys keep their full value.
__device__ __forceinline__ uint32_t radix_digit(uint32_t k, int shift, bool msd) {
    const uint32_t d = k >> shift;
    return msd ? (d < kRadixBuckets - 1u ? d : kRadixBuckets - 1u) : (d & (kRadixBuckets - 1u));
}

template <typename IdT>
__device__ __forceinline__ void radix_first_tile_body(const IdT *__restrict__ ids, int n, int nblk, int tile,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ hist,
                                                      int tile_major, uint32_t *s_h /* kRadixBuckets words */,
                                                      int shift = 0, bool msd = false) {
    const int nt = blockDim.x;   // 256 beside gather blocks (fused.hip), 1024 alone
    for (int d = threadIdx.x; d < kRadixBuckets; d += nt)
        s_h[d] = 0;
    __syncthreads();
    const int base = tile * kRadixTile;
    const int end = min(n, base + kRadixTile);
    for (int j = base + threadIdx.x; j < end; j += nt) {
        const uint32_t k = to_key<IdT>(ids[j]);
        keys[j] = k;
        atomicAdd(&s_h[radix_digit(k, shift, msd)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kRadixBuckets; d += nt)
        hist[tile_major ? tile * kRadixBuckets + d : d * nblk + tile] = s_h[d];
}

// plan.hip: the radix sort of n > kSmallMax keys; first_hist_done = keys and the pass-0 histograms were
// produced by radix_first_tile_body already
int plan_radix_sort(void *ws, int64_t n, int key_bits, bool sort_only, hipStream_t stream);
// plan.hip: bucket sort (one most-significant-digit scatter + rank-by-counting inside the buckets) for
// kBucketMin < n <= kBucketMax (= kSmallMax) keys below a known limit; the first launch (radix_first_tile_body with the
// shift of bucket_shift() and msd = true) has been made by the caller
constexpr int kBucketMin = 18432;   // measured (d = 128, Criteo ids): rank-by-counting 25.8 / 36.0 / 46.6 us per step at
                                    // 13,312 / 16,640 / 19,968 ids, bucket sort 38.4 / 42.0 / 45.9
constexpr int kBucketMax = kSmallMax;   // beyond: several low-cardinality fields share a range and its equal keys make
                                        // the in-range counting quadratic (measured at n = 106,496: 232 vs 129 us per step)
inline bool bucket_sort_applies(int64_t n, uint64_t key_limit) {
    return n > kBucketMin && n <= kBucketMax && key_limit > 0 && key_limit <= 0xFFFFFFFFull;
}
inline int bucket_shift(uint64_t key_limit) {
    int bits = 0;
    while (bits < 32 && (1ull << bits) < key_limit)
        ++bits;
    return bits > kRadixBits ? bits - kRadixBits : 0;
}
int plan_bucket_sort(void *ws, int64_t n, int shift, bool sort_only, hipStream_t stream);
inline int radix_tiles(int64_t n) { return static_cast<int>((n + kRadixTile - 1) / kRadixTile); }
inline int radix_tile_major(int64_t n) { return radix_tiles(n) <= kRadixFusedBlocks ? 1 : 0; }

// ===========================================================================
// Small path: stable rank by counting.
//
// Workgroup = 1024 threads = 16 waves, owns the 32 elements i in [32*b, 32*b+32).  Lane l works for
// element i0 + (l & 31); the two half-waves (h = l >> 5) of wave w interleave the 16-byte key chunks
// of the wave's j-range, so one ds_read_b128 serves two broadcast addresses and the before / diagonal
// / after regions stay wave-uniform.  rank(i) = #{j : key_j < key_i} + #{j < i : key_j == key_i}.
// ===========================================================================
constexpr int kRankTile = 32;

template <typename IdT>
__device__ __forceinline__ void rank_tile_body_int(
    const IdT *__restrict__ ids, int n, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ sorted, int32_t *__restrict__ perm, int tile,
    uint32_t *s_mem, uint32_t *r_out = nullptr, uint32_t *key_out = nullptr) {
    const int npad = (n + 127) & ~127;      // 32 j-splits x 4 keys
    uint32_t *s_keys = s_mem;               // [npad]
    uint32_t *s_part = s_mem + npad;        // [32 splits][32 elements]

    // stage all keys: every thread issues a batch of 8 loads before the matching LDS writes
    for (int base = 0; base < npad; base += 8192) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = base + threadIdx.x + k * 1024;
            const uint32_t kv = to_key<IdT>(ids[min(j, n - 1)]);  // branch-free load
            v[k] = j < n ? kv : kPadKey;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = base + threadIdx.x + k * 1024;
            if (j < npad)
                s_keys[j] = v[k];
        }
    }
    __syncthreads();

    const int lane = lane_id();
    const int il = lane & 31, h = lane >> 5;
    const int w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int i0 = tile * kRankTile;
    const int i = i0 + il;
    const uint32_t ki = s_keys[min(i, npad - 1)];

    // wave w owns chunk pairs [pb, pe); a pair = 8 consecutive keys, half h takes keys 4h..4h+3
    const int pairs = npad >> 3;
    const int ppw = pairs >> 4;  // npad % 128 == 0
    const int pb = w * ppw, pe = pb + ppw;
    const int tile_pb = i0 >> 3, tile_pe = (i0 + kRankTile) >> 3;

    uint32_t rank = 0;
    const uint32_t *kp = s_keys + 4 * h;
    // (a) before the tile: equal keys at smaller positions come first -> "<="
    for (int p = pb; p < min(pe, tile_pb); ++p) {
        const uint4 k = *reinterpret_cast<const uint4 *>(kp + 8 * p);
        rank += (k.x <= ki);
        rank += (k.y <= ki);
        rank += (k.z <= ki);
        rank += (k.w <= ki);
    }
    // (b) inside the tile
    for (int p = max(pb, tile_pb); p < min(pe, tile_pe); ++p) {
        const uint4 k = *reinterpret_cast<const uint4 *>(kp + 8 * p);
        const int j = 8 * p + 4 * h;
        rank += (k.x < ki) || (k.x == ki && (j + 0) < i);
        rank += (k.y < ki) || (k.y == ki && (j + 1) < i);
        rank += (k.z < ki) || (k.z == ki && (j + 2) < i);
        rank += (k.w < ki) || (k.w == ki && (j + 3) < i);
    }
    // (c) after the tile -> "<"
    for (int p = max(pb, tile_pe); p < pe; ++p) {
        const uint4 k = *reinterpret_cast<const uint4 *>(kp + 8 * p);
        rank += (k.x < ki);
        rank += (k.y < ki);
        rank += (k.z < ki);
        rank += (k.w < ki);
    }
    s_part[(w * 2 + h) * 32 + il] = rank;
    __syncthreads();
    if (threadIdx.x < 32 && i < n) {
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            r += s_part[k * 32 + il];
        sorted[r] = ki;
        perm[r] = i;
        keys[i] = ki;
        if (r_out) {
            *r_out = r;
            *key_out = ki;
        }
    }
}

// float32 ids (the operator boundary, python/hetu/dataloader.py:14): same tiling, half the VALU work.
// A key is the truncation of a float, i.e. itself a float, so the staged keys stay floats and
//   [k_j <  k_i] = clamp(k_i - k_j)     (distinct integer-valued floats differ by >= 1, equal ones by 0)
//   [k_j <= k_i] = 1 - clamp(k_j - k_i)
// One v_pk_add_f32 with a negated operand and the clamp modifier evaluates the indicator for TWO keys
// and a second packed add accumulates both: 1 VALU instruction per comparison where the integer form
// (v_cmp + v_addc) needs 2.  Counts stay below 2^24, so the float accumulators are exact.  Pad keys
// are +inf (clamp(k_i - inf) = 0).  Only the key pairs overlapping the tile need the index tie-break.
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2v pk_sub_clamp01(float2v a, float2v b) {  // clamp(a - b, 0, 1) per half
    float2v d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ void rank_tile_body_f32(
    const float *__restrict__ ids, int n, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ sorted, int32_t *__restrict__ perm, int tile,
    uint32_t *s_mem, uint32_t *r_out = nullptr, uint32_t *key_out = nullptr) {
    const int npad = (n + 127) & ~127;
    float *s_keys = reinterpret_cast<float *>(s_mem);   // [npad] keys as floats
    uint32_t *s_part = s_mem + npad;                    // [32 splits][32 elements]
    for (int base = 0; base < npad; base += 8192) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = base + threadIdx.x + k * 1024;
            const float kv = static_cast<float>(to_key<float>(ids[min(j, n - 1)]));  // exact: a truncated float
            v[k] = j < n ? kv : __builtin_inff();
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = base + threadIdx.x + k * 1024;
            if (j < npad)
                s_keys[j] = v[k];
        }
    }
    __syncthreads();

    const int lane = lane_id();
    const int il = lane & 31, h = lane >> 5;
    const int w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int i0 = tile * kRankTile;
    const int i = i0 + il;
    const float ki = s_keys[min(i, npad - 1)];
    const float2v ki2 = {ki, ki};

    const int pairs = npad >> 3;
    const int ppw = pairs >> 4;
    const int pb = w * ppw, pe = pb + ppw;
    const int tile_pb = i0 >> 3, tile_pe = (i0 + kRankTile) >> 3;

    const float *kp = s_keys + 4 * h;
    float2v gt = {0.f, 0.f};   // sum of [k_j > k_i] over the keys before the tile
    float2v lt = {0.f, 0.f};   // sum of [k_j < k_i] over the keys behind the tile
    const int pa = min(pe, tile_pb);
    {   // four key quads per trip: the LDS reads of a trip are issued before its arithmetic
        int p = pb;
        for (; p + 4 <= pa; p += 4) {
            float4v k[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                k[t] = *reinterpret_cast<const float4v *>(kp + 8 * (p + t));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                gt += pk_sub_clamp01(float2v{k[t][0], k[t][1]}, ki2);
                gt += pk_sub_clamp01(float2v{k[t][2], k[t][3]}, ki2);
            }
        }
        for (; p < pa; ++p) {
            const float4v k = *reinterpret_cast<const float4v *>(kp + 8 * p);
            gt += pk_sub_clamp01(float2v{k[0], k[1]}, ki2);
            gt += pk_sub_clamp01(float2v{k[2], k[3]}, ki2);
        }
    }
    uint32_t rank = pa > pb ? 4u * static_cast<uint32_t>(pa - pb) : 0u;   // "<=" = all minus ">"
    for (int p = max(pb, tile_pb); p < min(pe, tile_pe); ++p) {
        const float4v k = *reinterpret_cast<const float4v *>(kp + 8 * p);
        const int j = 8 * p + 4 * h;
        rank += (k[0] < ki) || (k[0] == ki && (j + 0) < i);
        rank += (k[1] < ki) || (k[1] == ki && (j + 1) < i);
        rank += (k[2] < ki) || (k[2] == ki && (j + 2) < i);
        rank += (k[3] < ki) || (k[3] == ki && (j + 3) < i);
    }
    {
        int p = max(pb, tile_pe);
        for (; p + 4 <= pe; p += 4) {
            float4v k[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                k[t] = *reinterpret_cast<const float4v *>(kp + 8 * (p + t));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                lt += pk_sub_clamp01(ki2, float2v{k[t][0], k[t][1]});
                lt += pk_sub_clamp01(ki2, float2v{k[t][2], k[t][3]});
            }
        }
        for (; p < pe; ++p) {
            const float4v k = *reinterpret_cast<const float4v *>(kp + 8 * p);
            lt += pk_sub_clamp01(ki2, float2v{k[0], k[1]});
            lt += pk_sub_clamp01(ki2, float2v{k[2], k[3]});
        }
    }
    rank += static_cast<uint32_t>(lt[0] + lt[1]) - static_cast<uint32_t>(gt[0] + gt[1]);
    s_part[(w * 2 + h) * 32 + il] = rank;   // partial ranks may wrap below zero; their sum does not
    __syncthreads();
    if (threadIdx.x < 32 && i < n) {
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            r += s_part[k * 32 + il];
        const uint32_t key = static_cast<uint32_t>(ki);
        sorted[r] = key;
        perm[r] = i;
        keys[i] = key;
        if (r_out) {
            *r_out = r;
            *key_out = key;
        }
    }
}

template <typename IdT>
__device__ __forceinline__ void rank_tile_body(
    const IdT *__restrict__ ids, int n, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ sorted, int32_t *__restrict__ perm, int tile,
    uint32_t *s_mem, uint32_t *r_out = nullptr, uint32_t *key_out = nullptr) {
    rank_tile_body_int<IdT>(ids, n, keys, sorted, perm, tile, s_mem, r_out, key_out);
}
template <>
__device__ __forceinline__ void rank_tile_body<float>(
    const float *__restrict__ ids, int n, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ sorted, int32_t *__restrict__ perm, int tile,
    uint32_t *s_mem, uint32_t *r_out, uint32_t *key_out) {
    rank_tile_body_f32(ids, n, keys, sorted, perm, tile, s_mem, r_out, key_out);
}

// Small-n finish (n <= kSmallMax): ceil(n/1024) independent workgroups, one sorted position per
// thread, no inter-workgroup communication.  Block b
//   * counts the heads (first position of every run) before its chunk by re-reading the sorted keys of
//     chunks 0..b-1 -- at most 14 x 2 L2-resident loads per thread, all independent, instead of a
//     dependent scan across workgroups;
//   * ranks its own heads with a ballot per wave + 16 wave totals in LDS, and writes
//     upos / inverse / uniq / seg for its positions;
//   * writes counts[u] = next head - this head for its heads; only the last run of the chunk can end
//     outside it, and its end is found by probing the following chunks (the first probe is requested
//     together with everything else, so the common case costs no extra round trip).
// Registers: a handful per thread -- this body shares a kernel with the apply blocks and must not
// raise their budget.  LDS: 64 words.  For batches up to kFinishChunkedMax ids the same body runs
// behind one counting launch (heads per chunk) instead of recounting: two launches in all.
constexpr int kFinishChunkedMax = 1 << 20;
constexpr int kPlanLongRun = 48;   // = kLongRun of scatter_dev.h: the chunked finish lists the keys with such runs

// Optional per-unique-key hook of the finish (the embedding cache): the thread that writes uniq[u] also
// probes the cache's direct map and, for a lookup, takes syncEmbedding's pull decision -- random,
// translation-bound accesses that spread over every finish workgroup here instead of running on the
// single bookkeeping workgroup (cache.hip).
struct HeadProbe {
    const int32_t *slot_of;   // key -> slot (-1 absent)
    long long length;
    int bypass;
    int32_t *uslot;           // [u] slot or -1
    uint32_t *flag;           // [u] 1 = miss
    int32_t *pull;            // [u] pull decision, or nullptr (update flow)
    const long long *version; // version of slot s at version[s * version_stride] (the cache keeps it inside a line record)
    int version_stride;
    const long long *srv_ver; // [row - row_start]
    long long row_start, store_rows, pull_bound;
};
__device__ __forceinline__ void head_probe(const HeadProbe &hp, int u, uint32_t k) {
    const bool known = k < static_cast<unsigned long long>(hp.length);
    const int sv = hp.slot_of[known ? k : 0];
    const int s = (!hp.bypass && known) ? sv : -1;
    hp.uslot[u] = s;
    hp.flag[u] = s < 0 ? 1u : 0u;
    if (hp.pull) {
        const long long lk = static_cast<long long>(k) - hp.row_start;
        const bool inr = lk >= 0 && lk < hp.store_rows;
        const long long v = s >= 0 ? hp.version[static_cast<long long>(s) * hp.version_stride] : -1;
        const long long srv = hp.srv_ver[inr ? lk : 0];
        hp.pull[u] = (inr && (v == -1 || srv - v > hp.pull_bound)) ? 1 : 0;
    }
}
// What the finish leaves with the thread of a sorted position for code that continues in the same launch (cache.hip,
// cache_finish_book_kernel): whether the position starts a run, the run's unique index and key.
struct HeadOut {
    bool head;
    int32_t ui;
    uint32_t key;
};
constexpr int kFinishLdsWords = 64;
inline int finish_blocks(int n) { return (n + 1023) / 1024; }

__device__ __forceinline__ void finish_block_body(
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    int n, PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq,
    int32_t *__restrict__ seg, int32_t *__restrict__ counts,
    int32_t *__restrict__ inverse, int32_t *__restrict__ upos,
    int b, uint32_t *s_w, const uint32_t *__restrict__ chunk_heads = nullptr,
    const HeadProbe *hp = nullptr, uint32_t *key_tab = nullptr, uint64_t key_rows = 0,
    uint32_t *long_list = nullptr, int long_min = 0, HeadOut *ho = nullptr) {
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    uint32_t *s_before = s_w, *s_heads = s_w + 16, *s_first = s_w + 32, *s_cand = s_w + 48;
    const int p = b * 1024 + tid;
    const int cp = min(p, n - 1);
    // everything this thread needs from memory, requested together (branch-free)
    const uint32_t k = sorted[cp];
    const uint32_t kprev = sorted[max(cp - 1, 0)];
    const int32_t pv = perm[cp];
    const int q1 = p + 1024;  // first probe behind the chunk; positions >= n act as heads
    const uint32_t k1 = sorted[min(q1, n - 1)];
    const uint32_t k1prev = sorted[max(min(q1, n - 1) - 1, 0)];
    uint32_t before = 0;
    // larger batches: the heads of every chunk were counted by a first launch (chunk_heads)
    for (int c = tid; chunk_heads != nullptr && c < b; c += 1024)
        before += chunk_heads[c];
    for (int c = 0; chunk_heads == nullptr && c < b; c += 4) {
        uint32_t x[4], y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = min(c + t, b - 1) * 1024 + tid;  // < n: chunks before b are complete
            x[t] = sorted[q];
            y[t] = sorted[max(q - 1, 0)];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            before += (c + t < b) && ((c + t) * 1024 + tid == 0 || x[t] != y[t]);
    }
    const bool valid = p < n;
    const bool head = valid && (p == 0 || k != kprev);
    const unsigned long long hm = __ballot(head);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        before += __shfl_xor(before, o, 64);
    const unsigned long long m1 = __ballot(q1 >= n || k1 != k1prev);
    if (lane == 0) {
        s_before[w] = before;
        s_heads[w] = __builtin_popcountll(hm);
        s_first[w] = hm ? static_cast<uint32_t>(w * 64 + __builtin_ctzll(hm)) : 0xFFFFFFFFu;
        s_cand[w] = m1 ? static_cast<uint32_t>(q1 + __builtin_ctzll(m1)) : 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t base = 0, woff = 0, total = 0, next_in_block = 0xFFFFFFFFu, next_after = 0xFFFFFFFFu;
    {
        const uint32_t vb = s_before[lane & 15], vh = s_heads[lane & 15];
        const uint32_t vf = s_first[lane & 15], vc = s_cand[lane & 15];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            base += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vb), t));
            const uint32_t h = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vh), t));
            woff += (t < w) ? h : 0u;
            total += h;
            const uint32_t f = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vf), t));
            if (t > w && next_in_block == 0xFFFFFFFFu)
                next_in_block = f;  // first head of a later wave of this chunk (chunk-relative)
            next_after = min(next_after, static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vc), t)));
        }
    }
    // the run that is open at the end of the chunk ends at the first head behind it
    for (int c = b + 2; next_after == 0xFFFFFFFFu; ++c) {
        const int q = c * 1024 + tid;
        const int cq = min(q, n - 1);
        const unsigned long long m = __ballot(q >= n || sorted[cq] != sorted[max(cq - 1, 0)]);
        __syncthreads();
        if (lane == 0)
            s_cand[w] = m ? static_cast<uint32_t>(q + __builtin_ctzll(m)) : 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t vc = s_cand[lane & 15];
#pragma unroll
        for (int t = 0; t < 16; ++t)
            next_after = min(next_after, static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vc), t)));
    }
    if (valid) {
        const unsigned long long upto = hm & ((2ull << lane) - 1ull);  // heads of this wave at lanes <= lane
        const int32_t ui = static_cast<int32_t>(base + woff + __builtin_popcountll(upto)) - 1;
        upos[p] = ui;
        inverse[pv] = ui;
        if (head) {
            uniq[ui] = k;
            seg[ui] = p;
            if (ho) {
                ho->head = true;
                ho->ui = ui;
                ho->key = k;
            }
            if (hp)
                head_probe(*hp, ui, k);
            const unsigned long long later = lane == 63 ? 0ull : (hm >> (lane + 1));
            int32_t nxt;
            if (later)
                nxt = p + 1 + __builtin_ctzll(later);
            else if (next_in_block != 0xFFFFFFFFu)
                nxt = b * 1024 + static_cast<int32_t>(next_in_block);
            else
                nxt = static_cast<int32_t>(min(next_after, static_cast<uint32_t>(n)));
            counts[ui] = nxt - p;
            // list of the keys with long runs (scatter.hip, apply of larger batches): counter in header word 0
            if (long_list != nullptr && nxt - p >= long_min) {
                const unsigned long long at =
                    atomicAdd(reinterpret_cast<unsigned long long *>(&hdr->reserved[0]), 1ull);
                long_list[at] = static_cast<uint32_t>(ui);
            }
            // key table of the batch (step.hip, ha_step_*): one entry per unique key of the table's range --
            // claim a slot (keys are unique here, so only different keys ever meet), then plain stores
            if (key_tab != nullptr && k < key_rows) {
                uint32_t sl = tab_slot(k);
                for (;;) {
                    uint32_t expect = kTabEmpty;
                    if (__hip_atomic_compare_exchange_strong(key_tab + 4 * sl, &expect, k, __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        break;
                    sl = (sl + 1) & kTabMask;
                }
                key_tab[4 * sl + 1] = static_cast<uint32_t>(p);
                key_tab[4 * sl + 2] = static_cast<uint32_t>(nxt - p - 1);
            }
        }
    }
    if (tid == 0 && (b + 1) * 1024 >= n) {  // the last chunk
        const uint32_t U = base + total;
        hdr->n_unique = U;
        seg[U] = n;
    }
}

}  // namespace ha
