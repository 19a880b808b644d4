// Device-side pieces of the index plan shared by plan.hip and fused.hip (see plan.hip for the
// reference semantics: np.unique / hetu::Unique<T> / PSAgent's std::map dedup).
#pragma once
#include "common.h"

namespace ha {

constexpr int kSmallMax = 15360;     // rank-by-counting up to here (keys + partial ranks = 64 KiB LDS)
constexpr uint32_t kPadKey = 0xFFFFFFFFu;

struct PlanHeader {
    int64_t n_unique;
    int64_t reserved[31];
};
static_assert(sizeof(PlanHeader) == 256, "plan header is one 256-byte line");

struct PlanPtrs {
    PlanHeader *hdr;
    uint32_t *keys, *sorted, *uniq;
    int32_t *perm, *inverse, *counts, *seg, *upos;
    // radix scratch
    uint32_t *keys_alt;
    int32_t *perm_alt;
    uint32_t *hist;        // [256 * nblocks]
    uint32_t *block_sums;  // finish scan scratch
    size_t bytes;
};

constexpr int kRadixTile = 4096;  // keys per workgroup per radix pass
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
    p.hist = reinterpret_cast<uint32_t *>(take((nblk * 256 + 1) * 4));
    const size_t nfin = (static_cast<size_t>(n) + kFinishTile - 1) / kFinishTile;
    p.block_sums = reinterpret_cast<uint32_t *>(take((nfin + 1) * 4));
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
inline size_t finish_small_lds_bytes(int n) {
    return (((static_cast<size_t>(n) + 7) & ~size_t(7)) + n + 2) * 2 + 16;
}

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
__device__ __forceinline__ void rank_tile_body(
    const IdT *__restrict__ ids, int n, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ sorted, int32_t *__restrict__ perm, int tile,
    uint32_t *s_mem) {
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
    }
}

// Small-n fused finish: one workgroup does phases 1-3 (n <= kSmallMax).
//   pass A (coalesced): head flag of every sorted position -> LDS
//   pass B (blocked)  : thread t owns positions t*15.., local count + wave/block scan -> unique index
//   pass C (coalesced): uniq / seg / counts / upos / inverse written out
// LDS: upos|head<<15 as u16[n] + seg as u16[n+2] (positions and unique indices are <= 15360 < 2^15).
__device__ __forceinline__ void finish_small_body(
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    int n, PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq,
    int32_t *__restrict__ seg, int32_t *__restrict__ counts,
    int32_t *__restrict__ inverse, int32_t *__restrict__ upos,
    uint32_t *s_dyn) {
    constexpr int kItems = kSmallMax / 1024;  // 15
    __shared__ uint32_t s_w[16];
    uint16_t *s_upos = reinterpret_cast<uint16_t *>(s_dyn);   // [n rounded up to 8]
    uint16_t *s_seg = s_upos + ((n + 7) & ~7);                // [n+2]
    const int lane = lane_id(), w = threadIdx.x >> 6;
    // ---- pass A (coalesced, batches of kBatch positions per thread to stay light on registers: this
    // body shares a kernel with the apply blocks and must not raise their register budget)
    constexpr int kBatch = 5;
    static_assert(kItems % kBatch == 0, "kItems must be a multiple of kBatch");
    for (int k0 = 0; k0 < kItems; k0 += kBatch) {
        uint32_t a[kBatch], b[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int p = threadIdx.x + (k0 + k) * 1024;
            a[k] = sorted[min(p, n - 1)];              // branch-free loads
            b[k] = sorted[max(min(p, n - 1) - 1, 0)];
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int p = threadIdx.x + (k0 + k) * 1024;
            if (p < n)
                s_upos[p] = (p == 0 || a[k] != b[k]) ? 0x8000u : 0u;
        }
        if ((k0 + kBatch) * 1024 >= n)
            break;
    }
    __syncthreads();
    // ---- pass B
    const int p0 = threadIdx.x * kItems;
    uint32_t c = 0;
    uint32_t flags = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int p = p0 + k;
        const uint32_t hd = (p < n) ? (s_upos[p] >> 15) : 0u;
        flags |= hd << k;
        c += hd;
    }
    uint32_t x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o)
            x += y;
    }
    if (lane == 63)
        s_w[w] = x;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < w)
            woff += s_w[k];
        total += s_w[k];
    }
    uint32_t u = woff + x - c;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int p = p0 + k;
        if (p < n) {
            const uint32_t hd = (flags >> k) & 1u;
            if (hd) {
                s_seg[u] = static_cast<uint16_t>(p);
                ++u;
            }
            s_upos[p] = static_cast<uint16_t>((u - 1) | (hd << 15));
        }
    }
    if (threadIdx.x == 0) {
        hdr->n_unique = total;
        s_seg[total] = static_cast<uint16_t>(n);
    }
    __syncthreads();
    // ---- pass C
    for (int k = threadIdx.x; k <= static_cast<int>(total); k += 1024) {
        const int32_t a = s_seg[k];
        seg[k] = a;
        if (k < static_cast<int>(total))
            counts[k] = static_cast<int32_t>(s_seg[k + 1]) - a;
    }
    for (int k0 = 0; k0 < kItems; k0 += kBatch) {
        uint32_t a[kBatch];
        int32_t pv[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int p = threadIdx.x + (k0 + k) * 1024;
            a[k] = sorted[min(p, n - 1)];
            pv[k] = perm[min(p, n - 1)];
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int p = threadIdx.x + (k0 + k) * 1024;
            if (p < n) {
                const uint32_t v = s_upos[p];
                const int32_t ui = static_cast<int32_t>(v & 0x7FFFu);
                upos[p] = ui;
                inverse[pv[k]] = ui;
                if (v >> 15)
                    uniq[ui] = a[k];
            }
        }
        if ((k0 + kBatch) * 1024 >= n)
            break;
    }
}

}  // namespace ha
