// Index plan: per-batch sorted-unique / inverse / counts / occurrence lists of the ids.
//
// Reference semantics reproduced bit-exactly (integer work):
//   * np.unique(ids, return_inverse=True)      python/hetu/ndarray.py:534,559
//   * hetu::Unique<T> (argsort + map)           src/hetu_cache/include/unqiue_tools.h:9-48
//   * std::map<idx, vector<pos>> dedup          ps-lite/include/ps/worker/PSAgent.h:124-183
// All three produce ASCENDING distinct keys, an inverse map, and (implicitly) the list of
// occurrence positions of each key in ascending position order.  The plan materialises
// exactly that:  sorted keys, a STABLE argsort `perm`, segment offsets `seg`, `uniq`,
// `counts`, `inverse`, and `upos` (unique index of every sorted position).
//
// Two sort engines:
//   n <= kSmallMax : rank-by-counting.  Every 1024-thread workgroup stages all keys in LDS and
//                    computes the final stable rank of 64 elements by comparing them against all
//                    n keys (16 waves split the j-range, partial ranks are summed through LDS).
//                    One launch, no inter-workgroup communication, O(n^2/64) VALU wave-instructions
//                    spread over ceil(n/64) CUs: ~2 us for the 6,656-id wdl_criteo batch, where a
//                    multi-pass radix sort would pay >= 12 dependent kernel boundaries.
//   larger n       : LSD radix sort, 11-bit digits (3 passes for any 32-bit key), wave-ballot
//                    multisplit ranking (stable); up to 64 tiles a pass is two launches.
#include "plan_dev.h"

namespace ha {

template <typename IdT>
__global__ __launch_bounds__(1024) void plan_rank_small_kernel(
    const IdT *__restrict__ ids, int n, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ sorted, int32_t *__restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    rank_tile_body<IdT>(ids, n, keys, sorted, perm, blockIdx.x, s_mem);
}

// ===========================================================================
// General path: LSD radix sort (8-bit digits), stable.
// ===========================================================================
// per-block digit histogram; hist[d * nblk + blk] (digit-major: the flat exclusive scan of the large
// path orders it) or hist[blk * kRadixBuckets + d] (tile-major: coalesced for the fused scatter)
__global__ __launch_bounds__(1024) void radix_hist_kernel(
    const uint32_t *__restrict__ keys, int n, int shift, int nblk,
    uint32_t *__restrict__ hist, int tile_major) {
    __shared__ uint32_t s_h[kRadixBuckets];
    for (int d = threadIdx.x; d < kRadixBuckets; d += 1024)
        s_h[d] = 0;
    __syncthreads();
    const int base = blockIdx.x * kRadixTile;
    const int end = min(n, base + kRadixTile);
    for (int j = base + threadIdx.x; j < end; j += 1024)
        atomicAdd(&s_h[(keys[j] >> shift) & (kRadixBuckets - 1u)], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < kRadixBuckets; d += 1024)
        hist[tile_major ? blockIdx.x * kRadixBuckets + d : d * nblk + blockIdx.x] = s_h[d];
}

// exclusive scan of `m` uint32 values in place, single workgroup of 1024 threads
__global__ __launch_bounds__(1024) void scan_exclusive_kernel(
    uint32_t *__restrict__ data, int m, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0)
        s_carry = 0;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int base = 0; base < m; base += 1024) {
        const int j = base + threadIdx.x;
        const uint32_t v = j < m ? data[j] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o)
                x += y;
        }
        if (lane == 63)
            s_w[w] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k)
            woff += s_w[k];
        const uint32_t carry = s_carry;
        if (j < m)
            data[j] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023)
            s_carry = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out)
        *total_out = s_carry;
}

// stable scatter of one digit pass (kRadixBits = 11: three passes sort any 32-bit key, two sort 22
// bits).  One 1024-thread workgroup per tile of 4,096 keys: each of the 16 waves owns a contiguous 256 keys
// and walks them in four rows of 64; within a row, lanes with equal digits are ranked by a ballot match (11
// ballots), the per-wave running digit counters live in LDS as 16-bit words (16 x 2048: a wave sees at most
// 256 keys).  Position of a key = global base of its digit in this tile (s_base) + keys of the digit in the
// tile's earlier waves (the counters after an exclusive prefix over the waves) + its rank inside the wave.
// PRESCANNED = false (up to kRadixFusedBlocks tiles): the block derives its own global digit bases from the
// raw per-block histograms -- every thread sums two digits over all tiles, one block scan orders the digits --
// so a pass is two launches (histogram, scatter) instead of three.  PRESCANNED = true: `hist` was scanned by
// scan_exclusive_kernel.  (256-thread blocks with sixteen rows per wave, the first version: 12.7 us per pass at
// 106,496 keys -- 26 blocks of four waves are a latency chain, not a throughput problem.)
constexpr int kScatterThreads = 1024;
constexpr int kScatterWaves = kScatterThreads / kWave;
constexpr size_t kScatterLdsBytes =
    static_cast<size_t>(kScatterWaves) * kRadixBuckets * 2 + 3 * kRadixBuckets * 4 + kScatterWaves * 4;

// FUSED (passes after the first, up to kRadixFusedBlocks tiles -- all of them resident at once): the pass is ONE launch.
// The ranking leaves the tile's own digit histogram in LDS; the workgroup publishes it (device-coherent stores, then
// its flag = `flag_value`), waits until every tile of the pass has done so, and sums the others' histograms with
// device-coherent loads -- a barrier across the 26 workgroups of a 106,496-key pass in the middle of the launch instead
// of a kernel boundary and a second read of the keys.  The flags are zeroed by the first pass's scatter (`flags_reset`),
// pass k waits for the value k: no epoch survives a call, so the launches replay from a hipGraph unchanged.
template <bool PRESCANNED, bool MSD = false, bool FUSED = false>
__device__ __forceinline__ void radix_scatter_body(
    const int bx,      // the tile (bx of a launch over ONE sort; of the batched launch: within the sort blockIdx.y names)
    const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ perm_in,
    int n, int shift, int nblk, const uint32_t *hist,
    uint32_t *__restrict__ keys_out, int32_t *__restrict__ perm_out,
    uint32_t *__restrict__ bucket_start, uint32_t *flags, uint32_t flag_value,
    uint32_t *flags_reset, long long *timeout_word) {
    constexpr int kRows = kRadixTile / kScatterThreads;        // rows of 64 keys per wave
    constexpr int kPerThread = kRadixBuckets / kScatterThreads;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    uint16_t *s_cnt = reinterpret_cast<uint16_t *>(s_dyn);                 // [waves][kRadixBuckets]
    uint32_t *s_base = s_dyn + kScatterWaves * kRadixBuckets / 2;          // [kRadixBuckets]
    uint32_t *s_tot = s_base + kRadixBuckets, *s_mine = s_tot + kRadixBuckets;
    uint32_t *s_scan = s_mine + kRadixBuckets;                             // [waves]
    const int lane = lane_id(), w = threadIdx.x >> 6;
    // HA_RADIX_STAMPS=1 (tools/radix_phases.py): the last workgroup's clock at its phase boundaries, in the bucket-start
    // scratch of the (unused here) bucket sort; slot 8 * pass + phase
    unsigned long long *stamps = (!MSD && bucket_start != nullptr && bx == nblk - 1 && threadIdx.x == 0)
                                     ? reinterpret_cast<unsigned long long *>(bucket_start) + 8 * flag_value : nullptr;
#define RADIX_STAMP(i)                                                  \
    do {                                                                \
        if (stamps)                                                     \
            stamps[i] = __builtin_amdgcn_s_memrealtime();               \
    } while (0)
    RADIX_STAMP(0);
    for (int k = threadIdx.x; k < kScatterWaves * kRadixBuckets / 2; k += kScatterThreads)
        s_dyn[k] = 0;
    if (flags_reset != nullptr && bx == 0 && threadIdx.x < kRadixFusedBlocks)
        flags_reset[threadIdx.x] = 0;      // the fused passes behind this launch count from here
    __syncthreads();

    const int wbase = bx * kRadixTile + w * (kRows * 64);
    uint32_t key[kRows];
    int32_t val[kRows];
    uint32_t lrank[kRows];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint16_t *my_cnt = s_cnt + w * kRadixBuckets;
    // all keys / values of the wave's rows first (branch-free, one batch of loads), then the ranking
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int j = min(wbase + r * 64 + lane, n - 1);
        key[r] = keys_in[j];
        val[r] = perm_in ? perm_in[j] : j;
    }
    RADIX_STAMP(1);
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int j = wbase + r * 64 + lane;
        const bool ok = j < n;
        const uint32_t d = radix_digit(key[r], shift, MSD);
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        // m: valid lanes with my digit
        const uint32_t before = __popcll(m & lt_mask);
        const uint32_t cnt = __popcll(m);
        uint32_t basec = 0;
        if (ok)
            basec = my_cnt[d];
        lrank[r] = basec + before;
        // all lanes of the group have read basec before the leader bumps it
        __builtin_amdgcn_wave_barrier();
        if (ok && before == 0)
            my_cnt[d] = static_cast<uint16_t>(basec + cnt);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    RADIX_STAMP(2);
    if (FUSED) {
        // this tile's histogram, then the flag.  A count is at most kRadixTile = 4096: two digits per word (thread t the
        // digits 2t, 2t + 1 -- the ones it owns below), half the bytes every workgroup pulls through its one compute
        // unit when it sums the tiles (26 x 8 KB as 32-bit counts: 3.6 us of an 11.9 us pass)
        static_assert(kPerThread == 2 && kRadixTile <= 0xFFFF, "two 16-bit counts per thread and word");
        uint32_t *mine_h = const_cast<uint32_t *>(hist) + static_cast<size_t>(bx) * kRadixBuckets;
        {
            uint32_t c0 = 0, c1 = 0;
#pragma unroll
            for (int k = 0; k < kScatterWaves; ++k) {
                const uint32_t pair = *reinterpret_cast<const uint32_t *>(s_cnt + k * kRadixBuckets + 2 * threadIdx.x);
                c0 += pair & 0xFFFFu;
                c1 += pair >> 16;
            }
            __hip_atomic_store(mine_h + threadIdx.x, c0 | (c1 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);      // the stores have left for memory before the flag follows them
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(flags + bx, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (static_cast<int>(threadIdx.x) < nblk) {
            // All tiles of a pass are resident at once (at most kRadixFusedBlocks workgroups), so every flag arrives; the
            // programming model does not promise that, hence the bound: about a second of polling, then the sticky word
            // of the plan header is set (ha_plan_handoff_timeout; IndexPlan.n_unique() raises) and the pass goes on with
            // whatever the histograms hold instead of hanging the device.
            unsigned spins = 0;
            while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != flag_value) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 20)) {
                    if (timeout_word)
                        *timeout_word = 1;
                    break;
                }
            }
        }
        __syncthreads();
    }
    RADIX_STAMP(3);
    // digits d = threadIdx.x * kPerThread ..: global base of the digit in this tile, then the exclusive
    // prefix over the waves
    {
        uint32_t run[kPerThread];
        const int d0 = threadIdx.x * kPerThread;
        if (PRESCANNED) {
#pragma unroll
            for (int q = 0; q < kPerThread; ++q)
                run[q] = hist[(d0 + q) * nblk + bx];
        } else {
            // coalesced pass over the tile-major histograms (thread t sums digits t, t+1024), then
            // through LDS to the thread that owns consecutive digits
            uint32_t tot[kPerThread], mine[kPerThread];
#pragma unroll
            for (int q = 0; q < kPerThread; ++q)
                tot[q] = mine[q] = 0;
            if (FUSED) {
                // packed counts, already with the thread that owns the two digits: one batch of device-coherent loads
                constexpr int kPk = 32;
                for (int k0 = 0; k0 < nblk; k0 += kPk) {
                    uint32_t v[kPk];
#pragma unroll
                    for (int kk = 0; kk < kPk; ++kk)
                        v[kk] = __hip_atomic_load(hist + static_cast<size_t>(min(k0 + kk, nblk - 1)) * kRadixBuckets + threadIdx.x,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int kk = 0; kk < kPk; ++kk) {
                        const int k = k0 + kk;
                        const uint32_t lo = k < nblk ? v[kk] & 0xFFFFu : 0u, hi = k < nblk ? v[kk] >> 16 : 0u;
                        tot[0] += lo;
                        tot[1] += hi;
                        mine[0] += k < static_cast<int>(bx) ? lo : 0u;
                        mine[1] += k < static_cast<int>(bx) ? hi : 0u;
                    }
                }
            }
            constexpr int kTrip = FUSED ? 1 : 8;               // tiles per trip: 2 x kTrip independent loads in flight (the fused
                                                     // pass reads past the L2: fewer, longer trips)
            for (int k0 = 0; !FUSED && k0 < nblk; k0 += kTrip) {
                uint32_t v[kTrip][kPerThread];
#pragma unroll
                for (int kk = 0; kk < kTrip; ++kk) {
                    const uint32_t *h = hist + static_cast<size_t>(min(k0 + kk, nblk - 1)) * kRadixBuckets + threadIdx.x;
#pragma unroll
                    for (int q = 0; q < kPerThread; ++q)
                        v[kk][q] = FUSED ? __hip_atomic_load(h + q * kScatterThreads, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)
                                         : h[q * kScatterThreads];
                }
#pragma unroll
                for (int kk = 0; kk < kTrip; ++kk) {
                    const int k = k0 + kk;
#pragma unroll
                    for (int q = 0; q < kPerThread; ++q) {
                        mine[q] += k < static_cast<int>(bx) ? v[kk][q] : 0u;   // k < bx implies k < nblk
                        tot[q] += k < nblk ? v[kk][q] : 0u;
                    }
                }
            }
            if (!FUSED) {
#pragma unroll
                for (int q = 0; q < kPerThread; ++q) {
                    s_tot[q * kScatterThreads + threadIdx.x] = tot[q];
                    s_mine[q * kScatterThreads + threadIdx.x] = mine[q];
                }
                __syncthreads();
            }
            RADIX_STAMP(4);
            if (!FUSED) {
#pragma unroll
                for (int q = 0; q < kPerThread; ++q) {
                    tot[q] = s_tot[d0 + q];
                    mine[q] = s_mine[d0 + q];
                }
            }
            uint32_t local = 0;
#pragma unroll
            for (int q = 0; q < kPerThread; ++q)
                local += tot[q];
            uint32_t x = local;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(x, o, 64);
                if (lane >= o)
                    x += y;
            }
            if (lane == 63)
                s_scan[w] = x;
            __syncthreads();
            uint32_t off = x - local;
            for (int k = 0; k < w; ++k)
                off += s_scan[k];
#pragma unroll
            for (int q = 0; q < kPerThread; ++q) {
                if (MSD && bx == 0)       // first sorted position of every bucket (mine == 0 here)
                    bucket_start[d0 + q] = off;
                run[q] = off + mine[q];
                off += tot[q];
            }
            if (MSD && bx == 0 && threadIdx.x == kScatterThreads - 1)
                bucket_start[kRadixBuckets] = off;
        }
#pragma unroll
        for (int q = 0; q < kPerThread; ++q) {
            s_base[d0 + q] = run[q];
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < kScatterWaves; ++k) {
                const uint32_t c = s_cnt[k * kRadixBuckets + d0 + q];
                s_cnt[k * kRadixBuckets + d0 + q] = static_cast<uint16_t>(acc);
                acc += c;
            }
        }
    }
    __syncthreads();
    RADIX_STAMP(5);
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int j = wbase + r * 64 + lane;
        if (j < n) {
            const uint32_t d = radix_digit(key[r], shift, MSD);
            const uint32_t pos = s_base[d] + my_cnt[d] + lrank[r];
            keys_out[pos] = key[r];
            perm_out[pos] = val[r];
        }
    }
    if (stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        stamps[6] = __builtin_amdgcn_s_memrealtime();
    }
#undef RADIX_STAMP
}

template <bool PRESCANNED, bool MSD = false, bool FUSED = false>
__global__ __launch_bounds__(kScatterThreads) void radix_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ perm_in,
    int n, int shift, int nblk, const uint32_t *hist,
    uint32_t *__restrict__ keys_out, int32_t *__restrict__ perm_out,
    uint32_t *__restrict__ bucket_start = nullptr, uint32_t *flags = nullptr, uint32_t flag_value = 0,
    uint32_t *flags_reset = nullptr, long long *timeout_word = nullptr) {
    radix_scatter_body<PRESCANNED, MSD, FUSED>(static_cast<int>(blockIdx.x), keys_in, perm_in, n, shift, nblk, hist, keys_out,
                                               perm_out, bucket_start, flags, flag_value, flags_reset, timeout_word);
}

// The same pass for up to kRadixBatchMax sorts in ONE launch (blockIdx.y = the sort; plan_build_batch: the plans of a block of
// batches built ahead -- 16 sorts of 106,496 keys as 16 x 6 launches of 26 workgroups each were 53 us apiece, a chain of launch
// latencies; as 6 launches of 416 workgroups they are a throughput problem).  A sort's tiles wait for each other only (FUSED),
// and they are consecutive in dispatch order.
constexpr int kRadixBatchMax = 16;
struct RadixBatch {
    int n[kRadixBatchMax], nblk[kRadixBatchMax];
    const uint32_t *kin[kRadixBatchMax];
    const int32_t *vin[kRadixBatchMax];
    uint32_t *hist[kRadixBatchMax], *kout[kRadixBatchMax], *flags[kRadixBatchMax];
    int32_t *vout[kRadixBatchMax];
    long long *timeout_word[kRadixBatchMax];
};
template <bool FUSED>
__global__ __launch_bounds__(kScatterThreads) void radix_scatter_batch_kernel(const RadixBatch b, int shift, uint32_t flag_value) {
    const int i = blockIdx.y;
    if (static_cast<int>(blockIdx.x) >= b.nblk[i])
        return;
    radix_scatter_body<false, false, FUSED>(static_cast<int>(blockIdx.x), b.kin[i], b.vin[i], b.n[i], shift, b.nblk[i], b.hist[i],
                                            b.kout[i], b.vout[i], nullptr, FUSED ? b.flags[i] : nullptr, flag_value,
                                            FUSED ? nullptr : b.flags[i], b.timeout_word[i]);
}

// ===========================================================================
// Bucket sort, last launch: stable rank by counting INSIDE the buckets.
//
// After one most-significant-digit scatter the keys are grouped into 2,048 ordered key ranges (stable inside
// a range); the final position of a key is the start of its range plus its rank among the keys of that range.
// A workgroup owns 32 consecutive grouped positions and counts against the union of the ranges its elements
// lie in -- every key of a lower range is smaller, every key of a higher one larger, so the plain comparison
// over the union gives the global rank minus the union's start.  The union streams through LDS in chunks of
// 8,192 keys, so a range of any size works; the work is the sum of (range size)^2 / 64 instead of n^2 / 64:
// for a Criteo batch of 26,624 (106,496) ids ~30x (~35x) less than ranking against all keys, and three
// launches in all where the 11-bit LSD sort needs six.
// ===========================================================================
constexpr int kBucketChunk = 8192;

__global__ __launch_bounds__(1024) void bucket_rank_kernel(
    const uint32_t *__restrict__ keys_g, const int32_t *__restrict__ perm_g, int n, int shift,
    const uint32_t *__restrict__ bucket_start, uint32_t *__restrict__ sorted, int32_t *__restrict__ perm) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[kBucketChunk];
    __shared__ uint32_t s_part[32 * 32];
    const int lane = lane_id();
    const int il = lane & 31, h = lane >> 5;
    const int w = uniform(static_cast<int>(threadIdx.x >> 6));
    const int i0 = blockIdx.x * kRankTile;
    const int i = i0 + il;
    const uint32_t ki = keys_g[min(i, n - 1)];
    const uint32_t kfirst = keys_g[i0], klast = keys_g[min(i0 + kRankTile - 1, n - 1)];
    const int lo = static_cast<int>(bucket_start[radix_digit(kfirst, shift, true)]);
    const int hi = static_cast<int>(bucket_start[radix_digit(klast, shift, true) + 1]);
    uint32_t rank = 0;
    for (int c0 = lo; c0 < hi; c0 += kBucketChunk) {
        const int cn = min(kBucketChunk, hi - c0);
        const int npad = (cn + 127) & ~127;
        {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = static_cast<int>(threadIdx.x) + k * 1024;
                const uint32_t kv = keys_g[min(c0 + t, n - 1)];   // branch-free load
                v[k] = t < cn ? kv : kPadKey;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = static_cast<int>(threadIdx.x) + k * 1024;
                if (t < npad)
                    s_keys[t] = v[k];
            }
        }
        __syncthreads();
        const int pairs = npad >> 3, ppw = pairs >> 4;   // npad % 128 == 0
        const uint32_t *kp = s_keys + 4 * h;
        for (int p = w * ppw; p < (w + 1) * ppw; ++p) {
            const uint4 k = *reinterpret_cast<const uint4 *>(kp + 8 * p);
            const int j = c0 + 8 * p + 4 * h;
            rank += (k.x < ki) || (k.x == ki && (j + 0) < i);
            rank += (k.y < ki) || (k.y == ki && (j + 1) < i);
            rank += (k.z < ki) || (k.z == ki && (j + 2) < i);
            rank += (k.w < ki) || (k.w == ki && (j + 3) < i);
        }
        __syncthreads();
    }
    s_part[(w * 2 + h) * 32 + il] = rank;
    __syncthreads();
    if (threadIdx.x < 32 && i < n) {
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            r += s_part[k * 32 + il];
        sorted[lo + r] = ki;
        perm[lo + r] = perm_g[i];
    }
}

// ===========================================================================
// Finish: head flags -> scan -> uniq / seg / counts / inverse / upos.
// ===========================================================================
// Phase 1 (multi-block): number of heads per tile of kFinishTile sorted positions.
__global__ __launch_bounds__(1024) void finish_count_kernel(
    const uint32_t *__restrict__ sorted, int n,
    uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t s_w[16];
    const int base = blockIdx.x * kFinishTile;
    uint32_t c = 0;
    for (int k = 0; k < kFinishTile / 1024; ++k) {
        const int p = base + k * 1024 + threadIdx.x;
        if (p < n)
            c += (p == 0 || sorted[p] != sorted[p - 1]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_down(c, o, 64);
    if (lane_id() == 0)
        s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < 16; ++k)
            t += s_w[k];
        block_sums[blockIdx.x] = t;
    }
}

// Phase 2 (multi-block): block_sums has been exclusively scanned; block_sums[nblocks] = U.
// Each thread owns 8 CONSECUTIVE sorted positions.
__global__ __launch_bounds__(1024) void finish_write_kernel(
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    int n, const uint32_t *__restrict__ block_sums, int nblocks,
    PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq,
    int32_t *__restrict__ seg, int32_t *__restrict__ inverse,
    int32_t *__restrict__ upos) {
    constexpr int kItems = kFinishTile / 1024;
    __shared__ uint32_t s_w[16];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int p0 = blockIdx.x * kFinishTile + threadIdx.x * kItems;
    uint32_t key[kItems];
    bool head[kItems];
    uint32_t prev = (p0 > 0 && p0 - 1 < n) ? sorted[p0 - 1] : 0u;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int p = p0 + k;
        key[k] = p < n ? sorted[p] : 0u;
        head[k] = p < n && (p == 0 || key[k] != prev);
        prev = key[k];
        c += head[k];
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
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k)
        woff += s_w[k];
    uint32_t u = block_sums[blockIdx.x] + woff + x - c;  // heads before my first item
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int p = p0 + k;
        if (p < n) {
            if (head[k]) {
                uniq[u] = key[k];
                seg[u] = p;
                ++u;
            }
            const int32_t ui = static_cast<int32_t>(u) - 1;
            upos[p] = ui;
            inverse[perm[p]] = ui;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t U = block_sums[nblocks];
        hdr->n_unique = U;
        seg[U] = n;
    }
}

// Phase 3: counts[u] = seg[u+1] - seg[u]
__global__ __launch_bounds__(256) void finish_counts_kernel(
    const PlanHeader *__restrict__ hdr, const int32_t *__restrict__ seg,
    int32_t *__restrict__ counts, int n) {
    const int U = static_cast<int>(hdr->n_unique);
    int u = blockIdx.x * 256 + threadIdx.x;
    const int stride = gridDim.x * 256;
    for (; u < U; u += stride)
        counts[u] = seg[u + 1] - seg[u];
}

__global__ __launch_bounds__(1024) void finish_small_kernel(
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    int n, PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq,
    int32_t *__restrict__ seg, int32_t *__restrict__ counts,
    int32_t *__restrict__ inverse, int32_t *__restrict__ upos) {
    __shared__ uint32_t s_w[kFinishLdsWords];
    finish_block_body(sorted, perm, n, hdr, uniq, seg, counts, inverse, upos, blockIdx.x, s_w);
}

// heads per 1024-position chunk (first launch of the two-launch finish of medium-sized batches)
__device__ __forceinline__ void finish_chunk_heads_body(const uint32_t *__restrict__ sorted, int n,
                                                        uint32_t *__restrict__ chunk_heads, PlanHeader *hdr, uint32_t *s_w) {
    const int p = blockIdx.x * 1024 + threadIdx.x;
    const int cp = min(p, n - 1);
    const uint32_t k = sorted[cp], kprev = sorted[max(cp - 1, 0)];
    const unsigned long long hm = __ballot(p < n && (p == 0 || k != kprev));
    if (lane_id() == 0)
        s_w[threadIdx.x >> 6] = __builtin_popcountll(hm);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < 16; ++i)
            t += s_w[i];
        chunk_heads[blockIdx.x] = t;
        if (blockIdx.x == 0 && hdr != nullptr)
            hdr->reserved[0] = 0;   // counter of the long-run list the second launch fills
    }
}
__global__ __launch_bounds__(1024) void finish_chunk_heads_kernel(
    const uint32_t *__restrict__ sorted, int n, uint32_t *__restrict__ chunk_heads, PlanHeader *hdr) {
    __shared__ uint32_t s_w[16];
    finish_chunk_heads_body(sorted, n, chunk_heads, hdr, s_w);
}

__global__ __launch_bounds__(1024) void finish_chunked_kernel(
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm,
    int n, PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq,
    int32_t *__restrict__ seg, int32_t *__restrict__ counts,
    int32_t *__restrict__ inverse, int32_t *__restrict__ upos,
    const uint32_t *__restrict__ chunk_heads, uint32_t *__restrict__ long_list, int long_min) {
    __shared__ uint32_t s_w[kFinishLdsWords];
    finish_block_body(sorted, perm, n, hdr, uniq, seg, counts, inverse, upos, blockIdx.x, s_w, chunk_heads, nullptr,
                      nullptr, 0, long_list, long_min);
}

template <typename IdT>
__global__ __launch_bounds__(1024) void radix_first_kernel(
    const IdT *__restrict__ ids, int n, int nblk, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ hist, int tile_major, int shift, int msd) {
    __shared__ uint32_t s_h[kRadixBuckets];
    radix_first_tile_body<IdT>(ids, n, nblk, blockIdx.x, keys, hist, tile_major, s_h, shift, msd != 0);
}

__global__ void plan_empty_kernel(PlanHeader *hdr, int32_t *seg) {
    hdr->n_unique = 0;
    seg[0] = 0;
}

static int plan_finish(void *ws, int64_t n, hipStream_t stream);

template <typename IdT>
static int plan_build(const IdT *ids, int64_t n, void *ws, int key_bits,
                      bool sort_only, hipStream_t stream, uint64_t key_limit = 0) {
    HA_REQUIRE(n >= 0 && n < (1ll << 31), "plan: bad n=%ld", (long)n);
    HA_REQUIRE(ws != nullptr, "plan: null workspace");
    PlanPtrs p = plan_layout(ws, n);
    if (n == 0) {
        hipLaunchKernelGGL(plan_empty_kernel, dim3(1), dim3(1), 0, stream,
                           p.hdr, p.seg);
        HA_LAUNCH_CHECK();
        return 0;
    }
    HA_REQUIRE(ids != nullptr, "plan: null ids");
    const int ni = static_cast<int>(n);
    if (key_limit == 0 && key_bits < 32)
        key_limit = 1ull << key_bits;
    if (bucket_sort_applies(n, key_limit)) {
        const int shift = bucket_shift(key_limit);
        hipLaunchKernelGGL(radix_first_kernel<IdT>, dim3(radix_tiles(n)), dim3(1024), 0, stream, ids, ni,
                           radix_tiles(n), p.keys, p.hist, 1, shift, 1);
        HA_LAUNCH_CHECK();
        return plan_bucket_sort(ws, n, shift, sort_only, stream);
    }
    if (n <= kSmallMax) {
        const size_t lds = rank_small_lds_bytes(ni);
        HA_ALLOW_LDS(plan_rank_small_kernel<IdT>, lds);
        hipLaunchKernelGGL(plan_rank_small_kernel<IdT>,
                           dim3((ni + kRankTile - 1) / kRankTile), dim3(1024),
                           lds, stream, ids, ni, p.keys, p.sorted, p.perm);
        HA_LAUNCH_CHECK();
        if (!sort_only) {
            if (plan_finish(ws, n, stream))
                return -1;
        }
        return 0;
    }
    // ---- radix path: conversion + first histogram in one launch, then the passes
    hipLaunchKernelGGL(radix_first_kernel<IdT>, dim3(radix_tiles(n)), dim3(1024), 0, stream, ids, ni,
                       radix_tiles(n), p.keys, p.hist, radix_tile_major(n), 0, 0);
    HA_LAUNCH_CHECK();
    return plan_radix_sort(ws, n, key_bits, sort_only, stream);
}

}  // namespace ha

// keys[] and the pass-0 histograms are in place (radix_first_tile_body); runs the scatter of pass 0, the
// remaining passes and, unless sort_only, the finish
static int scatter_allow_lds() {
    using namespace ha;
    static DeviceOnce once;   // once per device, and outside any stream capture (the first call is eager)
    return once.run([]() -> int {
        HA_ALLOW_LDS((radix_scatter_kernel<false, false>), kScatterLdsBytes);
        HA_ALLOW_LDS((radix_scatter_kernel<true, false>), kScatterLdsBytes);
        HA_ALLOW_LDS((radix_scatter_kernel<false, true>), kScatterLdsBytes);
        HA_ALLOW_LDS((radix_scatter_kernel<false, false, true>), kScatterLdsBytes);
        HA_ALLOW_LDS((radix_scatter_batch_kernel<false>), kScatterLdsBytes);
        HA_ALLOW_LDS((radix_scatter_batch_kernel<true>), kScatterLdsBytes);
        return 0;
    });
}

// Diagnostics (HA_RADIX_STAMPS=1 in the environment): out_host[24] = the 100 MHz clock at the phase boundaries of the last
// radix sort's scatter launches, eight slots per pass: start, keys loaded, ranked, histograms exchanged (fused passes),
// sums read, digit bases ready, scattered.
extern "C" int ha_plan_radix_stamps(void *ws, int64_t n, uint64_t *out_host, ha_stream_t stream) {
    HA_REQUIRE(ws && out_host && n > 0, "plan_radix_stamps: bad arguments");
    ha::PlanPtrs p = ha::plan_layout(ws, n);
    HA_CHECK_HIP(hipMemcpyAsync(out_host, p.bucket_start, 24 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                ha::as_stream(stream)));
    HA_CHECK_HIP(hipStreamSynchronize(ha::as_stream(stream)));
    return 0;
}

// HA_RADIX_FUSED=0: every pass after the first as two launches (histogram, scatter) again
static bool radix_pass_fused() {
    static const bool on = [] {
        const char *e = getenv("HA_RADIX_FUSED");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

int ha::plan_radix_sort(void *ws, int64_t n, int key_bits, bool sort_only, hipStream_t stream) {
    if (scatter_allow_lds())
        return -1;
    PlanPtrs p = plan_layout(ws, n);
    const int ni = static_cast<int>(n);
    const int nblk = radix_tiles(n);
    int passes = (key_bits + kRadixBits - 1) / kRadixBits;
    if (passes < 1)
        passes = 1;
    if (passes > (32 + kRadixBits - 1) / kRadixBits)
        passes = (32 + kRadixBits - 1) / kRadixBits;
    // Pass 0 reads `keys` (kept intact) with the identity permutation; pass k writes buffer
    // B = (sorted, perm) when (passes-1-k) is even and A = (keys_alt, perm_alt) otherwise, so the
    // last pass lands in B and consecutive passes never alias.
    const uint32_t *kin = p.keys;
    const int32_t *vin = nullptr;
    for (int pass = 0; pass < passes; ++pass) {
        const bool toB = ((passes - 1 - pass) & 1) == 0;
        uint32_t *kout = toB ? p.sorted : p.keys_alt;
        int32_t *vout = toB ? p.perm : p.perm_alt;
        const int shift = pass * kRadixBits;
        const bool one_launch = radix_pass_fused() && nblk <= kRadixFusedBlocks;
        static const bool stamp_on = getenv("HA_RADIX_STAMPS") && atoi(getenv("HA_RADIX_STAMPS")) != 0;
        uint32_t *dbg = stamp_on ? p.bucket_start : nullptr;
        if (pass > 0 && !one_launch) {
            hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(1024), 0, stream,
                               kin, ni, shift, nblk, p.hist, nblk <= kRadixFusedBlocks ? 1 : 0);
            HA_LAUNCH_CHECK();
        }
        if (pass > 0 && one_launch) {
            hipLaunchKernelGGL((radix_scatter_kernel<false, false, true>), dim3(nblk), dim3(kScatterThreads),
                               kScatterLdsBytes, stream, kin, vin, ni, shift, nblk, p.hist, kout, vout,
                               dbg, p.pass_flags, static_cast<uint32_t>(pass), static_cast<uint32_t *>(nullptr),
                               reinterpret_cast<long long *>(&p.hdr->reserved[kHandoffFlagWord]));
        } else if (nblk <= kRadixFusedBlocks) {
            hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(nblk), dim3(kScatterThreads), kScatterLdsBytes,
                               stream, kin, vin, ni, shift, nblk, p.hist, kout, vout,
                               dbg, static_cast<uint32_t *>(nullptr), 0u,
                               one_launch ? p.pass_flags : static_cast<uint32_t *>(nullptr));
        } else {
            hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, stream,
                               p.hist, kRadixBuckets * nblk, static_cast<uint32_t *>(nullptr));
            HA_LAUNCH_CHECK();
            hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(nblk), dim3(kScatterThreads), kScatterLdsBytes,
                               stream, kin, vin, ni, shift, nblk, p.hist, kout, vout);
        }
        HA_LAUNCH_CHECK();
        kin = kout;
        vin = vout;
    }
    if (sort_only)
        return 0;
    return plan_finish(ws, n, stream);
}

// keys[] and the tile-major histograms of the most significant digit are in place: scatter, then rank inside
// the buckets; unless sort_only, the finish
int ha::plan_bucket_sort(void *ws, int64_t n, int shift, bool sort_only, hipStream_t stream) {
    if (scatter_allow_lds())
        return -1;
    PlanPtrs p = plan_layout(ws, n);
    const int ni = static_cast<int>(n);
    const int nblk = radix_tiles(n);
    hipLaunchKernelGGL((radix_scatter_kernel<false, true>), dim3(nblk), dim3(kScatterThreads), kScatterLdsBytes, stream, p.keys,
                       static_cast<const int32_t *>(nullptr), ni, shift, nblk, p.hist, p.keys_alt, p.perm_alt,
                       p.bucket_start);
    hipLaunchKernelGGL(bucket_rank_kernel, dim3((ni + kRankTile - 1) / kRankTile), dim3(1024), 0, stream,
                       p.keys_alt, p.perm_alt, ni, shift, p.bucket_start, p.sorted, p.perm);
    HA_LAUNCH_CHECK();
    if (sort_only)
        return 0;
    return ha_plan_finish(ws, n, stream);
}

namespace ha {

static int plan_finish(void *ws, int64_t n, hipStream_t stream) {
    HA_REQUIRE(ws != nullptr && n >= 0, "plan_finish: bad arguments");
    PlanPtrs p = plan_layout(ws, n);
    if (n == 0) {
        hipLaunchKernelGGL(plan_empty_kernel, dim3(1), dim3(1), 0, stream,
                           p.hdr, p.seg);
        HA_LAUNCH_CHECK();
        return 0;
    }
    const int ni = static_cast<int>(n);
    if (n <= kSmallMax) {
        hipLaunchKernelGGL(finish_small_kernel, dim3(finish_blocks(ni)), dim3(1024), 0, stream,
                           p.sorted, p.perm, ni, p.hdr, p.uniq, p.seg,
                           p.counts, p.inverse, p.upos);
        HA_LAUNCH_CHECK();
        return 0;
    }
    if (n <= kFinishChunkedMax) {
        // heads per chunk into the radix histogram scratch (free after the sort, >= n/16 words)
        const int chunks = finish_blocks(ni);
        // the second launch also lists the keys with runs of kPlanLongRun or more occurrences (keys_alt, counter
        // in header word 0): the apply of larger batches starts with them (scatter.hip)
        hipLaunchKernelGGL(finish_chunk_heads_kernel, dim3(chunks), dim3(1024), 0, stream, p.sorted, ni, p.hist, p.hdr);
        hipLaunchKernelGGL(finish_chunked_kernel, dim3(chunks), dim3(1024), 0, stream, p.sorted, p.perm, ni,
                           p.hdr, p.uniq, p.seg, p.counts, p.inverse, p.upos, p.hist, p.keys_alt, kPlanLongRun);
        HA_LAUNCH_CHECK();
        return 0;
    }
    const int nfin = (ni + kFinishTile - 1) / kFinishTile;
    hipLaunchKernelGGL(finish_count_kernel, dim3(nfin), dim3(1024), 0, stream,
                       p.sorted, ni, p.block_sums);
    HA_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, stream,
                       p.block_sums, nfin, p.block_sums + nfin);
    HA_LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_write_kernel, dim3(nfin), dim3(1024), 0, stream,
                       p.sorted, p.perm, ni, p.block_sums, nfin, p.hdr, p.uniq,
                       p.seg, p.inverse, p.upos);
    HA_LAUNCH_CHECK();
    {
        int blocks = (ni + 255) / 256;
        if (blocks > 2048)
            blocks = 2048;
        hipLaunchKernelGGL(finish_counts_kernel, dim3(blocks), dim3(256), 0,
                           stream, p.hdr, p.seg, p.counts, ni);
        HA_LAUNCH_CHECK();
    }
    return 0;
}

__global__ __launch_bounds__(256) void plan_export_f32_kernel(
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq,
    const int32_t *__restrict__ inverse, int n, float *__restrict__ uniq_f32,
    float *__restrict__ inverse_f32) {
    const int U = static_cast<int>(hdr->n_unique);
    int i = blockIdx.x * 256 + threadIdx.x;
    const int stride = gridDim.x * 256;
    for (; i < n; i += stride) {
        if (uniq_f32 && i < U)
            uniq_f32[i] = static_cast<float>(uniq[i]);
        if (inverse_f32)
            inverse_f32[i] = static_cast<float>(inverse[i]);
    }
}

}  // namespace ha

using namespace ha;

extern "C" size_t ha_plan_bytes(int64_t n) {
    if (n < 0)
        n = 0;
    return plan_layout(nullptr, n).bytes;
}

extern "C" int ha_plan_view_of(void *ws, int64_t n, ha_plan_view *view) {
    HA_REQUIRE(ws && view && n >= 0, "plan_view_of: bad arguments");
    PlanPtrs p = plan_layout(ws, n);
    view->n = n;
    view->n_unique = &p.hdr->n_unique;
    view->keys = p.keys;
    view->sorted = p.sorted;
    view->perm = p.perm;
    view->inverse = p.inverse;
    view->uniq = p.uniq;
    view->counts = p.counts;
    view->seg = p.seg;
    view->upos = p.upos;
    return 0;
}

extern "C" int ha_plan_build_f32ids(const float *ids, int64_t n, void *ws,
                                    ha_stream_t stream) {
    return plan_build<float>(ids, n, ws, 32, false, as_stream(stream));
}

extern "C" int ha_plan_build_u64ids(const uint64_t *ids, int64_t n, void *ws,
                                    ha_stream_t stream) {
    return plan_build<uint64_t>(ids, n, ws, 32, false, as_stream(stream));
}

extern "C" int ha_plan_build_u32keys(const uint32_t *keys, int64_t n, void *ws,
                                     int key_bits, ha_stream_t stream) {
    return plan_build<uint32_t>(keys, n, ws, key_bits, false, as_stream(stream));
}

extern "C" int ha_plan_sort_u32keys(const uint32_t *keys, int64_t n, void *ws, int key_bits,
                                    ha_stream_t stream) {
    return plan_build<uint32_t>(keys, n, ws, key_bits, true, as_stream(stream));
}

extern "C" int ha_plan_sort_f32ids(const float *ids, int64_t n, void *ws,
                                   ha_stream_t stream) {
    return plan_build<float>(ids, n, ws, 32, true, as_stream(stream));
}

extern "C" int ha_plan_sort_u64ids(const uint64_t *ids, int64_t n, void *ws,
                                   ha_stream_t stream) {
    return plan_build<uint64_t>(ids, n, ws, 32, true, as_stream(stream));
}

// ---- several small plans in two launches -------------------------------------------------------------------------
// The plans of up to kPlanBatchMax batches of at most kSmallMax ids each: blockIdx.y names the batch, blockIdx.x the
// tile of plan_rank_small_kernel / the block of finish_small_kernel.  What a caller that knows its ids several
// batches ahead uses (herald_amd/sharded.py routes a block of batches at a time beside the steps): two launches per
// block instead of two per batch.
namespace ha {

constexpr int kPlanBatchMax = 16;
struct PlanBatch {
    const void *ids[kPlanBatchMax];
    int n[kPlanBatchMax];
    PlanHeader *hdr[kPlanBatchMax];
    uint32_t *keys[kPlanBatchMax], *sorted[kPlanBatchMax], *uniq[kPlanBatchMax];
    int32_t *perm[kPlanBatchMax], *inverse[kPlanBatchMax], *counts[kPlanBatchMax], *seg[kPlanBatchMax],
        *upos[kPlanBatchMax];
};

template <typename IdT>
__global__ __launch_bounds__(1024) void plan_rank_small_batch_kernel(const PlanBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    const int i = blockIdx.y, n = b.n[i];
    if (static_cast<int>(blockIdx.x) * kRankTile >= n)
        return;
    rank_tile_body<IdT>(static_cast<const IdT *>(b.ids[i]), n, b.keys[i], b.sorted[i], b.perm[i], blockIdx.x, s_mem);
}

// The stable sort of a batch of at most kWgSortMax ids by ONE workgroup, everything in LDS: an LSD radix sort whose digit
// width follows the batch's largest key (26-bit Criteo keys: three passes of 9 bits).  plan_rank_small_batch_kernel ranks by
// counting -- n^2 / 2 comparisons spread over n / 32 workgroups with 148 KB of LDS each: 42 us of the WHOLE chip for a block of
// sixteen 6,656-id batches.  That is the right trade on the critical path of one batch (8 us) and the wrong one for plans
// that are built a block AHEAD on a side stream beside row launches that want every compute unit (the sharded step's routing,
// the cache's planned flow): sixteen workgroups for ~25 us leave the other 240 compute units alone.  Same output (keys,
// sorted keys, perm: positions ascending inside equal keys).
//   per pass: wave w counts the digits of ITS contiguous chunk (LDS atomics), one scan over (digit, wave) gives every
//   (wave, digit) its first output position, wave w walks its chunk again 64 elements at a time -- a lane's rank among the
//   lanes of its group with the same digit from dbits ballots -- and scatters.
constexpr int kWgSortMax = 8192;
constexpr int kWgSortDigitBits = 9;
constexpr size_t kWgSortLds = static_cast<size_t>(kWgSortMax) * (4 + 4 + 2 + 2) + 16 * (1 << kWgSortDigitBits) * 4;
template <typename IdT>
__global__ __launch_bounds__(1024) void plan_sort_wg_batch_kernel(const PlanBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    __shared__ uint32_t s_x[16];
    const int i = blockIdx.x, n = b.n[i];
    if (n == 0)
        return;
    uint32_t *kA = s_mem, *kB = kA + kWgSortMax;
    uint16_t *iA = reinterpret_cast<uint16_t *>(kB + kWgSortMax), *iB = iA + kWgSortMax;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(iB + kWgSortMax);      // [16 waves][D digits]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const IdT *ids = static_cast<const IdT *>(b.ids[i]);
    uint32_t mx = 0;
    for (int j = tid; j < n; j += 1024) {
        const uint32_t k = to_key<IdT>(ids[j]);
        b.keys[i][j] = k;
        kA[j] = k;
        iA[j] = static_cast<uint16_t>(j);
        mx = max(mx, k);
    }
    for (int o = 32; o >= 1; o >>= 1)
        mx = max(mx, static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx), o, 64)));
    if (lane == 0)
        s_x[w] = mx;
    __syncthreads();
    mx = 0;
    for (int k = 0; k < 16; ++k)
        mx = max(mx, s_x[k]);
    const int bits = 32 - __builtin_clz(mx | 1u);
    const int passes = (bits + kWgSortDigitBits - 1) / kWgSortDigitBits;
    const int dbits = max((bits + passes - 1) / passes, 6), D = 1 << dbits;      // (>= 64 digits: the scan below gives every thread a share)
    const int chunk = (((n + 15) / 16) + 63) & ~63;                       // elements per wave, whole groups of 64
    const int c0 = min(w * chunk, n), c1 = min(c0 + chunk, n);
    for (int p = 0; p < passes; ++p) {
        const int shift = p * dbits;
        for (int c = tid; c < 16 * D; c += 1024)
            cnt[c] = 0;
        __syncthreads();
        for (int j = c0 + lane; j < c1; j += 64)
            atomicAdd(&cnt[w * D + ((kA[j] >> shift) & (D - 1))], 1u);
        __syncthreads();
        // exclusive scan in (digit, wave) order: element e = d * 16 + wave; thread t owns e = t * E .. t * E + E - 1
        const int E = 16 * D / 1024;       // 1, 2, 4 or 8 counters per thread
        uint32_t v[8], sum = 0;
        for (int q = 0; q < 8; ++q) {
            const int e = tid * E + q;
            v[q] = (q < E) ? cnt[(e & 15) * D + (e >> 4)] : 0u;
            sum += v[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += y;
        }
        if (lane == 63)
            s_x[w] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int k = 0; k < w; ++k)
            base += s_x[k];
        uint32_t run = base + incl - sum;
        for (int q = 0; q < 8; ++q) {
            const int e = tid * E + q;
            if (q < E)
                cnt[(e & 15) * D + (e >> 4)] = run;
            run += v[q];
        }
        __syncthreads();
        // stable scatter: the wave walks its chunk in order
        for (int j0 = c0; j0 < c1; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < c1;
            const uint32_t k = valid ? kA[j] : 0u;
            const uint32_t idx = valid ? iA[j] : 0u;
            const uint32_t d = (k >> shift) & (D - 1);
            unsigned long long mask = __ballot(valid);
            for (int bit = 0; bit < dbits; ++bit) {
                const unsigned long long m = __ballot(valid && ((d >> bit) & 1u));
                mask &= ((d >> bit) & 1u) ? m : ~m;
            }
            const uint32_t rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            const uint32_t group = __builtin_popcountll(mask);
            if (valid) {
                const uint32_t at = cnt[w * D + d];
                kB[at + rank] = k;
                iB[at + rank] = static_cast<uint16_t>(idx);
                if (rank + 1u == group)
                    cnt[w * D + d] = at + group;
            }
        }
        __syncthreads();
        uint32_t *tk = kA;
        kA = kB;
        kB = tk;
        uint16_t *ti = iA;
        iA = iB;
        iB = ti;
    }
    for (int j = tid; j < n; j += 1024) {
        b.sorted[i][j] = kA[j];
        b.perm[i][j] = static_cast<int32_t>(iA[j]);
    }
}

__global__ __launch_bounds__(1024) void finish_small_batch_kernel(const PlanBatch b) {
    __shared__ uint32_t s_w[kFinishLdsWords];
    const int i = blockIdx.y, n = b.n[i];
    if (n == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            b.hdr[i]->n_unique = 0;
            b.seg[i][0] = 0;
        }
        return;
    }
    if (static_cast<int>(blockIdx.x) >= (n + 1023) / 1024)   // finish_blocks(n)
        return;
    finish_block_body(b.sorted[i], b.perm[i], n, b.hdr[i], b.uniq[i], b.seg[i], b.counts[i], b.inverse[i], b.upos[i],
                      blockIdx.x, s_w);
}

// ---- plans of kSmallMax < n <= kFinishChunkedMax ids (radix sort in one-launch passes + the two-launch finish), up to
// kRadixBatchMax of them per launch: see radix_scatter_batch_kernel
struct RadixFirstBatch {
    const void *ids[kRadixBatchMax];
    int n[kRadixBatchMax], nblk[kRadixBatchMax];
    uint32_t *keys[kRadixBatchMax], *hist[kRadixBatchMax];
};
template <typename IdT>
__global__ __launch_bounds__(1024) void radix_first_batch_kernel(const RadixFirstBatch b) {
    __shared__ uint32_t s_h[kRadixBuckets];
    const int i = blockIdx.y;
    if (static_cast<int>(blockIdx.x) >= b.nblk[i])
        return;
    radix_first_tile_body<IdT>(static_cast<const IdT *>(b.ids[i]), b.n[i], b.nblk[i], blockIdx.x, b.keys[i], b.hist[i], 1, s_h, 0,
                               false);
}
struct FinishChunkedBatch {
    int n[kRadixBatchMax];
    PlanHeader *hdr[kRadixBatchMax];
    uint32_t *sorted[kRadixBatchMax], *uniq[kRadixBatchMax], *chunk_heads[kRadixBatchMax], *long_list[kRadixBatchMax];
    int32_t *perm[kRadixBatchMax], *seg[kRadixBatchMax], *counts[kRadixBatchMax], *inverse[kRadixBatchMax], *upos[kRadixBatchMax];
};
__global__ __launch_bounds__(1024) void finish_chunk_heads_batch_kernel(const FinishChunkedBatch b) {
    __shared__ uint32_t s_w[16];
    const int i = blockIdx.y;
    if (static_cast<int>(blockIdx.x) * 1024 >= b.n[i])
        return;
    finish_chunk_heads_body(b.sorted[i], b.n[i], b.chunk_heads[i], b.hdr[i], s_w);
}
__global__ __launch_bounds__(1024) void finish_chunked_batch_kernel(const FinishChunkedBatch b, int long_min) {
    __shared__ uint32_t s_w[kFinishLdsWords];
    const int i = blockIdx.y;
    if (static_cast<int>(blockIdx.x) * 1024 >= b.n[i])
        return;
    finish_block_body(b.sorted[i], b.perm[i], b.n[i], b.hdr[i], b.uniq[i], b.seg[i], b.counts[i], b.inverse[i], b.upos[i], blockIdx.x,
                      s_w, b.chunk_heads[i], nullptr, nullptr, 0, b.long_list[i], long_min);
}

template <typename IdT>
static int plan_build_batch_radix(const IdT *const *ids, const int64_t *n, void *const *ws, int count, bool sort_only,
                                  hipStream_t stream) {
    if (scatter_allow_lds())
        return -1;
    constexpr int passes = (32 + kRadixBits - 1) / kRadixBits;
    for (int at = 0; at < count; at += kRadixBatchMax) {
        const int m = count - at < kRadixBatchMax ? count - at : kRadixBatchMax;
        RadixFirstBatch fb;
        RadixBatch rb[passes];
        FinishChunkedBatch cb;
        memset(&fb, 0, sizeof(fb));
        memset(rb, 0, sizeof(rb));
        memset(&cb, 0, sizeof(cb));
        int tiles_max = 0, chunks_max = 0;
        for (int i = 0; i < m; ++i) {
            const int ni = static_cast<int>(n[at + i]);
            PlanPtrs p = plan_layout(ws[at + i], ni);
            const int nblk = radix_tiles(ni);
            fb.ids[i] = ids[at + i]; fb.n[i] = ni; fb.nblk[i] = nblk; fb.keys[i] = p.keys; fb.hist[i] = p.hist;
            // (as plan_radix_sort: pass 0 reads `keys` with the identity permutation; pass k writes B = (sorted, perm) when
            // (passes - 1 - k) is even and A = (keys_alt, perm_alt) otherwise)
            const uint32_t *kin = p.keys;
            const int32_t *vin = nullptr;
            for (int pass = 0; pass < passes; ++pass) {
                const bool toB = ((passes - 1 - pass) & 1) == 0;
                RadixBatch &r = rb[pass];
                r.n[i] = ni; r.nblk[i] = nblk; r.kin[i] = kin; r.vin[i] = vin; r.hist[i] = p.hist;
                r.kout[i] = toB ? p.sorted : p.keys_alt;
                r.vout[i] = toB ? p.perm : p.perm_alt;
                r.flags[i] = p.pass_flags;
                r.timeout_word[i] = reinterpret_cast<long long *>(&p.hdr->reserved[kHandoffFlagWord]);
                kin = r.kout[i];
                vin = r.vout[i];
            }
            cb.n[i] = ni; cb.hdr[i] = p.hdr; cb.sorted[i] = p.sorted; cb.uniq[i] = p.uniq; cb.chunk_heads[i] = p.hist;
            cb.long_list[i] = p.keys_alt; cb.perm[i] = p.perm; cb.seg[i] = p.seg; cb.counts[i] = p.counts;
            cb.inverse[i] = p.inverse; cb.upos[i] = p.upos;
            tiles_max = nblk > tiles_max ? nblk : tiles_max;
            const int ch = finish_blocks(ni);
            chunks_max = ch > chunks_max ? ch : chunks_max;
        }
        hipLaunchKernelGGL(radix_first_batch_kernel<IdT>, dim3(tiles_max, m), dim3(1024), 0, stream, fb);
        hipLaunchKernelGGL(radix_scatter_batch_kernel<false>, dim3(tiles_max, m), dim3(kScatterThreads), kScatterLdsBytes, stream,
                           rb[0], 0, 0u);
        for (int pass = 1; pass < passes; ++pass)
            hipLaunchKernelGGL(radix_scatter_batch_kernel<true>, dim3(tiles_max, m), dim3(kScatterThreads), kScatterLdsBytes, stream,
                               rb[pass], pass * kRadixBits, static_cast<uint32_t>(pass));
        if (!sort_only) {
            hipLaunchKernelGGL(finish_chunk_heads_batch_kernel, dim3(chunks_max, m), dim3(1024), 0, stream, cb);
            hipLaunchKernelGGL(finish_chunked_batch_kernel, dim3(chunks_max, m), dim3(1024), 0, stream, cb, kPlanLongRun);
        }
        HA_LAUNCH_CHECK();
    }
    return 0;
}

template <typename IdT>
static int plan_build_batch(const IdT *const *ids, const int64_t *n, void *const *ws, int count, uint64_t key_limit,
                            hipStream_t stream, bool sort_only = false) {
    HA_REQUIRE(count >= 0 && (count == 0 || (ids && n && ws)), "plan_build_batch: bad arguments");
    bool small = true;
    for (int i = 0; i < count; ++i) {
        HA_REQUIRE(n[i] >= 0 && ws[i] != nullptr && (n[i] == 0 || ids[i] != nullptr), "plan_build_batch: bad batch %d", i);
        small = small && n[i] <= kSmallMax && !bucket_sort_applies(n[i], key_limit);
    }
    bool medium = count > 1;
    for (int i = 0; i < count; ++i)
        medium = medium && n[i] > kSmallMax && n[i] <= kFinishChunkedMax && radix_tiles(n[i]) <= kRadixFusedBlocks &&
                 !bucket_sort_applies(n[i], key_limit);
    static const bool batch_radix = !(getenv("HA_PLAN_BATCH_RADIX") && atoi(getenv("HA_PLAN_BATCH_RADIX")) == 0);
    if (medium && batch_radix)      // every batch takes the radix sort in one-launch passes: the passes of all of them per launch
        return plan_build_batch_radix<IdT>(ids, n, ws, count, sort_only, stream);
    if (!small) {   // some batch takes a multi-launch sort: one plan at a time
        for (int i = 0; i < count; ++i)
            if (plan_build<IdT>(ids[i], n[i], ws[i], 32, sort_only, stream, key_limit))
                return -1;
        return 0;
    }
    for (int at = 0; at < count; at += kPlanBatchMax) {
        const int m = count - at < kPlanBatchMax ? count - at : kPlanBatchMax;
        PlanBatch b;
        memset(&b, 0, sizeof(b));
        int nmax = 0;
        for (int i = 0; i < m; ++i) {
            const int ni = static_cast<int>(n[at + i]);
            PlanPtrs p = plan_layout(ws[at + i], ni);
            b.ids[i] = ids[at + i];
            b.n[i] = ni;
            b.hdr[i] = p.hdr; b.keys[i] = p.keys; b.sorted[i] = p.sorted; b.uniq[i] = p.uniq; b.perm[i] = p.perm;
            b.inverse[i] = p.inverse; b.counts[i] = p.counts; b.seg[i] = p.seg; b.upos[i] = p.upos;
            nmax = ni > nmax ? ni : nmax;
        }
        static const bool wg_sort = !(getenv("HA_PLAN_WG_SORT") && atoi(getenv("HA_PLAN_WG_SORT")) == 0);
        if (nmax > 0 && nmax <= kWgSortMax && wg_sort) {      // plans built ahead, beside other launches: a workgroup per batch
            HA_ALLOW_LDS(plan_sort_wg_batch_kernel<IdT>, kWgSortLds);
            hipLaunchKernelGGL(plan_sort_wg_batch_kernel<IdT>, dim3(m), dim3(1024), kWgSortLds, stream, b);
        } else if (nmax > 0) {
            const size_t lds = rank_small_lds_bytes(nmax);
            HA_ALLOW_LDS(plan_rank_small_batch_kernel<IdT>, lds);
            hipLaunchKernelGGL(plan_rank_small_batch_kernel<IdT>, dim3((nmax + kRankTile - 1) / kRankTile, m), dim3(1024),
                               lds, stream, b);
        }
        if (!sort_only)
            hipLaunchKernelGGL(finish_small_batch_kernel, dim3(nmax > 0 ? finish_blocks(nmax) : 1, m), dim3(1024), 0,
                               stream, b);
        HA_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace ha

extern "C" int ha_plan_build_batch_f32ids_lim(const float *const *ids, const int64_t *n, void *const *ws, int count,
                                              uint64_t key_limit, ha_stream_t stream) {
    return ha::plan_build_batch<float>(ids, n, ws, count, key_limit, ha::as_stream(stream));
}
extern "C" int ha_plan_build_batch_u64ids_lim(const uint64_t *const *ids, const int64_t *n, void *const *ws, int count,
                                              uint64_t key_limit, ha_stream_t stream) {
    return ha::plan_build_batch<uint64_t>(ids, n, ws, count, key_limit, ha::as_stream(stream));
}

// the stable sorts alone (keys / sorted / perm of every workspace; what ha_plan_sort_*_lim leaves), ONE launch for up to
// 16 batches of at most 36,864 ids
extern "C" int ha_plan_sort_batch_f32ids_lim(const float *const *ids, const int64_t *n, void *const *ws, int count,
                                             uint64_t key_limit, ha_stream_t stream) {
    return ha::plan_build_batch<float>(ids, n, ws, count, key_limit, ha::as_stream(stream), true);
}
extern "C" int ha_plan_sort_batch_u64ids_lim(const uint64_t *const *ids, const int64_t *n, void *const *ws, int count,
                                             uint64_t key_limit, ha_stream_t stream) {
    return ha::plan_build_batch<uint64_t>(ids, n, ws, count, key_limit, ha::as_stream(stream), true);
}

extern "C" int ha_plan_build_f32ids_lim(const float *ids, int64_t n, void *ws, uint64_t key_limit,
                                        ha_stream_t stream) {
    return plan_build<float>(ids, n, ws, 32, false, as_stream(stream), key_limit);
}
extern "C" int ha_plan_sort_f32ids_lim(const float *ids, int64_t n, void *ws, uint64_t key_limit,
                                       ha_stream_t stream) {
    return plan_build<float>(ids, n, ws, 32, true, as_stream(stream), key_limit);
}
extern "C" int ha_plan_build_u64ids_lim(const uint64_t *ids, int64_t n, void *ws, uint64_t key_limit,
                                        ha_stream_t stream) {
    return plan_build<uint64_t>(ids, n, ws, 32, false, as_stream(stream), key_limit);
}
extern "C" int ha_plan_sort_u64ids_lim(const uint64_t *ids, int64_t n, void *ws, uint64_t key_limit,
                                       ha_stream_t stream) {
    return plan_build<uint64_t>(ids, n, ws, 32, true, as_stream(stream), key_limit);
}

extern "C" int ha_plan_finish(void *ws, int64_t n, ha_stream_t stream) {
    return plan_finish(ws, n, as_stream(stream));
}

extern "C" int ha_plan_export_f32(const void *ws, int64_t n, float *uniq_f32,
                                  float *inverse_f32, ha_stream_t stream) {
    HA_REQUIRE(ws && n >= 0, "plan_export: bad arguments");
    if (n == 0 || (!uniq_f32 && !inverse_f32))
        return 0;
    PlanPtrs p = plan_layout(const_cast<void *>(ws), n);
    int blocks = static_cast<int>((n + 255) / 256);
    if (blocks > 2048)
        blocks = 2048;
    hipLaunchKernelGGL(plan_export_f32_kernel, dim3(blocks), dim3(256), 0,
                       as_stream(stream), p.hdr, p.uniq, p.inverse,
                       static_cast<int>(n), uniq_f32, inverse_f32);
    HA_LAUNCH_CHECK();
    return 0;
}
