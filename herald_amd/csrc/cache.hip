// HET embedding cache on the GPU (reference: src/hetu_cache).
//
// Reference model.  A worker-side cache of `limit` embedding lines in front of the parameter server.
// A line (Line<T>, include/embedding.h:18-149) has data[width], a lazily created grad[width], a
// `version` (-1 until first pulled) and an `updates` counter.  Per batch (src/cache.cc):
//   lookup (60-107):  Unique(keys) -> policy lookup of every unique key IN SORTED ORDER (hit = touch)
//       -> new empty lines for misses -> syncEmbedding: the server returns a row when the line's
//       version is -1 or lags by more than pull_bound; the client stores version and row and re-adds
//       its un-pushed gradient (hetu_client.cc:25-30, Line::addup) -> copy the line to every
//       occurrence -> policy insert of the misses (sorted order), evicting when size > limit;
//       evicted lines with updates != 0 wait in `evict_` for the next push.
//   update (132-197): Unique -> policy lookup (touch) -> per occurrence, in order: grad += g;
//       data += g; updates++ -> push = lines with updates > push_bound (or without data), merged with
//       the pending evictions -> server: ver[row] += updates, row += grad (PSFhandle_embedding.cc:5-28)
//       -> pushed lines: version += updates, zeroGrad.
// LRUCache (src/lru_cache.cc): insert/lookup move to the list front, evict the back.
//
// MI355X layout.  Everything lives in HBM and every step is a kernel over the batch's unique keys;
// no host round trip is needed between the kernels of one call (counts stay on the device):
//   slot_of[length]     direct map row id -> slot (-1 absent): one load per probe, no hashing
//   key/version/updates/stamp/state/hasgrad [S], data[S,width], grad[S,width]   S = limit + slack
//   free_list[S]        stack of free slots
//   LRU order = (stamp).  Every touch stamps the line with a monotone counter and appends
//       (slot, stamp) to a ring log; the log is ordered by stamp, an entry is stale when the line was
//       touched again (stamp mismatch) or left the cache.  Evicting E lines = taking the first E
//       valid entries from the log head -- exactly the list-back order of the reference, found with
//       a block scan instead of a pointer chase.  The log is compacted in place when it fills.
// LFUCache (src/lfu_cache.cc) and LFUOptCache (src/lfuopt_cache.cc) evict the oldest line of the lowest
// use bucket BEFORE each insert once size == limit.  New lines enter the lowest bucket with the newest
// stamps, so within one batch the victims are (exactly, see cache_insert_evict_kernel): the old lines of
// the lowest bucket in arrival order, then -- once those are gone -- the batch's own first inserts; a
// line of a higher bucket is taken at most once per batch, when the cache is full and the lowest
// bucket empty.  The lowest bucket therefore gets the same stamp log as LRU (an entry is stale once the
// line was promoted or evicted) and the rare higher-bucket victim is found by one scan over the slots.
// The batch index plan (plan.hip) supplies sorted unique keys / inverse / counts; occurrence-order
// accumulation and the ordered server `+=` reuse the apply kernels (scatter_dev.h, ha_apply_mapped).
#include "cache_dev.h"

extern "C" int ha_apply_mapped(float *dst, int64_t dst_rows, int64_t width,
                               const void *plan_ws, int64_t n, const float *src,
                               float lr, const int32_t *rowmap,
                               const int32_t *valmap, const uint8_t *dst_init,
                               ha_stream_t stream);
extern "C" int ha_apply_mapped2(float *dst, int64_t dst_rows, float *dst2, int64_t width,
                                const void *plan_ws, int64_t n, const float *src, float lr,
                                const int32_t *rowmap, const int32_t *rowmap2,
                                const uint8_t *dst_init, ha_stream_t stream);

namespace ha {

// phase boundary i of a single-workgroup bookkeeping body (ha_cache_phase_times); `ctl` in scope
#define CACHE_PH(i)                                                  \
    do {                                                             \
        if (threadIdx.x == 0)                                        \
            ctl->ph[i] = __builtin_amdgcn_s_memrealtime();           \
    } while (0)

#ifndef HA_CACHE_BOOK_FUSED
#define HA_CACHE_BOOK_FUSED 1
#endif
#define CACHE_GRID(n) dim3(static_cast<unsigned>(((n) + 255) / 256 > 2048 ? 2048 : ((n) + 255) / 256 < 1 ? 1 : ((n) + 255) / 256))

// ---- block-wide exclusive scan helper (1024 threads) ----------------------------------------------
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *s_w, uint32_t *total) {
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o)
            x += y;
    }
    __syncthreads();
    if (lane == 63)
        s_w[w] = x;
    __syncthreads();
    uint32_t t = s_w[lane & 15];      // the sixteen wave totals: one read per lane, four scan steps over lanes 0..15
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const uint32_t y = __shfl_up(t, o, 64);
        if ((lane & 15) >= o)
            t += y;
    }
    *total = __shfl(t, 15, 64);
    const uint32_t woff = w > 0 ? __shfl(t, w - 1, 64) : 0u;
    return woff + x - v;
}

// ring position base + off for base < cap, 0 <= off < cap: one 64-bit division per caller (for the base) instead of one
// per entry
__device__ __forceinline__ long long ring_at(long long base_mod, long long off, long long cap) {
    const long long q = base_mod + off;
    return q >= cap ? q - cap : q;
}

// block-wide exclusive scan of four 16-bit counters packed into 64 bits (every total < 65536)
__device__ __forceinline__ unsigned long long block_scan_1024_x4(unsigned long long v, unsigned long long *s_w64,
                                                                 unsigned long long *total) {
    const int lane = lane_id(), w = threadIdx.x >> 6;
    unsigned long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o, 64);
        if (lane >= o)
            x += y;
    }
    __syncthreads();
    if (lane == 63)
        s_w64[w] = x;
    __syncthreads();
    unsigned long long t = s_w64[lane & 15];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const unsigned long long y = __shfl_up(t, o, 64);
        if ((lane & 15) >= o)
            t += y;
    }
    *total = __shfl(t, 15, 64);
    const unsigned long long woff = w > 0 ? __shfl(t, w - 1, 64) : 0ull;
    return woff + x - v;
}

// ---- probe: slot of every unique key --------------------------------------------------------------
// (bodies take the calling thread's index and the number of threads that share the loop, so that the
// same code runs as its own grid or as one phase of a single-workgroup bookkeeping kernel)
__device__ __forceinline__ void cache_probe_body(
    CacheCtl *ctl, const PlanHeader *hdr, const uint32_t *uniq,
    const int32_t *slot_of, long long length, int bypass, int32_t *uslot,
    uint32_t *flag, int tid0, int nthr) {
    const int U = static_cast<int>(hdr->n_unique);
    for (int u = tid0; u < U; u += nthr) {
        const uint32_t k = uniq[u];
        int s = -1;
        if (!bypass && k < static_cast<unsigned long long>(length))
            s = slot_of[k];
        uslot[u] = s;
        flag[u] = s < 0 ? 1u : 0u;
    }
    if (tid0 == 0)
        ctl->U = U;
}
__global__ __launch_bounds__(256) void cache_probe_kernel(
    CacheCtl *ctl, const PlanHeader *hdr, const uint32_t *uniq,
    const int32_t *slot_of, long long length, int bypass, int32_t *uslot,
    uint32_t *flag) {
    cache_probe_body(ctl, hdr, uniq, slot_of, length, bypass, uslot, flag,
                     blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// exclusive scan of flag[0..U) -> rank, total -> *total_out.  Single workgroup.
__device__ __forceinline__ void cache_scan_body(
    const PlanHeader *hdr, const uint32_t *flag, uint32_t *rank, long long *total_out,
    long long *nhit_out) {
    __shared__ uint32_t s_w[16];
    const int U = static_cast<int>(hdr->n_unique);
    uint32_t carry = 0;
    if (U <= 16 * 1024) {
        // thread t owns the K = ceil(U/1024) consecutive flags t*K ..: one block scan of the per-thread
        // sums instead of one per 1024 flags
        const int K = (U + 1023) >> 10;
        const int u0 = threadIdx.x * K;
        uint32_t f[16];
        uint32_t local = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f[i] = (i < K && u0 + i < U) ? flag[u0 + i] : 0u;
            local += f[i];
        }
        uint32_t tot;
        uint32_t ex = block_scan_1024(local, s_w, &tot);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < K && u0 + i < U)
                rank[u0 + i] = ex;
            ex += f[i];
        }
        carry = tot;
    } else {
        for (int base = 0; base < U; base += 1024) {
            const int u = base + threadIdx.x;
            const uint32_t v = u < U ? flag[u] : 0u;
            uint32_t tot;
            const uint32_t ex = block_scan_1024(v, s_w, &tot);
            if (u < U)
                rank[u] = carry + ex;
            carry += tot;
        }
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        if (nhit_out)
            *nhit_out = U - static_cast<long long>(carry);
    }
}
__global__ __launch_bounds__(1024) void cache_scan_kernel(
    const PlanHeader *hdr, const uint32_t *flag, uint32_t *rank, long long *total_out,
    long long *nhit_out) {
    cache_scan_body(hdr, flag, rank, total_out, nhit_out);
}

// hits: LRU touch (stamp + log append).  misses: take a slot from the free stack, new line.
__device__ __forceinline__ void cache_assign_body(
    const CacheCtl *ctl, CacheCtl *ctl_mut, const Cache &c, const uint32_t *uniq, const uint32_t *flag,
    const uint32_t *rank, int miss_state, int tid0, int nthr) {
    const int U = static_cast<int>(ctl->U);
    const long long clock = ctl->clock, tail = ctl->log_tail, ftop = ctl->free_top;
    const long long tail_mod = tail % c.Lcap;       // the batch (<= nmax keys) is shorter than the ring
    // eight keys per thread and round: their flags / slots / ranks / keys in ONE batch of loads, the free slots of the
    // misses in a second one, then the stores -- a key-by-key loop is two dependent trips to memory per key (seven keys
    // per thread at the criteo batch: the single-workgroup bookkeeping kernels spent most of their time there)
    constexpr int R = 8;
    for (int base = tid0; base < U; base += nthr * R) {
        uint32_t f[R], rk[R], kk[R];
        int us[R], fs[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int u = min(base + i * nthr, U - 1);
            f[i] = flag[u];
            us[i] = c.uslot[u];
            rk[i] = rank[u];
            kk[i] = uniq[u];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const long long idx = ftop - 1 - static_cast<long long>(rk[i]);
            // running out of slots is a sizing error reported by the host wrapper (nmax)
            fs[i] = (f[i] && idx >= 0 && base + i * nthr < U) ? c.free_list[idx] : 0;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int u = base + i * nthr;
            if (u >= U)
                break;
            if (!f[i]) {
                const int s = us[i];
                const unsigned long long st = static_cast<unsigned long long>(clock + u);
                if (c.policy == kLRU) {
                    // lru_cache.cc:27-39: move to the list front
                    c.line[s].stamp = st;
                    const long long pos = ring_at(tail_mod, u - static_cast<long long>(rk[i]), c.Lcap);
                    c.log_slot[pos] = static_cast<uint32_t>(s);
                    c.log_stamp[pos] = st;
                } else if (c.policy == kLFU) {
                    // lfu_cache.cc:22-29,51-68: use+1, front of the next bucket
                    const int fq = c.line[s].freq;
                    if (fq == 1)
                        atomicAdd(reinterpret_cast<unsigned long long *>(&ctl_mut->n_base), ~0ull);
                    c.line[s].freq = fq + 1;
                    c.line[s].stamp = st;
                } else if (c.line[s].state == kResident) {
                    // lfuopt_cache.cc:26-41: use+1 or promotion to the never-evicted store
                    const int fq = c.line[s].freq;
                    if (fq == 0)
                        atomicAdd(reinterpret_cast<unsigned long long *>(&ctl_mut->n_base), ~0ull);
                    if (fq + 1 < kUseCntMax) {
                        c.line[s].freq = fq + 1;
                        c.line[s].stamp = st;
                    } else {
                        c.line[s].state = kStored;
                        atomicAdd(reinterpret_cast<unsigned long long *>(&ctl_mut->n_hash), ~0ull);
                    }
                }
            } else {
                const int s = fs[i];
                c.uslot[u] = s;
                c.line[s].key = kk[i];
                c.line[s].version = -1;
                c.line[s].updates = 0;
                c.hasgrad[s] = 0;
                c.line[s].hg = 0;
                c.line[s].state = static_cast<uint8_t>(miss_state);
            }
        }
    }
}
__global__ __launch_bounds__(256) void cache_assign_kernel(
    const CacheCtl *ctl, CacheCtl *ctl_mut, Cache c, const uint32_t *uniq, const uint32_t *flag,
    const uint32_t *rank, int miss_state) {
    cache_assign_body(ctl, ctl_mut, c, uniq, flag, rank, miss_state, blockIdx.x * 256 + threadIdx.x,
                      gridDim.x * 256);
}

// Makes the touches of cache_assign_kernel permanent: the stamps clock .. clock+U-1 are used, the LRU
// log grew by one entry per hit.
__device__ __forceinline__ void cache_commit_touch_body(CacheCtl *ctl, const Cache &c) {
    if (c.policy == kLRU)
        ctl->log_tail += ctl->nhit;
    ctl->clock += ctl->U;
}
__global__ void cache_commit_touch_kernel(CacheCtl *ctl, Cache c) {
    cache_commit_touch_body(ctl, c);
}
// push_pull: the pull phase's new lines keep their stack entries out of reach of the push phase
__global__ void cache_retire_kernel(CacheCtl *ctl, int retire) {
    if (retire)
        ctl->free_top -= ctl->parked[1];
    else
        ctl->free_top += ctl->parked[1];
}
// embedding_push_pull runs a pull phase and a push phase over two scratch sets; the pull phase's
// counters are parked while the push phase uses the control block.
__global__ void cache_park_kernel(CacheCtl *ctl, int op) {
    if (op == 0) {          // park the pull phase
        ctl->parked[0] = ctl->U;
        ctl->parked[1] = ctl->M;
        ctl->parked[2] = ctl->nhit;
    } else if (op == 1) {   // resume the pull phase
        ctl->U = ctl->parked[0];
        ctl->M = ctl->parked[1];
        ctl->nhit = ctl->parked[2];
    } else if (op == 2) {   // park the push phase's unique count
        ctl->parked[3] = ctl->U;
    } else {                // resume it for the deferred clean-up
        ctl->U = ctl->parked[3];
    }
}

// syncEmbedding: pull rows whose cached version is -1 or lags by more than pull_bound.
// One wave per unique key.
__global__ __launch_bounds__(256) void cache_sync_kernel(CacheCtl *ctl, Cache c,
                                                         const uint32_t *uniq) {
    const int U = static_cast<int>(ctl->U);
    const int lane = lane_id();
    const int wpb = 4;
    for (int u = blockIdx.x * wpb + (threadIdx.x >> 6); u < U; u += gridDim.x * wpb) {
        const int s = c.uslot[u];
        const uint32_t k = uniq[u];
        const long long lk = static_cast<long long>(k) - c.row_start;
        long long sv;
        const float *src;
        if (c.remote) {   // the owner took the decision (ha_store_serve_sync); its answer is in the inbox
            if (!c.inbox_pull[u])
                continue;
            sv = c.inbox_ver[u];
            src = c.inbox_rows + static_cast<long long>(c.inbox_idx[u]) * c.width;
        } else {
            if (lk < 0 || lk >= c.store_rows)
                continue;
            const long long v = c.line[s].version;
            sv = c.srv_ver[lk];
            if (!(v == -1 || sv - v > c.pull_bound))
                continue;
            src = c.table + lk * c.width;
        }
        const bool hg = c.hasgrad[s] != 0;
        float *dst = c.data + static_cast<long long>(s) * c.width;
        const float *g = c.grad + static_cast<long long>(s) * c.width;
        for (long long j = lane; j < c.width; j += kWave) {
            float x = src[j];
            if (hg)
                x = __fadd_rn(x, g[j]);  // Line::addup(): data += grad
            dst[j] = x;
        }
        if (lane == 0) {
            c.line[s].version = sv;
            atomicAdd(reinterpret_cast<unsigned long long *>(&ctl->pulled), 1ull);
        }
    }
}

// dest[i,:] = data[uslot[inverse[i]],:]
__global__ __launch_bounds__(256) void cache_dest_kernel(Cache c, const int32_t *inverse,
                                                         long long n, float *dest) {
    const long long total = n * c.width;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        const long long i = e / c.width, j = e - i * c.width;
        dest[e] = c.data[static_cast<long long>(c.uslot[inverse[i]]) * c.width + j];
    }
}

// LFU / LFUOpt, rare path: the cache is full and its lowest use bucket is empty, so the first insert of
// this batch evicts the oldest line of the lowest non-empty bucket (lfu_cache.cc:31-42,
// lfuopt_cache.cc:48-60): argmin (use, stamp) over the resident, non-stored lines.  Single workgroup;
// exits immediately when the situation does not arise.
constexpr int kScanParts = 512;            // workgroups of the partial scan
constexpr long long kScanWideFrom = 1 << 16;   // caches of at least this many slots use it
__device__ __forceinline__ bool cache_scan_needed(const CacheCtl *ctl, const Cache &c) {
    return c.policy != kLRU && ctl->M > 0 && ctl->n_base == 0 && ctl->size >= c.limit;
}
// `parts`: the partial results of cache_scan_victim_part_kernel, launched between the bookkeeping of this call and the
// kernel this runs in (same condition, same lines -- nothing touches freq / stamp / state in between)
__device__ __forceinline__ void cache_scan_victim_body(CacheCtl *ctl, const Cache &c, bool parts = false) {
    __shared__ unsigned long long s_best[16];
    __shared__ int s_slot[16];
    if (threadIdx.x == 0)
        ctl->scan_victim = -1;
    const bool needed = cache_scan_needed(ctl, c);
    if (!needed)
        return;
    unsigned long long best = ~0ull;
    int slot = -1;
    if (parts) {
        for (int b = threadIdx.x; b < kScanParts; b += 1024) {
            const unsigned long long k = c.scan_key[b];
            if (k < best) {      // (ties cannot happen: stamps are unique)
                best = k;
                slot = c.scan_slot[b];
            }
        }
    } else {
        for (long long s = threadIdx.x; s < c.S; s += 1024) {
            if (c.line[s].state != kResident)
                continue;
            // stamps stay far below 2^48
            const unsigned long long k = (static_cast<unsigned long long>(c.line[s].freq) << 48) | c.line[s].stamp;
            if (k < best) {
                best = k;
                slot = static_cast<int>(s);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_down(best, o, 64);
        const int os = __shfl_down(slot, o, 64);
        if (ob < best) {
            best = ob;
            slot = os;
        }
    }
    if (lane_id() == 0) {
        s_best[threadIdx.x >> 6] = best;
        s_slot[threadIdx.x >> 6] = slot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k)
            if (s_best[k] < s_best[0]) {
                s_best[0] = s_best[k];
                s_slot[0] = s_slot[k];
            }
        ctl->scan_victim = s_slot[0];
    }
}
__global__ __launch_bounds__(1024) void cache_scan_victim_kernel(CacheCtl *ctl, Cache c, int parts) {
    cache_scan_victim_body(ctl, c, parts != 0);
}
// the same argmin, every workgroup over its stripe of the lines: scan_key / scan_slot [kScanParts]
__global__ __launch_bounds__(1024) void cache_scan_victim_part_kernel(const CacheCtl *ctl, Cache c) {
    __shared__ unsigned long long s_best[16];
    __shared__ int s_slot[16];
    if (!cache_scan_needed(ctl, c))
        return;
    unsigned long long best = ~0ull;
    int slot = -1;
    for (long long s = static_cast<long long>(blockIdx.x) * 1024 + threadIdx.x; s < c.S; s += static_cast<long long>(gridDim.x) * 1024) {
        const LineMeta m = c.line[s];
        if (m.state != kResident)
            continue;
        const unsigned long long k = (static_cast<unsigned long long>(m.freq) << 48) | m.stamp;
        if (k < best) {
            best = k;
            slot = static_cast<int>(s);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_down(best, o, 64);
        const int os = __shfl_down(slot, o, 64);
        if (ob < best) {
            best = ob;
            slot = os;
        }
    }
    if (lane_id() == 0) {
        s_best[threadIdx.x >> 6] = best;
        s_slot[threadIdx.x >> 6] = slot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k)
            if (s_best[k] < s_best[0]) {
                s_best[0] = s_best[k];
                s_slot[0] = s_slot[k];
            }
        c.scan_key[blockIdx.x] = s_best[0];
        c.scan_slot[blockIdx.x] = s_slot[0];
    }
}

// batchedInsert of the misses (sorted order) + LRU eviction + log compaction.  Single workgroup.
// lines_done = 1: the bookkeeping kernel already wrote the new lines' records and slot_of (cache_lookup_book_kernel with
// defer_evict == 2), only their log entries are left; 2: those as well (cache_finish_book_kernel) -- nothing to insert.
__device__ __forceinline__ void cache_insert_evict_body(
    CacheCtl *ctl, const Cache &c, const uint32_t *uniq, const uint32_t *flag,
    const uint32_t *rank, int do_insert, int lines_done = 0) {
    __shared__ uint32_t s_w[16];
    __shared__ long long s_head, s_need, s_clean, s_dirty;
    const int U = static_cast<int>(ctl->U);
    const long long M = do_insert ? ctl->M : 0;
    // the touches of this call (stamps clock-U .. clock-1, LRU log entries) were committed by
    // cache_commit_touch_kernel: misses are stamped clock + rank and logged from the current tail
    const long long clock = ctl->clock, tail0 = ctl->log_tail;
    const long long size0 = ctl->size;
    const long long evict_n0 = ctl->evict_n, free_top0 = ctl->free_top;   // (read once: not a trip per round)
    const long long log_head0 = ctl->log_head;
    // Number of evictions and how many of the batch's own first inserts they consume (v_new), see the
    // file header: LRU evicts after an insert while size > limit, LFU/LFUOpt before it while
    // size == limit, and prefer old lines of the lowest bucket, then their own new lines.
    long long need, v_new = 0, drop_all = 0, scan_take = 0;
    if (c.policy == kLRU) {
        need = size0 + M > c.limit ? size0 + M - c.limit : 0;
    } else {
        const long long free0 = c.limit > size0 ? c.limit - size0 : 0;
        long long ev = M > free0 ? M - free0 : 0;
        if (c.policy == kLFUOpt && ev > 0 && free0 == 0 && ctl->n_hash == 0)
            drop_all = 1;  // only the permanent store is left: new lines are dropped (lfuopt_cache.cc:18-24)
        if (drop_all) {
            ev = 0;
            v_new = M;
        } else if (ev > 0 && ctl->n_base == 0 && free0 == 0) {
            scan_take = ctl->scan_victim >= 0 ? 1 : 0;
            v_new = ev - scan_take;
            ev = 0;
        } else {
            const long long from_old = ev < ctl->n_base ? ev : ctl->n_base;
            v_new = ev - from_old;
            ev = from_old;
        }
        need = ev;
    }
    const int base_use = c.policy == kLFU ? 1 : 0;
    const long long tail0_mod = tail0 % c.Lcap;
    if (do_insert && lines_done < 2) {
        constexpr int R = 8;       // batched loads, see cache_assign_body
        for (int base = threadIdx.x; base < U; base += 1024 * R) {
            uint32_t f[R], rk[R], kk[R];
            int us[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int u = min(base + i * 1024, U - 1);
                f[i] = flag[u];
                us[i] = c.uslot[u];
                rk[i] = rank[u];
                kk[i] = uniq[u];
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (base + i * 1024 >= U || !f[i])
                    continue;
                const int s = us[i];
                const long long q = rk[i];
                if (q < v_new) {
                    // inserted and evicted again inside this batch (or dropped): never becomes resident;
                    // a fresh line has updates == 0, so it does not enter evict_
                    c.line[s].state = kFree;
                    continue;
                }
                const unsigned long long st = static_cast<unsigned long long>(clock + q);
                if (!lines_done) {
                    c.slot_of[kk[i]] = s;
                    c.line[s].stamp = st;
                    c.line[s].freq = base_use;
                    c.line[s].state = kResident;
                }
                const long long pos = ring_at(tail0_mod, q - v_new, c.Lcap);
                c.log_slot[pos] = static_cast<uint32_t>(s);
                c.log_stamp[pos] = st;
            }
        }
    }
    __syncthreads();
    CACHE_PH(9);
    const long long inserted = M - v_new;
    const long long tail = tail0 + inserted;
    long long size = size0 + inserted;
    if (threadIdx.x == 0) {
        s_head = log_head0;
        s_need = need;
        s_clean = 0;
        s_dirty = 0;
        if (scan_take) {
            // the one victim outside the lowest bucket
            const int vs = static_cast<int>(ctl->scan_victim);
            c.slot_of[c.line[vs].key] = -1;
            if (c.line[vs].updates != 0) {
                c.line[vs].state = kEvictedDirty;
                c.evict_slots[ctl->evict_n] = vs;
                s_dirty = 1;
            } else {
                c.line[vs].state = kFree;
                c.free_list[ctl->free_top - M] = vs;
                s_clean = 1;
            }
        }
    }
    __syncthreads();
    const long long E = need + scan_take;
    // ---- evict the `need` oldest valid log entries
    if (threadIdx.x == 0) {
        ctl->ph[13] = 0;                                            // rounds
        ctl->ph[14] = static_cast<unsigned long long>(need);        // lines to evict
        ctl->ph[15] = static_cast<unsigned long long>(s_head);      // log head before the walk
    }
    // Two log entries per thread and round (2,048 entries: a Criteo batch's ~1,000 victims in ONE round instead of 1.3), and
    // ONE packed scan per round -- valid entries in the low half, valid-and-dirty ones in the high half: every valid entry in
    // front of a victim is a victim too, so a victim's rank among the dirty (clean) victims is its rank among the valid-dirty
    // (valid-clean) entries.
    __shared__ uint32_t s_cut_dirty;
    while (true) {
        const long long head = s_head, left = s_need;
        if (left <= 0 || head >= tail)
            break;
        if (threadIdx.x == 0) {
            ctl->ph[13] += 1;
            s_cut_dirty = 0xFFFFFFFFu;
        }
        const long long pos0 = head + 2ll * threadIdx.x;
        int sl[2] = {-1, -1};
        bool valid[2] = {false, false};
        int upd_s[2] = {0, 0};
        uint32_t key_s[2] = {0, 0};
        {
            // everything a victim needs in the same trip as the validity test, both entries' loads in one batch
            uint8_t st8[2] = {0, 0};
            unsigned long long stp[2] = {0, 0}, lst[2] = {0, 0};
            uint32_t fq[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (pos0 + i < tail) {
                    sl[i] = static_cast<int>(c.log_slot[(pos0 + i) % c.Lcap]);
                    lst[i] = c.log_stamp[(pos0 + i) % c.Lcap];
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (sl[i] >= 0) {
                    st8[i] = c.line[sl[i]].state;
                    stp[i] = c.line[sl[i]].stamp;
                    upd_s[i] = c.line[sl[i]].updates;
                    key_s[i] = c.line[sl[i]].key;
                    fq[i] = static_cast<uint32_t>(c.line[sl[i]].freq);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                valid[i] = sl[i] >= 0 && st8[i] == kResident && stp[i] == lst[i] &&
                           (c.policy == kLRU || fq[i] == static_cast<uint32_t>(base_use));
        }
        const uint32_t vd0 = valid[0] && upd_s[0] != 0 ? 1u : 0u, vd1 = valid[1] && upd_s[1] != 0 ? 1u : 0u;
        const uint32_t v0 = valid[0] ? 1u : 0u, v1 = valid[1] ? 1u : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan_1024((v0 + v1) | ((vd0 + vd1) << 16), s_w, &tot);
        const uint32_t tot_valid = tot & 0xFFFFu, tot_vd = tot >> 16;
        const uint32_t r[2] = {ex & 0xFFFFu, (ex & 0xFFFFu) + v0}, rd[2] = {ex >> 16, (ex >> 16) + vd0};
        bool take[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            take[i] = valid[i] && static_cast<long long>(r[i]) < left;
            const bool dirty = (i == 0 ? vd0 : vd1) != 0u;
            if (take[i]) {
                c.slot_of[key_s[i]] = -1;
                if (dirty) {
                    c.line[sl[i]].state = kEvictedDirty;
                    c.evict_slots[evict_n0 + s_dirty + rd[i]] = sl[i];
                } else {
                    c.line[sl[i]].state = kFree;
                    // freed slots go on top of the stack AFTER this call's allocations are retired
                    c.free_list[free_top0 - M + s_clean + (r[i] - rd[i])] = sl[i];
                }
                if (static_cast<long long>(r[i]) == left - 1)       // the last victim of the walk: the dirty ones up to it
                    s_cut_dirty = rd[i] + (dirty ? 1u : 0u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const long long taken = static_cast<long long>(tot_valid) < left ? tot_valid : left;
            const uint32_t taken_dirty = static_cast<long long>(tot_valid) <= left && s_cut_dirty == 0xFFFFFFFFu ? tot_vd
                                         : (s_cut_dirty != 0xFFFFFFFFu ? s_cut_dirty : tot_vd);
            s_need = left - taken;
            s_dirty += taken_dirty;
            s_clean += static_cast<uint32_t>(taken) - taken_dirty;
            s_head = head + 2048 < tail ? head + 2048 : tail;
        }
        __syncthreads();
        // entries past the last victim of a chunk that satisfied `need` are skipped too; they are
        // either stale or still valid-and-resident -- the latter must stay reachable, so rewind
        if (s_need == 0) {
            // find the position right after the last taken entry
            __shared__ long long s_last;
            if (threadIdx.x == 0)
                s_last = head;
            __syncthreads();
            // one LDS atomic per wave (its last victim), not one per victim on the same word
            const long long mine = take[1] ? pos0 + 2 : (take[0] ? pos0 + 1 : 0);
            const unsigned long long tm = __ballot(mine != 0);
            if (tm != 0ull && lane_id() == 63 - __builtin_clzll(tm))
                atomicMax(reinterpret_cast<unsigned long long *>(&s_last), static_cast<unsigned long long>(mine));
            __syncthreads();
            if (threadIdx.x == 0)
                s_head = s_last;
            __syncthreads();
        }
    }
    __syncthreads();
    CACHE_PH(10);
    long long head = s_head;
    if (threadIdx.x == 0)
        ctl->ph[15] = static_cast<unsigned long long>(head) - ctl->ph[15];       // entries the walk consumed
    long long new_tail = tail;
    // ---- compact the log in place when it is nearly full (valid entries keep their order)
    if (tail - head > c.Lcap - 4 * c.nmax - 2048) {
        __shared__ long long s_wr;
        if (threadIdx.x == 0)
            s_wr = head;
        __syncthreads();
        for (long long base = head; base < tail; base += 1024) {
            const long long pos = base + threadIdx.x;
            int s = -1;
            unsigned long long st = 0;
            bool valid = false;
            if (pos < tail) {
                s = static_cast<int>(c.log_slot[pos % c.Lcap]);
                st = c.log_stamp[pos % c.Lcap];
                valid = c.line[s].state == kResident && c.line[s].stamp == st &&
                        (c.policy == kLRU || c.line[s].freq == base_use);
            }
            uint32_t tot;
            const uint32_t r = block_scan_1024(valid ? 1u : 0u, s_w, &tot);
            const long long wr = s_wr;
            __syncthreads();
            if (valid) {
                c.log_slot[(wr + r) % c.Lcap] = static_cast<uint32_t>(s);
                c.log_stamp[(wr + r) % c.Lcap] = st;
            }
            __syncthreads();
            if (threadIdx.x == 0)
                s_wr = wr + tot;
            __syncthreads();
        }
        new_tail = s_wr;
    }
    // the slots of the batch's dropped inserts go back on the stack right above the freed victims
    if (do_insert && v_new > 0) {
        const long long base = ctl->free_top - M + s_clean;
        for (int u = threadIdx.x; u < U; u += 1024)
            if (flag[u] && static_cast<long long>(rank[u]) < v_new)
                c.free_list[base + rank[u]] = c.uslot[u];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long evicted = E - s_need;
        ctl->log_head = head;
        ctl->log_tail = new_tail;
        // slots of the batch's dropped inserts return to the stack as well
        ctl->free_top = ctl->free_top - M + s_clean + v_new;
        ctl->evict_n += s_dirty;
        ctl->size = size - evicted;
        ctl->clock = clock + M;
        ctl->E = evicted;
        if (c.policy != kLRU) {
            ctl->n_base += inserted - (evicted - scan_take);
            ctl->n_hash += inserted - evicted;
        }
    }
}
__global__ __launch_bounds__(1024) void cache_insert_evict_kernel(
    CacheCtl *ctl, Cache c, const uint32_t *uniq, const uint32_t *flag,
    const uint32_t *rank, int do_insert) {
    cache_insert_evict_body(ctl, c, uniq, flag, rank, do_insert);
}

// after a lookup: report (type 0)
__device__ __forceinline__ void cache_report_pull_body(CacheCtl *ctl, const Cache &c, long long n) {
    ctl->perf[0] = 0;
    ctl->perf[1] = n;
    ctl->perf[2] = ctl->U;
    ctl->perf[3] = ctl->M;
    ctl->perf[4] = ctl->pulled;
    ctl->perf[5] = 0;
    ctl->perf[6] = ctl->size == c.limit;
    ctl->perf[7] = 0;
    ctl->pulled = 0;
    for (int i = 0; i < 64; ++i)          // cache_finish_book_kernel's exchange words: every lookup starts from zero
        ctl->fb_xw[i] = 0;
    ctl->snap[0] = ctl->clock;
    ctl->snap[1] = ctl->log_tail;
    ctl->snap[2] = ctl->free_top;
    ctl->snap[3] = ctl->evict_n;
}
__global__ void cache_report_pull_kernel(CacheCtl *ctl, Cache c, long long n) {
    cache_report_pull_body(ctl, c, n);
}

// plan finish of the cache's batch with the probe (and the pull decision of a lookup) done by the thread
// that writes each unique key (HeadProbe, plan_dev.h)
__global__ __launch_bounds__(1024) void cache_finish_probe_kernel(
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm, int n,
    PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq, int32_t *__restrict__ seg,
    int32_t *__restrict__ counts, int32_t *__restrict__ inverse, int32_t *__restrict__ upos,
    HeadProbe hp) {
    __shared__ uint32_t s_w[kFinishLdsWords];
    finish_block_body(sorted, perm, n, hdr, uniq, seg, counts, inverse, upos, blockIdx.x, s_w, nullptr, &hp);
}

// ---- LRU lookup, local store, not bypassed: the finish of the index plan IS the bookkeeping ------------------------------
// cache_finish_probe_kernel + cache_lookup_book_kernel in one launch.  The finish's workgroups (one per 1024 sorted
// positions, at most 36: all resident) hold, per run head, the unique index and the key; they probe the direct map and take
// the pull decision as before, count their misses and pulls, and EXCHANGE the counts: every workgroup publishes one word
// (valid | heads | pulls | misses, a device-coherent store) and waits for the words of all the others -- the same
// barrier-inside-a-launch the one-launch radix passes use (plan.hip).  With the misses of the earlier chunks known, the
// rank of every miss is known, and each workgroup does the slot assignment / LRU touch / new-line records of its OWN keys
// (cache_assign_body's work, spread over the finish's compute units instead of one); the last chunk's workgroup knows the
// totals and commits the control block.  What is left of the bookkeeping kernel -- nothing: victim scan, insert, eviction
// and report already run beside the row copies.  The report at the end of every lookup zeroes the exchange words, so the
// words of one lookup never satisfy the next (and a captured lookup replays).
__global__ __launch_bounds__(1024) void cache_finish_book_kernel(
    const uint32_t *__restrict__ sorted, const int32_t *__restrict__ perm, int n,
    PlanHeader *__restrict__ hdr, uint32_t *__restrict__ uniq, int32_t *__restrict__ seg,
    int32_t *__restrict__ counts, int32_t *__restrict__ inverse, int32_t *__restrict__ upos, Cache c, int pre_insert) {
    __shared__ uint32_t s_w[kFinishLdsWords];
    __shared__ uint32_t s_sc[16];
    __shared__ uint32_t s_tot[4];       // misses / pulls of the chunks before this one, of all chunks; heads of all chunks
    CacheCtl *ctl = c.ctl;
    const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
    // the control block as the previous call left it (committed below, behind the exchange: by then every workgroup has read it)
    const long long clock = ctl->clock, tail = ctl->log_tail, ftop = ctl->free_top;
    if (b == nb - 1)
        CACHE_PH(0);
    HeadOut ho{false, 0, 0u};
    finish_block_body(sorted, perm, n, hdr, uniq, seg, counts, inverse, upos, b, s_w, nullptr, nullptr, nullptr, 0, nullptr, 0, &ho);
    // probe + syncEmbedding's pull decision (head_probe, plan_dev.h), branch-free
    const uint32_t k = ho.key;
    const bool known = ho.head && k < static_cast<unsigned long long>(c.length);
    const int sv = c.slot_of[known ? k : 0u];
    const int s = known ? sv : -1;
    const long long lk = static_cast<long long>(k) - c.row_start;
    const bool inr = ho.head && lk >= 0 && lk < c.store_rows;
    const long long ver = c.line[s >= 0 ? s : 0].version;
    const long long srv = c.srv_ver[inr ? lk : 0];
    const bool miss = ho.head && s < 0;
    const bool pull = inr && (s < 0 || ver == -1 || srv - ver > c.pull_bound);
    unsigned long long tot4;
    __shared__ unsigned long long s_w64[16];
    const unsigned long long ex4 = block_scan_1024_x4((miss ? 1ull : 0ull) | (pull ? 1ull << 16 : 0ull) | (ho.head ? 1ull << 32 : 0ull),
                                                      s_w64, &tot4);
    // the exchange word: valid | heads | pulls | misses of this chunk (the report at the end of every lookup zeroes the words)
    if (tid == 0)
        __hip_atomic_store(&ctl->fb_xw[b], (1ull << 63) | tot4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        uint32_t mb = 0, ma = 0, pa = 0, ha = 0;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int cb = c0 + tid;
            unsigned long long w = 0;
            if (cb < nb) {
                // every finish workgroup is resident (at most 36), so every word arrives; the programming model does not
                // promise that: after about a second of polling the sticky flag is set (ha_cache_perf / ha_cache_state
                // then fail) and the launch goes on instead of hanging the device
                unsigned spins = 0;
                do {
                    w = __hip_atomic_load(&ctl->fb_xw[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (++spins > (1u << 20)) {
                        ctl->fb_timeout = 1;
                        break;
                    }
                } while (!(w >> 63));
            }
            const uint32_t m = static_cast<uint32_t>(w & 0xFFFFull);
            mb += cb < b ? m : 0u;
            ma += m;
            pa += static_cast<uint32_t>((w >> 16) & 0xFFFFull);
            ha += static_cast<uint32_t>((w >> 32) & 0xFFFFull);
        }
        for (int o = 32; o >= 1; o >>= 1) {
            mb += __shfl_xor(mb, o, 64);
            ma += __shfl_xor(ma, o, 64);
            pa += __shfl_xor(pa, o, 64);
            ha += __shfl_xor(ha, o, 64);
        }
        if (tid == 0) {
            s_tot[0] = mb;
            s_tot[1] = ma;
            s_tot[2] = pa;
            s_tot[3] = ha;
        }
    }
    __syncthreads();
    if (b == nb - 1)
        CACHE_PH(1);
    const uint32_t rank = s_tot[0] + static_cast<uint32_t>(ex4 & 0xFFFFull);
    const long long U = s_tot[3], M = s_tot[1];
    const long long fidx = ftop - 1 - static_cast<long long>(rank);
    // running out of slots is a sizing error reported by the host wrapper (nmax)
    const int fs = (miss && fidx >= 0) ? c.free_list[fidx] : 0;
    if (ho.head) {
        const int u = ho.ui;
        c.flag[u] = miss ? 1u : 0u;
        c.rank[u] = rank;
        c.data_row[u] = pull ? 1 : 0;
        if (!miss) {                    // lru_cache.cc:27-39: move to the list front
            const unsigned long long st = static_cast<unsigned long long>(clock + u);
            const long long pos = ring_at(tail % c.Lcap, u - static_cast<long long>(rank), c.Lcap);
            c.uslot[u] = s;
            c.line[s].stamp = st;
            c.log_slot[pos] = static_cast<uint32_t>(s);
            c.log_stamp[pos] = st;
        } else {
            c.uslot[u] = fs;
            c.line[fs].key = k;
            c.line[fs].version = -1;
            c.line[fs].updates = 0;
            c.hasgrad[fs] = 0;
            c.line[fs].hg = 0;
            if (pre_insert) {
                // the new line's side of batchedInsert as well (cache_insert_evict_body then only appends the log
                // entries): LRU with limit >= batch inserts every miss, stamp = clock after the touches + its rank
                const unsigned long long st = static_cast<unsigned long long>(clock + U + rank);
                c.line[fs].stamp = st;
                c.line[fs].freq = 0;
                c.line[fs].state = kResident;
                c.slot_of[k] = fs;
                // ... and its log entry, behind the U - M touches of this lookup (cache_insert_evict_body's tail0 + rank)
                const long long pos = ring_at((tail + (U - M)) % c.Lcap, rank, c.Lcap);
                c.log_slot[pos] = static_cast<uint32_t>(fs);
                c.log_stamp[pos] = st;
            } else {
                c.line[fs].state = static_cast<uint8_t>(kPending);
            }
        }
    }
    if (b == nb - 1) {      // commit: cache_scan_body's totals, cache_commit_touch_body, the pull count
        if (tid == 0) {
            ctl->U = U;
            ctl->M = M;
            ctl->nhit = U - M;
            ctl->log_tail = tail + (U - M);
            ctl->clock = clock + U;
            ctl->pulled = s_tot[2];
        }
        CACHE_PH(4);
    }
}

// ---- fused lookup: one bookkeeping workgroup + one row kernel ---------------------------------------
// ha_cache_lookup = plan (2 launches) + cache_lookup_book_kernel + cache_lookup_rows_kernel.  The
// bookkeeping phases (probe, miss scan, slot assignment / LRU touch, pull decision, victim scan,
// insert + evict, report) touch a few KB per phase and depend on each other through grid-wide
// results, so they run as phases of ONE 1024-thread workgroup separated by workgroup barriers
// instead of as eight dependent launches.  The pull decision of syncEmbedding (cache.cc:84-93:
// version -1 or lagging by more than pull_bound) is taken here, once per unique key, and parked in
// data_row[u]; the row kernel only moves rows.
// defer_evict: the kernel ends with the pull decisions; victim scan, insert + evict and the report run as workgroup 0
// of cache_lookup_rows_kernel, beside the row copies (which need slots and pull decisions only).
__global__ __launch_bounds__(1024) void cache_lookup_book_kernel(
    Cache c, const PlanHeader *hdr, const uint32_t *uniq, long long n, int bypass, int probed, int defer_evict) {
    __shared__ uint32_t s_cnt[16];
    CacheCtl *ctl = c.ctl;
    const int tid = threadIdx.x;
    CACHE_PH(0);
    // probed: uslot / flag / the pull decisions (data_row) were written by cache_finish_probe_kernel
    if (!probed)
        cache_probe_body(ctl, hdr, uniq, c.slot_of, c.length, bypass, c.uslot, c.flag, tid, 1024);
    else if (tid == 0)
        ctl->U = hdr->n_unique;
    __syncthreads();
    CACHE_PH(1);
    const int U = static_cast<int>(hdr->n_unique);
    if (probed && c.policy == kLRU && U <= 8 * 1024 && (HA_CACHE_BOOK_FUSED)) {
        // Up to 8192 unique keys, flags and pull decisions in place, LRU: thread t owns the keys t, t + 1024, ... from
        // the loads to the stores -- flags / slots / keys / decisions in ONE batch, the miss ranks from one packed scan
        // (per 1024-key slice a 16-bit counter), the free slots of the misses in a second batch; nothing passes through
        // memory between what used to be three phases with a trip each at their start.
        const long long clock = ctl->clock, tail = ctl->log_tail, ftop = ctl->free_top;
        const long long tail_mod = tail % c.Lcap;
        uint32_t f[8], kk[8];
        int us[8], fs[8];
        uint32_t pc = 0;
        unsigned long long v_lo = 0, v_hi = 0;      // misses of slices 0..3 / 4..7, 16 bits each
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = min(tid + i * 1024, max(U - 1, 0));
            const bool on = tid + i * 1024 < U;
            f[i] = c.flag[u];
            us[i] = c.uslot[u];
            kk[i] = uniq[u];
            const int pl = c.data_row[u];
            f[i] = on ? f[i] : 0u;
            pc += on ? static_cast<uint32_t>(pl) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v_lo |= static_cast<unsigned long long>(f[i]) << (16 * i);
            v_hi |= static_cast<unsigned long long>(f[i + 4]) << (16 * i);
        }
        __shared__ unsigned long long s_w64[16];
        unsigned long long t_lo, t_hi;
        const unsigned long long e_lo = block_scan_1024_x4(v_lo, s_w64, &t_lo);
        const unsigned long long e_hi = block_scan_1024_x4(v_hi, s_w64, &t_hi);
        uint32_t ptot;
        block_scan_1024(pc, s_cnt, &ptot);
        uint32_t rk[8];
        uint32_t before = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned long long e = i < 4 ? e_lo : e_hi, t = i < 4 ? t_lo : t_hi;
            rk[i] = before + static_cast<uint32_t>((e >> (16 * (i & 3))) & 0xFFFFull);
            before += static_cast<uint32_t>((t >> (16 * (i & 3))) & 0xFFFFull);
        }
        const uint32_t M = before;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long idx = ftop - 1 - static_cast<long long>(rk[i]);
            // running out of slots is a sizing error reported by the host wrapper (nmax)
            fs[i] = (f[i] && idx >= 0) ? c.free_list[idx] : 0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = tid + i * 1024;
            if (u >= U)
                break;
            c.rank[u] = rk[i];
            if (!f[i]) {                            // the LRU hit branch of cache_assign_body
                const int sl = us[i];
                const unsigned long long st = static_cast<unsigned long long>(clock + u);
                const long long pos = ring_at(tail_mod, u - static_cast<long long>(rk[i]), c.Lcap);
                c.line[sl].stamp = st;
                c.log_slot[pos] = static_cast<uint32_t>(sl);
                c.log_stamp[pos] = st;
            } else {
                const int sl = fs[i];
                c.uslot[u] = sl;
                c.line[sl].key = kk[i];
                c.line[sl].version = -1;
                c.line[sl].updates = 0;
                c.hasgrad[sl] = 0;
                c.line[sl].hg = 0;
                if (defer_evict == 2) {
                    // the new line's side of batchedInsert as well (the same record; cache_insert_evict_body then only
                    // appends the log entries): LRU with limit >= batch inserts every miss, stamp = clock after the
                    // touches + its rank
                    c.line[sl].stamp = static_cast<unsigned long long>(clock + U + rk[i]);
                    c.line[sl].freq = 0;
                    c.line[sl].state = kResident;
                    c.slot_of[kk[i]] = sl;
                } else {
                    c.line[sl].state = static_cast<uint8_t>(kPending);
                }
            }
        }
        __syncthreads();        // every thread has read the control block
        if (tid == 0) {         // cache_scan_body's totals, cache_commit_touch_body, the pull count
            ctl->M = M;
            ctl->nhit = static_cast<long long>(U) - M;
            ctl->log_tail = tail + (static_cast<long long>(U) - M);
            ctl->clock = clock + U;
            ctl->pulled = ptot;
        }
        CACHE_PH(4);
        if (defer_evict)
            return;
        __syncthreads();
        cache_insert_evict_body(ctl, c, uniq, c.flag, c.rank, bypass ? 0 : 1, 0);
        __syncthreads();
        if (tid == 0)
            cache_report_pull_body(ctl, c, n);
        return;
    }
    cache_scan_body(hdr, c.flag, c.rank, &ctl->M, &ctl->nhit);
    __syncthreads();
    CACHE_PH(2);
    cache_assign_body(ctl, ctl, c, uniq, c.flag, c.rank, static_cast<int>(kPending), tid, 1024);
    __syncthreads();
    CACHE_PH(3);
    if (tid == 0)
        cache_commit_touch_body(ctl, c);
    // pull decision per unique key
    uint32_t cnt = 0;
    if (probed) {   // the decisions are there: count them, eight independent loads per thread and round
        for (int base = tid; base < U; base += 8 * 1024) {
            int pl[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                pl[i] = c.data_row[min(base + i * 1024, U - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                cnt += base + i * 1024 < U ? static_cast<uint32_t>(pl[i]) : 0u;
        }
    }
    for (int u = tid; u < U && !probed; u += 1024) {
        int pull = 0;
        {
            const int s = c.uslot[u];
            const long long lk = static_cast<long long>(uniq[u]) - c.row_start;
            if (lk >= 0 && lk < c.store_rows) {
                const long long v = c.line[s].version;
                pull = (v == -1 || c.srv_ver[lk] - v > c.pull_bound) ? 1 : 0;
            }
            c.data_row[u] = pull;
        }
        cnt += pull;
    }
    for (int o = 32; o >= 1; o >>= 1)
        cnt += __shfl_xor(cnt, o, 64);
    if (lane_id() == 0)
        s_cnt[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < 16; ++k)
            tot += s_cnt[k];
        ctl->pulled = tot;
    }
    CACHE_PH(4);
    if (defer_evict)
        return;
    if (c.policy != kLRU) {
        __syncthreads();
        cache_scan_victim_body(ctl, c);
    }
    __syncthreads();
    cache_insert_evict_body(ctl, c, uniq, c.flag, c.rank, bypass ? 0 : 1);
    __syncthreads();
    if (tid == 0)
        cache_report_pull_body(ctl, c, n);
}

// (Tried: finish + probe + pull decisions + slot assignment of a lookup of <= 8192 keys as ONE workgroup whose threads keep
// their keys in registers from the sorted list to the stores -- 38 us instead of 6.8 + 8.6: the ~5,000 probes of slot_of /
// version / srv_ver are random accesses into 135 / 27 / 270 MB and one compute unit's address translation serves them one
// after the other.  Random accesses belong on many compute units -- the finish's probe hook, the row kernel --, the single
// bookkeeping workgroup keeps to scans and sequential lists.)
// One wave per SORTED position p of the batch: dest[perm[p],:] = the row of its key.  Keys marked for
// a pull read the store row (+ the line's pending gradient, Line::addup) and the wave of the key's
// first position also refreshes the cache line and its version; the others copy the cached row.
// evict_block: workgroup 0 is the rest of the bookkeeping instead (victim scan, insert of the misses + eviction, report --
// see cache_lookup_book_kernel).  It touches slot_of / stamp / state / log / free stack / evict list / control block,
// the row waves read uslot / data_row / hasgrad and write rows and versions: disjoint, in the reference's order as well
// (cache.cc:95-104: the rows are copied before the policy inserts and evicts).
template <int VEC>
__global__ __launch_bounds__(1024) void cache_lookup_rows_kernel(
    Cache c, const uint32_t *__restrict__ uniq, const int32_t *__restrict__ upos,
    const int32_t *__restrict__ perm, long long n, float *__restrict__ dest, int evict_block, int bypass) {
    if (evict_block && blockIdx.x == 0) {
        CacheCtl *ctl = c.ctl;
        CACHE_PH(8);
        if (c.policy != kLRU) {
            cache_scan_victim_body(ctl, c, c.S >= kScanWideFrom);      // (the partial scan ran in front of this launch)
            __syncthreads();
        }
        cache_insert_evict_body(ctl, c, uniq, c.flag, c.rank, bypass ? 0 : 1, evict_block - 1);
        __syncthreads();
        if (threadIdx.x == 0)
            cache_report_pull_body(ctl, c, n);
        CACHE_PH(12);
        return;
    }
    const int lane = lane_id();
    const long long p = (static_cast<long long>(blockIdx.x) - (evict_block ? 1 : 0)) * 16ll + (threadIdx.x >> 6);
    if (p >= n)
        return;
    const int u = upos[p];
    const bool head = p == 0 || upos[p - 1] != u;
    const int s = c.uslot[u];
    const long long lk = static_cast<long long>(uniq[u]) - c.row_start;
    const bool pull = c.data_row[u] != 0;
    float *line = c.data + static_cast<long long>(s) * c.width;
    float *out = dest + static_cast<long long>(perm[p]) * c.width;
    if (!pull) {
        if (VEC == 4) {
            for (long long j = lane * 4; j < c.width; j += kWave * 4)
                st4(out + j, ld4(line + j));
        } else {
            for (long long j = lane; j < c.width; j += kWave)
                out[j] = line[j];
        }
        return;
    }
    const bool hg = c.hasgrad[s] != 0;
    const float *src = c.remote ? c.inbox_rows + static_cast<long long>(c.inbox_idx[u]) * c.width
                                : c.table + lk * c.width;
    const float *g = c.grad + static_cast<long long>(s) * c.width;
    if (VEC == 4) {
        for (long long j = lane * 4; j < c.width; j += kWave * 4) {
            float4v x = ld4(src + j);
            if (hg) {
                const float4v gv = ld4(g + j);
                x = float4v{__fadd_rn(x[0], gv[0]), __fadd_rn(x[1], gv[1]), __fadd_rn(x[2], gv[2]),
                            __fadd_rn(x[3], gv[3])};
            }
            st4(out + j, x);
            if (head)
                st4(line + j, x);
        }
    } else {
        for (long long j = lane; j < c.width; j += kWave) {
            float x = src[j];
            if (hg)
                x = __fadd_rn(x, g[j]);  // Line::addup(): data += grad
            out[j] = x;
            if (head)
                line[j] = x;
        }
    }
    if (head && lane == 0)
        c.line[s].version = c.remote ? c.inbox_ver[u] : c.srv_ver[lk];
}

// ---- update ----------------------------------------------------------------------------------------
// rows for the two accumulate passes: grad row = slot (every line), data row = slot for lines with data
__device__ __forceinline__ void cache_update_rows_body(const CacheCtl *ctl, const Cache &c, int tid0, int nthr) {
    const int U = static_cast<int>(ctl->U);
    for (int u = tid0; u < U; u += nthr) {
        const int s = c.uslot[u];
        c.data_row[u] = c.line[s].state == kTransient ? -1 : s;
    }
}
__global__ __launch_bounds__(256) void cache_update_rows_kernel(const CacheCtl *ctl, Cache c) {
    cache_update_rows_body(ctl, c, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// fused update, phase 1: probe, miss scan, slot assignment (misses become transient lines without
// data, cache.cc:146-150) / LRU touch, accumulate-row maps -- one workgroup, see cache_lookup_book_kernel
__global__ __launch_bounds__(1024) void cache_update_book_kernel(
    Cache c, const PlanHeader *hdr, const uint32_t *uniq, int bypass, int probed) {
    CacheCtl *ctl = c.ctl;
    const int tid = threadIdx.x;
    if (!probed)
        cache_probe_body(ctl, hdr, uniq, c.slot_of, c.length, bypass, c.uslot, c.flag, tid, 1024);
    else if (tid == 0)
        ctl->U = hdr->n_unique;
    __syncthreads();
    cache_scan_body(hdr, c.flag, c.rank, &ctl->M, &ctl->nhit);
    __syncthreads();
    cache_assign_body(ctl, ctl, c, uniq, c.flag, c.rank, static_cast<int>(kTransient), tid, 1024);
    __syncthreads();
    if (tid == 0)
        cache_commit_touch_body(ctl, c);
    cache_update_rows_body(ctl, c, tid, 1024);
}

// bookkeeping after the accumulate: updates += count, push decision
__global__ __launch_bounds__(256) void cache_update_flags_kernel(
    const CacheCtl *ctl, Cache c, const uint32_t *uniq, const int32_t *counts,
    const uint32_t *push_keys, long long n_push_keys, int with_push_keys) {
    const int U = static_cast<int>(ctl->U);
    for (int u = blockIdx.x * 256 + threadIdx.x; u < U; u += gridDim.x * 256) {
        const int s = c.uslot[u];
        c.hasgrad[s] = 1;
        c.line[s].hg = 1;
        const int upd = c.line[s].updates + counts[u];
        c.line[s].updates = upd;
        const bool has_data = c.line[s].state != kTransient;
        bool push;
        if (!with_push_keys) {
            push = upd > c.push_bound || !has_data;  // cache.cc:159
        } else {
            // lines whose key is listed in the (sorted) push keys and that hold data (cache.cc:295-299)
            const uint32_t k = uniq[u];
            long long lo = 0, hi = n_push_keys;
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (push_keys[mid] < k)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            push = has_data && lo < n_push_keys && push_keys[lo] == k;
        }
        c.pushflag[u] = push ? 1 : 0;
        c.flag[u] = push ? 1u : 0u;
    }
}

// server side of pushEmbedding for the batch's own lines (unique keys: no conflicts):
//   ver[row] += updates;  row += grad      (PSFhandle_embedding.cc:23-27).  One wave per unique key.
__global__ __launch_bounds__(256) void cache_push_lines_kernel(const CacheCtl *ctl, Cache c,
                                                               const uint32_t *uniq) {
    const int U = static_cast<int>(ctl->U);
    const int lane = lane_id();
    for (int u = blockIdx.x * 4 + (threadIdx.x >> 6); u < U; u += gridDim.x * 4) {
        if (!c.pushflag[u])
            continue;
        const int s = c.uslot[u];
        const long long lk = static_cast<long long>(uniq[u]) - c.row_start;
        if (lk < 0 || lk >= c.store_rows)
            continue;
        float *row = c.table + lk * c.width;
        const float *g = c.grad + static_cast<long long>(s) * c.width;
        for (long long j = lane; j < c.width; j += kWave)
            row[j] = __fadd_rn(row[j], g[j]);
        if (lane == 0)
            c.srv_ver[lk] += c.line[s].updates;
    }
}

// fused update, phase 3: cache_update_flags_kernel + cache_push_lines_kernel, one wave per unique key
// (the push decision of a line and the server side of its push need nothing from other lines)
__global__ __launch_bounds__(256) void cache_update_flags_push_kernel(
    const CacheCtl *ctl, Cache c, const uint32_t *uniq, const int32_t *counts,
    const uint32_t *push_keys, long long n_push_keys, int with_push_keys) {
    const int U = static_cast<int>(ctl->U);
    const int lane = lane_id();
    for (int u = blockIdx.x * 4 + (threadIdx.x >> 6); u < U; u += gridDim.x * 4) {
        const int s = c.uslot[u];
        const int upd = c.line[s].updates + counts[u];
        const bool has_data = c.line[s].state != kTransient;
        const uint32_t k = uniq[u];
        bool push;
        if (!with_push_keys) {
            push = upd > c.push_bound || !has_data;  // cache.cc:159
        } else {
            long long lo = 0, hi = n_push_keys;      // cache.cc:295-299
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (push_keys[mid] < k)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            push = has_data && lo < n_push_keys && push_keys[lo] == k;
        }
        if (lane == 0) {
            c.hasgrad[s] = 1;
            c.line[s].hg = 1;
            c.line[s].updates = upd;
            c.pushflag[u] = push ? 1 : 0;
            c.flag[u] = push ? 1u : 0u;
        }
        if (c.remote) {   // outbox entry u: the line's gradient row and update count, or "not pushed"
            if (lane == 0) {
                c.out_keys[u] = push ? k : kNoPush;
                c.out_upd[u] = upd;
            }
            if (push) {
                const float *g = c.grad + static_cast<long long>(s) * c.width;
                float *o = c.out_rows + static_cast<long long>(u) * c.width;
                for (long long j = lane; j < c.width; j += kWave)
                    o[j] = g[j];
            }
            continue;
        }
        if (!push)
            continue;
        const long long lk = static_cast<long long>(k) - c.row_start;
        if (lk < 0 || lk >= c.store_rows)
            continue;
        float *row = c.table + lk * c.width;
        const float *g = c.grad + static_cast<long long>(s) * c.width;
        for (long long j = lane; j < c.width; j += kWave)
            row[j] = __fadd_rn(row[j], g[j]);
        if (lane == 0)
            c.srv_ver[lk] += upd;
    }
}

// pending evicted lines, pushed after the batch's lines (same-key order of the reference's merge:
// the current line first, then the older evicted ones in eviction order).  Entry j is applied by the
// wave of the FIRST entry with its key, which then walks the later duplicates in order.
__global__ __launch_bounds__(256) void cache_push_evicted_kernel(const CacheCtl *ctl, Cache c, long long pad_to) {
    const int En = static_cast<int>(ctl->evict_n);
    const int lane = lane_id();
    if (c.remote && blockIdx.x == 0 && threadIdx.x == 0)
        c.ctl->out_n = ctl->U + En <= c.out_cap ? ctl->U + En : -1;
    // entries [U + E, pad_to) are marked "not pushed": an owner that takes the outbox without a host-side count
    // (ha_cache_outbox_pad) serves pad_to entries
    if (c.remote) {
        const long long lim = pad_to < c.out_cap ? pad_to : c.out_cap;
        for (long long e = ctl->U + En + blockIdx.x * 256ll + threadIdx.x; e < lim; e += gridDim.x * 256ll) {
            c.out_keys[e] = kNoPush;
            c.out_upd[e] = 0;
        }
    }
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < En; j += gridDim.x * 4) {
        const int s = c.evict_slots[j];
        const uint32_t k = c.line[s].key;
        if (c.remote) {   // one outbox entry per evicted line, in eviction order behind the batch's lines
            const long long e = ctl->U + j;
            if (e < c.out_cap) {
                if (lane == 0) {
                    c.out_keys[e] = k;
                    c.out_upd[e] = c.line[s].updates;
                }
                const float *g = c.grad + static_cast<long long>(s) * c.width;
                float *o = c.out_rows + e * c.width;
                for (long long q = lane; q < c.width; q += kWave)
                    o[q] = g[q];
            }
            continue;
        }
        bool earlier = false;
        for (int b = 0; b < j && !earlier; b += kWave) {
            const int t = b + lane;
            const bool m = t < j && c.line[c.evict_slots[t]].key == k;
            earlier = __ballot(m) != 0ull;
        }
        if (earlier)
            continue;
        const long long lk = static_cast<long long>(k) - c.row_start;
        if (lk < 0 || lk >= c.store_rows)
            continue;
        float *row = c.table + lk * c.width;
        long long vadd = 0;
        for (int t = j; t < En; ++t) {
            const int st = c.evict_slots[t];
            if (c.line[st].key != k)
                continue;
            const float *g = c.grad + static_cast<long long>(st) * c.width;
            for (long long q = lane; q < c.width; q += kWave)
                row[q] = __fadd_rn(row[q], g[q]);
            vadd += c.line[st].updates;
        }
        if (lane == 0)
            c.srv_ver[lk] += vadd;
    }
}

// ---- update of the keys of the preceding lookup, LRU, local store (ha_cache_update_same_keys' usual case) ----------
// After a lookup whose evict list was empty before it, with limit >= max_batch: every key of the batch is resident
// with data (the batch's lines are the newest of the cache and at most `limit` lines survive an eviction), its slot
// is still in uslot[], and the pending evicted lines are the victims of that one lookup -- distinct keys, none of them
// in the batch.  Nothing of the update then depends on another line: no probe, no miss scan, no slot assignment.  The
// accumulate (ha_apply_mapped2 with uslot as both row maps) is followed by THIS launch, one wave per unique key:
//   touch (stamp clock + u, log entry tail + u: cache_assign_body's hit branch with rank 0), updates += count, the
//   bounded push (cache.cc:159: updates > push_bound) with its server side (ver += updates, row += grad) and its
//   clean-up (version += updates, zeroGrad),
// further waves: one per pending evicted line (server side of its push, slot back on the free stack), and workgroup 0
// commits what cache_commit_touch_body / cache_update_commit_body commit.  Two launches instead of five; results
// identical to the general path (tests/test_gpu_cache.py: LRU traces with same_as_lookup against the model).
// row[0..width) += g[0..width) (and g = 0 when `zero`), by one wave: all loads of a pass issued before its first store
template <int VEC>
__device__ __forceinline__ void cache_row_add(float *__restrict__ row, float *__restrict__ g, long long width, int lane,
                                              bool zero) {
    if (VEC == 4) {
        for (long long j0 = 0; j0 < width; j0 += kWave * 8) {
            const long long j = j0 + lane * 4, j2 = j + kWave * 4;
            const bool a = j < width, b = j2 < width;
            float4v r0{0.f, 0.f, 0.f, 0.f}, r1 = r0, g0 = r0, g1 = r0;
            if (a) {
                r0 = ld4(row + j);
                g0 = ld4(g + j);
            }
            if (b) {
                r1 = ld4(row + j2);
                g1 = ld4(g + j2);
            }
            const float4v z{0.f, 0.f, 0.f, 0.f};
            if (a) {
                st4(row + j, float4v{__fadd_rn(r0[0], g0[0]), __fadd_rn(r0[1], g0[1]), __fadd_rn(r0[2], g0[2]),
                                     __fadd_rn(r0[3], g0[3])});
                if (zero)
                    st4(g + j, z);
            }
            if (b) {
                st4(row + j2, float4v{__fadd_rn(r1[0], g1[0]), __fadd_rn(r1[1], g1[1]), __fadd_rn(r1[2], g1[2]),
                                      __fadd_rn(r1[3], g1[3])});
                if (zero)
                    st4(g + j2, z);
            }
        }
    } else {
        for (long long j = lane; j < width; j += kWave) {
            row[j] = __fadd_rn(row[j], g[j]);
            if (zero)
                g[j] = 0.f;
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(1024) void cache_update_same_post_kernel(
    Cache c, const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq,
    const int32_t *__restrict__ counts, long long n, long long pad_to) {
    CacheCtl *ctl = c.ctl;
    const int lane = lane_id();
    const int U = static_cast<int>(hdr->n_unique);
    // the control block as the lookup left it (cache_report_pull_body's copy): every wave reads the COPY, workgroup 0
    // commits to the block itself -- no wave waits for another, no grid-wide counter (one same-address atomic per
    // workgroup to elect the last one cost 14.5 us for 520 workgroups: device-scope atomics on one address serialise
    // at ~30 ns each; with a __threadfence before it, 47 us)
    const long long clock = ctl->snap[0], tail = ctl->snap[1], ftop = ctl->snap[2];
    const int En = static_cast<int>(ctl->snap[3]);
    const long long tailU_mod = (tail - U) % c.Lcap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->perf[0] = 1;
        ctl->perf[1] = n;
        ctl->perf[2] = U;
        ctl->perf[3] = 0;            // lines not in the cache
        ctl->perf[4] = En;           // + perf[7], the batch's pushed lines (counted by their waves; ha_cache_perf adds)
        ctl->perf[5] = En;
        ctl->perf[6] = ctl->size == c.limit;
        ctl->U = U;
        ctl->M = 0;
        ctl->nhit = U;
        ctl->clock = clock + U;
        ctl->free_top = ftop + En;       // log_tail stays: the touches replace the lookup's entries
        ctl->evict_n = 0;
        if (c.remote)
            ctl->out_n = U + En <= c.out_cap ? U + En : -1;
    }
    // REMOTE store: nothing here touches the store; the pushes go to the OUTBOX as cache_update_flags_push_kernel /
    // cache_push_evicted_kernel leave them -- entry u = batch line u (kNoPush when it is not pushed), entry U + j = pending
    // evicted line j, entries up to pad_to marked not pushed (an owner that takes the outbox without a host-side count)
    if (c.remote && blockIdx.x == 0) {
        const long long lim = pad_to < c.out_cap ? pad_to : c.out_cap;
        for (long long e = static_cast<long long>(U) + En + threadIdx.x; e < lim; e += 1024) {
            c.out_keys[e] = kNoPush;
            c.out_upd[e] = 0;
        }
    }
    for (int item = blockIdx.x * 16 + static_cast<int>(threadIdx.x >> 6); item < U + En; item += gridDim.x * 16) {
        if (item < U) {
            const int u = item;
            const int s = uniform(c.uslot[u]);
            const uint32_t k = uniform(uniq[u]);
            const int upd = uniform(c.line[s].updates + counts[u]);
            const bool push = upd > c.push_bound;
            if (lane == 0) {
                const unsigned long long st = static_cast<unsigned long long>(clock + u);
                // The log's last U entries are the lookup's touches and inserts of exactly these lines: all stale the
                // moment this update touches them.  The new entries REPLACE them (same positions, stamps still ascending)
                // instead of following them: the log does not collect U dead entries per step for the eviction to wade
                // through (the general path appends; the order of the valid entries -- all eviction sees -- is the same).
                const long long pos = ring_at(tailU_mod, u, c.Lcap);
                c.line[s].stamp = st;
                c.log_slot[pos] = static_cast<uint32_t>(s);
                c.log_stamp[pos] = st;
                c.hasgrad[s] = 1;
                c.line[s].hg = 1;
                c.line[s].updates = push ? 0 : upd;
                if (push) {
                    c.line[s].version += upd;
                    atomicAdd(reinterpret_cast<unsigned long long *>(&ctl->perf[7]), 1ull);
                }
            }
            if (c.remote && lane == 0) {
                c.out_keys[u] = push ? k : kNoPush;
                c.out_upd[u] = upd;
            }
            if (!push)
                continue;
            float *g = c.grad + static_cast<long long>(s) * c.width;
            if (c.remote) {         // the gradient row travels, the line's buffer starts from zero again
                float *o = c.out_rows + static_cast<long long>(u) * c.width;
                for (long long j = lane; j < c.width; j += kWave) {
                    o[j] = g[j];
                    g[j] = 0.f;
                }
                continue;
            }
            const long long lk = static_cast<long long>(k) - c.row_start;
            if (lk >= 0 && lk < c.store_rows) {
                cache_row_add<VEC>(c.table + lk * c.width, g, c.width, lane, true);
                if (lane == 0)
                    c.srv_ver[lk] += upd;
            } else {
                for (long long j = lane; j < c.width; j += kWave)
                    g[j] = 0.f;
            }
        } else {
            const int j = item - U;
            const int s = uniform(c.evict_slots[j]);
            if (c.remote) {
                const long long e = static_cast<long long>(U) + j;
                if (e < c.out_cap) {
                    if (lane == 0) {
                        c.out_keys[e] = c.line[s].key;
                        c.out_upd[e] = c.line[s].updates;
                    }
                    const float *g = c.grad + static_cast<long long>(s) * c.width;
                    float *o = c.out_rows + e * c.width;
                    for (long long q = lane; q < c.width; q += kWave)
                        o[q] = g[q];
                }
                if (lane == 0) {
                    c.line[s].state = kFree;
                    c.free_list[ftop + j] = s;
                }
                continue;
            }
            const long long lk = static_cast<long long>(uniform(c.line[s].key)) - c.row_start;
            if (lk >= 0 && lk < c.store_rows) {
                cache_row_add<VEC>(c.table + lk * c.width, c.grad + static_cast<long long>(s) * c.width, c.width, lane,
                                   false);
                if (lane == 0)
                    c.srv_ver[lk] += c.line[s].updates;
            }
            if (lane == 0) {
                c.line[s].state = kFree;
                c.free_list[ftop + j] = s;
            }
        }
    }
}

// after the push: version bump, zeroGrad, transient lines dropped, evicted slots freed.
__device__ __forceinline__ void cache_update_cleanup_body(const CacheCtl *ctl, const Cache &c,
                                                          int with_push_keys, int wave0, int nwaves) {
    const int U = static_cast<int>(ctl->U);
    const int lane = lane_id();
    for (int u = wave0; u < U; u += nwaves) {
        const int s = c.uslot[u];
        const bool has_data = c.line[s].state != kTransient;
        const bool pushed = c.pushflag[u] != 0;
        bool zero;
        if (!with_push_keys) {
            zero = pushed && has_data;                    // cache.cc:171-177
            if (zero && lane == 0)
                c.line[s].version += c.line[s].updates;
        } else {
            if (lane == 0)
                c.line[s].version += c.line[s].updates;             // every touched line (cache.cc:308-310)
            zero = pushed;
        }
        if (zero) {
            float *g = c.grad + static_cast<long long>(s) * c.width;
            for (long long j = lane; j < c.width; j += kWave)
                g[j] = 0.f;
            if (lane == 0)
                c.line[s].updates = 0;
        }
        if (!has_data && lane == 0)
            c.line[s].state = kFree;  // its stack entry was never retired (free_top unchanged)
    }
}
__global__ __launch_bounds__(256) void cache_update_cleanup_kernel(const CacheCtl *ctl, Cache c,
                                                                   int with_push_keys) {
    cache_update_cleanup_body(ctl, c, with_push_keys, blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4);
}

__device__ __forceinline__ void cache_update_commit_body(CacheCtl *ctl, const Cache &c, long long n) {
    __shared__ uint32_t s_w[16];
    const int U = static_cast<int>(ctl->U);
    const long long En = ctl->evict_n;
    const long long ftop = ctl->free_top;
    for (long long j = threadIdx.x; j < En; j += 1024) {
        const int s = c.evict_slots[j];
        c.line[s].state = kFree;
        c.free_list[ftop + j] = s;
    }
    // number of pushed lines of the batch
    uint32_t cnt = 0;
    for (int u = threadIdx.x; u < U; u += 1024)
        cnt += c.pushflag[u];
    uint32_t tot;
    block_scan_1024(cnt, s_w, &tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl->perf[0] = 1;
        ctl->perf[1] = n;
        ctl->perf[2] = U;
        ctl->perf[3] = ctl->M;       // lines not in the cache
        ctl->perf[4] = tot + En;     // num_transfered
        ctl->perf[5] = En;           // num_evict
        ctl->perf[6] = ctl->size == c.limit;
        ctl->perf[7] = 0;
        ctl->free_top = ftop + En;
        ctl->evict_n = 0;
    }
}
__global__ __launch_bounds__(1024) void cache_update_commit_kernel(CacheCtl *ctl, Cache c,
                                                                   long long n) {
    cache_update_commit_body(ctl, c, n);
}
// fused update, last phase: block 0 = commit (evicted slots freed, report), the others = clean-up
// (version bump, zeroGrad, transient lines dropped); the two touch disjoint state
__global__ __launch_bounds__(1024) void cache_update_finish_kernel(CacheCtl *ctl, Cache c, long long n,
                                                                   int with_push_keys) {
    if (blockIdx.x == 0)
        cache_update_commit_body(ctl, c, n);
    else
        cache_update_cleanup_body(ctl, c, with_push_keys,
                                  (blockIdx.x - 1) * 16 + (threadIdx.x >> 6), (gridDim.x - 1) * 16);
}

__global__ __launch_bounds__(256) void cache_f32_to_u32_kernel(const float *in, long long n,
                                                               uint32_t *out) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
        out[i] = f32_to_key(in[i]);
}
__global__ __launch_bounds__(256) void cache_u64_to_u32_kernel(const uint64_t *in, long long n,
                                                               uint32_t *out) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
        out[i] = in[i] > 0xFFFFFFFEull ? 0xFFFFFFFEu : static_cast<uint32_t>(in[i]);
}

__global__ __launch_bounds__(256) void cache_init_kernel(Cache c) {
    const long long tid = blockIdx.x * 256ll + threadIdx.x, stride = gridDim.x * 256ll;
    for (long long i = tid; i < c.length; i += stride)
        c.slot_of[i] = -1;
    for (long long i = tid; i < c.S; i += stride) {
        c.free_list[i] = static_cast<int32_t>(c.S - 1 - i);  // slot 0 is handed out first
        c.line[i].state = kFree;
        c.line[i].updates = 0;
        c.line[i].freq = 0;
        c.hasgrad[i] = 0;
        c.line[i].hg = 0;
        c.line[i].version = -1;
        c.line[i].stamp = 0;
    }
    if (tid == 0) {
        memset(c.ctl, 0, sizeof(CacheCtl));
        c.ctl->free_top = c.S;
        c.ctl->clock = 1;
    }
}

// snapshot of the resident lines (unordered) for tests / keys() / debugging
__global__ __launch_bounds__(256) void cache_snapshot_kernel(Cache c, long long cap, uint32_t *keys,
                                                             long long *version, int32_t *updates,
                                                             unsigned long long *stamp,
                                                             int32_t *slots,
                                                             unsigned long long *count) {
    for (long long s = blockIdx.x * 256ll + threadIdx.x; s < c.S; s += gridDim.x * 256ll) {
        if (c.line[s].state != kResident && c.line[s].state != kStored)
            continue;
        const unsigned long long i = atomicAdd(count, 1ull);
        if (static_cast<long long>(i) < cap) {
            keys[i] = c.line[s].key;
            version[i] = c.line[s].version;
            updates[i] = c.line[s].updates;
            stamp[i] = c.line[s].stamp;
            slots[i] = static_cast<int32_t>(s);
        }
    }
}

__global__ __launch_bounds__(256) void cache_set_line_kernel(Cache c, long long key, long long version,
                                                             const float *data) {
    if (key < 0 || key >= c.length)
        return;
    const int s = c.slot_of[key];
    if (s < 0 || (c.line[s].state != kResident && c.line[s].state != kStored))
        return;
    if (threadIdx.x == 0)
        c.line[s].version = version;
    for (long long j = threadIdx.x; j < c.width; j += 256)
        c.data[static_cast<long long>(s) * c.width + j] = data[j];
}

}  // namespace ha

using namespace ha;

extern "C" ha_cache *ha_cache_create(int policy, int64_t limit, int64_t length,
                                     int64_t width, int64_t max_batch) {
    if (policy < 0 || policy > 2) {
        set_error("ha_cache_create: policy must be 0 (LRU), 1 (LFU) or 2 (LFUOpt)");
        return nullptr;
    }
    if (limit < 0 || length <= 0 || length > 0xFFFFFFFEll || width <= 0 || max_batch <= 0) {
        set_error("ha_cache_create: bad arguments");
        return nullptr;
    }
    ha_cache *h = new ha_cache();
    if (const char *e = getenv("HA_CACHE_FUSED"))
        h->fused_update = atoi(e);
    Cache &c = h->c;
    memset(&c, 0, sizeof(c));
    c.policy = policy;
    c.limit = limit;
    c.length = length;
    c.width = width;
    c.nmax = max_batch;
    c.S = limit + 4 * max_batch + 64;
    c.Lcap = 2 * c.S + 8 * max_batch + 4096;
    c.pull_bound = 5;   // include/cache.h:27-28
    c.push_bound = 5;
    const size_t plan_bytes = ha_plan_bytes(max_batch);
    h->plan_bytes = plan_bytes;
    bool ok = true;
#define CACHE_ALLOC(field, count)                                                   \
    do {                                                                            \
        if (ok && dmalloc(&c.field, static_cast<size_t>(count)) != 0)               \
            ok = false;                                                             \
        else if (ok)                                                                \
            h->allocs.push_back(c.field);                                           \
    } while (0)
    CACHE_ALLOC(ctl, 1);
    CACHE_ALLOC(slot_of, length);
    CACHE_ALLOC(line, c.S);
    CACHE_ALLOC(hasgrad, c.S);
    CACHE_ALLOC(data, c.S * width);
    CACHE_ALLOC(grad, c.S * width);
    CACHE_ALLOC(free_list, c.S);
    CACHE_ALLOC(log_slot, c.Lcap);
    CACHE_ALLOC(log_stamp, c.Lcap);
    CACHE_ALLOC(evict_slots, c.S);
    CACHE_ALLOC(uslot, max_batch);
    CACHE_ALLOC(data_row, max_batch);
    CACHE_ALLOC(flag, max_batch);
    CACHE_ALLOC(rank, max_batch);
    CACHE_ALLOC(pushflag, max_batch);
    CACHE_ALLOC(pushkeys_u32, max_batch);
    CACHE_ALLOC(uslot_b, max_batch);
    CACHE_ALLOC(data_row_b, max_batch);
    CACHE_ALLOC(flag_b, max_batch);
    CACHE_ALLOC(rank_b, max_batch);
    CACHE_ALLOC(pushflag_b, max_batch);
    CACHE_ALLOC(scan_key, kScanParts);
    CACHE_ALLOC(scan_slot, kScanParts);
    if (ok) {
        char *p = nullptr;
        if (dmalloc(&p, plan_bytes) == 0) {
            c.plan_ws = p;
            h->allocs.push_back(p);
        } else {
            ok = false;
        }
        char *p2 = nullptr;
        if (ok && dmalloc(&p2, plan_bytes) == 0) {
            c.plan_ws_b = p2;
            h->allocs.push_back(p2);
        } else {
            ok = false;
        }
        char *p3 = nullptr;
        if (ok && dmalloc(&p3, plan_bytes) == 0) {
            h->plan_ws_alt = p3;
            h->allocs.push_back(p3);
            ok = dzero(p3, 256) == 0;      // the plan header (sticky flags)
        } else {
            ok = false;
        }
    }
#undef CACHE_ALLOC
    if (!ok) {
        for (void *p : h->allocs)
            (void)hipFree(p);
        delete h;
        return nullptr;
    }
    hipLaunchKernelGGL(cache_init_kernel, dim3(2048), dim3(256), 0, nullptr, c);
    if (hipDeviceSynchronize() != hipSuccess) {
        set_error("ha_cache_create: init kernel failed");
        for (void *p : h->allocs)
            (void)hipFree(p);
        delete h;
        return nullptr;
    }
    return h;
}

extern "C" void ha_cache_destroy(ha_cache *h) {
    if (!h)
        return;
    (void)hipDeviceSynchronize();
    if (h->ahead_stream)
        (void)hipStreamDestroy(h->ahead_stream);
    if (h->ahead_fork)
        (void)hipEventDestroy(h->ahead_fork);
    if (h->ahead_join)
        (void)hipEventDestroy(h->ahead_join);
    if (h->plan_fork)
        (void)hipEventDestroy(h->plan_fork);
    for (PlanSlot &sl : h->plan) {
        if (sl.booked)
            (void)hipEventDestroy(sl.booked);
        if (sl.rows_done)
            (void)hipEventDestroy(sl.rows_done);
    }
    for (void *p : h->allocs)
        (void)hipFree(p);
    delete h;
}

extern "C" int ha_cache_set_bounds(ha_cache *h, int64_t pull_bound, int64_t push_bound) {
    HA_REQUIRE(h, "cache: null handle");
    HA_REQUIRE(ha_cache_plan_pending(h) == 0, "cache_set_bounds: planned batches are outstanding (their bookkeeping used the bounds)");
    h->c.pull_bound = pull_bound;
    h->c.push_bound = push_bound;
    return 0;
}

extern "C" int ha_cache_set_bypass(ha_cache *h, int bypass) {
    HA_REQUIRE(h, "cache: null handle");
    HA_REQUIRE(ha_cache_plan_pending(h) == 0, "cache_set_bypass: planned batches are outstanding");
    h->c.bypass = bypass != 0;
    return 0;
}

extern "C" int ha_cache_bind_store(ha_cache *h, float *table, int64_t *versions,
                                   int64_t store_rows, int64_t row_start) {
    HA_REQUIRE(h && table && versions && store_rows >= 0 && row_start >= 0, "cache_bind_store: bad arguments");
    h->c.table = table;
    h->c.srv_ver = reinterpret_cast<long long *>(versions);
    h->c.store_rows = store_rows;
    h->c.row_start = row_start;
    return 0;
}

// key_kind: 0 = float32 ids (the *_raw entry points, cache.cc:49-58), 1 = uint64 keys
// Index plan of the batch.  *probed = 1 when the finish also probed the cache for every unique key
// (batches up to kSmallMax keys: sort, then cache_finish_probe_kernel); want_pull adds the pull decision.
static int cache_plan(ha_cache *h, const void *keys, int key_kind, int64_t n, hipStream_t s,
                      int want_pull = 0, int *probed = nullptr, bool presorted = false) {
    HA_REQUIRE(n <= h->c.nmax, "cache: batch of %ld keys exceeds max_batch %ld", (long)n, (long)h->c.nmax);
    Cache &c = h->c;
    if (probed)
        *probed = 0;
    if (probed && n > 0 && n <= kSmallMax) {
        const uint64_t lim = static_cast<uint64_t>(c.length);
        const int rc = presorted ? 0
                       : key_kind == 0 ? ha_plan_sort_f32ids_lim(static_cast<const float *>(keys), n, c.plan_ws, lim, s)
                                       : ha_plan_sort_u64ids_lim(static_cast<const uint64_t *>(keys), n, c.plan_ws, lim, s);
        if (rc)
            return rc;
        PlanPtrs p = plan_layout(c.plan_ws, n);
        const HeadProbe hp{c.slot_of, (long long)c.length, c.bypass ? 1 : 0, c.uslot, c.flag,
                           want_pull ? c.data_row : nullptr, &c.line[0].version,
                           static_cast<int>(sizeof(LineMeta) / sizeof(long long)), c.srv_ver,
                           (long long)c.row_start, (long long)c.store_rows, (long long)c.pull_bound};
        hipLaunchKernelGGL(cache_finish_probe_kernel, dim3(finish_blocks((int)n)), dim3(1024), 0, s, p.sorted,
                           p.perm, (int)n, p.hdr, p.uniq, p.seg, p.counts, p.inverse, p.upos, hp);
        HA_LAUNCH_CHECK();
        *probed = 1;
        return 0;
    }
    if (key_kind == 0)
        return ha_plan_build_f32ids_lim(static_cast<const float *>(keys), n, c.plan_ws, static_cast<uint64_t>(c.length), s);
    return ha_plan_build_u64ids_lim(static_cast<const uint64_t *>(keys), n, c.plan_ws, static_cast<uint64_t>(c.length), s);
}

// How a lookup's victim scan / insert / eviction / report run: 0 = at the end of cache_lookup_book_kernel, 1 = as workgroup 0
// of cache_lookup_rows_kernel, beside the row copies, 2 = the same with the new lines' records and slot_of entries already
// written by the bookkeeping kernel (its register-resident path: probed plan of <= 8192 keys, LRU; every miss is inserted
// when limit >= batch and the cache is not bypassed).
static int cache_evict_mode(const ha_cache *h, int64_t n, int probed) {
    const Cache &c = h->c;
    if (!(h->fused_update & 2) || n <= 0)
        return 0;
    if (HA_CACHE_BOOK_FUSED && probed && c.policy == kLRU && !c.bypass && c.limit >= n && n <= 8 * 1024)
        return 2;
    return 1;
}

// ---- remote store: inbox / outbox ------------------------------------------------------------------------
// request of a lookup: (key, cached version) of every unique key -- what the reference's client hands to
// syncEmbedding (hetu_client.cc:6-23: keys + the lines' versions)
// Entries [n_unique, n) are padded with a key no store owns (0xFFFFFFFF: never pulled), so that an owner that
// needs no host-side count (a local or host-resident store) can serve all n entries without a read-back.
__global__ __launch_bounds__(256) void cache_export_req_kernel(Cache c, const PlanHeader *hdr,
                                                               const uint32_t *uniq, int n) {
    const int U = static_cast<int>(hdr->n_unique);
    for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
        if (u < U) {
            const int s = c.uslot[u];
            c.req_keys[u] = uniq[u];
            c.req_ver[u] = s >= 0 ? c.line[s].version : -1;
        } else {
            c.req_keys[u] = kNoPush;
            c.req_ver[u] = 0;
        }
    }
}

extern "C" int ha_cache_set_remote(ha_cache *h) {
    HA_REQUIRE(h, "cache_set_remote: null handle");
    Cache &c = h->c;
    if (c.remote)
        return 0;
    c.out_cap = 5 * c.nmax;
    bool ok = true;
#define REMOTE_ALLOC(field, count)                                                  \
    do {                                                                            \
        if (ok && dmalloc(&c.field, static_cast<size_t>(count)) != 0)               \
            ok = false;                                                             \
        else if (ok)                                                                \
            h->allocs.push_back(c.field);                                           \
    } while (0)
    REMOTE_ALLOC(req_keys, c.nmax);
    REMOTE_ALLOC(req_ver, c.nmax);
    REMOTE_ALLOC(inbox_pull, c.nmax);
    REMOTE_ALLOC(inbox_idx, c.nmax);
    REMOTE_ALLOC(inbox_ver, c.nmax);
    REMOTE_ALLOC(inbox_rows, c.nmax * c.width);
    REMOTE_ALLOC(out_keys, c.out_cap);
    REMOTE_ALLOC(out_upd, c.out_cap);
    REMOTE_ALLOC(out_rows, c.out_cap * c.width);
#undef REMOTE_ALLOC
    HA_REQUIRE(ok, "cache_set_remote: out of device memory");
    c.remote = 1;
    return 0;
}

extern "C" int ha_cache_remote_buffers(ha_cache *h, ha_cache_remote *out) {
    HA_REQUIRE(h && out && h->c.remote, "cache_remote_buffers: the cache is not in remote-store mode");
    const Cache &c = h->c;
    out->req_keys = c.req_keys;
    out->req_versions = reinterpret_cast<int64_t *>(c.req_ver);
    out->inbox_pull = c.inbox_pull;
    out->inbox_idx = c.inbox_idx;
    out->inbox_versions = reinterpret_cast<int64_t *>(c.inbox_ver);
    out->inbox_rows = c.inbox_rows;
    out->out_keys = c.out_keys;
    out->out_updates = c.out_upd;
    out->out_rows = c.out_rows;
    out->out_capacity = c.out_cap;
    out->max_batch = c.nmax;
    return 0;
}

// first half of a lookup against a remote store: plan, probe, request export.  *n_unique_host (optional)
// receives the number of unique keys = request entries (synchronises the stream).
extern "C" int ha_cache_lookup_begin(ha_cache *h, const void *keys, int key_kind, int64_t n,
                                     int64_t *n_unique_host, ha_stream_t stream) {
    HA_REQUIRE(h && h->c.remote, "cache_lookup_begin: the cache is not in remote-store mode");
    HA_REQUIRE(n >= 0 && (n == 0 || keys), "cache_lookup_begin: bad arguments");
    Cache &c = h->c;
    hipStream_t s = as_stream(stream);
    int probed = 0;
    cache_mark(h, kTStart, s, true);
    if (cache_plan(h, keys, key_kind, n, s, 0, &probed))
        return -1;
    cache_mark(h, kTSort, s);
    h->plan_n = n;
    // (remote store: the update of these keys can take the two-launch path as well -- its pushes go to the outbox)
    h->same_fast = (h->fused_update & 1) && h->evict_empty && c.policy == kLRU && !c.bypass && c.limit >= n && n > 0;
    h->evict_empty = false;
    PlanPtrs p = plan_layout(c.plan_ws, n);
    if (!probed)   // larger batches: the plan was built unfused, probe separately
        hipLaunchKernelGGL(cache_probe_kernel, CACHE_GRID(n), dim3(256), 0, s, c.ctl, p.hdr, p.uniq, c.slot_of,
                           (long long)c.length, c.bypass ? 1 : 0, c.uslot, c.flag);
    hipLaunchKernelGGL(cache_export_req_kernel, CACHE_GRID(n), dim3(256), 0, s, c, p.hdr, p.uniq, (int)n);
    cache_mark(h, kTLookup, s);
    HA_LAUNCH_CHECK();
    if (n_unique_host) {
        HA_CHECK_HIP(hipMemcpyAsync(n_unique_host, &p.hdr->n_unique, 8, hipMemcpyDeviceToHost, s));
        HA_CHECK_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

// second half: the inbox holds the owner's answer for the request of ha_cache_lookup_begin
extern "C" int ha_cache_lookup_finish(ha_cache *h, int64_t n, float *dest, ha_stream_t stream) {
    HA_REQUIRE(h && h->c.remote, "cache_lookup_finish: the cache is not in remote-store mode");
    HA_REQUIRE(h->plan_n == n && (n == 0 || dest), "cache_lookup_finish: no ha_cache_lookup_begin of %ld keys precedes",
               (long)n);
    Cache c = h->c;
    c.data_row = c.inbox_pull;   // the decisions the bookkeeping and the row kernel read
    hipStream_t s = as_stream(stream);
    PlanPtrs p = plan_layout(c.plan_ws, n);
    const int evb = cache_evict_mode(h, n, 1);
    cache_mark(h, kTTransfer, s);        // between ha_cache_lookup_begin's last launch and here: the exchange with the owners
    hipLaunchKernelGGL(cache_lookup_book_kernel, dim3(1), dim3(1024), 0, s, c, p.hdr, p.uniq,
                       (long long)n, c.bypass ? 1 : 0, 1, evb);
    if (n > 0 && evb && c.policy != kLRU && c.S >= kScanWideFrom)      // LFU / LFUOpt: the victim scan of a large cache, in parallel
        hipLaunchKernelGGL(cache_scan_victim_part_kernel, dim3(kScanParts), dim3(1024), 0, s, c.ctl, c);
    if (n > 0) {
        const unsigned blocks = static_cast<unsigned>((n + 15) / 16) + (evb ? 1 : 0);
        const bool vec_ok = (c.width % 4 == 0) && (reinterpret_cast<uintptr_t>(dest) % 16 == 0);
        if (vec_ok)
            hipLaunchKernelGGL(cache_lookup_rows_kernel<4>, dim3(blocks), dim3(1024), 0, s, c, p.uniq, p.upos,
                               p.perm, (long long)n, dest, evb, c.bypass ? 1 : 0);
        else
            hipLaunchKernelGGL(cache_lookup_rows_kernel<1>, dim3(blocks), dim3(1024), 0, s, c, p.uniq, p.upos,
                               p.perm, (long long)n, dest, evb, c.bypass ? 1 : 0);
    }
    cache_mark(h, kTEnd, s);
    HA_LAUNCH_CHECK();
    return 0;
}

// entries of the outbox the last ha_cache_update* left (synchronises the stream)
extern "C" int ha_cache_outbox_count(ha_cache *h, int64_t *count_host, ha_stream_t stream) {
    HA_REQUIRE(h && h->c.remote && count_host, "cache_outbox_count: the cache is not in remote-store mode");
    long long v = 0;
    HA_CHECK_HIP(hipMemcpyAsync(&v, &h->c.ctl->out_n, 8, hipMemcpyDeviceToHost, as_stream(stream)));
    HA_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    HA_REQUIRE(v >= 0, "cache outbox overflow: more pending evicted lines than the outbox holds (%ld entries)",
               (long)h->c.out_cap);
    *count_host = v;
    return 0;
}

// Remote store without host-side counts: the following updates pad the outbox to `entries` (<= its capacity), so
// the owner can be handed outbox[0, entries) straight away.  The caller guarantees entries >= U + E: the batch
// has at most n unique keys and every lookup since the last update evicted at most as many lines as it has keys.
extern "C" int ha_cache_outbox_pad(ha_cache *h, int64_t entries) {
    HA_REQUIRE(h && h->c.remote, "cache_outbox_pad: the cache is not in remote-store mode");
    HA_REQUIRE(entries >= 0 && entries <= h->c.out_cap, "cache_outbox_pad: %ld entries exceed the outbox (%ld)",
               (long)entries, (long)h->c.out_cap);
    h->out_pad = entries;
    return 0;
}

// The sort of a lookup's keys, one batch early.  The caller's stream forks into the cache's own stream here (everything
// enqueued on `stream` so far -- the calls that last used the second workspace among it -- precedes the sort) and joins it
// again in the ha_cache_lookup_presorted of the same keys; both are event edges, so the pair can be captured into a hipGraph
// as long as the join is captured too.  Batches the counting sort does not take (n > 36,864) are accepted and ignored: their
// lookup sorts by itself.  The keys must not change between this call and their lookup.
extern "C" int ha_cache_sort_ahead(ha_cache *h, const void *keys, int key_kind, int64_t n, ha_stream_t stream) {
    HA_REQUIRE(h && (key_kind == 0 || key_kind == 1) && n >= 0 && (n == 0 || keys), "cache_sort_ahead: bad arguments");
    HA_REQUIRE(n <= h->c.nmax, "cache: batch of %ld keys exceeds max_batch %ld", (long)n, (long)h->c.nmax);
    hipStream_t s = as_stream(stream);
    if (!h->ahead_stream) {
        HA_REQUIRE(hipStreamCreateWithFlags(&h->ahead_stream, hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&h->ahead_fork, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&h->ahead_join, hipEventDisableTiming) == hipSuccess,
                   "cache_sort_ahead: cannot create the stream / events");
    }
    if (h->ahead_n >= 0)      // a sort nobody consumed: the caller's stream joins it before the workspace is reused
        HA_REQUIRE(hipStreamWaitEvent(s, h->ahead_join, 0) == hipSuccess, "cache_sort_ahead: join failed");
    h->ahead_n = -1;
    if (n == 0 || n > kSmallMax)
        return 0;
    HA_REQUIRE(hipEventRecord(h->ahead_fork, s) == hipSuccess &&
               hipStreamWaitEvent(h->ahead_stream, h->ahead_fork, 0) == hipSuccess, "cache_sort_ahead: fork failed");
    const uint64_t lim = static_cast<uint64_t>(h->c.length);
    if (key_kind == 0 ? ha_plan_sort_f32ids_lim(static_cast<const float *>(keys), n, h->plan_ws_alt, lim, h->ahead_stream)
                      : ha_plan_sort_u64ids_lim(static_cast<const uint64_t *>(keys), n, h->plan_ws_alt, lim, h->ahead_stream))
        return -1;
    HA_REQUIRE(hipEventRecord(h->ahead_join, h->ahead_stream) == hipSuccess, "cache_sort_ahead: join record failed");
    h->ahead_keys = keys;
    h->ahead_n = n;
    h->ahead_kind = key_kind;
    return 0;
}

// The sorts of the next `count` (<= 16) lookups at once: ONE launch on the caller's stream (no second stream, no event edge --
// inside a hipGraph of whole steps an edge costs more than the sort it hides), into a ring of plan workspaces.  The
// ha_cache_lookup_presorted calls that follow must name these key buffers IN THIS ORDER (anything else there is an error);
// plain ha_cache_lookup calls in between leave the ring alone.  A caller that has its ids a block of batches early -- the
// work-queue step's requirement, INTEGRATION.md -- uses this; the reference's loader has them one batch early
// (dataloader.py:63-98: ha_cache_sort_ahead).  The keys must not change until their lookups.  A call while sorted batches are
// still unconsumed drops those.
extern "C" int ha_cache_sort_ahead_batch(ha_cache *h, const void *const *keys, int key_kind, const int64_t *n, int count,
                                         ha_stream_t stream) {
    HA_REQUIRE(h && (key_kind == 0 || key_kind == 1) && count >= 0 && count <= ha_cache::kAheadRing &&
               (count == 0 || (keys && n)), "cache_sort_ahead_batch: bad arguments (at most %d batches)", ha_cache::kAheadRing);
    h->ring_count = 0;
    h->ring_head = 0;
    if (count == 0)
        return 0;
    for (int i = 0; i < count; ++i) {
        HA_REQUIRE(n[i] > 0 && n[i] <= kSmallMax && n[i] <= h->c.nmax && keys[i] != nullptr,
                   "cache_sort_ahead_batch: batch %d of %ld keys (1 .. min(max_batch, %d) per batch)", i, (long)n[i], kSmallMax);
        if (h->ring_ws[i] == nullptr) {
            char *p = nullptr;
            HA_REQUIRE(dmalloc(&p, h->plan_bytes) == 0, "cache_sort_ahead_batch: out of device memory");
            HA_REQUIRE(dzero(p, 256) == 0, "cache_sort_ahead_batch: device memset failed");      // the plan header (sticky flags)
            h->ring_ws[i] = p;
            h->allocs.push_back(p);
        }
    }
    const uint64_t lim = static_cast<uint64_t>(h->c.length);
    if (key_kind == 0 ? ha_plan_sort_batch_f32ids_lim(reinterpret_cast<const float *const *>(keys), n, h->ring_ws, count, lim, stream)
                      : ha_plan_sort_batch_u64ids_lim(reinterpret_cast<const uint64_t *const *>(keys), n, h->ring_ws, count, lim,
                                                      stream))
        return -1;
    for (int i = 0; i < count; ++i) {
        h->ring_keys[i] = keys[i];
        h->ring_n[i] = n[i];
    }
    h->ring_kind = key_kind;
    h->ring_count = count;
    return 0;
}

static int cache_lookup_impl(ha_cache *h, const void *keys, int key_kind, int64_t n, float *dest, hipStream_t s, bool presorted,
                             bool from_ring = false);

extern "C" int ha_cache_lookup(ha_cache *h, const void *keys, int key_kind, int64_t n,
                               float *dest, ha_stream_t stream) {
    return cache_lookup_impl(h, keys, key_kind, n, dest, as_stream(stream), false);
}

// The lookup of the keys ha_cache_sort_ahead was last given (same pointer, kind and count -- anything else is an error).
extern "C" int ha_cache_lookup_presorted(ha_cache *h, const void *keys, int key_kind, int64_t n,
                                         float *dest, ha_stream_t stream) {
    HA_REQUIRE(h, "cache_lookup_presorted: null handle");
    if (h->ring_head < h->ring_count) {      // the next batch of ha_cache_sort_ahead_batch's ring
        const int i = h->ring_head;
        HA_REQUIRE(h->ring_keys[i] == keys && h->ring_n[i] == n && h->ring_kind == key_kind,
                   "cache_lookup_presorted: batch %d of the last ha_cache_sort_ahead_batch is another one (%ld keys)", i,
                   (long)h->ring_n[i]);
        HA_REQUIRE(h->c.table && !h->c.remote, "cache_lookup: no local store bound");
        void *ws = h->ring_ws[i];
        h->ring_ws[i] = h->c.plan_ws;        // (all workspaces are of one size: the ring keeps the lookup's old one)
        h->c.plan_ws = ws;
        ++h->ring_head;
        return cache_lookup_impl(h, keys, key_kind, n, dest, as_stream(stream), true, true);
    }
    if (n == 0 || n > kSmallMax)      // sort_ahead ignored this batch
        return cache_lookup_impl(h, keys, key_kind, n, dest, as_stream(stream), false);
    HA_REQUIRE(h->ahead_n == n && h->ahead_keys == keys && h->ahead_kind == key_kind,
               "cache_lookup_presorted: no ha_cache_sort_ahead of these %ld keys precedes", (long)n);
    return cache_lookup_impl(h, keys, key_kind, n, dest, as_stream(stream), true);
}

static int cache_lookup_impl(ha_cache *h, const void *keys, int key_kind, int64_t n, float *dest, hipStream_t s,
                             bool presorted, bool from_ring) {
    HA_REQUIRE(h && h->c.table && !h->c.remote, "cache_lookup: no local store bound (remote stores use "
               "ha_cache_lookup_begin / ha_cache_lookup_finish)");
    HA_REQUIRE(n >= 0 && (n == 0 || (keys && dest)), "cache_lookup: bad arguments");
    HA_REQUIRE(ha_cache_plan_pending(h) == 0, "cache: %d planned calls are outstanding (ha_cache_plan_block): ha_cache_lookup_planned / "
               "ha_cache_update_planned come first", ha_cache_plan_pending(h));
    h->last_planned_type = -1;
    h->lfu_tree_ok = false;
    Cache &c = h->c;
    // (a batch from the ring of ha_cache_sort_ahead_batch has its sorted keys in c.plan_ws already: a pending sort of
    // ha_cache_sort_ahead -- another batch's -- stays where it is)
    if (presorted && !from_ring && h->ahead_n >= 0) {   // the sorted keys are in the second workspace: it becomes the plan of this batch
        HA_REQUIRE(hipStreamWaitEvent(s, h->ahead_join, 0) == hipSuccess, "cache_lookup_presorted: join failed");
        std::swap(c.plan_ws, h->plan_ws_alt);
        h->ahead_n = -1;
    }
    int probed = 0;
    PlanPtrs p = plan_layout(c.plan_ws, n);
    cache_mark(h, kTStart, s, true);
    // LRU, not bypassed, a batch the counting sort takes: sort, then the finish that is the bookkeeping as well
    const bool finish_book = (h->fused_update & 4) && (h->fused_update & 2) && c.policy == kLRU && !c.bypass && n > 0 &&
                             n <= kSmallMax && finish_blocks((int)n) <= 64;
    if (finish_book) {
        HA_REQUIRE(n <= c.nmax, "cache: batch of %ld keys exceeds max_batch %ld", (long)n, (long)c.nmax);
        const uint64_t lim = static_cast<uint64_t>(c.length);
        if (!presorted && (key_kind == 0 ? ha_plan_sort_f32ids_lim(static_cast<const float *>(keys), n, c.plan_ws, lim, s)
                                         : ha_plan_sort_u64ids_lim(static_cast<const uint64_t *>(keys), n, c.plan_ws, lim, s)))
            return -1;
        cache_mark(h, kTSort, s);
        hipLaunchKernelGGL(cache_finish_book_kernel, dim3(finish_blocks((int)n)), dim3(1024), 0, s, p.sorted, p.perm, (int)n,
                           p.hdr, p.uniq, p.seg, p.counts, p.inverse, p.upos, c, c.limit >= n ? 1 : 0);
        probed = 1;
    } else if (cache_plan(h, keys, key_kind, n, s, 1, &probed, presorted)) {
        return -1;
    } else {
        cache_mark(h, kTSort, s);
    }
    const int evb = finish_book ? (c.limit >= n ? 3 : 1) : cache_evict_mode(h, n, probed);
    h->plan_n = n;
    h->same_fast = (h->fused_update & 1) && h->evict_empty && c.policy == kLRU && !c.bypass && c.limit >= n && n > 0;
    h->evict_empty = false;
    if (!finish_book)
        hipLaunchKernelGGL(cache_lookup_book_kernel, dim3(1), dim3(1024), 0, s, c, p.hdr, p.uniq,
                           (long long)n, c.bypass ? 1 : 0, probed, evb);
    cache_mark(h, kTLookup, s);
    if (n > 0 && evb && c.policy != kLRU && c.S >= kScanWideFrom)      // LFU / LFUOpt: the victim scan of a large cache, in parallel
        hipLaunchKernelGGL(cache_scan_victim_part_kernel, dim3(kScanParts), dim3(1024), 0, s, c.ctl, c);
    if (n > 0) {
        const unsigned blocks = static_cast<unsigned>((n + 15) / 16) + (evb ? 1 : 0);
        const bool vec_ok = (c.width % 4 == 0) && (reinterpret_cast<uintptr_t>(dest) % 16 == 0) &&
                            (reinterpret_cast<uintptr_t>(c.table) % 16 == 0);
        if (vec_ok)
            hipLaunchKernelGGL(cache_lookup_rows_kernel<4>, dim3(blocks), dim3(1024), 0, s, c, p.uniq, p.upos,
                               p.perm, (long long)n, dest, evb, c.bypass ? 1 : 0);
        else
            hipLaunchKernelGGL(cache_lookup_rows_kernel<1>, dim3(blocks), dim3(1024), 0, s, c, p.uniq, p.upos,
                               p.perm, (long long)n, dest, evb, c.bypass ? 1 : 0);
    }
    cache_mark(h, kTEnd, s);
    HA_LAUNCH_CHECK();
    return 0;
}

// defer_cleanup: embedding_push_pull runs the version bump / zeroGrad of the pushed lines only after
// its pull phase, as the reference does (cache.cc:413-421 after 398-411).
static int cache_update_impl(ha_cache *h, const void *keys, int key_kind, int64_t n,
                             const float *grads, const void *push_keys, int push_kind,
                             int64_t n_push, int with_push_keys, hipStream_t s,
                             bool defer_cleanup = false) {
    HA_REQUIRE(h && (h->c.table || h->c.remote), "cache_update: no store bound");
    HA_REQUIRE(n >= 0 && (n == 0 || grads), "cache_update: bad arguments");
    HA_REQUIRE(ha_cache_plan_pending(h) == 0, "cache: %d planned calls are outstanding (ha_cache_plan_block): ha_cache_lookup_planned / "
               "ha_cache_update_planned come first", ha_cache_plan_pending(h));
    h->last_planned_type = -1;
    h->lfu_tree_ok = false;
    Cache &c = h->c;
    // keys == nullptr: the batch of the preceding ha_cache_lookup, whose plan is still in the workspace
    int probed = 0;
    cache_mark(h, kTStart, s, true);
    if (keys != nullptr) {
        if (cache_plan(h, keys, key_kind, n, s, 0, &probed))
            return -1;
        cache_mark(h, kTSort, s);
    } else {
        HA_REQUIRE(h->plan_n == n && n > 0, "cache_update_same_keys: no lookup of %ld keys precedes this update", (long)n);
    }
    h->plan_n = -1;
    PlanPtrs p = plan_layout(c.plan_ws, n);
    const dim3 b(256);
    const bool fast = keys == nullptr && h->same_fast && !with_push_keys && !defer_cleanup && !c.bypass;
    h->same_fast = false;
    h->evict_empty = true;     // every path below pushes the pending evicted lines
    if (fast) {
        ++h->fused_count;
        // every key is resident with data in uslot[]: accumulate, then touch / flags / push / commit in one launch
        if (ha_apply_mapped2(c.grad, c.S, c.data, c.width, c.plan_ws, n, grads, -1.0f, c.uslot, c.uslot, c.hasgrad, s))
            return -1;
        cache_mark(h, kTCopy, s);
        const unsigned pblocks = static_cast<unsigned>((2 * n + 15) / 16 > 1024 ? 1024 : (2 * n + 15) / 16);
        const bool vec_ok = !c.remote && (c.width % 4 == 0) && (reinterpret_cast<uintptr_t>(c.table) % 16 == 0);
        if (vec_ok)
            hipLaunchKernelGGL(cache_update_same_post_kernel<4>, dim3(pblocks), dim3(1024), 0, s, c, p.hdr, p.uniq,
                               p.counts, (long long)n, (long long)h->out_pad);
        else
            hipLaunchKernelGGL(cache_update_same_post_kernel<1>, dim3(pblocks), dim3(1024), 0, s, c, p.hdr, p.uniq,
                               p.counts, (long long)n, (long long)h->out_pad);
        cache_mark(h, kTEnd, s);
        HA_LAUNCH_CHECK();
        return 0;
    }
    const uint32_t *pk = nullptr;
    if (with_push_keys) {
        HA_REQUIRE(n_push <= c.nmax, "cache_update: too many push keys");
        if (n_push > 0) {
            if (push_kind == 0)
                hipLaunchKernelGGL(cache_f32_to_u32_kernel, CACHE_GRID(n_push), b, 0, s,
                                   static_cast<const float *>(push_keys), (long long)n_push, c.pushkeys_u32);
            else
                hipLaunchKernelGGL(cache_u64_to_u32_kernel, CACHE_GRID(n_push), b, 0, s,
                                   static_cast<const uint64_t *>(push_keys), (long long)n_push, c.pushkeys_u32);
        }
        pk = c.pushkeys_u32;
    }
    hipLaunchKernelGGL(cache_update_book_kernel, dim3(1), dim3(1024), 0, s, c, p.hdr, p.uniq,
                       c.bypass ? 1 : 0, probed);
    cache_mark(h, kTLookup, s);
    HA_LAUNCH_CHECK();
    // Line::accumulate per occurrence, occurrence order: grad += g (every line), data += g (lines
    // with data).  lr = -1 turns the SGD chain `acc - lr*g` into `acc + g` bit for bit.
    if (n > 0) {
        if (ha_apply_mapped2(c.grad, c.S, c.data, c.width, c.plan_ws, n, grads, -1.0f, c.uslot, c.data_row,
                             c.hasgrad, s))
            return -1;
    }
    cache_mark(h, kTCopy, s);
    hipLaunchKernelGGL(cache_update_flags_push_kernel, CACHE_GRID(n * 64), b, 0, s, c.ctl, c, p.uniq, p.counts,
                       pk, (long long)n_push, with_push_keys);
    hipLaunchKernelGGL(cache_push_evicted_kernel, CACHE_GRID(c.nmax * 64), b, 0, s, c.ctl, c, (long long)h->out_pad);
    cache_mark(h, kTTransfer, s);
    // the evicted lines that were pending before this call are pushed now: their slots are free again
    if (!defer_cleanup) {
        const unsigned cblocks = 1u + static_cast<unsigned>((n + 15) / 16 > 1024 ? 1024 : (n + 15) / 16);
        hipLaunchKernelGGL(cache_update_finish_kernel, dim3(cblocks), dim3(1024), 0, s, c.ctl, c, (long long)n,
                           with_push_keys);
    } else {
        hipLaunchKernelGGL(cache_update_commit_kernel, dim3(1), dim3(1024), 0, s, c.ctl, c, (long long)n);
    }
    cache_mark(h, kTEnd, s);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_cache_update(ha_cache *h, const void *keys, int key_kind, int64_t n,
                               const float *grads, ha_stream_t stream) {
    HA_REQUIRE(n == 0 || keys, "cache_update: null keys");
    return cache_update_impl(h, keys, key_kind, n, grads, nullptr, 0, 0, 0, as_stream(stream));
}

extern "C" int ha_cache_update_same_keys(ha_cache *h, int64_t n, const float *grads,
                                         ha_stream_t stream) {
    HA_REQUIRE(h, "cache_update_same_keys: null handle");
    return cache_update_impl(h, nullptr, 0, n, grads, nullptr, 0, 0, 0, as_stream(stream));
}

extern "C" int ha_cache_update_with_push_keys(ha_cache *h, const void *keys, int key_kind,
                                              int64_t n, const void *push_keys, int push_kind,
                                              int64_t n_push, const float *grads,
                                              ha_stream_t stream) {
    return cache_update_impl(h, keys, key_kind, n, grads, push_keys, push_kind, n_push, 1,
                             as_stream(stream));
}

// embedding_push_pull (cache.cc:356-422): pull phase (touch + new lines) on scratch set B, push phase
// (touch, accumulate, push incl. pending evictions) on scratch set A, then -- as
// PSHandler::serve(kPushSyncEmbedding) pushes before it syncs (PSFhandle_embedding.cc:66-79) -- the
// staleness-bounded pull, the copy to dest and the insert of the pull misses.
static Cache scratch_b_view(const Cache &c) {
    Cache cb = c;  // view with the B scratch set
    cb.plan_ws = c.plan_ws_b;
    cb.uslot = c.uslot_b;
    cb.data_row = c.data_row_b;
    cb.flag = c.flag_b;
    cb.rank = c.rank_b;
    cb.pushflag = c.pushflag_b;
    return cb;
}

// pull phase part 1 + the push phase.  Local store: the push lands in the table; remote: in the outbox,
// and the request of the pull keys is exported.
static int push_pull_begin(ha_cache *h, const void *pull_keys, int pull_kind, int64_t n_pull,
                           const void *push_keys, int push_kind, int64_t n_push, const float *grads,
                           hipStream_t s) {
    HA_REQUIRE(n_pull >= 0 && n_push >= 0 && n_pull <= h->c.nmax && n_push <= h->c.nmax,
               "cache_push_pull: bad sizes");
    HA_REQUIRE((n_pull == 0 || pull_keys) && (n_push == 0 || (push_keys && grads)),
               "cache_push_pull: null pointer");
    HA_REQUIRE(ha_cache_plan_pending(h) == 0, "cache: %d planned calls are outstanding (ha_cache_plan_block): ha_cache_lookup_planned / "
               "ha_cache_update_planned come first", ha_cache_plan_pending(h));
    h->last_planned_type = -1;
    h->lfu_tree_ok = false;
    Cache &c = h->c;
    Cache cb = scratch_b_view(c);
    const dim3 b(256);
    // ---- pull phase, part 1
    if (pull_kind == 0 ? ha_plan_build_f32ids(static_cast<const float *>(pull_keys), n_pull, cb.plan_ws, s)
                       : ha_plan_build_u64ids(static_cast<const uint64_t *>(pull_keys), n_pull, cb.plan_ws, s))
        return -1;
    PlanPtrs pp = plan_layout(cb.plan_ws, n_pull);
    hipLaunchKernelGGL(cache_probe_kernel, CACHE_GRID(n_pull), b, 0, s, c.ctl, pp.hdr, pp.uniq, c.slot_of,
                       (long long)c.length, c.bypass ? 1 : 0, cb.uslot, cb.flag);
    hipLaunchKernelGGL(cache_scan_kernel, dim3(1), dim3(1024), 0, s, pp.hdr, cb.flag, cb.rank, &c.ctl->M,
                       &c.ctl->nhit);
    hipLaunchKernelGGL(cache_assign_kernel, CACHE_GRID(n_pull), b, 0, s, c.ctl, c.ctl, cb, pp.uniq, cb.flag,
                       cb.rank, static_cast<int>(kPending));
    hipLaunchKernelGGL(cache_commit_touch_kernel, dim3(1), dim3(1), 0, s, c.ctl, c);
    // the pull misses hold stack entries until they are inserted: retire them now so that the push
    // phase's transient lines take different slots
    hipLaunchKernelGGL(cache_park_kernel, dim3(1), dim3(1), 0, s, c.ctl, 0);
    hipLaunchKernelGGL(cache_retire_kernel, dim3(1), dim3(1), 0, s, c.ctl, 1);
    HA_LAUNCH_CHECK();
    // ---- push phase (a complete embedding_update on scratch set A)
    if (cache_update_impl(h, push_keys, push_kind, n_push, grads, nullptr, 0, 0, 0, s, true))
        return -1;
    h->evict_empty = false;    // the pull phase's insert evicts after this push
    // ---- pull phase, part 2 starts: restore the parked pull state
    hipLaunchKernelGGL(cache_park_kernel, dim3(1), dim3(1), 0, s, c.ctl, 2);  // park the push phase's U
    hipLaunchKernelGGL(cache_retire_kernel, dim3(1), dim3(1), 0, s, c.ctl, 0);
    hipLaunchKernelGGL(cache_park_kernel, dim3(1), dim3(1), 0, s, c.ctl, 1);
    if (c.remote)   // the request: the pull keys and the versions their lines hold AFTER the push phase
        hipLaunchKernelGGL(cache_export_req_kernel, CACHE_GRID(n_pull), b, 0, s, cb, pp.hdr, pp.uniq, (int)n_pull);
    HA_LAUNCH_CHECK();
    return 0;
}

static int push_pull_finish(ha_cache *h, int64_t n_pull, float *dest, int64_t n_push, hipStream_t s) {
    Cache &c = h->c;
    Cache cb = scratch_b_view(c);
    const dim3 b(256);
    PlanPtrs pp = plan_layout(cb.plan_ws, n_pull);
    hipLaunchKernelGGL(cache_sync_kernel, CACHE_GRID(n_pull * 64), b, 0, s, c.ctl, cb, pp.uniq);
    if (n_pull > 0)
        hipLaunchKernelGGL(cache_dest_kernel, CACHE_GRID(n_pull * c.width), b, 0, s, cb, pp.inverse,
                           (long long)n_pull, dest);
    if (c.policy != kLRU) {
        const int parts = c.S >= kScanWideFrom ? 1 : 0;
        if (parts)
            hipLaunchKernelGGL(cache_scan_victim_part_kernel, dim3(kScanParts), dim3(1024), 0, s, c.ctl, c);
        hipLaunchKernelGGL(cache_scan_victim_kernel, dim3(1), dim3(1024), 0, s, c.ctl, c, parts);
    }
    hipLaunchKernelGGL(cache_insert_evict_kernel, dim3(1), dim3(1024), 0, s, c.ctl, cb, pp.uniq, cb.flag,
                       cb.rank, c.bypass ? 0 : 1);
    hipLaunchKernelGGL(cache_report_pull_kernel, dim3(1), dim3(1), 0, s, c.ctl, c, (long long)n_pull);
    // ---- deferred clean-up of the push phase (its unique count was parked before the pull phase resumed)
    hipLaunchKernelGGL(cache_park_kernel, dim3(1), dim3(1), 0, s, c.ctl, 3);
    hipLaunchKernelGGL(cache_update_cleanup_kernel, CACHE_GRID(n_push * 64), b, 0, s, c.ctl, c, 0);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_cache_push_pull(ha_cache *h, const void *pull_keys, int pull_kind, int64_t n_pull,
                                  float *dest, const void *push_keys, int push_kind, int64_t n_push,
                                  const float *grads, ha_stream_t stream) {
    HA_REQUIRE(h && h->c.table && !h->c.remote, "cache_push_pull: needs a local store (over a remote store: "
               "ha_cache_push_pull_begin, the exchange, ha_cache_push_pull_finish)");
    HA_REQUIRE(n_pull == 0 || dest, "cache_push_pull: null dest");
    hipStream_t s = as_stream(stream);
    if (push_pull_begin(h, pull_keys, pull_kind, n_pull, push_keys, push_kind, n_push, grads, s))
        return -1;
    return push_pull_finish(h, n_pull, dest, n_push, s);
}

// embedding_push_pull over a remote store.  begin: pull phase part 1 + push phase; the OUTBOX holds the
// lines to push (ha_cache_outbox_count), the REQUEST the pull keys.  The host pushes the outbox, then
// syncs the request (the server pushes before it syncs, PSFhandle_embedding.cc:66-79), then calls finish.
extern "C" int ha_cache_push_pull_begin(ha_cache *h, const void *pull_keys, int pull_kind, int64_t n_pull,
                                        const void *push_keys, int push_kind, int64_t n_push,
                                        const float *grads, int64_t *n_unique_pull_host, ha_stream_t stream) {
    HA_REQUIRE(h && h->c.remote, "cache_push_pull_begin: the cache is not in remote-store mode");
    hipStream_t s = as_stream(stream);
    if (push_pull_begin(h, pull_keys, pull_kind, n_pull, push_keys, push_kind, n_push, grads, s))
        return -1;
    h->pp_pull = n_pull;
    h->pp_push = n_push;
    if (n_unique_pull_host) {
        PlanPtrs pp = plan_layout(h->c.plan_ws_b, n_pull);
        HA_CHECK_HIP(hipMemcpyAsync(n_unique_pull_host, &pp.hdr->n_unique, 8, hipMemcpyDeviceToHost, s));
        HA_CHECK_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

extern "C" int ha_cache_push_pull_finish(ha_cache *h, float *dest, ha_stream_t stream) {
    HA_REQUIRE(h && h->c.remote && h->pp_pull >= 0, "cache_push_pull_finish: no ha_cache_push_pull_begin precedes");
    HA_REQUIRE(h->pp_pull == 0 || dest, "cache_push_pull_finish: null dest");
    const int64_t n_pull = h->pp_pull, n_push = h->pp_push;
    h->pp_pull = -1;
    return push_pull_finish(h, n_pull, dest, n_push, as_stream(stream));
}

// out[8]: last op report {type, num_all, num_unique, num_miss, num_transfered, num_evict, is_full, size}
extern "C" int ha_cache_perf(ha_cache *h, int64_t *out_host, ha_stream_t stream) {
    HA_REQUIRE(h && out_host, "cache_perf: bad arguments");
    if (h->last_planned_type >= 0 && h->last_planned != nullptr)      // the last call was a planned lookup / update
        return cache_perf_planned(h, out_host, as_stream(stream));
    CacheCtl ctl;
    HA_CHECK_HIP(hipMemcpyAsync(&ctl, h->c.ctl, sizeof(ctl), hipMemcpyDeviceToHost, as_stream(stream)));
    HA_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    HA_REQUIRE(ctl.fb_timeout == 0, "cache: a bookkeeping launch gave up (code %ld: 1 = a wait between its workgroups timed out, 2 = the "
               "LFU victim search found no line, 3 = a planned row launch met an item out of range; words %llu %llu %llu %llx %llx %llx "
               "%llu %llu); the cache's state is not to be trusted; HA_CACHE_FUSED=3 keeps the call-by-call bookkeeping in a launch "
               "of its own", (long)ctl.fb_timeout, ctl.ph[8], ctl.ph[9], ctl.ph[10], ctl.ph[11], ctl.ph[12], ctl.ph[13], ctl.ph[14], ctl.ph[15]);
    for (int i = 0; i < 7; ++i)
        out_host[i] = ctl.perf[i];
    out_host[4] += ctl.perf[7];
    out_host[7] = ctl.size;
    return 0;
}

// Stage times of the last lookup / update (milliseconds, as the reference's perf dict reports them: cache.cc:99-105,
// 189-194), measured with HIP events between the call's launches on the caller's stream.  Off by default (an event record
// between two launches costs about what a short launch costs).  out_ms[6]: for every stage boundary that the call passed,
// the time since the boundary before it -- [1] sort (index plan), [2] lookup (probe, miss scan, slot assignment: the
// reference's lookup + prepare), [3] copy (rows to dest + insert + eviction / the accumulate of an update), [4] transfer
// (remote stores: the exchange with the owners; updates: the push launches), [5] the rest of the call (an update's
// clean-up; a lookup's row launch); [0] = the whole call; -1 = boundary not passed.
extern "C" int ha_cache_set_timing(ha_cache *h, int on) {
    HA_REQUIRE(h, "cache_set_timing: null handle");
    if (on && h->tev[0] == nullptr)
        for (int i = 0; i < 6; ++i)
            HA_CHECK_HIP(hipEventCreate(&h->tev[i]));
    h->timing = on != 0;
    h->tmask = 0;
    return 0;
}
extern "C" int ha_cache_stage_times(ha_cache *h, double *out_ms) {
    HA_REQUIRE(h && out_ms, "cache_stage_times: bad arguments");
    for (int i = 0; i < 6; ++i)
        out_ms[i] = -1.0;
    if (!h->timing || !(h->tmask & 1u))
        return 0;
    int last = 0;
    for (int i = 1; i < 6; ++i)
        if (h->tmask & (1u << i))
            last = i;
    HA_CHECK_HIP(hipEventSynchronize(h->tev[last]));
    int prev = 0;
    for (int i = 1; i < 6; ++i) {
        if (!(h->tmask & (1u << i)))
            continue;
        float ms = 0.f;
        HA_CHECK_HIP(hipEventElapsedTime(&ms, h->tev[prev], h->tev[i]));
        out_ms[i] = ms;
        prev = i;
    }
    float all = 0.f;
    HA_CHECK_HIP(hipEventElapsedTime(&all, h->tev[0], h->tev[last]));
    out_ms[0] = all;
    return 0;
}

// out[8]: {size, evict_n, free_top, log_head, log_tail, clock, S, Lcap}
// out[16]: the 100 MHz clock at the phase boundaries of the last lookup's bookkeeping (tools/cache_phases.py):
// [0..4] cache_lookup_book_kernel: start, probe / U, miss scan, slot assignment + touch, commit + pull count;
// [8..12] workgroup 0 of cache_lookup_rows_kernel: start, insert of the misses, eviction, -, report;
// [13..15] the eviction walk: rounds, lines to evict, log entries consumed
extern "C" int ha_cache_phase_times(ha_cache *h, uint64_t *out_host, ha_stream_t stream) {
    HA_REQUIRE(h && out_host, "cache_phase_times: bad arguments");
    CacheCtl ctl;
    HA_CHECK_HIP(hipMemcpyAsync(&ctl, h->c.ctl, sizeof(ctl), hipMemcpyDeviceToHost, as_stream(stream)));
    HA_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    for (int i = 0; i < 16; ++i)
        out_host[i] = ctl.ph[i];
    return 0;
}

extern "C" int ha_cache_state(ha_cache *h, int64_t *out_host, ha_stream_t stream) {
    HA_REQUIRE(h && out_host, "cache_state: bad arguments");
    for (PlanSlot &sl : h->plan)       // (a bookkeeping launch of the planned flow owns the control block while it runs)
        if (sl.booked && sl.count > 0)
            HA_CHECK_HIP(hipEventSynchronize(sl.booked));
    CacheCtl ctl;
    HA_CHECK_HIP(hipMemcpyAsync(&ctl, h->c.ctl, sizeof(ctl), hipMemcpyDeviceToHost, as_stream(stream)));
    HA_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    HA_REQUIRE(ctl.fb_timeout == 0, "cache: a bookkeeping launch gave up (code %ld: 1 = a wait between its workgroups timed out, 2 = the "
               "LFU victim search found no line, 3 = a planned row launch met an item out of range; words %llu %llu %llu %llx %llx %llx "
               "%llu %llu); the cache's state is not to be trusted; HA_CACHE_FUSED=3 keeps the call-by-call bookkeeping in a launch "
               "of its own", (long)ctl.fb_timeout, ctl.ph[8], ctl.ph[9], ctl.ph[10], ctl.ph[11], ctl.ph[12], ctl.ph[13], ctl.ph[14], ctl.ph[15]);
    out_host[0] = ctl.size;
    out_host[1] = ctl.evict_n;
    out_host[2] = ctl.free_top;
    out_host[3] = ctl.log_head;
    out_host[4] = ctl.log_tail;
    out_host[5] = ctl.clock;
    out_host[6] = h->c.S;
    out_host[7] = h->c.Lcap;
    return 0;
}

// Resident lines, unordered: device arrays of capacity cap; returns the count through *count_dev
// (device u64, zeroed by the caller).  slots[i] indexes ha_cache_data / ha_cache_grad rows.
extern "C" int ha_cache_snapshot(ha_cache *h, int64_t cap, uint32_t *keys, int64_t *version,
                                 int32_t *updates, uint64_t *stamp, int32_t *slots,
                                 uint64_t *count_dev, ha_stream_t stream) {
    HA_REQUIRE(h && keys && version && updates && stamp && slots && count_dev, "cache_snapshot: null pointer");
    for (PlanSlot &sl : h->plan)
        if (sl.booked && sl.count > 0)
            HA_CHECK_HIP(hipStreamWaitEvent(as_stream(stream), sl.booked, 0));
    hipLaunchKernelGGL(cache_snapshot_kernel, dim3(1024), dim3(256), 0, as_stream(stream), h->c,
                       (long long)cap, keys, reinterpret_cast<long long *>(version), updates,
                       reinterpret_cast<unsigned long long *>(stamp), slots,
                       reinterpret_cast<unsigned long long *>(count_dev));
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_cache_set_line(ha_cache *h, int64_t key, int64_t version, const float *data_dev,
                                 ha_stream_t stream) {
    HA_REQUIRE(h && data_dev, "cache_set_line: null pointer");
    hipLaunchKernelGGL(cache_set_line_kernel, dim3(1), dim3(256), 0, as_stream(stream), h->c,
                       static_cast<long long>(key), static_cast<long long>(version), data_dev);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" float *ha_cache_data(ha_cache *h) { return h ? h->c.data : nullptr; }
extern "C" float *ha_cache_grad(ha_cache *h) { return h ? h->c.grad : nullptr; }
extern "C" int64_t ha_cache_limit(ha_cache *h) { return h ? h->c.limit : -1; }
extern "C" int64_t ha_cache_width(ha_cache *h) { return h ? h->c.width : -1; }
extern "C" int64_t ha_cache_fused_updates(ha_cache *h) { return h ? h->fused_count : -1; }
