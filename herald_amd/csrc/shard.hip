// Row-range sharding helpers for the in-node replacement of the PS/worker split.
//
// Reference: AveragePartitioner::partitionDense (ps-lite/include/ps/partitioner.h:46-57) gives shard i
// `len/S + (i < len%S)` contiguous rows; PSAgent routes SORTED unique keys to shards with
// std::lower_bound on the cumulative lengths and rebases them to shard-local offsets
// (ps-lite/include/ps/worker/PSAgent.h:537-560, 185-237).  ha_shard_bucket does the same on the
// device: for the plan's sorted unique keys it writes offsets[W+1] (offsets[g] = first unique key
// owned by shard g) and local[u] = uniq[u] - start[owner(u)].
#include "plan_dev.h"

namespace ha {

constexpr int kMaxShards = 64;

struct ShardStarts {
    uint32_t start[kMaxShards + 1];
};

__global__ __launch_bounds__(256) void shard_bucket_kernel(
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq,
    ShardStarts st, int nshard, int32_t *__restrict__ offsets,
    uint32_t *__restrict__ local) {
    const int U = static_cast<int>(hdr->n_unique);
    const int tid = blockIdx.x * 256 + threadIdx.x;
    // offsets[g] = lower_bound(uniq, start[g]) -- one thread per boundary
    if (tid <= nshard) {
        const uint32_t target = st.start[tid];
        int lo = 0, hi = U;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        offsets[tid] = tid == nshard ? U : lo;
    }
    const int stride = gridDim.x * 256;
    for (int u = tid; u < U; u += stride) {
        const uint32_t k = uniq[u];
        int g = 0;
        // nshard is tiny (<= 64): linear scan over the boundaries
        while (g + 1 < nshard && k >= st.start[g + 1])
            ++g;
        local[u] = k - st.start[g];
    }
}

// meta[0] = n_unique, meta[1+g] = number of unique keys owned by shard g (int64: the dtype the counts
// all-to-all and the host read-back use)
__global__ __launch_bounds__(64) void shard_meta_kernel(
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq,
    ShardStarts st, int nshard, int64_t *__restrict__ meta) {
    const int U = static_cast<int>(hdr->n_unique);
    const int g = threadIdx.x;
    if (g == 0)
        meta[0] = U;
    if (g < nshard) {
        int bound[2];
        for (int e = 0; e < 2; ++e) {
            const uint32_t target = st.start[g + e];
            int lo = 0, hi = U;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (uniq[mid] < target)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            bound[e] = (g + e == nshard) ? U : lo;
        }
        meta[1 + g] = bound[1] - bound[0];
    }
}

// Routing message of one batch, one launch: send[g] = [count_g, the shard-local keys owner g is asked for ...]
// in a FIXED frame of 1 + cap words per owner, so that the keys travel in one equal-split all-to-all together
// with their counts (no counts exchange, no host read-back before the keys can move); meta = n_unique and
// the W counts for the one asynchronous read-back that sizes the row exchanges.
__global__ __launch_bounds__(256) void shard_pack_kernel(
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq, ShardStarts st, int nshard, int cap,
    int64_t *__restrict__ meta, int32_t *__restrict__ send) {
    __shared__ int s_off[kMaxShards + 1];
    const int U = static_cast<int>(hdr->n_unique);
    if (threadIdx.x <= static_cast<unsigned>(nshard)) {
        const uint32_t target = st.start[threadIdx.x];
        int lo = 0, hi = U;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        s_off[threadIdx.x] = static_cast<int>(threadIdx.x) == nshard ? U : lo;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0)
            meta[0] = U;
        if (threadIdx.x < static_cast<unsigned>(nshard)) {
            const int c = s_off[threadIdx.x + 1] - s_off[threadIdx.x];
            meta[1 + threadIdx.x] = c;
            send[static_cast<size_t>(threadIdx.x) * (1 + cap)] = c;
        }
    }
    const int stride = gridDim.x * 256;
    for (int u = blockIdx.x * 256 + threadIdx.x; u < U; u += stride) {
        const uint32_t k = uniq[u];
        int g = 0;
        while (g + 1 < nshard && u >= s_off[g + 1])
            ++g;
        send[static_cast<size_t>(g) * (1 + cap) + 1 + (u - s_off[g])] = static_cast<int32_t>(k - st.start[g]);
    }
}

// Receiving side: recv[g] = [count_g, keys ...] from rank g -> the W key lists concatenated in rank order
// (what the owner-side gather / serve_push take) and the W counts next to the sender's meta.
__global__ __launch_bounds__(256) void shard_unpack_kernel(const int32_t *__restrict__ recv, int nshard, int cap,
                                                           int64_t *__restrict__ recv_cnt,
                                                           uint32_t *__restrict__ keys_out) {
    __shared__ int s_pre[kMaxShards + 1];
    if (threadIdx.x == 0) {
        int run = 0;
        for (int g = 0; g < nshard; ++g) {
            s_pre[g] = run;
            int c = recv[static_cast<size_t>(g) * (1 + cap)];
            c = c < 0 ? 0 : (c > cap ? cap : c);
            run += c;
        }
        s_pre[nshard] = run;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < static_cast<unsigned>(nshard))
        recv_cnt[threadIdx.x] = s_pre[threadIdx.x + 1] - s_pre[threadIdx.x];
    const int total = s_pre[nshard];
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < total; j += stride) {
        int g = 0;
        while (g + 1 < nshard && j >= s_pre[g + 1])
            ++g;
        keys_out[j] = static_cast<uint32_t>(recv[static_cast<size_t>(g) * (1 + cap) + 1 + (j - s_pre[g])]);
    }
}

}  // namespace ha

using namespace ha;

template <typename IdT>
static int shard_route_pack(const IdT *ids, int64_t n, void *plan_ws, const int64_t *starts_host, int nshard,
                            int64_t cap, int64_t *meta, int32_t *send, ha_stream_t stream,
                            int (*build)(const IdT *, int64_t, void *, uint64_t, ha_stream_t)) {
    HA_REQUIRE(plan_ws && starts_host && meta && send, "shard_route_pack: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards, "shard_route_pack: nshard must be in [1,%d]", kMaxShards);
    HA_REQUIRE(n >= 0 && cap >= n && cap < (1ll << 30), "shard_route_pack: the frame (%ld keys) must hold the batch (%ld ids)",
               static_cast<long>(cap), static_cast<long>(n));
    if (build(ids, n, plan_ws, static_cast<uint64_t>(starts_host[nshard]), stream))   // keys < total rows
        return -1;
    PlanPtrs p = plan_layout(plan_ws, n);
    ShardStarts st;
    for (int g = 0; g <= nshard; ++g) {
        HA_REQUIRE(starts_host[g] >= 0 && starts_host[g] <= 0xFFFFFFFEll, "shard_route_pack: start out of range");
        st.start[g] = static_cast<uint32_t>(starts_host[g]);
    }
    int blocks = static_cast<int>((n + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(shard_pack_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p.hdr, p.uniq, st, nshard,
                       static_cast<int>(cap), meta, send);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_shard_route_pack_f32ids(const float *ids, int64_t n, void *plan_ws, const int64_t *starts_host,
                                          int nshard, int64_t cap, int64_t *meta, int32_t *send,
                                          ha_stream_t stream) {
    return shard_route_pack<float>(ids, n, plan_ws, starts_host, nshard, cap, meta, send, stream,
                                   ha_plan_build_f32ids_lim);
}

extern "C" int ha_shard_route_pack_u64ids(const uint64_t *ids, int64_t n, void *plan_ws, const int64_t *starts_host,
                                          int nshard, int64_t cap, int64_t *meta, int32_t *send,
                                          ha_stream_t stream) {
    return shard_route_pack<uint64_t>(ids, n, plan_ws, starts_host, nshard, cap, meta, send, stream,
                                      ha_plan_build_u64ids_lim);
}

extern "C" int ha_shard_route_unpack(const int32_t *recv, int nshard, int64_t cap, int64_t *recv_cnt,
                                     uint32_t *keys_out, ha_stream_t stream) {
    HA_REQUIRE(recv && recv_cnt && keys_out, "shard_route_unpack: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && cap >= 0 && cap < (1ll << 30), "shard_route_unpack: bad sizes");
    long long want = (static_cast<long long>(nshard) * cap + 255) / 256;
    const int blocks = want < 1 ? 1 : (want > 256 ? 256 : static_cast<int>(want));
    hipLaunchKernelGGL(shard_unpack_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), recv, nshard,
                       static_cast<int>(cap), recv_cnt, keys_out);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_shard_bucket(const void *plan_ws, int64_t n,
                               const int64_t *starts_host, int nshard,
                               int32_t *offsets, uint32_t *local_keys,
                               ha_stream_t stream) {
    HA_REQUIRE(plan_ws && starts_host && offsets && local_keys, "shard_bucket: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards, "shard_bucket: nshard must be in [1,%d]", kMaxShards);
    HA_REQUIRE(n >= 0, "shard_bucket: bad n");
    PlanPtrs p = plan_layout(const_cast<void *>(plan_ws), n);
    ShardStarts st;
    for (int g = 0; g <= nshard; ++g) {
        HA_REQUIRE(starts_host[g] >= 0 && starts_host[g] <= 0xFFFFFFFEll, "shard_bucket: start out of range");
        st.start[g] = static_cast<uint32_t>(starts_host[g]);
    }
    int blocks = static_cast<int>((n + 255) / 256);
    if (blocks < 1)
        blocks = 1;
    if (blocks > 1024)
        blocks = 1024;
    hipLaunchKernelGGL(shard_bucket_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       p.hdr, p.uniq, st, nshard, offsets, local_keys);
    HA_LAUNCH_CHECK();
    return 0;
}

template <typename IdT>
static int shard_route(const IdT *ids, int64_t n, void *plan_ws, const int64_t *starts_host, int nshard,
                       int64_t *meta, uint32_t *local_keys, ha_stream_t stream,
                       int (*build)(const IdT *, int64_t, void *, uint64_t, ha_stream_t)) {
    HA_REQUIRE(plan_ws && starts_host && meta && local_keys, "shard_route: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards, "shard_route: nshard must be in [1,%d]", kMaxShards);
    HA_REQUIRE(n >= 0, "shard_route: bad n");
    if (build(ids, n, plan_ws, static_cast<uint64_t>(starts_host[nshard]), stream))   // keys < total rows
        return -1;
    PlanPtrs p = plan_layout(plan_ws, n);
    ShardStarts st;
    for (int g = 0; g <= nshard; ++g) {
        HA_REQUIRE(starts_host[g] >= 0 && starts_host[g] <= 0xFFFFFFFEll, "shard_route: start out of range");
        st.start[g] = static_cast<uint32_t>(starts_host[g]);
    }
    if (n == 0) {
        HA_CHECK_HIP(hipMemsetAsync(meta, 0, sizeof(int64_t) * (1 + nshard), as_stream(stream)));
        return 0;
    }
    int blocks = static_cast<int>((n + 255) / 256);
    if (blocks > 1024)
        blocks = 1024;
    // the per-shard offsets of ha_shard_bucket are not needed here; they land in the radix scratch
    // of the plan (free after the build, >= 257 words)
    hipLaunchKernelGGL(shard_bucket_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       p.hdr, p.uniq, st, nshard, reinterpret_cast<int32_t *>(p.hist), local_keys);
    hipLaunchKernelGGL(shard_meta_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                       p.hdr, p.uniq, st, nshard, meta);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_shard_route_f32ids(const float *ids, int64_t n, void *plan_ws,
                                     const int64_t *starts_host, int nshard, int64_t *meta,
                                     uint32_t *local_keys, ha_stream_t stream) {
    return shard_route<float>(ids, n, plan_ws, starts_host, nshard, meta, local_keys, stream,
                              ha_plan_build_f32ids_lim);
}

extern "C" int ha_shard_route_u64ids(const uint64_t *ids, int64_t n, void *plan_ws,
                                     const int64_t *starts_host, int nshard, int64_t *meta,
                                     uint32_t *local_keys, ha_stream_t stream) {
    return shard_route<uint64_t>(ids, n, plan_ws, starts_host, nshard, meta, local_keys, stream,
                                 ha_plan_build_u64ids_lim);
}

// ---- fixed-size frames: the sharded step without host-known counts (graph-replayable) ---------------------------
// The sized exchanges above need the per-owner counts on the host before the rows can move.  Here every owner gets a
// FIXED key frame [count, overflow flag, rcap shard-local keys padded with kNoKey] and a fixed row frame of rcap rows,
// so pull and push are equal-split all-to-alls whose sizes the host knows without a read-back, and every kernel runs
// over W * rcap slots (padding slots: key beyond any table -> zero row on the pull side, skipped on the push side).
// A batch that names more than rcap unique keys of one owner sets the flag; every frame of the sender carries it, so
// after the key exchange every rank knows whether ANY rank overflowed (herald_amd/sharded.py then takes the sized
// exchange for that batch).  Routing semantics as above (PSAgent.h:185-237, partitioner.h:46-57).
namespace ha {

constexpr uint32_t kNoKey = 0xFFFFFFFFu;

// send[g] = [count_g, flag, keys ...]; rowmap[u] = frame slot (g * rcap + j) of unique key u, -1 beyond rcap;
// posmap[i] = frame slot of the unique key of position i (what the expand gathers from), the zero row W * rcap
// beyond rcap
__device__ __forceinline__ void shard_pack_frames_body(
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq, const int32_t *__restrict__ inverse, int n,
    const ShardStarts &st, int nshard, int rcap, size_t fw, int32_t *__restrict__ send, int32_t *__restrict__ rowmap,
    int32_t *__restrict__ posmap, int *s_off, int *s_flag_p) {
    int &s_flag = *s_flag_p;
    const int U = n > 0 ? static_cast<int>(hdr->n_unique) : 0;
    if (threadIdx.x <= static_cast<unsigned>(nshard)) {
        const uint32_t target = st.start[threadIdx.x];
        int lo = 0, hi = U;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        s_off[threadIdx.x] = static_cast<int>(threadIdx.x) == nshard ? U : lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int f = 0;
        for (int g = 0; g < nshard; ++g)
            f |= (s_off[g + 1] - s_off[g]) > rcap;
        s_flag = f;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < static_cast<unsigned>(nshard)) {
        send[threadIdx.x * fw] = s_off[threadIdx.x + 1] - s_off[threadIdx.x];
        send[threadIdx.x * fw + 1] = s_flag;
    }
    const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    const int slots = nshard * rcap;
    for (int p = tid; p < slots; p += stride) {
        const int g = p / rcap, j = p - g * rcap;
        const bool live = j < s_off[g + 1] - s_off[g];
        send[g * fw + 2 + j] = static_cast<int32_t>(live ? uniq[s_off[g] + j] - st.start[g] : kNoKey);
    }
    for (int u = tid; u < U; u += stride) {
        int g = 0;
        while (g + 1 < nshard && u >= s_off[g + 1])
            ++g;
        const int j = u - s_off[g];
        rowmap[u] = j < rcap ? g * rcap + j : -1;
    }
    for (int i = tid; i < n; i += stride) {
        const int u = inverse[i];
        int g = 0;
        while (g + 1 < nshard && u >= s_off[g + 1])
            ++g;
        const int j = u - s_off[g];
        posmap[i] = j < rcap ? g * rcap + j : slots;
    }
}

__global__ __launch_bounds__(256) void shard_pack_frames_kernel(
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq, const int32_t *__restrict__ inverse, int n,
    ShardStarts st, int nshard, int rcap, size_t fw, int32_t *__restrict__ send, int32_t *__restrict__ rowmap,
    int32_t *__restrict__ posmap) {
    __shared__ int s_off[kMaxShards + 1];
    __shared__ int s_flag;
    shard_pack_frames_body(hdr, uniq, inverse, n, st, nshard, rcap, fw, send, rowmap, posmap, s_off, &s_flag);
}

// the same for up to kFrameBatchMax batches in one launch (blockIdx.y = batch; batch i's frames start i * (2 + rcap)
// words into every owner's stride)
constexpr int kFrameBatchMax = 16;
struct FramePackBatch {
    const PlanHeader *hdr[kFrameBatchMax];
    const uint32_t *uniq[kFrameBatchMax];
    const int32_t *inverse[kFrameBatchMax];
    int n[kFrameBatchMax];
    int32_t *rowmap[kFrameBatchMax], *posmap[kFrameBatchMax];
};
__global__ __launch_bounds__(256) void shard_pack_frames_batch_kernel(const FramePackBatch b, ShardStarts st, int nshard,
                                                                      int rcap, size_t fw, int32_t *__restrict__ send) {
    __shared__ int s_off[kMaxShards + 1];
    __shared__ int s_flag;
    const int i = blockIdx.y;
    shard_pack_frames_body(b.hdr[i], b.uniq[i], b.inverse[i], b.n[i], st, nshard, rcap, fw,
                           send + static_cast<size_t>(i) * (2 + static_cast<size_t>(rcap)), b.rowmap[i], b.posmap[i], s_off,
                           &s_flag);
}

// recv[g] = [count_g, flag_g, keys ...] from rank g -> keys_fixed[g * rcap + j] (kNoKey beyond the count) and
// state = {any rank overflowed, keys received}
__device__ __forceinline__ void shard_unpack_frames_body(const int32_t *__restrict__ recv, int nshard, int rcap, size_t fw,
                                                         uint32_t *__restrict__ keys_fixed, int32_t *__restrict__ state) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int f = 0, total = 0;
        for (int g = 0; g < nshard; ++g) {
            const int c = recv[g * fw];
            f |= (recv[g * fw + 1] != 0) | (c > rcap) | (c < 0);
            total += c < 0 ? 0 : (c > rcap ? rcap : c);
        }
        state[0] = f;
        state[1] = total;
    }
    const int slots = nshard * rcap, stride = gridDim.x * 256;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < slots; p += stride) {
        const int g = p / rcap, j = p - g * rcap;
        const int c = recv[g * fw];
        keys_fixed[p] = j < c ? static_cast<uint32_t>(recv[g * fw + 2 + j]) : kNoKey;
    }
}

__global__ __launch_bounds__(256) void shard_unpack_frames_kernel(const int32_t *__restrict__ recv, int nshard, int rcap,
                                                                  size_t fw, uint32_t *__restrict__ keys_fixed,
                                                                  int32_t *__restrict__ state) {
    shard_unpack_frames_body(recv, nshard, rcap, fw, keys_fixed, state);
}

struct FrameUnpackBatch {
    uint32_t *keys_fixed[kFrameBatchMax];
    int32_t *state[kFrameBatchMax];
};
__global__ __launch_bounds__(256) void shard_unpack_frames_batch_kernel(const int32_t *__restrict__ recv, int nshard,
                                                                        int rcap, size_t fw, const FrameUnpackBatch b) {
    const int i = blockIdx.y;
    shard_unpack_frames_body(recv + static_cast<size_t>(i) * (2 + static_cast<size_t>(rcap)), nshard, rcap, fw,
                             b.keys_fixed[i], b.state[i]);
}

// Owner side of a framed pull, one launch: received key frames -> keys_fixed / state (what shard_unpack_frames_kernel
// writes) AND the rows of the live slots into the row frames (PSHandler::serve(SparsePull), PSFHandle.h:101-128).
// One wave per slot; unused slots are not written (the expand never reads them).
template <int VEC>
__global__ __launch_bounds__(256) void shard_serve_pull_frames_kernel(
    const float *__restrict__ table, uint64_t rows, int width, const int32_t *__restrict__ recv, int nshard, int rcap,
    size_t fw, float *__restrict__ rows_out, uint32_t *__restrict__ keys_fixed, int32_t *__restrict__ state) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int f = 0, total = 0;
        for (int g = 0; g < nshard; ++g) {
            const int c = recv[g * fw];
            f |= (recv[g * fw + 1] != 0) | (c > rcap) | (c < 0);
            total += c < 0 ? 0 : (c > rcap ? rcap : c);
        }
        state[0] = f;
        state[1] = total;
    }
    const int lane = threadIdx.x & 63;
    const int slots = nshard * rcap;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= slots)
        return;
    const int g = p / rcap, j = p - g * rcap;
    const int c = recv[g * fw];
    const uint32_t key = j < c ? static_cast<uint32_t>(recv[g * fw + 2 + j]) : kNoKey;
    if (lane == 0)
        keys_fixed[p] = key;
    if (key == kNoKey)
        return;
    float *dst = rows_out + static_cast<uint64_t>(p) * static_cast<uint64_t>(width);
    const bool ok = key < rows;     // a key beyond the shard reads as zeros (the library's definition)
    const float *src = table + static_cast<uint64_t>(ok ? key : 0u) * static_cast<uint64_t>(width);
    if (VEC == 4) {
        for (int col = 4 * lane; col < width; col += 256) {
            const float4v v = ok ? ld4(src + col) : float4v{0.f, 0.f, 0.f, 0.f};
            st4_nt(dst + col, v);
        }
    } else {
        for (int col = lane; col < width; col += 64)
            dst[col] = ok ? src[col] : 0.f;
    }
}

// Owner side of a framed push, the apply: the merged list holds every key at most once per sender (W <= kMaxShards
// entries per run, in rank order), so a wave per run HEAD adds the run's value rows to the table row in that order --
// table[key,:] = ((table[key,:] + v_0) + v_1) ... : bit for bit what ha_sgd_apply_finish computes with lr = -1
// (acc - (-1 * v)), without the finish of an index plan nobody reads and without its treatment of the padding (the
// unused slots of the frames carry one key beyond any table: a "run" of tens of thousands of positions to the
// generic apply).  perm == nullptr: the list is one sender's frame as it arrived.
template <int VEC>
__global__ __launch_bounds__(256) void shard_frames_apply_kernel(float *__restrict__ table, unsigned long long rows, int width,
                                                                 const uint32_t *__restrict__ sorted,
                                                                 const int32_t *__restrict__ perm, int n,
                                                                 const float *__restrict__ values) {
    const int lane = threadIdx.x & 63;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += gridDim.x * 4) {
        const uint32_t key = sorted[p];
        if (key >= rows || (p > 0 && sorted[p - 1] == key))
            continue;           // padding / beyond the table / not the head of its run
        int len = 1;
        while (p + len < n && sorted[p + len] == key)
            ++len;
        float *row = table + static_cast<unsigned long long>(key) * width;
        if (VEC == 4) {
            for (int c = lane * 4; c < width; c += 256) {
                float4v acc = ld4(row + c);
                for (int t = 0; t < len; ++t) {
                    const long long src = perm ? perm[p + t] : p + t;
                    const float4v v = ld4(values + src * width + c);
                    acc = float4v{__fadd_rn(acc[0], v[0]), __fadd_rn(acc[1], v[1]), __fadd_rn(acc[2], v[2]),
                                  __fadd_rn(acc[3], v[3])};
                }
                st4(row + c, acc);
            }
        } else {
            for (int c = lane; c < width; c += 64) {
                float acc = row[c];
                for (int t = 0; t < len; ++t) {
                    const long long src = perm ? perm[p + t] : p + t;
                    acc = __fadd_rn(acc, values[src * width + c]);
                }
                row[c] = acc;
            }
        }
    }
}

static int shard_frames_apply(float *table, int64_t rows, int64_t width, const uint32_t *sorted, const int32_t *perm,
                              int64_t n, const float *values, hipStream_t s) {
    unsigned blocks = static_cast<unsigned>((n + 3) / 4);
    if (blocks > 16384)
        blocks = 16384;
    const bool vec_ok = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(table) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(values) % 16 == 0);
    if (vec_ok)
        hipLaunchKernelGGL(shard_frames_apply_kernel<4>, dim3(blocks), dim3(256), 0, s, table,
                           static_cast<unsigned long long>(rows), static_cast<int>(width), sorted, perm,
                           static_cast<int>(n), values);
    else
        hipLaunchKernelGGL(shard_frames_apply_kernel<1>, dim3(blocks), dim3(256), 0, s, table,
                           static_cast<unsigned long long>(rows), static_cast<int>(width), sorted, perm,
                           static_cast<int>(n), values);
    HA_LAUNCH_CHECK();
    return 0;
}

// Stable merge of the W received key lists of a framed push (each ascending, unique inside a list, kNoKey = the
// largest key in the unused slots): slot p = (s, j) gets rank j + sum over s' < s of upper_bound(list s', key) + sum
// over s' > s of lower_bound(list s', key) -- the position of (key, s) in the rank-ordered merge -- by binary searches
// over the lists staged in LDS; sorted[rank] = key, perm[rank] = p: what the sort of an index plan produces, in one
// short launch instead of a rank-by-counting of W * rcap keys against each other.
__global__ __launch_bounds__(256) void shard_merge_rank_kernel(const uint32_t *__restrict__ keys, int nshard, int rcap,
                                                               uint32_t *__restrict__ sorted, int32_t *__restrict__ perm) {
    extern __shared__ uint32_t s_keys[];
    const int slots = nshard * rcap;
    for (int q = threadIdx.x; q < slots; q += 256)
        s_keys[q] = keys[q];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= slots)
        return;
    const int s = p / rcap, j = p - s * rcap;
    const uint32_t key = s_keys[p];
    int rank = j;
    for (int t = 0; t < nshard; ++t) {
        if (t == s)
            continue;
        const uint32_t *list = s_keys + t * rcap;
        int lo = 0, hi = rcap;
        if (t < s) {            // upper bound: equal keys of earlier ranks come first
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (list[mid] <= key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
        } else {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (list[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
        }
        rank += lo;
    }
    sorted[rank] = key;
    perm[rank] = p;
}

}  // namespace ha

static int shard_starts_of(const int64_t *starts_host, int nshard, ShardStarts *st, const char *who) {
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards, "%s: nshard must be in [1,%d]", who, kMaxShards);
    for (int g = 0; g <= nshard; ++g) {
        HA_REQUIRE(starts_host[g] >= 0 && starts_host[g] <= 0xFFFFFFFEll, "%s: start out of range", who);
        st->start[g] = static_cast<uint32_t>(starts_host[g]);
    }
    return 0;
}

extern "C" int ha_shard_frames_pack(const void *plan_ws, int64_t n, const int64_t *starts_host, int nshard, int64_t rcap,
                                    int64_t frame_stride, int32_t *send, int32_t *rowmap, int32_t *posmap,
                                    ha_stream_t stream) {
    HA_REQUIRE(plan_ws && starts_host && send && rowmap && posmap, "shard_frames_pack: null pointer");
    HA_REQUIRE(n >= 0 && rcap >= 1 && static_cast<int64_t>(nshard) * rcap < (1ll << 30) && frame_stride >= 2 + rcap,
               "shard_frames_pack: bad sizes");
    ShardStarts st;
    if (shard_starts_of(starts_host, nshard, &st, "shard_frames_pack"))
        return -1;
    PlanPtrs p = plan_layout(const_cast<void *>(plan_ws), n > 0 ? n : 1);
    const int64_t work = n > nshard * rcap ? n : nshard * rcap;
    int blocks = static_cast<int>((work + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(shard_pack_frames_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p.hdr, p.uniq, p.inverse,
                       static_cast<int>(n), st, nshard, static_cast<int>(rcap), static_cast<size_t>(frame_stride), send,
                       rowmap, posmap);
    HA_LAUNCH_CHECK();
    return 0;
}

template <typename IdT>
static int shard_frames_route(const IdT *ids, int64_t n, void *plan_ws, const int64_t *starts_host, int nshard,
                              int64_t rcap, int64_t frame_stride, int32_t *send, int32_t *rowmap, int32_t *posmap,
                              ha_stream_t stream,
                              int (*build)(const IdT *, int64_t, void *, uint64_t, ha_stream_t)) {
    HA_REQUIRE(plan_ws && starts_host, "shard_frames_route: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards, "shard_frames_route: nshard must be in [1,%d]", kMaxShards);
    if (n > 0 && build(ids, n, plan_ws, static_cast<uint64_t>(starts_host[nshard]), stream))   // keys < total rows
        return -1;
    return ha_shard_frames_pack(plan_ws, n, starts_host, nshard, rcap, frame_stride, send, rowmap, posmap, stream);
}

extern "C" int ha_shard_frames_route_f32ids(const float *ids, int64_t n, void *plan_ws, const int64_t *starts_host,
                                            int nshard, int64_t rcap, int64_t frame_stride, int32_t *send,
                                            int32_t *rowmap, int32_t *posmap, ha_stream_t stream) {
    return shard_frames_route<float>(ids, n, plan_ws, starts_host, nshard, rcap, frame_stride, send, rowmap, posmap, stream,
                                     ha_plan_build_f32ids_lim);
}

extern "C" int ha_shard_frames_route_u64ids(const uint64_t *ids, int64_t n, void *plan_ws, const int64_t *starts_host,
                                            int nshard, int64_t rcap, int64_t frame_stride, int32_t *send,
                                            int32_t *rowmap, int32_t *posmap, ha_stream_t stream) {
    return shard_frames_route<uint64_t>(ids, n, plan_ws, starts_host, nshard, rcap, frame_stride, send, rowmap, posmap, stream,
                                        ha_plan_build_u64ids_lim);
}

extern "C" int ha_shard_frames_pack_batch(const void *const *plan_ws, const int64_t *n, int count,
                                          const int64_t *starts_host, int nshard, int64_t rcap, int64_t frame_stride,
                                          int32_t *send, int32_t *const *rowmap, int32_t *const *posmap,
                                          ha_stream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (plan_ws && n && send && rowmap && posmap)) && starts_host,
               "shard_frames_pack_batch: null pointer");
    HA_REQUIRE(rcap >= 1 && static_cast<int64_t>(nshard) * rcap < (1ll << 30) && frame_stride >= count * (2 + rcap),
               "shard_frames_pack_batch: the stride must hold the frames of all batches");
    ShardStarts st;
    if (shard_starts_of(starts_host, nshard, &st, "shard_frames_pack_batch"))
        return -1;
    for (int at = 0; at < count; at += kFrameBatchMax) {
        const int m = count - at < kFrameBatchMax ? count - at : kFrameBatchMax;
        FramePackBatch b;
        memset(&b, 0, sizeof(b));
        int64_t work = static_cast<int64_t>(nshard) * rcap;
        for (int i = 0; i < m; ++i) {
            HA_REQUIRE(plan_ws[at + i] && n[at + i] >= 0 && rowmap[at + i] && posmap[at + i],
                       "shard_frames_pack_batch: bad batch %d", at + i);
            PlanPtrs p = plan_layout(const_cast<void *>(plan_ws[at + i]), n[at + i] > 0 ? n[at + i] : 1);
            b.hdr[i] = p.hdr; b.uniq[i] = p.uniq; b.inverse[i] = p.inverse;
            b.n[i] = static_cast<int>(n[at + i]);
            b.rowmap[i] = rowmap[at + i];
            b.posmap[i] = posmap[at + i];
            work = n[at + i] > work ? n[at + i] : work;
        }
        int blocks = static_cast<int>((work + 255) / 256);
        blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
        hipLaunchKernelGGL(shard_pack_frames_batch_kernel, dim3(blocks, m), dim3(256), 0, as_stream(stream), b, st, nshard,
                           static_cast<int>(rcap), static_cast<size_t>(frame_stride),
                           send + static_cast<size_t>(at) * (2 + static_cast<size_t>(rcap)));
        HA_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int ha_shard_frames_unpack_batch(const int32_t *recv, int count, int nshard, int64_t rcap, int64_t frame_stride,
                                            uint32_t *const *keys_fixed, int32_t *const *state, ha_stream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (recv && keys_fixed && state)), "shard_frames_unpack_batch: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && rcap >= 1 && static_cast<int64_t>(nshard) * rcap < (1ll << 30) &&
                   frame_stride >= count * (2 + rcap), "shard_frames_unpack_batch: bad sizes");
    for (int at = 0; at < count; at += kFrameBatchMax) {
        const int m = count - at < kFrameBatchMax ? count - at : kFrameBatchMax;
        FrameUnpackBatch b;
        memset(&b, 0, sizeof(b));
        for (int i = 0; i < m; ++i) {
            HA_REQUIRE(keys_fixed[at + i] && state[at + i], "shard_frames_unpack_batch: bad batch %d", at + i);
            b.keys_fixed[i] = keys_fixed[at + i];
            b.state[i] = state[at + i];
        }
        long long want = (static_cast<long long>(nshard) * rcap + 255) / 256;
        const int blocks = want < 1 ? 1 : (want > 64 ? 64 : static_cast<int>(want));
        hipLaunchKernelGGL(shard_unpack_frames_batch_kernel, dim3(blocks, m), dim3(256), 0, as_stream(stream),
                           recv + static_cast<size_t>(at) * (2 + static_cast<size_t>(rcap)), nshard, static_cast<int>(rcap),
                           static_cast<size_t>(frame_stride), b);
        HA_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int ha_shard_frames_serve_pull(const float *table, int64_t rows, int64_t width, const int32_t *recv, int nshard,
                                          int64_t rcap, int64_t frame_stride, float *rows_out, uint32_t *keys_fixed,
                                          int32_t *state, ha_stream_t stream) {
    HA_REQUIRE(table && recv && rows_out && keys_fixed && state, "shard_frames_serve_pull: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && rcap >= 1 && static_cast<int64_t>(nshard) * rcap < (1ll << 30) &&
                   rows >= 0 && width >= 1 && width < (1 << 30) && frame_stride >= 2 + rcap,
               "shard_frames_serve_pull: bad sizes");
    const unsigned blocks = static_cast<unsigned>((static_cast<int64_t>(nshard) * rcap + 3) / 4);
    const bool vec = width % 4 == 0 && reinterpret_cast<uintptr_t>(table) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(rows_out) % 16 == 0;
    if (vec)
        hipLaunchKernelGGL(shard_serve_pull_frames_kernel<4>, dim3(blocks), dim3(256), 0, as_stream(stream), table,
                           static_cast<uint64_t>(rows), static_cast<int>(width), recv, nshard, static_cast<int>(rcap),
                           static_cast<size_t>(frame_stride), rows_out, keys_fixed, state);
    else
        hipLaunchKernelGGL(shard_serve_pull_frames_kernel<1>, dim3(blocks), dim3(256), 0, as_stream(stream), table,
                           static_cast<uint64_t>(rows), static_cast<int>(width), recv, nshard, static_cast<int>(rcap),
                           static_cast<size_t>(frame_stride), rows_out, keys_fixed, state);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_shard_frames_unpack(const int32_t *recv, int nshard, int64_t rcap, int64_t frame_stride,
                                      uint32_t *keys_fixed, int32_t *state, ha_stream_t stream) {
    HA_REQUIRE(recv && keys_fixed && state, "shard_frames_unpack: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && rcap >= 1 && static_cast<int64_t>(nshard) * rcap < (1ll << 30) &&
                   frame_stride >= 2 + rcap, "shard_frames_unpack: bad sizes");
    long long want = (static_cast<long long>(nshard) * rcap + 255) / 256;
    const int blocks = want < 1 ? 1 : (want > 256 ? 256 : static_cast<int>(want));
    hipLaunchKernelGGL(shard_unpack_frames_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), recv, nshard,
                       static_cast<int>(rcap), static_cast<size_t>(frame_stride), keys_fixed, state);
    HA_LAUNCH_CHECK();
    return 0;
}

// Owner side of a sparse push in one call: index plan of the received shard-local keys (the W sorted
// lists concatenated in rank order) + `row = (row + v_a) + v_b ...` in list order -- the server `+=`
// of PSHandler::serve(SparsePush), PSFHandle.h:130-164, made deterministic (rank order).
extern "C" int ha_shard_serve_push(float *table, int64_t rows, int64_t width, const uint32_t *keys,
                                   int64_t n, const float *values, void *plan_ws, ha_stream_t stream) {
    HA_REQUIRE(n >= 0, "shard_serve_push: bad n");
    if (n == 0)
        return 0;
    HA_REQUIRE(table && keys && values && plan_ws, "shard_serve_push: null pointer");
    // sort, then the fused apply + finish launch (lr = -1 turns `acc - lr*v` into `acc + v` bit for bit)
    if (ha_plan_sort_u32keys(keys, n, plan_ws, 32, stream))
        return -1;
    return ha_sgd_apply_finish(table, rows, width, plan_ws, n, values, -1.0f, stream);
}

// Owner side of a FRAMED push: keys_fixed = nshard lists of rcap slots in rank order (ha_shard_frames_unpack), values
// = the received row frames.  As ha_shard_serve_push, with the sort replaced by the merge of the sorted lists.
extern "C" int ha_shard_frames_serve_push(float *table, int64_t rows, int64_t width, const uint32_t *keys_fixed, int nshard,
                                          int64_t rcap, const float *values, void *plan_ws, ha_stream_t stream) {
    HA_REQUIRE(table && keys_fixed && values && plan_ws, "shard_frames_serve_push: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && rcap >= 1 && static_cast<int64_t>(nshard) * rcap < (1ll << 30),
               "shard_frames_serve_push: bad sizes");
    const int64_t n = static_cast<int64_t>(nshard) * rcap;
    const size_t lds = static_cast<size_t>(n) * 4;
    PlanPtrs p = plan_layout(plan_ws, n);
    if (nshard == 1)       // one list: it IS the merged order, its keys are distinct
        return shard_frames_apply(table, rows, width, keys_fixed, nullptr, n, values, as_stream(stream));
    if (lds > (size_t(64) << 10))     // lists that do not fit 64 KiB of LDS: the general sort
        return ha_shard_serve_push(table, rows, width, keys_fixed, n, values, plan_ws, stream);
    HA_ALLOW_LDS(shard_merge_rank_kernel, lds);
    hipLaunchKernelGGL(shard_merge_rank_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), lds,
                       as_stream(stream), keys_fixed, nshard, static_cast<int>(rcap), p.sorted, p.perm);
    HA_LAUNCH_CHECK();
    return shard_frames_apply(table, rows, width, p.sorted, p.perm, n, values, as_stream(stream));
}

// ---- sized frames: the same routing, the rows travel in exchanges sized by the REAL counts -----------------------------
// The key frames above are routed a block of batches ahead of the steps that use them, so the per-owner counts of a
// batch -- how many unique keys this rank names of every owner (send counts) and how many every peer names of this
// rank's range (receive counts) -- are on the device a block early; the unpack launch copies them to pinned host
// memory, and the host sizes the two row exchanges of the step by them without a stall (the reference's messages
// carry exactly U_s keys and U_s x d floats per server: PSAgent::vecPullSparse / vecPushSparse, ps-lite/include/ps/
// worker/PSAgent.h:167-172,217-226; ps/psf/sparse.h:9-32).  Rows are laid out COMPACTLY in rank order, and the keys
// a rank owns itself never enter a row exchange:
//   meta (int32[2 + 2W], one copy on the device, one in pinned host memory):
//       [0] any rank overflowed its key frames  [1] keys received  [2 + g] send count g  [2 + W + g] receive count g
//   pull  owner: rows_out[roff(g) + j] = table[key j of peer g], g != self, roff = running sum of the receive counts of
//         the peers before g (self excluded)  ->  all-to-all with the real splits  ->  expand by posmap:
//         posmap[i] = 0x80000000 | shard-local key   (position i names a key of THIS rank: read from the table itself)
//                   = compact index of its unique key among the unique keys of the other owners, in key order
//                   = 0xFFFFFFFF (a zero row) beyond the frames
//   push  reduce by rowmap into ONE buffer of (2W + 1) * rcap rows:  A = [0, W rcap) rows for the other owners, compact
//         in key order;  S = [W rcap, (W + 1) rcap) this rank's own keys;  B = [(W + 1) rcap, ...) the rows received
//         (compact, rank order);  all-to-all A -> B with the real splits; owner: merge of the W key lists in rank
//         order, values of list g at B + roff(g) + j, of this rank's own list at S + j.
namespace ha {

__device__ __forceinline__ void shard_pack_sized_maps(
    const PlanHeader *__restrict__ hdr, const uint32_t *__restrict__ uniq, const int32_t *__restrict__ inverse, int n,
    const ShardStarts &st, int nshard, int self, int rcap, int32_t *__restrict__ rowmap, int32_t *__restrict__ posmap,
    int32_t *__restrict__ meta_dev, int32_t *__restrict__ meta_host, const int *s_off, int s_flag) {
    const int U = n > 0 ? static_cast<int>(hdr->n_unique) : 0;
    if (blockIdx.x == 0 && threadIdx.x < static_cast<unsigned>(nshard)) {
        const int c = s_off[threadIdx.x + 1] - s_off[threadIdx.x];
        meta_dev[2 + threadIdx.x] = c;
        meta_host[2 + threadIdx.x] = c;
    }
    const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    const int own0 = s_off[self], own1 = s_off[self + 1];
    for (int u = tid; u < U; u += stride) {
        int r = -1;
        if (!s_flag)
            r = u < own0 ? u : u < own1 ? nshard * rcap + (u - own0) : u - (own1 - own0);
        rowmap[u] = r;
    }
    for (int i = tid; i < n; i += stride) {
        const int u = inverse[i];
        uint32_t e = 0xFFFFFFFFu;
        if (!s_flag) {
            const uint32_t local = uniq[u] - st.start[self];       // (the last owner also gets the keys beyond the table)
            e = u < own0 ? static_cast<uint32_t>(u)
                         : u < own1 ? (local < 0x7FFFFFFFu ? (0x80000000u | local) : 0xFFFFFFFFu)
                                    : static_cast<uint32_t>(u - (own1 - own0));
        }
        posmap[i] = static_cast<int32_t>(e);
    }
}

struct FrameSizedBatch {
    int32_t *meta_dev[kFrameBatchMax], *meta_host[kFrameBatchMax];
};
__global__ __launch_bounds__(256) void shard_pack_sized_batch_kernel(const FramePackBatch b, const FrameSizedBatch m,
                                                                     ShardStarts st, int nshard, int self, int rcap,
                                                                     size_t fw, int32_t *__restrict__ send) {
    __shared__ int s_off[kMaxShards + 1];
    __shared__ int s_flag;
    const int i = blockIdx.y;
    const int n = b.n[i];
    const int U = n > 0 ? static_cast<int>(b.hdr[i]->n_unique) : 0;
    if (threadIdx.x <= static_cast<unsigned>(nshard)) {
        const uint32_t target = st.start[threadIdx.x];
        int lo = 0, hi = U;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (b.uniq[i][mid] < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        s_off[threadIdx.x] = static_cast<int>(threadIdx.x) == nshard ? U : lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int f = 0;
        for (int g = 0; g < nshard; ++g)
            f |= (s_off[g + 1] - s_off[g]) > rcap;
        s_flag = f;
    }
    __syncthreads();
    int32_t *snd = send + static_cast<size_t>(i) * (2 + static_cast<size_t>(rcap));
    if (blockIdx.x == 0 && threadIdx.x < static_cast<unsigned>(nshard)) {
        snd[threadIdx.x * fw] = s_off[threadIdx.x + 1] - s_off[threadIdx.x];
        snd[threadIdx.x * fw + 1] = s_flag;
    }
    const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    const int slots = nshard * rcap;
    for (int p = tid; p < slots; p += stride) {
        const int g = p / rcap, j = p - g * rcap;
        const bool live = j < s_off[g + 1] - s_off[g];
        snd[g * fw + 2 + j] = static_cast<int32_t>(live ? b.uniq[i][s_off[g] + j] - st.start[g] : kNoKey);
    }
    shard_pack_sized_maps(b.hdr[i], b.uniq[i], b.inverse[i], n, st, nshard, self, rcap, b.rowmap[i], b.posmap[i],
                          m.meta_dev[i], m.meta_host[i], s_off, s_flag);
}

__global__ __launch_bounds__(256) void shard_unpack_sized_batch_kernel(const int32_t *__restrict__ recv, int nshard,
                                                                       int rcap, size_t fw, const FrameUnpackBatch b,
                                                                       const FrameSizedBatch m) {
    const int i = blockIdx.y;
    const int32_t *rcv = recv + static_cast<size_t>(i) * (2 + static_cast<size_t>(rcap));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int f = 0, total = 0;
        for (int g = 0; g < nshard; ++g) {
            const int c = rcv[g * fw];
            f |= (rcv[g * fw + 1] != 0) | (c > rcap) | (c < 0);
            const int cc = c < 0 ? 0 : (c > rcap ? rcap : c);
            total += cc;
            m.meta_dev[i][2 + nshard + g] = cc;
            m.meta_host[i][2 + nshard + g] = cc;
        }
        m.meta_dev[i][0] = f;
        m.meta_dev[i][1] = total;
        m.meta_host[i][1] = total;
        __threadfence_system();
        m.meta_host[i][0] = f;          // the word the host polls last
    }
    const int slots = nshard * rcap, stride = gridDim.x * 256;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < slots; p += stride) {
        const int g = p / rcap, j = p - g * rcap;
        const int c = rcv[g * fw];
        b.keys_fixed[i][p] = j < c ? static_cast<uint32_t>(rcv[g * fw + 2 + j]) : kNoKey;
    }
}

// running sum of the receive counts of the peers before g, this rank's own list excluded (wave-uniform loop, W <= 64)
__device__ __forceinline__ int shard_roff(const int32_t *__restrict__ meta, int nshard, int self, int g) {
    int off = 0;
    for (int t = 0; t < g; ++t)
        off += t == self ? 0 : meta[2 + nshard + t];
    return off;
}

// Owner side of a sized pull: one wave per live slot of the OTHER ranks' key lists (PSHandler::serve(SparsePull),
// PSFHandle.h:101-128); this rank's own keys are read by the expand straight from the table.
template <int VEC>
__global__ __launch_bounds__(256) void shard_serve_pull_sized_kernel(
    const float *__restrict__ table, uint64_t rows, int width, const uint32_t *__restrict__ keys_fixed, int nshard, int self,
    int rcap, const int32_t *__restrict__ meta, float *__restrict__ rows_out) {
    const int lane = threadIdx.x & 63;
    const int slots = nshard * rcap;
    const int p = uniform(static_cast<int>(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (p >= slots)
        return;
    const int g = p / rcap, j = p - g * rcap;
    if (g == self || j >= meta[2 + nshard + g])
        return;
    const uint32_t key = keys_fixed[p];
    float *dst = rows_out + static_cast<uint64_t>(shard_roff(meta, nshard, self, g) + j) * static_cast<uint64_t>(width);
    const bool ok = key < rows;
    const float *src = table + static_cast<uint64_t>(ok ? key : 0u) * static_cast<uint64_t>(width);
    if (VEC == 4) {
        for (int col = 4 * lane; col < width; col += 256) {
            const float4v v = ok ? ld4(src + col) : float4v{0.f, 0.f, 0.f, 0.f};
            st4_nt(dst + col, v);
        }
    } else {
        for (int col = lane; col < width; col += 64)
            dst[col] = ok ? src[col] : 0.f;
    }
}

// Merge of the W received key lists (each ascending over its first count_g slots) in rank order, as
// shard_merge_rank_kernel, over the LIVE slots only and with the value row of every entry: sorted[rank] = key,
// perm[rank] = row of the push buffer that holds the entry's values.  The lists sit in LDS where they fit (lds != 0).
__global__ __launch_bounds__(256) void shard_merge_rank_sized_kernel(const uint32_t *__restrict__ keys, int nshard, int self,
                                                                     int rcap, const int32_t *__restrict__ meta, int lds,
                                                                     uint32_t *__restrict__ sorted, int32_t *__restrict__ perm) {
    extern __shared__ uint32_t s_keys[];
    const int slots = nshard * rcap;
    if (lds) {
        for (int q = threadIdx.x; q < slots; q += 256)
            s_keys[q] = keys[q];
        __syncthreads();
    }
    const uint32_t *kk = lds ? s_keys : keys;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= slots)
        return;
    const int s = p / rcap, j = p - s * rcap;
    if (j >= meta[2 + nshard + s])
        return;
    const uint32_t key = kk[p];
    int rank = j;
    for (int t = 0; t < nshard; ++t) {
        if (t == s)
            continue;
        const uint32_t *list = kk + t * rcap;
        int lo = 0, hi = meta[2 + nshard + t];
        if (t < s) {            // upper bound: equal keys of earlier ranks come first
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (list[mid] <= key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
        } else {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (list[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
        }
        rank += lo;
    }
    sorted[rank] = key;
    perm[rank] = s == self ? nshard * rcap + j : (nshard + 1) * rcap + shard_roff(meta, nshard, self, s) + j;
}

}  // namespace ha

extern "C" int ha_shard_frames_pack_batch_sized(const void *const *plan_ws, const int64_t *n, int count,
                                                const int64_t *starts_host, int nshard, int self, int64_t rcap,
                                                int64_t frame_stride, int32_t *send, int32_t *const *rowmap,
                                                int32_t *const *posmap, int32_t *const *meta_dev,
                                                int32_t *const *meta_host, ha_stream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (plan_ws && n && send && rowmap && posmap && meta_dev && meta_host)) &&
                   starts_host, "shard_frames_pack_batch_sized: null pointer");
    HA_REQUIRE(rcap >= 1 && static_cast<int64_t>(2 * nshard + 1) * rcap < (1ll << 30) && frame_stride >= count * (2 + rcap) &&
                   self >= 0 && self < nshard, "shard_frames_pack_batch_sized: bad sizes");
    ShardStarts st;
    if (shard_starts_of(starts_host, nshard, &st, "shard_frames_pack_batch_sized"))
        return -1;
    HA_REQUIRE(starts_host[self + 1] - starts_host[self] < (1ll << 31), "shard_frames_pack_batch_sized: a shard holds at most 2^31 rows");
    for (int at = 0; at < count; at += kFrameBatchMax) {
        const int m = count - at < kFrameBatchMax ? count - at : kFrameBatchMax;
        FramePackBatch b;
        FrameSizedBatch ms;
        memset(&b, 0, sizeof(b));
        memset(&ms, 0, sizeof(ms));
        int64_t work = static_cast<int64_t>(nshard) * rcap;
        for (int i = 0; i < m; ++i) {
            HA_REQUIRE(plan_ws[at + i] && n[at + i] >= 0 && rowmap[at + i] && posmap[at + i] && meta_dev[at + i] &&
                           meta_host[at + i], "shard_frames_pack_batch_sized: bad batch %d", at + i);
            PlanPtrs p = plan_layout(const_cast<void *>(plan_ws[at + i]), n[at + i] > 0 ? n[at + i] : 1);
            b.hdr[i] = p.hdr; b.uniq[i] = p.uniq; b.inverse[i] = p.inverse;
            b.n[i] = static_cast<int>(n[at + i]);
            b.rowmap[i] = rowmap[at + i];
            b.posmap[i] = posmap[at + i];
            ms.meta_dev[i] = meta_dev[at + i];
            ms.meta_host[i] = meta_host[at + i];
            work = n[at + i] > work ? n[at + i] : work;
        }
        int blocks = static_cast<int>((work + 255) / 256);
        blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
        hipLaunchKernelGGL(shard_pack_sized_batch_kernel, dim3(blocks, m), dim3(256), 0, as_stream(stream), b, ms, st, nshard,
                           self, static_cast<int>(rcap), static_cast<size_t>(frame_stride),
                           send + static_cast<size_t>(at) * (2 + static_cast<size_t>(rcap)));
        HA_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int ha_shard_frames_unpack_batch_sized(const int32_t *recv, int count, int nshard, int64_t rcap,
                                                  int64_t frame_stride, uint32_t *const *keys_fixed,
                                                  int32_t *const *meta_dev, int32_t *const *meta_host,
                                                  ha_stream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (recv && keys_fixed && meta_dev && meta_host)),
               "shard_frames_unpack_batch_sized: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && rcap >= 1 && static_cast<int64_t>(nshard) * rcap < (1ll << 30) &&
                   frame_stride >= count * (2 + rcap), "shard_frames_unpack_batch_sized: bad sizes");
    for (int at = 0; at < count; at += kFrameBatchMax) {
        const int m = count - at < kFrameBatchMax ? count - at : kFrameBatchMax;
        FrameUnpackBatch b;
        FrameSizedBatch ms;
        memset(&b, 0, sizeof(b));
        memset(&ms, 0, sizeof(ms));
        for (int i = 0; i < m; ++i) {
            HA_REQUIRE(keys_fixed[at + i] && meta_dev[at + i] && meta_host[at + i], "shard_frames_unpack_batch_sized: bad batch %d",
                       at + i);
            b.keys_fixed[i] = keys_fixed[at + i];
            ms.meta_dev[i] = meta_dev[at + i];
            ms.meta_host[i] = meta_host[at + i];
        }
        long long want = (static_cast<long long>(nshard) * rcap + 255) / 256;
        const int blocks = want < 1 ? 1 : (want > 64 ? 64 : static_cast<int>(want));
        hipLaunchKernelGGL(shard_unpack_sized_batch_kernel, dim3(blocks, m), dim3(256), 0, as_stream(stream),
                           recv + static_cast<size_t>(at) * (2 + static_cast<size_t>(rcap)), nshard, static_cast<int>(rcap),
                           static_cast<size_t>(frame_stride), b, ms);
        HA_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int ha_shard_sized_serve_pull(const float *table, int64_t rows, int64_t width, const uint32_t *keys_fixed,
                                         int nshard, int self, int64_t rcap, const int32_t *meta_dev, float *rows_out,
                                         ha_stream_t stream) {
    HA_REQUIRE(table && keys_fixed && meta_dev && rows_out, "shard_sized_serve_pull: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && self >= 0 && self < nshard && rcap >= 1 &&
                   static_cast<int64_t>(nshard) * rcap < (1ll << 30) && rows >= 0 && width >= 1 && width < (1 << 30),
               "shard_sized_serve_pull: bad sizes");
    if (nshard == 1)
        return 0;      // nobody else to serve
    const unsigned blocks = static_cast<unsigned>((static_cast<int64_t>(nshard) * rcap + 3) / 4);
    const bool vec = width % 4 == 0 && reinterpret_cast<uintptr_t>(table) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(rows_out) % 16 == 0;
    if (vec)
        hipLaunchKernelGGL(shard_serve_pull_sized_kernel<4>, dim3(blocks), dim3(256), 0, as_stream(stream), table,
                           static_cast<uint64_t>(rows), static_cast<int>(width), keys_fixed, nshard, self,
                           static_cast<int>(rcap), meta_dev, rows_out);
    else
        hipLaunchKernelGGL(shard_serve_pull_sized_kernel<1>, dim3(blocks), dim3(256), 0, as_stream(stream), table,
                           static_cast<uint64_t>(rows), static_cast<int>(width), keys_fixed, nshard, self,
                           static_cast<int>(rcap), meta_dev, rows_out);
    HA_LAUNCH_CHECK();
    return 0;
}

// Owner side of a sized push: `total` = the keys received (host-known: the sum of the receive counts), push_buf = the
// (2W + 1) * rcap-row buffer described above, plan_ws: ha_plan_bytes(nshard * rcap) of scratch.
extern "C" int ha_shard_sized_serve_push(float *table, int64_t rows, int64_t width, const uint32_t *keys_fixed, int nshard,
                                         int self, int64_t rcap, const int32_t *meta_dev, int64_t total,
                                         const float *push_buf, void *plan_ws, ha_stream_t stream) {
    HA_REQUIRE(table && keys_fixed && meta_dev && push_buf && plan_ws, "shard_sized_serve_push: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards && self >= 0 && self < nshard && rcap >= 1 &&
                   static_cast<int64_t>(2 * nshard + 1) * rcap < (1ll << 30) && total >= 0 &&
                   total <= static_cast<int64_t>(nshard) * rcap, "shard_sized_serve_push: bad sizes");
    if (total == 0)
        return 0;
    const int64_t slots = static_cast<int64_t>(nshard) * rcap;
    if (nshard == 1)       // this rank's own list IS the merged order: distinct keys, values in region S
        return shard_frames_apply(table, rows, width, keys_fixed, nullptr, total,
                                  push_buf + static_cast<size_t>(rcap) * static_cast<size_t>(width), as_stream(stream));
    PlanPtrs p = plan_layout(plan_ws, slots);
    const size_t lds = static_cast<size_t>(slots) * 4 <= (size_t(64) << 10) ? static_cast<size_t>(slots) * 4 : 0;
    hipLaunchKernelGGL(shard_merge_rank_sized_kernel, dim3(static_cast<unsigned>((slots + 255) / 256)), dim3(256), lds,
                       as_stream(stream), keys_fixed, nshard, self, static_cast<int>(rcap), meta_dev, lds ? 1 : 0, p.sorted,
                       p.perm);
    HA_LAUNCH_CHECK();
    return shard_frames_apply(table, rows, width, p.sorted, p.perm, total, push_buf, as_stream(stream));
}

// ---- owner side of the cache protocol over a sharded / host-resident store --------------------------------
// kSyncEmbedding (ps-lite/src/PSFhandle_embedding.cc:30-64): for every requested (key, client version) the
// server answers "pull" when the client has no data (version -1) or lags by more than the bound, and then
// returns its version and the row.  One workgroup takes the decisions and packs the answer positions in
// request order (running prefix over chunks of 1024 requests); a second launch copies the pulled rows,
// one wave per row.
namespace ha {

__global__ __launch_bounds__(1024) void store_sync_decide_kernel(
    const long long *__restrict__ srv_ver, long long rows, const uint32_t *__restrict__ keys,
    const long long *__restrict__ versions, long long m, long long bound, int32_t *__restrict__ pull,
    int32_t *__restrict__ idx, long long *__restrict__ ver_out, long long *__restrict__ count) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    if (m <= 8 * 1024) {
        // up to 8192 entries: thread t owns the consecutive entries Kt .. Kt+K-1 -- keys and cached versions in one batch
        // of loads, the server versions in a second, ONE block scan of the per-thread pull counts (a chunk of 1024 entries at
        // a time it was two dependent trips and a scan per chunk: 11 us for the 5,720-entry request of the cold-tier batch)
        const int K = static_cast<int>((m + 1023) >> 10);
        const long long j0 = static_cast<long long>(tid) * K;
        uint32_t kk[8];
        long long vv[8], sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long j = min(j0 + i, m - 1);
            kk[i] = keys[j];
            vv[i] = versions[j];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            sv[i] = srv_ver[kk[i] < static_cast<unsigned long long>(rows) ? kk[i] : 0];
        uint32_t local = 0, pm = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool on = i < K && j0 + i < m && kk[i] < static_cast<unsigned long long>(rows);
            const bool p = on && (vv[i] == -1 || sv[i] - vv[i] > bound);
            pm |= p ? (1u << i) : 0u;
            local += p ? 1u : 0u;
        }
        uint32_t x = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o)
                x += y;
        }
        if (lane == 63)
            s_w[w] = x;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 16; ++k) {
            if (k < w)
                woff += s_w[k];
            tot += s_w[k];
        }
        uint32_t at = woff + x - local;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < K && j0 + i < m) {
                const bool on = kk[i] < static_cast<unsigned long long>(rows);
                pull[j0 + i] = (pm >> i) & 1u;
                idx[j0 + i] = static_cast<int32_t>(at);
                ver_out[j0 + i] = on ? sv[i] : 0;
                at += (pm >> i) & 1u;
            }
        }
        if (tid == 0)
            *count = tot;
        return;
    }
    if (tid == 0)
        s_base = 0;
    __syncthreads();
    for (long long c0 = 0; c0 < m; c0 += 1024) {
        const long long j = c0 + tid;
        int p = 0;
        long long sv = 0;
        if (j < m) {
            const uint32_t k = keys[j];
            if (k < static_cast<unsigned long long>(rows)) {
                sv = srv_ver[k];
                const long long v = versions[j];
                p = (v == -1 || sv - v > bound) ? 1 : 0;
            }
        }
        // exclusive scan of p over the chunk
        uint32_t x = static_cast<uint32_t>(p);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o)
                x += y;
        }
        if (lane == 63)
            s_w[w] = x;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 16; ++k) {
            if (k < w)
                woff += s_w[k];
            tot += s_w[k];
        }
        const uint32_t base = s_base;
        if (j < m) {
            pull[j] = p;
            idx[j] = static_cast<int32_t>(base + woff + x - static_cast<uint32_t>(p));
            ver_out[j] = sv;
        }
        __syncthreads();
        if (tid == 0)
            s_base = base + tot;
        __syncthreads();
    }
    if (tid == 0)
        *count = s_base;
}

__global__ __launch_bounds__(256) void store_sync_rows_kernel(
    const float *__restrict__ table, int width, const uint32_t *__restrict__ keys,
    const int32_t *__restrict__ pull, const int32_t *__restrict__ idx, long long m,
    float *__restrict__ rows_out) {
    const int lane = lane_id();
    const long long nw = static_cast<long long>(gridDim.x) * 4;
    for (long long j = blockIdx.x * 4ll + (threadIdx.x >> 6); j < m; j += nw) {
        if (!pull[j])
            continue;
        const float *src = table + static_cast<uint64_t>(keys[j]) * static_cast<uint64_t>(width);
        float *dst = rows_out + static_cast<uint64_t>(idx[j]) * static_cast<uint64_t>(width);
        if ((width & 3) == 0 && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(rows_out)) & 15) == 0) {
            for (int c = lane * 4; c < width; c += kWave * 4)
                st4(dst + c, ld4(src + c));
        } else {
            for (int c = lane; c < width; c += kWave)
                dst[c] = src[c];
        }
    }
}

__global__ __launch_bounds__(256) void store_add_versions_kernel(long long *__restrict__ srv_ver, long long rows,
                                                                 const uint32_t *__restrict__ keys,
                                                                 const int32_t *__restrict__ updates, long long m) {
    for (long long j = blockIdx.x * 256ll + threadIdx.x; j < m; j += gridDim.x * 256ll) {
        const uint32_t k = keys[j];
        if (k < static_cast<unsigned long long>(rows))   // integer adds: any order gives the same sum
            atomicAdd(reinterpret_cast<unsigned long long *>(srv_ver + k),
                      static_cast<unsigned long long>(static_cast<long long>(updates[j])));
    }
}

// pushEmbedding's server side (PSFhandle_embedding.cc:23-27) for a list whose live keys are pairwise DISTINCT: one wave per
// entry, row += gradient row, version += update count; no sort, no dedup, no order to keep.
__global__ __launch_bounds__(256) void store_push_distinct_kernel(float *__restrict__ table, long long *__restrict__ srv_ver,
                                                                  long long rows, int width,
                                                                  const uint32_t *__restrict__ keys,
                                                                  const int32_t *__restrict__ updates,
                                                                  const float *__restrict__ grad_rows, long long m) {
    const int lane = threadIdx.x & 63;
    for (long long j = blockIdx.x * 4ll + (threadIdx.x >> 6); j < m; j += gridDim.x * 4ll) {
        const uint32_t k = keys[j];
        if (k >= static_cast<unsigned long long>(rows))       // 0xFFFFFFFF = not pushed
            continue;
        float *row = table + static_cast<long long>(k) * width;
        const float *g = grad_rows + j * width;
        for (int c = lane; c < width; c += 64)
            row[c] = __fadd_rn(row[c], g[c]);
        if (lane == 0)
            srv_ver[k] += updates[j];
    }
}

}  // namespace ha

extern "C" int ha_store_push_distinct(float *table, int64_t *server_versions, int64_t rows, int64_t width,
                                      const uint32_t *keys, const int32_t *updates, const float *grad_rows, int64_t m,
                                      ha_stream_t stream) {
    HA_REQUIRE(m >= 0 && rows >= 0 && width >= 1 && width < (1ll << 30), "store_push_distinct: bad sizes");
    if (m == 0)
        return 0;
    HA_REQUIRE(table && server_versions && keys && updates && grad_rows, "store_push_distinct: null pointer");
    unsigned blocks = static_cast<unsigned>((m + 3) / 4);
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(ha::store_push_distinct_kernel, dim3(blocks), dim3(256), 0, ha::as_stream(stream), table,
                       reinterpret_cast<long long *>(server_versions), (long long)rows, (int)width, keys, updates,
                       grad_rows, (long long)m);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_store_serve_sync(const float *table, const int64_t *server_versions, int64_t rows,
                                   int64_t width, const uint32_t *keys, const int64_t *versions, int64_t m,
                                   int64_t bound, int32_t *pull, int32_t *idx, int64_t *ver_out,
                                   float *rows_out, int64_t *count_dev, void *scan_ws, ha_stream_t stream) {
    (void)scan_ws;
    HA_REQUIRE(m >= 0 && rows >= 0 && width >= 1 && width < (1ll << 30), "store_serve_sync: bad sizes");
    HA_REQUIRE(count_dev, "store_serve_sync: null count");
    if (m == 0) {
        HA_CHECK_HIP(hipMemsetAsync(count_dev, 0, 8, as_stream(stream)));
        return 0;
    }
    HA_REQUIRE(table && server_versions && keys && versions && pull && idx && ver_out && rows_out,
               "store_serve_sync: null pointer");
    hipLaunchKernelGGL(store_sync_decide_kernel, dim3(1), dim3(1024), 0, as_stream(stream),
                       reinterpret_cast<const long long *>(server_versions), (long long)rows, keys,
                       reinterpret_cast<const long long *>(versions), (long long)m, (long long)bound, pull, idx,
                       reinterpret_cast<long long *>(ver_out), reinterpret_cast<long long *>(count_dev));
    unsigned blocks = static_cast<unsigned>((m + 3) / 4);
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(store_sync_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), table, (int)width,
                       keys, pull, idx, (long long)m, rows_out);
    HA_LAUNCH_CHECK();
    return 0;
}

namespace ha {
__global__ __launch_bounds__(256) void store_count_valid_kernel(const uint32_t *__restrict__ keys, long long m,
                                                                long long rows, long long *__restrict__ acc) {
    long long c = 0;
    for (long long j = blockIdx.x * 256ll + threadIdx.x; j < m; j += gridDim.x * 256ll)
        c += keys[j] < static_cast<unsigned long long>(rows) ? 1 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        c += __shfl_xor(c, o, 64);
    if (lane_id() == 0 && c)
        atomicAdd(reinterpret_cast<unsigned long long *>(acc), static_cast<unsigned long long>(c));
}
}  // namespace ha

// *acc += number of keys below `rows` (entries a store applies; 0xFFFFFFFF marks "not pushed"): device-side
// traffic accounting of the stores, no host synchronisation
extern "C" int ha_store_count_valid(const uint32_t *keys, int64_t m, int64_t rows, int64_t *acc, ha_stream_t stream) {
    HA_REQUIRE(m >= 0 && rows >= 0 && acc, "store_count_valid: bad arguments");
    if (m == 0)
        return 0;
    HA_REQUIRE(keys, "store_count_valid: null keys");
    unsigned blocks = static_cast<unsigned>((m + 255) / 256);
    if (blocks > 256)
        blocks = 256;
    hipLaunchKernelGGL(store_count_valid_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), keys, (long long)m,
                       (long long)rows, reinterpret_cast<long long *>(acc));
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_store_add_versions(int64_t *server_versions, int64_t rows, const uint32_t *keys,
                                     const int32_t *updates, int64_t m, ha_stream_t stream) {
    HA_REQUIRE(m >= 0 && rows >= 0, "store_add_versions: bad sizes");
    if (m == 0)
        return 0;
    HA_REQUIRE(server_versions && keys && updates, "store_add_versions: null pointer");
    unsigned blocks = static_cast<unsigned>((m + 255) / 256);
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(store_add_versions_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<long long *>(server_versions), (long long)rows, keys, updates,
                       (long long)m);
    HA_LAUNCH_CHECK();
    return 0;
}

// ---- the sharded step as ONE native call ------------------------------------------------------------------------------------
// What PSAgent::vecPullSparse / vecPushSparse do inside one C++ call each (ps-lite/include/ps/worker/PSAgent.h:124-237: dedup,
// route, send, wait, scatter): the step's launches and its two row exchanges enqueued on the caller's stream by the library,
// from the per-owner counts the routing left in pinned host memory a block ago -- no Python between them.
//   pull half  owner gather of the rows the other ranks name (ha_shard_sized_serve_pull) -> rows exchange (ha_xchg_rows) ->
//              positions from the own shard / the received rows (ha_gather2_u32map);
//   push half  occurrence-ordered reduce of scale * values by unique key into the push buffer (ha_apply_mapped) -> rows
//              exchange -> merge of the W lists in rank order + apply (ha_shard_sized_serve_push);
//   world 1    the lookup + ONE reduce-and-add launch (ha_push_apply_scaled_finished); xchg may be NULL.
// ha_shard_step = both halves for a caller that holds the batch's gradients when it asks for its rows (a benchmark loop, a
// pipeline with staleness); ha_shard_steps = `count` such steps by one call.  slot->counts_host: the pinned words of the
// batch's routing ([2 + g] = unique keys this rank names at owner g, [2 + world + g] = unique keys rank g names here; the
// routing block's event must have completed: the caller waited for it when it checked the overflow word [0]).
extern "C" int ha_shard_step_pull(const float *table, int64_t rows, int64_t width, const ha_shard_slot *slot, void *xchg,
                                  float *pull_send, float *pull_recv, int64_t pull_rows, float *out, ha_stream_t stream) {
    HA_REQUIRE(table && slot && slot->world >= 1 && slot->world <= 1024 && slot->rank >= 0 && slot->rank < slot->world,
               "ha_shard_step_pull: bad arguments");
    const int w = slot->world, r = slot->rank;
    if (w > 1) {
        HA_REQUIRE(xchg && slot->counts_host && pull_send && pull_recv, "ha_shard_step_pull: world > 1 needs the exchange, the "
                   "counts and the row buffers");
        int64_t ins[1024], outs[1024];
        for (int g = 0; g < w; ++g) {
            ins[g] = g == r ? 0 : slot->counts_host[2 + w + g];       // rows this rank serves to peer g
            outs[g] = g == r ? 0 : slot->counts_host[2 + g];          // rows owner g sends back
        }
        if (ha_shard_sized_serve_pull(table, rows, width, slot->keys_fixed, w, r, slot->rcap, slot->meta_dev, pull_send, stream))
            return -1;
        if (ha_xchg_rows(xchg, pull_send, ins, pull_recv, outs, width, stream))
            return -1;
    }
    if (slot->n > 0) {
        HA_REQUIRE(out && slot->posmap, "ha_shard_step_pull: null output");
        if (ha_gather2_u32map(table, rows, pull_recv, pull_rows, width, slot->posmap, slot->n, out, stream))
            return -1;
    }
    return 0;
}

extern "C" int ha_shard_step_push(float *table, int64_t rows, int64_t width, const ha_shard_slot *slot, void *xchg,
                                  float *push_buf, int64_t push_rows, const uint8_t *zero_flags, const float *values,
                                  float scale, ha_stream_t stream) {
    HA_REQUIRE(table && slot && slot->world >= 1 && slot->world <= 1024 && slot->rank >= 0 && slot->rank < slot->world,
               "ha_shard_step_push: bad arguments");
    const int w = slot->world, r = slot->rank;
    if (w == 1) {      // nobody else pushes: reduce + server add of the own keys in one launch
        if (slot->n > 0)
            return ha_push_apply_scaled_finished(table, rows, width, slot->plan_ws, slot->n, values, scale, stream);
        return 0;
    }
    HA_REQUIRE(xchg && slot->counts_host && push_buf && zero_flags && slot->owner_plan_ws,
               "ha_shard_step_push: world > 1 needs the exchange, the counts, the push buffer and the owner's plan workspace");
    int64_t ins[1024], outs[1024], total = 0, n_in = 0;
    for (int g = 0; g < w; ++g) {
        ins[g] = g == r ? 0 : slot->counts_host[2 + g];               // reduced rows for owner g
        outs[g] = g == r ? 0 : slot->counts_host[2 + w + g];          // rows peer g pushes to this rank
        total += slot->counts_host[2 + w + g];
        n_in += ins[g];
    }
    (void)n_in;
    if (slot->n > 0 &&
        ha_apply_mapped(push_buf, push_rows, width, slot->plan_ws, slot->n, values, -scale, slot->rowmap, nullptr, zero_flags, stream))
        return -1;
    float *recv = push_buf + static_cast<int64_t>(w + 1) * slot->rcap * width;      // region B: rows received
    if (ha_xchg_rows(xchg, push_buf, ins, recv, outs, width, stream))
        return -1;
    return ha_shard_sized_serve_push(table, rows, width, slot->keys_fixed, w, r, slot->rcap, slot->meta_dev, total, push_buf,
                                     slot->owner_plan_ws, stream);
}

extern "C" int ha_shard_step(float *table, int64_t rows, int64_t width, const ha_shard_slot *slot, void *xchg, float *pull_send,
                             float *pull_recv, int64_t pull_rows, float *push_buf, int64_t push_rows, const uint8_t *zero_flags,
                             float *out, const float *values, float scale, ha_stream_t stream) {
    if (ha_shard_step_pull(table, rows, width, slot, xchg, pull_send, pull_recv, pull_rows, out, stream))
        return -1;
    return ha_shard_step_push(table, rows, width, slot, xchg, push_buf, push_rows, zero_flags, values, scale, stream);
}

extern "C" int ha_shard_steps(float *table, int64_t rows, int64_t width, int64_t count, const ha_shard_slot *const *slots,
                              void *xchg, float *pull_send, float *pull_recv, int64_t pull_rows, float *push_buf,
                              int64_t push_rows, const uint8_t *zero_flags, float *const *outs, const float *const *values,
                              float scale, ha_stream_t stream) {
    HA_REQUIRE(count >= 0 && (count == 0 || (slots && outs && values)), "ha_shard_steps: null pointer");
    for (int64_t k = 0; k < count; ++k)
        if (ha_shard_step(table, rows, width, slots[k], xchg, pull_send, pull_recv, pull_rows, push_buf, push_rows, zero_flags,
                          outs[k], values[k], scale, stream))
            return -1;
    return 0;
}
