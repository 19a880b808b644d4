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

}  // namespace ha

using namespace ha;

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
                       int (*build)(const IdT *, int64_t, void *, ha_stream_t)) {
    HA_REQUIRE(plan_ws && starts_host && meta && local_keys, "shard_route: null pointer");
    HA_REQUIRE(nshard >= 1 && nshard <= kMaxShards, "shard_route: nshard must be in [1,%d]", kMaxShards);
    HA_REQUIRE(n >= 0, "shard_route: bad n");
    if (build(ids, n, plan_ws, stream))
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
                              ha_plan_build_f32ids);
}

extern "C" int ha_shard_route_u64ids(const uint64_t *ids, int64_t n, void *plan_ws,
                                     const int64_t *starts_host, int nshard, int64_t *meta,
                                     uint32_t *local_keys, ha_stream_t stream) {
    return shard_route<uint64_t>(ids, n, plan_ws, starts_host, nshard, meta, local_keys, stream,
                                 ha_plan_build_u64ids);
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
