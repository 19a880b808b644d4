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
