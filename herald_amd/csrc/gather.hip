// Forward sparse gather: out[i,:] = table[(size_t)ids[i],:].
//
// Replaces cpu_EmbeddingLookup (reference src/dnnl_ops/EmbeddingLookup.cpp:16-35: OpenMP
// memcpy per id) and embedding_lookup_kernel (src/ops/EmbeddingLookup.cu:3-14: one
// thread per id looping over the row, i.e. lanes stride `width` floats apart).
//
// MI355X layout: the output is viewed as a flat array of 16-byte vectors; lane e copies
// vector e, so a wavefront moves 1 KiB of one (or two adjacent) output row(s) per
// instruction, and the matching table reads are 1 KiB contiguous pieces of the source
// row(s).  Every thread keeps UNROLL independent 16-byte loads in flight before its
// first store.  Output stores are non-temporal: the gathered rows are consumed by a
// different kernel (the dense network) and should not evict table rows from L2.
//
// Algorithmic bytes per id: 4 (id) + 4*width (row read) + 4*width (row write).
#include "common.h"

namespace ha {

}  // namespace ha
#include "gather_dev.h"
namespace ha {

template <typename IdT, int UNROLL, int NV_SHIFT>
__global__ __launch_bounds__(256) void gather_vec4_kernel(
    const float *__restrict__ table, uint64_t rows, uint32_t nv,
    const IdT *__restrict__ ids, uint64_t total_vec, float *__restrict__ out) {
    gather_vec4_body<IdT, UNROLL, NV_SHIFT, 256>(table, rows, nv, ids, total_vec, out, blockIdx.x);
}

// Any width (not a multiple of 4, or unaligned buffers): one float per lane.
template <typename IdT>
__global__ __launch_bounds__(256) void gather_scalar_kernel(
    const float *__restrict__ table, uint64_t rows, uint32_t width,
    const IdT *__restrict__ ids, uint64_t total, float *__restrict__ out) {
    uint64_t e = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (; e < total; e += stride) {
        const uint64_t i = e / width;
        const uint64_t c = e - i * width;
        const uint64_t r = id_to_row<IdT>(ids[i]);
        out[e] = r < rows ? table[r * width + c] : 0.f;
    }
}

template <typename IdT>
static int gather_launch(const float *table, int64_t rows, int64_t width,
                         const IdT *ids, int64_t n, float *out,
                         hipStream_t stream) {
    HA_REQUIRE(rows >= 0 && width >= 1 && n >= 0,
               "gather: bad sizes rows=%ld width=%ld n=%ld", (long)rows,
               (long)width, (long)n);
    if (n == 0)
        return 0;
    HA_REQUIRE(table && ids && out, "gather: null pointer");
    const bool vec_ok = (width % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(table) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (!vec_ok) {
        const uint64_t total = static_cast<uint64_t>(n) * width;
        uint64_t blocks = (total + 255) / 256;
        if (blocks > 16384)
            blocks = 16384;
        hipLaunchKernelGGL(gather_scalar_kernel<IdT>, dim3((unsigned)blocks),
                           dim3(256), 0, stream, table, (uint64_t)rows,
                           (uint32_t)width, ids, total, out);
        HA_LAUNCH_CHECK();
        return 0;
    }
    const uint32_t nv = static_cast<uint32_t>(width / 4);
    const uint64_t total_vec = static_cast<uint64_t>(n) * nv;
    constexpr int UNROLL = 4;
    const uint64_t blocks64 = (total_vec + 256 * UNROLL - 1) / (256 * UNROLL);
    HA_REQUIRE(blocks64 < (1ull << 31), "gather: batch too large");
    const dim3 grid(static_cast<unsigned>(blocks64)), block(256);
#define HA_GATHER_CASE(S)                                                      \
    hipLaunchKernelGGL((gather_vec4_kernel<IdT, UNROLL, S>), grid, block, 0,   \
                       stream, table, (uint64_t)rows, nv, ids, total_vec, out)
    switch (nv) {
    case 4: HA_GATHER_CASE(2); break;     // width 16
    case 8: HA_GATHER_CASE(3); break;     // width 32
    case 16: HA_GATHER_CASE(4); break;    // width 64
    case 32: HA_GATHER_CASE(5); break;    // width 128
    case 64: HA_GATHER_CASE(6); break;    // width 256
    case 128: HA_GATHER_CASE(7); break;   // width 512
    case 256: HA_GATHER_CASE(8); break;   // width 1024
    default: HA_GATHER_CASE(-1); break;
    }
#undef HA_GATHER_CASE
    HA_LAUNCH_CHECK();
    return 0;
}


// new_values[(int)ids[i], :] = values[i, :]  (indexedslices2dense_kernel,
// reference src/ops/OptimizersSparse.cu:233-246; ids are deduplicated by the caller)
template <typename IdT>
__global__ __launch_bounds__(256) void scatter_rows_kernel(
    const float *__restrict__ values, const IdT *__restrict__ ids,
    uint64_t total, uint32_t width, uint64_t rows, float *__restrict__ dst) {
    uint64_t e = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (; e < total; e += stride) {
        const uint64_t i = e / width;
        const uint64_t c = e - i * width;
        const uint64_t r = id_to_row<IdT>(ids[i]);
        if (r < rows)
            dst[r * width + c] = values[e];
    }
}

}  // namespace ha

extern "C" int ha_scatter_rows_f32ids(const float *values, const float *ids,
                                      int64_t n, int64_t width, float *dst,
                                      int64_t rows, ha_stream_t stream) {
    HA_REQUIRE(n >= 0 && width >= 1 && rows >= 0, "scatter_rows: bad sizes");
    if (n == 0)
        return 0;
    HA_REQUIRE(values && ids && dst, "scatter_rows: null pointer");
    const uint64_t total = static_cast<uint64_t>(n) * width;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 16384)
        blocks = 16384;
    hipLaunchKernelGGL(ha::scatter_rows_kernel<float>, dim3((unsigned)blocks),
                       dim3(256), 0, ha::as_stream(stream), values, ids, total,
                       (uint32_t)width, (uint64_t)rows, dst);
    HA_LAUNCH_CHECK();
    return 0;
}

namespace ha {
__global__ __launch_bounds__(256) void scale_kernel(float *__restrict__ x, uint64_t n, float s) {
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (; i < n; i += stride)
        x[i] = __fmul_rn(x[i], s);
}
}  // namespace ha

// values[:] = values * scale -- `input_val.values[:] = input_val.values.asnumpy() * self.learning_rate`
// of ParameterServerCommunicateOp._mult_lr_sparse_cpu (python/hetu/gpu_ops/ParameterServerCommunicate.py:58-59)
extern "C" int ha_scale_f32(float *values, int64_t n, float scale, ha_stream_t stream) {
    HA_REQUIRE(n >= 0 && (n == 0 || values), "scale: bad arguments");
    if (n == 0)
        return 0;
    uint64_t blocks = (static_cast<uint64_t>(n) + 255) / 256;
    if (blocks > 16384)
        blocks = 16384;
    hipLaunchKernelGGL(ha::scale_kernel, dim3((unsigned)blocks), dim3(256), 0, ha::as_stream(stream), values,
                       (uint64_t)n, scale);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_gather_f32ids(const float *table, int64_t rows, int64_t width,
                                const float *ids, int64_t n, float *out,
                                ha_stream_t stream) {
    return ha::gather_launch<float>(table, rows, width, ids, n, out,
                                    ha::as_stream(stream));
}

extern "C" int ha_gather_u64ids(const float *table, int64_t rows, int64_t width,
                                const uint64_t *ids, int64_t n, float *out,
                                ha_stream_t stream) {
    return ha::gather_launch<uint64_t>(table, rows, width, ids, n, out,
                                       ha::as_stream(stream));
}

// out[i,:] = (map[i] & 0x80000000) ? table[map[i] & 0x7FFFFFFF,:] : recv[map[i],:]; an index beyond its array reads as
// a zero row.  The expand of a sized pull (shard.hip): positions whose key THIS rank owns read the table itself, the
// others the rows received from their owners -- PSAgent::vecPullSparse's scatter of the returned rows to all positions
// (ps-lite/include/ps/psf/sparse.h:17-31).  Same lane mapping as the forward gather (flat 16-byte vectors).
namespace ha {
template <int UNROLL>
__global__ __launch_bounds__(256) void gather2_vec4_kernel(const float *__restrict__ table, uint64_t rows,
                                                           const float *__restrict__ recv, uint64_t recv_rows, uint32_t nv,
                                                           const uint32_t *__restrict__ map, uint64_t total_vec,
                                                           float *__restrict__ out) {
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * (256 * UNROLL) + threadIdx.x;
    const uint64_t last = total_vec - 1;
    float4v v[UNROLL];
    bool ok[UNROLL];
    uint32_t m[UNROLL];
    uint64_t col[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        uint64_t e = base + static_cast<uint64_t>(u) * 256;
        e = e < last ? e : last;
        const uint64_t i = e / nv;
        col[u] = e - i * nv;
        m[u] = map[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const bool local = (m[u] & 0x80000000u) != 0;
        const uint64_t r = m[u] & 0x7FFFFFFFu;
        ok[u] = r < (local ? rows : recv_rows);
        const float *src = local ? table : recv;
        v[u] = ld4(src + ((ok[u] ? r : 0) * nv + col[u]) * 4u);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint64_t e = base + static_cast<uint64_t>(u) * 256;
        if (e < total_vec)
            st4_nt(out + e * 4u, ok[u] ? v[u] : float4v{0.f, 0.f, 0.f, 0.f});
    }
}
__global__ __launch_bounds__(256) void gather2_scalar_kernel(const float *__restrict__ table, uint64_t rows,
                                                             const float *__restrict__ recv, uint64_t recv_rows, uint32_t width,
                                                             const uint32_t *__restrict__ map, uint64_t total,
                                                             float *__restrict__ out) {
    uint64_t e = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (; e < total; e += stride) {
        const uint64_t i = e / width, c = e - i * width;
        const uint32_t m = map[i];
        const bool local = (m & 0x80000000u) != 0;
        const uint64_t r = m & 0x7FFFFFFFu;
        out[e] = r < (local ? rows : recv_rows) ? (local ? table : recv)[r * width + c] : 0.f;
    }
}
}  // namespace ha

extern "C" int ha_gather2_u32map(const float *table, int64_t rows, const float *recv, int64_t recv_rows, int64_t width,
                                 const uint32_t *map, int64_t n, float *out, ha_stream_t stream) {
    HA_REQUIRE(rows >= 0 && recv_rows >= 0 && width >= 1 && n >= 0, "gather2: bad sizes");
    if (n == 0)
        return 0;
    HA_REQUIRE(map && out && (table || rows == 0) && (recv || recv_rows == 0), "gather2: null pointer");
    // a side that has no rows is never dereferenced beyond its clamped row 0: give it a valid address
    if (!table) table = recv;
    if (!recv) recv = table;
    HA_REQUIRE(table != nullptr, "gather2: no source at all");
    const bool vec_ok = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(table) % 16 == 0) &&
                        (reinterpret_cast<uintptr_t>(recv) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (!vec_ok) {
        const uint64_t total = static_cast<uint64_t>(n) * width;
        uint64_t blocks = (total + 255) / 256;
        if (blocks > 16384)
            blocks = 16384;
        hipLaunchKernelGGL(ha::gather2_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, ha::as_stream(stream), table,
                           (uint64_t)rows, recv, (uint64_t)recv_rows, (uint32_t)width, map, total, out);
        HA_LAUNCH_CHECK();
        return 0;
    }
    const uint32_t nv = static_cast<uint32_t>(width / 4);
    const uint64_t total_vec = static_cast<uint64_t>(n) * nv;
    constexpr int UNROLL = 4;
    const uint64_t blocks64 = (total_vec + 256 * UNROLL - 1) / (256 * UNROLL);
    HA_REQUIRE(blocks64 < (1ull << 31), "gather2: batch too large");
    hipLaunchKernelGGL((ha::gather2_vec4_kernel<UNROLL>), dim3(static_cast<unsigned>(blocks64)), dim3(256), 0,
                       ha::as_stream(stream), table, (uint64_t)rows, recv, (uint64_t)recv_rows, nv, map, total_vec, out);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_gather_u32keys(const float *table, int64_t rows,
                                 int64_t width, const uint32_t *keys, int64_t n,
                                 float *out, ha_stream_t stream) {
    return ha::gather_launch<uint32_t>(table, rows, width, keys, n, out,
                                       ha::as_stream(stream));
}


// A gate for a stream: the launch returns at once, the stream behind it does not move on until the HOST has written a
// non-zero value to *flag (pinned, device-visible host memory) -- or ~2 s have passed.  Lets a caller enqueue a whole
// sequence (events, graphs) behind the gate and release it when everything is queued, so that the device never sits
// inside the sequence waiting for the host to enqueue its next piece (bench.py: the timed region of a short run).
namespace ha {
__global__ void stream_gate_kernel(const volatile uint32_t *flag) {
    for (int i = 0; i < (1 << 21); ++i) {
        if (__hip_atomic_load(const_cast<const uint32_t *>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
            break;
        __builtin_amdgcn_s_sleep(32);
    }
}
}  // namespace ha

// Measurement aid (tools/occupy_ab.py): `wgs` workgroups of `threads` threads and `lds_bytes` of LDS each that do nothing
// but hold their wave slots for `ticks` ticks of the 100 MHz clock -- what a latency-bound preparation kernel takes from a
// launch running beside it, without its memory traffic.
namespace ha {
__global__ void occupy_kernel(unsigned long long ticks) {
    extern __shared__ uint32_t s_occ[];
    if (threadIdx.x == 0)
        s_occ[0] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
}
}  // namespace ha
extern "C" int ha_debug_occupy(int64_t wgs, int64_t threads, int64_t lds_bytes, int64_t ticks, ha_stream_t stream) {
    HA_REQUIRE(wgs >= 1 && threads >= 64 && threads <= 1024 && lds_bytes >= 4 && lds_bytes <= 160 * 1024 && ticks >= 0,
               "ha_debug_occupy: bad arguments");
    HA_ALLOW_LDS(ha::occupy_kernel, static_cast<size_t>(lds_bytes));
    hipLaunchKernelGGL(ha::occupy_kernel, dim3(static_cast<unsigned>(wgs)), dim3(static_cast<unsigned>(threads)),
                       static_cast<size_t>(lds_bytes), ha::as_stream(stream), static_cast<unsigned long long>(ticks));
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int ha_stream_gate(const uint32_t *flag, ha_stream_t stream) {
    HA_REQUIRE(flag != nullptr, "ha_stream_gate: null flag");
    hipLaunchKernelGGL(ha::stream_gate_kernel, dim3(1), dim3(1), 0, ha::as_stream(stream), flag);
    HA_LAUNCH_CHECK();
    return 0;
}
