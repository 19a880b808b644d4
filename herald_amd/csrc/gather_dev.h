// Device body of the forward gather, shared by gather.hip and fused.hip (semantics: gather.hip).
#pragma once
#include "common.h"

namespace ha {

template <typename IdT>
__device__ __forceinline__ uint64_t id_to_row(IdT v);
template <>
__device__ __forceinline__ uint64_t id_to_row<float>(float v) {
    return static_cast<uint64_t>(f32_to_key(v));
}
template <>
__device__ __forceinline__ uint64_t id_to_row<uint64_t>(uint64_t v) {
    return v;
}
template <>
__device__ __forceinline__ uint64_t id_to_row<uint32_t>(uint32_t v) {
    return v;
}

// NV_SHIFT >= 0: vectors per row is 1 << NV_SHIFT (shift/mask instead of div/mod).
template <typename IdT, int UNROLL, int NV_SHIFT, int BLOCK>
__device__ __forceinline__ void gather_vec4_body(
    const float *__restrict__ table, uint64_t rows, uint32_t nv,
    const IdT *__restrict__ ids, uint64_t total_vec, float *__restrict__ out,
    uint32_t vblock) {
    const uint64_t base =
        static_cast<uint64_t>(vblock) * (BLOCK * UNROLL) + threadIdx.x;
    // Loads are branch-free (clamped addresses): hipcc puts an s_waitcnt vmcnt(0) behind every load
    // that sits in its own exec-masked branch, which would serialise the UNROLL row reads.
    const uint64_t last = total_vec - 1;
    uint64_t row[UNROLL], col[UNROLL];
    IdT idv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        uint64_t e = base + static_cast<uint64_t>(u) * BLOCK;
        e = e < last ? e : last;
        uint64_t i;
        if (NV_SHIFT >= 0) {
            i = e >> NV_SHIFT;
            col[u] = e & ((1u << NV_SHIFT) - 1u);
        } else {
            i = e / nv;
            col[u] = e - i * nv;
        }
        idv[u] = ids[i];
    }
    float4v v[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint64_t r = id_to_row<IdT>(idv[u]);
        ok[u] = r < rows;
        row[u] = ok[u] ? r : 0;
        v[u] = ld4(table + (row[u] * nv + col[u]) * 4u);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint64_t e = base + static_cast<uint64_t>(u) * BLOCK;
        if (e < total_vec)
            st4_nt(out + e * 4u, ok[u] ? v[u] : float4v{0.f, 0.f, 0.f, 0.f});
    }
}

}  // namespace ha
