// Shared host/device helpers for the herald_amd HIP sources (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>

#include "../../include/herald_amd.h"

namespace ha {

constexpr int kWave = 64;  // CDNA wavefront

// ---- error plumbing -------------------------------------------------------
void set_error(const char *fmt, ...);
bool g_err_is_empty();

#define HA_CHECK_HIP(expr)                                                     \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            ::ha::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,      \
                            hipGetErrorString(_e));                            \
            return -1;                                                         \
        }                                                                      \
    } while (0)

#define HA_REQUIRE(cond, ...)                                                  \
    do {                                                                       \
        if (!(cond)) {                                                         \
            ::ha::set_error(__VA_ARGS__);                                      \
            return -1;                                                         \
        }                                                                      \
    } while (0)

#define HA_LAUNCH_CHECK() HA_CHECK_HIP(hipGetLastError())

// Dynamic LDS beyond 64 KiB (gfx950 offers 160 KiB per workgroup) has to be allowed per kernel.
#define HA_ALLOW_LDS(kernel, bytes)                                                              \
    do {                                                                                         \
        if ((bytes) > 65536)                                                                     \
            HA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),             \
                                             hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                             static_cast<int>(bytes)));                          \
    } while (0)

// "Once per DEVICE" guard for per-device function attributes (hipFuncSetAttribute applies to the current device
// only).  run(f): f() is executed once per device (f returns 0 on success); a thread that arrives while another is
// still inside f WAITS for it -- it must not launch a kernel whose LDS attribute is not set yet --, and a failed
// run is retried by the next caller.
struct DeviceOnce {
    std::atomic<unsigned long long> done[4];   // 256 devices
    std::mutex mu;
    template <typename F>
    int run(F f) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess)
            d = 0;
        const unsigned long long bit = 1ull << (d & 63);
        std::atomic<unsigned long long> &word = done[(d >> 6) & 3];
        if (word.load(std::memory_order_acquire) & bit)
            return 0;
        std::lock_guard<std::mutex> lk(mu);
        if (word.load(std::memory_order_acquire) & bit)
            return 0;
        const int rc = f();
        if (rc == 0)
            word.fetch_or(bit, std::memory_order_release);
        return rc;
    }
};

static inline hipStream_t as_stream(ha_stream_t s) {
    return reinterpret_cast<hipStream_t>(s);
}
static inline hipStream_t dl_stream(DLStreamHandle h) {
    // reference: *(cudaStream_t *)stream_handle->handle (src/ops/EmbeddingLookup.cu:46)
    if (h == nullptr || h->handle == nullptr)
        return nullptr;
    return *reinterpret_cast<hipStream_t *>(h->handle);
}
static inline int64_t dl_numel(const DLArray *a) {
    int64_t n = 1;
    for (int i = 0; i < a->ndim; ++i)
        n *= a->shape[i];
    return n;
}

static inline size_t align_up(size_t x, size_t a) {
    return (x + a - 1) / a * a;
}

// ---- device helpers -------------------------------------------------------
// (size_t)f of the reference (src/dnnl_ops/EmbeddingLookup.cpp:31) for the ids a
// table can hold: truncation toward zero, ids < 2^32.
__device__ __forceinline__ uint32_t f32_to_key(float f) {
    return static_cast<uint32_t>(f);
}

__device__ __forceinline__ int lane_id() {
    return threadIdx.x & (kWave - 1);
}

// Wave-uniform value -> SGPR.
__device__ __forceinline__ int uniform(int v) {
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) {
    return static_cast<uint32_t>(
        __builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}

typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4v ld4(const float *p) {
    return *reinterpret_cast<const float4v *>(p);
}
__device__ __forceinline__ void st4(float *p, float4v v) {
    *reinterpret_cast<float4v *>(p) = v;
}
__device__ __forceinline__ void st4_nt(float *p, float4v v) {
    __builtin_nontemporal_store(v, reinterpret_cast<float4v *>(p));
}

// Device-coherent (agent scope, `sc1`) accesses for data another workgroup of the SAME launch reads or
// wrote: the store is written through the XCD's L2, the load bypasses the CU's L1
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  The
// compiler does not count inline-asm memory instructions in its vmcnt bookkeeping: a 16-byte load's
// result may only be used behind wait_loads(), and stores are followed by an explicit s_waitcnt
// (signal_done, scatter_dev.h) before anything that publishes them.
__device__ __forceinline__ void st4_sc1(float *p, float4v v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st1_sc1(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4v ld4_sc1_async(const float *p) {
    float4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_loads(float4v &a, float4v &b, float4v &c, float4v &d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// Pending table of a batch (step.hip): 2^kPendBits words indexed by a multiplicative hash of the key.
constexpr int kPendBits = 16;
__device__ __forceinline__ uint32_t pend_slot(uint32_t key) {
    return (key * 0x9E3779B1u) >> (32 - kPendBits);
}


// Key table of a batch (step.hip, ha_step_*): open addressing, 2^kTabBits entries {key, first sorted
// position, occurrences - 1, unused}, all-ones = empty; filled by the rank tiles of the launch that sorts
// the batch (32-bit atomics: claim the key, min the position, add the count), read by later launches only.
constexpr int kTabBits = 15;
constexpr uint32_t kTabMask = (1u << kTabBits) - 1u;
constexpr uint32_t kTabEmpty = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t tab_slot(uint32_t key) {
    return (key * 0x9E3779B1u) >> (32 - kTabBits);
}

}  // namespace ha
