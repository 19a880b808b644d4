// Library plumbing + the reference-named operator symbols (DLArray / DLStream ABI).
//
// Every extern "C" function below that carries a reference name keeps the reference's argument
// order and meaning (src/common/c_runtime_api.h, cited per function in include/herald_amd.h); the
// body validates shapes the way the reference asserts them and forwards to the ha_* engine entry
// points on the caller's stream.
#include "common.h"

#include <stdarg.h>
#include <mutex>
#include <unordered_map>

namespace ha {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool g_err_is_empty() {
    return g_err[0] == 0;
}

// Per-(device, stream) scratch that the one-call reference-named ops use for their index plan.
// It only grows; a stream serialises its own users, so reuse across calls on one stream is safe.
struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
};
static std::mutex g_scratch_mu;
static std::unordered_map<uint64_t, Scratch> g_scratch;

int scratch_get(hipStream_t stream, size_t bytes, void **out) {
    int dev = 0;
    HA_CHECK_HIP(hipGetDevice(&dev));
    const uint64_t key = (static_cast<uint64_t>(dev) << 56) ^
                         reinterpret_cast<uint64_t>(stream);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    Scratch &s = g_scratch[key];
    if (s.bytes < bytes) {
        if (s.ptr) {
            // the old buffer may still be in use by queued kernels of this stream
            HA_CHECK_HIP(hipStreamSynchronize(stream));
            HA_CHECK_HIP(hipFree(s.ptr));
            s.ptr = nullptr;
            s.bytes = 0;
        }
        size_t want = bytes + bytes / 2;
        if (want < (1u << 20))
            want = 1u << 20;
        HA_CHECK_HIP(hipMalloc(&s.ptr, want));
        s.bytes = want;
    }
    *out = s.ptr;
    return 0;
}

static int check_f32_gpu(const DLArray *a, const char *what) {
    HA_REQUIRE(a != nullptr && a->data != nullptr, "%s: null array", what);
    HA_REQUIRE(a->ctx.device_type == kGPU, "%s: array is not on the GPU (device_type=%d)",
               what, (int)a->ctx.device_type);
    return 0;
}

}  // namespace ha

using namespace ha;

extern "C" const char *ha_version(void) {
    return "herald_amd 0.1 (gfx950)";
}

extern "C" const char *ha_last_error(void) {
    return ha::g_err;
}

extern "C" int ha_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

namespace ha {
void host_map_release();
}

extern "C" int ha_scratch_release(void) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto &kv : g_scratch)
        if (kv.second.ptr)
            (void)hipFree(kv.second.ptr);
    g_scratch.clear();
    host_map_release();
    return 0;
}

// ---------------------------------------------------------------------------
extern "C" int DLGpuEmbeddingLookUp(const DLArrayHandle input,
                                    const DLArrayHandle ids,
                                    DLArrayHandle output,
                                    DLStreamHandle stream_handle) {
    if (check_f32_gpu(input, "DLGpuEmbeddingLookUp(input)") ||
        check_f32_gpu(ids, "DLGpuEmbeddingLookUp(ids)") ||
        check_f32_gpu(output, "DLGpuEmbeddingLookUp(output)"))
        return -1;
    // shape checks of the reference: src/ops/EmbeddingLookup.cu:19-27
    HA_REQUIRE(input->ndim == 2, "DLGpuEmbeddingLookUp: table must be 2-D");
    HA_REQUIRE(output->ndim == ids->ndim + 1,
               "DLGpuEmbeddingLookUp: output.ndim must be ids.ndim + 1");
    for (int i = 0; i < ids->ndim; ++i)
        HA_REQUIRE(output->shape[i] == ids->shape[i],
                   "DLGpuEmbeddingLookUp: output/ids shape mismatch at dim %d", i);
    HA_REQUIRE(output->shape[output->ndim - 1] == input->shape[1],
               "DLGpuEmbeddingLookUp: output width != table width");
    return ha_gather_f32ids(static_cast<const float *>(input->data),
                            input->shape[0], input->shape[1],
                            static_cast<const float *>(ids->data),
                            dl_numel(ids), static_cast<float *>(output->data),
                            dl_stream(stream_handle));
}

// Shared body of the "scatter-add rows into a dense [rows,width] array" ops.
static int scatter_add_rows(float *dst, int64_t rows, int64_t width,
                            const float *ids, int64_t n, const float *vals,
                            hipStream_t stream) {
    if (n == 0)
        return 0;
    void *ws = nullptr;
    if (scratch_get(stream, ha_plan_bytes(n), &ws))
        return -1;
    if (ha_plan_sort_f32ids(ids, n, ws, stream))
        return -1;
    return ha_push_apply(dst, rows, width, ws, n, vals, stream);
}

extern "C" int DLGpuEmbeddingLookUp_Gradient(const DLArrayHandle output_grad,
                                             const DLArrayHandle ids,
                                             DLArrayHandle input_grad,
                                             DLStreamHandle stream_handle) {
    if (check_f32_gpu(output_grad, "DLGpuEmbeddingLookUp_Gradient(output_grad)") ||
        check_f32_gpu(ids, "DLGpuEmbeddingLookUp_Gradient(ids)") ||
        check_f32_gpu(input_grad, "DLGpuEmbeddingLookUp_Gradient(input_grad)"))
        return -1;
    HA_REQUIRE(input_grad->ndim == 2, "DLGpuEmbeddingLookUp_Gradient: input_grad must be 2-D");
    const int64_t rows = input_grad->shape[0], width = input_grad->shape[1];
    const int64_t n = dl_numel(ids);
    HA_REQUIRE(dl_numel(output_grad) == n * width,
               "DLGpuEmbeddingLookUp_Gradient: output_grad size mismatch");
    hipStream_t stream = dl_stream(stream_handle);
    // the reference zeroes the whole dense gradient first (EmbeddingLookup.cu:98-115)
    HA_CHECK_HIP(hipMemsetAsync(input_grad->data, 0,
                                static_cast<size_t>(rows) * width * 4, stream));
    return scatter_add_rows(static_cast<float *>(input_grad->data), rows, width,
                            static_cast<const float *>(ids->data), n,
                            static_cast<const float *>(output_grad->data),
                            stream);
}

extern "C" int IndexedSlicesOneSideAdd(const DLArrayHandle indices,
                                       const DLArrayHandle values,
                                       DLArrayHandle output,
                                       DLStreamHandle stream_handle) {
    if (check_f32_gpu(indices, "IndexedSlicesOneSideAdd(indices)") ||
        check_f32_gpu(values, "IndexedSlicesOneSideAdd(values)") ||
        check_f32_gpu(output, "IndexedSlicesOneSideAdd(output)"))
        return -1;
    HA_REQUIRE(output->ndim == 2, "IndexedSlicesOneSideAdd: output must be 2-D");
    const int64_t rows = output->shape[0], width = output->shape[1];
    const int64_t n = dl_numel(indices);
    HA_REQUIRE(dl_numel(values) == n * width,
               "IndexedSlicesOneSideAdd: values size mismatch");
    return scatter_add_rows(static_cast<float *>(output->data), rows, width,
                            static_cast<const float *>(indices->data), n,
                            static_cast<const float *>(values->data),
                            dl_stream(stream_handle));
}

extern "C" int DeduplicateIndexedSlices(const DLArrayHandle origin,
                                        const DLArrayHandle inverse,
                                        DLArrayHandle compressed,
                                        DLStreamHandle stream_handle) {
    if (check_f32_gpu(origin, "DeduplicateIndexedSlices(origin)") ||
        check_f32_gpu(inverse, "DeduplicateIndexedSlices(inverse)") ||
        check_f32_gpu(compressed, "DeduplicateIndexedSlices(compressed)"))
        return -1;
    HA_REQUIRE(compressed->ndim >= 1, "DeduplicateIndexedSlices: bad compressed");
    const int64_t width = compressed->shape[compressed->ndim - 1];
    const int64_t rows = dl_numel(compressed) / (width ? width : 1);
    const int64_t n = dl_numel(inverse);
    HA_REQUIRE(dl_numel(origin) == n * width,
               "DeduplicateIndexedSlices: origin size mismatch");
    // compressed[inverse[i],:] += origin[i,:]; caller zero-filled it (ndarray.py:547-548)
    return scatter_add_rows(static_cast<float *>(compressed->data), rows, width,
                            static_cast<const float *>(inverse->data), n,
                            static_cast<const float *>(origin->data),
                            dl_stream(stream_handle));
}

extern "C" int ha_scatter_rows_f32ids(const float *values, const float *ids,
                                      int64_t n, int64_t width, float *dst,
                                      int64_t rows, ha_stream_t stream);

extern "C" int IndexedSlices2Dense(const DLArrayHandle values,
                                   const DLArrayHandle indices,
                                   DLArrayHandle new_values,
                                   DLStreamHandle stream_handle) {
    if (check_f32_gpu(values, "IndexedSlices2Dense(values)") ||
        check_f32_gpu(indices, "IndexedSlices2Dense(indices)") ||
        check_f32_gpu(new_values, "IndexedSlices2Dense(new_values)"))
        return -1;
    HA_REQUIRE(new_values->ndim >= 1, "IndexedSlices2Dense: bad new_values");
    const int64_t width = new_values->shape[new_values->ndim - 1];
    const int64_t rows = dl_numel(new_values) / (width ? width : 1);
    const int64_t n = dl_numel(indices);
    HA_REQUIRE(dl_numel(values) == n * width,
               "IndexedSlices2Dense: values size mismatch");
    return ha_scatter_rows_f32ids(static_cast<const float *>(values->data),
                                  static_cast<const float *>(indices->data), n,
                                  width, static_cast<float *>(new_values->data),
                                  rows, dl_stream(stream_handle));
}

extern "C" int SGDOptimizerSparseUpdate(DLArrayHandle param,
                                        const DLArrayHandle grad_indices,
                                        const DLArrayHandle grad_values,
                                        float lr,
                                        DLStreamHandle stream_handle) {
    if (check_f32_gpu(param, "SGDOptimizerSparseUpdate(param)") ||
        check_f32_gpu(grad_indices, "SGDOptimizerSparseUpdate(grad_indices)") ||
        check_f32_gpu(grad_values, "SGDOptimizerSparseUpdate(grad_values)"))
        return -1;
    HA_REQUIRE(param->ndim == 2, "SGDOptimizerSparseUpdate: param must be 2-D");
    const int64_t n = dl_numel(grad_indices);
    HA_REQUIRE(dl_numel(grad_values) == n * param->shape[1],
               "SGDOptimizerSparseUpdate: grad_values size mismatch");
    return ha_sgd_sparse_update_f32ids(
        static_cast<float *>(param->data), param->shape[0], param->shape[1],
        static_cast<const float *>(grad_indices->data), n,
        static_cast<const float *>(grad_values->data), lr,
        dl_stream(stream_handle));
}

// ---- the reference's CPU operator names (src/common/c_runtime_api.h:811-818) -------------------------
// python/hetu/_base.py:8-11,72 feature-probes these two symbols with hasattr and, when present, routes
// EmbeddingLookUp.compute / SGDOptimizer.update of host-context nodes to them (EmbeddingLookUp.py:16-28,
// optimizer.py:204-214).  This library is the GPU engine and has no CPU arithmetic: arrays whose context is
// the GPU are used where they lie; HOST arrays are made visible to the device for the call --
//   * below 64 MiB (ids, gradients, looked-up rows): a device copy, H2D before the kernel, D2H behind it;
//   * from 64 MiB (tables): the host range is page-locked and mapped (hipHostRegister, kept registered until
//     ha_scratch_release: callers pass the same parameter array every step), and the kernels read / write only
//     the rows the ids name across PCIe -- a 69 GB table is never copied;
// the kernels are the ones the GPU-context call runs (null stream, complete on return, as a host operator is).
namespace ha {

constexpr size_t kHostMapFrom = size_t(64) << 20;
static std::mutex g_hostmap_mu;
static std::unordered_map<void *, size_t> g_hostmap;   // registered host ranges (base -> bytes)

static int host_map(void *host, size_t bytes, void **dev) {
    std::lock_guard<std::mutex> lk(g_hostmap_mu);
    auto it = g_hostmap.find(host);
    if (it != g_hostmap.end() && it->second < bytes) {   // the same base with a larger extent: register anew
        HA_CHECK_HIP(hipHostUnregister(host));
        g_hostmap.erase(it);
        it = g_hostmap.end();
    }
    if (it == g_hostmap.end()) {
        HA_CHECK_HIP(hipHostRegister(host, bytes, hipHostRegisterMapped));
        g_hostmap[host] = bytes;
    }
    HA_CHECK_HIP(hipHostGetDevicePointer(dev, host, 0));
    return 0;
}

// Drops the registration of a host array (cpu_* entry points register arrays of >= 64 MiB instead of copying them, and keep
// the registration for the next call).  Call it BEFORE freeing such an array: a later allocation at the same address would
// otherwise be seen through the old mapping.  Unknown pointers are ignored.
extern "C" int ha_host_unmap(void *host) {
    std::lock_guard<std::mutex> lk(g_hostmap_mu);
    auto it = g_hostmap.find(host);
    if (it == g_hostmap.end())
        return 0;
    HA_CHECK_HIP(hipDeviceSynchronize());
    HA_CHECK_HIP(hipHostUnregister(host));
    g_hostmap.erase(it);
    return 0;
}

void host_map_release() {
    std::lock_guard<std::mutex> lk(g_hostmap_mu);
    for (auto &kv : g_hostmap)
        (void)hipHostUnregister(kv.first);
    g_hostmap.clear();
}

// One array of a cpu_* call as the device sees it.
struct DeviceView {
    DLArray arr;            // the caller's array with `data` replaced by a device-visible address, ctx = GPU
    void *copy = nullptr;   // device copy (small host arrays)
    void *host = nullptr;
    size_t bytes = 0;
    ~DeviceView() {
        if (copy)
            (void)hipFree(copy);
    }
    int open(const DLArray *a, bool read, const char *name) {
        HA_REQUIRE(a && a->data, "%s: null array", name);
        arr = *a;
        if (a->ctx.device_type == kGPU)
            return 0;
        HA_REQUIRE(a->ctx.device_type == kCPU, "%s: unknown device_type %d", name, (int)a->ctx.device_type);
        bytes = static_cast<size_t>(dl_numel(a)) * 4;
        host = a->data;
        int dev = 0;
        HA_CHECK_HIP(hipGetDevice(&dev));
        arr.ctx.device_type = kGPU;
        arr.ctx.device_id = dev;
        if (bytes >= kHostMapFrom)
            return host_map(host, bytes, &arr.data);
        HA_CHECK_HIP(hipMalloc(&copy, bytes ? bytes : 4));
        arr.data = copy;
        if (read && bytes)
            HA_CHECK_HIP(hipMemcpyAsync(copy, host, bytes, hipMemcpyHostToDevice, nullptr));
        return 0;
    }
    int close(bool written) {   // behind the kernel, on the null stream
        if (copy && written && bytes)
            HA_CHECK_HIP(hipMemcpyAsync(host, copy, bytes, hipMemcpyDeviceToHost, nullptr));
        return 0;
    }
};

}  // namespace ha

extern "C" int cpu_EmbeddingLookup(const DLArrayHandle in_mat, const DLArrayHandle ids, DLArrayHandle out_mat) {
    DeviceView t, i, o;
    if (t.open(in_mat, true, "cpu_EmbeddingLookup(in_mat)") || i.open(ids, true, "cpu_EmbeddingLookup(ids)") ||
        o.open(out_mat, false, "cpu_EmbeddingLookup(out_mat)"))
        return -1;
    if (DLGpuEmbeddingLookUp(&t.arr, &i.arr, &o.arr, nullptr))
        return -1;
    if (o.close(true))
        return -1;
    HA_CHECK_HIP(hipStreamSynchronize(nullptr));
    return 0;
}

extern "C" int cpu_SGDOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                            const DLArrayHandle grad_values, float lr) {
    DeviceView p, i, g;
    if (p.open(param, true, "cpu_SGDOptimizerSparseUpdate(param)") ||
        i.open(grad_indices, true, "cpu_SGDOptimizerSparseUpdate(grad_indices)") ||
        g.open(grad_values, true, "cpu_SGDOptimizerSparseUpdate(grad_values)"))
        return -1;
    if (SGDOptimizerSparseUpdate(&p.arr, &i.arr, &g.arr, lr, nullptr))
        return -1;
    if (p.close(true))
        return -1;
    HA_CHECK_HIP(hipStreamSynchronize(nullptr));
    return 0;
}

extern "C" int ha_sgd_sparse_update_f32ids(float *table, int64_t rows,
                                           int64_t width, const float *ids,
                                           int64_t n, const float *grads,
                                           float lr, ha_stream_t stream) {
    if (n == 0)
        return 0;
    void *ws = nullptr;
    if (scratch_get(as_stream(stream), ha_plan_bytes(n), &ws))
        return -1;
    if (ha_plan_sort_f32ids(ids, n, ws, stream))
        return -1;
    return ha_sgd_apply(table, rows, width, ws, n, grads, lr, stream);
}
