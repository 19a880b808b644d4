// Sparse optimizers on DEDUPLICATED indexed slices: AdaGrad / Adam / AdamW.
//
// Reference: adagrad_sparse_update / adam_sparse_update / adamw_sparse_update,
// src/ops/OptimizersSparse.cu:331-349, 391-416, 457-484 -- one thread per element, the element's row
// taken from indices[ind].  The callers deduplicate first (python/hetu/gpu_links/OptimizerLink.py:60,78,95:
// grad.deduplicate(stream)), so rows are unique and the read-modify-write needs no atomics.
// Here: flat 16-byte vectors over the [n, width] gradient, state rows addressed through the index;
// arithmetic in the reference's expression order (no FMA contraction), sqrtf/division as written.
// Floating-point tolerance vs the numpy oracle: 1e-5 (tests/test_gpu_optim.py), as in the reference's
// own tests (tests/test_optimizer.py:117-198).
#include "common.h"

namespace ha {

enum OptKind { kAdaGrad = 0, kAdam = 1, kAdamW = 2 };

struct OptArgs {
    float lr, eps, beta1, beta2, beta1t, beta2t, weight_decay;
};

template <int KIND>
__device__ __forceinline__ void opt_step(float &p, float g, float &s1, float &s2, const OptArgs &a) {
    if (KIND == kAdaGrad) {
        const float acc = s1 + g * g;
        s1 = acc;
        p = p - a.lr * g / (sqrtf(acc) + a.eps);
    } else {
        float m = a.beta1 * s1 + (1.f - a.beta1) * g;
        float v = a.beta2 * s2 + (1.f - a.beta2) * g * g;
        s1 = m;
        s2 = v;
        m = m / (1.f - a.beta1t);
        v = v / (1.f - a.beta2t);
        if (KIND == kAdam) {
            p = p - a.lr * m / (sqrtf(v) + a.eps);
        } else {
            const float update = m / (sqrtf(v) + a.eps);
            p = p - a.lr * (update + a.weight_decay * p);
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void sparse_opt_kernel(float *__restrict__ param, uint64_t rows,
                                                         const float *__restrict__ ids,
                                                         const float *__restrict__ grads, uint64_t total,
                                                         uint32_t width, float *__restrict__ s1,
                                                         float *__restrict__ s2, OptArgs a) {
    uint64_t e = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (; e < total; e += stride) {
        const uint64_t i = e / width, c = e - i * width;
        const uint64_t r = static_cast<uint64_t>(f32_to_key(ids[i]));
        if (r >= rows)
            continue;
        const uint64_t off = r * width + c;
        float p = param[off], x1 = s1[off], x2 = KIND == kAdaGrad ? 0.f : s2[off];
        opt_step<KIND>(p, grads[e], x1, x2, a);
        param[off] = p;
        s1[off] = x1;
        if (KIND != kAdaGrad)
            s2[off] = x2;
    }
}

template <int KIND>
static int opt_launch(DLArrayHandle param, const DLArrayHandle ids, const DLArrayHandle grads, DLArrayHandle s1,
                      DLArrayHandle s2, OptArgs a, DLStreamHandle sh, const char *name) {
    HA_REQUIRE(param && ids && grads && s1 && (KIND == kAdaGrad || s2), "%s: null array", name);
    HA_REQUIRE(param->ndim == 2, "%s: param must be 2-D", name);
    const int64_t n = dl_numel(ids), width = param->shape[1];
    HA_REQUIRE(dl_numel(grads) == n * width, "%s: grad_values size mismatch", name);
    if (n == 0)
        return 0;
    const uint64_t total = static_cast<uint64_t>(n) * width;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536)
        blocks = 65536;
    hipLaunchKernelGGL(sparse_opt_kernel<KIND>, dim3((unsigned)blocks), dim3(256), 0, dl_stream(sh),
                       static_cast<float *>(param->data), (uint64_t)param->shape[0],
                       static_cast<const float *>(ids->data), static_cast<const float *>(grads->data), total,
                       (uint32_t)width, static_cast<float *>(s1->data),
                       s2 ? static_cast<float *>(s2->data) : nullptr, a);
    HA_LAUNCH_CHECK();
    return 0;
}

}  // namespace ha

using namespace ha;

extern "C" int AdaGradOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                            const DLArrayHandle grad_values, DLArrayHandle acc, float lr,
                                            float eps, DLStreamHandle stream_handle) {
    OptArgs a{lr, eps, 0, 0, 0, 0, 0};
    return opt_launch<kAdaGrad>(param, grad_indices, grad_values, acc, nullptr, a, stream_handle,
                                "AdaGradOptimizerSparseUpdate");
}

extern "C" int AdamOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                         const DLArrayHandle grad_values, DLArrayHandle expavg,
                                         DLArrayHandle expavgsq, float lr, float beta1, float beta2,
                                         float beta1t, float beta2t, float eps, DLStreamHandle stream_handle) {
    OptArgs a{lr, eps, beta1, beta2, beta1t, beta2t, 0};
    return opt_launch<kAdam>(param, grad_indices, grad_values, expavg, expavgsq, a, stream_handle,
                             "AdamOptimizerSparseUpdate");
}

extern "C" int AdamWOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                          const DLArrayHandle grad_values, DLArrayHandle expavg,
                                          DLArrayHandle expavgsq, float lr, float beta1, float beta2,
                                          float beta1t, float beta2t, float eps, float weight_decay,
                                          DLStreamHandle stream_handle) {
    OptArgs a{lr, eps, beta1, beta2, beta1t, beta2t, weight_decay};
    return opt_launch<kAdamW>(param, grad_indices, grad_values, expavg, expavgsq, a, stream_handle,
                              "AdamWOptimizerSparseUpdate");
}
