// Sparse optimizers and sparse L2 on indexed slices: AddL2RegularizationSparse,
// Momentum (plain and Nesterov), AdaGrad, Adam, AdamW, Lamb.
//
// Reference: src/ops/OptimizersSparse.cu -- add_l2_regularization_sparse :3-51, momentum / nesterov
// :101-231, adagrad_sparse_update :331-389, adam_sparse_update :391-455, adamw_sparse_update :457-522,
// Lamb :524-722; declarations src/common/c_runtime_api.h:639-700.  There every kernel is one thread per
// element with a 64-bit divide per element and, where ids may repeat (SGD, Momentum), float atomics.
// The callers deduplicate before AdaGrad / Adam / AdamW / Lamb / L2 (python/hetu/gpu_links/
// OptimizerLink.py:15,60,78,95,111) and do NOT before Momentum (:37-49).
//
// Here: ONE WAVE PER ROW.  The row id is wave-uniform (one scalar load), every lane moves 16-byte
// vectors of the gradient row and of the parameter / state rows it addresses (two vectors per lane per
// trip, all loads of a trip issued before the arithmetic), so a 2 KiB row is two 1-KiB wave
// instructions per array and nothing is divided.  Rows are unique for the deduplicated ops, so the
// read-modify-write needs no atomics; Momentum, whose ids repeat, goes through the index plan and the
// occurrence-ordered apply of scatter.hip (deterministic: the reference's atomics pick an arbitrary
// order).  Arithmetic keeps the reference's expression order (library built with -ffp-contract=off);
// tolerance against the numpy oracle 1e-5 as in the reference's own tests
// (tests/test_optimizer.py:117-300), Lamb's two norms are accumulated in double.
#include "optim_dev.h"

namespace ha {

// capi.hip
int scratch_get(hipStream_t stream, size_t bytes, void **out);

template <int KIND>
struct Uses {
    static constexpr bool s1 = KIND == kAdaGrad || KIND == kAdam || KIND == kAdamW || KIND == kLambUpdate;
    static constexpr bool s2 = KIND == kAdam || KIND == kAdamW || KIND == kLambUpdate;
    static constexpr bool grad_in = KIND != kLambStep;
    static constexpr bool grad_out = KIND == kL2;
    static constexpr bool param_out = KIND != kL2 && KIND != kLambUpdate;
    static constexpr bool upd_in = KIND == kLambStep;
    static constexpr bool upd_out = KIND == kLambUpdate;
};

// VEC = 4: width % 4 == 0 and 16-byte aligned arrays; VEC = 1: anything else.
template <int KIND, int VEC>
__global__ __launch_bounds__(256) void sparse_row_kernel(float *__restrict__ param, uint64_t rows,
                                                         const float *__restrict__ ids,
                                                         float *__restrict__ grads, int n, int width,
                                                         float *__restrict__ s1, float *__restrict__ s2,
                                                         OptArgs a) {
    typedef Uses<KIND> U;
    const int lane = lane_id();
    const int wave = static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6);
    const int nwaves = static_cast<int>(gridDim.x) * 4;
    float ratio = 0.f;
    if (KIND == kLambStep)   // norm2(param) / norm2(update), both over the indexed rows (:577)
        ratio = static_cast<float>(sqrt(a.norms[0])) / static_cast<float>(sqrt(a.norms[1]));
    for (int i = wave; i < n; i += nwaves) {
        const uint64_t r = static_cast<uint64_t>(f32_to_key(ids[i]));
        double sp = 0.0, su = 0.0;
        if (r < rows) {   // wave-uniform
            const uint64_t prow = r * static_cast<uint64_t>(width);
            const uint64_t grow = static_cast<uint64_t>(i) * static_cast<uint64_t>(width);
            for (int c0 = 0; c0 < width; c0 += 2 * kWave * VEC) {
                float p[2][VEC], g[2][VEC], x1[2][VEC], x2[2][VEC], up[2][VEC];
                int col[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    col[t] = c0 + (t * kWave + lane) * VEC;
                    const int lc = col[t] < width ? col[t] : 0;   // branch-free loads
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        g[t][k] = 0.f, x1[t][k] = 0.f, x2[t][k] = 0.f, up[t][k] = 0.f;
                    }
                    if (VEC == 4) {
                        *reinterpret_cast<float4v *>(p[t]) = ld4(param + prow + lc);
                        if (U::grad_in)
                            *reinterpret_cast<float4v *>(g[t]) = ld4(grads + grow + lc);
                        if (U::s1)
                            *reinterpret_cast<float4v *>(x1[t]) = ld4(s1 + prow + lc);
                        if (U::s2)
                            *reinterpret_cast<float4v *>(x2[t]) = ld4(s2 + prow + lc);
                        if (U::upd_in)
                            *reinterpret_cast<float4v *>(up[t]) = ld4(a.update + grow + lc);
                    } else {
                        p[t][0] = param[prow + lc];
                        if (U::grad_in)
                            g[t][0] = grads[grow + lc];
                        if (U::s1)
                            x1[t][0] = s1[prow + lc];
                        if (U::s2)
                            x2[t][0] = s2[prow + lc];
                        if (U::upd_in)
                            up[t][0] = a.update[grow + lc];
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (col[t] >= width)
                        continue;
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        opt_step<KIND>(p[t][k], g[t][k], x1[t][k], x2[t][k], a, up[t][k], sp, su, ratio);
                    if (VEC == 4) {
                        if (U::param_out)
                            st4(param + prow + col[t], *reinterpret_cast<float4v *>(p[t]));
                        if (U::grad_out)
                            st4(grads + grow + col[t], *reinterpret_cast<float4v *>(g[t]));
                        if (U::s1)
                            st4(s1 + prow + col[t], *reinterpret_cast<float4v *>(x1[t]));
                        if (U::s2)
                            st4(s2 + prow + col[t], *reinterpret_cast<float4v *>(x2[t]));
                        if (U::upd_out)
                            st4(a.update + grow + col[t], *reinterpret_cast<float4v *>(up[t]));
                    } else {
                        if (U::param_out)
                            param[prow + col[t]] = p[t][0];
                        if (U::grad_out)
                            grads[grow + col[t]] = g[t][0];
                        if (U::s1)
                            s1[prow + col[t]] = x1[t][0];
                        if (U::s2)
                            s2[prow + col[t]] = x2[t][0];
                        if (U::upd_out)
                            a.update[grow + col[t]] = up[t][0];
                    }
                }
            }
        }
        if (KIND == kLambUpdate) {   // this row's share of the two norms, fixed lane order
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                sp += __shfl_xor(sp, o, kWave);
                su += __shfl_xor(su, o, kWave);
            }
            if (lane == 0) {
                a.part_param[i] = sp;
                a.part_update[i] = su;
            }
        }
    }
}

// Lamb: sums of the per-row partials in a fixed order (one workgroup; n is a batch's unique count).
__global__ __launch_bounds__(1024) void lamb_norms_kernel(const double *__restrict__ pp,
                                                          const double *__restrict__ pu, int n,
                                                          double *__restrict__ norms) {
    __shared__ double s_p[16], s_u[16];
    double sp = 0.0, su = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        sp += pp[i];
        su += pu[i];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sp += __shfl_xor(sp, o, kWave);
        su += __shfl_xor(su, o, kWave);
    }
    if (lane_id() == 0) {
        s_p[threadIdx.x >> 6] = sp;
        s_u[threadIdx.x >> 6] = su;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tp = 0.0, tu = 0.0;
        for (int w = 0; w < 16; ++w) {
            tp += s_p[w];
            tu += s_u[w];
        }
        norms[0] = tp;
        norms[1] = tu;
    }
}

// Dense second phase of the reference's momentum update (:122-131 Nesterov, :147-155 plain): it runs over
// the WHOLE parameter array every step, as the reference does.  float4 grid-stride stream.
template <bool NESTEROV>
__global__ __launch_bounds__(256) void momentum_dense_kernel(float *__restrict__ param,
                                                             float *__restrict__ veloc, float momentum,
                                                             uint64_t nvec, uint64_t total) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; e < nvec; e += stride) {
        float4v p = ld4(param + e * 4), v = ld4(veloc + e * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NESTEROV) {
                const float t = momentum * v[k];
                v[k] = t;
                p[k] = p[k] + t;
            } else {
                p[k] = p[k] + v[k];
                v[k] = momentum * v[k];
            }
        }
        st4(param + e * 4, p);
        st4(veloc + e * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3u)) {   // tail elements
        const uint64_t e = (total & ~3ull) + threadIdx.x;
        if (NESTEROV) {
            const float t = momentum * veloc[e];
            veloc[e] = t;
            param[e] = param[e] + t;
        } else {
            param[e] = param[e] + veloc[e];
            veloc[e] = momentum * veloc[e];
        }
    }
}

static bool vec_ok(int64_t width, const void *a, const void *b, const void *c, const void *d, const void *e) {
    auto al = [](const void *p) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    return width % 4 == 0 && al(a) && al(b) && al(c) && al(d) && al(e);
}

template <int KIND>
static int row_launch(float *param, int64_t rows, const float *ids, float *grads, int64_t n, int64_t width,
                      float *s1, float *s2, const OptArgs &a, hipStream_t stream) {
    if (n == 0)
        return 0;
    unsigned blocks = static_cast<unsigned>((n + 3) / 4);
    if (blocks > 16384)
        blocks = 16384;
    if (vec_ok(width, param, grads, s1, s2, a.update))
        hipLaunchKernelGGL((sparse_row_kernel<KIND, 4>), dim3(blocks), dim3(256), 0, stream, param,
                           (uint64_t)rows, ids, grads, (int)n, (int)width, s1, s2, a);
    else
        hipLaunchKernelGGL((sparse_row_kernel<KIND, 1>), dim3(blocks), dim3(256), 0, stream, param,
                           (uint64_t)rows, ids, grads, (int)n, (int)width, s1, s2, a);
    HA_LAUNCH_CHECK();
    return 0;
}

static int check_args(const char *name, const DLArray *param, const DLArray *ids, const DLArray *grads,
                      const DLArray *s1, const DLArray *s2, bool need_s1, bool need_s2, int64_t *n,
                      int64_t *width) {
    HA_REQUIRE(param && ids && grads && (!need_s1 || s1) && (!need_s2 || s2), "%s: null array", name);
    HA_REQUIRE(param->data && ids->data && grads->data, "%s: null data pointer", name);
    HA_REQUIRE(param->ctx.device_type == kGPU && ids->ctx.device_type == kGPU && grads->ctx.device_type == kGPU,
               "%s: arrays must be on the GPU", name);
    HA_REQUIRE(param->ndim == 2, "%s: param must be 2-D", name);
    *n = dl_numel(ids);
    *width = param->shape[1];
    HA_REQUIRE(*n < (1ll << 31) && *width < (1ll << 30), "%s: sizes out of range", name);
    HA_REQUIRE(dl_numel(grads) == *n * *width, "%s: grad_values size mismatch", name);
    if (need_s1)
        HA_REQUIRE(dl_numel(s1) == dl_numel(param), "%s: state size mismatch", name);
    if (need_s2)
        HA_REQUIRE(dl_numel(s2) == dl_numel(param), "%s: state size mismatch", name);
    return 0;
}

}  // namespace ha

using namespace ha;

// src/ops/OptimizersSparse.cu:3-51 -- grad_values[i,:] += l2reg * param[indices[i],:]
extern "C" int AddL2RegularizationSparse(const DLArrayHandle param, const DLArrayHandle grad_indices,
                                         DLArrayHandle grad_values, float l2reg,
                                         DLStreamHandle stream_handle) {
    int64_t n, width;
    if (check_args("AddL2RegularizationSparse", param, grad_indices, grad_values, nullptr, nullptr, false, false,
                   &n, &width))
        return -1;
    OptArgs a{};
    a.weight_decay = l2reg;
    return row_launch<kL2>(static_cast<float *>(param->data), param->shape[0],
                           static_cast<const float *>(grad_indices->data),
                           static_cast<float *>(grad_values->data), n, width, nullptr, nullptr, a,
                           dl_stream(stream_handle));
}

// src/ops/OptimizersSparse.cu:101-231.  First phase (ids may repeat): velocity[id,:] += -lr * g per
// occurrence (Nesterov: param[id,:] too) -- here in occurrence order through the index plan instead of
// float atomics; acc - lr*g is bit for bit acc + (-lr*g).  Second phase: dense, over the whole table.
extern "C" int MomentumOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                             const DLArrayHandle grad_values, DLArrayHandle velocity,
                                             float lr, float momentum, bool nesterov,
                                             DLStreamHandle stream_handle) {
    int64_t n, width;
    if (check_args("MomentumOptimizerSparseUpdate", param, grad_indices, grad_values, velocity, nullptr, true,
                   false, &n, &width))
        return -1;
    hipStream_t stream = dl_stream(stream_handle);
    float *p = static_cast<float *>(param->data), *v = static_cast<float *>(velocity->data);
    const int64_t rows = param->shape[0];
    if (n > 0) {
        void *ws = nullptr;
        if (scratch_get(stream, ha_plan_bytes(n), &ws))
            return -1;
        if (ha_plan_sort_f32ids(static_cast<const float *>(grad_indices->data), n, ws, stream))
            return -1;
        const float *g = static_cast<const float *>(grad_values->data);
        if (ha_sgd_apply(v, rows, width, ws, n, g, lr, stream))
            return -1;
        if (nesterov && ha_sgd_apply(p, rows, width, ws, n, g, lr, stream))
            return -1;
    }
    const uint64_t total = static_cast<uint64_t>(rows) * static_cast<uint64_t>(width);
    if (total == 0)
        return 0;
    HA_REQUIRE(reinterpret_cast<uintptr_t>(p) % 16 == 0 && reinterpret_cast<uintptr_t>(v) % 16 == 0,
               "MomentumOptimizerSparseUpdate: param and velocity must be 16-byte aligned");
    const uint64_t nvec = total / 4;
    uint64_t blocks = (nvec + 255) / 256;
    if (blocks > 65536)
        blocks = 65536;
    if (blocks == 0)
        blocks = 1;
    if (nesterov)
        hipLaunchKernelGGL(momentum_dense_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, p, v,
                           momentum, nvec, total);
    else
        hipLaunchKernelGGL(momentum_dense_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, p, v,
                           momentum, nvec, total);
    HA_LAUNCH_CHECK();
    return 0;
}

extern "C" int AdaGradOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                            const DLArrayHandle grad_values, DLArrayHandle acc, float lr,
                                            float eps, DLStreamHandle stream_handle) {
    int64_t n, width;
    if (check_args("AdaGradOptimizerSparseUpdate", param, grad_indices, grad_values, acc, nullptr, true, false,
                   &n, &width))
        return -1;
    OptArgs a{};
    a.lr = lr;
    a.eps = eps;
    return row_launch<kAdaGrad>(static_cast<float *>(param->data), param->shape[0],
                                static_cast<const float *>(grad_indices->data),
                                static_cast<float *>(grad_values->data), n, width,
                                static_cast<float *>(acc->data), nullptr, a, dl_stream(stream_handle));
}

static OptArgs adam_args(float lr, float beta1, float beta2, float beta1t, float beta2t, float eps, float wd) {
    OptArgs a{};
    a.lr = lr, a.eps = eps, a.beta1 = beta1, a.beta2 = beta2, a.beta1t = beta1t, a.beta2t = beta2t;
    a.weight_decay = wd;
    return a;
}

extern "C" int AdamOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                         const DLArrayHandle grad_values, DLArrayHandle expavg,
                                         DLArrayHandle expavgsq, float lr, float beta1, float beta2,
                                         float beta1t, float beta2t, float eps, DLStreamHandle stream_handle) {
    int64_t n, width;
    if (check_args("AdamOptimizerSparseUpdate", param, grad_indices, grad_values, expavg, expavgsq, true, true, &n,
                   &width))
        return -1;
    return row_launch<kAdam>(static_cast<float *>(param->data), param->shape[0],
                             static_cast<const float *>(grad_indices->data),
                             static_cast<float *>(grad_values->data), n, width,
                             static_cast<float *>(expavg->data), static_cast<float *>(expavgsq->data),
                             adam_args(lr, beta1, beta2, beta1t, beta2t, eps, 0.f), dl_stream(stream_handle));
}

extern "C" int AdamWOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                          const DLArrayHandle grad_values, DLArrayHandle expavg,
                                          DLArrayHandle expavgsq, float lr, float beta1, float beta2,
                                          float beta1t, float beta2t, float eps, float weight_decay,
                                          DLStreamHandle stream_handle) {
    int64_t n, width;
    if (check_args("AdamWOptimizerSparseUpdate", param, grad_indices, grad_values, expavg, expavgsq, true, true,
                   &n, &width))
        return -1;
    return row_launch<kAdamW>(static_cast<float *>(param->data), param->shape[0],
                              static_cast<const float *>(grad_indices->data),
                              static_cast<float *>(grad_values->data), n, width,
                              static_cast<float *>(expavg->data), static_cast<float *>(expavgsq->data),
                              adam_args(lr, beta1, beta2, beta1t, beta2t, eps, weight_decay),
                              dl_stream(stream_handle));
}

// src/ops/OptimizersSparse.cu:524-722 (deduplicated slices): norm2 of the indexed parameter rows, Adam
// moments + update direction, norm2 of the direction, then
//   param[id,:] -= lr * (norm2_param / norm2_update) * (update + weight_decay * param[id,:]).
// Three launches: direction + per-row partial norms, the two sums, the step.
extern "C" int LambOptimizerSparseUpdate(DLArrayHandle param, const DLArrayHandle grad_indices,
                                         const DLArrayHandle grad_values, DLArrayHandle expavg,
                                         DLArrayHandle expavgsq, float lr, float beta1, float beta2,
                                         float beta1t, float beta2t, float eps, float weight_decay,
                                         DLStreamHandle stream_handle) {
    int64_t n, width;
    if (check_args("LambOptimizerSparseUpdate", param, grad_indices, grad_values, expavg, expavgsq, true, true, &n,
                   &width))
        return -1;
    if (n == 0)
        return 0;
    hipStream_t stream = dl_stream(stream_handle);
    const size_t upd_bytes = align_up(static_cast<size_t>(n) * width * 4, 256);
    const size_t part_bytes = align_up(static_cast<size_t>(n) * 8, 256);
    void *ws = nullptr;
    if (scratch_get(stream, upd_bytes + 2 * part_bytes + 256, &ws))
        return -1;
    char *b = static_cast<char *>(ws);
    OptArgs a = adam_args(lr, beta1, beta2, beta1t, beta2t, eps, weight_decay);
    a.update = reinterpret_cast<float *>(b);
    a.part_param = reinterpret_cast<double *>(b + upd_bytes);
    a.part_update = reinterpret_cast<double *>(b + upd_bytes + part_bytes);
    double *norms = reinterpret_cast<double *>(b + upd_bytes + 2 * part_bytes);
    a.norms = norms;
    float *p = static_cast<float *>(param->data);
    const float *ids = static_cast<const float *>(grad_indices->data);
    float *g = static_cast<float *>(grad_values->data);
    if (row_launch<kLambUpdate>(p, param->shape[0], ids, g, n, width, static_cast<float *>(expavg->data),
                                static_cast<float *>(expavgsq->data), a, stream))
        return -1;
    hipLaunchKernelGGL(lamb_norms_kernel, dim3(1), dim3(1024), 0, stream, a.part_param, a.part_update, (int)n,
                       norms);
    HA_LAUNCH_CHECK();
    return row_launch<kLambStep>(p, param->shape[0], ids, g, n, width, nullptr, nullptr, a, stream);
}
