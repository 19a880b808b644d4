// Element-wise steps of the sparse optimizers, shared by optim.hip (one wave per deduplicated row) and by the
// fused dedup-reduce + optimizer apply (scatter_dev.h, kModeOpt).  Expression order of the reference:
// src/ops/OptimizersSparse.cu:331-349 (AdaGrad), 391-416 (Adam), 457-484 (AdamW), 539-577 (Lamb).
#pragma once
#include "common.h"

#include <math.h>

namespace ha {

enum OptKind { kAdaGrad = 0, kAdam = 1, kAdamW = 2, kL2 = 3, kLambUpdate = 4, kLambStep = 5 };

struct OptArgs {
    float lr, eps, beta1, beta2, beta1t, beta2t, weight_decay;
    float *update;        // Lamb: [n, width] scratch holding the update direction
    double *part_param;   // Lamb: per-row sums of param^2
    double *part_update;  // Lamb: per-row sums of update^2
    const double *norms;  // Lamb: {sum param^2, sum update^2}
};

// One element.  p = parameter, g = gradient (in/out for kL2), s1 / s2 = optimizer state.
template <int KIND>
__device__ __forceinline__ void opt_step(float &p, float &g, float &s1, float &s2, const OptArgs &a,
                                         float &upd, double &sp, double &su, float ratio) {
    if (KIND == kL2) {
        g = g + a.weight_decay * p;                       // grad += l2reg * param  (:17)
    } else if (KIND == kAdaGrad) {
        const float acc = s1 + g * g;
        s1 = acc;
        p = p - a.lr * g / (sqrtf(acc) + a.eps);
    } else if (KIND == kLambStep) {
        p = p - a.lr * ratio * (upd + a.weight_decay * p);   // :577
    } else {
        float m = a.beta1 * s1 + (1.f - a.beta1) * g;
        float v = a.beta2 * s2 + (1.f - a.beta2) * g * g;
        s1 = m;
        s2 = v;
        m = m / (1.f - a.beta1t);
        v = v / (1.f - a.beta2t);
        if (KIND == kAdam) {
            p = p - a.lr * m / (sqrtf(v) + a.eps);
        } else if (KIND == kAdamW) {
            const float update = m / (sqrtf(v) + a.eps);
            p = p - a.lr * (update + a.weight_decay * p);
        } else {  // kLambUpdate: the direction and the two squared norms (:539-561)
            upd = m / (sqrtf(v) + a.eps);
            sp += static_cast<double>(p) * static_cast<double>(p);
            su += static_cast<double>(upd) * static_cast<double>(upd);
        }
    }
}

// runtime kind (wave-uniform) for the fused apply: AdaGrad / Adam / AdamW
__device__ __forceinline__ void opt_step_rt(int kind, float &p, float g, float &s1, float &s2, const OptArgs &a) {
    float upd = 0.f;
    double sp = 0.0, su = 0.0;
    if (kind == kAdaGrad)
        opt_step<kAdaGrad>(p, g, s1, s2, a, upd, sp, su, 0.f);
    else if (kind == kAdam)
        opt_step<kAdam>(p, g, s1, s2, a, upd, sp, su, 0.f);
    else
        opt_step<kAdamW>(p, g, s1, s2, a, upd, sp, su, 0.f);
}

}  // namespace ha
