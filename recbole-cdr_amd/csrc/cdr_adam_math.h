// The per-element arithmetic of torch.optim.Adam (single-tensor path, amsgrad = False) shared by the dense sweep
// (cdr_rows.hip: adam_multi_dev_kernel) and the deferred per-row form (cdr_lazyadam.hip).  Floating-point contraction is OFF
// inside: every operation rounds on its own, so the two kernels produce bit-identical results from identical inputs whatever
// the compiler does around the call (an fma formed in one of them and not in the other would break that).
#pragma once
#include <hip/hip_runtime.h>

//   g += wd*p ; m = m + (g - m)(1 - b1) ; v = b2 v + (1 - b2) g g ; p -= step_size * m / (sqrt(v)/bc2_sqrt + eps)
__device__ __forceinline__ float cdr_adam_elem(float pv, float gv, float& m, float& v, float b1, float b2, float eps, float wd,
                                               float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
    if (wd != 0.f) gv = gv + wd * pv;
    const float mv = m + (gv - m) * (1.0f - b1);          // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vv = b2 * v + ((1.0f - b2) * gv) * gv;    // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    m = mv; v = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    return pv - step_size * (mv / denom);
}

// bias corrections of update number `st` (>= 1), as the capturable dense kernel computes them
__device__ __forceinline__ void cdr_adam_hp(double st, float lr, float b1, float b2, float& step_size, float& bc2_sqrt) {
    step_size = (float)((double)lr / (1.0 - pow((double)b1, st)));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, st));
}
