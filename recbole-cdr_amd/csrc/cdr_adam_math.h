// The per-element arithmetic of torch.optim.Adam (single-tensor path, amsgrad = False) shared by the dense sweep
// (cdr_rows.hip: adam_multi_dev_kernel) and the deferred per-row form (cdr_lazyadam.hip).  Floating-point contraction is OFF
// inside: every operation rounds on its own, so the two kernels produce bit-identical results from identical inputs whatever
// the compiler does around the call (an fma formed in one of them and not in the other would break that).
//
// Round 4: the square root and the two divisions of the update term use the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 and a
// host-exact reciprocal of the bias correction instead of the IEEE-correct sequences (~11 instructions per division, ~10 per
// square root: 32 of the ~40 instructions of one element-update).  Why it matters: the deferred form replays every postponed
// update of a row before the row is read -- with real (non-repeating) batches a C3 user row is ~95 updates behind when it comes up
// again, and that replay (lz_prepare_kernel) is VALU-issue bound: 157 us per step with the IEEE sequences.  What it changes: the
// update term lr * m_hat / (sqrt(v_hat) + eps) carries <= ~3 ulp of ITS OWN magnitude (it is ~lr, the weight ~10-100 x larger), i.e.
// below one ulp of the weight; dense and deferred forms still share this one function, so they stay bit-identical to each other.
// -DCDR_ADAM_IEEE restores the IEEE sequences (A/B runs: profiles/r04_ab_adam_math.txt).
#pragma once
#include <hip/hip_runtime.h>

// THE update term of every Adam kernel of this library (dense sweep, deferred per-row form, the row-wise fused steps of cdr_step.hip /
// cdr_kstep.hip, the OVERLAP step of cdr_mapstep.hip):   step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// `bc2` is the value cdr_adam_hp returns for update t.  Default: v_sqrt_f32 / v_rcp_f32 (1 ulp each) and bc2 = 1 / sqrt(1 - b2^t) rounded
// once from fp64; -DCDR_ADAM_IEEE: sqrtf and two IEEE divisions with bc2 = sqrt(1 - b2^t) (torch's own sequence).  One function, one
// rounding: two optimizers of this product never round the same expression differently.
// Drift against torch.optim.Adam over 2,000 free-running updates: tests/test_gpu_parity.py::test_adam_long_horizon_vs_torch.
__device__ __forceinline__ float cdr_adam_term(float mv, float vv, float step_size, float bc2, float eps) {
#pragma clang fp contract(off)
#ifdef CDR_ADAM_IEEE
    const float denom = sqrtf(vv) / bc2 + eps;
    return step_size * (mv / denom);
#else
    const float denom = __builtin_amdgcn_sqrtf(vv) * bc2 + eps;      // v_sqrt_f32 (1 ulp; a denormal v counts as 0: its update is < 1e-19 lr)
    return step_size * (mv * __builtin_amdgcn_rcpf(denom));          // v_rcp_f32 (1 ulp); denom >= eps
#endif
}

//   g += wd*p ; m = m + (g - m)(1 - b1) ; v = b2 v + (1 - b2) g g ; p -= cdr_adam_term(m, v)
__device__ __forceinline__ float cdr_adam_elem(float pv, float gv, float& m, float& v, float b1, float b2, float eps, float wd,
                                               float step_size, float bc2) {
#pragma clang fp contract(off)
    if (wd != 0.f) gv = gv + wd * pv;
    const float mv = m + (gv - m) * (1.0f - b1);          // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vv = b2 * v + ((1.0f - b2) * gv) * gv;    // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    m = mv; v = vv;
    return pv - cdr_adam_term(mv, vv, step_size, bc2, eps);
}

// bias corrections of update number `st` (>= 1): on the host (launch arguments) and on the device (capturable kernels) alike
__host__ __device__ __forceinline__ void cdr_adam_hp(double st, float lr, float b1, float b2, float& step_size, float& bc2) {
    step_size = (float)((double)lr / (1.0 - pow((double)b1, st)));
#ifdef CDR_ADAM_IEEE
    bc2 = (float)sqrt(1.0 - pow((double)b2, st));
#else
    bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, st)));
#endif
}
