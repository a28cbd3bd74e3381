// Fused row-wise training step for tables too large for dense gradients / dense Adam (BASELINE config C5):
//
//   cdr_bpr_fwd_grad   one pass over the batch: gather 3 rows per triple, loss partials, and the two compact gradient
//                      rows every other row gradient is made of:  GU[b] = g_b (p - n),  GP[b] = g_b u
//                      (dI[pid_b] += GP[b], dI[nid_b] -= GP[b]).  Nothing table-sized is ever written.
//   cdr_sort_ids       stable LSD radix sort (rocPRIM) of the touched row ids with their occurrence index, over only the
//                      significant key bits -> every table row's occurrences become one contiguous segment.
//   cdr_rowwise_apply  one lane-group per segment head: sums the segment's gradient rows in occurrence order (fixed
//                      order => deterministic, no float atomics), adds the EmbLoss term count * c * W[r], and applies
//                      the optimizer (SGD, or Adam with per-row moments) in place: one read-modify-write per touched row.
//
// Semantics: exactly "sum of the batch's gradients evaluated at the pre-step weights", as autograd + a dense optimizer
// would compute for the touched rows.  Rows not in the batch are not touched (for Adam this is the usual lazy/sparse
// variant: untouched rows keep their moments and do not move) -- documented in DESIGN.md as the one deliberate
// difference from the reference's dense torch.optim.Adam, which is O(table) per step.
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "cdr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

inline int grid_for(int64_t units, int per_block) {
    int64_t g = (units + per_block - 1) / per_block;
    const int64_t cap = CDR_NUM_CU * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------ forward + compact grads
// SCATTER (row-sharded step): the item operand is the buffer of received rows, one entry per occurrence, and the item
// gradients are written straight into the send buffer at the same positions: GP[pid[t]] = g u, GP[nid[t]] = -g u.
template <int LPR, bool SCATTER>
__global__ __launch_bounds__(kBlock) void bpr_fwd_grad_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                              int D, const int64_t* __restrict__ uid,
                                                              const int64_t* __restrict__ pid,
                                                              const int64_t* __restrict__ nid, int64_t B, float gamma,
                                                              float invB, float* __restrict__ GU,
                                                              float* __restrict__ GP, double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool live = sub < D4;

    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        float4 u[kUnroll], p[kUnroll], n[kUnroll];
        // ids of all kUnroll triples first, THEN all row loads: vmcnt is an in-order counter, so interleaving
        // "ids(r) -> rows(r)" makes the wait for ids(r+1) also wait for rows(r) and serialises the gathers
        int64_t iu[kUnroll], ip[kUnroll], in[kUnroll];
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ip[r] = pid[tc]; in[r] = nid[tc];
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            u[r] = p[r] = n[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < B && live) {
                u[r] = ld4(U + iu[r] * D + 4 * sub);
                p[r] = ld4(I + ip[r] * D + 4 * sub);
                n[r] = ld4(I + in[r] * D + 4 * sub);
            }
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const float dp = group_sum<LPR>(dot4(u[r], p[r]));
            const float dn = group_sum<LPR>(dot4(u[r], n[r]));
            const float su = group_sum<LPR>(dot4(u[r], u[r]));
            const float sp = group_sum<LPR>(dot4(p[r], p[r]));
            if (t < B) {
                const float s = sigmoidf_(dp - dn);
                const float g = -invB * (s * (1.0f - s)) / (gamma + s);
                if (live) {
                    st4(GU + t * D + 4 * sub, make_float4(g * (p[r].x - n[r].x), g * (p[r].y - n[r].y),
                                                          g * (p[r].z - n[r].z), g * (p[r].w - n[r].w)));
                    if (SCATTER) {
                        st4(GP + ip[r] * D + 4 * sub, make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w));
                        st4(GP + in[r] * D + 4 * sub, make_float4(-g * u[r].x, -g * u[r].y, -g * u[r].z, -g * u[r].w));
                    } else {
                        st4(GP + t * D + 4 * sub, make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w));
                    }
                }
                if (sub == 0) {
                    acc[0] += (double)(-logf(gamma + s));
                    acc[1] += (double)su;
                    acc[2] += (double)sp;
                }
            }
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// out9 = {total, main, ||U_b||, ||I_b||, c_u, c_i, sum loss, sum u^2, sum p^2} with c = reg_weight / (B * norm)
// (0 when the norm is 0).  B is the batch size the mean and the EmbLoss are taken over (the GLOBAL batch when the
// step is sharded: then out9[0..5] are provisional and cdr_loss_finish_sums recomputes them from all-reduced sums).
__global__ __launch_bounds__(kBlock) void step_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t B,
                                                             float reg_weight, float* __restrict__ out6) {
    __shared__ double smem[3 * (kBlock / 64)];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2];
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        const float main_loss = (float)(acc[0] / (double)B);
        const float nu = (float)sqrt(acc[1]), ni = (float)sqrt(acc[2]);
        out6[1] = main_loss; out6[2] = nu; out6[3] = ni;
        out6[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
        out6[4] = (reg_weight != 0.f && nu > 0.f) ? reg_weight / ((float)B * nu) : 0.f;
        out6[5] = (reg_weight != 0.f && ni > 0.f) ? reg_weight / ((float)B * ni) : 0.f;
        out6[6] = (float)acc[0]; out6[7] = (float)acc[1]; out6[8] = (float)acc[2];
    }
}

__global__ void finish_sums_kernel(const float* __restrict__ sums3, int64_t B, float reg_weight, float* __restrict__ out6) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float main_loss = sums3[0] / (float)B;
        const float nu = sqrtf(sums3[1]), ni = sqrtf(sums3[2]);
        out6[1] = main_loss; out6[2] = nu; out6[3] = ni;
        out6[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
        out6[4] = (reg_weight != 0.f && nu > 0.f) ? reg_weight / ((float)B * nu) : 0.f;
        out6[5] = (reg_weight != 0.f && ni > 0.f) ? reg_weight / ((float)B * ni) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ keys for the sort
__global__ __launch_bounds__(kBlock) void make_keys_kernel(const int64_t* __restrict__ ids0, int64_t n0,
                                                           const int64_t* __restrict__ ids1, int64_t n1,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t n = n0 + n1, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        keys[e] = (uint32_t)(e < n0 ? ids0[e] : ids1[e - n0]);
        vals[e] = (uint32_t)e;
    }
}

// ------------------------------------------------------------------------------------------------ segmented apply
// OPT 0: SGD  w -= lr * grad            OPT 1: Adam on the touched rows (torch.optim.Adam arithmetic per element)
template <int LPR, int OPT, bool SIGNED>
__global__ __launch_bounds__(kBlock) void rowwise_apply_kernel(float* __restrict__ W, float* __restrict__ Mo,
                                                               float* __restrict__ Vo, int D,
                                                               const uint32_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ perm, int64_t n,
                                                               const float* __restrict__ G, int64_t neg_start,
                                                               int64_t reg_limit, const float* __restrict__ reg_coef,
                                                               float lr, float b1, float b2, float eps, float wd,
                                                               float step_size, float bc2_sqrt,
                                                               const int64_t* __restrict__ occ_ids) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    for (int64_t q = gg; q < n; q += TG) {
        const uint32_t row = keys[q];
        if (q > 0 && keys[q - 1] == row) continue;          // not a segment head (uniform inside the group)
        for (int ch = sub; ch < D4; ch += LPR) {
            float* wp = W + (int64_t)row * D + 4 * ch;
            const float4 w = ld4(wp);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnt = 0;
            for (int64_t e = q; e < n && keys[e] == row; ++e) {
                const int64_t o = perm[e];
                const bool neg = SIGNED && o >= neg_start;
                const float4 g = ld4(G + (neg ? o - neg_start : o) * D + 4 * ch);
                if (neg) { acc.x -= g.x; acc.y -= g.y; acc.z -= g.z; acc.w -= g.w; }
                else { acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w; }
                cnt += occ_ids ? (int)((occ_ids[o] >> 62) & 1) : ((o < reg_limit) ? 1 : 0);
            }
            const float rc = c * (float)cnt;
            float4 gr = make_float4(acc.x + rc * w.x, acc.y + rc * w.y, acc.z + rc * w.z, acc.w + rc * w.w);
            float4 wn;
            if (OPT == 0) {
                if (wd != 0.f) { gr.x += wd * w.x; gr.y += wd * w.y; gr.z += wd * w.z; gr.w += wd * w.w; }
                wn = make_float4(w.x - lr * gr.x, w.y - lr * gr.y, w.z - lr * gr.z, w.w - lr * gr.w);
            } else {
                float* mp = Mo + (int64_t)row * D + 4 * ch;
                float* vp = Vo + (int64_t)row * D + 4 * ch;
                float4 m = ld4(mp), v = ld4(vp);
                if (wd != 0.f) { gr.x += wd * w.x; gr.y += wd * w.y; gr.z += wd * w.z; gr.w += wd * w.w; }
                m.x += (gr.x - m.x) * (1.0f - b1); m.y += (gr.y - m.y) * (1.0f - b1);
                m.z += (gr.z - m.z) * (1.0f - b1); m.w += (gr.w - m.w) * (1.0f - b1);
                v.x = b2 * v.x + (1.0f - b2) * gr.x * gr.x; v.y = b2 * v.y + (1.0f - b2) * gr.y * gr.y;
                v.z = b2 * v.z + (1.0f - b2) * gr.z * gr.z; v.w = b2 * v.w + (1.0f - b2) * gr.w * gr.w;
                st4(mp, m); st4(vp, v);
                wn = make_float4(w.x - step_size * (m.x / (sqrtf(v.x) / bc2_sqrt + eps)),
                                 w.y - step_size * (m.y / (sqrtf(v.y) / bc2_sqrt + eps)),
                                 w.z - step_size * (m.z / (sqrtf(v.z) / bc2_sqrt + eps)),
                                 w.w - step_size * (m.w / (sqrtf(v.w) / bc2_sqrt + eps)));
            }
            st4(wp, wn);
        }
    }
}

}  // namespace

#define DISPATCH_LPR(lpr, ...)                                  \
    switch (lpr) {                                              \
        case 1: { constexpr int L = 1; __VA_ARGS__; } break;    \
        case 2: { constexpr int L = 2; __VA_ARGS__; } break;    \
        case 4: { constexpr int L = 4; __VA_ARGS__; } break;    \
        case 8: { constexpr int L = 8; __VA_ARGS__; } break;    \
        case 16: { constexpr int L = 16; __VA_ARGS__; } break;  \
        case 32: { constexpr int L = 32; __VA_ARGS__; } break;  \
        default: { constexpr int L = 64; __VA_ARGS__; } break;  \
    }

extern "C" int cdr_bpr_fwd_grad(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D,
                                const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, int64_t B_mean,
                                float gamma, float reg_weight, float* out6, float* GU, float* GP, int scatter) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out6 && GU && GP);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    if (B_mean <= 0) B_mean = B;
    const float invB = 1.0f / (float)B_mean;
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_for((B + kUnroll - 1) / kUnroll, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_GRAD, s);
        if (scatter) {
            DISPATCH_LPR(lpr, bpr_fwd_grad_kernel<L, true><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid,
                                                                                                 B, gamma, invB, GU, GP, ctx->partials));
        } else {
            DISPATCH_LPR(lpr, bpr_fwd_grad_kernel<L, false><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid,
                                                                                                  B, gamma, invB, GU, GP, ctx->partials));
        }
    }
    CDR_LAUNCH_CHECK();
    step_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B_mean, reg_weight, out6);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_loss_finish_sums(void* stream, const float* sums3, int64_t B_mean, float reg_weight, float* out6) {
    CDR_CHECK_ARG(sums3 && out6 && B_mean > 0);
    finish_sums_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(sums3, B_mean, reg_weight, out6);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

static inline unsigned bits_for(int64_t num_rows) {
    unsigned b = 1;
    while (b < 32 && ((int64_t)1 << b) < num_rows) ++b;
    return b;
}

extern "C" int cdr_sort_workspace_bytes(int64_t n, int64_t num_rows, size_t* bytes) {
    CDR_CHECK_ARG(bytes && n > 0 && num_rows > 0 && num_rows <= (int64_t)0xFFFFFFFFu && n <= (int64_t)0x7FFFFFFF);
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, bits_for(num_rows));
    if (e != hipSuccess) { cdr_set_error("cdr_sort_workspace_bytes: %s", hipGetErrorString(e)); return (int)e; }
    tmp = (tmp + 255) & ~(size_t)255;
    *bytes = tmp + 2 * (((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255);
    return CDR_OK;
}

extern "C" int cdr_sort_ids(cdr_ctx* ctx, void* stream, const int64_t* ids0, int64_t n0, const int64_t* ids1, int64_t n1,
                            int64_t num_rows, uint32_t* keys_sorted, uint32_t* perm, void* workspace,
                            size_t workspace_bytes) {
    const int64_t n = n0 + n1;
    CDR_CHECK_ARG(ids0 && n0 > 0 && (n1 == 0 || ids1) && keys_sorted && perm && workspace);
    size_t need = 0;
    int rc = cdr_sort_workspace_bytes(n, num_rows, &need);
    if (rc) return rc;
    CDR_CHECK_ARG(workspace_bytes >= need);
    hipStream_t s = (hipStream_t)stream;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t* keys_in = (uint32_t*)workspace;
    uint32_t* vals_in = (uint32_t*)((char*)workspace + arr);
    void* tmp = (char*)workspace + 2 * arr;
    size_t tmp_bytes = workspace_bytes - 2 * arr;
    cdr_time_scope ts(ctx, CDR_TAG_SORT, s);
    make_keys_kernel<<<dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s>>>(ids0, n0, ids1, n1, keys_in, vals_in);
    CDR_LAUNCH_CHECK();
    CDR_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, (const uint32_t*)keys_in, keys_sorted, (const uint32_t*)vals_in, perm,
                                      (size_t)n, 0u, bits_for(num_rows), s));
    return CDR_OK;
}

extern "C" int cdr_rowwise_apply(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D,
                                 const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, const float* G,
                                 int64_t neg_start, int64_t reg_limit, const float* reg_coef, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int64_t step, const int64_t* occ_ids) {
    CDR_CHECK_ARG(table && keys_sorted && perm && G && n > 0);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && exp_avg && exp_avg_sq && step > 0));
    hipStream_t s = (hipStream_t)stream;
    float step_size = lr, bc2_sqrt = 1.f;
    if (opt == 1) {
        const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_for(n, kBlock / lpr);
    const bool is_signed = neg_start < n;
    cdr_time_scope ts(ctx, is_signed ? CDR_TAG_APPLY_SIGNED : CDR_TAG_APPLY_UNSIGNED, s);
#define APPLY_ARGS table, exp_avg, exp_avg_sq, D, keys_sorted, perm, n, G, neg_start, reg_limit, reg_coef, lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, occ_ids
    if (opt == 0 && !is_signed) { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 0, false><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
    else if (opt == 0) { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 0, true><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
    else if (!is_signed) { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 1, false><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
    else { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 1, true><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
#undef APPLY_ARGS
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
