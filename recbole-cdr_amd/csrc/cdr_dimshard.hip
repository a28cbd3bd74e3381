// BPR step over DIMENSION-sharded tables (DESIGN.md 6 "dimension sharding"): rank r holds columns
// [r Ds, (r+1) Ds) of EVERY row of both tables, Ds = D / world, and every rank walks the GLOBAL batch.
//
//   row sharding moves rows:   2 rows in + 2 gradient rows out per triple  ~ 2.1 KB over xGMI
//   dimension sharding moves:  the triple's ids (24 B, all-gather) + ONE partial score (4 B, all-reduce)
//
// because the loss needs only x_t = <u, p> - <u, n> summed over the column slices, and every gradient element
// dL/dU[u,c] = g_t (p_c - n_c), dL/dI[p,c] = g_t u_c, dL/dI[n,c] = -g_t u_c lives on the rank that holds column c.  The
// step is therefore the single-GPU fused step (gather -> compact gradient rows -> sort -> row-wise apply, cdr_step.hip)
// cut in two around one all-reduce:
//
//   cdr_bpr_partial_diff     diff[t] = <u,p> - <u,n> over this rank's columns; diff[B], diff[B+1] = its share of
//                            sum |u|^2, sum |p|^2 (the EmbLoss norms, emb_loss of recbole: App. A) -> all-reduce(sum)
//   cdr_bpr_grad_from_diff   s = sigmoid(x), g = -(1/B) s (1-s) / (gamma + s)   (identical on every rank), the compact
//                            gradient rows GU[t] = g (p - n), GP[t] = g u on this rank's columns, the loss, and the
//                            EmbLoss coefficients from the all-reduced norms; cdr_sort_ids_two_tables +
//                            cdr_rowwise_apply then run unchanged on [rows, Ds] tables.
//
// All sizes are static (no bucket counts, no host sync); the rows are re-gathered after the all-reduce (2 x 3 Ds floats
// per triple instead of 3), which is the price for not keeping 3 B Ds floats alive across the collective.
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

template <int LPR>
__global__ __launch_bounds__(kBlock) void bpr_partial_diff_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                  const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                                  const int64_t* __restrict__ nid, int64_t B,
                                                                  float* __restrict__ diff, double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[2 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    double acc[2] = {0.0, 0.0};
    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        int64_t iu[kUnroll], ip[kUnroll], in[kUnroll];
        float4 u[kUnroll], p[kUnroll], n[kUnroll];
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {                   // all ids first, then all rows (see cdr_step.hip)
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
            if (t < B && sub == 0) {
                diff[t] = dp - dn;
                acc[0] += (double)su;
                acc[1] += (double)sp;
            }
        }
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1];
    }
}

__global__ __launch_bounds__(kBlock) void partial_norms_kernel(const double* __restrict__ partials, int nblocks,
                                                               float* __restrict__ norms2) {
    __shared__ double smem[2 * (kBlock / 64)];
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1];
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) { norms2[0] = (float)acc[0]; norms2[1] = (float)acc[1]; }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void bpr_grad_from_diff_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                    const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                                    const int64_t* __restrict__ nid, int64_t B, float gamma,
                                                                    float invB, const float* __restrict__ diff,
                                                                    float* __restrict__ GU, float* __restrict__ GP,
                                                                    double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[kBlock / 64];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    double acc[1] = {0.0};
    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        int64_t iu[kUnroll], ip[kUnroll], in[kUnroll];
        float x[kUnroll];
        float4 u[kUnroll], p[kUnroll], n[kUnroll];
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ip[r] = pid[tc]; in[r] = nid[tc]; x[r] = diff[tc];
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
            if (t < B) {
                const float s = sigmoidf_(x[r]);
                const float g = -invB * (s * (1.0f - s)) / (gamma + s);
                if (live) {
                    st4(GU + t * D + 4 * sub, make_float4(g * (p[r].x - n[r].x), g * (p[r].y - n[r].y),
                                                          g * (p[r].z - n[r].z), g * (p[r].w - n[r].w)));
                    st4(GP + t * D + 4 * sub, make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w));
                }
                if (sub == 0) acc[0] += (double)(-logf(gamma + s));
            }
        }
    }
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
}

// out9 as cdr_bpr_fwd_grad's: {total, main, ||U_b||, ||I_b||, c_u, c_i, sum loss, sum u^2, sum p^2}; the two norms come
// from the all-reduced diff tail, the loss sum is the same on every rank.
__global__ __launch_bounds__(kBlock) void dimshard_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t B,
                                                                 float reg_weight, const float* __restrict__ norms2,
                                                                 float* __restrict__ out9) {
    __shared__ double smem[kBlock / 64];
    double acc[1] = {0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) acc[0] += partials[(size_t)b * CDR_PARTIAL_STRIDE];
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) {
        const float main_loss = (float)(acc[0] / (double)B);
        const float nu = sqrtf(norms2[0]), ni = sqrtf(norms2[1]);
        out9[1] = main_loss; out9[2] = nu; out9[3] = ni;
        out9[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
        out9[4] = (reg_weight != 0.f && nu > 0.f) ? reg_weight / ((float)B * nu) : 0.f;
        out9[5] = (reg_weight != 0.f && ni > 0.f) ? reg_weight / ((float)B * ni) : 0.f;
        out9[6] = (float)acc[0]; out9[7] = norms2[0]; out9[8] = norms2[1];
    }
}

// Pointwise counterpart (EMCDR's default MF latent factor model, emcdr.py:111-122: MSE on the raw dot; BCE on sigmoid(dot) as
// cmf.py:75-99): dot[t] = <u, i> over this rank's columns (+ the two EmbLoss norms behind it) -> all-reduce -> the loss
// derivative per row and the compact gradient rows GU[t] = g i, GI[t] = g u.  Same arithmetic as point_fwd_grad_kernel.
template <int LPR>
__global__ __launch_bounds__(kBlock) void point_partial_dot_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                   const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                                   int64_t B, float* __restrict__ dot, double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[2 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    double acc[2] = {0.0, 0.0};
    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        int64_t iu[kUnroll], ii[kUnroll];
        float4 u[kUnroll], v[kUnroll];
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ii[r] = iid[tc];
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            u[r] = v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < B && live) {
                u[r] = ld4(U + iu[r] * D + 4 * sub);
                v[r] = ld4(I + ii[r] * D + 4 * sub);
            }
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const float dx = group_sum<LPR>(dot4(u[r], v[r]));
            const float su = group_sum<LPR>(dot4(u[r], u[r]));
            const float si = group_sum<LPR>(dot4(v[r], v[r]));
            if (t < B && sub == 0) {
                dot[t] = dx;
                acc[0] += (double)su;
                acc[1] += (double)si;
            }
        }
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1];
    }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void point_grad_from_dot_kernel(int loss_kind, const float* __restrict__ U, const float* __restrict__ I,
                                                                     int D, const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                                     const float* __restrict__ label, int64_t B, float invB,
                                                                     const float* __restrict__ dot, float* __restrict__ GU,
                                                                     float* __restrict__ GI, double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[kBlock / 64];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    double acc[1] = {0.0};
    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        int64_t iu[kUnroll], ii[kUnroll];
        float x[kUnroll], yl[kUnroll];
        float4 u[kUnroll], v[kUnroll];
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ii[r] = iid[tc]; x[r] = dot[tc]; yl[r] = label[tc];
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            u[r] = v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < B && live) {
                u[r] = ld4(U + iu[r] * D + 4 * sub);
                v[r] = ld4(I + ii[r] * D + 4 * sub);
            }
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            if (t < B) {
                const float y = yl[r];
                float l, g;
                if (loss_kind == CDR_LOSS_MSE) {
                    const float d = x[r] - y;
                    l = d * d; g = 2.0f * d * invB;
                } else {                                       // torch BCELoss on sigmoid(dot): -100 log clamp, 1e-12 backward clamp
                    const float p = sigmoidf_(x[r]);
                    l = (y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f);
                    const float pq = (1.0f - p) * p;
                    g = (p - y) / fmaxf(pq, 1e-12f) * invB * pq;
                }
                if (live) {
                    st4(GU + t * D + 4 * sub, make_float4(g * v[r].x, g * v[r].y, g * v[r].z, g * v[r].w));
                    st4(GI + t * D + 4 * sub, make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w));
                }
                if (sub == 0) acc[0] += (double)l;
            }
        }
    }
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
}

// ids travel as int32 (table rows < 2^31 everywhere in this library: the sort keys are 32-bit): 12 B per triple on the links.
//   pack   : out32[j * Bl + t] = (int32) src_j[t]                      j = 0 (user), 1 (positive), 2 (negative)
//   unpack : the all-gathered buffer is rank-major [G][3][Bl]; the kernels want field-major int64 [3][G * Bl]
// Pointwise rows carry (user, item, label): the label's fp32 bit pattern takes the third slot (n == nullptr, label != nullptr).
__global__ __launch_bounds__(kBlock) void ids_pack32_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ p,
                                                            const int64_t* __restrict__ n, const float* __restrict__ label,
                                                            int64_t Bl, int32_t* __restrict__ out, int* __restrict__ bad) {
    const int64_t total = 3 * Bl, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t j = e / Bl, t = e - j * Bl;
        if (j == 2 && n == nullptr) { out[e] = __float_as_int(label[t]); continue; }
        const int64_t v = j == 0 ? u[t] : j == 1 ? p[t] : n[t];
        if (v < 0 || v > 0x7FFFFFFFll) atomicExch(bad, 1);
        out[e] = (int32_t)v;
    }
}

__global__ __launch_bounds__(kBlock) void ids_unpack32_kernel(const int32_t* __restrict__ in, int G, int64_t Bl,
                                                              int64_t* __restrict__ out, float* __restrict__ label_out) {
    const int64_t Bg = (int64_t)G * Bl, total = 3 * Bg, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t j = e / Bg, r = e - j * Bg;            // out[j][g * Bl + t]
        const int64_t g = r / Bl, t = r - g * Bl;
        const int32_t v = in[(g * 3 + j) * Bl + t];
        if (j == 2 && label_out) label_out[r] = __int_as_float(v);
        else out[e] = (int64_t)v;
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

extern "C" int cdr_bpr_partial_diff(cdr_ctx* ctx, void* stream, const float* user_cols, const float* item_cols, int Ds,
                                    const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, float* diff) {
    CDR_CHECK_ARG(ctx && user_cols && item_cols && uid && pid && nid && diff);
    CDR_CHECK_ARG(Ds > 0 && (Ds & 3) == 0 && Ds <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(Ds);
    const int grid = grid_for((B + kUnroll - 1) / kUnroll, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_PARTIAL_DIFF, s);
        DISPATCH_LPR(lpr, bpr_partial_diff_kernel<L><<<dim3(grid), dim3(kBlock), 0, s>>>(user_cols, item_cols, Ds, uid, pid, nid, B, diff,
                                                                                          ctx->partials));
    }
    CDR_LAUNCH_CHECK();
    partial_norms_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, diff + B);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bpr_grad_from_diff(cdr_ctx* ctx, void* stream, const float* user_cols, const float* item_cols, int Ds,
                                      const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, float gamma,
                                      float reg_weight, const float* diff, float* out9, float* GU, float* GP) {
    CDR_CHECK_ARG(ctx && user_cols && item_cols && uid && pid && nid && diff && out9 && GU && GP);
    CDR_CHECK_ARG(Ds > 0 && (Ds & 3) == 0 && Ds <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(Ds);
    const int grid = grid_for((B + kUnroll - 1) / kUnroll, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_GRAD_FROM_DIFF, s);
        DISPATCH_LPR(lpr, bpr_grad_from_diff_kernel<L><<<dim3(grid), dim3(kBlock), 0, s>>>(user_cols, item_cols, Ds, uid, pid, nid, B, gamma,
                                                                                            1.0f / (float)B, diff, GU, GP, ctx->partials));
    }
    CDR_LAUNCH_CHECK();
    dimshard_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, diff + B, out9);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_ids_pack32(void* stream, const int64_t* uid, const int64_t* pid, const int64_t* nid, const float* label, int64_t Bl,
                              int32_t* out32, int* bad_flag) {
    CDR_CHECK_ARG(uid && pid && out32 && bad_flag && Bl > 0);
    CDR_CHECK_ARG((nid != nullptr) != (label != nullptr));
    ids_pack32_kernel<<<dim3(grid_for(3 * Bl, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(uid, pid, nid, label, Bl, out32, bad_flag);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_ids_unpack32(void* stream, const int32_t* gathered, int world, int64_t Bl, int64_t* out64, float* label_out) {
    CDR_CHECK_ARG(gathered && out64 && world >= 1 && Bl > 0);
    ids_unpack32_kernel<<<dim3(grid_for(3 * Bl * world, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(gathered, world, Bl, out64, label_out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_point_partial_dot(cdr_ctx* ctx, void* stream, const float* user_cols, const float* item_cols, int Ds,
                                     const int64_t* uid, const int64_t* iid, int64_t B, float* dot) {
    CDR_CHECK_ARG(ctx && user_cols && item_cols && uid && iid && dot);
    CDR_CHECK_ARG(Ds > 0 && (Ds & 3) == 0 && Ds <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(Ds);
    const int grid = grid_for((B + kUnroll - 1) / kUnroll, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_POINT_PARTIAL_DOT, s);
        DISPATCH_LPR(lpr, point_partial_dot_kernel<L><<<dim3(grid), dim3(kBlock), 0, s>>>(user_cols, item_cols, Ds, uid, iid, B, dot, ctx->partials));
    }
    CDR_LAUNCH_CHECK();
    partial_norms_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, dot + B);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_point_grad_from_dot(cdr_ctx* ctx, void* stream, int loss_kind, const float* user_cols, const float* item_cols, int Ds,
                                       const int64_t* uid, const int64_t* iid, const float* label, int64_t B, float reg_weight,
                                       const float* dot, float* out9, float* GU, float* GI) {
    CDR_CHECK_ARG(ctx && user_cols && item_cols && uid && iid && label && dot && out9 && GU && GI);
    CDR_CHECK_ARG((loss_kind == CDR_LOSS_MSE || loss_kind == CDR_LOSS_BCE) && Ds > 0 && (Ds & 3) == 0 && Ds <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(Ds);
    const int grid = grid_for((B + kUnroll - 1) / kUnroll, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_POINT_GRAD_FROM_DOT, s);
        DISPATCH_LPR(lpr, point_grad_from_dot_kernel<L><<<dim3(grid), dim3(kBlock), 0, s>>>(loss_kind, user_cols, item_cols, Ds, uid, iid, label,
                                                                                             B, 1.0f / (float)B, dot, GU, GI, ctx->partials));
    }
    CDR_LAUNCH_CHECK();
    dimshard_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, dot + B, out9);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
