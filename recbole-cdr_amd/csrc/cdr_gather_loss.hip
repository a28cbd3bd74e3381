// Fused dual-table gather -> dot -> loss kernels (K1+K2+K3/K4+K5 of SURVEY.md 2.2) and their dense backward.
//
// Mapping: a row of D fp32 is moved by LPR = D/4 lanes as one float4 (16 B) each, so one wave-instruction fetches
// 64/LPR whole rows (D=128: 2 rows, 1 KiB; D=64: 4 rows).  A "group" of LPR lanes owns one interaction at a time,
// walks the batch with a grid stride and keeps UNR interactions' row loads in flight before it touches any of them
// (HBM-latency hiding by memory-level parallelism: the gather is random 256..512-B reads over tables >> L2/MALL).
// Dots are reduced inside the group with xor-shuffles; per-interaction losses and the two EmbLoss square sums are
// accumulated in fp64 per lane, reduced per block, and finished by one block in a fixed order (no float atomics).
#include "cdr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;
constexpr int kUnrollPoint = 8;      // pointwise rows carry 2 row loads each (BPR triples 3): keep as many bytes in flight

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ BPR forward
// out4 = {total, main, ||U_b||_F, ||I_b||_F}
template <bool SYS = false>
__device__ __forceinline__ void loss_finish_body(const double* __restrict__ partials, int nblocks, int64_t B,
                                                 float reg_weight, float* __restrict__ out4) {
    __shared__ double smem[3 * (kBlock / 64)];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        if (SYS) { acc[0] += cdr_load_sys(o); acc[1] += cdr_load_sys(o + 1); acc[2] += cdr_load_sys(o + 2); }      // (past the L2s: cdr_sign_in_last)
        else { acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        const float main_loss = (float)(acc[0] / (double)B);
        const float nu = (float)sqrt(acc[1]), ni = (float)sqrt(acc[2]);
        out4[1] = main_loss; out4[2] = nu; out4[3] = ni;
        out4[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
    }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void bpr_fwd_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                         int D, const int64_t* __restrict__ uid,
                                                         const int64_t* __restrict__ pid,
                                                         const int64_t* __restrict__ nid, int64_t B, float gamma,
                                                         float* __restrict__ gcoef, double* __restrict__ partials,
                                                         unsigned* __restrict__ ticket, float reg_weight, float* __restrict__ out4,
                                                         uint4* __restrict__ scrub, int64_t scrub_n16) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    cdr_scrub(scrub, scrub_n16);
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float invB = 1.0f / (float)B;
    double acc[3] = {0.0, 0.0, 0.0};   // loss sum, sum u^2, sum p^2

    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        if (D4 <= LPR) {
            // one float4 per lane per row: issue all loads of kUnroll interactions, then reduce
            float4 u[kUnroll], p[kUnroll], n[kUnroll];
            const bool live = sub < D4;
            int64_t iu[kUnroll], ip[kUnroll], in[kUnroll];      // ids first, then every row load (see cdr_step.hip)
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
                float dp = group_sum<LPR>(dot4(u[r], p[r]));
                float dn = group_sum<LPR>(dot4(u[r], n[r]));
                float su = group_sum<LPR>(dot4(u[r], u[r]));
                float sp = group_sum<LPR>(dot4(p[r], p[r]));
                if (t < B && sub == 0) {
                    const float s = sigmoidf_(dp - dn);
                    acc[0] += (double)(-logf(gamma + s));
                    acc[1] += (double)su;
                    acc[2] += (double)sp;
                    if (gcoef) gcoef[t] = -invB * (s * (1.0f - s)) / (gamma + s);
                }
            }
        } else {
            // long rows (D > 256): chunked, one interaction at a time per group
#pragma unroll 1
            for (int r = 0; r < kUnroll; ++r) {
                const int64_t t = base + (int64_t)r * TG;
                if (t >= B) break;       // uniform across the group
                const int64_t iu = uid[t], ip = pid[t], in = nid[t];
                float dp = 0.f, dn = 0.f, su = 0.f, sp = 0.f;
                for (int c = sub; c < D4; c += LPR) {
                    const float4 a = ld4(U + iu * D + 4 * c), b = ld4(I + ip * D + 4 * c), q = ld4(I + in * D + 4 * c);
                    dp += dot4(a, b); dn += dot4(a, q); su += dot4(a, a); sp += dot4(b, b);
                }
                dp = group_sum<LPR>(dp); dn = group_sum<LPR>(dn); su = group_sum<LPR>(su); sp = group_sum<LPR>(sp);
                if (sub == 0) {
                    const float s = sigmoidf_(dp - dn);
                    acc[0] += (double)(-logf(gamma + s));
                    acc[1] += (double)su;
                    acc[2] += (double)sp;
                    if (gcoef) gcoef[t] = -invB * (s * (1.0f - s)) / (gamma + s);
                }
            }
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        cdr_store_sys(o, acc[0]); cdr_store_sys(o + 1, acc[1]); cdr_store_sys(o + 2, acc[2]);
    }
    // small grids: the block that signs in last is the finishing pass (no second launch)
    if (ticket && cdr_sign_in_last_wide(ticket, gridDim.x)) loss_finish_body<true>(partials, gridDim.x, B, reg_weight, out4);
}

// generic scalar path (D not a multiple of 4): one wave per interaction
__global__ __launch_bounds__(kBlock) void bpr_fwd_scalar_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                                int D, const int64_t* __restrict__ uid,
                                                                const int64_t* __restrict__ pid,
                                                                const int64_t* __restrict__ nid, int64_t B, float gamma,
                                                                float* __restrict__ gcoef, double* __restrict__ partials) {
    __shared__ double smem[3 * (kBlock / 64)];
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t TW = (int64_t)gridDim.x * (kBlock / 64);
    const float invB = 1.0f / (float)B;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t t = gw; t < B; t += TW) {
        const float* u = U + uid[t] * D; const float* p = I + pid[t] * D; const float* n = I + nid[t] * D;
        float dp = 0.f, dn = 0.f, su = 0.f, sp = 0.f;
        for (int c = lane; c < D; c += 64) { const float a = u[c], b = p[c], q = n[c]; dp += a * b; dn += a * q; su += a * a; sp += b * b; }
        dp = group_sum<64>(dp); dn = group_sum<64>(dn); su = group_sum<64>(su); sp = group_sum<64>(sp);
        if (lane == 0) {
            const float s = sigmoidf_(dp - dn);
            acc[0] += (double)(-logf(gamma + s)); acc[1] += (double)su; acc[2] += (double)sp;
            if (gcoef) gcoef[t] = -invB * (s * (1.0f - s)) / (gamma + s);
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

__global__ __launch_bounds__(kBlock) void loss_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t B,
                                                             float reg_weight, float* __restrict__ out4) {
    loss_finish_body(partials, nblocks, B, reg_weight, out4);
}

// ------------------------------------------------------------------------------------------------ BPR dense backward
// dU[u] += go*g*(p-n) + cu*u ; dI[p] += go*g*u + ci*p ; dI[n] -= go*g*u      (cu = go*reg/(B*||U_b||), ci likewise)
template <int LPR>
__global__ __launch_bounds__(kBlock) void bpr_bwd_dense_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                               int D, const int64_t* __restrict__ uid,
                                                               const int64_t* __restrict__ pid,
                                                               const int64_t* __restrict__ nid, int64_t B,
                                                               const float* __restrict__ gcoef,
                                                               const float* __restrict__ out4, float reg_weight,
                                                               const float* __restrict__ grad_out,
                                                               float* __restrict__ gU, float* __restrict__ gI) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float go = grad_out ? grad_out[0] : 1.0f;
    const float nu = out4[2], ni = out4[3];
    const float cu = (reg_weight != 0.f && nu > 0.f) ? go * reg_weight / ((float)B * nu) : 0.f;
    const float ci = (reg_weight != 0.f && ni > 0.f) ? go * reg_weight / ((float)B * ni) : 0.f;
    for (int64_t t = gg; t < B; t += TG) {
        const int64_t iu = uid[t], ip = pid[t], in = nid[t];
        const float g = go * gcoef[t];
        for (int c = sub; c < D4; c += LPR) {
            const float4 u = ld4(U + iu * D + 4 * c), p = ld4(I + ip * D + 4 * c), n = ld4(I + in * D + 4 * c);
            float* du = gU + iu * D + 4 * c; float* dp = gI + ip * D + 4 * c; float* dn = gI + in * D + 4 * c;
            atomicAdd(du + 0, g * (p.x - n.x) + cu * u.x); atomicAdd(du + 1, g * (p.y - n.y) + cu * u.y);
            atomicAdd(du + 2, g * (p.z - n.z) + cu * u.z); atomicAdd(du + 3, g * (p.w - n.w) + cu * u.w);
            atomicAdd(dp + 0, g * u.x + ci * p.x); atomicAdd(dp + 1, g * u.y + ci * p.y);
            atomicAdd(dp + 2, g * u.z + ci * p.z); atomicAdd(dp + 3, g * u.w + ci * p.w);
            atomicAdd(dn + 0, -g * u.x); atomicAdd(dn + 1, -g * u.y); atomicAdd(dn + 2, -g * u.z); atomicAdd(dn + 3, -g * u.w);
        }
    }
}

__global__ __launch_bounds__(kBlock) void bpr_bwd_dense_scalar_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                                      int D, const int64_t* __restrict__ uid,
                                                                      const int64_t* __restrict__ pid,
                                                                      const int64_t* __restrict__ nid, int64_t B,
                                                                      const float* __restrict__ gcoef,
                                                                      const float* __restrict__ out4, float reg_weight,
                                                                      const float* __restrict__ grad_out,
                                                                      float* __restrict__ gU, float* __restrict__ gI) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t TW = (int64_t)gridDim.x * (kBlock / 64);
    const float go = grad_out ? grad_out[0] : 1.0f;
    const float nu = out4[2], ni = out4[3];
    const float cu = (reg_weight != 0.f && nu > 0.f) ? go * reg_weight / ((float)B * nu) : 0.f;
    const float ci = (reg_weight != 0.f && ni > 0.f) ? go * reg_weight / ((float)B * ni) : 0.f;
    for (int64_t t = gw; t < B; t += TW) {
        const int64_t iu = uid[t], ip = pid[t], in = nid[t];
        const float g = go * gcoef[t];
        for (int c = lane; c < D; c += 64) {
            const float u = U[iu * D + c], p = I[ip * D + c], n = I[in * D + c];
            atomicAdd(gU + iu * D + c, g * (p - n) + cu * u);
            atomicAdd(gI + ip * D + c, g * u + ci * p);
            atomicAdd(gI + in * D + c, -g * u);
        }
    }
}

// ------------------------------------------------------------------------------------------------ pointwise forward
// SAME: the EmbLoss tables are the dot tables (EMCDR-MF, CMF); otherwise (BiTGCF) the reg rows come from other tables, DR <= D floats
// wide (BiTGCF scores rows of the (L + 1) D-wide layer stack and regularises the D-wide ego rows: bitgcf.py:222-240).
template <int LPR, bool SAME>
__device__ __forceinline__ void point_fwd_body(int loss_kind, const float* __restrict__ U,
                                                           const float* __restrict__ I, const float* __restrict__ RU,
                                                           const float* __restrict__ RI, int D, int DR,
                                                           const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                           const float* __restrict__ label, int64_t B,
                                                           float* __restrict__ gcoef, float* __restrict__ scores,
                                                           double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float invB = 1.0f / (float)B;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t base = gg; base < B; base += TG * kUnrollPoint) {
        // ids of the kUnrollPoint rows first, then every row load, then the reductions (vmcnt is in-order: see cdr_step.hip)
        int64_t iu[kUnrollPoint], ii[kUnrollPoint];
        float yl[kUnrollPoint];                  // labels too: a load issued after the reductions would expose a full
#pragma unroll                                   // memory latency at the tail of every iteration
        for (int r = 0; r < kUnrollPoint; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ii[r] = iid[tc]; yl[r] = label[tc];
        }
        float dxs[kUnrollPoint], sus[kUnrollPoint], sis[kUnrollPoint];
        if (D4 <= LPR) {
            float4 a[kUnrollPoint], b[kUnrollPoint], ra[kUnrollPoint], rb[kUnrollPoint];
            const bool live = sub < D4;
#pragma unroll
            for (int r = 0; r < kUnrollPoint; ++r) {
                const int64_t t = base + (int64_t)r * TG;
                a[r] = b[r] = ra[r] = rb[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < B && live) {
                    a[r] = ld4(U + iu[r] * D + 4 * sub);
                    b[r] = ld4(I + ii[r] * D + 4 * sub);
                    if (!SAME && 4 * sub < DR) { ra[r] = ld4(RU + iu[r] * DR + 4 * sub); rb[r] = ld4(RI + ii[r] * DR + 4 * sub); }
                }
            }
#pragma unroll
            for (int r = 0; r < kUnrollPoint; ++r) {
                dxs[r] = dot4(a[r], b[r]);
                sus[r] = SAME ? dot4(a[r], a[r]) : dot4(ra[r], ra[r]);
                sis[r] = SAME ? dot4(b[r], b[r]) : dot4(rb[r], rb[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < kUnrollPoint; ++r) {
                const int64_t t = base + (int64_t)r * TG;
                dxs[r] = sus[r] = sis[r] = 0.f;
                if (t < B) {
                    for (int c = sub; c < D4; c += LPR) {
                        const float4 a = ld4(U + iu[r] * D + 4 * c), b = ld4(I + ii[r] * D + 4 * c);
                        dxs[r] += dot4(a, b);
                        if (SAME) { sus[r] += dot4(a, a); sis[r] += dot4(b, b); }
                        else if (4 * c < DR) {
                            const float4 ra = ld4(RU + iu[r] * DR + 4 * c), rb = ld4(RI + ii[r] * DR + 4 * c);
                            sus[r] += dot4(ra, ra); sis[r] += dot4(rb, rb);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kUnrollPoint; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            float dx = dxs[r], su = sus[r], si = sis[r];
            dx = group_sum<LPR>(dx); su = group_sum<LPR>(su); si = group_sum<LPR>(si);
            if (t < B && sub == 0) {
                const float y = yl[r];
                float l, g, sc;
                if (loss_kind == CDR_LOSS_MSE) {
                    const float d = dx - y;
                    l = d * d; g = 2.0f * d * invB; sc = dx;
                } else {
                    const float p = sigmoidf_(dx);
                    // torch BCELoss: (y-1)*max(log(1-p),-100) - y*max(log(p),-100); backward (p-y)/max((1-p)p,1e-12)
                    l = (y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f);
                    const float pq = (1.0f - p) * p;
                    g = (p - y) / fmaxf(pq, 1e-12f) * invB * pq;
                    sc = p;
                }
                acc[0] += (double)l; acc[1] += (double)su; acc[2] += (double)si;
                if (gcoef) gcoef[t] = g;
                if (scores) scores[t] = sc;
            }
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        cdr_store_sys(o, acc[0]); cdr_store_sys(o + 1, acc[1]); cdr_store_sys(o + 2, acc[2]);       // (see cdr_sign_in_last)
    }
}

template <int LPR, bool SAME>
__global__ __launch_bounds__(kBlock) void point_fwd_kernel(int loss_kind, const float* __restrict__ U, const float* __restrict__ I,
                                                           const float* __restrict__ RU, const float* __restrict__ RI, int D,
                                                           const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                           const float* __restrict__ label, int64_t B, float* __restrict__ gcoef,
                                                           float* __restrict__ scores, double* __restrict__ partials,
                                                           unsigned* __restrict__ ticket, float reg_weight, float* __restrict__ out4,
                                                           uint4* __restrict__ scrub, int64_t scrub_n16) {
    cdr_scrub(scrub, scrub_n16);
    point_fwd_body<LPR, SAME>(loss_kind, U, I, RU, RI, D, D, uid, iid, label, B, gcoef, scores, partials);
    if (ticket && cdr_sign_in_last_wide(ticket, gridDim.x)) loss_finish_body<true>(partials, gridDim.x, B, reg_weight, out4);
}

// Two batches in one launch (blockIdx.y = batch; CMF's two domains on shared tables, BiTGCF's two stacks): same arithmetic per batch,
// half the launches of a launch-bound step.  Batch d's partials live at partials + d * kPairPartials.
struct point_pair {
    const float* U[2]; const float* I[2]; const float* RU[2]; const float* RI[2];
    const int64_t* uid[2]; const int64_t* iid[2]; const float* label[2]; int64_t B[2];
    float* gcoef[2]; float* scores[2]; float* out4[2]; float reg[2];
    const float* go[2]; float* gU[2]; float* gI[2]; float* gRU[2]; float* gRI[2]; float gscale[2];
    int DR;                                      // row width of RU / RI (= D unless the reg tables are narrower: cdr_point_fwd_pair_ex)
};
constexpr size_t kPairPartials = (size_t)(CDR_MAX_PARTIAL_BLOCKS / 2) * CDR_PARTIAL_STRIDE;

// Both batches' finishing pass as ONE pass: the six partial sums are read together and reduced together (block_sum_d<6> adds each
// of them in loss_finish_body's order, so every value is the one two passes give) -- the second pass's loads used to wait for the
// first pass's reduction, a memory latency and two barriers in a launch that is little else at 2 x 2,048 rows.
template <bool SYS>
__device__ __forceinline__ void loss_finish_pair_body(const double* __restrict__ partials, int nblocks, const point_pair& a,
                                                      const float* __restrict__ w, float* __restrict__ total) {
    __shared__ double smem[6 * (kBlock / 64)];
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o0 = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        const double* o1 = o0 + kPairPartials;
        if (SYS) {
            acc[0] += cdr_load_sys(o0); acc[1] += cdr_load_sys(o0 + 1); acc[2] += cdr_load_sys(o0 + 2);
            acc[3] += cdr_load_sys(o1); acc[4] += cdr_load_sys(o1 + 1); acc[5] += cdr_load_sys(o1 + 2);
        } else {
            acc[0] += o0[0]; acc[1] += o0[1]; acc[2] += o0[2]; acc[3] += o1[0]; acc[4] += o1[1]; acc[5] += o1[2];
        }
    }
    block_sum_d<6>(acc, smem);
    if (threadIdx.x == 0) {
        float tot[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float main_loss = (float)(acc[3 * d] / (double)a.B[d]);
            const float nu = (float)sqrt(acc[3 * d + 1]), ni = (float)sqrt(acc[3 * d + 2]);
            float* out4 = a.out4[d];
            out4[1] = main_loss; out4[2] = nu; out4[3] = ni;
            tot[d] = main_loss + a.reg[d] * ((nu + ni) / (float)a.B[d]);
            out4[0] = tot[d];
        }
        if (total) {
#pragma clang fp contract(off)
            total[0] = (0.f + tot[0] * w[0]) + tot[1] * w[1];
        }
    }
}

template <int LPR, bool SAME>
__global__ __launch_bounds__(kBlock) void point_fwd_pair_kernel(int loss_kind, point_pair a, int D, double* __restrict__ partials,
                                                                unsigned* __restrict__ ticket, const float* __restrict__ w,
                                                                float* __restrict__ total, uint4* __restrict__ scrub, int64_t scrub_n16) {
    const int d = blockIdx.y;
    cdr_scrub(scrub, scrub_n16);
    point_fwd_body<LPR, SAME>(loss_kind, a.U[d], a.I[d], a.RU[d], a.RI[d], D, a.DR, a.uid[d], a.iid[d], a.label[d], a.B[d], a.gcoef[d],
                              a.scores[d], partials + d * kPairPartials);
    // small grids: the block (of either batch) that signs in last finishes both losses and their weighted total
    if (ticket && cdr_sign_in_last_wide(ticket, gridDim.x * gridDim.y)) loss_finish_pair_body<true>(partials, gridDim.x, a, w, total);
}

__global__ __launch_bounds__(kBlock) void point_fwd_scalar_kernel(int loss_kind, const float* __restrict__ U,
                                                                  const float* __restrict__ I, const float* __restrict__ RU,
                                                                  const float* __restrict__ RI, int D,
                                                                  const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                                  const float* __restrict__ label, int64_t B,
                                                                  float* __restrict__ gcoef, float* __restrict__ scores,
                                                                  double* __restrict__ partials) {
    __shared__ double smem[3 * (kBlock / 64)];
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t TW = (int64_t)gridDim.x * (kBlock / 64);
    const float invB = 1.0f / (float)B;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t t = gw; t < B; t += TW) {
        const int64_t iu = uid[t], ii = iid[t];
        float dx = 0.f, su = 0.f, si = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float a = U[iu * D + c], b = I[ii * D + c], ra = RU[iu * D + c], rb = RI[ii * D + c];
            dx += a * b; su += ra * ra; si += rb * rb;
        }
        dx = group_sum<64>(dx); su = group_sum<64>(su); si = group_sum<64>(si);
        if (lane == 0) {
            const float y = label[t];
            float l, g, sc;
            if (loss_kind == CDR_LOSS_MSE) { const float d = dx - y; l = d * d; g = 2.0f * d * invB; sc = dx; }
            else {
                const float p = sigmoidf_(dx);
                l = (y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f);
                const float pq = (1.0f - p) * p;
                g = (p - y) / fmaxf(pq, 1e-12f) * invB * pq; sc = p;
            }
            acc[0] += (double)l; acc[1] += (double)su; acc[2] += (double)si;
            if (gcoef) gcoef[t] = g;
            if (scores) scores[t] = sc;
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// ------------------------------------------------------------------------------------------------ pointwise dense backward
// dU[u] += go*g*I[i] ; dI[i] += go*g*U[u] ; dRU[u] += cu*RU[u] ; dRI[i] += ci*RI[i]
__device__ __forceinline__ void point_bwd_dense_body(const float* __restrict__ U, const float* __restrict__ I,
                                                                 const float* __restrict__ RU, const float* __restrict__ RI,
                                                                 int D, const int64_t* __restrict__ uid,
                                                                 const int64_t* __restrict__ iid, int64_t B,
                                                                 const float* __restrict__ gcoef, const float* __restrict__ out4,
                                                                 float reg_weight, const float* __restrict__ grad_out,
                                                                 float* __restrict__ gU, float* __restrict__ gI,
                                                                 float* __restrict__ gRU, float* __restrict__ gRI, float gscale) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t TW = (int64_t)gridDim.x * (kBlock / 64);
    const float go = (grad_out ? grad_out[0] : 1.0f) * gscale;     // gscale: the batch's weight in the caller's total (1 = none)
    const float nu = out4[2], ni = out4[3];
    const float cu = (reg_weight != 0.f && nu > 0.f) ? go * reg_weight / ((float)B * nu) : 0.f;
    const float ci = (reg_weight != 0.f && ni > 0.f) ? go * reg_weight / ((float)B * ni) : 0.f;
    for (int64_t t = gw; t < B; t += TW) {
        const int64_t iu = uid[t], ii = iid[t];
        const float g = go * gcoef[t];
        for (int c = lane; c < D; c += 64) {
            const float u = U[iu * D + c], i = I[ii * D + c];
            float du = g * i, di = g * u;
            if (gRU == gU) du += cu * RU[iu * D + c]; else if (gRU && cu != 0.f) atomicAdd(gRU + iu * D + c, cu * RU[iu * D + c]);
            if (gRI == gI) di += ci * RI[ii * D + c]; else if (gRI && ci != 0.f) atomicAdd(gRI + ii * D + c, ci * RI[ii * D + c]);
            if (gU) atomicAdd(gU + iu * D + c, du);
            if (gI) atomicAdd(gI + ii * D + c, di);
        }
    }
}
__global__ __launch_bounds__(kBlock) void point_bwd_dense_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                                 const float* __restrict__ RU, const float* __restrict__ RI,
                                                                 int D, const int64_t* __restrict__ uid,
                                                                 const int64_t* __restrict__ iid, int64_t B,
                                                                 const float* __restrict__ gcoef, const float* __restrict__ out4,
                                                                 float reg_weight, const float* __restrict__ grad_out,
                                                                 float* __restrict__ gU, float* __restrict__ gI,
                                                                 float* __restrict__ gRU, float* __restrict__ gRI) {
    point_bwd_dense_body(U, I, RU, RI, D, uid, iid, B, gcoef, out4, reg_weight, grad_out, gU, gI, gRU, gRI, 1.0f);
}
__global__ __launch_bounds__(kBlock) void point_bwd_dense_pair_kernel(point_pair a, int D) {
    const int d = blockIdx.y;
    point_bwd_dense_body(a.U[d], a.I[d], a.RU[d], a.RI[d], D, a.uid[d], a.iid[d], a.B[d], a.gcoef[d], a.out4[d], a.reg[d], a.go[d],
                         a.gU[d], a.gI[d], a.gRU[d], a.gRI[d], a.gscale[d]);
}
// both batches' losses, then (optionally) total[0] = w[0] * L_0 + w[1] * L_1: one block, the second batch after the first
__global__ __launch_bounds__(kBlock) void loss_finish_pair_kernel(const double* __restrict__ partials, int nblocks, point_pair a,
                                                                  const float* __restrict__ w, float* __restrict__ total) {
    loss_finish_body(partials, nblocks, a.B[0], a.reg[0], a.out4[0]);
    __syncthreads();
    loss_finish_body(partials + kPairPartials, nblocks, a.B[1], a.reg[1], a.out4[1]);
    __syncthreads();
    if (total && threadIdx.x == 0) {
#pragma clang fp contract(off)
        total[0] = (0.f + a.out4[0][0] * w[0]) + a.out4[1][0] * w[1];
    }
}

// Rows a lane group walks per trip of its loop (<= unroll): small batches get kSmallGrid workgroups' worth of lane groups before any
// group takes a second row -- at the reference's batch (2,048 rows) the unrolled form put 16-32 workgroups on 256 CUs and every group
// worked through 4-8 rows one after the other; the loop is grid-stride, so a wider grid simply leaves the later unroll slots empty.
// Swept on the box at 2,048 rows (C1 through fit).  With every workgroup signing in on ONE word for the finishing pass: 16-32
// workgroups 0.0383 ms, 64: 0.0344, 128: 0.0355, 256: 0.0357 (same-address atomics queue up); with the two-level sign-in
// (cdr_sign_in_last_wide): 64: 0.0339, 128: 0.0335, 256: 0.0336.
constexpr int kSmallGrid = 128;
inline int64_t units_for(int64_t B, int unroll, int per_block) {
    int64_t u = (B + (int64_t)kSmallGrid * per_block - 1) / ((int64_t)kSmallGrid * per_block);
    if (u < 1) u = 1;
    if (u > unroll) u = unroll;
    return (B + u - 1) / u;
}

inline int grid_for(int64_t units, int per_block) {
    int64_t g = (units + per_block - 1) / per_block;
    const int64_t cap = CDR_NUM_CU * 8;       // 2048 blocks = 8 per CU, grid-stride beyond (guide G11)
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
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

extern "C" int cdr_bpr_fwd(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D,
                           const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, float gamma,
                           float reg_weight, float* out4, float* gcoef) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out4);
    CDR_CHECK_ARG(D > 0 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    int grid;
    bool fused_finish = false;
    uint4* zs; int64_t zn;
    cdr_take_scrub(ctx, &zs, &zn);
    if (zs && (D & 3) != 0) { CDR_HIP(cdr_zero_u32(zs, zn * 4, s)); }        // the scalar path has no side job: a plain fill
    cdr_time_scope* ts = new cdr_time_scope(ctx, CDR_TAG_BPR_FWD, s);
    if ((D & 3) == 0) {
        const int lpr = cdr_lpr_for(D);
        grid = grid_for(units_for(B, kUnroll, kBlock / lpr), kBlock / lpr);
        fused_finish = grid <= kSignInMaxBlocks;
        DISPATCH_LPR(lpr, bpr_fwd_kernel<L><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D,
                                              uid, pid, nid, B, gamma, gcoef, ctx->partials, fused_finish ? ctx->tickets : nullptr,
                                              reg_weight, out4, zs, zn));
    } else {
        grid = grid_for(B, kBlock / 64);
        hipLaunchKernelGGL(bpr_fwd_scalar_kernel, dim3(grid), dim3(kBlock), 0, s, user_tab, item_tab, D, uid, pid, nid, B,
                           gamma, gcoef, ctx->partials);
    }
    delete ts;
    CDR_LAUNCH_CHECK();
    if (!fused_finish) hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, s, ctx->partials, grid, B, reg_weight, out4);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bpr_bwd_dense(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D,
                                 const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B,
                                 const float* gcoef, const float* out4, float reg_weight, const float* grad_out,
                                 float* grad_user_tab, float* grad_item_tab) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && gcoef && out4 && grad_user_tab && grad_item_tab);
    CDR_CHECK_ARG(D > 0 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    // From 64 floats a row the wave-per-triple form is the faster one at every batch size measured: each of its atomic instructions
    // covers 64 CONSECUTIVE floats of a gradient row (one coalesced request), where a lane group's float4 form issues four per row with
    // its lanes 16 bytes apart -- forward + backward of a 2,048-triple batch 23.8 -> 19.0 us (D = 64), 36.5 -> 26.7 (D = 128); 8,192
    // triples 40.8 -> 24.1 and 69.2 -> 35.5 (tools/mb_ordered_bwd.py, atomic column; profiles/r05_ab_bwd_dense_form.txt).  Narrower rows
    // would leave most of a wave idle and keep the lane groups.
    if ((D & 3) == 0 && D < 64) {
        const int lpr = cdr_lpr_for(D);
        const int grid = grid_for(B, kBlock / lpr);
        DISPATCH_LPR(lpr, bpr_bwd_dense_kernel<L><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab,
                                              D, uid, pid, nid, B, gcoef, out4, reg_weight, grad_out, grad_user_tab,
                                              grad_item_tab));
    } else {
        const int grid = grid_for(B, kBlock / 64);
        hipLaunchKernelGGL(bpr_bwd_dense_scalar_kernel, dim3(grid), dim3(kBlock), 0, s, user_tab, item_tab, D, uid, pid, nid,
                           B, gcoef, out4, reg_weight, grad_out, grad_user_tab, grad_item_tab);
    }
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_point_fwd(cdr_ctx* ctx, void* stream, int loss_kind, const float* user_tab, const float* item_tab,
                             const float* reg_user_tab, const float* reg_item_tab, int D, const int64_t* uid,
                             const int64_t* iid, const float* label, int64_t B, float reg_weight, float* out4,
                             float* gcoef, float* scores) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && label && out4);
    CDR_CHECK_ARG(D > 0 && B > 0);
    CDR_CHECK_ARG(loss_kind == CDR_LOSS_MSE || loss_kind == CDR_LOSS_BCE);
    if (!reg_user_tab) reg_user_tab = user_tab;
    if (!reg_item_tab) reg_item_tab = item_tab;
    hipStream_t s = (hipStream_t)stream;
    const bool same = (reg_user_tab == user_tab) && (reg_item_tab == item_tab);
    int grid;
    bool fused_finish = false;
    uint4* zs; int64_t zn;
    cdr_take_scrub(ctx, &zs, &zn);
    if (zs && (D & 3) != 0) { CDR_HIP(cdr_zero_u32(zs, zn * 4, s)); }
    if ((D & 3) == 0) {
        const int lpr = cdr_lpr_for(D);
        grid = grid_for(units_for(B, kUnrollPoint, kBlock / lpr), kBlock / lpr);
        fused_finish = grid <= kSignInMaxBlocks;
        unsigned* tk = fused_finish ? ctx->tickets : nullptr;
        if (same) {
            DISPATCH_LPR(lpr, point_fwd_kernel<L, true><<<dim3(grid), dim3(kBlock), 0, s>>>(loss_kind,
                                                  user_tab, item_tab, reg_user_tab, reg_item_tab, D, uid, iid, label, B,
                                                  gcoef, scores, ctx->partials, tk, reg_weight, out4, zs, zn));
        } else {
            DISPATCH_LPR(lpr, point_fwd_kernel<L, false><<<dim3(grid), dim3(kBlock), 0, s>>>(loss_kind,
                                                  user_tab, item_tab, reg_user_tab, reg_item_tab, D, uid, iid, label, B,
                                                  gcoef, scores, ctx->partials, tk, reg_weight, out4, zs, zn));
        }
    } else {
        grid = grid_for(B, kBlock / 64);
        hipLaunchKernelGGL(point_fwd_scalar_kernel, dim3(grid), dim3(kBlock), 0, s, loss_kind, user_tab, item_tab,
                           reg_user_tab, reg_item_tab, D, uid, iid, label, B, gcoef, scores, ctx->partials);
    }
    CDR_LAUNCH_CHECK();
    if (!fused_finish) hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, s, ctx->partials, grid, B, reg_weight, out4);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// Two pointwise batches at once (see point_pair): arrays of two per argument; reg tables NULL = the dot tables.  D % 4 == 0.
extern "C" int cdr_point_fwd_pair(cdr_ctx* ctx, void* stream, int loss_kind, const float* const* user_tab, const float* const* item_tab,
                                  const float* const* reg_user_tab, const float* const* reg_item_tab, int D, const int64_t* const* uid,
                                  const int64_t* const* iid, const float* const* label, const int64_t* B, const float* reg_weight,
                                  float* const* out4, float* const* gcoef, float* const* scores, const float* w, float* total) {
    return cdr_point_fwd_pair_ex(ctx, stream, loss_kind, user_tab, item_tab, reg_user_tab, reg_item_tab, D, D, uid, iid, label, B, reg_weight, out4,
                                 gcoef, scores, w, total);
}

// ... with reg tables reg_D <= D floats wide (both % 4 == 0): BiTGCF's loss in ONE launch -- BCE on rows of the propagated (L + 1) D-wide
// stacks + reg_weight x EmbLoss of the batch's D-wide EGO rows (bitgcf.py:222-240) -- instead of a point loss, two EmbLoss passes with
// their finishing blocks and two scalar adds.
extern "C" int cdr_point_fwd_pair_ex(cdr_ctx* ctx, void* stream, int loss_kind, const float* const* user_tab, const float* const* item_tab,
                                     const float* const* reg_user_tab, const float* const* reg_item_tab, int D, int reg_D,
                                     const int64_t* const* uid, const int64_t* const* iid, const float* const* label, const int64_t* B,
                                     const float* reg_weight, float* const* out4, float* const* gcoef, float* const* scores, const float* w,
                                     float* total) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && label && B && reg_weight && out4 && D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(reg_D > 0 && reg_D <= D && (reg_D & 3) == 0 && (reg_D == D || (reg_user_tab && reg_item_tab)));
    CDR_CHECK_ARG(loss_kind == CDR_LOSS_MSE || loss_kind == CDR_LOSS_BCE);
    CDR_CHECK_ARG((total == nullptr) || w);
    point_pair a{};
    a.DR = reg_D;
    bool same = true;
    int64_t bmax = 0;
    for (int d = 0; d < 2; ++d) {
        CDR_CHECK_ARG(user_tab[d] && item_tab[d] && uid[d] && iid[d] && label[d] && out4[d] && B[d] > 0);
        a.U[d] = user_tab[d]; a.I[d] = item_tab[d];
        a.RU[d] = (reg_user_tab && reg_user_tab[d]) ? reg_user_tab[d] : user_tab[d];
        a.RI[d] = (reg_item_tab && reg_item_tab[d]) ? reg_item_tab[d] : item_tab[d];
        same = same && a.RU[d] == a.U[d] && a.RI[d] == a.I[d];
        CDR_CHECK_ARG(reg_D == D || (reg_user_tab[d] && reg_item_tab[d] && reg_user_tab[d] != user_tab[d] && reg_item_tab[d] != item_tab[d]));   // narrower reg rows: tables of their own
        a.uid[d] = uid[d]; a.iid[d] = iid[d]; a.label[d] = label[d]; a.B[d] = B[d];
        a.gcoef[d] = gcoef ? gcoef[d] : nullptr; a.scores[d] = scores ? scores[d] : nullptr; a.out4[d] = out4[d]; a.reg[d] = reg_weight[d];
        if (B[d] > bmax) bmax = B[d];
    }
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    int grid = grid_for(units_for(bmax, kUnrollPoint, kBlock / lpr), kBlock / lpr);
    if (grid > CDR_MAX_PARTIAL_BLOCKS / 2) grid = CDR_MAX_PARTIAL_BLOCKS / 2;
    const bool fused_finish = 2 * grid <= kSignInMaxBlocks;
    unsigned* tk = fused_finish ? ctx->tickets : nullptr;
    uint4* zs; int64_t zn;
    cdr_take_scrub(ctx, &zs, &zn);
    if (same) { DISPATCH_LPR(lpr, point_fwd_pair_kernel<L, true><<<dim3(grid, 2), dim3(kBlock), 0, s>>>(loss_kind, a, D, ctx->partials, tk, w, total, zs, zn)); }
    else { DISPATCH_LPR(lpr, point_fwd_pair_kernel<L, false><<<dim3(grid, 2), dim3(kBlock), 0, s>>>(loss_kind, a, D, ctx->partials, tk, w, total, zs, zn)); }
    CDR_LAUNCH_CHECK();
    if (!fused_finish) loss_finish_pair_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, a, w, total);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_point_bwd_dense_pair(cdr_ctx* ctx, void* stream, const float* const* user_tab, const float* const* item_tab,
                                        const float* const* reg_user_tab, const float* const* reg_item_tab, int D,
                                        const int64_t* const* uid, const int64_t* const* iid, const int64_t* B,
                                        const float* const* gcoef, const float* const* out4, const float* reg_weight,
                                        const float* const* grad_out, const float* grad_scale, float* const* grad_user_tab,
                                        float* const* grad_item_tab, float* const* grad_reg_user_tab, float* const* grad_reg_item_tab) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && B && gcoef && out4 && reg_weight && grad_out && grad_user_tab && grad_item_tab && D > 0);
    point_pair a{};
    int64_t bmax = 0;
    for (int d = 0; d < 2; ++d) {
        CDR_CHECK_ARG(user_tab[d] && item_tab[d] && uid[d] && iid[d] && gcoef[d] && out4[d] && B[d] > 0);
        a.U[d] = user_tab[d]; a.I[d] = item_tab[d];
        a.gU[d] = grad_user_tab[d]; a.gI[d] = grad_item_tab[d];
        const float* ru = reg_user_tab ? reg_user_tab[d] : nullptr; const float* ri = reg_item_tab ? reg_item_tab[d] : nullptr;
        if (!ru || ru == user_tab[d]) { a.RU[d] = user_tab[d]; a.gRU[d] = a.gU[d]; } else { a.RU[d] = ru; a.gRU[d] = grad_reg_user_tab ? grad_reg_user_tab[d] : nullptr; }
        if (!ri || ri == item_tab[d]) { a.RI[d] = item_tab[d]; a.gRI[d] = a.gI[d]; } else { a.RI[d] = ri; a.gRI[d] = grad_reg_item_tab ? grad_reg_item_tab[d] : nullptr; }
        a.uid[d] = uid[d]; a.iid[d] = iid[d]; a.B[d] = B[d]; a.gcoef[d] = const_cast<float*>(gcoef[d]); a.out4[d] = const_cast<float*>(out4[d]);
        a.reg[d] = reg_weight[d]; a.go[d] = grad_out[d]; a.gscale[d] = grad_scale ? grad_scale[d] : 1.0f;
        if (B[d] > bmax) bmax = B[d];
    }
    const int grid = grid_for(bmax, kBlock / 64);
    point_bwd_dense_pair_kernel<<<dim3(grid, 2), dim3(kBlock), 0, (hipStream_t)stream>>>(a, D);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_point_bwd_dense(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab,
                                   const float* reg_user_tab, const float* reg_item_tab, int D, const int64_t* uid,
                                   const int64_t* iid, int64_t B, const float* gcoef, const float* out4,
                                   float reg_weight, const float* grad_out, float* grad_user_tab, float* grad_item_tab,
                                   float* grad_reg_user_tab, float* grad_reg_item_tab) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && gcoef && out4);
    CDR_CHECK_ARG(D > 0 && B > 0);
    if (!reg_user_tab || reg_user_tab == user_tab) { reg_user_tab = user_tab; grad_reg_user_tab = grad_user_tab; }
    if (!reg_item_tab || reg_item_tab == item_tab) { reg_item_tab = item_tab; grad_reg_item_tab = grad_item_tab; }
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(B, kBlock / 64);
    hipLaunchKernelGGL(point_bwd_dense_kernel, dim3(grid), dim3(kBlock), 0, s, user_tab, item_tab, reg_user_tab,
                       reg_item_tab, D, uid, iid, B, gcoef, out4, reg_weight, grad_out, grad_user_tab, grad_item_tab,
                       grad_reg_user_tab, grad_reg_item_tab);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
