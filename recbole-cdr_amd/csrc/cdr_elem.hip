// Small fused row-wise / elementwise kernels used by the CoNet, SSCDR and BiTGCF mirrors (SURVEY.md 2.2 K4, K8, K10,
// K11, K12 minus the SpMM).  All HBM-bound streaming kernels: float4 (16 B) per lane where rows are 16-B aligned,
// one wave per row for row reductions, fixed-order two-pass reductions for scalars.
#include "cdr_common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_cap(int64_t blocks) {
    const int64_t cap = CDR_NUM_CU * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---- strided row gather / scatter-add (concatenated [u ; i] inputs of CoNet: conet.py:106-111) -------------------
__global__ __launch_bounds__(kBlock) void gather_rows_ld_kernel(const float* __restrict__ tab, int D,
                                                                const int64_t* __restrict__ ids, int64_t n,
                                                                float* __restrict__ out, int64_t ldo) {
    const int64_t total = n * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        out[r * ldo + c] = tab[ids[r] * D + c];
    }
}

__global__ __launch_bounds__(kBlock) void scatter_add_rows_ld_kernel(float* __restrict__ grad_tab, int D,
                                                                     const int64_t* __restrict__ ids, int64_t n,
                                                                     const float* __restrict__ src, int64_t lds) {
    const int64_t total = n * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        atomicAdd(grad_tab + ids[r] * D + c, src[r * lds + c]);
    }
}

__global__ __launch_bounds__(kBlock) void overlap_mask_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_overlap,
                                                              float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) out[e] = ids[e] < n_overlap ? 1.0f : 0.0f;
}

// out[m,:] = scale[m] * x[m,:]
__global__ __launch_bounds__(kBlock) void rowscale_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          int64_t M, int64_t N, float* __restrict__ out) {
    const int64_t total = M * N, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) out[e] = x[e] * scale[e / N];
}

// out[i,:] = act(P[i,:] + q[:])   (CoNet full-sort: first layer split into item part P and user part q)
__global__ __launch_bounds__(kBlock) void bcast_add_act_kernel(const float* __restrict__ P, const float* __restrict__ q,
                                                               int64_t N, int64_t H, int act, float* __restrict__ out) {
    const int64_t total = N * H, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        float v = P[e] + q[e % H];
        if (act == CDR_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == CDR_ACT_TANH) v = tanhf(v);
        else if (act == CDR_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        out[e] = v;
    }
}

// ---- BCE on probabilities (nn.BCELoss: conet.py:63,195-196) --------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bce_partial_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                             int64_t n, double* __restrict__ partials) {
    __shared__ double smem[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        const float pv = p[e], yv = y[e];
        acc[0] += (double)((yv - 1.0f) * fmaxf(logf(1.0f - pv), -100.0f) - yv * fmaxf(logf(pv), -100.0f));
    }
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
}

// mode 0: out = sum / n (mean) ; mode 1: out = sqrt(sum) (Frobenius norm)
__global__ __launch_bounds__(kBlock) void scalar_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t n,
                                                               int mode, float* __restrict__ out1) {
    __shared__ double smem[4];
    double acc[1] = {0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) acc[0] += partials[(size_t)b * CDR_PARTIAL_STRIDE];
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) out1[0] = mode == 0 ? (float)(acc[0] / (double)n) : (float)sqrt(acc[0]);
}

__global__ __launch_bounds__(kBlock) void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n,
                                                         const float* __restrict__ grad_out, float* __restrict__ gp) {
    const float go = (grad_out ? grad_out[0] : 1.0f) / (float)n;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        const float pv = p[e];
        gp[e] = go * (pv - y[e]) / fmaxf((1.0f - pv) * pv, 1e-12f);
    }
}

__global__ __launch_bounds__(kBlock) void sqsum_partial_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ partials) {
    __shared__ double smem[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) acc[0] += (double)(x[e] * x[e]);
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
}

// gx (+)= go * x / norm   (d||x||_F / dx; 0 where the norm is 0, as torch.norm's backward)
__global__ __launch_bounds__(kBlock) void frobenius_bwd_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ norm,
                                                               const float* __restrict__ grad_out, float* __restrict__ gx,
                                                               int accumulate) {
    const float nv = norm[0];
    const float c = nv > 0.f ? (grad_out ? grad_out[0] : 1.0f) / nv : 0.f;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride)
        gx[e] = (accumulate ? gx[e] : 0.f) + c * x[e];
}

// ---- SSCDR: squared-norm "normalize" (sscdr.py:120-124) ------------------------------------------------------------
//   len = sum x^2 ; y = x / (len > 1 ? len : 1)          one wave per row
__global__ __launch_bounds__(kBlock) void sqnorm_normalize_fwd_kernel(const float* __restrict__ x, int64_t rows, int D,
                                                                      float* __restrict__ y, float* __restrict__ len_out) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < rows; r += TW) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) { const float v = x[r * D + c]; s += v * v; }
        s = group_sum<64>(s);
        const float nrm = s > 1.0f ? s : 1.0f;
        for (int c = lane; c < D; c += 64) y[r * D + c] = x[r * D + c] / nrm;
        if (lane == 0 && len_out) len_out[r] = s;
    }
}

//   len > 1: gx = gy/len - 2 x (x . gy) / len^2 ; else gx = gy
__global__ __launch_bounds__(kBlock) void sqnorm_normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ len,
                                                                      const float* __restrict__ gy, int64_t rows, int D,
                                                                      float* __restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < rows; r += TW) {
        const float L = len[r];
        if (L > 1.0f) {
            float d = 0.f;
            for (int c = lane; c < D; c += 64) d += x[r * D + c] * gy[r * D + c];
            d = group_sum<64>(d);
            const float k = 2.0f * d / (L * L);
            for (int c = lane; c < D; c += 64) gx[r * D + c] = gy[r * D + c] / L - k * x[r * D + c];
        } else {
            for (int c = lane; c < D; c += 64) gx[r * D + c] = gy[r * D + c];
        }
    }
}

// ---- SSCDR: nn.TripletMarginLoss(margin, p=2, eps=1e-6) (sscdr.py:69,142-144) ------------------------------------
//   d(a,b) = || a - b + eps ||_2 ; l = max(d_ap - d_an + margin, 0) ; mean over rows
__global__ __launch_bounds__(kBlock) void triplet_fwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                             const float* __restrict__ n, int64_t rows, int D, float margin,
                                                             float eps, float* __restrict__ dap, float* __restrict__ dan,
                                                             double* __restrict__ partials) {
    __shared__ double smem[4];
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    double acc[1] = {0.0};
    for (int64_t r = w; r < rows; r += TW) {
        float sp = 0.f, sn = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float av = a[r * D + c];
            const float dp = av - p[r * D + c] + eps, dn = av - n[r * D + c] + eps;
            sp += dp * dp; sn += dn * dn;
        }
        sp = group_sum<64>(sp); sn = group_sum<64>(sn);
        if (lane == 0) {
            const float d1 = sqrtf(sp), d2 = sqrtf(sn);
            dap[r] = d1; dan[r] = d2;
            acc[0] += (double)fmaxf(d1 - d2 + margin, 0.0f);
        }
    }
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
}

__global__ __launch_bounds__(kBlock) void triplet_bwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                             const float* __restrict__ n, int64_t rows, int D, float margin,
                                                             float eps, const float* __restrict__ dap,
                                                             const float* __restrict__ dan, const float* __restrict__ grad_out,
                                                             float* __restrict__ ga, float* __restrict__ gp,
                                                             float* __restrict__ gn) {
    const float go = (grad_out ? grad_out[0] : 1.0f) / (float)rows;
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const float d1 = dap[r], d2 = dan[r];
        float va = 0.f, vp = 0.f, vn = 0.f;
        if (d1 - d2 + margin > 0.0f) {
            const float av = a[e];
            const float u1 = d1 > 0.f ? (av - p[e] + eps) / d1 : 0.f;
            const float u2 = d2 > 0.f ? (av - n[e] + eps) / d2 : 0.f;
            va = go * (u1 - u2); vp = -go * u1; vn = go * u2;
        }
        if (ga) ga[e] = va;
        if (gp) gp[e] = vp;
        if (gn) gn[e] = vn;
    }
}

// ---- SSCDR map phase in one pass (sscdr.py:161-172): loss_s = MSE(mapped source rows, target rows); loss_u = triplet(normalize(target
// rows), normalize(mapped interacted rows), normalize(mapped non-interacted rows)); total = loss_s + lambda loss_u -- and, in the same pass,
// d total / d inputs for a unit upstream gradient (the arithmetic of mse_bwd_kernel, triplet_bwd_kernel and sqnorm_normalize_bwd_kernel
// chained per row).  M3 [3 n, D] = the mapping of [source rows ; interacted ; non-interacted], T [n, D]; one wave per row.
//   partials: [0] sum (ms - t)^2, [1] sum of the hinge terms
__device__ __forceinline__ float sq_norm_bwd(float x, float g, float L, float dot) {       // sqnorm_normalize_bwd_kernel, one element
    return L > 1.0f ? g / L - (2.0f * dot / (L * L)) * x : g;
}

__global__ __launch_bounds__(kBlock) void sscdr_map_loss_kernel(const float* __restrict__ M3, const float* __restrict__ T, int64_t n, int D,
                                                                float margin, float eps, float lambda, float* __restrict__ G3,
                                                                float* __restrict__ GT, double* __restrict__ partials) {
    __shared__ double smem[8];
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    const float cs = 2.0f / (float)(n * D);                  // mse_bwd_kernel: go * 2 / numel
    const float cu = lambda / (float)n;                      // triplet_bwd_kernel: (lambda go) / rows
    double acc[2] = {0.0, 0.0};
    for (int64_t r = w; r < n; r += TW) {
        const float* ms = M3 + r * D;
        const float* mp = M3 + (n + r) * D;
        const float* mn = M3 + (2 * n + r) * D;
        const float* t = T + r * D;
        float se = 0.f, lt = 0.f, lp = 0.f, lq = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float d = ms[c] - t[c];
            se += d * d; lt += t[c] * t[c]; lp += mp[c] * mp[c]; lq += mn[c] * mn[c];
        }
        se = group_sum<64>(se); lt = group_sum<64>(lt); lp = group_sum<64>(lp); lq = group_sum<64>(lq);
        const float nt = lt > 1.0f ? lt : 1.0f, np_ = lp > 1.0f ? lp : 1.0f, nq = lq > 1.0f ? lq : 1.0f;
        float sp = 0.f, sn = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float av = t[c] / nt;
            const float dp = av - mp[c] / np_ + eps, dn = av - mn[c] / nq + eps;
            sp += dp * dp; sn += dn * dn;
        }
        sp = group_sum<64>(sp); sn = group_sum<64>(sn);
        const float d1 = sqrtf(sp), d2 = sqrtf(sn);
        const bool active = d1 - d2 + margin > 0.0f;
        if (lane == 0) { acc[0] += (double)se; acc[1] += (double)fmaxf(d1 - d2 + margin, 0.0f); }
        // gradients of the triplet term w.r.t. the three normalised rows, then through the normalisation (needs x . g per row)
        float da = 0.f, dpp = 0.f, dq = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float av = t[c] / nt;
            const float u1 = (active && d1 > 0.f) ? (av - mp[c] / np_ + eps) / d1 : 0.f;
            const float u2 = (active && d2 > 0.f) ? (av - mn[c] / nq + eps) / d2 : 0.f;
            da += t[c] * (cu * (u1 - u2)); dpp += mp[c] * (-cu * u1); dq += mn[c] * (cu * u2);
        }
        da = group_sum<64>(da); dpp = group_sum<64>(dpp); dq = group_sum<64>(dq);
        for (int c = lane; c < D; c += 64) {
            const float av = t[c] / nt;
            const float u1 = (active && d1 > 0.f) ? (av - mp[c] / np_ + eps) / d1 : 0.f;
            const float u2 = (active && d2 > 0.f) ? (av - mn[c] / nq + eps) / d2 : 0.f;
            const float gs = cs * (ms[c] - t[c]);
            G3[r * D + c] = gs;
            G3[(n + r) * D + c] = sq_norm_bwd(mp[c], -cu * u1, lp, dpp);
            G3[(2 * n + r) * D + c] = sq_norm_bwd(mn[c], cu * u2, lq, dq);
            GT[r * D + c] = -gs + sq_norm_bwd(t[c], cu * (u1 - u2), lt, da);
        }
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
        partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE + 1] = acc[1];
    }
}

// out3 = {loss_s + lambda loss_u, loss_s, loss_u}
__global__ __launch_bounds__(kBlock) void sscdr_map_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t n, int D,
                                                                  float lambda, float* __restrict__ out3) {
    __shared__ double smem[8];
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        acc[0] += partials[(size_t)b * CDR_PARTIAL_STRIDE];
        acc[1] += partials[(size_t)b * CDR_PARTIAL_STRIDE + 1];
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        const float ls = (float)(acc[0] / (double)(n * D)), lu = (float)(acc[1] / (double)n);
        out3[1] = ls; out3[2] = lu;
        out3[0] = ls + lambda * lu;
    }
}

// x *= s[0], y *= s[0] unless s[0] is exactly 1 (gradients made for a unit upstream gradient: `loss.backward()` passes 1)
__global__ __launch_bounds__(kBlock) void scale2_unless_one_kernel(const float* __restrict__ s, float* __restrict__ x, int64_t nx,
                                                                   float* __restrict__ y, int64_t ny) {
    const float sc = s[0];
    if (sc == 1.0f) return;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nx + ny; e += stride) {
        if (e < nx) x[e] *= sc; else y[e - nx] *= sc;
    }
}

// ---- EmbLoss alone (bitgcf.py:231-233: norms of the EGO rows, different width from the propagated rows) -----------
__global__ __launch_bounds__(kBlock) void embloss_partial_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                 const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                                 int64_t B, double* __restrict__ partials) {
    __shared__ double smem[8];
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    double acc[2] = {0.0, 0.0};
    for (int64_t t = w; t < B; t += TW) {
        const int64_t iu = uid[t], ii = iid[t];
        float su = 0.f, si = 0.f;
        for (int c = lane; c < D; c += 64) { const float a = U[iu * D + c], b = I[ii * D + c]; su += a * a; si += b * b; }
        su = group_sum<64>(su); si = group_sum<64>(si);
        if (lane == 0) { acc[0] += (double)su; acc[1] += (double)si; }
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
        partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE + 1] = acc[1];
    }
}

// out3 = {(nu + ni)/B, nu, ni}
__global__ __launch_bounds__(kBlock) void embloss_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t B,
                                                                float* __restrict__ out3) {
    __shared__ double smem[8];
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        acc[0] += partials[(size_t)b * CDR_PARTIAL_STRIDE];
        acc[1] += partials[(size_t)b * CDR_PARTIAL_STRIDE + 1];
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        const float nu = (float)sqrt(acc[0]), ni = (float)sqrt(acc[1]);
        out3[1] = nu; out3[2] = ni; out3[0] = (nu + ni) / (float)B;
    }
}

__global__ __launch_bounds__(kBlock) void embloss_bwd_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                             const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                             int64_t B, const float* __restrict__ out3,
                                                             const float* __restrict__ grad_out, float* __restrict__ gU,
                                                             float* __restrict__ gI) {
    const float go = grad_out ? grad_out[0] : 1.0f;
    const float cu = out3[1] > 0.f ? go / ((float)B * out3[1]) : 0.f;
    const float ci = out3[2] > 0.f ? go / ((float)B * out3[2]) : 0.f;
    const int64_t total = B * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t t = e / D;
        const int c = (int)(e - t * D);
        const int64_t iu = uid[t], ii = iid[t];
        atomicAdd(gU + iu * D + c, cu * U[iu * D + c]);
        atomicAdd(gI + ii * D + c, ci * I[ii * D + c]);
    }
}

// both domains' EmbLoss gradients in one launch (blockIdx.y = domain); norms[d] -> {||U_b||_F, ||I_b||_F}, the row coefficient is
// scale[d] * grad_out[d][0] / (B ||rows||)
struct embloss_pair { const float* U[2]; const float* I[2]; const int64_t* uid[2]; const int64_t* iid[2]; int64_t B[2];
                      const float* norms[2]; const float* go[2]; float scale[2]; float* gU[2]; float* gI[2]; };
__global__ __launch_bounds__(kBlock) void embloss_bwd_pair_kernel(embloss_pair a, int D) {
    const int d = blockIdx.y;
    const float* __restrict__ U = a.U[d]; const float* __restrict__ I = a.I[d];
    const int64_t* __restrict__ uid = a.uid[d]; const int64_t* __restrict__ iid = a.iid[d];
    float* __restrict__ gU = a.gU[d]; float* __restrict__ gI = a.gI[d];
    const int64_t B = a.B[d];
    const float go = (a.go[d] ? a.go[d][0] : 1.0f) * a.scale[d];
    const float nu = a.norms[d][0], ni = a.norms[d][1];
    const float cu = nu > 0.f ? go / ((float)B * nu) : 0.f;
    const float ci = ni > 0.f ? go / ((float)B * ni) : 0.f;
    const int64_t total = B * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t t = e / D;
        const int c = (int)(e - t * D);
        const int64_t iu = uid[t], ii = iid[t];
        atomicAdd(gU + iu * D + c, cu * U[iu * D + c]);
        atomicAdd(gI + ii * D + c, ci * I[ii * D + c]);
    }
}

}  // namespace

#define EL_GRID(total) dim3(grid_cap(((total) + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream

extern "C" int cdr_gather_rows_ld(void* stream, const float* tab, int D, const int64_t* ids, int64_t n, float* out, int64_t ldo) {
    CDR_CHECK_ARG(tab && ids && out && D > 0 && n > 0 && ldo >= D);
    gather_rows_ld_kernel<<<EL_GRID(n * D)>>>(tab, D, ids, n, out, ldo);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// Deterministic counterpart of cdr_scatter_add_rows_ld: the ids come SORTED (cdr_sort_ids / cdr_sort_ids_small), every distinct row
// is written once with the sum of its occurrences' rows in occurrence order -- no float atomics, run-to-run reproducible.
// grad_tab must be zero (rows the batch does not touch stay zero).
namespace {
__global__ __launch_bounds__(kBlock) void scatter_rows_sorted_kernel(float* __restrict__ grad_tab, int D, const uint32_t* __restrict__ keys,
                                                                     const uint32_t* __restrict__ perm, int64_t n,
                                                                     const float* __restrict__ src, int64_t lds) {
    const int D4 = D >> 2;
    const int64_t total = n * D4, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < total; w += stride) {
        const int64_t q = w / D4;
        const int ch = (int)(w - q * D4);
        const uint32_t row = keys[q];
        if (q > 0 && keys[q - 1] == row) continue;                     // segment heads only
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t e = q; e < n && keys[e] == row; ++e) {
            const float4 x = *reinterpret_cast<const float4*>(src + (int64_t)perm[e] * lds + 4 * ch);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        *reinterpret_cast<float4*>(grad_tab + (int64_t)row * D + 4 * ch) = acc;
    }
}
}  // namespace

extern "C" int cdr_scatter_rows_sorted(void* stream, float* grad_tab, int D, const uint32_t* keys_sorted, const uint32_t* perm, int64_t n,
                                       const float* src, int64_t lds) {
    CDR_CHECK_ARG(grad_tab && keys_sorted && perm && src && D > 0 && (D & 3) == 0 && n > 0 && lds >= D && (lds & 3) == 0);
    scatter_rows_sorted_kernel<<<EL_GRID(n * (D >> 2))>>>(grad_tab, D, keys_sorted, perm, n, src, lds);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_scatter_add_rows_ld(void* stream, float* grad_tab, int D, const int64_t* ids, int64_t n, const float* src,
                                       int64_t lds) {
    CDR_CHECK_ARG(grad_tab && ids && src && D > 0 && n > 0 && lds >= D);
    scatter_add_rows_ld_kernel<<<EL_GRID(n * D)>>>(grad_tab, D, ids, n, src, lds);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_overlap_mask(void* stream, const int64_t* ids, int64_t n, int64_t n_overlap, float* out) {
    CDR_CHECK_ARG(ids && out && n > 0);
    overlap_mask_kernel<<<EL_GRID(n)>>>(ids, n, n_overlap, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_rowscale(void* stream, const float* x, const float* scale, int64_t M, int64_t N, float* out) {
    CDR_CHECK_ARG(x && scale && out && M > 0 && N > 0);
    rowscale_kernel<<<EL_GRID(M * N)>>>(x, scale, M, N, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bcast_add_act(void* stream, const float* P, const float* q, int64_t N, int64_t H, int act, float* out) {
    CDR_CHECK_ARG(P && q && out && N > 0 && H > 0);
    bcast_add_act_kernel<<<EL_GRID(N * H)>>>(P, q, N, H, act, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

static inline int partial_grid(int64_t n) {
    int64_t g = (n + kBlock * 4 - 1) / (kBlock * 4);
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    if (g > CDR_MAX_PARTIAL_BLOCKS) g = CDR_MAX_PARTIAL_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int cdr_bce_prob_fwd(cdr_ctx* ctx, void* stream, const float* p, const float* y, int64_t n, float* out1) {
    CDR_CHECK_ARG(ctx && p && y && out1 && n > 0);
    hipStream_t s = (hipStream_t)stream;
    const int g = partial_grid(n);
    bce_partial_kernel<<<dim3(g), dim3(kBlock), 0, s>>>(p, y, n, ctx->partials);
    CDR_LAUNCH_CHECK();
    scalar_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, g, n, 0, out1);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bce_prob_bwd(void* stream, const float* p, const float* y, int64_t n, const float* grad_out, float* gp) {
    CDR_CHECK_ARG(p && y && gp && n > 0);
    bce_bwd_kernel<<<EL_GRID(n)>>>(p, y, n, grad_out, gp);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_frobenius_fwd(cdr_ctx* ctx, void* stream, const float* x, int64_t n, float* out1) {
    CDR_CHECK_ARG(ctx && x && out1 && n > 0);
    hipStream_t s = (hipStream_t)stream;
    const int g = partial_grid(n);
    sqsum_partial_kernel<<<dim3(g), dim3(kBlock), 0, s>>>(x, n, ctx->partials);
    CDR_LAUNCH_CHECK();
    scalar_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, g, n, 1, out1);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_frobenius_bwd(void* stream, const float* x, int64_t n, const float* norm, const float* grad_out, float* gx,
                                 int accumulate) {
    CDR_CHECK_ARG(x && norm && gx && n > 0);
    frobenius_bwd_kernel<<<EL_GRID(n)>>>(x, n, norm, grad_out, gx, accumulate);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_sqnorm_normalize_fwd(void* stream, const float* x, int64_t rows, int D, float* y, float* len_out) {
    CDR_CHECK_ARG(x && y && rows > 0 && D > 0);
    sqnorm_normalize_fwd_kernel<<<dim3(grid_cap((rows + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream>>>(x, rows, D, y, len_out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_sqnorm_normalize_bwd(void* stream, const float* x, const float* len, const float* gy, int64_t rows, int D,
                                        float* gx) {
    CDR_CHECK_ARG(x && len && gy && gx && rows > 0 && D > 0);
    sqnorm_normalize_bwd_kernel<<<dim3(grid_cap((rows + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream>>>(x, len, gy, rows, D, gx);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_triplet_fwd(cdr_ctx* ctx, void* stream, const float* a, const float* p, const float* n, int64_t rows, int D,
                               float margin, float eps, float* out1, float* dap, float* dan) {
    CDR_CHECK_ARG(ctx && a && p && n && out1 && dap && dan && rows > 0 && D > 0);
    hipStream_t s = (hipStream_t)stream;
    int g = grid_cap((rows + 3) / 4);
    if (g > CDR_MAX_PARTIAL_BLOCKS) g = CDR_MAX_PARTIAL_BLOCKS;
    triplet_fwd_kernel<<<dim3(g), dim3(kBlock), 0, s>>>(a, p, n, rows, D, margin, eps, dap, dan, ctx->partials);
    CDR_LAUNCH_CHECK();
    scalar_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, g, rows, 0, out1);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_sscdr_map_loss(cdr_ctx* ctx, void* stream, const float* mapped3, const float* target_rows, int64_t n, int D,
                                  float margin, float eps, float lambda, float* out3, float* g_mapped3, float* g_target_rows) {
    CDR_CHECK_ARG(ctx && mapped3 && target_rows && out3 && g_mapped3 && g_target_rows && n > 0 && D > 0);
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_cap((n + 3) / 4);
    sscdr_map_loss_kernel<<<dim3(g), dim3(kBlock), 0, s>>>(mapped3, target_rows, n, D, margin, eps, lambda, g_mapped3, g_target_rows,
                                                           ctx->partials);
    CDR_LAUNCH_CHECK();
    sscdr_map_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, g, n, D, lambda, out3);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_scale2_unless_one(void* stream, const float* scale_dev, float* x, int64_t nx, float* y, int64_t ny) {
    CDR_CHECK_ARG(scale_dev && nx >= 0 && ny >= 0 && (nx == 0 || x) && (ny == 0 || y));
    if (nx + ny == 0) return CDR_OK;
    scale2_unless_one_kernel<<<EL_GRID(nx + ny)>>>(scale_dev, x, nx, y, ny);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_triplet_bwd(void* stream, const float* a, const float* p, const float* n, int64_t rows, int D, float margin,
                               float eps, const float* dap, const float* dan, const float* grad_out, float* ga, float* gp,
                               float* gn) {
    CDR_CHECK_ARG(a && p && n && dap && dan && rows > 0 && D > 0 && (ga || gp || gn));
    triplet_bwd_kernel<<<EL_GRID(rows * D)>>>(a, p, n, rows, D, margin, eps, dap, dan, grad_out, ga, gp, gn);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_embloss_fwd(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D,
                               const int64_t* uid, const int64_t* iid, int64_t B, float* out3) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && out3 && B > 0 && D > 0);
    hipStream_t s = (hipStream_t)stream;
    int g = grid_cap((B + 3) / 4);
    if (g > CDR_MAX_PARTIAL_BLOCKS) g = CDR_MAX_PARTIAL_BLOCKS;
    embloss_partial_kernel<<<dim3(g), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, iid, B, ctx->partials);
    CDR_LAUNCH_CHECK();
    embloss_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, g, B, out3);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_embloss_bwd_dense(void* stream, const float* user_tab, const float* item_tab, int D, const int64_t* uid,
                                     const int64_t* iid, int64_t B, const float* out3, const float* grad_out,
                                     float* grad_user_tab, float* grad_item_tab) {
    CDR_CHECK_ARG(user_tab && item_tab && uid && iid && out3 && grad_user_tab && grad_item_tab && B > 0 && D > 0);
    embloss_bwd_kernel<<<EL_GRID(B * D)>>>(user_tab, item_tab, D, uid, iid, B, out3, grad_out, grad_user_tab, grad_item_tab);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_embloss_bwd_dense_pair(void* stream, const float* const* user_tab, const float* const* item_tab, int D,
                                          const int64_t* const* uid, const int64_t* const* iid, const int64_t* B, const float* const* norms,
                                          const float* const* grad_out, const float* scale, float* const* grad_user_tab,
                                          float* const* grad_item_tab) {
    CDR_CHECK_ARG(user_tab && item_tab && uid && iid && B && norms && scale && grad_user_tab && grad_item_tab && D > 0);
    embloss_pair a{};
    int64_t bmax = 0;
    for (int d = 0; d < 2; ++d) {
        CDR_CHECK_ARG(user_tab[d] && item_tab[d] && uid[d] && iid[d] && norms[d] && grad_user_tab[d] && grad_item_tab[d] && B[d] > 0);
        a.U[d] = user_tab[d]; a.I[d] = item_tab[d]; a.uid[d] = uid[d]; a.iid[d] = iid[d]; a.B[d] = B[d]; a.norms[d] = norms[d];
        a.go[d] = grad_out ? grad_out[d] : nullptr; a.scale[d] = scale[d]; a.gU[d] = grad_user_tab[d]; a.gI[d] = grad_item_tab[d];
        if (B[d] > bmax) bmax = B[d];
    }
    embloss_bwd_pair_kernel<<<dim3(grid_cap((bmax * D + kBlock - 1) / kBlock), 2), dim3(kBlock), 0, (hipStream_t)stream>>>(a, D);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
