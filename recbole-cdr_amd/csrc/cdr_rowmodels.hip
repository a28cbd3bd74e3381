// Row-wise kernels of the five models SURVEY.md 8f-4 lists after the BASELINE configs (DTCDR, DeepAPF, NATR, DCDCSR; CLFM needs none
// of its own).  Their tables are small (they sit in L2 / Infinity Cache) and their dense layers go through the fp32-MFMA
// contraction (cdr_gemm_f32_ex); what is left are per-row fusions of the reference's elementwise chains: one wave per batch row,
// lanes across the embedding dimension, wave reductions by DPP / permlane swaps (cdr_common.h), no atomics except the
// embedding-gradient scatter (same as cdr_scatter_add_rows), parameter-gradient reductions over the batch through per-row
// partial rows + cdr_colsum (fixed order).
#include "cdr_common.h"
#include <math.h>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / CDR_WAVE;
constexpr int kMaxJ = CDR_ROWMODEL_MAX_DIM / CDR_WAVE;       // registers per lane along D

inline int grid_cap(int64_t blocks) {
    const int64_t cap = CDR_NUM_CU * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
inline int wave_grid(int64_t rows) { return grid_cap((rows + kWaves - 1) / kWaves); }

__device__ __forceinline__ float wave_sum_f(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, CDR_WAVE));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, CDR_WAVE));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

// ---- DTCDR: torch.maximum of the two domains' rows (dtcdr.py:113-119) ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gather_max2_kernel(const float* __restrict__ A, const float* __restrict__ Bt, int D,
                                                             const int64_t* __restrict__ ids, int64_t n,
                                                             float* __restrict__ out, int64_t ldo) {
    const int64_t total = n * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        const int64_t o = ids[r] * D + c;
        out[r * ldo + c] = fmaxf(A[o], Bt[o]);
    }
}

// gradient of torch.maximum: to the larger operand; split evenly on an exact tie
__global__ __launch_bounds__(kBlock) void gather_max2_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Bt, int D,
                                                                 const int64_t* __restrict__ ids, int64_t n,
                                                                 const float* __restrict__ g, int64_t ldg,
                                                                 float* __restrict__ gA, float* __restrict__ gB) {
    const int64_t total = n * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        const int64_t o = ids[r] * D + c;
        const float a = A[o], b = Bt[o], gv = g[r * ldg + c];
        if (a > b) { if (gA) atomicAdd(gA + o, gv); }
        else if (b > a) { if (gB) atomicAdd(gB + o, gv); }
        else {
            if (gA) atomicAdd(gA + o, 0.5f * gv);
            if (gB) atomicAdd(gB + o, 0.5f * gv);
        }
    }
}

// ---- DeepAPF (deepapf.py:69-152) -------------------------------------------------------------------------------------------------
// X[b] = s[b] (.) t[b] ; X[B+b] = o[b] (.) t[b]      (inputs of the attention MLP, both candidates in one [2B, D] operand)
__global__ __launch_bounds__(kBlock) void apf_prod_kernel(const float* __restrict__ s, const float* __restrict__ o,
                                                          const float* __restrict__ t, int64_t B, int D, float* __restrict__ X) {
    const int64_t total = B * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const float tv = t[e];
        X[e] = s[e] * tv;
        X[total + e] = o[e] * tv;
    }
}

__global__ __launch_bounds__(kBlock) void apf_prod_bwd_kernel(const float* __restrict__ s, const float* __restrict__ o,
                                                              const float* __restrict__ t, const float* __restrict__ gX,
                                                              int64_t B, int D, float* __restrict__ gs, float* __restrict__ go,
                                                              float* __restrict__ gt) {
    const int64_t total = B * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const float tv = t[e], g0 = gX[e], g1 = gX[total + e];
        gs[e] = g0 * tv;
        go[e] = g1 * tv;
        gt[e] = g0 * s[e] + g1 * o[e];
    }
}

// softmax over (share, only) with the share score masked where id > n_overlap; e = a_s s + a_o o ; p = sigmoid(wp . (e (.) t))
__global__ __launch_bounds__(kBlock) void apf_combine_kernel(const float* __restrict__ a, const float* __restrict__ s,
                                                             const float* __restrict__ o, const float* __restrict__ t,
                                                             const float* __restrict__ wp, const int64_t* __restrict__ ids,
                                                             int64_t n_overlap, int64_t B, int D, float* __restrict__ p,
                                                             float* __restrict__ alpha_s) {
    const int lane = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * kWaves;
    for (int64_t b = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); b < B; b += wstride) {
        const float as = ids[b] > n_overlap ? -1e31f : a[b], ao = a[B + b];
        const float m = fmaxf(as, ao);
        const float es = expf(as - m), eo = expf(ao - m);
        const float al_s = es / (es + eo), al_o = eo / (es + eo);
        float acc = 0.f;
        for (int d = lane; d < D; d += CDR_WAVE) {
            const int64_t x = b * D + d;
            acc += wp[d] * ((al_s * s[x] + al_o * o[x]) * t[x]);
        }
        acc = wave_sum_f(acc);
        if (lane == 0) { p[b] = sigmoidf_(acc); alpha_s[b] = al_s; }
    }
}

__global__ __launch_bounds__(kBlock) void apf_combine_bwd_kernel(const float* __restrict__ s, const float* __restrict__ o,
                                                                 const float* __restrict__ t, const float* __restrict__ wp,
                                                                 const float* __restrict__ p, const float* __restrict__ alpha_s,
                                                                 const float* __restrict__ gp, int64_t B, int D,
                                                                 float* __restrict__ ga, float* __restrict__ gs,
                                                                 float* __restrict__ go, float* __restrict__ gt,
                                                                 float* __restrict__ gwp_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * kWaves;
    for (int64_t b = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); b < B; b += wstride) {
        const float pv = p[b], al_s = alpha_s[b], al_o = 1.0f - al_s;
        const float gz = gp[b] * pv * (1.0f - pv);
        float d_s = 0.f, d_o = 0.f;
        for (int d = lane; d < D; d += CDR_WAVE) {
            const int64_t x = b * D + d;
            const float sv = s[x], ov = o[x], tv = t[x], w = wp[d];
            const float ev = al_s * sv + al_o * ov;
            const float ge = gz * w * tv;
            gwp_rows[x] = gz * ev * tv;
            gt[x] = gz * w * ev;
            gs[x] = al_s * ge;
            go[x] = al_o * ge;
            d_s += ge * sv;
            d_o += ge * ov;
        }
        d_s = wave_sum_f(d_s);
        d_o = wave_sum_f(d_o);
        if (lane == 0) {
            const float dot = al_s * d_s + al_o * d_o;
            ga[b] = al_s * (d_s - dot);          // exactly 0 on masked rows (alpha_s == 0)
            ga[B + b] = al_o * (d_o - dot);
        }
    }
}

// ---- DCDCSR max-min row normalisation (dcdcsr.py:167-172): y = (x - mean) / (max - mean), mean = (max + min) / 2 -----------------
__global__ __launch_bounds__(kBlock) void maxmin_fwd_kernel(const float* __restrict__ x, int64_t n, int D, float* __restrict__ y,
                                                            float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * kWaves;
    for (int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); r < n; r += wstride) {
        float mx = -INFINITY, mn = INFINITY;
        for (int d = lane; d < D; d += CDR_WAVE) { const float v = x[r * D + d]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
        mx = wave_max_f(mx);
        mn = wave_min_f(mn);
        const float mean = (mx + mn) / 2.0f, h = mx - mean;
        for (int d = lane; d < D; d += CDR_WAVE) y[r * D + d] = (x[r * D + d] - mean) / h;
        if (stats && lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = mx; }
    }
}

// gx_j = gy_j / h - [j in argmax] (G + S) / (2 h n_max) - [j in argmin] (G - S) / (2 h n_min),  G = sum gy, S = sum gy y
// (amax / amin distribute their gradient evenly over ties, as torch does)
__global__ __launch_bounds__(kBlock) void maxmin_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, int64_t n,
                                                            int D, float* __restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * kWaves;
    for (int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); r < n; r += wstride) {
        float mx = -INFINITY, mn = INFINITY;
        for (int d = lane; d < D; d += CDR_WAVE) { const float v = x[r * D + d]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
        mx = wave_max_f(mx);
        mn = wave_min_f(mn);
        const float mean = (mx + mn) / 2.0f, h = mx - mean;
        float G = 0.f, S = 0.f, nmax = 0.f, nmin = 0.f;
        for (int d = lane; d < D; d += CDR_WAVE) {
            const float v = x[r * D + d], g = gy[r * D + d];
            G += g;
            S += g * ((v - mean) / h);
            nmax += v == mx ? 1.f : 0.f;
            nmin += v == mn ? 1.f : 0.f;
        }
        G = wave_sum_f(G); S = wave_sum_f(S); nmax = wave_sum_f(nmax); nmin = wave_sum_f(nmin);
        const float cmax = (G + S) / (2.0f * h * nmax), cmin = (G - S) / (2.0f * h * nmin);
        for (int d = lane; d < D; d += CDR_WAVE) {
            const float v = x[r * D + d];
            float g = gy[r * D + d] / h;
            if (v == mx) g -= cmax;
            if (v == mn) g -= cmin;
            gx[r * D + d] = g;
        }
    }
}

// ---- NATR phase 2 (natr.py:112-156): unit-level attention over the transferred history rows + domain-level gate -----------------
// one wave per batch row; lanes across Dt (J = ceil(Dt / 64) values per lane); history scores of the row in LDS
struct natr_args {
    const float* He;          // [B, L, D]  transfer_layer(source rows of the history)
    const float* pu;          // [B, D]     the row the history belongs to (user_e in overlap_items mode, item_e in overlap_users)
    const float* qi;          // [B, D]     the other side
    const float* mask;        // [B, L]     1 = real history entry, 0 = padding (-10000 added to the score)
    const float* wu; const float* bu;      // unit_attention_layer  [D], [1]
    const float* wd; const float* bd;      // domain_attention_layer [D], [1]
    int64_t B; int L; int D;
};

template <int J>
__global__ __launch_bounds__(kBlock) void natr_fwd_kernel(natr_args a, float* __restrict__ att, float* __restrict__ su_out,
                                                          float* __restrict__ beta_out, float* __restrict__ p_out) {
    __shared__ float sc_s[kWaves][CDR_NATR_MAX_HIST];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* sc = sc_s[wave];
    const int64_t wstride = (int64_t)gridDim.x * kWaves;
    float wu[J], wd[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = lane + j * CDR_WAVE;
        wu[j] = d < a.D ? a.wu[d] : 0.f;
        wd[j] = d < a.D ? a.wd[d] : 0.f;
    }
    const float bu = a.bu[0], bd = a.bd[0];
    for (int64_t b = (int64_t)blockIdx.x * kWaves + wave; b < a.B; b += wstride) {
        float pu[J], qi[J], su[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = lane + j * CDR_WAVE;
            pu[j] = d < a.D ? a.pu[b * a.D + d] : 0.f;
            qi[j] = d < a.D ? a.qi[b * a.D + d] : 0.f;
            su[j] = 0.f;
        }
        const float* He = a.He + b * a.L * a.D;
        float mx = -INFINITY;
        for (int l = 0; l < a.L; ++l) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int d = lane + j * CDR_WAVE;
                if (d < a.D) acc += wu[j] * fmaxf(pu[j] * He[(int64_t)l * a.D + d], 0.f);
            }
            acc = wave_sum_f(acc) + bu;
            acc += a.mask[b * a.L + l] != 0.f ? 0.f : -10000.0f;
            if (lane == 0) sc[l] = acc;
            mx = fmaxf(mx, acc);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes are visible to its own lanes
        float den = 0.f;
        for (int l = 0; l < a.L; ++l) den += expf(sc[l] - mx);
        for (int l = 0; l < a.L; ++l) {
            const float w = expf(sc[l] - mx) / den;
            if (lane == 0) att[b * a.L + l] = w;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int d = lane + j * CDR_WAVE;
                if (d < a.D) su[j] += w * He[(int64_t)l * a.D + d];
            }
        }
        float bs = 0.f, bp = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            bs += wd[j] * fmaxf(su[j] * qi[j], 0.f);
            bp += wd[j] * fmaxf(pu[j] * qi[j], 0.f);
        }
        bs = wave_sum_f(bs) + bd;
        bp = wave_sum_f(bp) + bd;
        const float es = expf(bs), ep = expf(bp);
        const float beta = es / (es + ep);
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = lane + j * CDR_WAVE;
            z += (beta * su[j] + (1.0f - beta) * pu[j]) * qi[j];
            if (d < a.D) su_out[b * a.D + d] = su[j];
        }
        z = wave_sum_f(z);
        if (lane == 0) { beta_out[b] = beta; p_out[b] = sigmoidf_(z); }
    }
}

template <int J>
__global__ __launch_bounds__(kBlock) void natr_bwd_kernel(natr_args a, const float* __restrict__ att, const float* __restrict__ su_in,
                                                          const float* __restrict__ beta_in, const float* __restrict__ p_in,
                                                          const float* __restrict__ gp, float* __restrict__ gHe,
                                                          float* __restrict__ gpu, float* __restrict__ gqi,
                                                          float* __restrict__ gwu_rows, float* __restrict__ gwd_rows,
                                                          float* __restrict__ gb_rows /* [B,2]: d bu, d bd */) {
    __shared__ float ga_s[kWaves][CDR_NATR_MAX_HIST];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* gatt = ga_s[wave];
    const int64_t wstride = (int64_t)gridDim.x * kWaves;
    float wu[J], wd[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = lane + j * CDR_WAVE;
        wu[j] = d < a.D ? a.wu[d] : 0.f;
        wd[j] = d < a.D ? a.wd[d] : 0.f;
    }
    for (int64_t b = (int64_t)blockIdx.x * kWaves + wave; b < a.B; b += wstride) {
        float pu[J], qi[J], su[J], g_su[J], g_pu[J], g_qi[J], g_wu[J];
        const float beta = beta_in[b], pv = p_in[b];
        const float gz = gp[b] * pv * (1.0f - pv);
        float g_beta = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = lane + j * CDR_WAVE;
            pu[j] = d < a.D ? a.pu[b * a.D + d] : 0.f;
            qi[j] = d < a.D ? a.qi[b * a.D + d] : 0.f;
            su[j] = d < a.D ? su_in[b * a.D + d] : 0.f;
            const float g_zu = gz * qi[j];
            g_qi[j] = gz * (beta * su[j] + (1.0f - beta) * pu[j]);
            g_beta += g_zu * (su[j] - pu[j]);
            g_su[j] = beta * g_zu;
            g_pu[j] = (1.0f - beta) * g_zu;
            g_wu[j] = 0.f;
        }
        g_beta = wave_sum_f(g_beta);
        // beta = e^bs / (e^bs + e^bp)
        const float g_bs = g_beta * beta * (1.0f - beta), g_bp = -g_bs;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = lane + j * CDR_WAVE;
            const float xs = su[j] * qi[j], xp = pu[j] * qi[j];
            const float rs = xs > 0.f ? 1.f : 0.f, rp = xp > 0.f ? 1.f : 0.f;
            if (d < a.D) gwd_rows[b * a.D + d] = g_bs * fmaxf(xs, 0.f) + g_bp * fmaxf(xp, 0.f);
            g_su[j] += g_bs * wd[j] * rs * qi[j];
            g_qi[j] += g_bs * wd[j] * rs * su[j] + g_bp * wd[j] * rp * pu[j];
            g_pu[j] += g_bp * wd[j] * rp * qi[j];
        }
        const float* He = a.He + b * a.L * a.D;
        float* gH = gHe + b * a.L * a.D;
        // g_att[l] = g_su . He[l] ; dot = sum_l att[l] g_att[l]
        float dot = 0.f;
        for (int l = 0; l < a.L; ++l) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int d = lane + j * CDR_WAVE;
                if (d < a.D) acc += g_su[j] * He[(int64_t)l * a.D + d];
            }
            acc = wave_sum_f(acc);
            if (lane == 0) gatt[l] = acc;
            dot += att[b * a.L + l] * acc;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        float g_bu = 0.f;
        for (int l = 0; l < a.L; ++l) {
            const float w = att[b * a.L + l];
            const float g_sc = w * (gatt[l] - dot);
            g_bu += g_sc;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int d = lane + j * CDR_WAVE;
                if (d < a.D) {
                    const float h = He[(int64_t)l * a.D + d];
                    const float x = pu[j] * h;
                    const float r = x > 0.f ? 1.f : 0.f;
                    g_wu[j] += g_sc * fmaxf(x, 0.f);
                    g_pu[j] += g_sc * wu[j] * r * h;
                    gH[(int64_t)l * a.D + d] = w * g_su[j] + g_sc * wu[j] * r * pu[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = lane + j * CDR_WAVE;
            if (d < a.D) {
                gpu[b * a.D + d] = g_pu[j];
                gqi[b * a.D + d] = g_qi[j];
                gwu_rows[b * a.D + d] = g_wu[j];
            }
        }
        if (lane == 0) { gb_rows[2 * b] = g_bu; gb_rows[2 * b + 1] = g_bs + g_bp; }
    }
}

}  // namespace

extern "C" {

int cdr_gather_max2(void* stream, const float* A, const float* B, int D, const int64_t* ids, int64_t n, float* out, int64_t ldo) {
    CDR_CHECK_ARG(A && B && ids && out && D > 0 && n >= 0 && ldo >= D);
    if (n == 0) return CDR_OK;
    hipLaunchKernelGGL(gather_max2_kernel, dim3(grid_cap((n * D + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       A, B, D, ids, n, out, ldo);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_gather_max2_bwd(void* stream, const float* A, const float* B, int D, const int64_t* ids, int64_t n, const float* g,
                        int64_t ldg, float* gA, float* gB) {
    CDR_CHECK_ARG(A && B && ids && g && D > 0 && n >= 0 && ldg >= D && (gA || gB));
    if (n == 0) return CDR_OK;
    hipLaunchKernelGGL(gather_max2_bwd_kernel, dim3(grid_cap((n * D + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, A, B, D, ids, n, g, ldg, gA, gB);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_apf_prod(void* stream, const float* s, const float* o, const float* t, int64_t B, int D, float* X) {
    CDR_CHECK_ARG(s && o && t && X && B >= 0 && D > 0);
    if (B == 0) return CDR_OK;
    hipLaunchKernelGGL(apf_prod_kernel, dim3(grid_cap((B * D + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       s, o, t, B, D, X);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_apf_prod_bwd(void* stream, const float* s, const float* o, const float* t, const float* gX, int64_t B, int D,
                     float* gs, float* go, float* gt) {
    CDR_CHECK_ARG(s && o && t && gX && gs && go && gt && B >= 0 && D > 0);
    if (B == 0) return CDR_OK;
    hipLaunchKernelGGL(apf_prod_bwd_kernel, dim3(grid_cap((B * D + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       s, o, t, gX, B, D, gs, go, gt);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_apf_combine(void* stream, const float* a, const float* s, const float* o, const float* t, const float* wp,
                    const int64_t* ids, int64_t n_overlap, int64_t B, int D, float* p, float* alpha_s) {
    CDR_CHECK_ARG(a && s && o && t && wp && ids && p && alpha_s && B >= 0 && D > 0);
    if (B == 0) return CDR_OK;
    hipLaunchKernelGGL(apf_combine_kernel, dim3(wave_grid(B)), dim3(kBlock), 0, (hipStream_t)stream, a, s, o, t, wp, ids,
                       n_overlap, B, D, p, alpha_s);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_apf_combine_bwd(void* stream, const float* s, const float* o, const float* t, const float* wp, const float* p,
                        const float* alpha_s, const float* gp, int64_t B, int D, float* ga, float* gs, float* go, float* gt,
                        float* gwp_rows) {
    CDR_CHECK_ARG(s && o && t && wp && p && alpha_s && gp && ga && gs && go && gt && gwp_rows && B >= 0 && D > 0);
    if (B == 0) return CDR_OK;
    hipLaunchKernelGGL(apf_combine_bwd_kernel, dim3(wave_grid(B)), dim3(kBlock), 0, (hipStream_t)stream, s, o, t, wp, p, alpha_s,
                       gp, B, D, ga, gs, go, gt, gwp_rows);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_maxmin_norm(void* stream, const float* x, int64_t n, int D, float* y, float* stats) {
    CDR_CHECK_ARG(x && y && n >= 0 && D > 0);
    if (n == 0) return CDR_OK;
    hipLaunchKernelGGL(maxmin_fwd_kernel, dim3(wave_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, x, n, D, y, stats);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_maxmin_norm_bwd(void* stream, const float* x, const float* gy, int64_t n, int D, float* gx) {
    CDR_CHECK_ARG(x && gy && gx && n >= 0 && D > 0);
    if (n == 0) return CDR_OK;
    hipLaunchKernelGGL(maxmin_bwd_kernel, dim3(wave_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, x, gy, n, D, gx);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

#define NATR_DISPATCH(KERNEL, ...)                                                                                       \
    switch ((D + CDR_WAVE - 1) / CDR_WAVE) {                                                                             \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(wave_grid(B)), dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(wave_grid(B)), dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(wave_grid(B)), dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL(KERNEL<4>, dim3(wave_grid(B)), dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    }

int cdr_natr_att_fwd(void* stream, const float* He, const float* pu, const float* qi, const float* mask, const float* wu,
                     const float* bu, const float* wd, const float* bd, int64_t B, int L, int D, float* att, float* su,
                     float* beta, float* p) {
    CDR_CHECK_ARG(He && pu && qi && mask && wu && bu && wd && bd && att && su && beta && p);
    CDR_CHECK_ARG(B >= 0 && L >= 1 && L <= CDR_NATR_MAX_HIST && D >= 1 && D <= CDR_ROWMODEL_MAX_DIM);
    static_assert(kMaxJ == 4, "NATR_DISPATCH covers J = 1..4");
    if (B == 0) return CDR_OK;
    natr_args a{He, pu, qi, mask, wu, bu, wd, bd, B, L, D};
    NATR_DISPATCH(natr_fwd_kernel, a, att, su, beta, p);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

int cdr_natr_att_bwd(void* stream, const float* He, const float* pu, const float* qi, const float* mask, const float* wu,
                     const float* bu, const float* wd, const float* bd, int64_t B, int L, int D, const float* att,
                     const float* su, const float* beta, const float* p, const float* gp, float* gHe, float* gpu, float* gqi,
                     float* gwu_rows, float* gwd_rows, float* gb_rows) {
    CDR_CHECK_ARG(He && pu && qi && mask && wu && bu && wd && bd && att && su && beta && p && gp);
    CDR_CHECK_ARG(gHe && gpu && gqi && gwu_rows && gwd_rows && gb_rows);
    CDR_CHECK_ARG(B >= 0 && L >= 1 && L <= CDR_NATR_MAX_HIST && D >= 1 && D <= CDR_ROWMODEL_MAX_DIM);
    if (B == 0) return CDR_OK;
    natr_args a{He, pu, qi, mask, wu, bu, wd, bd, B, L, D};
    NATR_DISPATCH(natr_bwd_kernel, a, att, su, beta, p, gp, gHe, gpu, gqi, gwu_rows, gwd_rows, gb_rows);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

}  // extern "C"
