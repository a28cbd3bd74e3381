// BiTGCF propagation kernels (recbole_cdr/model/cross_domain_recommender/bitgcf.py:130-205):
//   spmm_csr_kernel      side = A x E over a CSR adjacency, with the graph-layer math fused into the epilogue
//                          fwd : new = E + side + E (.) side            (bitgcf.py:130-135, dropout = identity)
//                          bwd : gE  = gnew (.) (1 + side) + A x tmp     (A is symmetric: A^T = A), tmp = gnew (.) (1 + E)
//   transfer_*_kernel    bi-directional transfer on the overlapped rows (bitgcf.py:137-172), other rows pass through
//   l2_normalize_*       F.normalize(x, p=2, dim=1) written straight into a column block of the concat buffer
//   colblock_mean_*      connect_way == 'mean'
// One lane group of LPR = D/4 lanes per output row (float4 per lane): 64/LPR rows per wave-instruction; the gather of
// E[col] rows is the HBM/L2 traffic that bounds the kernel (4D+12 bytes per non-zero).
#include "cdr_common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_cap(int64_t blocks) {
    const int64_t cap = CDR_NUM_CU * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// MODE 0: out = acc ; MODE 1: side_out = acc, out = x + acc + x*acc (x = X[r]) ; MODE 2: out = G[r]*(1 + S[r]) + acc
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                          const float* __restrict__ values, int64_t n_rows,
                                                          const float* __restrict__ E, int D, const float* __restrict__ X,
                                                          const float* __restrict__ S, float* __restrict__ side_out,
                                                          float* __restrict__ out) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    for (int64_t r = gg; r < n_rows; r += TG) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        for (int ch = sub; ch < D4; ch += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int64_t j = b;
            for (; j + 1 < e; j += 2) {          // two independent row fetches in flight
                const int64_t c0 = indices[j], c1 = indices[j + 1];
                const float v0 = values[j], v1 = values[j + 1];
                const float4 x0 = ld4(E + c0 * D + 4 * ch), x1 = ld4(E + c1 * D + 4 * ch);
                acc.x += v0 * x0.x; acc.y += v0 * x0.y; acc.z += v0 * x0.z; acc.w += v0 * x0.w;
                acc.x += v1 * x1.x; acc.y += v1 * x1.y; acc.z += v1 * x1.z; acc.w += v1 * x1.w;
            }
            if (j < e) {
                const int64_t c0 = indices[j];
                const float v0 = values[j];
                const float4 x0 = ld4(E + c0 * D + 4 * ch);
                acc.x += v0 * x0.x; acc.y += v0 * x0.y; acc.z += v0 * x0.z; acc.w += v0 * x0.w;
            }
            float* o = out + r * D + 4 * ch;
            if (MODE == 0) {
                st4(o, acc);
            } else if (MODE == 1) {
                const float4 x = ld4(X + r * D + 4 * ch);
                st4(side_out + r * D + 4 * ch, acc);
                // reference order: new = side + x*side ; new = x + new
                st4(o, make_float4(x.x + (acc.x + x.x * acc.x), x.y + (acc.y + x.y * acc.y),
                                   x.z + (acc.z + x.z * acc.z), x.w + (acc.w + x.w * acc.w)));
            } else {
                const float4 g = ld4(X + r * D + 4 * ch), s = ld4(S + r * D + 4 * ch);
                st4(o, make_float4(g.x * (1.f + s.x) + acc.x, g.y * (1.f + s.y) + acc.y, g.z * (1.f + s.z) + acc.z,
                                   g.w * (1.f + s.w) + acc.w));
            }
        }
    }
}

// tmp = g (.) (1 + x)
__global__ __launch_bounds__(kBlock) void mul_one_plus_kernel(const float* __restrict__ g, const float* __restrict__ x, int64_t n,
                                                              float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) out[e] = g[e] * (1.0f + x[e]);
}

// rows < n_overlap : lam/lap mix (reference operation order) ; other rows: copy
__global__ __launch_bounds__(kBlock) void transfer_fwd_kernel(const float* __restrict__ S, const float* __restrict__ T,
                                                              const float* __restrict__ ds, const float* __restrict__ dt,
                                                              int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t,
                                                              float* __restrict__ So, float* __restrict__ To) {
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const float s = S[e], t = T[e];
        if (r < n_overlap) {
            const float a = ds[r], b = dt[r];
            const float lap = (a * s + b * t) / ((a + b) + 1e-7f);
            const float s_lam = lam_s * s + (1.0f - lam_s) * t;
            const float t_lam = lam_t * t + (1.0f - lam_t) * s;
            So[e] = (s_lam + lap) / 2.0f;
            To[e] = (t_lam + lap) / 2.0f;
        } else {
            So[e] = s; To[e] = t;
        }
    }
}

__global__ __launch_bounds__(kBlock) void transfer_bwd_kernel(const float* __restrict__ gSo, const float* __restrict__ gTo,
                                                              const float* __restrict__ ds, const float* __restrict__ dt,
                                                              int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t,
                                                              float* __restrict__ gS, float* __restrict__ gT) {
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const float a = gSo[e], b = gTo[e];
        if (r < n_overlap) {
            const float dsv = ds[r], dtv = dt[r];
            const float dl = (dsv + dtv) + 1e-7f;
            const float ws = dsv / dl, wt = dtv / dl;
            gS[e] = 0.5f * (a * (lam_s + ws) + b * ((1.0f - lam_t) + ws));
            gT[e] = 0.5f * (a * ((1.0f - lam_s) + wt) + b * (lam_t + wt));
        } else {
            gS[e] = a; gT[e] = b;
        }
    }
}

// y = x / max(||x||_2, 1e-12), y written with leading dimension ldo (column block of the concat buffer); one wave per row
__global__ __launch_bounds__(kBlock) void l2_normalize_fwd_kernel(const float* __restrict__ x, int64_t rows, int D,
                                                                  float* __restrict__ y, int64_t ldo, float* __restrict__ norm_out) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < rows; r += TW) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) { const float v = x[r * D + c]; s += v * v; }
        s = group_sum<64>(s);
        const float nrm = sqrtf(s);
        const float den = fmaxf(nrm, 1e-12f);
        for (int c = lane; c < D; c += 64) y[r * ldo + c] = x[r * D + c] / den;
        if (lane == 0 && norm_out) norm_out[r] = nrm;
    }
}

// gx (+)= (gy - y (y . gy)) / max(norm, eps), y = x / max(norm, eps) ; gy read with leading dimension ldg
__global__ __launch_bounds__(kBlock) void l2_normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norm,
                                                                  const float* __restrict__ gy, int64_t ldg, int64_t rows, int D,
                                                                  float* __restrict__ gx, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < rows; r += TW) {
        const float den = fmaxf(norm[r], 1e-12f);
        float d = 0.f;
        for (int c = lane; c < D; c += 64) d += (x[r * D + c] / den) * gy[r * ldg + c];
        d = group_sum<64>(d);
        // below the clamp the forward is x / eps (linear): no projection term
        const float proj = norm[r] > 1e-12f ? d : 0.f;
        for (int c = lane; c < D; c += 64) {
            const float v = (gy[r * ldg + c] - (x[r * D + c] / den) * proj) / den;
            gx[r * D + c] = (accumulate ? gx[r * D + c] : 0.f) + v;
        }
    }
}

// out[r, c] (ldo) = src[r, c] (lds)     /     dst[r,c] (+)= src[r,c]
__global__ __launch_bounds__(kBlock) void copy_cols_kernel(const float* __restrict__ src, int64_t lds, int64_t rows, int D,
                                                           float* __restrict__ dst, int64_t ldo, int accumulate) {
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        const float v = src[r * lds + c];
        dst[r * ldo + c] = accumulate ? dst[r * ldo + c] + v : v;
    }
}

// out[r,c] = (sum_l cat[r, l*D + c]) / nb  (torch.mean over the stacked layer outputs, in layer order)
__global__ __launch_bounds__(kBlock) void colblock_mean_fwd_kernel(const float* __restrict__ cat, int64_t rows, int D, int nb,
                                                                   float* __restrict__ out) {
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        float s = 0.f;
        for (int l = 0; l < nb; ++l) s += cat[r * (int64_t)nb * D + (int64_t)l * D + c];
        out[e] = s / (float)nb;
    }
}

__global__ __launch_bounds__(kBlock) void colblock_mean_bwd_kernel(const float* __restrict__ gout, int64_t rows, int D, int nb,
                                                                   float* __restrict__ gcat) {
    const int64_t total = rows * D * nb, stride = (int64_t)gridDim.x * kBlock;
    const int64_t W = (int64_t)nb * D;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / W;
        const int c = (int)((e - r * W) % D);
        gcat[e] = gout[r * D + c] / (float)nb;
    }
}

// nn.Dropout(p) in training: out = keep ? x / (1-p) : 0 with a counter-based mask -- element e of call `seed` keeps iff
// splitmix64(seed + e * golden) maps to a uniform >= p.  Stateless, so the backward regenerates the same mask from the
// same seed (out = dropout(gy)) instead of storing it.  (The reference draws from torch's Philox stream; masks are not
// bit-comparable across implementations -- SURVEY App. A.1 -- only their distribution is.)
__global__ __launch_bounds__(kBlock) void dropout_kernel(const float* __restrict__ x, int64_t n, float p, uint64_t seed,
                                                         float* __restrict__ out) {
    const float scale = 1.0f / (1.0f - p);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        uint64_t z = seed + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const float u = (float)(z >> 40) * (1.0f / 16777216.0f);      // 24 uniform bits in [0,1)
        out[e] = u >= p ? x[e] * scale : 0.0f;
    }
}

// The same mask family with the seed held in DEVICE memory: a hipGraph replay bakes host scalars into its launches, so a
// host-drawn seed would repeat one mask on every replayed step; here the counter is bumped by a captured cdr_inc_i64 and each
// replay draws a fresh mask.  `salt` separates the masks of one step (layer, domain).
__global__ __launch_bounds__(kBlock) void dropout_dev_kernel(const float* __restrict__ x, int64_t n, float p,
                                                             const int64_t* __restrict__ seed_dev, uint64_t salt,
                                                             float* __restrict__ out) {
    uint64_t s0 = (uint64_t)seed_dev[0] * 0xD1342543DE82EF95ull + 0x9E3779B97F4A7C15ull;      // consecutive counters -> unrelated seeds
    s0 = (s0 ^ (s0 >> 30)) * 0xBF58476D1CE4E5B9ull;
    s0 = (s0 ^ (s0 >> 27)) * 0x94D049BB133111EBull;
    const uint64_t seed = (s0 ^ (s0 >> 31)) + salt * 0xA24BAED4963EE407ull;
    const float scale = 1.0f / (1.0f - p);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        uint64_t z = seed + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
        out[e] = u >= p ? x[e] * scale : 0.0f;
    }
}

// ---- the same two transfer kernels with nn.Dropout of the layer output folded in (bitgcf.py:134 precedes the transfer layer):
// forward: the inputs are read THROUGH the mask (s = drop(S[e]), t = drop(T[e])); backward: the outputs are written through it.
// Same mask family and the same (seed, salt, element) -> keep mapping as dropout_kernel / dropout_dev_kernel; `elem0` is the
// element offset of this row block inside the [n, D] layer output (the item block starts at nu * D).
struct drop_arg { float p; uint64_t seed; const int64_t* seed_dev; uint64_t salt_s, salt_t; int64_t elem0; };
__device__ __forceinline__ uint64_t drop_seed_of(const drop_arg& d, uint64_t salt) {
    if (!d.seed_dev) return d.seed + salt;
    uint64_t s0 = (uint64_t)d.seed_dev[0] * 0xD1342543DE82EF95ull + 0x9E3779B97F4A7C15ull;
    s0 = (s0 ^ (s0 >> 30)) * 0xBF58476D1CE4E5B9ull;
    s0 = (s0 ^ (s0 >> 27)) * 0x94D049BB133111EBull;
    return (s0 ^ (s0 >> 31)) + salt * 0xA24BAED4963EE407ull;
}
__device__ __forceinline__ float drop_factor(uint64_t seed, int64_t e, float p, float scale) {
    uint64_t z = seed + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f) >= p ? scale : 0.0f;
}

__global__ __launch_bounds__(kBlock) void transfer_drop_fwd_kernel(const float* __restrict__ S, const float* __restrict__ T,
                                                                   const float* __restrict__ ds, const float* __restrict__ dt,
                                                                   int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t,
                                                                   drop_arg dr, float* __restrict__ So, float* __restrict__ To) {
    const uint64_t seed_s = drop_seed_of(dr, dr.salt_s), seed_t = drop_seed_of(dr, dr.salt_t);
    const float scale = 1.0f / (1.0f - dr.p);
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const float fs = drop_factor(seed_s, dr.elem0 + e, dr.p, scale), ft = drop_factor(seed_t, dr.elem0 + e, dr.p, scale);
        const float s = fs != 0.f ? S[e] * scale : 0.0f, t = ft != 0.f ? T[e] * scale : 0.0f;
        if (r < n_overlap) {
            const float a = ds[r], b = dt[r];
            const float lap = (a * s + b * t) / ((a + b) + 1e-7f);
            const float s_lam = lam_s * s + (1.0f - lam_s) * t;
            const float t_lam = lam_t * t + (1.0f - lam_t) * s;
            So[e] = (s_lam + lap) / 2.0f;
            To[e] = (t_lam + lap) / 2.0f;
        } else {
            So[e] = s; To[e] = t;
        }
    }
}

__global__ __launch_bounds__(kBlock) void transfer_drop_bwd_kernel(const float* __restrict__ gSo, const float* __restrict__ gTo,
                                                                   const float* __restrict__ ds, const float* __restrict__ dt,
                                                                   int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t,
                                                                   drop_arg dr, float* __restrict__ gS, float* __restrict__ gT) {
    const uint64_t seed_s = drop_seed_of(dr, dr.salt_s), seed_t = drop_seed_of(dr, dr.salt_t);
    const float scale = 1.0f / (1.0f - dr.p);
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const float a = gSo[e], b = gTo[e];
        float gs_, gt_;
        if (r < n_overlap) {
            const float dsv = ds[r], dtv = dt[r];
            const float dl = (dsv + dtv) + 1e-7f;
            const float ws = dsv / dl, wt = dtv / dl;
            gs_ = 0.5f * (a * (lam_s + ws) + b * ((1.0f - lam_t) + ws));
            gt_ = 0.5f * (a * ((1.0f - lam_s) + wt) + b * (lam_t + wt));
        } else {
            gs_ = a; gt_ = b;
        }
        gS[e] = drop_factor(seed_s, dr.elem0 + e, dr.p, scale) != 0.f ? gs_ * scale : 0.0f;
        gT[e] = drop_factor(seed_t, dr.elem0 + e, dr.p, scale) != 0.f ? gt_ * scale : 0.0f;
    }
}

// ---- one pass per layer and direction for everything between the graph layer and the layer stack (bitgcf.py:134,137-172,190-199):
// [dropout of the layer output] -> transfer on the overlapped rows -> L2-normalised copy into the stack, users and items, both
// domains; one wave per row of the stacked [users ; items] table (row < nu: user block, else item block).  Same arithmetic, in the
// same order, as dropout_kernel / transfer_*_kernel / l2_normalize_*_kernel run one after the other -- four launches and two extra
// passes over the [n, D] buffers less per layer and direction.
struct mix_arg {
    const float* deg_su; const float* deg_tu; const float* deg_si; const float* deg_ti;
    int64_t nu, ni, OU, OI; int D; float lam_s, lam_t;
    drop_arg dr;                                   // dr.p == 0: no dropout
};
constexpr int kMixMaxJ = 8;                       // D <= 512

template <int J>
__global__ __launch_bounds__(kBlock) void mix_fwd_kernel(mix_arg a, const float* __restrict__ newS, const float* __restrict__ newT,
                                                         float* __restrict__ S2, float* __restrict__ T2, float* __restrict__ catS,
                                                         float* __restrict__ catT, int64_t ldc, float* __restrict__ nS, float* __restrict__ nT) {
    const int lane = threadIdx.x & 63, D = a.D;
    const int64_t n = a.nu + a.ni;
    const bool drop = a.dr.p > 0.f;
    const uint64_t seed_s = drop ? drop_seed_of(a.dr, a.dr.salt_s) : 0, seed_t = drop ? drop_seed_of(a.dr, a.dr.salt_t) : 0;
    const float scale = drop ? 1.0f / (1.0f - a.dr.p) : 1.0f;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < n; r += TW) {
        const bool user = r < a.nu;
        const int64_t rl = user ? r : r - a.nu;
        const bool mixrow = rl < (user ? a.OU : a.OI);
        float da = 0.f, db = 0.f;
        if (mixrow) { da = (user ? a.deg_su : a.deg_si)[rl]; db = (user ? a.deg_tu : a.deg_ti)[rl]; }
        float so[J], to[J], ss = 0.f, st = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = lane + 64 * j;
            so[j] = to[j] = 0.f;
            if (c < D) {
                const int64_t e = r * D + c;
                float s_ = newS[e], t_ = newT[e];
                if (drop) {
                    s_ = drop_factor(seed_s, e, a.dr.p, scale) != 0.f ? s_ * scale : 0.0f;
                    t_ = drop_factor(seed_t, e, a.dr.p, scale) != 0.f ? t_ * scale : 0.0f;
                }
                if (mixrow) {
                    const float lap = (da * s_ + db * t_) / ((da + db) + 1e-7f);
                    const float s_lam = a.lam_s * s_ + (1.0f - a.lam_s) * t_;
                    const float t_lam = a.lam_t * t_ + (1.0f - a.lam_t) * s_;
                    so[j] = (s_lam + lap) / 2.0f; to[j] = (t_lam + lap) / 2.0f;
                } else { so[j] = s_; to[j] = t_; }
                S2[e] = so[j]; T2[e] = to[j];
                ss += so[j] * so[j]; st += to[j] * to[j];
            }
        }
        ss = group_sum<64>(ss); st = group_sum<64>(st);
        const float ns_ = sqrtf(ss), nt_ = sqrtf(st);
        const float ds_ = fmaxf(ns_, 1e-12f), dt_ = fmaxf(nt_, 1e-12f);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = lane + 64 * j;
            if (c < D) { catS[r * ldc + c] = so[j] / ds_; catT[r * ldc + c] = to[j] / dt_; }
        }
        if (lane == 0) { nS[r] = ns_; nT[r] = nt_; }
    }
}

template <int J>
__global__ __launch_bounds__(kBlock) void mix_bwd_kernel(mix_arg a, const float* __restrict__ S2, const float* __restrict__ T2,
                                                         const float* __restrict__ nS, const float* __restrict__ nT,
                                                         const float* __restrict__ gcatS, const float* __restrict__ gcatT, int64_t ldg,
                                                         const float* __restrict__ gS_prev, const float* __restrict__ gT_prev,
                                                         float* __restrict__ gnS, float* __restrict__ gnT) {
    const int lane = threadIdx.x & 63, D = a.D;
    const int64_t n = a.nu + a.ni;
    const bool drop = a.dr.p > 0.f;
    const uint64_t seed_s = drop ? drop_seed_of(a.dr, a.dr.salt_s) : 0, seed_t = drop ? drop_seed_of(a.dr, a.dr.salt_t) : 0;
    const float scale = drop ? 1.0f / (1.0f - a.dr.p) : 1.0f;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < n; r += TW) {
        const bool user = r < a.nu;
        const int64_t rl = user ? r : r - a.nu;
        const bool mixrow = rl < (user ? a.OU : a.OI);
        const float ns_ = nS[r], nt_ = nT[r];
        const float ds_ = fmaxf(ns_, 1e-12f), dt_ = fmaxf(nt_, 1e-12f);
        float xs[J], xt[J], gys[J], gyt[J], dS = 0.f, dT = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = lane + 64 * j;
            xs[j] = xt[j] = gys[j] = gyt[j] = 0.f;
            if (c < D) {
                xs[j] = S2[r * D + c]; xt[j] = T2[r * D + c];
                gys[j] = gcatS[r * ldg + c]; gyt[j] = gcatT[r * ldg + c];
                dS += (xs[j] / ds_) * gys[j]; dT += (xt[j] / dt_) * gyt[j];
            }
        }
        dS = group_sum<64>(dS); dT = group_sum<64>(dT);
        const float pS = ns_ > 1e-12f ? dS : 0.f, pT = nt_ > 1e-12f ? dT : 0.f;
        float wsv = 0.f, wtv = 0.f;
        if (mixrow) {
            const float dsv = (user ? a.deg_su : a.deg_si)[rl], dtv = (user ? a.deg_tu : a.deg_ti)[rl];
            const float dl = (dsv + dtv) + 1e-7f;
            wsv = dsv / dl; wtv = dtv / dl;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                const int64_t e = r * D + c;
                const float ga = (gS_prev ? gS_prev[e] : 0.f) + (gys[j] - (xs[j] / ds_) * pS) / ds_;
                const float gb = (gT_prev ? gT_prev[e] : 0.f) + (gyt[j] - (xt[j] / dt_) * pT) / dt_;
                float os, ot;
                if (mixrow) {
                    os = 0.5f * (ga * (a.lam_s + wsv) + gb * ((1.0f - a.lam_t) + wsv));
                    ot = 0.5f * (ga * ((1.0f - a.lam_s) + wtv) + gb * (a.lam_t + wtv));
                } else { os = ga; ot = gb; }
                if (drop) {
                    os = drop_factor(seed_s, e, a.dr.p, scale) != 0.f ? os * scale : 0.0f;
                    ot = drop_factor(seed_t, e, a.dr.p, scale) != 0.f ? ot * scale : 0.0f;
                }
                gnS[e] = os; gnT[e] = ot;
            }
        }
    }
}

}  // namespace

#define GR_GRID(total) dim3(grid_cap(((total) + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream

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

extern "C" int cdr_spmm_csr_f32(void* stream, const int64_t* indptr, const int64_t* indices, const float* values,
                                int64_t n_rows, const float* E, int D, float* out) {
    CDR_CHECK_ARG(indptr && indices && values && E && out && n_rows > 0 && D > 0 && (D & 3) == 0);
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_cap((n_rows + kBlock / lpr - 1) / (kBlock / lpr));
    DISPATCH_LPR(lpr, spmm_csr_kernel<L, 0><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows, E,
                                                                                                 D, nullptr, nullptr, nullptr, out));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_graph_layer_fwd(void* stream, const int64_t* indptr, const int64_t* indices, const float* values,
                                   int64_t n_rows, const float* E, int D, float* side_out, float* new_out) {
    CDR_CHECK_ARG(indptr && indices && values && E && side_out && new_out && n_rows > 0 && D > 0 && (D & 3) == 0);
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_cap((n_rows + kBlock / lpr - 1) / (kBlock / lpr));
    DISPATCH_LPR(lpr, spmm_csr_kernel<L, 1><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows, E,
                                                                                                 D, E, nullptr, side_out, new_out));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_graph_layer_bwd(void* stream, const int64_t* indptr, const int64_t* indices, const float* values,
                                   int64_t n_rows, const float* E, const float* side, const float* gnew, int D, float* tmp,
                                   float* gE) {
    CDR_CHECK_ARG(indptr && indices && values && E && side && gnew && tmp && gE && n_rows > 0 && D > 0 && (D & 3) == 0);
    mul_one_plus_kernel<<<GR_GRID(n_rows * D)>>>(gnew, E, n_rows * D, tmp);
    CDR_LAUNCH_CHECK();
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_cap((n_rows + kBlock / lpr - 1) / (kBlock / lpr));
    DISPATCH_LPR(lpr, spmm_csr_kernel<L, 2><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows, tmp,
                                                                                                 D, gnew, side, nullptr, gE));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// Row-sharded forms (BASELINE configs[3]: tables and CSR sharded by destination row): the rank holds n_rows rows of the CSR with
// column indices into the ALL-GATHERED embedding buffer; the row's own value comes from the local slice.
extern "C" int cdr_graph_layer_fwd_rows(void* stream, const int64_t* indptr, const int64_t* indices, const float* values,
                                        int64_t n_rows, const float* E_gathered, const float* E_rows, int D, float* side_out,
                                        float* new_out) {
    CDR_CHECK_ARG(indptr && indices && values && E_gathered && E_rows && side_out && new_out && n_rows > 0 && D > 0 && (D & 3) == 0);
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_cap((n_rows + kBlock / lpr - 1) / (kBlock / lpr));
    DISPATCH_LPR(lpr, spmm_csr_kernel<L, 1><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows,
                                                                                                 E_gathered, D, E_rows, nullptr, side_out,
                                                                                                 new_out));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_mul_one_plus(void* stream, const float* g, const float* x, int64_t n, float* out) {
    CDR_CHECK_ARG(g && x && out && n > 0);
    mul_one_plus_kernel<<<GR_GRID(n)>>>(g, x, n, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_graph_layer_bwd_rows(void* stream, const int64_t* indptr, const int64_t* indices, const float* values,
                                        int64_t n_rows, const float* tmp_gathered, const float* gnew_rows, const float* side_rows, int D,
                                        float* gE_rows) {
    CDR_CHECK_ARG(indptr && indices && values && tmp_gathered && gnew_rows && side_rows && gE_rows && n_rows > 0 && D > 0 && (D & 3) == 0);
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_cap((n_rows + kBlock / lpr - 1) / (kBlock / lpr));
    DISPATCH_LPR(lpr, spmm_csr_kernel<L, 2><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows,
                                                                                                 tmp_gathered, D, gnew_rows, side_rows,
                                                                                                 nullptr, gE_rows));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_transfer_fwd(void* stream, const float* S, const float* T, const float* deg_s, const float* deg_t,
                                int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t, float* S_out, float* T_out) {
    CDR_CHECK_ARG(S && T && deg_s && deg_t && S_out && T_out && rows > 0 && D > 0);
    transfer_fwd_kernel<<<GR_GRID(rows * D)>>>(S, T, deg_s, deg_t, rows, D, n_overlap, lam_s, lam_t, S_out, T_out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_transfer_bwd(void* stream, const float* gS_out, const float* gT_out, const float* deg_s, const float* deg_t,
                                int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t, float* gS, float* gT) {
    CDR_CHECK_ARG(gS_out && gT_out && deg_s && deg_t && gS && gT && rows > 0 && D > 0);
    transfer_bwd_kernel<<<GR_GRID(rows * D)>>>(gS_out, gT_out, deg_s, deg_t, rows, D, n_overlap, lam_s, lam_t, gS, gT);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

#define MIX_DISPATCH(KERNEL, ...)                                                                                                   \
    switch ((D + 63) / 64) {                                                                                                      \
        case 1: KERNEL<1><<<dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream>>>(__VA_ARGS__); break;                    \
        case 2: KERNEL<2><<<dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream>>>(__VA_ARGS__); break;                    \
        case 3: KERNEL<3><<<dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream>>>(__VA_ARGS__); break;                    \
        case 4: KERNEL<4><<<dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream>>>(__VA_ARGS__); break;                    \
        default: KERNEL<8><<<dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream>>>(__VA_ARGS__); break;                   \
    }

extern "C" int cdr_bitgcf_mix_fwd(void* stream, const float* newS, const float* newT, const float* deg_su, const float* deg_tu,
                                  const float* deg_si, const float* deg_ti, int64_t nu, int64_t ni, int D, int64_t OU, int64_t OI,
                                  float lam_s, float lam_t, float p, uint64_t seed, const int64_t* seed_dev, uint64_t salt_s,
                                  uint64_t salt_t, float* S2, float* T2, float* catS_block, float* catT_block, int64_t ldc, float* nS,
                                  float* nT) {
    CDR_CHECK_ARG(newS && newT && deg_su && deg_tu && deg_si && deg_ti && S2 && T2 && catS_block && catT_block && nS && nT);
    CDR_CHECK_ARG(nu > 0 && ni > 0 && D > 0 && D <= 64 * kMixMaxJ && ldc >= D && p >= 0.f && p < 1.f);
    const mix_arg a{deg_su, deg_tu, deg_si, deg_ti, nu, ni, OU, OI, D, lam_s, lam_t, drop_arg{p, seed, seed_dev, salt_s, salt_t, 0}};
    int64_t grid = (nu + ni + 3) / 4;
    if (grid > CDR_NUM_CU * 16) grid = CDR_NUM_CU * 16;
    MIX_DISPATCH(mix_fwd_kernel, a, newS, newT, S2, T2, catS_block, catT_block, ldc, nS, nT);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bitgcf_mix_bwd(void* stream, const float* S2, const float* T2, const float* nS, const float* nT,
                                  const float* gcatS_block, const float* gcatT_block, int64_t ldg, const float* gS_prev,
                                  const float* gT_prev, const float* deg_su, const float* deg_tu, const float* deg_si,
                                  const float* deg_ti, int64_t nu, int64_t ni, int D, int64_t OU, int64_t OI, float lam_s, float lam_t,
                                  float p, uint64_t seed, const int64_t* seed_dev, uint64_t salt_s, uint64_t salt_t, float* gnS,
                                  float* gnT) {
    CDR_CHECK_ARG(S2 && T2 && nS && nT && gcatS_block && gcatT_block && deg_su && deg_tu && deg_si && deg_ti && gnS && gnT);
    CDR_CHECK_ARG(nu > 0 && ni > 0 && D > 0 && D <= 64 * kMixMaxJ && ldg >= D && p >= 0.f && p < 1.f && ((gS_prev == nullptr) == (gT_prev == nullptr)));
    const mix_arg a{deg_su, deg_tu, deg_si, deg_ti, nu, ni, OU, OI, D, lam_s, lam_t, drop_arg{p, seed, seed_dev, salt_s, salt_t, 0}};
    int64_t grid = (nu + ni + 3) / 4;
    if (grid > CDR_NUM_CU * 16) grid = CDR_NUM_CU * 16;
    MIX_DISPATCH(mix_bwd_kernel, a, S2, T2, nS, nT, gcatS_block, gcatT_block, ldg, gS_prev, gT_prev, gnS, gnT);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_transfer_drop_fwd(void* stream, const float* S, const float* T, const float* deg_s, const float* deg_t, int64_t rows,
                                     int D, int64_t n_overlap, float lam_s, float lam_t, float p, uint64_t seed, const int64_t* seed_dev,
                                     uint64_t salt_s, uint64_t salt_t, int64_t elem0, float* S_out, float* T_out) {
    CDR_CHECK_ARG(S && T && deg_s && deg_t && S_out && T_out && rows > 0 && D > 0 && p > 0.f && p < 1.f && elem0 >= 0);
    const drop_arg dr{p, seed, seed_dev, salt_s, salt_t, elem0};
    transfer_drop_fwd_kernel<<<GR_GRID(rows * D)>>>(S, T, deg_s, deg_t, rows, D, n_overlap, lam_s, lam_t, dr, S_out, T_out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_transfer_drop_bwd(void* stream, const float* gS_out, const float* gT_out, const float* deg_s, const float* deg_t,
                                     int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t, float p, uint64_t seed,
                                     const int64_t* seed_dev, uint64_t salt_s, uint64_t salt_t, int64_t elem0, float* gS, float* gT) {
    CDR_CHECK_ARG(gS_out && gT_out && deg_s && deg_t && gS && gT && rows > 0 && D > 0 && p > 0.f && p < 1.f && elem0 >= 0);
    const drop_arg dr{p, seed, seed_dev, salt_s, salt_t, elem0};
    transfer_drop_bwd_kernel<<<GR_GRID(rows * D)>>>(gS_out, gT_out, deg_s, deg_t, rows, D, n_overlap, lam_s, lam_t, dr, gS, gT);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_l2_normalize_fwd(void* stream, const float* x, int64_t rows, int D, float* y, int64_t ldo, float* norm_out) {
    CDR_CHECK_ARG(x && y && rows > 0 && D > 0 && ldo >= D);
    l2_normalize_fwd_kernel<<<dim3(grid_cap((rows + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream>>>(x, rows, D, y, ldo, norm_out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_l2_normalize_bwd(void* stream, const float* x, const float* norm, const float* gy, int64_t ldg, int64_t rows,
                                    int D, float* gx, int accumulate) {
    CDR_CHECK_ARG(x && norm && gy && gx && rows > 0 && D > 0 && ldg >= D);
    l2_normalize_bwd_kernel<<<dim3(grid_cap((rows + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream>>>(x, norm, gy, ldg, rows, D, gx,
                                                                                                     accumulate);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_copy_cols(void* stream, const float* src, int64_t lds, int64_t rows, int D, float* dst, int64_t ldo,
                             int accumulate) {
    CDR_CHECK_ARG(src && dst && rows > 0 && D > 0 && lds >= D && ldo >= D);
    copy_cols_kernel<<<GR_GRID(rows * D)>>>(src, lds, rows, D, dst, ldo, accumulate);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_colblock_mean_fwd(void* stream, const float* cat, int64_t rows, int D, int nb, float* out) {
    CDR_CHECK_ARG(cat && out && rows > 0 && D > 0 && nb > 0);
    colblock_mean_fwd_kernel<<<GR_GRID(rows * D)>>>(cat, rows, D, nb, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_colblock_mean_bwd(void* stream, const float* gout, int64_t rows, int D, int nb, float* gcat) {
    CDR_CHECK_ARG(gout && gcat && rows > 0 && D > 0 && nb > 0);
    colblock_mean_bwd_kernel<<<GR_GRID(rows * D * nb)>>>(gout, rows, D, nb, gcat);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_dropout_dev(void* stream, const float* x, int64_t n, float p, const int64_t* seed_dev, uint64_t salt, float* out) {
    CDR_CHECK_ARG(x && out && seed_dev && n > 0 && p >= 0.0f && p < 1.0f);
    dropout_dev_kernel<<<GR_GRID(n)>>>(x, n, p, seed_dev, salt, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_dropout(void* stream, const float* x, int64_t n, float p, uint64_t seed, float* out) {
    CDR_CHECK_ARG(x && out && n > 0 && p >= 0.0f && p < 1.0f);
    dropout_kernel<<<GR_GRID(n)>>>(x, n, p, seed, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
