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

// acc = sum_j values[j] * E[indices[j]] over one CSR row, for one 16-B chunk `ch` of the row, in index order.  The lane group reads
// LPR indices and values with ONE coalesced request each (the next batch's before this batch's gathers) and hands them round inside the
// group, so no gather waits for "its" index load.  At D = 64 a row is 16 lanes and a wave serves four rows: the kernel is bound by the
// instructions it issues per non-zero, not by bytes (the two-at-a-time loop this replaces spent ~19 per non-zero and ran at a fifth of
// the gather bandwidth: 88 us per SpMM at C4).  With 16-lane groups the hand-round is a DPP row broadcast (one VALU move, no LDS
// round trip), full batches run without a predicate and the row offset is a 32-bit multiply-add on a wave-uniform base: ~8 per
// non-zero.  Same additions in the same order in every variant.
template <int Q>
__device__ __forceinline__ int row16_bcast(int v) {              // lane Q of the caller's 16-lane row (DPP row_newbcast / row_share)
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + Q, 0xF, 0xF, false);
}
template <int Q>
__device__ __forceinline__ float row16_bcastf(float v) { return __int_as_float(row16_bcast<Q>(__float_as_int(v))); }

#define CDR_NZ8(Q0)                                                                                                        \
    {                                                                                                                      \
        float4 x0 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 0>(ci) * D + c4)), x1 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 1>(ci) * D + c4)); \
        float4 x2 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 2>(ci) * D + c4)), x3 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 3>(ci) * D + c4)); \
        float4 x4 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 4>(ci) * D + c4)), x5 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 5>(ci) * D + c4)); \
        float4 x6 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 6>(ci) * D + c4)), x7 = ld4(E + (size_t)(unsigned)(row16_bcast<Q0 + 7>(ci) * D + c4)); \
        CDR_NZ_FMA(x0, Q0 + 0) CDR_NZ_FMA(x1, Q0 + 1) CDR_NZ_FMA(x2, Q0 + 2) CDR_NZ_FMA(x3, Q0 + 3)                          \
        CDR_NZ_FMA(x4, Q0 + 4) CDR_NZ_FMA(x5, Q0 + 5) CDR_NZ_FMA(x6, Q0 + 6) CDR_NZ_FMA(x7, Q0 + 7)                          \
    }
#define CDR_NZ_FMA(X, Q) { const float v_ = row16_bcastf<Q>(vi); acc.x += v_ * X.x; acc.y += v_ * X.y; acc.z += v_ * X.z; acc.w += v_ * X.w; }

template <int LPR>
__device__ __forceinline__ float4 csr_row_dot(const int64_t* __restrict__ indices, const float* __restrict__ values, int64_t b, int64_t e,
                                              const float* __restrict__ E, int D, int ch, int sub, bool small) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cn = 0; float vn = 0.f;
    if (b + sub < e) { cn = (int)indices[b + sub]; vn = values[b + sub]; }
    int64_t j = b;
    if (LPR == 16 && small) {                                    // small: every element offset of E fits 32 bits
        const int c4 = 4 * ch;
        for (; j + 16 <= e; j += 16) {
            const int ci = cn; const float vi = vn;
            const int64_t jn = j + 16 + sub;
            cn = 0; vn = 0.f;
            if (jn < e) { cn = (int)indices[jn]; vn = values[jn]; }
            CDR_NZ8(0)
            CDR_NZ8(8)
        }
    }
    for (; j < e; j += LPR) {                                    // any group width; with 16 lanes: the row's last, partial batch
        const int ci = cn; const float vi = vn;
        const int64_t jn = j + LPR + sub;
        cn = 0; vn = 0.f;
        if (jn < e) { cn = (int)indices[jn]; vn = values[jn]; }
        const int cnt = (int)((e - j) < (int64_t)LPR ? (e - j) : (int64_t)LPR);
        for (int q0 = 0; q0 < cnt; q0 += 8) {
            float4 x[8]; float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = __shfl(ci, q0 + q, LPR);
                v[q] = __shfl(vi, q0 + q, LPR);
                x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q0 + q < cnt) x[q] = ld4(E + (int64_t)c * D + 4 * ch);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q0 + q < cnt) { acc.x += v[q] * x[q].x; acc.y += v[q] * x[q].y; acc.z += v[q] * x[q].z; acc.w += v[q] * x[q].w; }
            }
        }
    }
    return acc;
}
#undef CDR_NZ8
#undef CDR_NZ_FMA

// the same for the flagged backward: the term of column c is values[j] * (G[c] (.) (1 + E[c])) where flags[c] != 0 and an exact zero
// elsewhere (skipped).  Every lane probes the flag of its own index (LPR probes in flight per group, one batch ahead of the gathers,
// the indices two batches ahead); the group then walks the set bits of its flag mask in ascending order -- the non-zeros that
// matter, a fifth of them at C4 -- instead of testing every position.
template <int LPR>
__device__ __forceinline__ float4 csr_row_dot_flagged(const int64_t* __restrict__ indices, const float* __restrict__ values, int64_t b,
                                                      int64_t e, const float* __restrict__ G, const float* __restrict__ E,
                                                      const uint32_t* flags, int D, int ch, int sub) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c1 = 0, c2 = 0, f1 = 0; float v1 = 0.f, v2 = 0.f;
    if (b + sub < e) { c1 = (int)indices[b + sub]; v1 = values[b + sub]; }
    if (b + LPR + sub < e) { c2 = (int)indices[b + LPR + sub]; v2 = values[b + LPR + sub]; }
    if (b + sub < e) f1 = (int)((flags[c1 >> 5] >> (c1 & 31)) & 1u);
    const int gbase = (int)(threadIdx.x & 63) - sub;             // first lane of this group inside the wave
    for (int64_t j = b; j < e; j += LPR) {
        const int ci = c1, fi = f1; const float vi = v1;
        c1 = c2; v1 = v2;
        f1 = (j + LPR + sub < e) ? (int)((flags[c1 >> 5] >> (c1 & 31)) & 1u) : 0;
        const int64_t j2 = j + 2 * LPR + sub;
        c2 = 0; v2 = 0.f;
        if (j2 < e) { c2 = (int)indices[j2]; v2 = values[j2]; }
        unsigned long long m = (__ballot(fi != 0) >> gbase) & ((LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull));
        while (m) {                                               // group-uniform: every lane of the group holds the same mask
            const int q = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const int c = __shfl(ci, q, LPR);
            const float v = __shfl(vi, q, LPR);
            const float4 g = ld4(G + (int64_t)c * D + 4 * ch), x = ld4(E + (int64_t)c * D + 4 * ch);
            const float4 t = make_float4(g.x * (1.0f + x.x), g.y * (1.0f + x.y), g.z * (1.0f + x.z), g.w * (1.0f + x.w));
            acc.x += v * t.x; acc.y += v * t.y; acc.z += v * t.z; acc.w += v * t.w;
        }
    }
    return acc;
}

// MODE 0: out = acc ; MODE 1: side_out = acc, out = x + acc + x*acc (x = X[r]) ; MODE 2: out = G[r]*(1 + S[r]) + acc
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                          const float* __restrict__ values, int64_t n_rows,
                                                          const float* __restrict__ E, int D, const float* __restrict__ X,
                                                          const float* __restrict__ S, float* __restrict__ side_out,
                                                          float* __restrict__ out, bool small) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    for (int64_t r = gg; r < n_rows; r += TG) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        for (int ch0 = 0; ch0 < D4; ch0 += LPR) {
            // every lane of the group takes part in handing the indices round, also the ones past the row's last chunk (D / 4 not a
            // power of two): they redo chunk 0 and store nothing
            const bool live = ch0 + sub < D4;
            const int ch = live ? ch0 + sub : 0;
            const float4 acc = csr_row_dot<LPR>(indices, values, b, e, E, D, ch, sub, small);
            if (!live) continue;
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

// The LAST propagation layer only feeds the rows the loss gathers (a few thousand of n): `flags[r] != 0` marks them.
//   forward  (graph layer): only flagged rows are computed; the others' side / new rows are left unwritten and never read
//   backward (graph layer): gnew is zero outside the flagged rows, so a non-zero's gather (gnew[c], E[c]: tmp is formed on the fly)
//                           happens only where flags[c] != 0 -- every skipped term is an exact zero of the unmasked sum, the
//                           remaining ones are added in the same order: bit-identical gE, 4D + 12 -> ~13 B per non-zero
template <int LPR>
__global__ __launch_bounds__(kBlock) void spmm_rows_fwd_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                               const float* __restrict__ values, int64_t n_rows,
                                                               const float* __restrict__ E, int D, const int32_t* __restrict__ rowlist,
                                                               const int32_t* __restrict__ rowcount,
                                                               float* __restrict__ side_out, float* __restrict__ out, bool small) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const int64_t nsel = rowcount[0];                            // the flagged rows, compacted (any order: a row's result is its own)
    for (int64_t i = gg; i < nsel; i += TG) {
        const int64_t r = rowlist[i];
        const int64_t b = indptr[r], e = indptr[r + 1];
        for (int ch0 = 0; ch0 < D4; ch0 += LPR) {
            const bool live = ch0 + sub < D4;                    // (see spmm_csr_kernel)
            const int ch = live ? ch0 + sub : 0;
            const float4 acc = csr_row_dot<LPR>(indices, values, b, e, E, D, ch, sub, small);       // same order as spmm_csr_kernel
            if (!live) continue;
            const float4 x = ld4(E + r * D + 4 * ch);
            st4(side_out + r * D + 4 * ch, acc);
            st4(out + r * D + 4 * ch, make_float4(x.x + (acc.x + x.x * acc.x), x.y + (acc.y + x.y * acc.y),
                                                  x.z + (acc.z + x.z * acc.z), x.w + (acc.w + x.w * acc.w)));
        }
    }
}

// (Tried: the flagged rows' products g (.) (1 + E) formed once per row by a pass over the flag list, one gather per flagged non-zero
//  instead of two -- 27.3 -> 25.2 and 49.2 -> 45.5 us at C4 for 5.4 + 5.7 us of extra launches: the kernel is bound by its index walk and
//  flag probes, not by the gathers.  Not kept.)
template <int LPR>
__global__ __launch_bounds__(kBlock) void spmm_flagged_bwd_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                                  const float* __restrict__ values, int64_t n_rows,
                                                                  const float* __restrict__ E, int D, const float* __restrict__ G,
                                                                  const float* __restrict__ S, const uint32_t* __restrict__ bitmap,
                                                                  int lds_words, float* __restrict__ out) {
    extern __shared__ uint32_t bm_lds[];
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    // one probe per non-zero: from a copy of the bitmap in LDS (n / 8 bytes: 10 KB at C4) instead of one L2 request each -- as many
    // requests as the gathers they were meant to save
    for (int w = threadIdx.x; w < lds_words; w += kBlock) bm_lds[w] = bitmap[w];
    if (lds_words) __syncthreads();
    const uint32_t* flags = lds_words ? bm_lds : bitmap;
    for (int64_t r = gg; r < n_rows; r += TG) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        const bool own = (flags[r >> 5] >> (r & 31)) & 1u;
        for (int ch0 = 0; ch0 < D4; ch0 += LPR) {
            const bool live = ch0 + sub < D4;                    // (see spmm_csr_kernel)
            const int ch = live ? ch0 + sub : 0;
            const float4 acc = csr_row_dot_flagged<LPR>(indices, values, b, e, G, E, flags, D, ch, sub);
            if (!live) continue;
            float4 o = acc;
            if (own) {
                const float4 g = ld4(G + r * D + 4 * ch), s_ = ld4(S + r * D + 4 * ch);
                o = make_float4(g.x * (1.f + s_.x) + acc.x, g.y * (1.f + s_.y) + acc.y, g.z * (1.f + s_.z) + acc.z,
                                g.w * (1.f + s_.w) + acc.w);
            }
            st4(out + r * D + 4 * ch, o);
        }
    }
}

// Marks the rows off[i] + ids[i][k]: flags[r] = 1 (bytes, for the row-wise kernels), bit r of the bitmap (for the SpMM's column probes)
// and, for the thread that set the bit first, an entry of the compacted row list.  work = [flags | bitmap | count | list], see
// cdr_row_flags_layout; the first three are zero on entry.
struct flag_lists { const int64_t* ids[8]; int64_t n[8]; int64_t off[8]; int count; };

__global__ __launch_bounds__(kBlock) void row_flags_kernel(flag_lists fl, int64_t rows, uint8_t* __restrict__ flags,
                                                           uint32_t* __restrict__ bitmap, int32_t* __restrict__ count,
                                                           int32_t* __restrict__ list) {
    const int li = blockIdx.y;
    const int64_t n = fl.n[li];
    const int64_t* ids = fl.ids[li];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = fl.off[li] + ids[i];
        if (r < 0 || r >= rows) continue;
        const uint32_t bit = 1u << (r & 31);
        const uint32_t old = atomicOr(&bitmap[r >> 5], bit);
        if (!(old & bit)) {
            flags[r] = 1;
            list[atomicAdd(count, 1)] = (int32_t)r;
        }
    }
}

// tmp = g (.) (1 + x)
__global__ __launch_bounds__(kBlock) void mul_one_plus_kernel(const float* __restrict__ g, const float* __restrict__ x, int64_t n,
                                                              float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if (!(n & 3) && !(((uintptr_t)g | (uintptr_t)x | (uintptr_t)out) & 15)) {                        // 16-B requests
        const int64_t n4 = n >> 2;
        for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n4; e += stride) {
            const float4 a = ld4(g + 4 * e), b = ld4(x + 4 * e);
            st4(out + 4 * e, make_float4(a.x * (1.0f + b.x), a.y * (1.0f + b.y), a.z * (1.0f + b.z), a.w * (1.0f + b.w)));
        }
        return;
    }
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

// The ego layer of both domains in ONE launch: S = [su ; si], T = [tu ; ti] (the first graph layer's input) and the same rows as
// block 0 of the two layer stacks (six copy_cols launches before).  float4 per thread; D % 4 == 0.
__global__ __launch_bounds__(kBlock) void bitgcf_stack_kernel(const float* __restrict__ su, const float* __restrict__ si,
                                                              const float* __restrict__ tu, const float* __restrict__ ti, int64_t nu,
                                                              int64_t ni, int D, float* __restrict__ S, float* __restrict__ T,
                                                              float* __restrict__ catS, float* __restrict__ catT, int64_t ldc) {
    const int D4 = D >> 2;
    const int64_t total = (nu + ni) * D4, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D4;
        const int c = 4 * (int)(e - r * D4);
        const bool user = r < nu;
        const int64_t o = (user ? r : r - nu) * D + c;
        const float4 a = ld4((user ? su : si) + o), b = ld4((user ? tu : ti) + o);
        st4(S + r * D + c, a); st4(T + r * D + c, b);
        st4(catS + r * ldc + c, a); st4(catT + r * ldc + c, b);
    }
}

// gS += block 0 of gcatS, gT += block 0 of gcatT (the ego rows' share of the stack's gradient), one launch
__global__ __launch_bounds__(kBlock) void bitgcf_unstack_bwd_kernel(const float* __restrict__ gcatS, const float* __restrict__ gcatT,
                                                                    int64_t ldg, int64_t n, int D, float* __restrict__ gS,
                                                                    float* __restrict__ gT) {
    const int D4 = D >> 2;
    const int64_t total = n * D4, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D4;
        const int c = 4 * (int)(e - r * D4);
        const float4 a = ld4(gcatS + r * ldg + c), b = ld4(gcatT + r * ldg + c);
        float4 x = ld4(gS + r * D + c), y = ld4(gT + r * D + c);
        x.x += a.x; x.y += a.y; x.z += a.z; x.w += a.w;
        y.x += b.x; y.y += b.y; y.z += b.z; y.w += b.w;
        st4(gS + r * D + c, x); st4(gT + r * D + c, y);
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
                                                         float* __restrict__ catT, int64_t ldc, float* __restrict__ nS, float* __restrict__ nT,
                                                         const uint8_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63, D = a.D;
    const int64_t n = a.nu + a.ni;
    const bool drop = a.dr.p > 0.f;
    const uint64_t seed_s = drop ? drop_seed_of(a.dr, a.dr.salt_s) : 0, seed_t = drop ? drop_seed_of(a.dr, a.dr.salt_t) : 0;
    const float scale = drop ? 1.0f / (1.0f - a.dr.p) : 1.0f;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < n; r += TW) {
        if (flags && !flags[r]) {                  // a row the loss never reads (last layer): its block of the stack is zero, nothing else exists
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = lane + 64 * j;
                if (c < D) { catS[r * ldc + c] = 0.f; catT[r * ldc + c] = 0.f; }
            }
            continue;
        }
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
                                                         float* __restrict__ gnS, float* __restrict__ gnT, const uint8_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63, D = a.D;
    const int64_t n = a.nu + a.ni;
    const bool drop = a.dr.p > 0.f;
    const uint64_t seed_s = drop ? drop_seed_of(a.dr, a.dr.salt_s) : 0, seed_t = drop ? drop_seed_of(a.dr, a.dr.salt_t) : 0;
    const float scale = drop ? 1.0f / (1.0f - a.dr.p) : 1.0f;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < n; r += TW) {
        if (flags && !flags[r]) {                  // (only with gS_prev == nullptr) no gradient reaches this row: exact zeros
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = lane + 64 * j;
                if (c < D) { gnS[r * D + c] = 0.f; gnT[r * D + c] = 0.f; }
            }
            continue;
        }
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

// The same two kernels with one lane group of LPR = D / 4 lanes per row (a float4 per lane; D % 4 == 0, D <= 256): at D = 64 a wave
// moves four rows per request (1 KB) where the one-float-per-lane form above moves one (256 B) -- the C4 step spent 120 us in these
// elementwise passes at 3.5 TB/s.  Same arithmetic per element, same counter-based dropout draw per element index; the squared norms are
// summed per lane (x, y, z, w) and then across the group, i.e. in a different order than above (1 ulp).
template <int LPR>
__global__ __launch_bounds__(kBlock) void mix_fwd_vec_kernel(mix_arg a, const float* __restrict__ newS, const float* __restrict__ newT,
                                                             float* __restrict__ S2, float* __restrict__ T2, float* __restrict__ catS,
                                                             float* __restrict__ catT, int64_t ldc, float* __restrict__ nS, float* __restrict__ nT,
                                                             const uint8_t* __restrict__ flags) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR, D = a.D;
    const bool live = sub < (D >> 2);
    const int64_t n = a.nu + a.ni;
    const bool drop = a.dr.p > 0.f;
    const uint64_t seed_s = drop ? drop_seed_of(a.dr, a.dr.salt_s) : 0, seed_t = drop ? drop_seed_of(a.dr, a.dr.salt_t) : 0;
    const float scale = drop ? 1.0f / (1.0f - a.dr.p) : 1.0f;
    const int64_t g0 = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR, TG = (int64_t)gridDim.x * GPB;
    const int64_t rounds = (n + TG - 1) / TG;                     // every group runs the same trips (the group sums are wave-wide instructions)
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t r = g0 + it * TG;
        const bool row = r < n;
        const bool skip = row && flags && !flags[r];             // a row the loss never reads (last layer): its block of the stack is zero
        const bool work = row && !skip && live;
        float so[4] = {0.f, 0.f, 0.f, 0.f}, to[4] = {0.f, 0.f, 0.f, 0.f};
        if (work) {
            const bool user = r < a.nu;
            const int64_t rl = user ? r : r - a.nu;
            const bool mixrow = rl < (user ? a.OU : a.OI);
            float da = 0.f, db = 0.f;
            if (mixrow) { da = (user ? a.deg_su : a.deg_si)[rl]; db = (user ? a.deg_tu : a.deg_ti)[rl]; }
            const int64_t e0 = r * D + 4 * sub;
            const float4 sv = ld4(newS + e0), tv = ld4(newT + e0);
            const float s4[4] = {sv.x, sv.y, sv.z, sv.w}, t4[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s_ = s4[k], t_ = t4[k];
                if (drop) {
                    s_ = drop_factor(seed_s, e0 + k, a.dr.p, scale) != 0.f ? s_ * scale : 0.0f;
                    t_ = drop_factor(seed_t, e0 + k, a.dr.p, scale) != 0.f ? t_ * scale : 0.0f;
                }
                if (mixrow) {
                    const float lap = (da * s_ + db * t_) / ((da + db) + 1e-7f);
                    const float s_lam = a.lam_s * s_ + (1.0f - a.lam_s) * t_;
                    const float t_lam = a.lam_t * t_ + (1.0f - a.lam_t) * s_;
                    so[k] = (s_lam + lap) / 2.0f; to[k] = (t_lam + lap) / 2.0f;
                } else { so[k] = s_; to[k] = t_; }
            }
            st4(S2 + e0, make_float4(so[0], so[1], so[2], so[3]));
            st4(T2 + e0, make_float4(to[0], to[1], to[2], to[3]));
        }
        float ss = (so[0] * so[0] + so[1] * so[1]) + (so[2] * so[2] + so[3] * so[3]);
        float st = (to[0] * to[0] + to[1] * to[1]) + (to[2] * to[2] + to[3] * to[3]);
        ss = group_sum<LPR>(ss); st = group_sum<LPR>(st);
        const float ns_ = sqrtf(ss), nt_ = sqrtf(st);
        const float ds_ = fmaxf(ns_, 1e-12f), dt_ = fmaxf(nt_, 1e-12f);
        if (row && live) {
            st4(catS + r * ldc + 4 * sub, make_float4(so[0] / ds_, so[1] / ds_, so[2] / ds_, so[3] / ds_));       // (skipped rows: so = 0 -> zeros)
            st4(catT + r * ldc + 4 * sub, make_float4(to[0] / dt_, to[1] / dt_, to[2] / dt_, to[3] / dt_));
        }
        if (row && !skip && sub == 0) { nS[r] = ns_; nT[r] = nt_; }
    }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void mix_bwd_vec_kernel(mix_arg a, const float* __restrict__ S2, const float* __restrict__ T2,
                                                             const float* __restrict__ nS, const float* __restrict__ nT,
                                                             const float* __restrict__ gcatS, const float* __restrict__ gcatT, int64_t ldg,
                                                             const float* __restrict__ gS_prev, const float* __restrict__ gT_prev,
                                                             float* __restrict__ gnS, float* __restrict__ gnT, const uint8_t* __restrict__ flags) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR, D = a.D;
    const bool live = sub < (D >> 2);
    const int64_t n = a.nu + a.ni;
    const bool drop = a.dr.p > 0.f;
    const uint64_t seed_s = drop ? drop_seed_of(a.dr, a.dr.salt_s) : 0, seed_t = drop ? drop_seed_of(a.dr, a.dr.salt_t) : 0;
    const float scale = drop ? 1.0f / (1.0f - a.dr.p) : 1.0f;
    const int64_t g0 = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR, TG = (int64_t)gridDim.x * GPB;
    const int64_t rounds = (n + TG - 1) / TG;
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t r = g0 + it * TG;
        const bool row = r < n;
        const bool skip = row && flags && !flags[r];             // (only with gS_prev == nullptr) no gradient reaches this row: exact zeros
        const bool work = row && !skip && live;
        float xs[4] = {0.f, 0.f, 0.f, 0.f}, xt[4] = {0.f, 0.f, 0.f, 0.f}, gys[4] = {0.f, 0.f, 0.f, 0.f}, gyt[4] = {0.f, 0.f, 0.f, 0.f};
        float ns_ = 0.f, nt_ = 0.f;
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
        if (work) {
            ns_ = nS[r]; nt_ = nT[r];
            const int64_t e0 = r * D + 4 * sub;
            const float4 a4 = ld4(S2 + e0), b4 = ld4(T2 + e0), c4 = ld4(gcatS + r * ldg + 4 * sub), d4 = ld4(gcatT + r * ldg + 4 * sub);
            if (gS_prev) { pa = ld4(gS_prev + e0); pb = ld4(gT_prev + e0); }
            xs[0] = a4.x; xs[1] = a4.y; xs[2] = a4.z; xs[3] = a4.w; xt[0] = b4.x; xt[1] = b4.y; xt[2] = b4.z; xt[3] = b4.w;
            gys[0] = c4.x; gys[1] = c4.y; gys[2] = c4.z; gys[3] = c4.w; gyt[0] = d4.x; gyt[1] = d4.y; gyt[2] = d4.z; gyt[3] = d4.w;
        }
        const float ds_ = fmaxf(ns_, 1e-12f), dt_ = fmaxf(nt_, 1e-12f);
        float dS = 0.f, dT = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dS += (xs[k] / ds_) * gys[k]; dT += (xt[k] / dt_) * gyt[k]; }
        dS = group_sum<LPR>(dS); dT = group_sum<LPR>(dT);
        if (!(row && live)) continue;                            // (behind the group sums: nothing wave-wide follows)
        const int64_t e0 = r * D + 4 * sub;
        if (skip) {
            st4(gnS + e0, make_float4(0.f, 0.f, 0.f, 0.f)); st4(gnT + e0, make_float4(0.f, 0.f, 0.f, 0.f));
            continue;
        }
        const float pS = ns_ > 1e-12f ? dS : 0.f, pT = nt_ > 1e-12f ? dT : 0.f;
        const bool user = r < a.nu;
        const int64_t rl = user ? r : r - a.nu;
        const bool mixrow = rl < (user ? a.OU : a.OI);
        float wsv = 0.f, wtv = 0.f;
        if (mixrow) {
            const float dsv = (user ? a.deg_su : a.deg_si)[rl], dtv = (user ? a.deg_tu : a.deg_ti)[rl];
            const float dl = (dsv + dtv) + 1e-7f;
            wsv = dsv / dl; wtv = dtv / dl;
        }
        const float pas[4] = {pa.x, pa.y, pa.z, pa.w}, pbs[4] = {pb.x, pb.y, pb.z, pb.w};
        float os[4], ot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ga = pas[k] + (gys[k] - (xs[k] / ds_) * pS) / ds_;
            const float gb = pbs[k] + (gyt[k] - (xt[k] / dt_) * pT) / dt_;
            if (mixrow) {
                os[k] = 0.5f * (ga * (a.lam_s + wsv) + gb * ((1.0f - a.lam_t) + wsv));
                ot[k] = 0.5f * (ga * ((1.0f - a.lam_s) + wtv) + gb * (a.lam_t + wtv));
            } else { os[k] = ga; ot[k] = gb; }
            if (drop) {
                os[k] = drop_factor(seed_s, e0 + k, a.dr.p, scale) != 0.f ? os[k] * scale : 0.0f;
                ot[k] = drop_factor(seed_t, e0 + k, a.dr.p, scale) != 0.f ? ot[k] * scale : 0.0f;
            }
        }
        st4(gnS + e0, make_float4(os[0], os[1], os[2], os[3]));
        st4(gnT + e0, make_float4(ot[0], ot[1], ot[2], ot[3]));
    }
}

// the row-flag work buffer (cdr_row_flags): byte flags [rows], bit map [ceil(rows / 32)] words, the list's length, the list [rows]
struct row_flag_views { uint8_t* flags; uint32_t* bitmap; int32_t* count; int32_t* list; size_t zero_bytes, bytes; };
inline row_flag_views flag_views(const uint8_t* work, int64_t rows) {
    const size_t o_bm = ((size_t)rows + 15) & ~(size_t)15;
    const size_t o_cnt = o_bm + (((size_t)(rows + 31) / 32 * 4 + 15) & ~(size_t)15);
    const size_t o_list = o_cnt + 16;
    uint8_t* w = const_cast<uint8_t*>(work);
    return row_flag_views{w, (uint32_t*)(w + o_bm), (int32_t*)(w + o_cnt), (int32_t*)(w + o_list), o_list, o_list + (size_t)rows * 4};
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
                                                                                                 D, nullptr, nullptr, nullptr, out, false));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_graph_layer_fwd(void* stream, const int64_t* indptr, const int64_t* indices, const float* values,
                                   int64_t n_rows, const float* E, int D, float* side_out, float* new_out, const uint8_t* row_flags) {
    CDR_CHECK_ARG(indptr && indices && values && E && side_out && new_out && n_rows > 0 && D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(n_rows < ((int64_t)1 << 31));                 // column indices travel between lanes as 32-bit values
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_cap((n_rows + kBlock / lpr - 1) / (kBlock / lpr));
    const bool small = n_rows * (int64_t)D < ((int64_t)1 << 30);        // square adjacency: every column index < n_rows
    if (row_flags) {
        row_flag_views v = flag_views(row_flags, n_rows);
        DISPATCH_LPR(lpr, spmm_rows_fwd_kernel<L><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows, E, D,
                                                                                                       v.list, v.count, side_out, new_out,
                                                                                                       small));
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    DISPATCH_LPR(lpr, spmm_csr_kernel<L, 1><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows, E,
                                                                                                 D, E, nullptr, side_out, new_out, small));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_graph_layer_bwd(void* stream, const int64_t* indptr, const int64_t* indices, const float* values,
                                   int64_t n_rows, const float* E, const float* side, const float* gnew, int D, float* tmp,
                                   float* gE, const uint8_t* row_flags) {
    CDR_CHECK_ARG(indptr && indices && values && E && side && gnew && (tmp || row_flags) && gE && n_rows > 0 && D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(n_rows < ((int64_t)1 << 31));
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_cap((n_rows + kBlock / lpr - 1) / (kBlock / lpr));
    const bool small = n_rows * (int64_t)D < ((int64_t)1 << 30);
    if (row_flags) {                                           // gnew is zero outside the flagged rows (caller's contract)
        row_flag_views v = flag_views(row_flags, n_rows);
        const int64_t words = (n_rows + 31) / 32;
        const int lds_words = words * 4 <= 20 * 1024 ? (int)words : 0;          // <= 20 KB per block: 8 blocks per CU keep their LDS
        DISPATCH_LPR(lpr, spmm_flagged_bwd_kernel<L><<<dim3(grid), dim3(kBlock), (size_t)lds_words * 4, (hipStream_t)stream>>>(
            indptr, indices, values, n_rows, E, D, gnew, side, v.bitmap, lds_words, gE));
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    mul_one_plus_kernel<<<GR_GRID(n_rows * D)>>>(gnew, E, n_rows * D, tmp);
    CDR_LAUNCH_CHECK();
    DISPATCH_LPR(lpr, spmm_csr_kernel<L, 2><<<dim3(grid), dim3(kBlock), 0, (hipStream_t)stream>>>(indptr, indices, values, n_rows, tmp,
                                                                                                 D, gnew, side, nullptr, gE, small));
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
                                                                                                 new_out, false));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_row_flags_layout(int64_t rows, size_t* bytes) {
    CDR_CHECK_ARG(rows > 0 && bytes);
    *bytes = flag_views(nullptr, rows).bytes;
    return CDR_OK;
}

extern "C" int cdr_row_flags(void* stream, int n_lists, const int64_t* const* ids, const int64_t* counts, const int64_t* offsets,
                             int64_t rows, uint8_t* work, size_t work_bytes) {
    CDR_CHECK_ARG(n_lists >= 0 && n_lists <= 8 && rows > 0 && rows < ((int64_t)1 << 31) && work && (n_lists == 0 || (ids && counts && offsets)));
    row_flag_views v = flag_views(work, rows);
    CDR_CHECK_ARG(work_bytes >= v.bytes && ((uintptr_t)work & 15) == 0);
    // (a kernel, not hipMemsetAsync: see cdr_zero_u32)
    CDR_HIP(cdr_zero_u32(work, (int64_t)(v.zero_bytes / 4), (hipStream_t)stream));
    flag_lists fl{};
    int64_t nmax = 0;
    for (int i = 0; i < n_lists; ++i) {
        CDR_CHECK_ARG(counts[i] <= 0 || ids[i]);
        fl.ids[i] = ids[i]; fl.n[i] = counts[i] > 0 ? counts[i] : 0; fl.off[i] = offsets[i];
        if (fl.n[i] > nmax) nmax = fl.n[i];
    }
    fl.count = n_lists;
    if (nmax > 0) {
        row_flags_kernel<<<dim3(grid_cap((nmax + kBlock - 1) / kBlock), n_lists), dim3(kBlock), 0, (hipStream_t)stream>>>(
            fl, rows, v.flags, v.bitmap, v.count, v.list);
        CDR_LAUNCH_CHECK();
    }
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
                                                                                                 nullptr, gE_rows, false));
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

// the float4 form: D a multiple of 4 up to 256, every operand 16-byte aligned with a row stride that keeps it so (NULL operands pass)
static inline bool mix_vec_ok(int D, int64_t ld, const float* p0, const float* p1, const float* p2, const float* p3, const float* p4,
                              const float* p5) {
    if (D <= 0 || (D & 3) || D > 256 || (ld & 3)) return false;
    const uintptr_t m = (uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3 | (uintptr_t)p4 | (uintptr_t)p5;
    return (m & 15) == 0;
}
static inline int64_t mix_vec_grid(int64_t rows, int lpr) {
    const int gpb = kBlock / lpr;
    int64_t grid = (rows + gpb - 1) / gpb;
    if (grid > CDR_NUM_CU * 16) grid = CDR_NUM_CU * 16;
    return grid < 1 ? 1 : grid;
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
                                  float* nT, const uint8_t* row_flags) {
    CDR_CHECK_ARG(newS && newT && deg_su && deg_tu && deg_si && deg_ti && S2 && T2 && catS_block && catT_block && nS && nT);
    CDR_CHECK_ARG(nu > 0 && ni > 0 && D > 0 && D <= 64 * kMixMaxJ && ldc >= D && p >= 0.f && p < 1.f);
    const mix_arg a{deg_su, deg_tu, deg_si, deg_ti, nu, ni, OU, OI, D, lam_s, lam_t, drop_arg{p, seed, seed_dev, salt_s, salt_t, 0}};
    if (mix_vec_ok(D, ldc, newS, newT, S2, T2, catS_block, catT_block)) {
        const int lpr = cdr_lpr_for(D);
        const int64_t grid = mix_vec_grid(nu + ni, lpr);
        DISPATCH_LPR(lpr, mix_fwd_vec_kernel<L><<<dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream>>>(a, newS, newT, S2, T2, catS_block, catT_block, ldc, nS, nT, row_flags));
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    int64_t grid = (nu + ni + 3) / 4;
    if (grid > CDR_NUM_CU * 16) grid = CDR_NUM_CU * 16;
    MIX_DISPATCH(mix_fwd_kernel, a, newS, newT, S2, T2, catS_block, catT_block, ldc, nS, nT, row_flags);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bitgcf_mix_bwd(void* stream, const float* S2, const float* T2, const float* nS, const float* nT,
                                  const float* gcatS_block, const float* gcatT_block, int64_t ldg, const float* gS_prev,
                                  const float* gT_prev, const float* deg_su, const float* deg_tu, const float* deg_si,
                                  const float* deg_ti, int64_t nu, int64_t ni, int D, int64_t OU, int64_t OI, float lam_s, float lam_t,
                                  float p, uint64_t seed, const int64_t* seed_dev, uint64_t salt_s, uint64_t salt_t, float* gnS,
                                  float* gnT, const uint8_t* row_flags) {
    CDR_CHECK_ARG(S2 && T2 && nS && nT && gcatS_block && gcatT_block && deg_su && deg_tu && deg_si && deg_ti && gnS && gnT);
    CDR_CHECK_ARG(nu > 0 && ni > 0 && D > 0 && D <= 64 * kMixMaxJ && ldg >= D && p >= 0.f && p < 1.f && ((gS_prev == nullptr) == (gT_prev == nullptr)));
    const mix_arg a{deg_su, deg_tu, deg_si, deg_ti, nu, ni, OU, OI, D, lam_s, lam_t, drop_arg{p, seed, seed_dev, salt_s, salt_t, 0}};
    CDR_CHECK_ARG(!row_flags || !gS_prev);                    // flags mean "no gradient reaches the other rows": last layer only
    if (mix_vec_ok(D, ldg, S2, T2, gcatS_block, gcatT_block, gnS, gnT) && mix_vec_ok(D, ldg, gS_prev, gT_prev, gnS, gnT, gnS, gnT)) {
        const int lpr = cdr_lpr_for(D);
        const int64_t grid = mix_vec_grid(nu + ni, lpr);
        DISPATCH_LPR(lpr, mix_bwd_vec_kernel<L><<<dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream>>>(a, S2, T2, nS, nT, gcatS_block, gcatT_block, ldg, gS_prev, gT_prev, gnS, gnT, row_flags));
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    int64_t grid = (nu + ni + 3) / 4;
    if (grid > CDR_NUM_CU * 16) grid = CDR_NUM_CU * 16;
    MIX_DISPATCH(mix_bwd_kernel, a, S2, T2, nS, nT, gcatS_block, gcatT_block, ldg, gS_prev, gT_prev, gnS, gnT, row_flags);
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

extern "C" int cdr_bitgcf_stack(void* stream, const float* su, const float* si, const float* tu, const float* ti, int64_t nu, int64_t ni,
                                int D, float* S, float* T, float* catS, float* catT, int64_t ldc) {
    CDR_CHECK_ARG(su && si && tu && ti && S && T && catS && catT && nu > 0 && ni > 0 && D > 0 && (D & 3) == 0 && ldc >= D && (ldc & 3) == 0);
    bitgcf_stack_kernel<<<GR_GRID((nu + ni) * (D / 4))>>>(su, si, tu, ti, nu, ni, D, S, T, catS, catT, ldc);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bitgcf_unstack_bwd(void* stream, const float* gcatS, const float* gcatT, int64_t ldg, int64_t n, int D, float* gS,
                                      float* gT) {
    CDR_CHECK_ARG(gcatS && gcatT && gS && gT && n > 0 && D > 0 && (D & 3) == 0 && ldg >= D && (ldg & 3) == 0);
    bitgcf_unstack_bwd_kernel<<<GR_GRID(n * (D / 4))>>>(gcatS, gcatT, ldg, n, D, gS, gT);
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
