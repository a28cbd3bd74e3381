// fp32 MFMA contraction with fused epilogue (K6/K8/K9 of SURVEY.md 2.2).
//
// C[M,N] = epi(op(A) x op(B)) on v_mfma_f32_32x32x2_f32 (exact fp32, one rounding per product: the same numerics class
// as the reference's sgemm).  Both operand tiles live in LDS K-contiguous, rows padded to 36 dwords: that stride makes
// the 16-B fragment reads (ds_read_b128) bank-conflict-free for every 16-lane service group (36*i mod 64 is a distinct
// multiple of 4 for 16 rows that differ mod 16) and keeps rows 16-B aligned for ds_write_b128.  The contraction index
// is permuted inside each 8-wide K step so that ONE float4 per lane feeds FOUR MFMAs: lane (i = l&31, h = l>>5) reads
// k = 8s+4h..8s+4h+3 of row i, and MFMA c consumes component c of both operands, i.e. the k pair {8s+c, 8s+4+c}.
//
// Block = 256 threads = 4 waves.  Tile configurations:
//   <BM=32 ,BN=128>: 1x4 waves, one 32x32 accumulator each     -- few users per call (full-sort at reference defaults)
//   <BM=128,BN=128>: 2x2 waves, 2x2 accumulators (64 regs) each -- throughput shape (many users, MLP layers)
// A-operand rows = M side (users / batch rows), B-operand rows = N side (items / output features), so a lane's
// accumulator column is an N index and every store instruction writes 32 consecutive floats per half-wave.
#include "cdr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;   // 36 dwords

__device__ __forceinline__ float act_apply(int act, float v) {
    switch (act) {
        case CDR_ACT_TANH: return tanhf(v);
        case CDR_ACT_RELU: return v > 0.f ? v : 0.f;
        case CDR_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// Stage a ROWS x BK tile of op(X) into LDS (row-major, K contiguous).
//   !TRANS: X is [rows, K] with leading dimension ld (K contiguous)  -> float4 along K, ds_write_b128
//    TRANS: X is [K, rows] with leading dimension ld (rows contiguous) -> float4 along rows, 4x ds_write_b32
template <int ROWS, bool TRANS>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ X, int64_t ld,
                                           int64_t row0, int64_t nrows, int64_t k0, int64_t K, bool vec_ok) {
    constexpr int NLOAD = ROWS / 32;   // float4 per thread
    const int t = threadIdx.x;
    if (!TRANS) {
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int r = (t >> 3) + 32 * q, kc = t & 7;
            const int64_t gr = row0 + r, gk = k0 + 4 * kc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < nrows) {
                const float* p = X + gr * ld + gk;
                if (vec_ok && gk + 3 < K) v = ld4(p);
                else {
                    if (gk + 0 < K) v.x = p[0];
                    if (gk + 1 < K) v.y = p[1];
                    if (gk + 2 < K) v.z = p[2];
                    if (gk + 3 < K) v.w = p[3];
                }
            }
            st4(lds + r * LDS_STRIDE + 4 * kc, v);
        }
    } else {
        constexpr int RC = ROWS / 4;   // float4 per k-row
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int e = t + 256 * q;
            const int k = e / RC, rc = e % RC;
            const int64_t gk = k0 + k, gr = row0 + 4 * rc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < K) {
                const float* p = X + gk * ld + gr;
                if (vec_ok && gr + 3 < nrows) v = ld4(p);
                else {
                    if (gr + 0 < nrows) v.x = p[0];
                    if (gr + 1 < nrows) v.y = p[1];
                    if (gr + 2 < nrows) v.z = p[2];
                    if (gr + 3 < nrows) v.w = p[3];
                }
            }
            float* d = lds + (4 * rc) * LDS_STRIDE + k;
            d[0] = v.x; d[LDS_STRIDE] = v.y; d[2 * LDS_STRIDE] = v.z; d[3 * LDS_STRIDE] = v.w;
        }
    }
}

// EPI_SQDIST: v = -(((-2*acc) + rown[m]) + coln[n])   (sscdr.py:253-259)
template <int BM, int BN, bool TA, bool TB, bool EPI_SQDIST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A,
                                                       int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ C, int64_t ldc,
                                                       const float* __restrict__ bias, int act, int accumulate,
                                                       const float* __restrict__ rown, const float* __restrict__ coln,
                                                       int vecA, int vecB, const float* __restrict__ rowscale) {
    constexpr int WM = (BM == 128) ? 2 : 1;            // waves along M
    constexpr int WN = 4 / WM;                         // waves along N
    constexpr int TM = BM / (32 * WM);                 // 32x32 tiles per wave along M
    constexpr int TN = BN / (32 * WN);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + BM * LDS_STRIDE;

    // N tiles vary fastest across blockIdx.x so that consecutive blocks stream consecutive item rows
    const int64_t n_tiles = (N + BN - 1) / BN;
    const int64_t bm = blockIdx.x / n_tiles, bn = blockIdx.x % n_tiles;
    const int64_t m0 = bm * BM, n0 = bn * BN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        stage_tile<BM, TA>(As, A, lda, m0, M, k0, K, vecA != 0);
        stage_tile<BN, !TB>(Bs, B, ldb, n0, N, k0, K, vecB != 0);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ld4(As + ((wm * TM + i) * 32 + li) * LDS_STRIDE + 8 * s + 4 * lh);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = ld4(Bs + ((wn * TN + j) * 32 + li) * LDS_STRIDE + 8 * s + 4 * lh);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // epilogue: lane column = N index; register r -> row (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t n = n0 + (wn * TN + j) * 32 + li;
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
            const float cn = EPI_SQDIST ? coln[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= M) continue;
                float v = acc[i][j][r];
                if (EPI_SQDIST) {
                    v = -(((-2.0f * v) + rown[m]) + cn);
                } else {
                    if (rowscale) v *= rowscale[m];
                    if (accumulate == 2) v = act_apply(act, (C[m * ldc + n] + v) + bv);       // pre-activation add
                    else {
                        v = act_apply(act, v + bv);
                        if (accumulate == 1) v += C[m * ldc + n];                             // post-activation add
                    }
                }
                C[m * ldc + n] = v;
            }
        }
}

template <bool TA, bool TB, bool SQ>
int launch(hipStream_t s, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
           float* C, int64_t ldc, const float* bias, int act, int accumulate, const float* rown, const float* coln,
           const float* rowscale = nullptr) {
    // float4 global loads need 16-B aligned rows along the contiguous dimension
    const int vecA = ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
    const int vecB = ((ldb & 3) == 0) && (((uintptr_t)B & 15) == 0);
    if (M <= 64) {
        constexpr int BM = 32, BN = 128;
        const int64_t grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
        const size_t lds = (size_t)(BM + BN) * LDS_STRIDE * sizeof(float);
        gemm_f32_kernel<BM, BN, TA, TB, SQ><<<dim3((unsigned)grid), dim3(256), lds, s>>>(M, N, K, A, lda, B, ldb, C, ldc, bias,
                                                                                      act, accumulate, rown, coln, vecA, vecB, rowscale);
    } else {
        constexpr int BM = 128, BN = 128;
        const int64_t grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
        const size_t lds = (size_t)(BM + BN) * LDS_STRIDE * sizeof(float);
        gemm_f32_kernel<BM, BN, TA, TB, SQ><<<dim3((unsigned)grid), dim3(256), lds, s>>>(M, N, K, A, lda, B, ldb, C, ldc, bias,
                                                                                      act, accumulate, rown, coln, vecA, vecB, rowscale);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdr_set_error("cdr_gemm_f32: launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return CDR_OK;
}

// row squared norms: out[r] = sum_d X[r,d]^2  (one wave per row)
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ X, int64_t rows, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), TW = (int64_t)gridDim.x * 4;
    for (int64_t r = w; r < rows; r += TW) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) { const float v = X[r * D + c]; s += v * v; }
        s = group_sum<64>(s);
        if (lane == 0) out[r] = s;
    }
}

}  // namespace

extern "C" int cdr_gemm_f32_ex(void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                               int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                               const float* rowscale, int act, int accumulate) {
    CDR_CHECK_ARG(A && B && C);
    CDR_CHECK_ARG(M > 0 && N > 0 && K > 0);
    CDR_CHECK_ARG(act >= CDR_ACT_NONE && act <= CDR_ACT_SIGMOID);
    CDR_CHECK_ARG(accumulate >= 0 && accumulate <= 2);
    CDR_CHECK_ARG(((M + 127) / 128) * ((N + 127) / 128) < (int64_t)1 << 31);
    hipStream_t s = (hipStream_t)stream;
    if (!transA && transB) return launch<false, true, false>(s, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, nullptr, nullptr, rowscale);
    if (!transA && !transB) return launch<false, false, false>(s, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, nullptr, nullptr, rowscale);
    if (transA && !transB) return launch<true, false, false>(s, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, nullptr, nullptr, rowscale);
    return launch<true, true, false>(s, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, nullptr, nullptr, rowscale);
}

extern "C" int cdr_gemm_f32(void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                            int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                            int accumulate) {
    return cdr_gemm_f32_ex(stream, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, act, accumulate);
}

extern "C" int cdr_fullsort_scores_f32(void* stream, const float* user_e, int64_t U, int D, const float* slab0, int64_t n0,
                                       const float* slab1, int64_t n1, float* scores) {
    CDR_CHECK_ARG(user_e && scores && U > 0 && D > 0);
    CDR_CHECK_ARG((slab0 && n0 > 0) || (slab1 && n1 > 0));
    hipStream_t s = (hipStream_t)stream;
    const int64_t N = (slab0 ? n0 : 0) + (slab1 ? n1 : 0);
    int rc = CDR_OK;
    int64_t off = 0;
    if (slab0 && n0 > 0) {
        rc = launch<false, true, false>(s, U, n0, D, user_e, D, slab0, D, scores, N, nullptr, CDR_ACT_NONE, 0, nullptr, nullptr);
        if (rc) return rc;
        off = n0;
    }
    if (slab1 && n1 > 0)
        rc = launch<false, true, false>(s, U, n1, D, user_e, D, slab1, D, scores + off, N, nullptr, CDR_ACT_NONE, 0, nullptr, nullptr);
    return rc;
}

extern "C" int cdr_fullsort_neg_sqdist_f32(void* stream, const float* user_e, int64_t U, int D, const float* items,
                                           int64_t N, float* norm_scratch, float* scores) {
    CDR_CHECK_ARG(user_e && items && scores && norm_scratch && U > 0 && N > 0 && D > 0);
    hipStream_t s = (hipStream_t)stream;
    float* rown = norm_scratch;
    float* coln = norm_scratch + U;
    row_sqnorm_kernel<<<dim3((unsigned)((U + 3) / 4 > 2048 ? 2048 : (U + 3) / 4)), dim3(256), 0, s>>>(user_e, U, D, rown);
    row_sqnorm_kernel<<<dim3((unsigned)((N + 3) / 4 > 2048 ? 2048 : (N + 3) / 4)), dim3(256), 0, s>>>(items, N, D, coln);
    return launch<false, true, true>(s, U, N, D, user_e, D, items, D, scores, N, nullptr, CDR_ACT_NONE, 0, rown, coln);
}
