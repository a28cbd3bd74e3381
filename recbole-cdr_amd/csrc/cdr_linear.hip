// Weight and bias gradients of y = act(x W^T + b) for SMALL batches (recbole MLPLayers / nn.Linear in the map phases and the small
// models: emcdr.py:86-93, sscdr.py:60-66, dcdcsr.py, clfm.py ...):  dW [out, in] = gz^T x,  db [out] = column sums of gz,
// gz [rows, out], x [rows, in], rows of a few hundred.  Through the general contraction that is three launches -- a transposed-A
// GEMM whose 2-8 workgroups walk the whole batch (28 us for 300 rows x 128 x 64: one long latency chain per workgroup), a column
// sum in slabs and its finishing block -- in steps that are launch-bound to begin with (SSCDR's map step: 56 + 18 of 147 us).
// Here ONE launch: a workgroup per 32 x 32 tile of dW, its eight waves taking an eighth of the batch rows each
// (v_mfma_f32_32x32x2_f32: both operands are read down a column, 128 contiguous bytes per half-wave and row), partial tiles added
// through LDS in wave order (no float atomics: run-to-run identical), the bias gradient as the running sum of the A operand.
#include "cdr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kBlock = 256;
constexpr int kWgPart = 32 * 32 + 32;                          // floats of one partial: a dW tile + its bias row
constexpr int kWgWaves = 8, kWgBlock = 64 * kWgWaves;     // linear_wgrad_small_kernel: waves (= batch-row slices) per 32 x 32 tile

// gz = gy * act'(y) on the way into the contractions (act_bwd_kernel's expressions): the activation backward as a launch of its own
// is a tenth of a small step
__device__ __forceinline__ float dact(int act, float y, float g) {
    float d;
    switch (act) {
        case CDR_ACT_TANH: d = 1.0f - y * y; break;
        case CDR_ACT_RELU: d = y > 0.f ? 1.0f : 0.f; break;
        case CDR_ACT_SIGMOID: d = y * (1.0f - y); break;
        default: d = 1.0f;
    }
    return g * d;
}

__global__ __launch_bounds__(kWgBlock) void linear_wgrad_small_kernel(const float* __restrict__ gz, const float* __restrict__ yact, int act,
                                                                     const float* __restrict__ x, int64_t rows, int dout, int din,
                                                                     float* __restrict__ dW, float* __restrict__ db, int64_t chunk,
                                                                     float* __restrict__ part, unsigned* __restrict__ tickets) {
    __shared__ float red[kWgWaves][32 * 33];
    __shared__ float bred[kWgWaves][32];
    __shared__ int last_block;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int oc = o0 + li < dout ? o0 + li : dout - 1, ic = i0 + li < din ? i0 + li : din - 1;     // clamped: dropped at the store
    // batch rows [c0, c1) of this workgroup (blockIdx.z: larger batches are cut into chunks), [r0, r1) of this wave, both even-aligned
    const int64_t c0 = (int64_t)blockIdx.z * chunk, c1 = c0 + chunk < rows ? c0 + chunk : rows;
    const int64_t per = ((c1 - c0 + kWgWaves - 1) / kWgWaves + 1) & ~(int64_t)1;
    const int64_t r0 = c0 + wave * per < c1 ? c0 + wave * per : c1, r1 = r0 + per < c1 ? r0 + per : c1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    constexpr int UN = 8;                                          // 16 rows per round; the next round is requested before this one's MFMAs
    float a0[UN], b0[UN], a1[UN], b1[UN];
    auto load = [&](int64_t r, float (&a)[UN], float (&b)[UN]) {
        float g[UN], yv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t row = r + 2 * u + lh;
            const int64_t rc = row < r1 ? row : r0;
            g[u] = gz[rc * dout + oc];
            b[u] = x[rc * din + ic];
            yv[u] = yact ? yact[rc * dout + oc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = r + 2 * u + lh < r1;
            a[u] = ok ? (yact ? dact(act, yv[u], g[u]) : g[u]) : 0.f;          // gz given as the output gradient gy (+ the outputs)
            if (!ok) b[u] = 0.f;
        }
    };
    auto fma_round = [&](const float (&a)[UN], const float (&b)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            bsum += a[u];
        }
    };
    if (r0 < r1) {
        load(r0, a0, b0);
        for (int64_t r = r0; r < r1; r += 4 * UN) {
            if (r + 2 * UN < r1) load(r + 2 * UN, a1, b1);
            fma_round(a0, b0);
            if (r + 2 * UN < r1) {
                if (r + 4 * UN < r1) load(r + 4 * UN, a0, b0);
                fma_round(a1, b1);
            }
        }
    }
    // acc[r]: tile row (out) = (r & 3) + 8 (r >> 2) + 4 lh, tile column (in) = li
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
    bsum += __shfl_xor(bsum, 32);                                   // the two row parities of out column li
    if (lh == 0) bred[wave][li] = bsum;
    __syncthreads();
    const int nz = gridDim.z;
    if (nz == 1) {
        for (int e = threadIdx.x; e < 32 * 32; e += kWgBlock) {
            const int ro = e >> 5, ci = e & 31;
            if (o0 + ro < dout && i0 + ci < din) {
                const int q = ro * 33 + ci;
                float v = red[0][q];
#pragma unroll
                for (int w = 1; w < kWgWaves; ++w) v += red[w][q]; // wave order: run-to-run identical
                dW[(int64_t)(o0 + ro) * din + i0 + ci] = v;
            }
        }
        if (db && blockIdx.y == 0 && threadIdx.x < 32 && o0 + (int)threadIdx.x < dout) {
            const int q = threadIdx.x;
            float v = bred[0][q];
#pragma unroll
            for (int w = 1; w < kWgWaves; ++w) v += bred[w][q];
            db[o0 + q] = v;
        }
        return;
    }
    // several chunks per tile: every workgroup leaves its partial tile (+ bias row) in the workspace, and the one that signs in LAST on
    // the tile's counter adds them in chunk order -- fixed order again.  The partials travel with system-scope stores and loads (written
    // through / read past the per-XCD L2s): no release fence, which on this part is a write-back of the whole L2 per workgroup; one
    // counter per tile, so no more than gridDim.z atomics meet on an address.
    // (this hand-over is relaxed in the language's memory model; it is ordered by the hardware's behaviour -- sc0 sc1 stores are
    // acknowledged by memory before s_waitcnt vmcnt(0) retires, sc0 sc1 loads never hit a cache -- so it is tied to the one target below.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "cdr_linear.hip: the fence-free partial-tile hand-over is only valid on gfx950 (write-through system-scope stores); add __threadfence() pairs for another target"
#endif
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* mine = part + ((size_t)tile * nz + blockIdx.z) * kWgPart;
    for (int e = threadIdx.x; e < 32 * 32 + 32; e += kWgBlock) {
        float v;
        if (e < 1024) {
            const int q = (e >> 5) * 33 + (e & 31);
            v = red[0][q];
#pragma unroll
            for (int w = 1; w < kWgWaves; ++w) v += red[w][q];
        } else {
            v = bred[0][e - 1024];
#pragma unroll
            for (int w = 1; w < kWgWaves; ++w) v += bred[w][e - 1024];
        }
        __hip_atomic_store(mine + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's partials have been written through ...
    __syncthreads();                                   // ... every thread's have
    if (threadIdx.x == 0) last_block = atomicInc(tickets + tile, (unsigned)nz - 1) == (unsigned)nz - 1 ? 1 : 0;   // wraps to 0 for the next launch
    __syncthreads();
    if (!last_block) return;
    const float* all = part + (size_t)tile * nz * kWgPart;
    for (int e = threadIdx.x; e < 32 * 32 + 32; e += kWgBlock) {
        float v = __hip_atomic_load(all + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int z = 1; z < nz; ++z) v += __hip_atomic_load(all + (size_t)z * kWgPart + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (e < 1024) {
            const int ro = e >> 5, ci = e & 31;
            if (o0 + ro < dout && i0 + ci < din) dW[(int64_t)(o0 + ro) * din + i0 + ci] = v;
        } else if (db && blockIdx.y == 0 && o0 + (e - 1024) < dout) db[o0 + e - 1024] = v;
    }
}

// ---- y = act(x W^T + b) and dx = gz W for the same small batches: one WAVE per 32 x 32 output tile, operands straight from global
// memory (the general contraction stages tiles through LDS behind workgroup barriers: 12 us for 300 x 128 x 64, most of it start-up).
//   NT: C[m][n] = sum_k A[m][k] W[n][k]   (W [N, K] row-major: the forward)      NN: C[m][n] = sum_k A[m][k] W[k][n]   (W [K, N]: dx)
// K in steps of 8 (v_mfma_f32_32x32x2_f32 x 4: lane half lh takes k = 8 s + 4 lh .. + 3), four steps' operands requested before
// their MFMAs.  K % 4 == 0; rows and columns past the edge are clamped on the way in and dropped at the store.
template <bool NN>
__global__ __launch_bounds__(64) void linear_small_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
                                                          int64_t M, int N, int K, const float* __restrict__ bias, int act,
                                                          float* __restrict__ C, int64_t ldc, const float* __restrict__ a_out,
                                                          int a_act) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    const int n0 = blockIdx.y * 32;
    const int64_t mr = m0 + li < M ? m0 + li : M - 1;
    const int nc = n0 + li < N ? n0 + li : N - 1;
    const float* __restrict__ a = A + mr * lda + 4 * lh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int UN = 4;                                          // 32 K per round; the next round is requested before this one's MFMAs
    float4 av0[UN], bv0[UN], av1[UN], bv1[UN];
    auto load = [&](int k0, float4 (&av)[UN], float4 (&bv)[UN]) {
        float4 yv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + 8 * u + 4 * lh;
            const int kc = k < K ? k : 0;
            av[u] = ld4(a + (kc - 4 * lh));
            if (NN) { const float* p = W + (int64_t)kc * ldw + nc; bv[u] = make_float4(p[0], p[ldw], p[2 * ldw], p[3 * ldw]); }
            else bv[u] = ld4(W + (int64_t)nc * ldw + kc);
            if (a_out) yv[u] = ld4(a_out + mr * lda + kc);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (a_out)                                                              // A = gy (.) act'(outputs): dx = gz W without a gz launch
                av[u] = make_float4(dact(a_act, yv[u].x, av[u].x), dact(a_act, yv[u].y, av[u].y), dact(a_act, yv[u].z, av[u].z),
                                    dact(a_act, yv[u].w, av[u].w));
            if (k0 + 8 * u + 4 * lh >= K) av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto fma_round = [&](const float4 (&av)[UN], const float4 (&bv)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    };
    load(0, av0, bv0);
    for (int k0 = 0; k0 < K; k0 += 16 * UN) {
        if (k0 + 8 * UN < K) load(k0 + 8 * UN, av1, bv1);
        fma_round(av0, bv0);
        if (k0 + 8 * UN < K) {
            if (k0 + 16 * UN < K) load(k0 + 16 * UN, av0, bv0);
            fma_round(av1, bv1);
        }
    }
    if (n0 + li < N) {
        const float b = bias ? bias[n0 + li] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < M) {
                float v = acc[r] + b;
                if (act == CDR_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (act == CDR_ACT_TANH) v = tanhf(v);
                else if (act == CDR_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                C[m * ldc + n0 + li] = v;
            }
        }
    }
}

}  // namespace

namespace {
// up to 512 rows per workgroup (64 per wave); more rows -> more chunks per tile, added by the last workgroup to finish (256-row chunks
// were slower: 15.3 -> 18.0 us at 4,096 rows)
inline void wgrad_split(int64_t rows, int64_t* chunk, int64_t* nz) {
    int64_t n = (rows + 511) / 512;
    if (n > 64) n = 64;
    *chunk = (((rows + n - 1) / n) + 1) & ~(int64_t)1;
    *nz = (rows + *chunk - 1) / *chunk;
}
}  // namespace

extern "C" int cdr_linear_wgrad_small_workspace(int64_t rows, int dout, int din, size_t* bytes) {
    CDR_CHECK_ARG(bytes && rows > 0 && dout > 0 && din > 0);
    int64_t chunk, nz;
    wgrad_split(rows, &chunk, &nz);
    *bytes = nz > 1 ? (size_t)((dout + 31) / 32) * ((din + 31) / 32) * nz * kWgPart * sizeof(float) : 0;
    return CDR_OK;
}

extern "C" int cdr_linear_wgrad_small(cdr_ctx* ctx, void* stream, const float* gz, const float* y_out, int act, const float* x, int64_t rows,
                                      int dout, int din, float* dW, float* db, void* workspace, size_t workspace_bytes) {
    CDR_CHECK_ARG(ctx && gz && x && dW && rows > 0 && dout > 0 && din > 0);
    const int tx = (dout + 31) / 32, ty = (din + 31) / 32;
    CDR_CHECK_ARG(tx <= 65535 && ty <= 65535);
    if ((int64_t)tx * ty > CDR_TICKETS) { cdr_set_error("cdr_linear_wgrad_small: %d x %d tiles exceed the %d sign-in counters", tx, ty, CDR_TICKETS); return CDR_EINVAL; }
    int64_t chunk, nz;
    wgrad_split(rows, &chunk, &nz);
    size_t need = 0;
    int rc = cdr_linear_wgrad_small_workspace(rows, dout, din, &need);
    if (rc) return rc;
    // the partial tiles live in the CALLER's workspace (not in the context's grow-on-demand scratch: a captured hipGraph keeps the
    // pointer it was recorded with)
    CDR_CHECK_ARG(need == 0 || (workspace && workspace_bytes >= need));
    linear_wgrad_small_kernel<<<dim3(tx, ty, (unsigned)nz), dim3(kWgBlock), 0, (hipStream_t)stream>>>(gz, y_out, act, x, rows, dout, din, dW, db,
                                                                                                     chunk, (float*)workspace, ctx->tickets);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_linear_small(void* stream, int w_is_k_major, const float* A, int64_t lda, const float* W, int64_t ldw, int64_t M, int N,
                                int K, const float* bias, int act, float* C, int64_t ldc, const float* a_out, int a_act) {
    CDR_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0 && (K & 3) == 0 && (lda & 3) == 0 && ((uintptr_t)A & 15) == 0);
    CDR_CHECK_ARG(w_is_k_major || ((ldw & 3) == 0 && ((uintptr_t)W & 15) == 0));
    CDR_CHECK_ARG((M + 31) / 32 <= 0x7FFFFFFF && (N + 31) / 32 <= 65535);
    const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N + 31) / 32));
    if (w_is_k_major) linear_small_kernel<true><<<grid, dim3(64), 0, (hipStream_t)stream>>>(A, lda, W, ldw, M, N, K, bias, act, C, ldc, a_out, a_act);
    else linear_small_kernel<false><<<grid, dim3(64), 0, (hipStream_t)stream>>>(A, lda, W, ldw, M, N, K, bias, act, C, ldc, a_out, a_act);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
