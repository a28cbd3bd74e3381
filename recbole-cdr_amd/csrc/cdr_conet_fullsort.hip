// CoNet full-sort scoring in ONE launch (conet.py:222-242): every evaluated user against every target item through the target
// tower WITHOUT cross terms -- the reference loops over users in Python and pushes a repeat()ed [N, 2D] input through the tower
// per user; the product's previous path hoisted the separable first layer but still ran a Python loop of generic contractions with
// [N, h] intermediates in HBM per user.
//
// The first layer is separable: W1 [u ; i] + b1 = (W1u u + b1) + W1i i =: Q[u] + P[i].  P [N, h1] and Q [U, h1] come from two plain
// contractions (cdr_gemm_f32_ex); this kernel does everything behind them for all U x N pairs:
//      h1 = relu(P[i] + Q[u]) -> h2 = relu(W2 h1 + b2) -> ... -> sigmoid(wo . hL + bo)            ~2 (h1 d2 + d2 d3 + ...) FLOP per pair
//
// Layout.  A wave owns a tile of 32 items and keeps their P rows in registers for all the users it serves; the layers run
// TRANSPOSED on v_mfma_f32_32x32x2_f32 -- A = the layer's weights (rows = output features), B = the activations (columns = the 32
// items) -- so a layer's accumulator (C[feature][item]: lane = item + 32 * ((feature / 4) % 2), register = 4 * (feature / 8) +
// feature % 4) IS the next layer's B operand, register for register: lane half h holds features 8 g + 4 h + (0..3) in registers
// 4 g + (0..3), and K step r of the next layer contracts "register r of both halves" -- only the order of the K sum changes, and the
// weights are loaded once per lane in exactly that order.  No activation ever leaves the registers; nothing but P, Q and the
// scores touches memory.  Q rows of the workgroup's user chunk sit in LDS (read as wave-half broadcasts).
//
// Work: MFMA steps per (32 items x 1 user) = h1/2 + 4 ceil(d2/8) + 4 ceil(d3/8) (...): 56 x 64 cycles for [64,32,16,8] (75 % of the
// executed MFMA flops are algorithmic: the 16- and 8-row layers use half / a quarter of a 32-row tile).  The VALU side (P + q, ReLU,
// bias: ~135 instructions per tile-user) runs under the MFMAs of the SIMD's other wave.
// (Tried: two users per pass, i.e. two independent accumulator chains per wave: 260 VGPRs -> one wave per SIMD instead of two, 0.57 ->
//  0.47 of the fp32 MFMA peak at 64 users x 1 M items.  Two resident waves hide more than two chains in one.)
#include "cdr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kFsBlock = 256;                    // 4 waves: 4 item tiles per workgroup pass, one user chunk
constexpr int kFsUsers = 32;                     // users per LDS chunk (32 x 64 floats = 8 KB)

struct fs_args {
    const float* P; int64_t ldp;                 // [N, >= h1]  item part of the first layer (no bias)
    const float* Q; int64_t ldq;                 // [U, >= h1]  user part + bias
    int64_t U, N;
    int h1, n_tail;                              // n_tail in 1..3 layers behind the first
    int d[3];                                    // their widths
    const float* W[3]; const float* b[3];        // W[t]: [d[t], d_in] row-major (d_in = h1 or d[t-1])
    const float* wo; const float* bo;            // output unit [d_last], [1]
    float* out; int64_t ldo;                     // [U, N]
    int users_per_block;                         // a multiple of 8
    // QF (a few users per call -- recbole's evaluation hands over ONE user at the default eval_batch_size): Q is not read but formed while
    // the chunk is staged, Q[u][f] = b1[f] + <W1u[f, :], UT[uid[u], :]>, so a call is this one launch (no gather, no [U, h1] contraction)
    const float* UT; int64_t ldu; const int64_t* uid; const float* W1u; int64_t ldw1; const float* b1; int D;
};

#define FS_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

__device__ __forceinline__ f32x16 fs_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// feature held by (register r, lane half h) of a 32x32 accumulator
__device__ __forceinline__ int fs_feat(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }

// S1: K steps of the first tail layer (h1 <= 2 S1; half h contracts features h S1 + s).  G2/G3/G4: 8-feature register groups of the
// tail layers' outputs (G = 0: layer absent); the next layer contracts 4 G steps.
// The Q rows of users [u0, u0 + nu) into LDS.  QF: formed here from the table rows -- four lanes per (user, feature) element, each a quarter
// of the D-long dot with its loads issued together, summed inside the quad (the first form, one lane per element walking D dependent
// iterations, cost ~40 us per staging: twice the whole scoring launch at one user x 1 M items).
// (inlined as first written, its float4 temporaries pushed the default tower's kernel from 184 + 32 to 228 + 32 registers -- one resident
//  wave per SIMD instead of two, 1.5x on the scoring loop; as a call, 249 + 32 and scratch.  The kernel now carries the two-waves bound.)
template <int S1>
__device__ __forceinline__ void fs_form_q(const fs_args& a, float* __restrict__ Ql, int64_t u0, int nu) {
    {
        const int Dq = ((a.D + 15) >> 4) << 2;                       // a quarter of the row, rounded up to whole float4s
        const bool vec = !(a.ldu & 3) && !(a.ldw1 & 3) && !((uintptr_t)a.UT & 15) && !((uintptr_t)a.W1u & 15);
        for (int t = threadIdx.x; t < nu * 2 * S1 * 4; t += kFsBlock) {      // (8 S1 is a multiple of 64: whole waves run the same trips)
            const int e = t >> 2, part = t & 3;
            const int uu = e / (2 * S1), f = e - uu * 2 * S1;
            float q = 0.f;
            if (f < a.h1) {
                const float* __restrict__ ur = a.UT + a.uid[u0 + uu] * a.ldu;
                const float* __restrict__ wr = a.W1u + (int64_t)f * a.ldw1;
                const int d0 = part * Dq, d1 = d0 + Dq < a.D ? d0 + Dq : a.D;
                int d = d0;
                if (vec) {
                    float q1 = 0.f, q2 = 0.f, q3 = 0.f;
                    for (; d + 16 <= d1; d += 16) {
                        const float4 w0 = ld4(wr + d), w1 = ld4(wr + d + 4), w2 = ld4(wr + d + 8), w3 = ld4(wr + d + 12);
                        const float4 x0 = ld4(ur + d), x1 = ld4(ur + d + 4), x2 = ld4(ur + d + 8), x3 = ld4(ur + d + 12);
                        q += w0.x * x0.x + w0.y * x0.y + w0.z * x0.z + w0.w * x0.w;
                        q1 += w1.x * x1.x + w1.y * x1.y + w1.z * x1.z + w1.w * x1.w;
                        q2 += w2.x * x2.x + w2.y * x2.y + w2.z * x2.z + w2.w * x2.w;
                        q3 += w3.x * x3.x + w3.y * x3.y + w3.z * x3.z + w3.w * x3.w;
                    }
                    for (; d + 4 <= d1; d += 4) {
                        const float4 w0 = ld4(wr + d), x0 = ld4(ur + d);
                        q += w0.x * x0.x + w0.y * x0.y + w0.z * x0.z + w0.w * x0.w;
                    }
                    q += (q1 + q2) + q3;
                }
                for (; d < d1; ++d) q += wr[d] * ur[d];
            }
            q += __shfl_xor(q, 1, 64);
            q += __shfl_xor(q, 2, 64);
            if (part == 0) Ql[e] = f < a.h1 ? q + a.b1[f] : 0.f;
        }
    }
}

template <int S1, bool QF>
__device__ __forceinline__ void fs_stage_q(const fs_args& a, float* __restrict__ Ql, int64_t u0, int nu) {
    if (QF) {
        fs_form_q<S1>(a, Ql, u0, nu);
    } else {
        for (int e = threadIdx.x; e < nu * 2 * S1; e += kFsBlock) {
            const int uu = e / (2 * S1), f = e - uu * 2 * S1;
            Ql[e] = f < a.h1 ? a.Q[(u0 + uu) * a.ldq + f] : 0.f;
        }
    }
}

template <int S1, int G2, int G3, int G4, bool QF>
__global__ __launch_bounds__(kFsBlock, (G3 <= 2 && G4 <= 1) ? 2 : 1) void conet_fullsort_kernel(fs_args a) {
    __shared__ __attribute__((aligned(16))) float Ql[kFsUsers * 2 * S1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const int d2 = a.d[0], d3 = G3 ? a.d[1] : 0, d4 = G4 ? a.d[2] : 0;
    constexpr int R2 = 4 * G2, R3 = 4 * G3, R4 = 4 * G4;
    const int64_t u_lo = (int64_t)blockIdx.y * a.users_per_block;
    const int64_t u_hi = u_lo + a.users_per_block < a.U ? u_lo + a.users_per_block : a.U;
    // a workgroup whose users fit ONE chunk stages them once, not once per tile round -- first thing, so that its loads (QF: id -> table
    // row, a dependent pair) are in flight under the ~100 weight-fragment loads below
    const bool one_chunk = u_hi - u_lo <= kFsUsers;
    if (one_chunk) fs_stage_q<S1, QF>(a, Ql, u_lo, (int)(u_hi - u_lo));

    // ---- static per-lane weight fragments, in the K order of the register chain ------------------------------------------------
    // (every address is clamped into its array and the value masked afterwards: straight-line loads -- a bounds BRANCH per element made
    //  this prologue ~200 divergent branches, most of the launch for one-user calls)
    const auto ldz = [](const float* __restrict__ base, int64_t idx, bool ok) { const float v = base[ok ? idx : 0]; return ok ? v : 0.f; };
    float w2[S1], b2[R2];
    if (!(a.h1 & 3) && !((uintptr_t)a.W[0] & 15)) {                  // 16-byte requests: a group of four features is wholly inside or outside
#pragma unroll
        for (int s = 0; s < S1; s += 4) {
            const int f = h * S1 + s;
            const bool ok = c < d2 && f < a.h1;
            const float4 v = ld4(a.W[0] + (ok ? (int64_t)c * a.h1 + f : 0));
            w2[s] = ok ? v.x : 0.f; w2[s + 1] = ok ? v.y : 0.f; w2[s + 2] = ok ? v.z : 0.f; w2[s + 3] = ok ? v.w : 0.f;
        }
    } else {
#pragma unroll
        for (int s = 0; s < S1; ++s) { const int f = h * S1 + s; w2[s] = ldz(a.W[0], (int64_t)c * a.h1 + f, c < d2 && f < a.h1); }
    }
#pragma unroll
    for (int r = 0; r < R2; ++r) { const int j = fs_feat(r, h); b2[r] = ldz(a.b[0], j, j < d2); }
    float w3[R2 ? R2 : 1], b3[R3 ? R3 : 1], w4[R3 ? R3 : 1], b4[R4 ? R4 : 1];
    if (G3) {
#pragma unroll
        for (int r = 0; r < R2; ++r) { const int f = fs_feat(r, h); w3[r] = ldz(a.W[1], (int64_t)c * d2 + f, c < d3 && f < d2); }
#pragma unroll
        for (int r = 0; r < R3; ++r) { const int j = fs_feat(r, h); b3[r] = ldz(a.b[1], j, j < d3); }
    }
    if (G4) {
#pragma unroll
        for (int r = 0; r < R3; ++r) { const int f = fs_feat(r, h); w4[r] = ldz(a.W[2], (int64_t)c * d3 + f, c < d4 && f < d3); }
#pragma unroll
        for (int r = 0; r < R4; ++r) { const int j = fs_feat(r, h); b4[r] = ldz(a.b[2], j, j < d4); }
    }
    constexpr int RL = G4 ? R4 : (G3 ? R3 : R2);                      // registers of the last layer's output
    const int dl = G4 ? d4 : (G3 ? d3 : d2);
    float wo[RL];
#pragma unroll
    for (int r = 0; r < RL; ++r) { const int j = fs_feat(r, h); wo[r] = ldz(a.wo, j, j < dl); }
    const float bo = a.bo[0];
    const bool p_vec = !(a.ldp & 3) && !(a.h1 & 3) && !((uintptr_t)a.P & 15);

    const int64_t n_tiles = (a.N + 31) >> 5;
    const int64_t tile_stride = (int64_t)gridDim.x * 4;
    // every wave of the workgroup runs the same number of tile rounds (the Q chunk is staged behind workgroup barriers)
    const int64_t rounds = (n_tiles + tile_stride - 1) / tile_stride;
    if (one_chunk) __syncthreads();                                  // (staged at the top of the kernel)
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t tile = ((int64_t)rd * gridDim.x + blockIdx.x) * 4 + wave;
        const bool live = tile < n_tiles;
        const int64_t item = tile * 32 + c;
        const bool iv = live && item < a.N;
        // the tile's P rows: lane (item c, half h) keeps features h S1 .. h S1 + S1 - 1
        float p[S1];
        {
            const float* pr = a.P + (iv ? item : 0) * a.ldp;
#pragma unroll
            for (int s = 0; s < S1; s += 4) {
                const int f = h * S1 + s;
                if (p_vec) {                                        // h1 % 4 == 0: a 16-byte group is wholly inside or wholly outside the row
                    const bool ok = iv && f < a.h1;
                    const float4 v = ld4(pr + (ok ? f : 0));
                    p[s] = ok ? v.x : 0.f; p[s + 1] = ok ? v.y : 0.f; p[s + 2] = ok ? v.z : 0.f; p[s + 3] = ok ? v.w : 0.f;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const bool ok = iv && f + q < a.h1; const float v = pr[ok ? f + q : 0]; p[s + q] = ok ? v : 0.f; }
                }
            }
        }
        for (int64_t u0 = u_lo; u0 < u_hi; u0 += kFsUsers) {
            const int nu = (int)(u_hi - u0 < kFsUsers ? u_hi - u0 : kFsUsers);
            if (!one_chunk) {
                __syncthreads();                                   // the previous chunk has been consumed by every wave
                fs_stage_q<S1, QF>(a, Ql, u0, nu);
                __syncthreads();
            }
            if (!live) continue;
            for (int uu = 0; uu < nu; ++uu) {
                const float* q = Ql + uu * 2 * S1 + h * S1;
                f32x16 acc = fs_zero();
#pragma unroll
                for (int s = 0; s < S1; s += 4) {
                    const float4 qv = *reinterpret_cast<const float4*>(q + s);
                    FS_MFMA(acc, w2[s], fmaxf(p[s] + qv.x, 0.f));
                    FS_MFMA(acc, w2[s + 1], fmaxf(p[s + 1] + qv.y, 0.f));
                    FS_MFMA(acc, w2[s + 2], fmaxf(p[s + 2] + qv.z, 0.f));
                    FS_MFMA(acc, w2[s + 3], fmaxf(p[s + 3] + qv.w, 0.f));
                }
                if (G3) {
                    f32x16 a3 = fs_zero();
#pragma unroll
                    for (int r = 0; r < R2; ++r) FS_MFMA(a3, w3[r], fmaxf(acc[r] + b2[r], 0.f));
                    if (G4) {
                        f32x16 a4 = fs_zero();
#pragma unroll
                        for (int r = 0; r < R3; ++r) FS_MFMA(a4, w4[r], fmaxf(a3[r] + b3[r], 0.f));
#pragma unroll
                        for (int r = 0; r < R4; ++r) acc[r] = fmaxf(a4[r] + b4[r], 0.f);
                    } else {
#pragma unroll
                        for (int r = 0; r < R3; ++r) acc[r] = fmaxf(a3[r] + b3[r], 0.f);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R2; ++r) acc[r] = fmaxf(acc[r] + b2[r], 0.f);
                }
                // output unit: this half's features in register order, then the other half's partial (lane ^ 32)
                float z = 0.f;
#pragma unroll
                for (int r = 0; r < RL; ++r) z += wo[r] * acc[r];
                z += __shfl_xor(z, 32, 64);
                if (h == 0 && iv) a.out[(u0 + uu) * a.ldo + item] = 1.0f / (1.0f + expf(-(z + bo)));
            }
        }
    }
}

template <int S1, int G2, int G3, int G4, bool QF>
int fs_launch_q(const fs_args& a, hipStream_t s) {
    const int64_t n_tiles = (a.N + 31) >> 5;
    // user chunks per block: enough workgroups to cover the chip's 1,024 SIMDs about twice when the problem allows it
    fs_args b = a;
    int64_t gx = (n_tiles + 3) / 4;
    // users per workgroup: a multiple of 8, as few as it takes to give every SIMD ~4 waves (two are resident at this register count;
    // the rest evens out the tail) -- a wave's prologue (its weight fragments: ~100 loads) is paid once per (tile, user group)
    const int64_t groups = (a.U + 7) / 8;
    int64_t gy = 1;
    while (gx * 4 * gy < 4 * 4 * CDR_NUM_CU && gy < groups) gy *= 2;
    if (gy > groups) gy = groups;
    const int64_t per = (groups + gy - 1) / gy;
    gy = (groups + per - 1) / per;
    b.users_per_block = (int)(per * 8);
    if (gx > 8 * CDR_NUM_CU) gx = 8 * CDR_NUM_CU;            // waves walk the remaining tiles (their weights stay loaded)
    conet_fullsort_kernel<S1, G2, G3, G4, QF><<<dim3((unsigned)gx, (unsigned)gy), dim3(kFsBlock), 0, s>>>(b);
    return 0;
}

template <int S1, int G2, int G3, int G4>
int fs_launch(const fs_args& a, hipStream_t s) {
    return a.uid ? fs_launch_q<S1, G2, G3, G4, true>(a, s) : fs_launch_q<S1, G2, G3, G4, false>(a, s);
}

int fs_dispatch(const fs_args& a, hipStream_t s) {
    const int n_tail = a.n_tail;
    const int g2 = (a.d[0] + 7) / 8, g3 = n_tail > 1 ? (a.d[1] + 7) / 8 : 0, g4 = n_tail > 2 ? (a.d[2] + 7) / 8 : 0;
    const bool small1 = a.h1 <= 32;
    // the reference's default tower [.., 64, 32, 16, 8] (properties/model/CoNet.yaml) gets its exact instantiation; anything else within
    // range runs zero-padded on the next larger one
    if (!small1 && n_tail == 3 && g2 <= 4 && g3 <= 2 && g4 <= 1) fs_launch<32, 4, 2, 1>(a, s);
    else if (n_tail == 3) small1 ? fs_launch<16, 4, 4, 4>(a, s) : fs_launch<32, 4, 4, 4>(a, s);
    else if (n_tail == 2) small1 ? fs_launch<16, 4, 4, 0>(a, s) : fs_launch<32, 4, 4, 0>(a, s);
    else small1 ? fs_launch<16, 4, 0, 0>(a, s) : fs_launch<32, 4, 0, 0>(a, s);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

}  // namespace

extern "C" int cdr_conet_fullsort_supported(int h1, int n_tail, const int* tail_dims) {
    if (h1 < 1 || h1 > 64 || n_tail < 1 || n_tail > 3 || !tail_dims) return 0;
    for (int t = 0; t < n_tail; ++t)
        if (tail_dims[t] < 1 || tail_dims[t] > 32) return 0;
    return 1;
}

extern "C" int cdr_conet_fullsort(void* stream, const float* P, int64_t ldp, const float* Q, int64_t ldq, int64_t U, int64_t N, int h1,
                                  int n_tail, const int* tail_dims, const float* const* W, const float* const* b, const float* wo,
                                  const float* bo, float* out, int64_t ldo) {
    CDR_CHECK_ARG(P && Q && out && W && b && wo && bo && tail_dims && U > 0 && N > 0 && ldp >= h1 && ldq >= h1 && ldo >= N);
    if (!cdr_conet_fullsort_supported(h1, n_tail, tail_dims)) {
        cdr_set_error("cdr_conet_fullsort: tower [%d -> %d layers] outside the kernel's range (first width <= 64, 1..3 further layers of width <= 32)",
                      h1, n_tail);
        return CDR_EINVAL;
    }
    fs_args a{};
    a.P = P; a.ldp = ldp; a.Q = Q; a.ldq = ldq; a.U = U; a.N = N; a.h1 = h1; a.n_tail = n_tail; a.wo = wo; a.bo = bo; a.out = out; a.ldo = ldo;
    for (int t = 0; t < n_tail; ++t) {
        CDR_CHECK_ARG(W[t] && b[t]);
        a.d[t] = tail_dims[t]; a.W[t] = W[t]; a.b[t] = b[t];
    }
    return fs_dispatch(a, (hipStream_t)stream);
}

// The same scores with the user half of the first layer formed inside the launch: uid [U] rows of user_table [*, ldu >= D],
// W1u = the first D columns of the first layer's weight [h1, ldw1 >= D], b1 [h1].  Meant for the few-users call of recbole's evaluation
// loop (conet.py:222-242 is entered once per eval batch: one user at the default eval_batch_size over a large catalogue); any U is correct,
// but every workgroup re-forms its users' Q rows, so large U belongs to cdr_conet_fullsort.
extern "C" int cdr_conet_fullsort_users(void* stream, const float* P, int64_t ldp, const float* user_table, int64_t ldu, const int64_t* uid,
                                        const float* W1u, int64_t ldw1, const float* b1, int D, int64_t U, int64_t N, int h1, int n_tail,
                                        const int* tail_dims, const float* const* W, const float* const* b, const float* wo, const float* bo,
                                        float* out, int64_t ldo) {
    CDR_CHECK_ARG(P && user_table && uid && W1u && b1 && out && W && b && wo && bo && tail_dims);
    CDR_CHECK_ARG(U > 0 && N > 0 && D > 0 && ldp >= h1 && ldu >= D && ldw1 >= D && ldo >= N);
    if (!cdr_conet_fullsort_supported(h1, n_tail, tail_dims)) {
        cdr_set_error("cdr_conet_fullsort_users: tower [%d -> %d layers] outside the kernel's range (first width <= 64, 1..3 further layers of width <= 32)",
                      h1, n_tail);
        return CDR_EINVAL;
    }
    fs_args a{};
    a.P = P; a.ldp = ldp; a.U = U; a.N = N; a.h1 = h1; a.n_tail = n_tail; a.wo = wo; a.bo = bo; a.out = out; a.ldo = ldo;
    a.UT = user_table; a.ldu = ldu; a.uid = uid; a.W1u = W1u; a.ldw1 = ldw1; a.b1 = b1; a.D = D;
    for (int t = 0; t < n_tail; ++t) {
        CDR_CHECK_ARG(W[t] && b[t]);
        a.d[t] = tail_dims[t]; a.W[t] = W[t]; a.b[t] = b[t];
    }
    return fs_dispatch(a, (hipStream_t)stream);
}
