// CoNet towers fused (conet.py:105-203): the whole `source_forward` + `target_forward` + BCE + reg of one step in ONE
// forward launch, and its backward in three (data gradients, weight gradients by rows, fixed-order reduction).
//
// Per step the reference runs, for a stack of R = B_source + B_target rows, both towers through L cross units
//     s' = relu(s Ws^T + bs + m (.) (t H^T)),   t' = relu(t Wt^T + bt + m (.) (s H^T)),   m = 1 on overlapped rows
// (four products per layer: 152.6 kFLOP per row at dims [256,64,32,16,8]) -- round 1 served that with 54 generic GEMM
// launches per step and ran at 1.5 % of the fp32 MFMA roofline.  Here:
//
//   conet_fwd_kernel     a workgroup owns 32 rows: gathers the four embedding rows of each (coalesced 512-B segments)
//                        into LDS as [s | t], then walks the layers with v_mfma_f32_32x32x2_f32 -- activations never
//                        leave LDS between layers; the first layer's weight fragments stream from L2 (four K steps per
//                        request so that every 128-B line is consumed while it is hot in L1, next group prefetched under
//                        the MFMAs), the small later layers' weights are staged in LDS once per workgroup -- output unit,
//                        sigmoid and the BCE terms in the same pass.  It also stores what the backward needs: x0 [R, 2*d0],
//                        the post-ReLU activations [R, actw], prob, mask.
//   conet_bwd_kernel     same 32-row ownership, layers in reverse: gz = g (.) (act > 0) kept in LDS as the MFMA A operand,
//                        g_in = gz_own W_own + m (.) gz_other H, a wave accumulating up to four 32-column tiles at once
//                        (one A fragment read feeds 32 MFMAs); gz goes to HBM once (for the weight gradients), the input
//                        gradient [R, 2*d0] once (for the embedding update).
//   conet_wgrad_kernel   every weight gradient is gz^T x in, a contraction over the R rows: one WAVE per (pair of 32-row
//                        m tiles, 32-column n tile, row chunk) accumulating its Ws, Wt and H tiles together, both operands
//                        read coalesced straight from HBM/L2 (lane = output column, 4 rows per lane per step), bias
//                        gradients as the A operand's running row sum; partials go to a workspace and
//                        conet_wgrad_finish_kernel adds them in chunk order (plus d||H||_F) -- no float atomics anywhere:
//                        the tower backward is bit-reproducible.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "cdr_common.h"
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMaxL = CDR_CONET_MAX_LAYERS;
constexpr int kRows = 32;                       // rows per workgroup pass = one MFMA M tile
constexpr int kJobFloats = 6 * 1024 + 128;      // a wgrad job's partial: six 32x32 tiles + four bias rows
constexpr size_t kLdsBudget = 150 * 1024;
constexpr int kConetPartial = 16;               // doubles per forward block in ctx->partials (2 BCE sums + kMaxL norms); grid <= 2048
static_assert(2 + CDR_CONET_MAX_LAYERS <= kConetPartial && 2048 * kConetPartial <= CDR_MAX_PARTIAL_BLOCKS * CDR_PARTIAL_STRIDE, "partials");

struct conet_net {
    int L, vec, wlds;                           // wlds: the weights of layers >= 1 are staged in LDS
    int wl_chunks;                              // 16-byte chunks of those weights (unpadded)
    int dims[kMaxL + 1];
    int act_off[kMaxL + 1];                     // column of layer l's outputs in acts / gz; act_off[L] = row width
    int wl_off[kMaxL + 1];                      // float offset of layer l's {Ws, Wt, H} block in the LDS weight area (l >= 1)
    const unsigned long long* dma_tab;          // stage_weights_dma: source address per padded chunk of the LDS weight area (null: computed per lane)
    int hsq_chunk[kMaxL];                       // conet_fb_kernel: elements of H_l per workgroup's slice of sum H_l^2 (host: ceil(n / grid) -- a division per layer and use otherwise)
    const float* Ws[kMaxL]; const float* bs[kMaxL]; const float* Wt[kMaxL]; const float* bt[kMaxL]; const float* H[kMaxL];
    const float* wo[2]; const float* bo[2];
};
struct conet_grads {
    float* Ws[kMaxL]; float* bs[kMaxL]; float* Wt[kMaxL]; float* bt[kMaxL]; float* H[kMaxL];
    float* wo[2]; float* bo[2];
};
struct conet_tiles { int ntiles; int off[kMaxL + 1]; };

__device__ __forceinline__ float4 ldw4(const float* p, bool vec) {
    return vec ? ld4(p) : make_float4(p[0], p[1], p[2], p[3]);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
#ifdef CDR_CONET_PROF
__device__ long long* g_conet_prof = nullptr;
#define STAMP(i) do { if (g_conet_prof && blockIdx.x == 0 && threadIdx.x == 0) g_conet_prof[i] = wall_clock64(); } while (0)
// every block's entry / exit time (conet_fb_kernel): is a launch its slowest block, or its start-up and drain?  (buffer: 64 + 2 * grid words)
#define STAMPB(i) do { if (g_conet_prof && threadIdx.x == 0) g_conet_prof[64 + 2 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMPB(i) do { } while (0)
#endif
#define MFMA4(acc, a, b)                                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).x, (b).x, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).y, (b).y, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).z, (b).z, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).w, (b).w, acc, 0, 0, 0)
#define MF1(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every global STORE of
// the phase (the saved activations / gz / gx0 rows, which nothing in the kernel reads back): ~1.5 us per layer, more than
// the small layers' MFMA time.  Register results of global LOADS are still waited for by their consumers.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The {Ws, Wt, H} of every layer >= 1 into LDS, rows padded to din + 4 floats (conflict-free 16-B fragment reads).
__device__ __forceinline__ void stage_weights(const conet_net& net, float* wl) {
    const bool vec = net.vec != 0;
    for (int l = 1; l < net.L; ++l) {
        const int din = net.dims[l], dout = net.dims[l + 1], WS = din + 4, q = din >> 2;
        for (int mat = 0; mat < 3; ++mat) {
            const float* src = mat == 0 ? net.Ws[l] : mat == 1 ? net.Wt[l] : net.H[l];
            float* dst = wl + net.wl_off[l] + mat * dout * WS;
            for (int e = threadIdx.x; e < dout * q; e += 256) {
                const int row = e / q, c = e - row * q;
                st4(dst + row * WS + 4 * c, ldw4(src + (int64_t)row * din + 4 * c, vec));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ forward
// One cross unit for the workgroup's 32 rows.  Xin [32][2 din + 4] = [s | t] in LDS, Xout likewise; a wave takes
// (tower, 32-column tile) jobs and keeps two accumulators, the tower's own product and the cross product.
template <bool WL>
__device__ __forceinline__ void fwd_layer(const conet_net& net, int l, const float* __restrict__ Xin, float* __restrict__ Xout,
                                          const float* __restrict__ wl, const float* __restrict__ mrow, float* __restrict__ acts,
                                          int64_t row0, int64_t R, int wave, int li, int lh) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int XS = 2 * din + 4, XO = 2 * dout + 4;
    const int NT = (dout + 31) >> 5, KS = (din + 7) >> 3;
    const int off = net.act_off[l], actw = net.act_off[net.L];
    const bool vec = net.vec != 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int job = wave; job < 2 * NT; job += 4) {
        const int tower = job / NT, n0 = (job - tower * NT) * 32;
        const int n = n0 + li;
        const bool nv = n < dout;
        const float* xo = Xin + li * XS + (tower ? din : 0);
        const float* xc = Xin + li * XS + (tower ? 0 : din);
        f32x16 am = zero16(), ac = zero16();
        if (WL) {
            const int WS = din + 4;
            const float* wm = wl + net.wl_off[l] + (tower ? dout * WS : 0) + (nv ? n : 0) * WS;
            const float* hc = wl + net.wl_off[l] + 2 * dout * WS + (nv ? n : 0) * WS;
            for (int s = 0; s < KS; ++s) {
                const int k = 8 * s + 4 * lh;
                const bool kv = k < din;
                const float4 b0 = (kv && nv) ? ld4(wm + k) : z4, b1 = (kv && nv) ? ld4(hc + k) : z4;
                const float4 a0 = kv ? ld4(xo + k) : z4, a1 = kv ? ld4(xc + k) : z4;
                MFMA4(am, a0, b0);
                MFMA4(ac, a1, b1);
            }
        } else {
            // fragments straight from L2: a lane reads 16 B of row n per K step, i.e. a wave touches 32 lines for 32 B each;
            // asking for four K steps at once consumes every 128-B line while it is hot (one K step per request re-fetched
            // each line four times through a thrashing L1); the next group is in flight under this group's 32 MFMAs
            const float* wm = (tower ? net.Wt[l] : net.Ws[l]) + (int64_t)(nv ? n : 0) * din;
            const float* hc = net.H[l] + (int64_t)(nv ? n : 0) * din;
            const int KG = (KS + 3) >> 2;
            float4 nm[4], nc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 8 * j + 4 * lh;
                const bool ok = nv && k < din;
                nm[j] = ok ? ldw4(wm + k, vec) : z4;
                nc[j] = ok ? ldw4(hc + k, vec) : z4;
            }
            for (int g = 0; g < KG; ++g) {
                float4 cm[4], cc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { cm[j] = nm[j]; cc[j] = nc[j]; }
                if (g + 1 < KG) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 8 * (4 * (g + 1) + j) + 4 * lh;
                        const bool ok = nv && k < din;
                        nm[j] = ok ? ldw4(wm + k, vec) : z4;
                        nc[j] = ok ? ldw4(hc + k, vec) : z4;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 8 * (4 * g + j) + 4 * lh;
                    const bool kv = k < din;
                    const float4 a0 = kv ? ld4(xo + k) : z4, a1 = kv ? ld4(xc + k) : z4;
                    MFMA4(am, a0, cm[j]);
                    MFMA4(ac, a1, cc[j]);
                }
            }
        }
        if (nv) {
            const float bv = (tower ? net.bt[l] : net.bs[l])[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = am[r] + bv;
                if (mrow[row] != 0.f) v += ac[r];                      // conet.py:127-129 / :132-134
                v = v > 0.f ? v : 0.f;
                Xout[row * XO + tower * dout + n] = v;
                if (row0 + row < R) acts[(row0 + row) * actw + off + tower * dout + n] = v;
            }
        }
    }
}

// The L2-streamed layer without a predicate in its loop, for din % 128 == 0, dout % 32 == 0 and 16-B aligned weights (C3's first
// layer, 256 -> 64).  fwd_layer<false> keeps one group of fragments (4 K steps = 32 MFMAs = 0.9 us) in flight, which is less
// than the latency of these loads (a wave's request touches 32 lines of 32 B each), behind `if (k < din)` exec masks; here
// four register sets rotate through a loop unrolled by four groups, so every group is requested two groups (1.8 us) before
// its MFMAs, and the A fragments of a group are read from LDS in one batch.  Same K order as fwd_layer: bit-identical sums.
__device__ __forceinline__ void fwd_layer_stream(const conet_net& net, int l, const float* __restrict__ Xin, float* __restrict__ Xout,
                                                 const float* __restrict__ mrow, float* __restrict__ acts,
                                                 int64_t row0, int64_t R, int wave, int li, int lh) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int XS = 2 * din + 4, XO = 2 * dout + 4;
    const int NT = dout >> 5, KQ = din >> 7;
    const int off = net.act_off[l], actw = net.act_off[net.L];
    for (int job = wave; job < 2 * NT; job += 4) {
        const int tower = job / NT, n = (job - tower * NT) * 32 + li;
        const float* xo = Xin + li * XS + (tower ? din : 0) + 4 * lh;
        const float* xc = Xin + li * XS + (tower ? 0 : din) + 4 * lh;
        const float* wm = tower ? net.Wt[l] : net.Ws[l];                  // wave-uniform bases + one 32-bit lane offset: the
        const float* hc = net.H[l];                                       // 32 requests of a round share a single address VGPR
        const unsigned vo = (unsigned)(n * din + 4 * lh);
        const float bv = (tower ? net.bt[l] : net.bs[l])[n];
        f32x16 am = zero16(), ac = zero16();
        float4 m0[4], c0[4], m1[4], c1[4], m2[4], c2[4], m3[4], c3[4];
#define CDR_LOADG(M, C, KB)                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { M[j] = ld4(wm + (KB) + (vo + 8 * j)); C[j] = ld4(hc + (KB) + (vo + 8 * j)); }
#define CDR_MFG(M, C, KB) {                                                                         \
            float4 a0[4], a1[4];                                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { a0[j] = ld4(xo + (KB) + 8 * j); a1[j] = ld4(xc + (KB) + 8 * j); } \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { MFMA4(am, a0[j], M[j]); MFMA4(ac, a1[j], C[j]); } }
        CDR_LOADG(m0, c0, 0);
        CDR_LOADG(m1, c1, 32);
        __builtin_amdgcn_sched_barrier(0);
        for (int q = 0; q < KQ; ++q) {
            const int kb = q << 7;
            const int kw = q + 1 < KQ ? kb + 128 : 0;         // the last round re-requests the first groups (unused) instead of branching
            // sched_barrier: hipcc's scheduler otherwise sinks every request down to just before its MFMAs (lower register pressure,
            // no latency hidden)
            CDR_LOADG(m2, c2, kb + 64); __builtin_amdgcn_sched_barrier(0);
            CDR_MFG(m0, c0, kb);        __builtin_amdgcn_sched_barrier(0);
            CDR_LOADG(m3, c3, kb + 96); __builtin_amdgcn_sched_barrier(0);
            CDR_MFG(m1, c1, kb + 32);   __builtin_amdgcn_sched_barrier(0);
            CDR_LOADG(m0, c0, kw);      __builtin_amdgcn_sched_barrier(0);
            CDR_MFG(m2, c2, kb + 64);   __builtin_amdgcn_sched_barrier(0);
            CDR_LOADG(m1, c1, kw + 32); __builtin_amdgcn_sched_barrier(0);
            CDR_MFG(m3, c3, kb + 96);   __builtin_amdgcn_sched_barrier(0);
        }
#undef CDR_LOADG
#undef CDR_MFG
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = am[r] + bv;
            if (mrow[row] != 0.f) v += ac[r];                          // conet.py:127-129 / :132-134
            v = v > 0.f ? v : 0.f;
            Xout[row * XO + tower * dout + n] = v;
            if (row0 + row < R) acts[(row0 + row) * actw + off + tower * dout + n] = v;
        }
    }
}

// The small layers on LDS-staged weights without a predicate in the K loop, for din % 8 == 0: a column past dout works on weight
// row 0 and is dropped at the store (the predicated fwd_layer<true> compiled to an exec-masked block per operand: 4.6 / 3.4 /
// 2.8 us for C3's 64->32->16->8 where the MFMAs take 1.7 / 0.9 / 0.4).  Same K order: bit-identical.
__device__ __forceinline__ void fwd_layer_lds(const conet_net& net, int l, const float* __restrict__ Xin, float* __restrict__ Xout,
                                              const float* __restrict__ wl, const float* __restrict__ mrow, float* __restrict__ acts,
                                              int64_t row0, int64_t R, int wave, int li, int lh) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int XS = 2 * din + 4, XO = 2 * dout + 4, WS = din + 4;
    const int NT = (dout + 31) >> 5, KS = din >> 3;
    const int off = net.act_off[l], actw = net.act_off[net.L];
    for (int job = wave; job < 2 * NT; job += 4) {
        const int tower = job / NT, n = (job - tower * NT) * 32 + li;
        const bool nv = n < dout;
        const int nc = nv ? n : 0;
        const float* xo = Xin + li * XS + (tower ? din : 0) + 4 * lh;
        const float* xc = Xin + li * XS + (tower ? 0 : din) + 4 * lh;
        const float* wm = wl + net.wl_off[l] + (tower ? dout * WS : 0) + nc * WS + 4 * lh;
        const float* hc = wl + net.wl_off[l] + 2 * dout * WS + nc * WS + 4 * lh;
        const float bv = (tower ? net.bt[l] : net.bs[l])[nc];      // requested before the MFMAs, used after them
        f32x16 am = zero16(), ac = zero16();
#pragma unroll 2
        for (int s = 0; s < KS; ++s) {
            const float4 b0 = ld4(wm + 8 * s), b1 = ld4(hc + 8 * s), a0 = ld4(xo + 8 * s), a1 = ld4(xc + 8 * s);
            MFMA4(am, a0, b0);
            MFMA4(ac, a1, b1);
        }
        if (nv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = am[r] + bv;
                if (mrow[row] != 0.f) v += ac[r];                      // conet.py:127-129 / :132-134
                v = v > 0.f ? v : 0.f;
                Xout[row * XO + tower * dout + n] = v;
                if (row0 + row < R) acts[(row0 + row) * actw + off + tower * dout + n] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void conet_fwd_kernel(conet_net net, const float* __restrict__ su, const float* __restrict__ si,
                                                        const float* __restrict__ tu, const float* __restrict__ ti, int D,
                                                        const int64_t* __restrict__ user_s, const int64_t* __restrict__ user_t,
                                                        const int64_t* __restrict__ item_s, const int64_t* __restrict__ item_t,
                                                        int64_t R, int64_t n_source, int64_t n_overlap, int overlap_users,
                                                        const float* __restrict__ label_s, const float* __restrict__ label_t,
                                                        float* __restrict__ label_cat, int64_t* __restrict__ ids_cat,
                                                        int strideA, int strideB,
                                                        float* __restrict__ x0, float* __restrict__ acts, float* __restrict__ prob,
                                                        float* __restrict__ maskf, double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float mrow[kRows];
    __shared__ double red[(2 + kMaxL) * 4];
    float* bufA = smem;                                  // inputs of the even layers
    float* bufB = smem + kRows * strideA;                // inputs of the odd layers
    float* wl = bufB + kRows * strideB;                  // weights of layers >= 1 (when net.wlds)
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int D4 = D >> 2;
    double lacc0 = 0.0, lacc1 = 0.0;
    if (net.wlds) stage_weights(net, wl);                // visible after the first gather's barrier
    const int64_t nrb = (R + kRows - 1) / kRows;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        STAMP(0);
        {   // ---- gather [su | si | tu | ti] of 32 rows: 8 threads per row, 16 B each, 128 B contiguous per octet
            const int row = t >> 3, c0 = t & 7;
            const int64_t g = rb * kRows + row;
            const bool valid = g < R;
            const int64_t gc = valid ? g : R - 1;
            // the source batch and the target batch arrive as separate tensors (conet.py:184-191): stacked here, not by a cat
            const bool src = gc < n_source;
            const int64_t uid = src ? user_s[gc] : user_t[gc - n_source], iid = src ? item_s[gc] : item_t[gc - n_source];
            float* xr = bufA + row * (4 * D + 4);
            for (int c = c0; c < D4; c += 8) {
                const float4 a = ld4(su + uid * D + 4 * c), b = ld4(si + iid * D + 4 * c);
                const float4 e = ld4(tu + uid * D + 4 * c), f = ld4(ti + iid * D + 4 * c);
                st4(xr + 4 * c, a); st4(xr + D + 4 * c, b); st4(xr + 2 * D + 4 * c, e); st4(xr + 3 * D + 4 * c, f);
                if (valid) {
                    float* xg = x0 + g * (4 * (int64_t)D);
                    st4(xg + 4 * c, a); st4(xg + D + 4 * c, b); st4(xg + 2 * D + 4 * c, e); st4(xg + 3 * D + 4 * c, f);
                }
            }
            if (c0 == 0) {
                const float m = ((overlap_users ? uid : iid) < n_overlap) ? 1.f : 0.f;     // PAD id 0 counts
                mrow[row] = m;
                if (valid) {
                    maskf[g] = m;
                    ids_cat[g] = uid; ids_cat[R + g] = iid;               // for the embedding update (scatter / row-wise sort)
                }
            }
        }
        lds_barrier();
        STAMP(1);
        for (int l = 0; l < net.L; ++l) {
            const float* Xin = (l & 1) ? bufB : bufA;
            float* Xout = (l & 1) ? bufA : bufB;
            if (l > 0 && net.wlds && !(net.dims[l] & 7)) fwd_layer_lds(net, l, Xin, Xout, wl, mrow, acts, rb * kRows, R, wave, li, lh);
            else if (l > 0 && net.wlds) fwd_layer<true>(net, l, Xin, Xout, wl, mrow, acts, rb * kRows, R, wave, li, lh);
            else if (net.vec && !(net.dims[l] & 127) && !(net.dims[l + 1] & 31))
                fwd_layer_stream(net, l, Xin, Xout, mrow, acts, rb * kRows, R, wave, li, lh);
            else fwd_layer<false>(net, l, Xin, Xout, wl, mrow, acts, rb * kRows, R, wave, li, lh);
            lds_barrier();
            STAMP(2 + l);
        }
        if (t < kRows) {   // ---- output unit + sigmoid + BCE term of the tower this row belongs to (conet.py:140,179,195-196)
            const int64_t g = rb * kRows + t;
            if (g < R) {
                const int tower = g >= n_source ? 1 : 0;
                const int dL = net.dims[net.L];
                const float* h = ((net.L & 1) ? bufB : bufA) + t * (2 * dL + 4) + tower * dL;
                const float* w = net.wo[tower];
                float z = 0.f;
                for (int j = 0; j < dL; ++j) z += h[j] * w[j];
                z += net.bo[tower][0];
                const float p = 1.0f / (1.0f + expf(-z));
                prob[g] = p;
                const float y = tower ? label_t[g - n_source] : label_s[g];
                label_cat[g] = y;
                const double term = (double)((y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f));
                if (tower) lacc1 += term; else lacc0 += term;
            }
        }
        lds_barrier();
        STAMP(2 + net.L);
    }
    // this block's slice of sum H_l^2 (conet.py:198-201), so that the finishing block adds gridDim.x numbers per layer instead of
    // walking every H with 256 threads (a chain of dependent cache misses: 9 us of a 185 us step)
    double lacc[2 + kMaxL];
    lacc[0] = lacc0; lacc[1] = lacc1;
#pragma unroll
    for (int l = 0; l < kMaxL; ++l) {
        double q = 0.0;
        if (l < net.L) {
            const int n = net.dims[l] * net.dims[l + 1];
            const int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x;
            const int lo = (int)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
            const float* h = net.H[l];
            for (int e = lo + t; e < hi; e += 256) q += (double)h[e] * (double)h[e];
        }
        lacc[2 + l] = q;
    }
    block_sum_d<2 + kMaxL>(lacc, red);
    if (t == 0) {
        double* o = partials + (size_t)blockIdx.x * kConetPartial;
#pragma unroll
        for (int i = 0; i < 2 + kMaxL; ++i) o[i] = lacc[i];
    }
}

// out = {total, bce_source, bce_target, reg, ||H_0||_F .. ||H_{L-1}||_F}: one workgroup adds the forward blocks' partials
__device__ __forceinline__ void conet_finish_block(const conet_net& net, const double* __restrict__ partials, int nblocks,
                                                   int64_t n_source, int64_t R, float* __restrict__ out, double* red) {
    double acc[2 + kMaxL];
#pragma unroll
    for (int i = 0; i < 2 + kMaxL; ++i) acc[i] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {        // per block: the two BCE sums and its slice of every sum H_l^2, all fp64
        const double* o = partials + (size_t)b * kConetPartial;
#pragma unroll
        for (int i = 0; i < 2 + kMaxL; ++i) acc[i] += o[i];
    }
    block_sum_d<2 + kMaxL>(acc, red);
    if (threadIdx.x == 0) {
        float reg = 0.f;
        for (int l = 0; l < net.L; ++l) {                     // conet.py:198-201: un-weighted sum of ||H_l||_F
            const float nv = (float)sqrt(acc[2 + l]);
            out[4 + l] = nv;
            reg += nv;
        }
        const float ls = (float)(acc[0] / (double)n_source), lt = (float)(acc[1] / (double)(R - n_source));
        out[1] = ls; out[2] = lt; out[3] = reg;
        out[0] = (ls + lt) + reg;
    }
}

__global__ __launch_bounds__(256) void conet_fwd_finish_kernel(conet_net net, const double* __restrict__ partials, int nblocks,
                                                               int64_t n_source, int64_t R, float* __restrict__ out) {
    __shared__ double red[(2 + kMaxL) * 4];
    conet_finish_block(net, partials, nblocks, n_source, R, out, red);
}

// ------------------------------------------------------------------------------------------------------------ backward (data)
// g_in of one cross unit: Gl [32][2 dout + 4] = gz of both towers in LDS.  A wave takes (tower, group of <= 4 column tiles)
// units: the two A fragments of a K step feed 8 accumulators; the B values (W[k][n], lane = n: coalesced rows) of the
// next K step are in flight under the current step's MFMAs.
template <bool WL>
__device__ __forceinline__ void bwd_layer(const conet_net& net, int l, const float* __restrict__ Gl, float* __restrict__ Gn,
                                          const float* __restrict__ wl, const float* __restrict__ mrow, float* __restrict__ gx0,
                                          int64_t row0, int64_t R, int wave, int li, int lh) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int GS = 2 * dout + 4, GN = 2 * din + 4;
    const int NT = (din + 31) >> 5, NG = (NT + 3) >> 2, KS = (dout + 7) >> 3;
    const int ldw = WL ? din + 4 : din;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u = wave; u < 2 * NG; u += 4) {
        const int tower = u / NG, t0 = (u - tower * NG) * 4;
        const float* Wm = WL ? wl + net.wl_off[l] + (tower ? dout * ldw : 0) : (tower ? net.Wt[l] : net.Ws[l]);
        const float* Hm = WL ? wl + net.wl_off[l] + 2 * dout * ldw : net.H[l];
        const float* ao = Gl + li * GS + (tower ? dout : 0);
        const float* ax = Gl + li * GS + (tower ? 0 : dout);
        int ncol[4]; bool nv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ncol[j] = (t0 + j) * 32 + li; nv[j] = ncol[j] < din; }
        f32x16 am[4], ac[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { am[j] = zero16(); ac[j] = zero16(); }
        float4 nm[4], nc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            nm[j] = nc[j] = z4;
            const int k = 4 * lh;
            if (nv[j] && k < dout) {                          // dout % 4 == 0: a chunk of 4 k is all in or all out
                const float* w0 = Wm + k * ldw + ncol[j];
                const float* h0 = Hm + k * ldw + ncol[j];
                nm[j] = make_float4(w0[0], w0[ldw], w0[2 * ldw], w0[3 * ldw]);
                nc[j] = make_float4(h0[0], h0[ldw], h0[2 * ldw], h0[3 * ldw]);
            }
        }
        for (int s = 0; s < KS; ++s) {
            const int k = 8 * s + 4 * lh;
            const bool kv = k < dout;
            float4 cm[4], cc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { cm[j] = nm[j]; cc[j] = nc[j]; }
            const int kn = k + 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nm[j] = nc[j] = z4;
                if (nv[j] && kn < dout) {
                    const float* w0 = Wm + kn * ldw + ncol[j];
                    const float* h0 = Hm + kn * ldw + ncol[j];
                    nm[j] = make_float4(w0[0], w0[ldw], w0[2 * ldw], w0[3 * ldw]);
                    nc[j] = make_float4(h0[0], h0[ldw], h0[2 * ldw], h0[3 * ldw]);
                }
            }
            const float4 a0 = kv ? ld4(ao + k) : z4, a1 = kv ? ld4(ax + k) : z4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (t0 + j < NT) {                            // wave-uniform
                    MFMA4(am[j], a0, cm[j]);
                    MFMA4(ac[j], a1, cc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!nv[j]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = am[j][r];
                if (mrow[row] != 0.f) v += ac[j][r];
                if (l > 0) Gn[row * GN + tower * din + ncol[j]] = v;
                else if (row0 + row < R) gx0[(row0 + row) * (2 * (int64_t)din) + tower * din + ncol[j]] = v;
            }
        }
    }
}

// bwd_layer without a predicate in its loop, for dout % 32 == 0 and din % (32 NJ) == 0: a wave owns NJ whole column tiles, the
// B values rotate through four register sets in a loop unrolled by four K steps (each set is requested two steps before its
// MFMAs; bwd_layer's exec-masked loads and wave-uniform branches made hipcc wait for every step's loads right before their use
// and shuttle the accumulators between VGPRs and AGPRs around every 8 MFMAs: 22 us for C3's first layer, 6.5 us for its
// second).  Used with NJ = 1 for din % 32 == 0; din % 128 == 0 takes bwd_layer_quad.  Same K order as bwd_layer:
// bit-identical sums.
template <bool WL, int NJ>
__device__ __forceinline__ void bwd_layer_tiles(const conet_net& net, int l, const float* __restrict__ Gl, float* __restrict__ Gn,
                                                const float* __restrict__ wl, const float* __restrict__ mrow, float* __restrict__ gx0,
                                                int64_t row0, int64_t R, int wave, int li, int lh) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int GS = 2 * dout + 4, GN = 2 * din + 4;
    const int NG = (din >> 5) / NJ, KQ = dout >> 5;
    const int ldw = WL ? din + 4 : din;
    for (int u = wave; u < 2 * NG; u += 4) {
        const int tower = u / NG, c0 = (u - tower * NG) * NJ * 32 + li;
        const float* Wm = WL ? wl + net.wl_off[l] + (tower ? dout * ldw : 0) : (tower ? net.Wt[l] : net.Ws[l]);   // wave-uniform
        const float* Hm = WL ? wl + net.wl_off[l] + 2 * dout * ldw : net.H[l];
        const unsigned vo = (unsigned)(4 * lh * ldw + c0);                // the one lane-dependent part of every B address
        const float* ao = Gl + li * GS + (tower ? dout : 0) + 4 * lh;
        const float* ax = Gl + li * GS + (tower ? 0 : dout) + 4 * lh;
        f32x16 am[NJ], ac[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { am[j] = zero16(); ac[j] = zero16(); }
        float4 m0[NJ], h0[NJ], m1[NJ], h1[NJ], m2[NJ], h2[NJ], m3[NJ], h3[NJ];
#define CDR_LOADS(M, Hh, K)                                                                         \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                            \
            const float* w_ = Wm + (K) * ldw;                                                       \
            const float* h_ = Hm + (K) * ldw;                                                       \
            const unsigned o_ = vo + 32 * j;                                                        \
            M[j] = make_float4(w_[o_], (w_ + ldw)[o_], (w_ + 2 * ldw)[o_], (w_ + 3 * ldw)[o_]);     \
            Hh[j] = make_float4(h_[o_], (h_ + ldw)[o_], (h_ + 2 * ldw)[o_], (h_ + 3 * ldw)[o_]); }
#define CDR_MFS(M, Hh, K) {                                                                         \
            const float4 a0 = ld4(ao + (K)), a1 = ld4(ax + (K));                                    \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) { MFMA4(am[j], a0, M[j]); MFMA4(ac[j], a1, Hh[j]); } }
        CDR_LOADS(m0, h0, 0);
        CDR_LOADS(m1, h1, 8);
        __builtin_amdgcn_sched_barrier(0);
        for (int q = 0; q < KQ; ++q) {
            const int kb = q << 5;
            const int kw = q + 1 < KQ ? kb + 32 : 0;
            CDR_LOADS(m2, h2, kb + 16); __builtin_amdgcn_sched_barrier(0);       // see fwd_layer_stream
            CDR_MFS(m0, h0, kb);        __builtin_amdgcn_sched_barrier(0);
            CDR_LOADS(m3, h3, kb + 24); __builtin_amdgcn_sched_barrier(0);
            CDR_MFS(m1, h1, kb + 8);    __builtin_amdgcn_sched_barrier(0);
            CDR_LOADS(m0, h0, kw);      __builtin_amdgcn_sched_barrier(0);
            CDR_MFS(m2, h2, kb + 16);   __builtin_amdgcn_sched_barrier(0);
            CDR_LOADS(m1, h1, kw + 8);  __builtin_amdgcn_sched_barrier(0);
            CDR_MFS(m3, h3, kb + 24);   __builtin_amdgcn_sched_barrier(0);
        }
#undef CDR_LOADS
#undef CDR_MFS
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = am[j][r];
                if (mrow[row] != 0.f) v += ac[j][r];
                if (l > 0) Gn[row * GN + tower * din + c0 + 32 * j] = v;
                else if (row0 + row < R) gx0[(row0 + row) * (2 * (int64_t)din) + tower * din + c0 + 32 * j] = v;
            }
        }
    }
}

// The 128-column form of bwd_layer_tiles: the wave's four accumulator tiles take columns cb + 4 li + j (j = tile) instead of
// cb + 32 j + li, so the four B values a lane needs from one row of W are 16 contiguous bytes -- 8 dwordx4 requests per K
// step instead of 32 dword ones (a returning request costs the CU's vector-memory path about as much either way, and with
// dwords that path, not the MFMAs, set the pace) -- and its outputs go out as one 16-B store per row.  Which columns share
// a tile does not enter any sum: bit-identical to bwd_layer.
__device__ __forceinline__ float comp4(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
template <bool WL>
__device__ __forceinline__ void bwd_layer_quad(const conet_net& net, int l, const float* __restrict__ Gl, float* __restrict__ Gn,
                                               const float* __restrict__ wl, const float* __restrict__ mrow, float* __restrict__ gx0,
                                               int64_t row0, int64_t R, int wave, int li, int lh) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int GS = 2 * dout + 4, GN = 2 * din + 4;
    const int NG = din >> 7, KQ = dout >> 5;
    const int ldw = WL ? din + 4 : din;
    for (int u = wave; u < 2 * NG; u += 4) {
        const int tower = u / NG, c0 = (u - tower * NG) * 128 + 4 * li;
        const float* Wm = WL ? wl + net.wl_off[l] + (tower ? dout * ldw : 0) : (tower ? net.Wt[l] : net.Ws[l]);   // wave-uniform
        const float* Hm = WL ? wl + net.wl_off[l] + 2 * dout * ldw : net.H[l];
        const unsigned vo = (unsigned)(4 * lh * ldw + c0);
        const float* ao = Gl + li * GS + (tower ? dout : 0) + 4 * lh;
        const float* ax = Gl + li * GS + (tower ? 0 : dout) + 4 * lh;
        f32x16 am[4], ac[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { am[j] = zero16(); ac[j] = zero16(); }
        float4 m0[4], h0[4], m1[4], h1[4], m2[4], h2[4], m3[4], h3[4];       // [r] = W[k + r][c0 .. c0 + 3]
#define CDR_LOADS(M, Hh, K)                                                                         \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                             \
            M[r] = ld4(Wm + ((K) + r) * ldw + vo);                                                  \
            Hh[r] = ld4(Hm + ((K) + r) * ldw + vo); }
#define CDR_MFS(M, Hh, K) {                                                                         \
            const float4 a0 = ld4(ao + (K)), a1 = ld4(ax + (K));                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
                MF1(am[j], a0.x, comp4(M[0], j)); MF1(am[j], a0.y, comp4(M[1], j));                 \
                MF1(am[j], a0.z, comp4(M[2], j)); MF1(am[j], a0.w, comp4(M[3], j));                 \
                MF1(ac[j], a1.x, comp4(Hh[0], j)); MF1(ac[j], a1.y, comp4(Hh[1], j));               \
                MF1(ac[j], a1.z, comp4(Hh[2], j)); MF1(ac[j], a1.w, comp4(Hh[3], j)); } }
        CDR_LOADS(m0, h0, 0);
        CDR_LOADS(m1, h1, 8);
        __builtin_amdgcn_sched_barrier(0);
        for (int q = 0; q < KQ; ++q) {
            const int kb = q << 5;
            const int kw = q + 1 < KQ ? kb + 32 : 0;
            CDR_LOADS(m2, h2, kb + 16); __builtin_amdgcn_sched_barrier(0);       // see fwd_layer_stream
            CDR_MFS(m0, h0, kb);        __builtin_amdgcn_sched_barrier(0);
            CDR_LOADS(m3, h3, kb + 24); __builtin_amdgcn_sched_barrier(0);
            CDR_MFS(m1, h1, kb + 8);    __builtin_amdgcn_sched_barrier(0);
            CDR_LOADS(m0, h0, kw);      __builtin_amdgcn_sched_barrier(0);
            CDR_MFS(m2, h2, kb + 16);   __builtin_amdgcn_sched_barrier(0);
            CDR_LOADS(m1, h1, kw + 8);  __builtin_amdgcn_sched_barrier(0);
            CDR_MFS(m3, h3, kb + 24);   __builtin_amdgcn_sched_barrier(0);
        }
#undef CDR_LOADS
#undef CDR_MFS
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool cross = mrow[row] != 0.f;
            const float4 v = make_float4(cross ? am[0][r] + ac[0][r] : am[0][r], cross ? am[1][r] + ac[1][r] : am[1][r],
                                         cross ? am[2][r] + ac[2][r] : am[2][r], cross ? am[3][r] + ac[3][r] : am[3][r]);
            if (l > 0) st4(Gn + row * GN + tower * din + c0, v);
            else if (row0 + row < R) st4(gx0 + (row0 + row) * (2 * (int64_t)din) + tower * din + c0, v);
        }
    }
}

// bwd_layer<true> for the small layers (dout % 8 == 0, any din): one column tile per unit, no predicate in the K loop (a column past
// din reads weight column 0 and is dropped at the store).  Same K order: bit-identical.
__device__ __forceinline__ void bwd_layer_lds(const conet_net& net, int l, const float* __restrict__ Gl, float* __restrict__ Gn,
                                              const float* __restrict__ wl, const float* __restrict__ mrow, float* __restrict__ gx0,
                                              int64_t row0, int64_t R, int wave, int li, int lh) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int GS = 2 * dout + 4, GN = 2 * din + 4, ldw = din + 4;
    const int NT = (din + 31) >> 5, KS = dout >> 3;
    for (int u = wave; u < 2 * NT; u += 4) {
        const int tower = u / NT, c0 = (u - tower * NT) * 32 + li;
        const bool cv = c0 < din;
        const int cc = cv ? c0 : 0;
        const float* Wm = wl + net.wl_off[l] + (tower ? dout * ldw : 0) + 4 * lh * ldw + cc;
        const float* Hm = wl + net.wl_off[l] + 2 * dout * ldw + 4 * lh * ldw + cc;
        const float* ao = Gl + li * GS + (tower ? dout : 0) + 4 * lh;
        const float* ax = Gl + li * GS + (tower ? 0 : dout) + 4 * lh;
        f32x16 am = zero16(), ac = zero16();
#pragma unroll 2
        for (int s = 0; s < KS; ++s) {
            const float* w_ = Wm + 8 * s * ldw;
            const float* h_ = Hm + 8 * s * ldw;
            const float4 m = make_float4(w_[0], w_[ldw], w_[2 * ldw], w_[3 * ldw]);
            const float4 h = make_float4(h_[0], h_[ldw], h_[2 * ldw], h_[3 * ldw]);
            const float4 a0 = ld4(ao + 8 * s), a1 = ld4(ax + 8 * s);
            MFMA4(am, a0, m);
            MFMA4(ac, a1, h);
        }
        if (cv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = am[r];
                if (mrow[row] != 0.f) v += ac[r];
                if (l > 0) Gn[row * GN + tower * din + c0] = v;
                else if (row0 + row < R) gx0[(row0 + row) * (2 * (int64_t)din) + tower * din + c0] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void conet_bwd_kernel(conet_net net, int64_t R, int64_t n_source, const float* __restrict__ label,
                                                        const float* __restrict__ prob, const float* __restrict__ maskf,
                                                        const float* __restrict__ acts, const float* __restrict__ grad_out,
                                                        int strideG0, int strideG1, float* __restrict__ gz, float* __restrict__ gx0,
                                                        float* __restrict__ ou_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float mrow[kRows], dzrow[kRows];
    __shared__ int trow[kRows];
    float* G0 = smem;                                    // output gradients of the even layers
    float* G1 = smem + kRows * strideG0;                 // ... of the odd layers
    float* hl = G1 + kRows * strideG1;                   // last layer's activations of the 32 rows [32][2 dL]
    float* wl = hl + kRows * 2 * net.dims[net.L];        // weights of layers >= 1 (when net.wlds)
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int L = net.L, actw = net.act_off[L], dL = net.dims[L];
    const float go = grad_out ? grad_out[0] : 1.0f;
    float ou_acc = 0.f;
    if (net.wlds) stage_weights(net, wl);
    const int64_t nrb = (R + kRows - 1) / kRows;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        STAMP(16);
        if (t < kRows) {
            const int64_t g = rb * kRows + t;
            float dz = 0.f, m = 0.f;
            int tower = 0;
            if (g < R) {
                tower = g >= n_source ? 1 : 0;
                const float p = prob[g], y = label[g];
                const float nd = (float)(tower ? R - n_source : n_source);
                const float gp = (go / nd) * (p - y) / fmaxf((1.0f - p) * p, 1e-12f);     // BCELoss backward (mean)
                dz = (gp * (1.0f - p)) * p;                                                // sigmoid backward
                m = maskf[g];
            }
            dzrow[t] = dz; mrow[t] = m; trow[t] = tower;
        }
        for (int e = t; e < kRows * 2 * dL; e += 256) {      // the output units' inputs, for their weight gradients
            const int row = e / (2 * dL), c = e - row * 2 * dL;
            const int64_t g = rb * kRows + row;
            hl[e] = g < R ? acts[g * actw + net.act_off[L - 1] + c] : 0.f;
        }
        lds_barrier();
        STAMP(17);
        {
            float* Gl = ((L - 1) & 1) ? G1 : G0;
            const int GS = 2 * dL + 4;
            for (int e = t; e < kRows * 2 * dL; e += 256) {
                const int row = e / (2 * dL), c = e - row * 2 * dL;
                const int tower = c >= dL ? 1 : 0, j = c - tower * dL;
                Gl[row * GS + c] = (trow[row] == tower) ? dzrow[row] * net.wo[tower][j] : 0.f;
            }
            if (t < 2 * (dL + 1)) {                          // output-unit gradients of this block's rows, in row order
                const int tower = t / (dL + 1), jj = t - tower * (dL + 1);
                for (int row = 0; row < kRows; ++row)
                    if (trow[row] == tower) ou_acc += dzrow[row] * (jj < dL ? hl[row * 2 * dL + tower * dL + jj] : 1.0f);
            }
        }
        lds_barrier();
        STAMP(18);
        for (int l = L - 1; l >= 0; --l) {
            const int dout = net.dims[l + 1];
            const int GS = 2 * dout + 4;
            float* Gl = (l & 1) ? G1 : G0;
            float* Gn = (l & 1) ? G0 : G1;
            const int off = net.act_off[l];
            const int q = (2 * dout) >> 2;                   // float4 per row: widths are multiples of 4, rows 16-B aligned
            for (int e = t; e < kRows * q; e += 256) {       // ReLU backward; gz kept for the weight gradients
                const int row = e / q, c = 4 * (e - row * q);
                const int64_t g = rb * kRows + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g < R) {
                    const float4 a = ld4(acts + g * actw + off + c), gg = ld4(Gl + row * GS + c);
                    v = make_float4(a.x > 0.f ? gg.x : 0.f, a.y > 0.f ? gg.y : 0.f, a.z > 0.f ? gg.z : 0.f, a.w > 0.f ? gg.w : 0.f);
                    st4(gz + g * actw + off + c, v);
                }
                st4(Gl + row * GS + c, v);
            }
            lds_barrier();
            STAMP(19 + 2 * (L - 1 - l));
            const int din = net.dims[l];
            const bool lds_w = l > 0 && net.wlds;
            if (!(dout & 31) && !(din & 127)) {              // whole tiles: the predicate-free loops
                if (lds_w) bwd_layer_quad<true>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
                else bwd_layer_quad<false>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            } else if (!(dout & 31) && !(din & 31)) {
                if (lds_w) bwd_layer_tiles<true, 1>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
                else bwd_layer_tiles<false, 1>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            } else if (lds_w && !(dout & 7)) bwd_layer_lds(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            else if (lds_w) bwd_layer<true>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            else bwd_layer<false>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            lds_barrier();
            STAMP(20 + 2 * (L - 1 - l));
        }
    }
    if (t < 2 * (dL + 1)) ou_part[(size_t)blockIdx.x * 2 * (dL + 1) + t] = ou_acc;
}

// ------------------------------------------------------------------------------------------------------------ eight-wave cross units
// conet_fb_kernel<8>: the same 32-row block on EIGHT waves (two per SIMD).  A cross unit is two independent products per (tower,
// column tile) -- the tower's own weights and the shared H -- which the four-wave functions above accumulate side by side in one
// wave (am, ac).  Here each product is a job of its own: "own" waves (which = 0) and "cross" waves (which = 1) run them concurrently,
// the cross wave parks its raw accumulator where the result belongs (the layer's output region in LDS; a scratch region for the
// layer-0 input gradient, which leaves for HBM), and behind one LDS barrier the own wave finishes exactly as before:
// v = am + b; if (mask) v += ac.  Each accumulator sees the same operands in the same K order, the sum am + ac is formed by the same
// instruction: bit-identical to the four-wave kernel.  What it buys: C3's 8,190 rows are one block per CU, so a launch is one block's
// chain of LDS / L2 round trips in front of its MFMAs; with a second wave on every SIMD one wave's waits sit behind the other's
// MFMAs, the small layers (two jobs for four waves before) occupy four SIMDs, and a block without an overlapped row (`anym` false)
// skips the cross product altogether.
#ifndef CDR_FB8_GJ
#define CDR_FB8_GJ 4
#endif
struct split_job { bool active; int which, u; };
__device__ __forceinline__ split_job split_pick(int wave, int u0, int NU) {
    const int rem = NU - u0;
    const bool wide = rem > 2;                   // >= 3 units left: own on waves 0-3, cross on waves 4-7 (one of each per SIMD)
    split_job j;
    j.which = wide ? wave >> 2 : (wave >> 1) & 1;   // <= 2 units: own on waves 0-1, cross on waves 2-3 (a SIMD each)
    const int uu = wide ? wave & 3 : wave & 1;
    j.active = (wide || wave < 4) && uu < rem;
    j.u = j.active ? u0 + uu : 0;
    return j;
}

// layer 0 of the forward (din % 128 == 0, dout % 32 == 0, weights streamed from L2): fwd_layer_stream, one product per wave
__device__ __forceinline__ void fwd_layer_stream8(const conet_net& net, int l, const float* __restrict__ Xin, float* __restrict__ Xout,
                                                  const float* __restrict__ mrow, float* __restrict__ acts,
                                                  int64_t row0, int64_t R, int wave, int li, int lh, bool anym) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int XS = 2 * din + 4, XO = 2 * dout + 4;
    constexpr int GJ = CDR_FB8_GJ;                    // K steps (of 8) per register set
    const int NT = dout >> 5, KQ = din / (32 * GJ), NU = 2 * NT;
    const int off = net.act_off[l], actw = net.act_off[net.L];
    for (int u0 = 0; u0 < NU; u0 += 4) {
        const split_job sj = split_pick(wave, u0, NU);
        const int tower = sj.u / NT, n = (sj.u - tower * NT) * 32 + li;
        const bool work = sj.active && (!sj.which || anym);
        f32x16 acc = zero16();
        if (work) {
            const float* xa = Xin + li * XS + ((tower ^ sj.which) ? din : 0) + 4 * lh;
            const float* wb = sj.which ? net.H[l] : (tower ? net.Wt[l] : net.Ws[l]);
            const unsigned vo = (unsigned)(n * din + 4 * lh);
            float4 m0[GJ], m1[GJ], m2[GJ], m3[GJ];
#define CDR_LOADG(M, KB) _Pragma("unroll") for (int j = 0; j < GJ; ++j) { M[j] = ld4(wb + (KB) + (vo + 8 * j)); }
#define CDR_MFG(M, KB) {                                                                            \
            float4 a0[GJ];                                                                          \
            _Pragma("unroll") for (int j = 0; j < GJ; ++j) { a0[j] = ld4(xa + (KB) + 8 * j); }      \
            _Pragma("unroll") for (int j = 0; j < GJ; ++j) { MFMA4(acc, a0[j], M[j]); } }
            CDR_LOADG(m0, 0);
            CDR_LOADG(m1, 8 * GJ);
            __builtin_amdgcn_sched_barrier(0);
            for (int q = 0; q < KQ; ++q) {
                const int kb = q * 32 * GJ;
                const int kw = q + 1 < KQ ? kb + 32 * GJ : 0;
                CDR_LOADG(m2, kb + 16 * GJ); __builtin_amdgcn_sched_barrier(0);       // see fwd_layer_stream
                CDR_MFG(m0, kb);             __builtin_amdgcn_sched_barrier(0);
                CDR_LOADG(m3, kb + 24 * GJ); __builtin_amdgcn_sched_barrier(0);
                CDR_MFG(m1, kb + 8 * GJ);    __builtin_amdgcn_sched_barrier(0);
                CDR_LOADG(m0, kw);           __builtin_amdgcn_sched_barrier(0);
                CDR_MFG(m2, kb + 16 * GJ);   __builtin_amdgcn_sched_barrier(0);
                CDR_LOADG(m1, kw + 8 * GJ);  __builtin_amdgcn_sched_barrier(0);
                CDR_MFG(m3, kb + 24 * GJ);   __builtin_amdgcn_sched_barrier(0);
            }
#undef CDR_LOADG
#undef CDR_MFG
            if (sj.which) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Xout[((r & 3) + 8 * (r >> 2) + 4 * lh) * XO + tower * dout + n] = acc[r];
            }
        }
        lds_barrier();
        if (sj.active && !sj.which) {
            const float bv = (tower ? net.bt[l] : net.bs[l])[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[r] + bv;
                if (mrow[row] != 0.f) v += Xout[row * XO + tower * dout + n];      // conet.py:127-129 / :132-134
                v = v > 0.f ? v : 0.f;
                Xout[row * XO + tower * dout + n] = v;
                if (row0 + row < R) acts[(row0 + row) * actw + off + tower * dout + n] = v;
            }
        }
    }
}

// the small layers of the forward on LDS-staged weights (din % 8 == 0): fwd_layer_lds, one product per wave
__device__ __forceinline__ void fwd_layer_lds8(const conet_net& net, int l, const float* __restrict__ Xin, float* __restrict__ Xout,
                                               const float* __restrict__ wl, const float* __restrict__ mrow, float* __restrict__ acts,
                                               int64_t row0, int64_t R, int wave, int li, int lh, bool anym) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int XS = 2 * din + 4, XO = 2 * dout + 4, WS = din + 4;
    const int NT = (dout + 31) >> 5, KS = din >> 3, NU = 2 * NT;
    const int off = net.act_off[l], actw = net.act_off[net.L];
    for (int u0 = 0; u0 < NU; u0 += 4) {
        const split_job sj = split_pick(wave, u0, NU);
        const int tower = sj.u / NT, n = (sj.u - tower * NT) * 32 + li;
        const bool nv = n < dout;
        const int nc = nv ? n : 0;
        const bool work = sj.active && (!sj.which || anym);
        f32x16 acc = zero16();
        float bv = 0.f;
        if (work) {
            const float* xa = Xin + li * XS + ((tower ^ sj.which) ? din : 0) + 4 * lh;
            const float* wb = wl + net.wl_off[l] + (sj.which ? 2 * dout * WS : (tower ? dout * WS : 0)) + nc * WS + 4 * lh;
            if (!sj.which) bv = (tower ? net.bt[l] : net.bs[l])[nc];            // requested before the MFMAs, used after them
#pragma unroll 2
            for (int s = 0; s < KS; ++s) {
                const float4 b0 = ld4(wb + 8 * s), a0 = ld4(xa + 8 * s);
                MFMA4(acc, a0, b0);
            }
            if (sj.which && nv) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Xout[((r & 3) + 8 * (r >> 2) + 4 * lh) * XO + tower * dout + n] = acc[r];
            }
        }
        lds_barrier();
        if (sj.active && !sj.which && nv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[r] + bv;
                if (mrow[row] != 0.f) v += Xout[row * XO + tower * dout + n];
                v = v > 0.f ? v : 0.f;
                Xout[row * XO + tower * dout + n] = v;
                if (row0 + row < R) acts[(row0 + row) * actw + off + tower * dout + n] = v;
            }
        }
    }
}

// bwd_layer_lds (l >= 1, dout % 8 == 0), one product per wave; the cross accumulator waits in Gn
__device__ __forceinline__ void bwd_layer_lds8(const conet_net& net, int l, const float* __restrict__ Gl, float* __restrict__ Gn,
                                               const float* __restrict__ wl, const float* __restrict__ mrow, int wave, int li, int lh,
                                               bool anym) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int GS = 2 * dout + 4, GN = 2 * din + 4, ldw = din + 4;
    const int NT = (din + 31) >> 5, KS = dout >> 3, NU = 2 * NT;
    for (int u0 = 0; u0 < NU; u0 += 4) {
        const split_job sj = split_pick(wave, u0, NU);
        const int tower = sj.u / NT, c0 = (sj.u - tower * NT) * 32 + li;
        const bool cv = c0 < din;
        const int cc = cv ? c0 : 0;
        const bool work = sj.active && (!sj.which || anym);
        f32x16 acc = zero16();
        if (work) {
            const float* Wb = wl + net.wl_off[l] + (sj.which ? 2 * dout * ldw : (tower ? dout * ldw : 0)) + 4 * lh * ldw + cc;
            const float* aa = Gl + li * GS + ((tower ^ sj.which) ? dout : 0) + 4 * lh;
#pragma unroll 2
            for (int s = 0; s < KS; ++s) {
                const float* w_ = Wb + 8 * s * ldw;
                const float4 m = make_float4(w_[0], w_[ldw], w_[2 * ldw], w_[3 * ldw]);
                const float4 a0 = ld4(aa + 8 * s);
                MFMA4(acc, a0, m);
            }
            if (sj.which && cv) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Gn[((r & 3) + 8 * (r >> 2) + 4 * lh) * GN + tower * din + c0] = acc[r];
            }
        }
        lds_barrier();
        if (sj.active && !sj.which && cv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[r];
                if (mrow[row] != 0.f) v += Gn[row * GN + tower * din + c0];
                Gn[row * GN + tower * din + c0] = v;
            }
        }
    }
}

// bwd_layer_tiles<true, 1> (l >= 1, dout % 32 == 0, din % 32 == 0 -- the 128-column shapes too), one product per wave
__device__ __forceinline__ void bwd_layer_tiles8(const conet_net& net, int l, const float* __restrict__ Gl, float* __restrict__ Gn,
                                                 const float* __restrict__ wl, const float* __restrict__ mrow, int wave, int li, int lh,
                                                 bool anym) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int GS = 2 * dout + 4, GN = 2 * din + 4;
    const int NG = din >> 5, KQ = dout >> 5, NU = 2 * NG;
    const int ldw = din + 4;
    for (int u0 = 0; u0 < NU; u0 += 4) {
        const split_job sj = split_pick(wave, u0, NU);
        const int tower = sj.u / NG, c0 = (sj.u - tower * NG) * 32 + li;
        const bool work = sj.active && (!sj.which || anym);
        f32x16 acc = zero16();
        if (work) {
            const float* Wb = wl + net.wl_off[l] + (sj.which ? 2 * dout * ldw : (tower ? dout * ldw : 0));
            const unsigned vo = (unsigned)(4 * lh * ldw + c0);
            const float* aa = Gl + li * GS + ((tower ^ sj.which) ? dout : 0) + 4 * lh;
            float4 m0, m1, m2, m3;
#define CDR_LOADS(M, K) { const float* w_ = Wb + (K) * ldw; M = make_float4(w_[vo], (w_ + ldw)[vo], (w_ + 2 * ldw)[vo], (w_ + 3 * ldw)[vo]); }
#define CDR_MFS(M, K) { const float4 a0 = ld4(aa + (K)); MFMA4(acc, a0, M); }
            CDR_LOADS(m0, 0);
            CDR_LOADS(m1, 8);
            for (int q = 0; q < KQ; ++q) {
                const int kb = q << 5;
                const int kw = q + 1 < KQ ? kb + 32 : 0;
                CDR_LOADS(m2, kb + 16);
                CDR_MFS(m0, kb);
                CDR_LOADS(m3, kb + 24);
                CDR_MFS(m1, kb + 8);
                CDR_LOADS(m0, kw);
                CDR_MFS(m2, kb + 16);
                CDR_LOADS(m1, kw + 8);
                CDR_MFS(m3, kb + 24);
            }
#undef CDR_LOADS
#undef CDR_MFS
            if (sj.which) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Gn[((r & 3) + 8 * (r >> 2) + 4 * lh) * GN + tower * din + c0] = acc[r];
            }
        }
        lds_barrier();
        if (sj.active && !sj.which) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[r];
                if (mrow[row] != 0.f) v += Gn[row * GN + tower * din + c0];
                Gn[row * GN + tower * din + c0] = v;
            }
        }
    }
}

// bwd_layer_quad for layer 0 (dout % 32 == 0, din % 128 == 0, weights streamed from L2), one product per wave.  The result leaves
// for gx0 in HBM, so the cross accumulator waits in a scratch region X [32][2 din + 4] of LDS (conet_fb_lds::x_off).
__device__ __forceinline__ void bwd_layer_quad8(const conet_net& net, int l, const float* __restrict__ Gl, float* __restrict__ X,
                                                const float* __restrict__ mrow, float* __restrict__ gx0, int64_t row0, int64_t R,
                                                int wave, int li, int lh, bool anym) {
    const int din = net.dims[l], dout = net.dims[l + 1];
    const int GS = 2 * dout + 4, GN = 2 * din + 4;
    const int NG = din >> 7, KQ = dout >> 5, NU = 2 * NG;
    const int ldw = din;
    for (int u0 = 0; u0 < NU; u0 += 4) {
        const split_job sj = split_pick(wave, u0, NU);
        const int tower = sj.u / NG, c0 = (sj.u - tower * NG) * 128 + 4 * li;
        const bool work = sj.active && (!sj.which || anym);
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = zero16();
        if (work) {
            const float* Wb = sj.which ? net.H[l] : (tower ? net.Wt[l] : net.Ws[l]);
            const unsigned vo = (unsigned)(4 * lh * ldw + c0);
            const float* aa = Gl + li * GS + ((tower ^ sj.which) ? dout : 0) + 4 * lh;
            float4 m0[4], m1[4], m2[4], m3[4];                                    // [r] = W[k + r][c0 .. c0 + 3]
#define CDR_LOADS(M, K) _Pragma("unroll") for (int r = 0; r < 4; ++r) { M[r] = ld4(Wb + ((K) + r) * ldw + vo); }
#define CDR_MFS(M, K) {                                                                             \
            const float4 a0 = ld4(aa + (K));                                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
                MF1(acc[j], a0.x, comp4(M[0], j)); MF1(acc[j], a0.y, comp4(M[1], j));               \
                MF1(acc[j], a0.z, comp4(M[2], j)); MF1(acc[j], a0.w, comp4(M[3], j)); } }
            CDR_LOADS(m0, 0);
            CDR_LOADS(m1, 8);
            __builtin_amdgcn_sched_barrier(0);
            for (int q = 0; q < KQ; ++q) {
                const int kb = q << 5;
                const int kw = q + 1 < KQ ? kb + 32 : 0;
                CDR_LOADS(m2, kb + 16); __builtin_amdgcn_sched_barrier(0);       // see fwd_layer_stream
                CDR_MFS(m0, kb);        __builtin_amdgcn_sched_barrier(0);
                CDR_LOADS(m3, kb + 24); __builtin_amdgcn_sched_barrier(0);
                CDR_MFS(m1, kb + 8);    __builtin_amdgcn_sched_barrier(0);
                CDR_LOADS(m0, kw);      __builtin_amdgcn_sched_barrier(0);
                CDR_MFS(m2, kb + 16);   __builtin_amdgcn_sched_barrier(0);
                CDR_LOADS(m1, kw + 8);  __builtin_amdgcn_sched_barrier(0);
                CDR_MFS(m3, kb + 24);   __builtin_amdgcn_sched_barrier(0);
            }
#undef CDR_LOADS
#undef CDR_MFS
            if (sj.which) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st4(X + ((r & 3) + 8 * (r >> 2) + 4 * lh) * GN + tower * din + c0, make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]));
            }
        }
        lds_barrier();
        if (sj.active && !sj.which) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float4 v = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                if (mrow[row] != 0.f) {
                    const float4 c = ld4(X + row * GN + tower * din + c0);
                    v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
                }
                if (row0 + row < R) st4(gx0 + (row0 + row) * (2 * (int64_t)din) + tower * din + c0, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ forward + data backward
// Training steps: conet_fwd_kernel's pass and conet_bwd_kernel's pass over the same 32 rows in ONE launch, for a unit upstream
// gradient (the loss is what `.backward()` is called on; any other factor is applied afterwards, conet_wgrad_finish_kernel).  Nothing
// about a row block's data gradient depends on another block -- BCELoss's mean divides by the batch size, known up front -- so the
// backward can start the moment the block's output unit is done: every layer's activations are still in LDS (each keeps its own
// region instead of two ping-pong buffers), the backward's prologue (probabilities, labels, masks and the last activations back from
// HBM: 4.5 us) and its copy of the staged weights disappear with the second launch.  The two gradient buffers take over the region
// of the layer-0 input, which is dead after the first cross unit.  Same device functions, same order of operations as the two
// kernels: bit-identical activations, gz, input gradients and loss.
struct conet_fb_lds { int a_off[kMaxL + 1]; int g0_off, g1_off, wl_off, x_off; };   // x_off: conet_fb_kernel<8>'s scratch (bwd_layer_quad8)

// The kernel's prologue with every global request in flight at once: stage_weights() walks its 35 KB one load -> one LDS store at a
// time (nine dependent L2 round trips per thread: 7.1 us, measured with wall_clock64 stamps), and the gather then starts its own two
// dependent rounds (ids, then rows: 5.9 us).  Here the small layers' weights travel by LDS DMA (global_load_lds_dwordx4: no
// registers, nothing to wait for until the first barrier) out of a loop of a few dozen instructions -- the prologue runs once per
// workgroup, so every instruction of it is an instruction-cache miss, and a register-staged version unrolled over eight chunks per
// thread was no faster than the dependent loop -- followed by the output units' weights, the thread's element of every sum H_l^2,
// and the gather's ids and rows; LDS stores come last.
struct fb_prologue {
    float wo;
    float he[kMaxL];
};

// One DMA instruction fills 64 consecutive 16-byte chunks of the row-PADDED weight area ([rows][din + 4] per matrix): the lane's
// source is found from its destination chunk (a pad chunk re-reads its row's first chunk; nobody reads it back).
__device__ __forceinline__ void stage_weights_dma(const conet_net& net, float* wl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)wl;
    const int total = net.wl_off[net.L] >> 2;                       // padded chunks
    if (net.dma_tab) {
        // the sources come from a table the host filled (conet_dma_table): the arithmetic below is ~100 dependent instructions per trip of
        // a wave that runs it once, at ~10 cycles each with two waves on a SIMD -- 4.5 us of every workgroup's 60
        constexpr int kTrips = 8;                                   // (the host offers the table only when this covers the area)
        const float* src[kTrips];
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int i0 = 64 * wave + k * (int)blockDim.x;
            int c = i0 + lane;
            if (c >= total) c = total - 1;
            src[k] = i0 < total ? reinterpret_cast<const float*>(net.dma_tab[c]) : nullptr;
        }
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int i0 = 64 * wave + k * (int)blockDim.x;
            if (i0 < total) {                                       // wave-uniform
                const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)i0 * 16u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src[k]), "s"(dst) : "memory");
            }
        }
        return;
    }
    for (int i0 = 64 * wave; i0 < total; i0 += (int)blockDim.x) {
        int c = i0 + lane;
        if (c >= total) c = total - 1;                              // (the last instruction's tail lanes: a valid source, dropped below)
        const float* src = net.Ws[1];
        // (l is a compile-time constant in every copy: indexed by a loop variable, each field of `net` is a dependent scalar load from
        //  the kernel-argument segment -- 6 us for this loop)
#pragma unroll
        for (int l = 1; l < kMaxL; ++l) {
            const int din = net.dims[l], dout = net.dims[l + 1], q1 = (din >> 2) + 1, n1 = dout * q1;
            const int r3 = c - (net.wl_off[l] >> 2);
            if (l < net.L && r3 >= 0 && r3 < 3 * n1) {
                const int mat = (r3 >= n1 ? 1 : 0) + (r3 >= 2 * n1 ? 1 : 0);
                const int r = r3 - mat * n1;
                const int row = (int)(((float)r + 0.5f) * (1.0f / (float)q1));         // exact: r < 2^20, q1 small
                const int col = r - row * q1;
                src = (mat == 0 ? net.Ws[l] : mat == 1 ? net.Wt[l] : net.H[l]) + (int64_t)row * din + 4 * (col < q1 - 1 ? col : 0);
            }
        }
        // the instruction writes lane-linear from M0: lanes past `total` would land in the next region -- the host sizes the weight
        // area to whole instructions (fill_net: 256 floats of slack)
        const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)i0 * 16u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
}

// this block's slice [lo, hi) of H_l (the sum of squares is split over the grid: conet_fwd_kernel)
__device__ __forceinline__ void hsq_slice(const conet_net& net, int l, int& lo, int& hi) {
    const int n = net.dims[l] * net.dims[l + 1];
    const int chunk = net.hsq_chunk[l];
    lo = (int)blockIdx.x * chunk;
    hi = lo + chunk < n ? lo + chunk : n;
}

// In two halves, with the first block's gather between them: {the first block's ids} -> weights' DMA -> {its rows} -> the small loads below
// -> LDS stores, so that the gather's two dependent round trips run under the prologue's instruction issue instead of behind it.  (Tried
// earlier in round 6, when the DMA loop and this half were 8 us of address arithmetic: no gain -- the prologue was issue-bound whatever the
// order.  With the DMA table and the host-side slice bounds it is 4.4 us of issue and the gather's latency is what is left.)
__device__ __forceinline__ void prologue_dma(const conet_net& net, float* wl) {
    STAMP(43);
    if (net.wlds) stage_weights_dma(net, wl);
    STAMP(44);
}
__device__ __forceinline__ void prologue_small(const conet_net& net, fb_prologue& pr) {
    const int t = threadIdx.x, dL = net.dims[net.L];
    pr.wo = 0.f;
    if (t < 2 * (dL + 1)) {
        const int tower = t / (dL + 1), jj = t - tower * (dL + 1);
        pr.wo = jj < dL ? net.wo[tower][jj] : net.bo[tower][0];
    }
#pragma unroll
    for (int l = 0; l < kMaxL; ++l) {
        pr.he[l] = 0.f;
        if (l < net.L) {
            int lo, hi;
            hsq_slice(net, l, lo, hi);
            if (t < 256 && lo + t < hi) pr.he[l] = net.H[l][lo + t];
        }
    }
    STAMP(45);
}

// hq [kMaxL][4]: per wave, this block's slice of every sum H_l^2 (added over the waves in wave order by the tail, as block_sum_d does)
__device__ __forceinline__ void prologue_commit(const conet_net& net, float* wo_sh, double* hq, fb_prologue& pr) {
    const int t = threadIdx.x;
    if (t < 2 * (net.dims[net.L] + 1)) wo_sh[t] = pr.wo;
    if (t >= 256) return;                      // (conet_fb_kernel<8>: the H^2 slices stay on the first four waves -- same sums in the same order)
#pragma unroll
    for (int l = 0; l < kMaxL; ++l) {
        if (l < net.L) {
            double q = (double)pr.he[l] * (double)pr.he[l];
            int lo, hi;
            hsq_slice(net, l, lo, hi);
            const float* h = net.H[l];
            for (int e = lo + t + 256; e < hi; e += 256) q += (double)h[e] * (double)h[e];
            q = wave_sum_d(q);
            if ((t & 63) == 0) hq[l * 4 + (t >> 6)] = q;
        }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void conet_fb_kernel(conet_net net, conet_fb_lds lo, const float* __restrict__ su,
                                                       const float* __restrict__ si, const float* __restrict__ tu,
                                                       const float* __restrict__ ti, int D, const int64_t* __restrict__ user_s,
                                                       const int64_t* __restrict__ user_t, const int64_t* __restrict__ item_s,
                                                       const int64_t* __restrict__ item_t, int64_t R, int64_t n_source,
                                                       int64_t n_overlap, int overlap_users, const float* __restrict__ label_s,
                                                       const float* __restrict__ label_t, float* __restrict__ label_cat,
                                                       int64_t* __restrict__ ids_cat, float* __restrict__ x0, float* __restrict__ acts,
                                                       float* __restrict__ prob, float* __restrict__ maskf, double* __restrict__ partials,
                                                       float* __restrict__ gz, float* __restrict__ gx0, float* __restrict__ ou_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float mrow[kRows], dzrow[kRows];
    __shared__ int trow[kRows];
    __shared__ float yrow[kRows], wo_sh[256];
    __shared__ double red[2 * NW], hq[kMaxL * 4];
    constexpr int NTH = 64 * NW, TPR = 2 * NW, NI = 32 / TPR;        // threads, threads per gathered row, 16-B chunks per thread and table
    float* wl = smem + lo.wl_off;
    float* G0 = smem + lo.g0_off;
    float* G1 = smem + lo.g1_off;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int D4 = D >> 2, L = net.L, actw = net.act_off[L], dL = net.dims[L];
    double lacc0 = 0.0, lacc1 = 0.0;
    float ou_acc = 0.f;
    STAMP(40);
    STAMPB(0);
    fb_prologue pr;
    const int64_t nrb = (R + kRows - 1) / kRows;
    // the first block's ids and label: requested before anything else (see prologue_dma)
    int64_t uid0 = 0, iid0 = 0;
    float y0 = 0.f;
    if ((int64_t)blockIdx.x < nrb) {
        const int64_t g = (int64_t)blockIdx.x * kRows + t / TPR, gc = g < R ? g : R - 1;
        const bool src = gc < n_source;
        uid0 = src ? user_s[gc] : user_t[gc - n_source]; iid0 = src ? item_s[gc] : item_t[gc - n_source];
        y0 = (t % TPR) == 0 ? (src ? label_s[gc] : label_t[gc - n_source]) : 0.f;
    }
    prologue_dma(net, wl);
    auto gather = [&](int64_t rb, auto first) {   // ---- gather [su | si | tu | ti] of 32 rows (8 threads per row, 16 B each): every row request of a 128-column pass is in
            //      flight before anything is parked
            float* bufA = smem + lo.a_off[0];
            const int row = t / TPR, c0 = t % TPR;
            const int64_t g = rb * kRows + row;
            const bool valid = g < R;
            const int64_t gc = valid ? g : R - 1;
            const bool src = gc < n_source;
            int64_t uid, iid;
            float y;
            if (decltype(first)::value) { uid = uid0; iid = iid0; y = y0; }
            else {
                uid = src ? user_s[gc] : user_t[gc - n_source]; iid = src ? item_s[gc] : item_t[gc - n_source];
                y = c0 == 0 ? (src ? label_s[gc] : label_t[gc - n_source]) : 0.f;
            }
            float* xr = bufA + row * (4 * D + 4);
            for (int cb = 0; cb < D4; cb += 32) {
                float4 a[NI], b[NI], e[NI], f[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int c = cb + c0 + TPR * i;
                    if (c < D4) {
                        a[i] = ld4(su + uid * D + 4 * c); b[i] = ld4(si + iid * D + 4 * c);
                        e[i] = ld4(tu + uid * D + 4 * c); f[i] = ld4(ti + iid * D + 4 * c);
                    }
                }
                if (decltype(first)::value && cb == 0) { prologue_small(net, pr); prologue_commit(net, wo_sh, hq, pr); }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int c = cb + c0 + TPR * i;
                    if (c < D4) {
                        st4(xr + 4 * c, a[i]); st4(xr + D + 4 * c, b[i]); st4(xr + 2 * D + 4 * c, e[i]); st4(xr + 3 * D + 4 * c, f[i]);
                        if (valid) {
                            float* xg = x0 + g * (4 * (int64_t)D);
                            st4(xg + 4 * c, a[i]); st4(xg + D + 4 * c, b[i]); st4(xg + 2 * D + 4 * c, e[i]); st4(xg + 3 * D + 4 * c, f[i]);
                        }
                    }
                }
            }
            if (decltype(first)::value) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weights' DMA has landed (nothing tracks it)
            if (c0 == 0) {
                const float m = ((overlap_users ? uid : iid) < n_overlap) ? 1.f : 0.f;     // PAD id 0 counts
                mrow[row] = m;
                yrow[row] = y;
                if (valid) {
                    maskf[g] = m;
                    label_cat[g] = y;
                    ids_cat[g] = uid; ids_cat[R + g] = iid;               // for the embedding update (scatter / row-wise sort)
                }
            }
        };
    // the loop is rotated: a block's gather sits at the END of the previous block's pass, the first one in front of the loop with the
    // prologue's LDS stores inside it (inside the loop, the prologue's registers would stay live through every MFMA phase)
    STAMP(0);
    if ((int64_t)blockIdx.x < nrb) gather((int64_t)blockIdx.x, std::true_type{});
    else { prologue_small(net, pr); prologue_commit(net, wo_sh, hq, pr); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        lds_barrier();
        STAMP(1);
        const bool anym = NW == 8 && __ballot(mrow[li] != 0.f) != 0ull;     // any overlapped row in this block (8 waves: else no cross product)
        for (int l = 0; l < L; ++l) {
            const float* Xin = smem + lo.a_off[l];
            float* Xout = smem + lo.a_off[l + 1];
            if (NW == 8) {                                        // (the host offers this kernel for the shape class of these two only)
                if (l == 0) fwd_layer_stream8(net, l, Xin, Xout, mrow, acts, rb * kRows, R, wave, li, lh, anym);
                else fwd_layer_lds8(net, l, Xin, Xout, wl, mrow, acts, rb * kRows, R, wave, li, lh, anym);
            } else
            if (l > 0 && net.wlds && !(net.dims[l] & 7)) fwd_layer_lds(net, l, Xin, Xout, wl, mrow, acts, rb * kRows, R, wave, li, lh);
            else if (l > 0 && net.wlds) fwd_layer<true>(net, l, Xin, Xout, wl, mrow, acts, rb * kRows, R, wave, li, lh);
            else if (net.vec && !(net.dims[l] & 127) && !(net.dims[l + 1] & 31))
                fwd_layer_stream(net, l, Xin, Xout, mrow, acts, rb * kRows, R, wave, li, lh);
            else fwd_layer<false>(net, l, Xin, Xout, wl, mrow, acts, rb * kRows, R, wave, li, lh);
            lds_barrier();
            STAMP(2 + l);
        }
        const float* hl = smem + lo.a_off[L];                       // [32][2 dL + 4]
        const int HS = 2 * dL + 4;
        if (t < kRows) {   // ---- output unit + sigmoid + BCE term (conet.py:140,179,195-196), and d loss / d logit for a unit upstream gradient
            const int64_t g = rb * kRows + t;
            float dz = 0.f;
            int tower = 0;
            if (g < R) {
                tower = g >= n_source ? 1 : 0;
                const float* h = hl + t * HS + tower * dL;
                const float* w = wo_sh + tower * (dL + 1);
                float z = 0.f;
                for (int j = 0; j < dL; ++j) z += h[j] * w[j];
                z += w[dL];
                const float p = 1.0f / (1.0f + expf(-z));
                prob[g] = p;
                const float y = yrow[t];
                const double term = (double)((y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f));
                if (tower) lacc1 += term; else lacc0 += term;
                const float nd = (float)(tower ? R - n_source : n_source);
                const float gp = (1.0f / nd) * (p - y) / fmaxf((1.0f - p) * p, 1e-12f);      // BCELoss backward (mean)
                dz = (gp * (1.0f - p)) * p;                                                  // sigmoid backward
            }
            dzrow[t] = dz; trow[t] = tower;
        }
        lds_barrier();
        STAMP(2 + L);
        {   // ---- the output units' input gradient and their own weight gradients (this block's rows, in row order)
            float* Gl = ((L - 1) & 1) ? G1 : G0;
            for (int e = t; e < kRows * 2 * dL; e += NTH) {
                const int row = e / (2 * dL), c = e - row * 2 * dL;
                const int tower = c >= dL ? 1 : 0, j = c - tower * dL;
                Gl[row * HS + c] = (trow[row] == tower) ? dzrow[row] * wo_sh[tower * (dL + 1) + j] : 0.f;
            }
            if (t < 2 * (dL + 1)) {
                const int tower = t / (dL + 1), jj = t - tower * (dL + 1);
                float dzv[kRows], hv[kRows];                          // all 64 LDS reads first, then the row-ordered sum
#pragma unroll
                for (int row = 0; row < kRows; ++row) {
                    dzv[row] = trow[row] == tower ? dzrow[row] : 0.f;
                    hv[row] = jj < dL ? hl[row * HS + tower * dL + jj] : 1.0f;
                }
#pragma unroll
                for (int row = 0; row < kRows; ++row) ou_acc = fmaf(dzv[row], hv[row], ou_acc);
            }
        }
        lds_barrier();
        for (int l = L - 1; l >= 0; --l) {
            const int dout = net.dims[l + 1];
            const int GS = 2 * dout + 4;
            float* Gl = (l & 1) ? G1 : G0;
            float* Gn = (l & 1) ? G0 : G1;
            const float* Al = smem + lo.a_off[l + 1];                // this layer's post-ReLU outputs, same row stride as Gl
            const int off = net.act_off[l];
            const int q = (2 * dout) >> 2;
            for (int e = t; e < kRows * q; e += NTH) {              // ReLU backward; gz kept for the weight gradients
                const int row = e / q, c = 4 * (e - row * q);
                const int64_t g = rb * kRows + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g < R) {
                    const float4 a = ld4(Al + row * GS + c), gg = ld4(Gl + row * GS + c);
                    v = make_float4(a.x > 0.f ? gg.x : 0.f, a.y > 0.f ? gg.y : 0.f, a.z > 0.f ? gg.z : 0.f, a.w > 0.f ? gg.w : 0.f);
                    st4(gz + g * actw + off + c, v);
                }
                st4(Gl + row * GS + c, v);
            }
            lds_barrier();
            const int din = net.dims[l];
            const bool lds_w = l > 0 && net.wlds;
            if (NW == 8) {
                if (l == 0) bwd_layer_quad8(net, l, Gl, smem + lo.x_off, mrow, gx0, rb * kRows, R, wave, li, lh, anym);
                else if (!(dout & 31) && !(din & 31)) bwd_layer_tiles8(net, l, Gl, Gn, wl, mrow, wave, li, lh, anym);
                else bwd_layer_lds8(net, l, Gl, Gn, wl, mrow, wave, li, lh, anym);
            } else
            if (!(dout & 31) && !(din & 127)) {
                if (lds_w) bwd_layer_quad<true>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
                else bwd_layer_quad<false>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            } else if (!(dout & 31) && !(din & 31)) {
                if (lds_w) bwd_layer_tiles<true, 1>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
                else bwd_layer_tiles<false, 1>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            } else if (lds_w && !(dout & 7)) bwd_layer_lds(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            else if (lds_w) bwd_layer<true>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            else bwd_layer<false>(net, l, Gl, Gn, wl, mrow, gx0, rb * kRows, R, wave, li, lh);
            lds_barrier();
            STAMP(3 + L + (L - 1 - l));
        }
        if (rb + gridDim.x < nrb) gather(rb + gridDim.x, std::false_type{});
    }
    STAMP(41);
    if (t < 2 * (dL + 1)) ou_part[(size_t)blockIdx.x * 2 * (dL + 1) + t] = ou_acc;
    double lacc[2];
    lacc[0] = lacc0; lacc[1] = lacc1;
    block_sum_d<2>(lacc, red);
    if (t == 0) {
        double* o = partials + (size_t)blockIdx.x * kConetPartial;
        o[0] = lacc[0]; o[1] = lacc[1];
#pragma unroll
        for (int l = 0; l < kMaxL; ++l) o[2 + l] = l < net.L ? ((hq[4 * l] + hq[4 * l + 1]) + hq[4 * l + 2]) + hq[4 * l + 3] : 0.0;
    }
    // (Tried: the block that finishes last -- a sign-in counter behind __threadfence() -- adds the partials instead of
    //  conet_fwd_finish_kernel's launch.  The release fence is an L2 write-back at agent scope, i.e. of EVERY dirty line of the XCD:
    //  this kernel's 40 MB of saved activations and gradients.  72 -> 82 us; the 5 us launch stays.)
    STAMP(42);
    STAMPB(1);
}

// ------------------------------------------------------------------------------------------------------------ weight gradients
// One WAVE per (layer, pair of 32-row m tiles, 32-column n tile, chunk of batch rows): it accumulates the Ws, Wt and H tiles
// of its m tiles at once, so a K step of 8 batch rows costs 16 + 8 operand loads for 32 MFMAs (the first version, one
// tile per wave, loaded 8 per 4 and re-read every operand panel for every tile: 105 us + a 146 us serial reduction).
// The four waves of a workgroup share the job and add their tiles through LDS before the partial is written (wgrad_job).
struct job_id { int l, mp, nt; };
__device__ __forceinline__ job_id decode_job(const conet_net& net, const conet_tiles& jl, int jb) {
    job_id d;
    d.l = 0;
    for (int l = 1; l < net.L; ++l) if (jb >= jl.off[l]) d.l = l;
    const int local = jb - jl.off[d.l];
    const int NT = (net.dims[d.l] + 31) >> 5;
    d.mp = local / NT;
    d.nt = local - d.mp * NT;
    return d;
}

struct wg_ops { float s0[4], t0[4], s1[4], t1[4], xs[4], xt[4], mk[4]; };
struct wg_src { const float* gs0; const float* gs1; const float* bsp; const float* maskf; int64_t actw, ldin, r_safe, r_end; int dout, din; };

// 8 batch rows of operands (lane half lh takes rows first + 0..3): no predicate -- rows past the chunk are re-read from a
// valid row and zeroed (A side) where they are used, columns past the layer are clamped and never read back.
template <bool TWO>
__device__ __forceinline__ void wg_load(wg_ops& o, const wg_src& q, int64_t first) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t row = first + c, rc = row < q.r_end ? row : q.r_safe;
        o.mk[c] = q.maskf[rc];
        o.s0[c] = q.gs0[rc * q.actw]; o.t0[c] = q.gs0[rc * q.actw + q.dout];
        o.xs[c] = q.bsp[rc * q.ldin]; o.xt[c] = q.bsp[rc * q.ldin + q.din];
        if (TWO) { o.s1[c] = q.gs1[rc * q.actw]; o.t1[c] = q.gs1[rc * q.actw + q.dout]; }
    }
}

// Ws += gz_s^T s_in ; Wt += gz_t^T t_in ; H += (m gz_s)^T t_in + (m gz_t)^T s_in      (conet.py:127-134 transposed)
template <bool TWO, bool TAIL>
__device__ __forceinline__ void wg_use(const wg_ops& cu, int64_t first, int64_t r_end, f32x16 (&acc)[6], float (&bias)[4]) {
    float s0[4], t0[4], s1[4], t1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool rv = !TAIL || first + c < r_end;
        s0[c] = rv ? cu.s0[c] : 0.f; t0[c] = rv ? cu.t0[c] : 0.f;
        if (TWO) { s1[c] = rv ? cu.s1[c] : 0.f; t1[c] = rv ? cu.t1[c] : 0.f; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float ms = cu.mk[c] != 0.f ? s0[c] : 0.f, mt = cu.mk[c] != 0.f ? t0[c] : 0.f;
        MF1(acc[0], s0[c], cu.xs[c]);
        MF1(acc[1], t0[c], cu.xt[c]);
        MF1(acc[2], ms, cu.xt[c]);
        MF1(acc[2], mt, cu.xs[c]);
    }
    bias[0] += (s0[0] + s0[1]) + (s0[2] + s0[3]);
    bias[1] += (t0[0] + t0[1]) + (t0[2] + t0[3]);
    if (TWO) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ms = cu.mk[c] != 0.f ? s1[c] : 0.f, mt = cu.mk[c] != 0.f ? t1[c] : 0.f;
            MF1(acc[3], s1[c], cu.xs[c]);
            MF1(acc[4], t1[c], cu.xt[c]);
            MF1(acc[5], ms, cu.xt[c]);
            MF1(acc[5], mt, cu.xs[c]);
        }
        bias[2] += (s1[0] + s1[1]) + (s1[2] + s1[3]);
        bias[3] += (t1[0] + t1[1]) + (t1[2] + t1[3]);
    }
}

// The row loop of one wave and the workgroup's reduction.  Three operand sets rotate through a loop unrolled by three steps, so a
// step's 20-28 requests are issued two steps (64 MFMAs with a second m tile) before they are used -- gz and x0 were written by
// the previous kernel and come from the Infinity Cache, further away than one step; sched_barrier keeps hipcc from sinking
// them back (see fwd_layer_stream).  The four waves of a workgroup take four consecutive row chunks of the SAME job and add
// their tiles through LDS in wave order before anything goes to HBM: a quarter of the partials to write here and to re-read in
// conet_wgrad_finish_kernel (26 MB -> 6.5 MB per step at C3).
template <bool TWO>
__device__ __forceinline__ void wgrad_job(const wg_src& q, int64_t r_begin, bool bias_job, float* __restrict__ red, float* __restrict__ o,
                                          int wave, int lane, int lh) {
    f32x16 acc[6];
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = zero16();
    const int64_t nrows = q.r_end > r_begin ? q.r_end - r_begin : 0;
    const int64_t nfull = nrows >> 3;
    const int64_t rf = r_begin + 4 * lh;
    wg_ops A, B, C;
    wg_load<TWO>(A, q, rf);
    wg_load<TWO>(B, q, rf + 8);
    __builtin_amdgcn_sched_barrier(0);
    int64_t st = 0;
    for (; st + 3 <= nfull; st += 3) {
        const int64_t r = rf + 8 * st;
        wg_load<TWO>(C, q, r + 16);              __builtin_amdgcn_sched_barrier(0);
        wg_use<TWO, false>(A, r, q.r_end, acc, bias);      __builtin_amdgcn_sched_barrier(0);
        wg_load<TWO>(A, q, r + 24);              __builtin_amdgcn_sched_barrier(0);
        wg_use<TWO, false>(B, r + 8, q.r_end, acc, bias);  __builtin_amdgcn_sched_barrier(0);
        wg_load<TWO>(B, q, r + 32);              __builtin_amdgcn_sched_barrier(0);
        wg_use<TWO, false>(C, r + 16, q.r_end, acc, bias); __builtin_amdgcn_sched_barrier(0);
    }
    if (st < nfull) {
        wg_use<TWO, false>(A, rf + 8 * st, q.r_end, acc, bias);
        ++st;
        if (st < nfull) { wg_use<TWO, false>(B, rf + 8 * st, q.r_end, acc, bias); ++st; }
    }
    if (nrows & 7) {                                          // the chunk that ends at R: rows past it contribute zeros
        wg_load<TWO>(C, q, rf + 8 * nfull);
        wg_use<TWO, true>(C, rf + 8 * nfull, q.r_end, acc, bias);
    }
    // this wave's partial in the layout of the job's partial: tiles in accumulator order {m tile, {Ws, Wt, H}}, then four bias rows
    float* my = red + (size_t)wave * kJobFloats;
    constexpr int NTILE = TWO ? 6 : 3;
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) my[i * 1024 + r * 64 + lane] = acc[i][r];
    }
    if (bias_job) {                                           // bias gradients = column sums of gz: the A operand's running sum
        const float a = bias[0] + __shfl_xor(bias[0], 32, 64), b = bias[1] + __shfl_xor(bias[1], 32, 64);
        const float c = bias[2] + __shfl_xor(bias[2], 32, 64), e = bias[3] + __shfl_xor(bias[3], 32, 64);
        if (lh == 0) { const int li = lane & 31; my[6144 + li] = a; my[6144 + 32 + li] = b; my[6144 + 64 + li] = c; my[6144 + 96 + li] = e; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NTILE * 1024; e += 256)
        o[e] = ((red[e] + red[kJobFloats + e]) + red[2 * kJobFloats + e]) + red[3 * kJobFloats + e];
    if (bias_job && threadIdx.x < 128) {
        const int e = 6144 + threadIdx.x;
        o[e] = ((red[e] + red[kJobFloats + e]) + red[2 * kJobFloats + e]) + red[3 * kJobFloats + e];
    }
}

// One WORKGROUP per (job, group of four row chunks); wave w takes chunk 4 g + w.
// `fin` (cdr_conet_defer_finish): the launch's LAST workgroup adds the forward blocks' loss partials instead of a weight-gradient job
// (conet_fwd_finish_kernel's work; the reader of `out`, conet_wgrad_finish_kernel, is the next launch)
struct conet_fin { const double* partials; float* out; int nblocks; int64_t n_source; };
__global__ __launch_bounds__(256) void conet_wgrad_kernel(conet_net net, conet_tiles jl, int64_t R, int64_t kc,
                                                          const float* __restrict__ x0, const float* __restrict__ acts,
                                                          const float* __restrict__ gz, const float* __restrict__ maskf,
                                                          float* __restrict__ wpart, conet_fin fin) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [4][kJobFloats]
    if (fin.out && blockIdx.x == gridDim.x - 1) {
        conet_finish_block(net, fin.partials, fin.nblocks, fin.n_source, R, fin.out, reinterpret_cast<double*>(red));
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int jb = (int)(blockIdx.x % (unsigned)jl.ntiles), g = (int)(blockIdx.x / (unsigned)jl.ntiles);
    const job_id d = decode_job(net, jl, jb);
    const int din = net.dims[d.l], dout = net.dims[d.l + 1];
    const int actw = net.act_off[net.L];
    const float* in = d.l == 0 ? x0 : acts + net.act_off[d.l - 1];
    const int m0 = d.mp * 64 + li, m1 = m0 + 32, n = d.nt * 32 + li;
    const bool two = d.mp * 64 + 32 < dout;                  // workgroup-uniform: the job has a second m tile
    wg_src q;
    q.gs0 = gz + net.act_off[d.l] + (m0 < dout ? m0 : 0);               // tower s columns; tower t at + dout
    q.gs1 = gz + net.act_off[d.l] + (m1 < dout ? m1 : 0);
    q.bsp = in + (n < din ? n : 0);                                     // s_in column n; t_in at + din
    q.maskf = maskf;
    q.actw = actw; q.ldin = d.l == 0 ? 2 * din : actw;
    q.dout = dout; q.din = din;
    const int64_t r_begin = (int64_t)(4 * g + wave) * kc;
    q.r_end = (r_begin + kc < R) ? r_begin + kc : R;
    q.r_safe = r_begin < R ? r_begin : R - 1;
    float* o = wpart + ((size_t)g * jl.ntiles + jb) * kJobFloats;
    if (two) wgrad_job<true>(q, r_begin, d.nt == 0, red, o, wave, lane, lh);
    else wgrad_job<false>(q, r_begin, d.nt == 0, red, o, wave, lane, lh);
}

// One THREAD per output element: the chunk partials added in chunk order (fixed order, independent loads), plus d||H||_F.
__global__ __launch_bounds__(256) void conet_wgrad_finish_kernel(conet_net net, conet_grads gr, conet_tiles jl, int nsplit,
                                                                 const float* __restrict__ wpart, const float* __restrict__ ou_part,
                                                                 int n_ou_blocks, const float* __restrict__ out,
                                                                 const float* __restrict__ grad_out, float* __restrict__ gx0_unit,
                                                                 int64_t n_gx0) {
    constexpr int kBlocksPerJob = (kJobFloats + 255) / 256;
    __shared__ float ou_sh[256];
    const int jb = blockIdx.x / kBlocksPerJob;
    const float go = grad_out ? grad_out[0] : 1.0f;
    // gx0_unit != null: the data gradients (gz, hence the partials here, and the input gradient) were made by conet_fb_kernel for a
    // unit upstream gradient; everything is linear in it, so any other value is applied now -- a no-op for `loss.backward()`
    const float unit_scale = gx0_unit ? go : 1.0f;
    if (gx0_unit && go != 1.0f)
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (n_gx0 >> 2); e += (int64_t)gridDim.x * 256) {
            float4 v = ld4(gx0_unit + 4 * e);
            v.x *= go; v.y *= go; v.z *= go; v.w *= go;
            st4(gx0_unit + 4 * e, v);
        }
    if (jb >= jl.ntiles) {                                    // output units: block partials added in a fixed two-level order
        const int dL = net.dims[net.L], w = 2 * (dL + 1), nch = 256 / w;
        const int j = threadIdx.x % w, ch = threadIdx.x / w;
        float s = 0.f;
        if (ch < nch) for (int b = ch; b < n_ou_blocks; b += nch) s += ou_part[(size_t)b * w + j];
        ou_sh[threadIdx.x] = s;
        __syncthreads();
        if ((int)threadIdx.x < w) {
            float tot = 0.f;
            for (int c = 0; c < nch; ++c) tot += ou_sh[c * w + j];
            if (unit_scale != 1.0f) tot *= unit_scale;
            const int tower = j / (dL + 1), jj = j - tower * (dL + 1);
            if (jj < dL) gr.wo[tower][jj] = tot; else gr.bo[tower][0] = tot;
        }
        return;
    }
    const int e = (blockIdx.x - jb * kBlocksPerJob) * 256 + threadIdx.x;
    if (e >= kJobFloats) return;
    const job_id d = decode_job(net, jl, jb);
    const int din = net.dims[d.l], dout = net.dims[d.l + 1];
    int m, n = 0, mat;
    if (e < 6144) {
        const int t6 = e >> 10, idx = e & 1023, r = idx >> 6, lane = idx & 63;
        mat = t6 % 3;
        m = d.mp * 64 + (t6 / 3) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        n = d.nt * 32 + (lane & 31);
        if (m >= dout || n >= din) return;
    } else {
        const int bi = e - 6144;
        if (d.nt != 0 || bi >= 128) return;
        mat = 3 + ((bi >> 5) & 1);                            // 3: bs, 4: bt
        m = d.mp * 64 + (bi >> 6) * 32 + (bi & 31);
        if (m >= dout) return;
    }
    const float* base = wpart + (size_t)jb * kJobFloats + e;
    const size_t stride = (size_t)jl.ntiles * kJobFloats;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base[(size_t)(sp + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; sp < nsplit; ++sp) s += base[(size_t)sp * stride];
    if (unit_scale != 1.0f) s *= unit_scale;
    if (mat == 2) {
        const float nv = out[4 + d.l];                        // d||H||_F / dH = H / ||H||_F (0 at 0, as torch.norm's backward)
        if (nv > 0.f) s += (go / nv) * net.H[d.l][(int64_t)m * din + n];
        gr.H[d.l][(int64_t)m * din + n] = s;
    } else if (mat == 0) gr.Ws[d.l][(int64_t)m * din + n] = s;
    else if (mat == 1) gr.Wt[d.l][(int64_t)m * din + n] = s;
    else if (mat == 3) gr.bs[d.l][m] = s;
    else gr.bt[d.l][m] = s;
}

// ------------------------------------------------------------------------------------------------------------ host side
struct lds_plan { int strideA, strideB, strideG0, strideG1; size_t wl_floats, fwd_bytes, bwd_bytes; conet_fb_lds fb; size_t fb_bytes; int fb8; };

int fill_net(conet_net& net, lds_plan& lp, int L, const int* dims, const float* const* params) {
    if (L < 1 || L > kMaxL || !dims || !params) return 0;
    net.L = L;
    net.vec = 1;
    net.dma_tab = nullptr;
    for (int l = 0; l < kMaxL; ++l) net.hsq_chunk[l] = 0;
    int off = 0;
    for (int l = 0; l <= L; ++l) {
        if (dims[l] <= 0 || (dims[l] & 3)) return 0;                 // float4 chunks along every contraction index
        net.dims[l] = dims[l];
    }
    for (int l = 0; l < L; ++l) { net.act_off[l] = off; off += 2 * dims[l + 1]; }
    net.act_off[L] = off;
    for (int l = L + 1; l <= kMaxL; ++l) { net.dims[l] = 0; net.act_off[l] = off; }
    for (int l = 0; l < kMaxL; ++l) { net.Ws[l] = net.bs[l] = net.Wt[l] = net.bt[l] = net.H[l] = nullptr; }
    for (int l = 0; l < L; ++l) {
        net.Ws[l] = params[5 * l]; net.bs[l] = params[5 * l + 1]; net.Wt[l] = params[5 * l + 2]; net.bt[l] = params[5 * l + 3];
        net.H[l] = params[5 * l + 4];
        if (!net.Ws[l] || !net.bs[l] || !net.Wt[l] || !net.bt[l] || !net.H[l]) return 0;
        if (((uintptr_t)net.Ws[l] | (uintptr_t)net.Wt[l] | (uintptr_t)net.H[l]) & 15) net.vec = 0;
    }
    net.wo[0] = params[5 * L]; net.bo[0] = params[5 * L + 1]; net.wo[1] = params[5 * L + 2]; net.bo[1] = params[5 * L + 3];
    if (!net.wo[0] || !net.bo[0] || !net.wo[1] || !net.bo[1]) return 0;
    if (2 * (dims[L] + 1) > 256) return 0;
    // LDS: forward = two ping-pong activation buffers (layer l reads buffer l & 1); backward = two gradient buffers (layer l's
    // output gradient in buffer l & 1) + the last activations; both + the staged weights of layers >= 1 when they fit
    int ea = 0, oa = 0, ge = 0, go = 0;
    for (int l = 0; l <= L; ++l) { int& m = (l & 1) ? oa : ea; if (dims[l] > m) m = dims[l]; }
    for (int l = 0; l < L; ++l) { int& m = (l & 1) ? go : ge; if (dims[l + 1] > m) m = dims[l + 1]; }
    lp.strideA = 2 * ea + 4; lp.strideB = 2 * oa + 4; lp.strideG0 = 2 * ge + 4; lp.strideG1 = 2 * go + 4;
    size_t wl = 0;
    for (int l = 0; l <= kMaxL; ++l) net.wl_off[l] = 0;
    for (int l = 1; l < L; ++l) { net.wl_off[l] = (int)wl; wl += (size_t)3 * dims[l + 1] * (dims[l] + 4); }
    net.wl_off[L] = (int)wl;
    net.wl_chunks = 0;
    for (int l = 1; l < L; ++l) net.wl_chunks += 3 * dims[l + 1] * (dims[l] >> 2);
    const size_t fwd = (size_t)kRows * (lp.strideA + lp.strideB) * sizeof(float);
    const size_t bwd = (size_t)kRows * (lp.strideG0 + lp.strideG1 + 2 * dims[L]) * sizeof(float);
    if (fwd > kLdsBudget || bwd > kLdsBudget) return 0;
    net.wlds = (L > 1 && fwd + wl * sizeof(float) <= kLdsBudget && bwd + wl * sizeof(float) <= kLdsBudget) ? 1 : 0;
    lp.wl_floats = net.wlds ? wl : 0;
    lp.fwd_bytes = fwd + lp.wl_floats * sizeof(float);
    lp.bwd_bytes = bwd + lp.wl_floats * sizeof(float);
    // conet_fb_kernel: the layer-0 input, one region per layer's activations, the staged weights; the two gradient buffers inside the
    // layer-0 input's region when they fit there (fb_bytes = 0: the shape takes the two-launch route)
    size_t o = 0;
    for (int l = 0; l <= L; ++l) { lp.fb.a_off[l] = (int)o; o += (size_t)kRows * (2 * dims[l] + 4); }
    for (int l = L + 1; l <= kMaxL; ++l) lp.fb.a_off[l] = (int)o;
    if ((size_t)kRows * (lp.strideG0 + lp.strideG1) <= (size_t)lp.fb.a_off[1]) { lp.fb.g0_off = 0; lp.fb.g1_off = kRows * lp.strideG0; }
    else { lp.fb.g0_off = (int)o; lp.fb.g1_off = (int)o + kRows * lp.strideG0; o += (size_t)kRows * (lp.strideG0 + lp.strideG1); }
    lp.fb.wl_off = (int)o;
    o += lp.wl_floats + (lp.wl_floats ? 256 : 0);          // whole DMA instructions (stage_weights_dma)
    lp.fb_bytes = (o * sizeof(float) <= kLdsBudget && (net.wlds || L == 1)) ? o * sizeof(float) : 0;
    // conet_fb_kernel<8> (one product per wave): layer 0 streamed in whole tiles, every later layer on LDS-staged weights in 8-wide K
    // steps, and room behind G0 -- over the dead G1, layer-0 input and activation regions -- for the layer-0 cross accumulator
    // [32][2 d0 + 4] (bwd_layer_quad8).  Other shapes keep the four-wave kernel.
    lp.fb8 = 0;
    lp.fb.x_off = lp.fb.g1_off;
    if (lp.fb_bytes && L > 1 && net.wlds && net.vec && !(dims[0] & 127) && !(dims[1] & 31) && lp.fb.g0_off == 0 &&
        (size_t)lp.fb.g1_off + (size_t)kRows * (2 * dims[0] + 4) <= (size_t)lp.fb.wl_off) {
        lp.fb8 = 1;
        for (int l = 1; l <= L; ++l) if (dims[l] & 7) lp.fb8 = 0;
    }
    return 1;
}

void fill_tiles(const conet_net& net, conet_tiles& tl) {        // wgrad jobs: (layer, pair of m tiles, n tile)
    int n = 0;
    for (int l = 0; l < net.L; ++l) {
        tl.off[l] = n;
        n += ((net.dims[l + 1] + 63) / 64) * ((net.dims[l] + 31) / 32);
    }
    for (int l = net.L; l <= kMaxL; ++l) tl.off[l] = n;
    tl.ntiles = n;
}

inline int64_t rows_grid(int64_t R) {
    int64_t g = (R + kRows - 1) / kRows;
    if (g > 2048) g = 2048;                      // <= CDR_MAX_PARTIAL_BLOCKS; 8 resident blocks per CU at most
    return g < 1 ? 1 : g;
}

inline void split_plan(int ntiles, int64_t R, int* nsplit, int64_t* kc) {
    int64_t ns = 4 * (int64_t)(CDR_NUM_CU / ntiles > 0 ? CDR_NUM_CU / ntiles : 1);   // one workgroup (four chunks, 100 KB of LDS) per CU
    const int64_t max_ns = (R + 63) / 64;
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    int64_t c = (R + ns - 1) / ns;
    c = (c + 7) & ~(int64_t)7;
    *kc = c;
    *nsplit = (int)((R + c - 1) / c);
}

int lds_opt_in(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return CDR_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { cdr_set_error("cdr_conet: %zu B of LDS refused: %s", bytes, hipGetErrorString(e)); return (int)e; }
    return CDR_OK;
}

}  // namespace

#ifdef CDR_CONET_PROF
extern "C" int cdr_conet_debug_prof(long long* dev_buf) {
    CDR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conet_prof), &dev_buf, sizeof(dev_buf)));
    return CDR_OK;
}
#endif

extern "C" int cdr_conet_plan(int L, const int* dims, int64_t R, int* act_width, size_t* workspace_bytes) {
    CDR_CHECK_ARG(dims && act_width && workspace_bytes && R > 0);
    conet_net net;
    lds_plan lp;
    const float* dummy[5 * kMaxL + 4];
    for (auto& p : dummy) p = (const float*)16;
    if (!fill_net(net, lp, L, dims, dummy)) { cdr_set_error("cdr_conet_plan: unsupported layer sizes"); return CDR_EINVAL; }
    conet_tiles tl;
    fill_tiles(net, tl);
    int nsplit; int64_t kc;
    split_plan(tl.ntiles, R, &nsplit, &kc);
    *act_width = net.act_off[L];
    const size_t wpart = (size_t)tl.ntiles * nsplit * kJobFloats * sizeof(float);
    const size_t ou = (size_t)rows_grid(R) * 2 * (dims[L] + 1) * sizeof(float);
    *workspace_bytes = ((wpart + 255) & ~(size_t)255) + ou;
    return CDR_OK;
}

// stage_weights_dma's per-chunk source addresses (its own arithmetic, on the host), kept in the context and re-sent only when they change
// (a parameter was re-allocated).  Returns the device table, or null: the kernel then computes the addresses itself -- the table does not
// cover the area, allocation failed, or the table would have to change while the stream is capturing.
static const unsigned long long* conet_dma_table(cdr_ctx* ctx, const conet_net& net, int threads, hipStream_t s) {
    const int total = net.wl_off[net.L] >> 2;
    if (!net.wlds || total <= 0 || total > 8 * threads) return nullptr;
    std::vector<unsigned long long> tab((size_t)total);
    for (int c = 0; c < total; ++c) {
        const float* src = net.Ws[1];
        for (int l = 1; l < net.L; ++l) {
            const int din = net.dims[l], dout = net.dims[l + 1], q1 = (din >> 2) + 1, n1 = dout * q1;
            const int r3 = c - (net.wl_off[l] >> 2);
            if (r3 >= 0 && r3 < 3 * n1) {
                const int mat = r3 / n1, r = r3 - mat * n1, row = r / q1, col = r - row * q1;
                src = (mat == 0 ? net.Ws[l] : mat == 1 ? net.Wt[l] : net.H[l]) + (int64_t)row * din + 4 * (col < q1 - 1 ? col : 0);
            }
        }
        tab[c] = (unsigned long long)(uintptr_t)src;
    }
    if (ctx->conet_tab_n == total && ctx->conet_tab_shadow && memcmp(ctx->conet_tab_shadow, tab.data(), (size_t)total * 8) == 0)
        return ctx->conet_tab_dev;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    if (ctx->conet_tab_cap < total) {
        if (ctx->conet_tab_dev) (void)hipFree(ctx->conet_tab_dev);
        if (ctx->conet_tab_pin) (void)hipHostFree(ctx->conet_tab_pin);
        free(ctx->conet_tab_shadow);
        ctx->conet_tab_dev = ctx->conet_tab_pin = ctx->conet_tab_shadow = nullptr;
        ctx->conet_tab_cap = ctx->conet_tab_n = 0;
        const size_t bytes = (size_t)total * 8;
        if (hipMalloc((void**)&ctx->conet_tab_dev, bytes) != hipSuccess || hipHostMalloc((void**)&ctx->conet_tab_pin, bytes) != hipSuccess ||
            !(ctx->conet_tab_shadow = (unsigned long long*)malloc(bytes))) {
            (void)hipGetLastError();
            return nullptr;
        }
        ctx->conet_tab_cap = total;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;         // (an earlier copy out of the pinned buffer; tables change once in a run)
    memcpy(ctx->conet_tab_pin, tab.data(), (size_t)total * 8);
    if (hipMemcpyAsync(ctx->conet_tab_dev, ctx->conet_tab_pin, (size_t)total * 8, hipMemcpyHostToDevice, s) != hipSuccess) { ctx->conet_tab_n = 0; return nullptr; }
    memcpy(ctx->conet_tab_shadow, tab.data(), (size_t)total * 8);
    ctx->conet_tab_n = total;
    return ctx->conet_tab_dev;
}

extern "C" int cdr_conet_defer_finish(cdr_ctx* ctx, int on) {
    CDR_CHECK_ARG(ctx);
    ctx->conet_defer = on ? 1 : 0;
    return CDR_OK;
}

extern "C" int cdr_conet_fwd(cdr_ctx* ctx, void* stream, const float* su_tab, const float* si_tab, const float* tu_tab,
                             const float* ti_tab, int D, const int64_t* user_s, const int64_t* user_t, const int64_t* item_s,
                             const int64_t* item_t, int64_t R, int64_t n_source,
                             int64_t n_overlap, int overlap_users, int L, const int* dims, const float* const* params,
                             const float* label_s, const float* label_t, float* x0, float* acts, float* prob, float* maskf,
                             float* label_cat, int64_t* ids_cat, float* out, float* gz, float* gx0, void* workspace,
                             size_t workspace_bytes, int* data_gradients_done) {
    CDR_CHECK_ARG(ctx && su_tab && si_tab && tu_tab && ti_tab && x0 && acts && prob && maskf && label_cat && ids_cat && out);
    CDR_CHECK_ARG((gz == nullptr) == (gx0 == nullptr) && (gz == nullptr || (workspace && data_gradients_done)));
    if (data_gradients_done) *data_gradients_done = 0;
    CDR_CHECK_ARG(R > 0 && n_source >= 0 && n_source <= R && D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG((n_source == 0 || (user_s && item_s && label_s)) && (n_source == R || (user_t && item_t && label_t)));
    conet_net net;
    lds_plan lp;
    if (!fill_net(net, lp, L, dims, params) || dims[0] != 2 * D) { cdr_set_error("cdr_conet_fwd: unsupported layer sizes"); return CDR_EINVAL; }
    int rc = lds_opt_in((const void*)conet_fwd_kernel, lp.fwd_bytes);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)rows_grid(R);
    if (ctx->conet_pending) {                    // a deferred forward whose backward never came: its loss is added now, before its partials go
        ctx->conet_pending = 0;
        conet_fwd_finish_kernel<<<dim3(1), dim3(256), 0, (hipStream_t)ctx->conet_fin_stream>>>(net, ctx->partials, ctx->conet_fin_grid, ctx->conet_fin_ns,
                                                                                              ctx->conet_fin_R, ctx->conet_fin_out);
        CDR_LAUNCH_CHECK();
    }
    if (gz && lp.fb_bytes) {                     // training step: forward and data backward of every 32-row block in one launch
        for (int l = 0; l < L; ++l) net.hsq_chunk[l] = (dims[l] * dims[l + 1] + grid - 1) / grid;
        int aw = 0; size_t need = 0;
        rc = cdr_conet_plan(L, dims, R, &aw, &need);
        if (rc) return rc;
        CDR_CHECK_ARG(workspace_bytes >= need);
        conet_tiles tl;
        fill_tiles(net, tl);
        int nsplit; int64_t kc;
        split_plan(tl.ntiles, R, &nsplit, &kc);
        const size_t wpart_bytes = ((size_t)tl.ntiles * nsplit * kJobFloats * sizeof(float) + 255) & ~(size_t)255;
        float* ou_part = (float*)((char*)workspace + wpart_bytes);
        const char* fbw = getenv("CDR_CONET_FB_WAVES");               // "4": the four-wave kernel (A/B runs and the bit-equality test only)
        const bool waves4 = fbw && atoi(fbw) == 4;
        const bool eight = lp.fb8 && !waves4;
        net.dma_tab = getenv("CDR_CONET_NO_DMA_TABLE") ? nullptr : conet_dma_table(ctx, net, eight ? 512 : 256, s);
        rc = lds_opt_in(eight ? (const void*)conet_fb_kernel<8> : (const void*)conet_fb_kernel<4>, lp.fb_bytes);
        if (rc) return rc;
        {
            cdr_time_scope ts(ctx, CDR_TAG_CONET_FB, s);
            if (eight)
                conet_fb_kernel<8><<<dim3(grid), dim3(512), lp.fb_bytes, s>>>(net, lp.fb, su_tab, si_tab, tu_tab, ti_tab, D, user_s, user_t, item_s,
                                                                              item_t, R, n_source, n_overlap, overlap_users, label_s, label_t,
                                                                              label_cat, ids_cat, x0, acts, prob, maskf, ctx->partials, gz, gx0,
                                                                              ou_part);
            else
                conet_fb_kernel<4><<<dim3(grid), dim3(256), lp.fb_bytes, s>>>(net, lp.fb, su_tab, si_tab, tu_tab, ti_tab, D, user_s, user_t, item_s,
                                                                              item_t, R, n_source, n_overlap, overlap_users, label_s, label_t,
                                                                              label_cat, ids_cat, x0, acts, prob, maskf, ctx->partials, gz, gx0,
                                                                              ou_part);
        }
        CDR_LAUNCH_CHECK();
        if (ctx->conet_defer) {                  // cdr_conet_bwd's weight-gradient launch adds the partials
            ctx->conet_pending = 1; ctx->conet_fin_grid = grid; ctx->conet_fin_ns = n_source; ctx->conet_fin_R = R; ctx->conet_fin_out = out;
            ctx->conet_fin_stream = stream;
        } else {
            conet_fwd_finish_kernel<<<dim3(1), dim3(256), 0, s>>>(net, ctx->partials, grid, n_source, R, out);
            CDR_LAUNCH_CHECK();
        }
        *data_gradients_done = 1;
        return CDR_OK;
    }
    {
        cdr_time_scope ts(ctx, CDR_TAG_CONET_FWD, s);
        conet_fwd_kernel<<<dim3(grid), dim3(256), lp.fwd_bytes, s>>>(net, su_tab, si_tab, tu_tab, ti_tab, D, user_s, user_t, item_s, item_t,
                                                                     R, n_source, n_overlap, overlap_users, label_s, label_t, label_cat,
                                                                     ids_cat, lp.strideA, lp.strideB, x0, acts, prob, maskf, ctx->partials);
    }
    CDR_LAUNCH_CHECK();
    conet_fwd_finish_kernel<<<dim3(1), dim3(256), 0, s>>>(net, ctx->partials, grid, n_source, R, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_conet_bwd(cdr_ctx* ctx, void* stream, int64_t R, int64_t n_source, int L, const int* dims,
                             const float* const* params, const float* label, const float* x0, const float* acts, const float* prob,
                             const float* maskf, const float* out, const float* grad_out, float* gz, float* gx0,
                             float* const* grads, void* workspace, size_t workspace_bytes, int data_gradients_done) {
    CDR_CHECK_ARG(ctx && label && x0 && acts && prob && maskf && out && gz && gx0 && grads && workspace);
    CDR_CHECK_ARG(R > 0 && n_source >= 0 && n_source <= R);
    conet_net net;
    lds_plan lp;
    if (!fill_net(net, lp, L, dims, params)) { cdr_set_error("cdr_conet_bwd: unsupported layer sizes"); return CDR_EINVAL; }
    int aw = 0; size_t need = 0;
    int rc = cdr_conet_plan(L, dims, R, &aw, &need);
    if (rc) return rc;
    CDR_CHECK_ARG(workspace_bytes >= need);
    conet_grads gr;
    memset(&gr, 0, sizeof(gr));
    for (int l = 0; l < L; ++l) {
        gr.Ws[l] = grads[5 * l]; gr.bs[l] = grads[5 * l + 1]; gr.Wt[l] = grads[5 * l + 2]; gr.bt[l] = grads[5 * l + 3]; gr.H[l] = grads[5 * l + 4];
        CDR_CHECK_ARG(gr.Ws[l] && gr.bs[l] && gr.Wt[l] && gr.bt[l] && gr.H[l]);
    }
    gr.wo[0] = grads[5 * L]; gr.bo[0] = grads[5 * L + 1]; gr.wo[1] = grads[5 * L + 2]; gr.bo[1] = grads[5 * L + 3];
    CDR_CHECK_ARG(gr.wo[0] && gr.bo[0] && gr.wo[1] && gr.bo[1]);
    conet_tiles tl;
    fill_tiles(net, tl);
    int nsplit; int64_t kc;
    split_plan(tl.ntiles, R, &nsplit, &kc);
    float* wpart = (float*)workspace;
    const size_t wpart_bytes = ((size_t)tl.ntiles * nsplit * kJobFloats * sizeof(float) + 255) & ~(size_t)255;
    float* ou_part = (float*)((char*)workspace + wpart_bytes);
    const int ngroup = (nsplit + 3) / 4;                      // a workgroup adds four chunks before writing: partials per job
    rc = lds_opt_in((const void*)conet_bwd_kernel, lp.bwd_bytes);
    if (rc) return rc;
    rc = lds_opt_in((const void*)conet_wgrad_kernel, 4 * kJobFloats * sizeof(float));
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)rows_grid(R);
    if (!data_gradients_done) {
        cdr_time_scope ts(ctx, CDR_TAG_CONET_BWD, s);
        conet_bwd_kernel<<<dim3(grid), dim3(256), lp.bwd_bytes, s>>>(net, R, n_source, label, prob, maskf, acts, grad_out, lp.strideG0,
                                                                     lp.strideG1, gz, gx0, ou_part);
    }
    CDR_LAUNCH_CHECK();
    conet_fin fin{nullptr, nullptr, 0, 0};
    if (ctx->conet_pending) {
        if (ctx->conet_fin_out != out || ctx->conet_fin_R != R || ctx->conet_fin_stream != stream) {
            cdr_set_error("cdr_conet_bwd: a deferred forward (cdr_conet_defer_finish) is pending for another loss / stream");
            return CDR_EINVAL;
        }
        fin = conet_fin{ctx->partials, ctx->conet_fin_out, ctx->conet_fin_grid, ctx->conet_fin_ns};
        ctx->conet_pending = 0;
    }
    {
        cdr_time_scope ts(ctx, CDR_TAG_CONET_WGRAD, s);
        conet_wgrad_kernel<<<dim3((unsigned)(tl.ntiles * ngroup + (fin.out ? 1 : 0))), dim3(256), 4 * kJobFloats * sizeof(float), s>>>(
            net, tl, R, kc, x0, acts, gz, maskf, wpart, fin);
    }
    CDR_LAUNCH_CHECK();
    conet_wgrad_finish_kernel<<<dim3((unsigned)(tl.ntiles * ((kJobFloats + 255) / 256) + 1)), dim3(256), 0, s>>>(
        net, gr, tl, ngroup, wpart, ou_part, grid, out, grad_out, data_gradients_done ? gx0 : nullptr,
        data_gradients_done ? R * 2 * (int64_t)dims[0] : 0);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
