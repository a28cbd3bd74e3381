// Fused row-wise training step for tables too large for dense gradients / dense Adam (BASELINE config C5):
//
//   cdr_bpr_fwd_grad   one pass over the batch: gather 3 rows per triple, loss partials, and the two compact gradient
//                      rows every other row gradient is made of:  GU[b] = g_b (p - n),  GP[b] = g_b u
//                      (dI[pid_b] += GP[b], dI[nid_b] -= GP[b]).  Nothing table-sized is ever written.
//   cdr_sort_ids       stable LSD radix sort (rocPRIM) of the touched row ids with their occurrence index, over only the
//                      significant key bits -> every table row's occurrences become one contiguous segment.
//   cdr_rowwise_apply  one lane-group per segment head: sums the segment's gradient rows in occurrence order (fixed
//                      order => deterministic, no float atomics), adds the EmbLoss term count * c * W[r], and applies
//                      the optimizer (SGD, or Adam with per-row moments) in place: one read-modify-write per touched row.
//
// Semantics: exactly "sum of the batch's gradients evaluated at the pre-step weights", as autograd + a dense optimizer
// would compute for the touched rows.  Rows not in the batch are not touched (for Adam this is the usual lazy/sparse
// variant: untouched rows keep their moments and do not move) -- documented in DESIGN.md as the one deliberate
// difference from the reference's dense torch.optim.Adam, which is O(table) per step.
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "cdr_common.h"
#include "cdr_adam_math.h"

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

// ------------------------------------------------------------------------------------------------ forward + compact grads
// SCATTER (row-sharded step): the item operand is the buffer of received rows, one entry per occurrence, and the item
// gradients are written straight into the send buffer at the same positions: GP[pid[t]] = g u, GP[nid[t]] = -g u.
template <int LPR, bool SCATTER>
__global__ __launch_bounds__(kBlock) void bpr_fwd_grad_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                              int D, const int64_t* __restrict__ uid,
                                                              const int64_t* __restrict__ pid,
                                                              const int64_t* __restrict__ nid, int64_t B, float gamma,
                                                              float invB, float* __restrict__ GU,
                                                              float* __restrict__ GP, double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool live = sub < D4;

    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        float4 u[kUnroll], p[kUnroll], n[kUnroll];
        // ids of all kUnroll triples first, THEN all row loads: vmcnt is an in-order counter, so interleaving
        // "ids(r) -> rows(r)" makes the wait for ids(r+1) also wait for rows(r) and serialises the gathers
        int64_t iu[kUnroll], ip[kUnroll], in[kUnroll];
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
                u[r] = ld4n<(LPR >= 32)>(U + iu[r] * D + 4 * sub);
                p[r] = ld4n<(LPR >= 32)>(I + ip[r] * D + 4 * sub);
                n[r] = ld4n<(LPR >= 32)>(I + in[r] * D + 4 * sub);
            }
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const float dp = group_sum<LPR>(dot4(u[r], p[r]));
            const float dn = group_sum<LPR>(dot4(u[r], n[r]));
            const float su = group_sum<LPR>(dot4(u[r], u[r]));
            const float sp = group_sum<LPR>(dot4(p[r], p[r]));
            if (t < B) {
                const float s = sigmoidf_(dp - dn);
                const float g = -invB * (s * (1.0f - s)) / (gamma + s);
                if (live) {
                    st4n<(LPR >= 32)>(GU + t * D + 4 * sub, make_float4(g * (p[r].x - n[r].x), g * (p[r].y - n[r].y),
                                                          g * (p[r].z - n[r].z), g * (p[r].w - n[r].w)));
                    if (SCATTER) {
                        st4n<(LPR >= 32)>(GP + ip[r] * D + 4 * sub, make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w));
                        st4n<(LPR >= 32)>(GP + in[r] * D + 4 * sub, make_float4(-g * u[r].x, -g * u[r].y, -g * u[r].z, -g * u[r].w));
                    } else {
                        st4n<(LPR >= 32)>(GP + t * D + 4 * sub, make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w));
                    }
                }
                if (sub == 0) {
                    acc[0] += (double)(-logf(gamma + s));
                    acc[1] += (double)su;
                    acc[2] += (double)sp;
                }
            }
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// Pointwise counterpart (EMCDR's default MF latent factor model: emcdr.py:111-122, MSE on the raw dot; BCE on sigmoid(dot) as in
// cmf.py:75-99): one pass over the batch's (user, item, label) rows; compact gradient rows GU[b] = g_b i_b, GI[b] = g_b u_b.
template <int LPR>
__global__ __launch_bounds__(kBlock) void point_fwd_grad_kernel(int loss_kind, const float* __restrict__ U, const float* __restrict__ I,
                                                                int D, const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                                const float* __restrict__ label, int64_t B, float invB,
                                                                float* __restrict__ GU, float* __restrict__ GI,
                                                                double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool live = sub < D4;
    for (int64_t base = gg; base < B; base += TG * kUnroll) {
        float4 u[kUnroll], v[kUnroll];
        int64_t iu[kUnroll], ii[kUnroll];
        float yl[kUnroll];
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {                  // ids and labels first, then every row load (vmcnt is in-order)
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ii[r] = iid[tc]; yl[r] = label[tc];
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            u[r] = v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < B && live) {
                u[r] = ld4n<(LPR >= 32)>(U + iu[r] * D + 4 * sub);
                v[r] = ld4n<(LPR >= 32)>(I + ii[r] * D + 4 * sub);
            }
        }
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const float dx = group_sum<LPR>(dot4(u[r], v[r]));
            const float su = group_sum<LPR>(dot4(u[r], u[r]));
            const float si = group_sum<LPR>(dot4(v[r], v[r]));
            if (t < B) {
                const float y = yl[r];
                float l, g;
                if (loss_kind == CDR_LOSS_MSE) {
                    const float d = dx - y;
                    l = d * d; g = 2.0f * d * invB;
                } else {                                       // torch BCELoss on sigmoid(dot): -100 log clamp, 1e-12 backward clamp
                    const float p = sigmoidf_(dx);
                    l = (y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f);
                    const float pq = (1.0f - p) * p;
                    g = (p - y) / fmaxf(pq, 1e-12f) * invB * pq;
                }
                if (live) {
                    st4n<(LPR >= 32)>(GU + t * D + 4 * sub, make_float4(g * v[r].x, g * v[r].y, g * v[r].z, g * v[r].w));
                    st4n<(LPR >= 32)>(GI + t * D + 4 * sub, make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w));
                }
                if (sub == 0) { acc[0] += (double)l; acc[1] += (double)su; acc[2] += (double)si; }
            }
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// out9 = {total, main, ||U_b||, ||I_b||, c_u, c_i, sum loss, sum u^2, sum p^2} with c = reg_weight / (B * norm)
// (0 when the norm is 0).  B is the batch size the mean and the EmbLoss are taken over (the GLOBAL batch when the
// step is sharded: then out9[0..5] are provisional and cdr_loss_finish_sums recomputes them from all-reduced sums).
__global__ __launch_bounds__(kBlock) void step_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t B,
                                                             float reg_weight, float* __restrict__ out6) {
    __shared__ double smem[3 * (kBlock / 64)];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2];
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        const float main_loss = (float)(acc[0] / (double)B);
        const float nu = (float)sqrt(acc[1]), ni = (float)sqrt(acc[2]);
        out6[1] = main_loss; out6[2] = nu; out6[3] = ni;
        out6[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
        out6[4] = (reg_weight != 0.f && nu > 0.f) ? reg_weight / ((float)B * nu) : 0.f;
        out6[5] = (reg_weight != 0.f && ni > 0.f) ? reg_weight / ((float)B * ni) : 0.f;
        out6[6] = (float)acc[0]; out6[7] = (float)acc[1]; out6[8] = (float)acc[2];
    }
}

__global__ void finish_sums_kernel(const float* __restrict__ sums3, int64_t B, float reg_weight, float* __restrict__ out6) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float main_loss = sums3[0] / (float)B;
        const float nu = sqrtf(sums3[1]), ni = sqrtf(sums3[2]);
        out6[1] = main_loss; out6[2] = nu; out6[3] = ni;
        out6[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
        out6[4] = (reg_weight != 0.f && nu > 0.f) ? reg_weight / ((float)B * nu) : 0.f;
        out6[5] = (reg_weight != 0.f && ni > 0.f) ? reg_weight / ((float)B * ni) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ keys for the sort
__global__ __launch_bounds__(kBlock) void make_keys_kernel(const int64_t* __restrict__ ids0, int64_t n0,
                                                           const int64_t* __restrict__ ids1, int64_t n1,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t n = n0 + n1, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        keys[e] = (uint32_t)(e < n0 ? ids0[e] : ids1[e - n0]);
        vals[e] = (uint32_t)e;
    }
}

// ------------------------------------------------------------------------------------------------ segmented apply
// OPT 0: SGD  w -= lr * grad            OPT 1: Adam on the touched rows (torch.optim.Adam arithmetic per element)
//
// Segments up to kLongSeg occurrences are summed by their head's lane group (the common case: a batch of 1-2 M ids over
// 10-50 M rows has segments of 1-5).  Skewed id streams (Zipf item popularity, SURVEY 8d synthetic inputs (ii)) put tens
// of thousands of occurrences into one segment; walking those with one lane group took 52 ms per step.  A head that
// sees a long segment therefore only registers it: the segment is cut into pieces of kPiece occurrences, every piece is
// summed by its own lane group (seg_piece_sum_kernel), and a finishing kernel adds the piece sums in piece order and
// applies the optimizer.  All three levels add in a fixed order: results stay bit-reproducible.
constexpr int kLongSeg = 32;
constexpr int kPiece = 256;

struct seg_long { int64_t head, len, base; };           // sorted position of the head, occurrences, first piece index
struct seg_piece { int64_t start; int64_t len; };

// dev (optional): {step_size, bc2_sqrt} in DEVICE memory, written earlier on the stream from device-resident update counts
// (coef_finish_kernel) -- the capturable form of the step: a hipGraph replay must not bake the update number into its launches.
struct apply_hp { float lr, b1, b2, eps, wd, step_size, bc2_sqrt; const float* dev; };
#define HP_FROM_DEV(h) do { if ((h).dev) { (h).step_size = (h).dev[0]; (h).bc2_sqrt = (h).dev[1]; } } while (0)

template <int LPR, int OPT>
__device__ __forceinline__ void apply_update(float* __restrict__ wp, float* __restrict__ mp, float* __restrict__ vp, float4 w,
                                             float4 acc, float rc, const apply_hp& h) {
    float4 gr = make_float4(acc.x + rc * w.x, acc.y + rc * w.y, acc.z + rc * w.z, acc.w + rc * w.w);
    float4 wn;
    if (OPT == 0) {
        if (h.wd != 0.f) { gr.x += h.wd * w.x; gr.y += h.wd * w.y; gr.z += h.wd * w.z; gr.w += h.wd * w.w; }
        wn = make_float4(w.x - h.lr * gr.x, w.y - h.lr * gr.y, w.z - h.lr * gr.z, w.w - h.lr * gr.w);
    } else {
        float4 m = ld4n<(LPR >= 32)>(mp), v = ld4n<(LPR >= 32)>(vp);
        if (h.wd != 0.f) { gr.x += h.wd * w.x; gr.y += h.wd * w.y; gr.z += h.wd * w.z; gr.w += h.wd * w.w; }
        m.x += (gr.x - m.x) * (1.0f - h.b1); m.y += (gr.y - m.y) * (1.0f - h.b1);
        m.z += (gr.z - m.z) * (1.0f - h.b1); m.w += (gr.w - m.w) * (1.0f - h.b1);
        v.x = h.b2 * v.x + (1.0f - h.b2) * gr.x * gr.x; v.y = h.b2 * v.y + (1.0f - h.b2) * gr.y * gr.y;
        v.z = h.b2 * v.z + (1.0f - h.b2) * gr.z * gr.z; v.w = h.b2 * v.w + (1.0f - h.b2) * gr.w * gr.w;
        st4n<(LPR >= 32)>(mp, m); st4n<(LPR >= 32)>(vp, v);
        wn = make_float4(w.x - cdr_adam_term(m.x, v.x, h.step_size, h.bc2_sqrt, h.eps), w.y - cdr_adam_term(m.y, v.y, h.step_size, h.bc2_sqrt, h.eps),
                         w.z - cdr_adam_term(m.z, v.z, h.step_size, h.bc2_sqrt, h.eps), w.w - cdr_adam_term(m.w, v.w, h.step_size, h.bc2_sqrt, h.eps));
    }
    st4n<(LPR >= 32)>(wp, wn);
}

template <int LPR, int OPT, bool SIGNED>
__global__ __launch_bounds__(kBlock) void rowwise_apply_kernel(float* __restrict__ W, float* __restrict__ Mo,
                                                               float* __restrict__ Vo, int D,
                                                               const uint32_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ perm, int64_t n,
                                                               const float* __restrict__ G, int64_t neg_start,
                                                               int64_t reg_limit, const float* __restrict__ reg_coef,
                                                               apply_hp hp, const int64_t* __restrict__ occ_ids,
                                                               unsigned* __restrict__ counters, seg_long* __restrict__ longs,
                                                               seg_piece* __restrict__ pieces) {
    HP_FROM_DEV(hp);
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    for (int64_t q = gg; q < n; q += TG) {
        // three independent probes, issued together (prefetching them one iteration ahead was measured slower: the copies
        // of the prefetched registers at the loop's back edge wait for this iteration's row stores)
        const uint32_t row = keys[q];
        const uint32_t before = keys[q > 0 ? q - 1 : 0];
        const uint32_t far = keys[q + kLongSeg < n ? q + kLongSeg : n - 1];
        const bool head = !(q > 0 && before == row);         // uniform inside the lane group
        const bool is_long = q + kLongSeg < n && far == row;  // registered below, summed by the piece kernels
        if (head && !is_long) {
            for (int ch = sub; ch < D4; ch += LPR) {
                float* wp = W + (int64_t)row * D + 4 * ch;
                const float4 w = ld4n<(LPR >= 32)>(wp);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                int cnt = 0;
                for (int64_t e = q; e < n && keys[e] == row; ++e) {
                    const int64_t o = perm[e];
                    const bool neg = SIGNED && o >= neg_start;
                    const float4 g = ld4n<(LPR >= 32)>(G + (neg ? o - neg_start : o) * D + 4 * ch);
                    if (neg) { acc.x -= g.x; acc.y -= g.y; acc.z -= g.z; acc.w -= g.w; }
                    else { acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w; }
                    cnt += occ_ids ? (int)((occ_ids[o] >> 62) & 1) : ((o < reg_limit) ? 1 : 0);
                }
                apply_update<LPR, OPT>(wp, OPT ? Mo + (int64_t)row * D + 4 * ch : nullptr, OPT ? Vo + (int64_t)row * D + 4 * ch : nullptr,
                                  w, acc, c * (float)cnt, hp);
            }
        }
    }
    // ---- registration of the long segments, one THREAD per sorted position.  Kept out of the loop above on purpose: with
    // the atomics inside it hipcc put an s_waitcnt vmcnt(0) on the loop's back edge (every iteration then waited for its
    // own row stores before issuing the next key loads: +25 % on the whole kernel).
    if (counters == nullptr) return;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q + kLongSeg < n; q += (int64_t)gridDim.x * kBlock) {
        const uint32_t row = keys[q];
        const uint32_t before = keys[q > 0 ? q - 1 : 0];
        const uint32_t far = keys[q + kLongSeg];
        if ((q > 0 && before == row) || far != row) continue;
        int64_t lo = q + kLongSeg, hi = n;                   // keys[lo] == row, keys[hi] != row (or hi == n): keys are sorted
        while (lo + 1 < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] == row) lo = mid; else hi = mid;
        }
        const int64_t len = hi - q;
        const unsigned np = (unsigned)((len + kPiece - 1) / kPiece);
        const unsigned base = atomicAdd(&counters[0], np);
        const unsigned li = atomicAdd(&counters[1], 1u);
        longs[li] = seg_long{q, len, (int64_t)base};
        for (unsigned k = 0; k < np; ++k) {
            const int64_t st = q + (int64_t)k * kPiece;
            pieces[base + k] = seg_piece{st, (hi - st) < kPiece ? (hi - st) : (int64_t)kPiece};
        }
    }
}

// One WORKGROUP per piece of a long segment (round 6; one lane group per piece before): the piece's kPiece = 256 occurrences are cut into
// eight sub-pieces of kSubPiece = 32; the workgroup's lane groups sum one sub-piece each in occurrence order (sixteen row loads in flight: two
// rounds), park the eight sub-sums in LDS, and the first lane group adds them in sub-piece order: partial[pi] = ((s0 + s1) + ...) + s7, a fixed
// association that does not depend on the launch shape.  This is north_star's "LDS staging of the hot rows" where it pays: a hot item's
// segment (4,068 occurrences at Zipf(1.05), B = 65,536) was 16 pieces of 16 dependent rounds each -- 44.8 us of the step's 300
// (profiles/r06_zipf65k_kernel_stats_before.csv); the staged sub-sums make it two rounds and one pass through LDS.  pcnt[pi] = the piece's
// EmbLoss occurrences.  Every long-segment path of this file (two-pass apply, fused duplicate apply, row-shard owner and requester) runs this
// body, so they keep agreeing bit for bit with each other.
constexpr int kSubPiece = 32, kSubs = kPiece / kSubPiece;
template <int LPR, bool SIGNED>
__device__ __forceinline__ void seg_piece_sum_body(int D, const uint32_t* __restrict__ perm,
                                                   const float* __restrict__ G, int64_t neg_start, int64_t reg_limit,
                                                   const int64_t* __restrict__ occ_ids,
                                                   const unsigned* __restrict__ counters,
                                                   const seg_piece* __restrict__ pieces, float* __restrict__ partial,
                                                   int* __restrict__ pcnt) {
    constexpr int GPB = kBlock / LPR;
    constexpr int UN = 16;
    __shared__ float4 sub_sum[kSubs][64];                     // one row of <= 256 floats per sub-piece (wider rows: chunk by chunk)
    __shared__ int sub_cnt[kSubs];
    const int sub = threadIdx.x % LPR, gid = threadIdx.x / LPR;
    const int D4 = D >> 2;
    const int64_t np = counters[0];
    for (int64_t pi = blockIdx.x; pi < np; pi += gridDim.x) {
        const seg_piece pc = pieces[pi];
        for (int ch0 = 0; ch0 < D4; ch0 += 64) {              // 64 float4 chunks of the row at a time (one trip for D <= 256)
            for (int sp = gid; sp < kSubs; sp += GPB) {
                const int64_t s0 = (int64_t)sp * kSubPiece, slen = pc.len - s0 < kSubPiece ? pc.len - s0 : (int64_t)kSubPiece;
                int cnt = 0;
                for (int ch = ch0 + sub; ch < D4 && ch < ch0 + 64; ch += LPR) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    cnt = 0;
                    for (int64_t e0 = 0; e0 < slen; e0 += UN) {
                        int64_t o[UN]; float4 g[UN]; bool neg[UN];
#pragma unroll
                        for (int j = 0; j < UN; ++j) o[j] = (e0 + j < slen) ? (int64_t)perm[pc.start + s0 + e0 + j] : -1;
#pragma unroll
                        for (int j = 0; j < UN; ++j) {
                            neg[j] = SIGNED && o[j] >= neg_start;
                            g[j] = o[j] >= 0 ? ld4n<(LPR >= 32)>(G + (neg[j] ? o[j] - neg_start : o[j]) * D + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int j = 0; j < UN; ++j) {
                            if (o[j] < 0) continue;
                            if (neg[j]) { acc.x -= g[j].x; acc.y -= g[j].y; acc.z -= g[j].z; acc.w -= g[j].w; }
                            else { acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; acc.w += g[j].w; }
                            cnt += occ_ids ? (int)((occ_ids[o[j]] >> 62) & 1) : ((o[j] < reg_limit) ? 1 : 0);
                        }
                    }
                    sub_sum[sp][ch - ch0] = acc;
                }
                if (sub == 0 && ch0 == 0) sub_cnt[sp] = slen > 0 ? cnt : 0;
            }
            __syncthreads();
            const int nsub = (int)((pc.len + kSubPiece - 1) / kSubPiece);
            for (int c = threadIdx.x; c < 64 && ch0 + c < D4; c += kBlock) {        // sub-piece order: the piece's sum
                float4 acc = sub_sum[0][c];
                for (int sp = 1; sp < nsub; ++sp) { const float4 x = sub_sum[sp][c]; acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w; }
                st4n<(LPR >= 32)>(partial + pi * D + 4 * (ch0 + c), acc);
            }
            if (threadIdx.x == 0 && ch0 == 0) {
                int cn = 0;
                for (int sp = 0; sp < nsub; ++sp) cn += sub_cnt[sp];
                pcnt[pi] = cn;
            }
            __syncthreads();
        }
    }
}
template <int LPR, bool SIGNED>
__global__ __launch_bounds__(kBlock) void seg_piece_sum_kernel(int D, const uint32_t* __restrict__ perm,
                                                               const float* __restrict__ G, int64_t neg_start, int64_t reg_limit,
                                                               const int64_t* __restrict__ occ_ids,
                                                               const unsigned* __restrict__ counters,
                                                               const seg_piece* __restrict__ pieces, float* __restrict__ partial,
                                                               int* __restrict__ pcnt) {
    seg_piece_sum_body<LPR, SIGNED>(D, perm, G, neg_start, reg_limit, occ_ids, counters, pieces, partial, pcnt);
}

// One lane group per long segment: piece sums added in piece order, then the same update as the head-only path.
template <int LPR, int OPT>
__device__ __forceinline__ void seg_long_finish_body(float* __restrict__ W, float* __restrict__ Mo, float* __restrict__ Vo,
                                                     int D, const uint32_t* __restrict__ keys,
                                                     const float* __restrict__ reg_coef, apply_hp hp,
                                                     const unsigned* __restrict__ counters,
                                                     const seg_long* __restrict__ longs,
                                                     const float* __restrict__ partial, const int* __restrict__ pcnt) {
    HP_FROM_DEV(hp);
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    const int64_t nl = counters[1];
    for (int64_t li = gg; li < nl; li += TG) {
        const seg_long sg = longs[li];
        const uint32_t row = keys[sg.head];
        const int64_t np = (sg.len + kPiece - 1) / kPiece;
        for (int ch = sub; ch < D4; ch += LPR) {
            float* wp = W + (int64_t)row * D + 4 * ch;
            const float4 w = ld4n<(LPR >= 32)>(wp);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnt = 0;
            for (int64_t k0 = 0; k0 < np; k0 += 8) {                  // eight piece sums in flight, added in piece order
                float4 g[8]; int c[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = k0 + j < np;
                    g[j] = in ? ld4n<(LPR >= 32)>(partial + (sg.base + k0 + j) * D + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    c[j] = in ? pcnt[sg.base + k0 + j] : 0;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j < np) { acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; acc.w += g[j].w; cnt += c[j]; }
                }
            }
            apply_update<LPR, OPT>(wp, OPT ? Mo + (int64_t)row * D + 4 * ch : nullptr, OPT ? Vo + (int64_t)row * D + 4 * ch : nullptr,
                              w, acc, c * (float)cnt, hp);
        }
    }
}
template <int LPR, int OPT>
__global__ __launch_bounds__(kBlock) void seg_long_finish_kernel(float* __restrict__ W, float* __restrict__ Mo, float* __restrict__ Vo,
                                                                 int D, const uint32_t* __restrict__ keys,
                                                                 const float* __restrict__ reg_coef, apply_hp hp,
                                                                 const unsigned* __restrict__ counters,
                                                                 const seg_long* __restrict__ longs,
                                                                 const float* __restrict__ partial, const int* __restrict__ pcnt) {
    seg_long_finish_body<LPR, OPT>(W, Mo, Vo, D, keys, reg_coef, hp, counters, longs, partial, pcnt);
}

// Two tables in ONE sort: rocPRIM's radix sort costs ~0.16 ms whether it sorts 1 M or 3 M pairs (a chain of small launches),
// so the user ids and the item ids of a step are sorted together; the item keys carry one extra high bit (key_base), which
// puts them behind every user key and is subtracted again by the apply kernel.
__global__ __launch_bounds__(kBlock) void make_keys2_kernel(const int64_t* __restrict__ a, int64_t na, const int64_t* __restrict__ b0,
                                                            int64_t nb0, const int64_t* __restrict__ b1, int64_t nb1, uint32_t key_base,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t n = na + nb0 + nb1, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        if (e < na) { keys[e] = (uint32_t)a[e]; vals[e] = (uint32_t)e; }
        else {
            const int64_t o = e - na;
            keys[e] = key_base + (uint32_t)(o < nb0 ? b0[o] : b1[o - nb0]);
            vals[e] = (uint32_t)o;                               // occurrence index inside table b's own list
        }
    }
}


// ================================================================================================ single-occurrence rows in the forward
// Round 3.  At C5 a batch of 1 M triples names ~0.98 M users that occur ONCE and ~1.6 M of its 2 M item occurrences are the only
// occurrence of their row.  For such a row the "segment sum" is one gradient row that the forward kernel has in registers, so the
// forward kernel applies the optimizer itself: it reads the row's two moments next to the row, writes all three back, and the
// compact gradient row (GU[b] / GP[b]) is neither written nor read again, nor is the row re-read by an apply kernel:
//
//   cdr_sort_ids_two_tables   as before, but FIRST (it needs only the ids)
//   occ_flags_kernel          one pass over the sorted keys: flags[occurrence] = "only occurrence of its row" and the heads of the
//                             remaining (duplicate) segments compacted into two lists
//   batch_norms_kernel        the EmbLoss norms ||U[uid]||, ||I[pid]|| of the batch (its gradient coefficient is a global scalar the
//                             optimizer needs BEFORE the first row is updated): a gather of 2 rows per triple.  (First version: a
//                             per-row squared-norm cache kept current by every writer -- its 2.6 M scattered 4-byte stores per
//                             step, partial-line read-modify-writes at the HBM, cost 0.35 ms per domain step against 0.18 ms for
//                             this gather, measured A/B.)
//   bpr_fwd_apply_kernel      gather 3 rows (+ 2 moments per single row) -> loss -> single rows: optimizer in place;
//                             duplicate rows: GU[b] / GP[b] as before
//   rowwise_apply_dups_kernel the segmented apply over the duplicate segments only (same sums, same order as rowwise_apply_kernel)
//
// Same arithmetic per row as cdr_bpr_fwd_grad + cdr_rowwise_apply (one occurrence: 0 + g, then the same update), fixed order
// everywhere: bit-reproducible.  Bytes per triple at D = 128, uniform ids: ~9.5 KB against 12.9 KB before (SURVEY 8d floor 9.2 KB).
struct tab_ptrs { float* W; float* M; float* V; };

template <int OPT>
__device__ __forceinline__ float4 upd_math(float4 w, float4& m, float4& v, float4 acc, float rc, const apply_hp& h) {
    float4 gr = make_float4(acc.x + rc * w.x, acc.y + rc * w.y, acc.z + rc * w.z, acc.w + rc * w.w);
    if (h.wd != 0.f) { gr.x += h.wd * w.x; gr.y += h.wd * w.y; gr.z += h.wd * w.z; gr.w += h.wd * w.w; }
    if (OPT == 0) return make_float4(w.x - h.lr * gr.x, w.y - h.lr * gr.y, w.z - h.lr * gr.z, w.w - h.lr * gr.w);
    m.x += (gr.x - m.x) * (1.0f - h.b1); m.y += (gr.y - m.y) * (1.0f - h.b1);
    m.z += (gr.z - m.z) * (1.0f - h.b1); m.w += (gr.w - m.w) * (1.0f - h.b1);
    v.x = h.b2 * v.x + (1.0f - h.b2) * gr.x * gr.x; v.y = h.b2 * v.y + (1.0f - h.b2) * gr.y * gr.y;
    v.z = h.b2 * v.z + (1.0f - h.b2) * gr.z * gr.z; v.w = h.b2 * v.w + (1.0f - h.b2) * gr.w * gr.w;
    return make_float4(w.x - cdr_adam_term(m.x, v.x, h.step_size, h.bc2_sqrt, h.eps), w.y - cdr_adam_term(m.y, v.y, h.step_size, h.bc2_sqrt, h.eps),
                       w.z - cdr_adam_term(m.z, v.z, h.step_size, h.bc2_sqrt, h.eps), w.w - cdr_adam_term(m.w, v.w, h.step_size, h.bc2_sqrt, h.eps));
}

// partials[block] = {sum_b ||U[uid[b]]||^2, sum_b ||I[pid[b]]||^2}: the EmbLoss norms of the batch (emcdr.py:129-131: reg_loss(user_e, pos_e))
template <int LPR>
__global__ __launch_bounds__(kBlock) void batch_norms_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                             const int64_t* __restrict__ uid, const int64_t* __restrict__ pid, int64_t B,
                                                             double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    constexpr int UNR = 8;
    __shared__ double smem[2 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    double acc[2] = {0.0, 0.0};
    for (int64_t base = gg; base < B; base += TG * UNR) {
        int64_t iu[UNR], ip[UNR];
        float4 u[UNR], p[UNR];
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ip[r] = pid[tc];
        }
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            u[r] = p[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < B && live) { u[r] = ld4n<(LPR >= 32)>(U + iu[r] * D + 4 * sub); p[r] = ld4n<(LPR >= 32)>(I + ip[r] * D + 4 * sub); }
        }
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const float su = group_sum<LPR>(dot4(u[r], u[r])), sp = group_sum<LPR>(dot4(p[r], p[r]));
            if (sub == 0) { acc[0] += (double)su; acc[1] += (double)sp; }
        }
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1];
    }
}

// keys / perm: the two-table sort's output (section A = positions [0, nA): user keys; section B = [nA, n): item keys + key_base).
// flags: fstride bytes per positive j = {its user occurrence (A occurrence j), its positive (B occurrence j), its negatives (B
// occurrence nA + j + m nA for m = 0 .. k - 1: recbole's k-major layout), padding}; the per-triple step is the case k = 1 of it
// (fstride = 4: one 32-bit load per triple in the forward kernel).  cnt[0] / cnt[1]: lengths of headsA / headsB (sorted
// positions, relative to their section, of the first occurrence of every row that occurs more than once; list order is
// irrelevant -- every segment is summed in occurrence order by whoever takes it).
// A thread takes kFlagIT consecutive positions and a block reserves its share of each list with ONE atomic: with an atomic per
// 256 positions the 24 k same-address atomics of a 3 M-position launch were most of its 0.157 ms.
constexpr int kFlagIT = 8;
__global__ __launch_bounds__(kBlock) void occ_flags_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                           int64_t nA, int64_t n, int fstride, uint8_t* __restrict__ flags,
                                                           uint32_t* __restrict__ headsA, uint32_t* __restrict__ headsB,
                                                           unsigned* __restrict__ cnt, bool stat = false) {
    constexpr int NW = kBlock / 64;
    __shared__ unsigned wcnt[3][NW], wbase[2][NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * kFlagIT; base < n; base += (int64_t)gridDim.x * kBlock * kFlagIT) {
        const int64_t q0 = base + (int64_t)threadIdx.x * kFlagIT;
        uint32_t k[kFlagIT + 2], o[kFlagIT];
#pragma unroll
        for (int j = 0; j < kFlagIT + 2; ++j) {
            const int64_t q = q0 - 1 + j;
            k[j] = (q >= 0 && q < n) ? keys[q] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kFlagIT; ++j) o[j] = q0 + j < n ? perm[q0 + j] : 0u;
        unsigned hA = 0, hB = 0, nd = 0;                   // bit j: position q0 + j heads a duplicate segment; nd: positions that are not alone
#pragma unroll
        for (int j = 0; j < kFlagIT; ++j) {
            const int64_t q = q0 + j;
            if (q < n) {
                const uint32_t row = k[j + 1];
                const bool first = q == 0 || k[j] != row, last = q + 1 >= n || k[j + 2] != row;
                nd += (first && last) ? 0u : 1u;
                if (q < nA) {
                    flags[(int64_t)fstride * o[j]] = (uint8_t)(first && last);
                    hA |= (unsigned)(first && !last) << j;
                } else {
                    const int64_t ob = o[j];
                    int64_t fi;
                    if (ob < nA) fi = fstride * ob + 1;
                    else { const int64_t r_ = ob - nA, m_ = r_ / nA; fi = fstride * (r_ - m_ * nA) + 2 + m_; }
                    flags[fi] = (uint8_t)(first && last);
                    hB |= (unsigned)(first && !last) << j;
                }
            }
        }
        // exclusive prefix of the per-thread head counts: inside the wave by shuffles, across waves through LDS, one atomic per list
        unsigned cA = (unsigned)__popc(hA), cB = (unsigned)__popc(hB), pA = cA, pB = cB;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned a = __shfl_up(pA, d, 64), b = __shfl_up(pB, d, 64);
            if (lane >= d) { pA += a; pB += b; }
        }
        if (stat) {                                            // (medium batches only: the statistic that picks their id path)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) nd += __shfl_xor(nd, d, 64);
        }
        if (lane == 63) { wcnt[0][wave] = pA; wcnt[1][wave] = pB; wcnt[2][wave] = nd; }
        __syncthreads();
        if (stat && threadIdx.x == 2) {                        // cnt[2] += duplicate occurrences: the host binding's statistic (which id path serves this stream)
            unsigned tot = 0;
            for (int w = 0; w < NW; ++w) tot += wcnt[2][w];
            if (tot) atomicAdd(&cnt[2], tot);
        }
        if (threadIdx.x < 2) {
            const int t = threadIdx.x;
            unsigned tot = 0;
            for (int w = 0; w < NW; ++w) { wbase[t][w] = tot; tot += wcnt[t][w]; }
            const unsigned b = tot ? atomicAdd(&cnt[t], tot) : 0u;
            for (int w = 0; w < NW; ++w) wbase[t][w] += b;
        }
        __syncthreads();
        unsigned sA = wbase[0][wave] + pA - cA, sB = wbase[1][wave] + pB - cB;
#pragma unroll
        for (int j = 0; j < kFlagIT; ++j) {
            if ((hA >> j) & 1u) headsA[sA++] = (uint32_t)(q0 + j);
            if ((hB >> j) & 1u) headsB[sB++] = (uint32_t)(q0 + j - nA);
        }
        __syncthreads();
    }
}

// out9[4], out9[5] = reg_weight / (B * ||rows||) from batch_norms_kernel's partial sums (0 when there is no EmbLoss or the norm is 0)
// (k-major batches: the partials are sums over the S positives, every one of which stands for k rows of the batch -- the norm is taken
// over B = S k rows and the coefficient is handed out pre-multiplied by k, per S-list occurrence)
// step_dev (optional; the capturable step): {user table's, item table's} update counts in device memory -- advanced here, and the Adam
// scalars of the new counts left in hp_dev[0..3] = {step_size_u, bc2_sqrt_u, step_size_i, bc2_sqrt_i} for the kernels behind (apply_hp::dev)
// norms2 (optional): {sum ||u||^2, sum ||p||^2} GIVEN (all-reduced over ranks) instead of summed from `partials`
__global__ __launch_bounds__(kBlock) void coef_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t B,
                                                             float reg_weight, float* __restrict__ out9, int kmul = 1,
                                                             int64_t* __restrict__ step_u_dev = nullptr, int64_t* __restrict__ step_i_dev = nullptr,
                                                             float* __restrict__ hp_dev = nullptr, float lr = 0.f, float b1 = 0.f, float b2 = 0.f,
                                                             unsigned* __restrict__ zero4 = nullptr, const float* __restrict__ norms2 = nullptr) {
    __shared__ double smem[2 * (kBlock / 64)];
    if (zero4 && threadIdx.x >= 64 && threadIdx.x < 68) zero4[threadIdx.x - 64] = 0u;       // the head lists' counters (occ_flags_kernel): no launch of their own
    if (hp_dev && threadIdx.x < 2) {
        int64_t* c = threadIdx.x == 0 ? step_u_dev : step_i_dev;
        const int64_t st = c[0] + 1;
        c[0] = st;
        cdr_adam_hp((double)st, lr, b1, b2, hp_dev[2 * threadIdx.x], hp_dev[2 * threadIdx.x + 1]);
    }
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1];
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        if (norms2) { acc[0] = (double)norms2[0]; acc[1] = (double)norms2[1]; }
        const float nu = (float)sqrt((double)kmul * acc[0]), ni = (float)sqrt((double)kmul * acc[1]);
        out9[4] = (reg_weight != 0.f && nu > 0.f) ? (float)kmul * (reg_weight / ((float)B * nu)) : 0.f;
        out9[5] = (reg_weight != 0.f && ni > 0.f) ? (float)kmul * (reg_weight / ((float)B * ni)) : 0.f;
    }
}

// loss scalars of the fused step: as step_finish_kernel, but out9[4..5] (the EmbLoss coefficients the kernels used) stay
__global__ __launch_bounds__(kBlock) void step_finish_keep_kernel(const double* __restrict__ partials, int nblocks, int64_t B,
                                                                  float reg_weight, float* __restrict__ out9,
                                                                  unsigned* __restrict__ zero_a = nullptr, unsigned* __restrict__ zero_b = nullptr,
                                                                  const float* __restrict__ norms2 = nullptr) {
    __shared__ double smem[3 * (kBlock / 64)];
    // the long-segment counters of the two duplicate-row applies behind this launch (apply_dups_pair): no launches of their own
    if (zero_a && threadIdx.x >= 64 && threadIdx.x < 68) zero_a[threadIdx.x - 64] = 0u;
    if (zero_b && threadIdx.x >= 128 && threadIdx.x < 132) zero_b[threadIdx.x - 128] = 0u;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2];
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        if (norms2) { acc[1] = (double)norms2[0]; acc[2] = (double)norms2[1]; }
        const float main_loss = (float)(acc[0] / (double)B);
        const float nu = (float)sqrt(acc[1]), ni = (float)sqrt(acc[2]);
        out9[1] = main_loss; out9[2] = nu; out9[3] = ni;
        out9[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
        out9[6] = (float)acc[0]; out9[7] = (float)acc[1]; out9[8] = (float)acc[2];
    }
}

// {0, sum u^2, sum p^2} of batch_norms_kernel's partials as three floats (the row shard all-reduces them before the update)
__global__ __launch_bounds__(kBlock) void norm_sums_kernel(const double* __restrict__ partials, int nblocks, float* __restrict__ sums3) {
    __shared__ double smem[2 * (kBlock / 64)];
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1];
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) { sums3[0] = 0.f; sums3[1] = (float)acc[0]; sums3[2] = (float)acc[1]; }
}

// out9[6..8] = this rank's {loss sum, sum u^2, sum p^2} from the forward's partials; nothing else of out9 moves (row shard: the caller
// all-reduces the three and finishes the loss with cdr_loss_finish_sums)
__global__ __launch_bounds__(kBlock) void shard_sums_kernel(const double* __restrict__ partials, int nblocks, float* __restrict__ out9,
                                                            unsigned* __restrict__ zero_a) {
    __shared__ double smem[3 * (kBlock / 64)];
    if (zero_a && threadIdx.x >= 64 && threadIdx.x < 68) zero_a[threadIdx.x - 64] = 0u;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2];
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) { out9[6] = (float)acc[0]; out9[7] = (float)acc[1]; out9[8] = (float)acc[2]; }
}

// XD (round 5, the dimension-sharded step): the triple's score x_t = <u,p> - <u,n> is GIVEN (xdiff[t]: the all-reduced sum of every
// rank's column-slice partials, cdr_dimshard.hip) instead of being formed from the rows held here; the norm partials are not produced.
// SH (round 6, the row-sharded step): the item "table" TI.W is the buffer of RECEIVED rows (one per distinct item this rank asked for,
// read-only) and an item occurrence flagged "only occurrence of its row" is not updated -- its finished gradient row (g u + c_i p for the
// positive, -g u for the negative) is written straight into the send slot of its row, GS[ip] / GS[in]; GP[t] = g u only when one of the
// triple's two item rows is a duplicate (the segmented sum over the duplicates reads it).
template <int LPR, int OPT, int UN, bool XD = false, bool SH = false>
__global__ __launch_bounds__(kBlock) void bpr_fwd_apply_kernel(tab_ptrs TU, tab_ptrs TI, int D, const int64_t* __restrict__ uid,
                                                               const int64_t* __restrict__ pid, const int64_t* __restrict__ nid,
                                                               const uint32_t* __restrict__ flags4, int64_t B, float gamma, float invB,
                                                               const float* __restrict__ coef, apply_hp hu, apply_hp hi,
                                                               float* __restrict__ GU, float* __restrict__ GP,
                                                               double* __restrict__ partials, const float* __restrict__ xdiff = nullptr,
                                                               float* __restrict__ GS = nullptr) {
    HP_FROM_DEV(hu); HP_FROM_DEV(hi);
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const bool live = sub < D4;
    const float cu = coef[0], ci = coef[1];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    double acc[3] = {0.0, 0.0, 0.0};

    // The ids + flags of the NEXT iteration are requested right behind this iteration's row loads and are waited for together
    // with them, BEFORE this iteration's stores are issued: vmcnt retires in order, so a wait for anything requested after the
    // stores also waits for every store's acknowledgement (the first version of this loop had s_waitcnt vmcnt(0) at its head).
    uint32_t iu[UN], ip[UN], in[UN], fl[UN];
    float xd[UN];
#pragma unroll
    for (int r = 0; r < UN; ++r) {
        const int64_t t = gg + (int64_t)r * TG;
        const int64_t tc = t < B ? t : B - 1;
        iu[r] = (uint32_t)uid[tc]; ip[r] = (uint32_t)pid[tc]; in[r] = (uint32_t)nid[tc];
        fl[r] = flags4[tc];
        xd[r] = XD ? xdiff[tc] : 0.f;
    }
    // (waited for HERE: a wait that the loop header inherits from this prologue is a static s_waitcnt vmcnt(0) on every iteration)
#pragma unroll
    for (int r = 0; r < UN; ++r) asm volatile("" : "+v"(iu[r]), "+v"(ip[r]), "+v"(in[r]), "+v"(fl[r]), "+v"(xd[r]));
    for (int64_t base = gg; base < B; base += TG * UN) {
        float4 u[UN], p[UN], n[UN], um[UN], uv[UN], pm[UN], pv[UN], nm[UN], nv[UN];
        int64_t ou[UN], op[UN], on[UN];
        bool fu[UN], fp[UN], fn[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const bool ok = t < B && live;
            fu[r] = (fl[r] & 0xFFu) != 0 && t < B; fp[r] = (fl[r] & 0xFF00u) != 0 && t < B; fn[r] = (fl[r] & 0xFF0000u) != 0 && t < B;
            ou[r] = (int64_t)iu[r] * D + 4 * sub; op[r] = (int64_t)ip[r] * D + 4 * sub; on[r] = (int64_t)in[r] * D + 4 * sub;
            u[r] = ok ? ld4n<(LPR >= 32)>(TU.W + ou[r]) : z4;
            p[r] = ok ? ld4n<(LPR >= 32)>(TI.W + op[r]) : z4;
            n[r] = ok ? ld4n<(LPR >= 32)>(TI.W + on[r]) : z4;
            um[r] = uv[r] = pm[r] = pv[r] = nm[r] = nv[r] = z4;
            if (OPT == 1) {
                if (ok && fu[r]) { um[r] = ld4n<(LPR >= 32)>(TU.M + ou[r]); uv[r] = ld4n<(LPR >= 32)>(TU.V + ou[r]); }
                if (!SH && ok && fp[r]) { pm[r] = ld4n<(LPR >= 32)>(TI.M + op[r]); pv[r] = ld4n<(LPR >= 32)>(TI.V + op[r]); }
                if (!SH && ok && fn[r]) { nm[r] = ld4n<(LPR >= 32)>(TI.M + on[r]); nv[r] = ld4n<(LPR >= 32)>(TI.V + on[r]); }
            }
        }
        uint32_t ju[UN], jp[UN], jn[UN], gl[UN];
        float yd[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int64_t t = base + (int64_t)(UN + r) * TG;
            const int64_t tc = t < B ? t : B - 1;
            ju[r] = (uint32_t)uid[tc]; jp[r] = (uint32_t)pid[tc]; jn[r] = (uint32_t)nid[tc];
            gl[r] = flags4[tc];
            yd[r] = XD ? xdiff[tc] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);                  // keep the requests above the arithmetic (the scheduler sinks them otherwise)
        float gco[UN], sus[UN], sps[UN], lss[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            float x;
            if (XD) { x = xd[r]; sus[r] = sps[r] = 0.f; }
            else {
                const float dp = group_sum<LPR>(dot4(u[r], p[r]));
                const float dn = group_sum<LPR>(dot4(u[r], n[r]));
                sus[r] = group_sum<LPR>(dot4(u[r], u[r]));
                sps[r] = group_sum<LPR>(dot4(p[r], p[r]));
                x = dp - dn;
            }
            const float s = sigmoidf_(x);
            gco[r] = -invB * (s * (1.0f - s)) / (gamma + s);
            lss[r] = -logf(gamma + s);
        }
        // every request of this iteration -- rows AND the next ids -- has returned here; nothing below waits on vmcnt again
#pragma unroll
        for (int r = 0; r < UN; ++r) asm volatile("" : "+v"(ju[r]), "+v"(jp[r]), "+v"(jn[r]), "+v"(gl[r]), "+v"(yd[r]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const float g = gco[r];
            const float4 gu = make_float4(g * (p[r].x - n[r].x), g * (p[r].y - n[r].y), g * (p[r].z - n[r].z), g * (p[r].w - n[r].w));
            const float4 gi = make_float4(g * u[r].x, g * u[r].y, g * u[r].z, g * u[r].w);
            const bool ok = t < B && live;
            // ---- user row
            if (fu[r]) {
                const float4 wu = upd_math<OPT>(u[r], um[r], uv[r], gu, cu, hu);
                if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TU.M + ou[r], um[r]); st4n<(LPR >= 32)>(TU.V + ou[r], uv[r]); } st4n<(LPR >= 32)>(TU.W + ou[r], wu); }
            } else if (ok) st4n<(LPR >= 32)>(GU + t * D + 4 * sub, gu);
            // ---- positive item row (EmbLoss occurrence), negative item row (gradient -g u, no EmbLoss)
            if (SH) {
                if (fp[r] && live) st4n<(LPR >= 32)>(GS + op[r], make_float4(__builtin_fmaf(ci, p[r].x, gi.x), __builtin_fmaf(ci, p[r].y, gi.y),
                                                               __builtin_fmaf(ci, p[r].z, gi.z), __builtin_fmaf(ci, p[r].w, gi.w)));
                if (fn[r] && live) st4n<(LPR >= 32)>(GS + on[r], make_float4(0.f - gi.x, 0.f - gi.y, 0.f - gi.z, 0.f - gi.w));
            } else {
            if (fp[r]) {
                const float4 wp = upd_math<OPT>(p[r], pm[r], pv[r], gi, ci, hi);
                if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TI.M + op[r], pm[r]); st4n<(LPR >= 32)>(TI.V + op[r], pv[r]); } st4n<(LPR >= 32)>(TI.W + op[r], wp); }
            }
            if (fn[r]) {
                const float4 wn = upd_math<OPT>(n[r], nm[r], nv[r], make_float4(0.f - gi.x, 0.f - gi.y, 0.f - gi.z, 0.f - gi.w), 0.f, hi);
                if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TI.M + on[r], nm[r]); st4n<(LPR >= 32)>(TI.V + on[r], nv[r]); } st4n<(LPR >= 32)>(TI.W + on[r], wn); }
            }
            }
            if (ok && !(fp[r] && fn[r])) st4n<(LPR >= 32)>(GP + t * D + 4 * sub, gi);
            if (t < B && sub == 0) {
                acc[0] += (double)lss[r];
                acc[1] += (double)sus[r];
                acc[2] += (double)sps[r];
            }
        }
#pragma unroll
        for (int r = 0; r < UN; ++r) { iu[r] = ju[r]; ip[r] = jp[r]; in[r] = jn[r]; fl[r] = gl[r]; xd[r] = yd[r]; }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// The pointwise rows (user, item, label) -- EMCDR's default MF latent factor model (emcdr.py:111-122: MSE on the raw dot) and CMF's BCE
// (cmf.py:75-99) -- through the same forward-and-update pass (round 5): flags4[t] byte 0 / byte 1 = the row's user / item occurs once in the
// batch; such a row is updated in place from registers (gradient g v resp. g u plus its EmbLoss term), a duplicate row's gradient row goes
// to GU[t] / GI[t] for the segmented apply.  Same prefetch discipline as bpr_fwd_apply_kernel (ids, labels and flags of the next iteration
// requested behind this iteration's row loads and waited for before its stores).
// XD: the row's dot <u, i> is GIVEN (xdot[t]: the all-reduced sum of every rank's column-slice partials, the dimension-sharded step).
template <int LPR, int OPT, bool XD = false>
__global__ __launch_bounds__(kBlock) void point_fwd_apply_kernel(int loss_kind, tab_ptrs TU, tab_ptrs TI, int D, const int64_t* __restrict__ uid,
                                                                 const int64_t* __restrict__ iid, const float* __restrict__ label,
                                                                 const uint32_t* __restrict__ flags4, int64_t B, float invB,
                                                                 const float* __restrict__ coef, apply_hp hu, apply_hp hi,
                                                                 float* __restrict__ GU, float* __restrict__ GI, double* __restrict__ partials,
                                                                 const float* __restrict__ xdot = nullptr) {
    HP_FROM_DEV(hu); HP_FROM_DEV(hi);
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    const float cu = coef[0], ci = coef[1];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    double acc[3] = {0.0, 0.0, 0.0};
    uint32_t iu, ii, fl; float yl, xd = 0.f;
    {
        const int64_t tc = gg < B ? gg : B - 1;
        iu = (uint32_t)uid[tc]; ii = (uint32_t)iid[tc]; fl = flags4[tc]; yl = label[tc];
        if (XD) xd = xdot[tc];
    }
    asm volatile("" : "+v"(iu), "+v"(ii), "+v"(fl), "+v"(yl), "+v"(xd));
    for (int64_t t = gg; t < B; t += TG) {
        const bool fu = (fl & 0xFFu) != 0, fi = (fl & 0xFF00u) != 0;
        const int64_t ou = (int64_t)iu * D + 4 * sub, oi = (int64_t)ii * D + 4 * sub;
        float4 u = live ? ld4n<(LPR >= 32)>(TU.W + ou) : z4, v = live ? ld4n<(LPR >= 32)>(TI.W + oi) : z4;
        float4 um = z4, uv = z4, im = z4, iv = z4;
        if (OPT == 1) {
            if (live && fu) { um = ld4n<(LPR >= 32)>(TU.M + ou); uv = ld4n<(LPR >= 32)>(TU.V + ou); }
            if (live && fi) { im = ld4n<(LPR >= 32)>(TI.M + oi); iv = ld4n<(LPR >= 32)>(TI.V + oi); }
        }
        uint32_t ju, ji, gl; float yn, xn = 0.f;
        {
            const int64_t tn = t + TG, tc = tn < B ? tn : B - 1;
            ju = (uint32_t)uid[tc]; ji = (uint32_t)iid[tc]; gl = flags4[tc]; yn = label[tc];
            if (XD) xn = xdot[tc];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float dx = XD ? xd : group_sum<LPR>(dot4(u, v));
        const float su = XD ? 0.f : group_sum<LPR>(dot4(u, u));
        const float si = XD ? 0.f : group_sum<LPR>(dot4(v, v));
        float l, g;
        if (loss_kind == CDR_LOSS_MSE) {
            const float d = dx - yl;
            l = d * d; g = 2.0f * d * invB;
        } else {                                               // torch BCELoss on sigmoid(dot): -100 log clamp, 1e-12 backward clamp
            const float p = sigmoidf_(dx);
            l = (yl - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - yl * fmaxf(logf(p), -100.0f);
            const float pq = (1.0f - p) * p;
            g = (p - yl) / fmaxf(pq, 1e-12f) * invB * pq;
        }
        asm volatile("" : "+v"(ju), "+v"(ji), "+v"(gl), "+v"(yn), "+v"(xn));        // every request of this iteration has returned: stores below wait on nothing older
        __builtin_amdgcn_sched_barrier(0);
        const float4 gu = make_float4(g * v.x, g * v.y, g * v.z, g * v.w);
        const float4 gi = make_float4(g * u.x, g * u.y, g * u.z, g * u.w);
        if (fu) {
            const float4 wu = upd_math<OPT>(u, um, uv, gu, cu, hu);
            if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TU.M + ou, um); st4n<(LPR >= 32)>(TU.V + ou, uv); } st4n<(LPR >= 32)>(TU.W + ou, wu); }
        } else if (live) st4n<(LPR >= 32)>(GU + t * D + 4 * sub, gu);
        if (fi) {
            const float4 wi = upd_math<OPT>(v, im, iv, gi, ci, hi);
            if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TI.M + oi, im); st4n<(LPR >= 32)>(TI.V + oi, iv); } st4n<(LPR >= 32)>(TI.W + oi, wi); }
        } else if (live) st4n<(LPR >= 32)>(GI + t * D + 4 * sub, gi);
        if (sub == 0) { acc[0] += (double)l; acc[1] += (double)su; acc[2] += (double)si; }
        iu = ju; ii = ji; fl = gl; yl = yn; xd = xn;
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// The same idea on recbole's pairwise batch layout (S positives tiled k times, negatives k-major: crossdomain_sampler.py:148-152;
// csrc/cdr_kstep.hip): one lane group per POSITIVE gathers u and p once and its k negatives; every row among them that occurs once
// in the step's lists (users [S]; items [pid | nid]) is updated in place from registers -- user: sum_m g_m (p - n_m) + k c_u u,
// positive: (sum_m g_m) u + k c_i p, negative m: -g_m u -- and a duplicate row's gradient row goes to GU[j] / GI[occurrence] for
// the segmented apply (no {user row, coefficient} records here: the user row an item gradient is made of is updated by this very
// kernel).  flags: fstride bytes per positive {user, positive, negative 0 .. k - 1} from occ_flags_kernel.
template <int LPR, int OPT, int KC>
__global__ __launch_bounds__(kBlock) void bpr_fwd_apply_kmajor_kernel(tab_ptrs TU, tab_ptrs TI, int D, const int64_t* __restrict__ uid,
                                                                      const int64_t* __restrict__ pid, const int64_t* __restrict__ nid,
                                                                      const uint8_t* __restrict__ flags, int fstride, int64_t S, int k,
                                                                      float gamma, float invB, const float* __restrict__ coef,
                                                                      apply_hp hu, apply_hp hi, float* __restrict__ GU,
                                                                      float* __restrict__ GI, double* __restrict__ partials) {
    HP_FROM_DEV(hu); HP_FROM_DEV(hi);
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const bool live = sub < D4;
    const float cu = coef[0], ci = coef[1];                 // pre-multiplied by k (coef_finish_kernel)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t j = gg; j < S; j += TG) {
        const uint8_t* fl = flags + j * fstride;
        const int64_t iu = uid[j], ip = pid[j];
        const bool fu = fl[0] != 0, fp = fl[1] != 0;
        int64_t in0[KC]; bool fn0[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) { const int m = c < k ? c : k - 1; in0[c] = nid[j + (int64_t)m * S]; fn0[c] = fl[2 + m] != 0; }
        const int64_t ou = iu * D + 4 * sub, op = ip * D + 4 * sub;
        float4 u = z4, p = z4, um = z4, uv = z4, pm = z4, pv = z4;
        if (live) {
            u = ld4n<(LPR >= 32)>(TU.W + ou); p = ld4n<(LPR >= 32)>(TI.W + op);
            if (OPT == 1) {
                if (fu) { um = ld4n<(LPR >= 32)>(TU.M + ou); uv = ld4n<(LPR >= 32)>(TU.V + ou); }
                if (fp) { pm = ld4n<(LPR >= 32)>(TI.M + op); pv = ld4n<(LPR >= 32)>(TI.V + op); }
            }
        }
        float4 gu = z4;
        float gs = 0.f, dp = 0.f;
        for (int m0 = 0; m0 < k; m0 += KC) {
            int64_t in[KC], on[KC]; bool fn[KC];
            float4 n[KC], nm[KC], nv[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (m0 == 0) { in[c] = in0[c]; fn[c] = fn0[c]; }
                else { const int m = m0 + c < k ? m0 + c : k - 1; in[c] = nid[j + (int64_t)m * S]; fn[c] = fl[2 + m] != 0; }
                fn[c] = fn[c] && m0 + c < k;
                on[c] = in[c] * D + 4 * sub;
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                n[c] = live ? ld4n<(LPR >= 32)>(TI.W + on[c]) : z4;
                nm[c] = nv[c] = z4;
                if (OPT == 1 && live && fn[c]) { nm[c] = ld4n<(LPR >= 32)>(TI.M + on[c]); nv[c] = ld4n<(LPR >= 32)>(TI.V + on[c]); }
            }
            if (m0 == 0) {
                dp = group_sum<LPR>(dot4(u, p));
                const float su = group_sum<LPR>(dot4(u, u)), sp = group_sum<LPR>(dot4(p, p));
                if (sub == 0) { acc[1] += (double)k * (double)su; acc[2] += (double)k * (double)sp; }      // EmbLoss sees k copies
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const float dn = group_sum<LPR>(dot4(u, n[c]));
                if (m0 + c < k) {
                    const float sg = sigmoidf_(dp - dn);
                    const float g = -invB * (sg * (1.0f - sg)) / (gamma + sg);
                    gu.x += g * (p.x - n[c].x); gu.y += g * (p.y - n[c].y); gu.z += g * (p.z - n[c].z); gu.w += g * (p.w - n[c].w);
                    gs += g;
                    if (sub == 0) acc[0] += (double)(-logf(gamma + sg));
                    const float4 gn = make_float4(0.f - g * u.x, 0.f - g * u.y, 0.f - g * u.z, 0.f - g * u.w);
                    if (fn[c]) {
                        const float4 wn = upd_math<OPT>(n[c], nm[c], nv[c], gn, 0.f, hi);
                        if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TI.M + on[c], nm[c]); st4n<(LPR >= 32)>(TI.V + on[c], nv[c]); } st4n<(LPR >= 32)>(TI.W + on[c], wn); }
                    } else if (live) st4n<(LPR >= 32)>(GI + (S + j + (int64_t)(m0 + c) * S) * D + 4 * sub, gn);
                }
            }
        }
        if (fu) {
            const float4 wu = upd_math<OPT>(u, um, uv, gu, cu, hu);
            if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TU.M + ou, um); st4n<(LPR >= 32)>(TU.V + ou, uv); } st4n<(LPR >= 32)>(TU.W + ou, wu); }
        } else if (live) st4n<(LPR >= 32)>(GU + j * D + 4 * sub, gu);
        const float4 gp = make_float4(gs * u.x, gs * u.y, gs * u.z, gs * u.w);
        if (fp) {
            const float4 wp = upd_math<OPT>(p, pm, pv, gp, ci, hi);
            if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TI.M + op, pm); st4n<(LPR >= 32)>(TI.V + op, pv); } st4n<(LPR >= 32)>(TI.W + op, wp); }
        } else if (live) st4n<(LPR >= 32)>(GI + j * D + 4 * sub, gp);
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// ---- pointwise rows per POSITIVE (round 5) ---------------------------------------------------------------------------------------------
// recbole's pointwise batch (TrainDataLoader._neg_sampling, driven by data/dataloader.py:114-162): S positives, the user column tiled 1 + k
// times, items = [positives | k-major negatives], labels = [1] * S + [0] * S k.  In that layout EVERY user row occurs 1 + k times, so the
// per-row kernel above never finds a user that occurs once; here one lane group takes a positive with its k negatives: the user row is
// gathered once, its gradient sum_r g_r i_r accumulated in registers, and a user that occurs in ONE positive is updated in place (EmbLoss
// count 1 + k); every item row among the 1 + k that occurs once in [pid | nid] is updated in place (g_r u, EmbLoss count 1).  Lists, flags
// and gradient buffers exactly as bpr_fwd_apply_kmajor_kernel's: users [S] -> GU [S, D]; items [pid | nid] -> GI [S + S k, D].
template <int LPR>
__global__ __launch_bounds__(kBlock) void point_norms_kmajor_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                    const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int64_t S,
                                                                    int k, double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[2 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const bool live = sub < (D >> 2);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    double acc[2] = {0.0, 0.0};
    for (int64_t j = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; j < S; j += (int64_t)gridDim.x * GPB) {
        const float4 u = live ? ld4n<(LPR >= 32)>(U + uid[j] * D + 4 * sub) : z4;
        float si = 0.f;
        for (int r0 = 0; r0 <= k; r0 += 4) {                       // four item rows in flight
            float4 v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (live && r0 + c <= k) ? ld4n<(LPR >= 32)>(I + iid[j + (int64_t)(r0 + c) * S] * D + 4 * sub) : z4;
#pragma unroll
            for (int c = 0; c < 4; ++c) si += dot4(v[c], v[c]);
        }
        const float su = group_sum<LPR>(dot4(u, u)), st = group_sum<LPR>(si);
        if (sub == 0) { acc[0] += (double)(1 + k) * (double)su; acc[1] += (double)st; }
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1];
    }
}

// out9[4], out9[5] = reg_weight / (B ||rows||) as coef_finish_kernel; out9[9] = (1 + k) out9[4]: the user coefficient per LIST occurrence
__global__ __launch_bounds__(kBlock) void point_coef_kmajor_kernel(const double* __restrict__ partials, int nblocks, int64_t B, int k,
                                                                   float reg_weight, float* __restrict__ out9, unsigned* __restrict__ zero4) {
    __shared__ double smem[2 * (kBlock / 64)];
    if (zero4 && threadIdx.x >= 64 && threadIdx.x < 68) zero4[threadIdx.x - 64] = 0u;
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1];
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        const float nu = (float)sqrt(acc[0]), ni = (float)sqrt(acc[1]);
        out9[4] = (reg_weight != 0.f && nu > 0.f) ? reg_weight / ((float)B * nu) : 0.f;
        out9[5] = (reg_weight != 0.f && ni > 0.f) ? reg_weight / ((float)B * ni) : 0.f;
        out9[9] = (float)(1 + k) * out9[4];
    }
}

template <int LPR, int OPT, int KC>
__global__ __launch_bounds__(kBlock) void point_fwd_apply_kmajor_kernel(int loss_kind, tab_ptrs TU, tab_ptrs TI, int D, const int64_t* __restrict__ uid,
                                                                        const int64_t* __restrict__ iid, const float* __restrict__ label,
                                                                        const uint8_t* __restrict__ flags, int fstride, int64_t S, int k,
                                                                        float invB, const float* __restrict__ coef, apply_hp hu, apply_hp hi,
                                                                        float* __restrict__ GU, float* __restrict__ GI,
                                                                        double* __restrict__ partials) {
    HP_FROM_DEV(hu); HP_FROM_DEV(hi);
    constexpr int GPB = kBlock / LPR;
    __shared__ double smem[3 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const bool live = sub < (D >> 2);
    const float cu = coef[5], ci = coef[1];                 // coef = out9 + 4: {c_u, c_i, ..., (1 + k) c_u at out9[9]}
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t j = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; j < S; j += (int64_t)gridDim.x * GPB) {
        const uint8_t* fl = flags + j * fstride;
        const int64_t iu = uid[j];
        const bool fu = fl[0] != 0;
        const int64_t ou = iu * D + 4 * sub;
        float4 u = z4, um = z4, uv = z4;
        if (live) {
            u = ld4n<(LPR >= 32)>(TU.W + ou);
            if (OPT == 1 && fu) { um = ld4n<(LPR >= 32)>(TU.M + ou); uv = ld4n<(LPR >= 32)>(TU.V + ou); }
        }
        float4 gu = z4;
        float su = 0.f;
        for (int r0 = 0; r0 <= k; r0 += KC) {                   // item rows r = 0 (the positive), 1 .. k (negatives), KC at a time
            int64_t oi[KC]; bool fi[KC]; float yl[KC];
            float4 v[KC], vm[KC], vv[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int r = r0 + c <= k ? r0 + c : k;
                oi[c] = iid[j + (int64_t)r * S] * D + 4 * sub;
                fi[c] = fl[1 + r] != 0 && r0 + c <= k;
                yl[c] = label[j + (int64_t)r * S];
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                v[c] = live ? ld4n<(LPR >= 32)>(TI.W + oi[c]) : z4;
                vm[c] = vv[c] = z4;
                if (OPT == 1 && live && fi[c]) { vm[c] = ld4n<(LPR >= 32)>(TI.M + oi[c]); vv[c] = ld4n<(LPR >= 32)>(TI.V + oi[c]); }
            }
            if (r0 == 0) su = group_sum<LPR>(dot4(u, u));
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const float dx = group_sum<LPR>(dot4(u, v[c]));
                const float si = group_sum<LPR>(dot4(v[c], v[c]));
                if (r0 + c <= k) {
                    const float y = yl[c];
                    float l, g;
                    if (loss_kind == CDR_LOSS_MSE) {
                        const float d = dx - y;
                        l = d * d; g = 2.0f * d * invB;
                    } else {
                        const float p = sigmoidf_(dx);
                        l = (y - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - y * fmaxf(logf(p), -100.0f);
                        const float pq = (1.0f - p) * p;
                        g = (p - y) / fmaxf(pq, 1e-12f) * invB * pq;
                    }
                    gu.x += g * v[c].x; gu.y += g * v[c].y; gu.z += g * v[c].z; gu.w += g * v[c].w;
                    if (sub == 0) { acc[0] += (double)l; acc[2] += (double)si; }
                    const float4 gi = make_float4(g * u.x, g * u.y, g * u.z, g * u.w);
                    if (fi[c]) {
                        const float4 wi = upd_math<OPT>(v[c], vm[c], vv[c], gi, ci, hi);
                        if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TI.M + oi[c], vm[c]); st4n<(LPR >= 32)>(TI.V + oi[c], vv[c]); } st4n<(LPR >= 32)>(TI.W + oi[c], wi); }
                    } else if (live) st4n<(LPR >= 32)>(GI + (j + (int64_t)(r0 + c) * S) * D + 4 * sub, gi);
                }
            }
        }
        if (sub == 0) acc[1] += (double)(1 + k) * (double)su;
        if (fu) {
            const float4 wu = upd_math<OPT>(u, um, uv, gu, cu, hu);
            if (live) { if (OPT == 1) { st4n<(LPR >= 32)>(TU.M + ou, um); st4n<(LPR >= 32)>(TU.V + ou, uv); } st4n<(LPR >= 32)>(TU.W + ou, wu); }
        } else if (live) st4n<(LPR >= 32)>(GU + j * D + 4 * sub, gu);
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// End of a MEDIUM segment (3 .. kLongSeg occurrences) headed at sorted position q: the first position past q + 2 whose key differs.
// One probe per lane and a ballot instead of a walk (round 6): the walk was a chain of up to 30 dependent key loads per segment -- most of
// the 60 us the duplicate apply took on a Zipf(1.05) batch of 65,536 triples, whose items of rank 100 .. 3,000 have such segments.
template <int LPR>
__device__ __forceinline__ int64_t seg_end_probe(const uint32_t* __restrict__ keys, int64_t q, int64_t n, uint32_t row) {
    if constexpr (LPR >= 32) {
        const int lane = threadIdx.x & 63, l = lane & 31;
        const int64_t p = q + 3 + l;
        const bool same = (LPR == 32 || lane < 32) && p < n && p <= q + kLongSeg && keys[p] == row;
        const unsigned long long m = __ballot(!same);
        const uint32_t mg = LPR == 32 ? (uint32_t)(m >> (lane & 32)) : (uint32_t)m;
        return q + 3 + (mg ? (int64_t)(__ffs((int)mg) - 1) : 32);
    } else {
        int64_t end = q + 3;
        while (end < n && end <= q + kLongSeg && keys[end] == row) ++end;
        return end;
    }
}

// The occurrence numbers of a medium segment's third and later entries, one per lane (lane l of the group: sorted position q + 2 + l),
// fetched together behind the end probe: the rounds below hand them round with lane shuffles instead of loading perm[] ahead of every
// round's row loads (one dependent hop per round less).
template <int LPR>
__device__ __forceinline__ uint32_t seg_occ_fetch(const uint32_t* __restrict__ perm, int64_t q, int64_t end) {
    if constexpr (LPR >= 32) {
        const int l = threadIdx.x & 31;
        return ((LPR == 32 || (threadIdx.x & 63) < 32) && q + 2 + l < end) ? perm[q + 2 + l] : 0u;
    } else {
        return 0u;
    }
}
template <int LPR>
__device__ __forceinline__ int64_t seg_occ_at(const uint32_t* __restrict__ perm, uint32_t mine, int64_t q, int64_t e) {
    if constexpr (LPR >= 32) return (int64_t)__shfl(mine, (int)(e - (q + 2)), LPR);      // (LPR = 64: the values sit in the wave's lower half)
    else return (int64_t)perm[e];
}

// The segmented apply over the DUPLICATE segments only: heads[0 .. *nheads) are the sorted positions of their first occurrences
// (any order: every segment is summed by one lane group in occurrence order, whoever takes it).  Long segments as in
// rowwise_apply_kernel.
template <int LPR, int OPT, bool SIGNED>
__device__ __forceinline__ void rowwise_apply_dups_body(float* __restrict__ W, float* __restrict__ Mo, float* __restrict__ Vo,
                                                        int D, const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ perm, int64_t n,
                                                        const uint32_t* __restrict__ heads, const unsigned* __restrict__ nheads,
                                                        const float* __restrict__ G, int64_t neg_start, int64_t reg_limit,
                                                        const float* __restrict__ reg_coef, apply_hp hp,
                                                        unsigned* __restrict__ counters, seg_long* __restrict__ longs,
                                                        seg_piece* __restrict__ pieces) {
    HP_FROM_DEV(hp);
    constexpr int GPB = kBlock / LPR;
    constexpr int SU = 4;                                 // segments in flight per lane group
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    const int64_t nh = (int64_t)nheads[0];
    if (D4 <= LPR) {
        // Whole rows per lane (D <= 256).  A segment is a chain of dependent random accesses (head -> keys -> perm -> gradient rows,
        // next to w, m, v): one segment at a time per lane group left the kernel latency-bound (0.197 ms for 0.8 GB at C5).  SU
        // segments go through the chain together: SU heads, then SU x 3 keys, then their rows, moments and the first two
        // occurrences' gradient rows (most duplicate rows have two or three occurrences) are all requested before anything is used.
        const bool live = sub < D4;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t h0 = gg * SU; h0 < nh; h0 += TG * SU) {
            int64_t q[SU]; uint32_t row[SU], k1[SU], k2[SU]; bool ok[SU];
#pragma unroll
            for (int j = 0; j < SU; ++j) { ok[j] = h0 + j < nh; q[j] = ok[j] ? (int64_t)heads[h0 + j] : 0; }
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                row[j] = keys[q[j]];
                k1[j] = q[j] + 1 < n ? keys[q[j] + 1] : ~0u;               // (a head of a duplicate segment: keys[q + 1] == row)
                k2[j] = q[j] + 2 < n ? keys[q[j] + 2] : ~0u;
            }
            uint32_t far[SU], o0[SU], o1[SU];
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                far[j] = q[j] + kLongSeg < n ? keys[q[j] + kLongSeg] : ~0u;
                o0[j] = perm[q[j]]; o1[j] = q[j] + 1 < n ? perm[q[j] + 1] : 0u;
            }
            float4 w[SU], m[SU], v[SU], g0[SU], g1[SU];
            int64_t off[SU];
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                ok[j] = ok[j] && !(far[j] == row[j] && row[j] != ~0u && q[j] + kLongSeg < n);      // long segments: registered below
                off[j] = (int64_t)row[j] * D + 4 * sub;
                w[j] = m[j] = v[j] = g0[j] = g1[j] = z4;
                if (ok[j] && live) {
                    w[j] = ld4n<(LPR >= 32)>(W + off[j]);
                    if (OPT == 1) { m[j] = ld4n<(LPR >= 32)>(Mo + off[j]); v[j] = ld4n<(LPR >= 32)>(Vo + off[j]); }
                    const bool n0 = SIGNED && (int64_t)o0[j] >= neg_start, n1 = SIGNED && (int64_t)o1[j] >= neg_start;
                    g0[j] = ld4n<(LPR >= 32)>(G + (n0 ? (int64_t)o0[j] - neg_start : (int64_t)o0[j]) * D + 4 * sub);
                    g1[j] = ld4n<(LPR >= 32)>(G + (n1 ? (int64_t)o1[j] - neg_start : (int64_t)o1[j]) * D + 4 * sub);
                }
            }
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                if (!ok[j]) continue;
                float4 acc = z4;
                int cnt = 0;
                {   // occurrences 0 and 1 (always present), in occurrence order, exactly as the one-at-a-time loop adds them
                    const bool n0 = SIGNED && (int64_t)o0[j] >= neg_start, n1 = SIGNED && (int64_t)o1[j] >= neg_start;
                    if (n0) { acc.x -= g0[j].x; acc.y -= g0[j].y; acc.z -= g0[j].z; acc.w -= g0[j].w; }
                    else { acc.x += g0[j].x; acc.y += g0[j].y; acc.z += g0[j].z; acc.w += g0[j].w; }
                    cnt += ((int64_t)o0[j] < reg_limit) ? 1 : 0;
                    if (n1) { acc.x -= g1[j].x; acc.y -= g1[j].y; acc.z -= g1[j].z; acc.w -= g1[j].w; }
                    else { acc.x += g1[j].x; acc.y += g1[j].y; acc.z += g1[j].z; acc.w += g1[j].w; }
                    cnt += ((int64_t)o1[j] < reg_limit) ? 1 : 0;
                }
                if (k2[j] == row[j]) {                                     // third and later occurrences (at most kLongSeg - 2 of them)
                    // the segment's end first (its keys are neighbours in memory), then the gradient rows eight at a time: one at a time
                    // this walk was a chain of up to 30 dependent ~1.5 us loads per segment, which is what a Zipf batch's step waited for
                    const int64_t end = seg_end_probe<LPR>(keys, q[j], n, row[j]);
                    const uint32_t occ_l = seg_occ_fetch<LPR>(perm, q[j], end);
                    for (int64_t e0 = q[j] + 2; e0 < end; e0 += 8) {
                        int64_t o[8]; float4 g[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const int64_t ov = seg_occ_at<LPR>(perm, occ_l, q[j], e0 + u < end ? e0 + u : q[j] + 2); o[u] = e0 + u < end ? ov : -1; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool neg = SIGNED && o[u] >= neg_start;
                            g[u] = (o[u] >= 0 && live) ? ld4n<(LPR >= 32)>(G + (neg ? o[u] - neg_start : o[u]) * D + 4 * sub) : z4;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (o[u] < 0) continue;
                            if (SIGNED && o[u] >= neg_start) { acc.x -= g[u].x; acc.y -= g[u].y; acc.z -= g[u].z; acc.w -= g[u].w; }
                            else { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
                            cnt += (o[u] < reg_limit) ? 1 : 0;
                        }
                    }
                }
                const float4 wn = upd_math<OPT>(w[j], m[j], v[j], acc, c * (float)cnt, hp);
                if (live) {
                    if (OPT == 1) { st4n<(LPR >= 32)>(Mo + off[j], m[j]); st4n<(LPR >= 32)>(Vo + off[j], v[j]); }
                    st4n<(LPR >= 32)>(W + off[j], wn);
                }
            }
        }
    } else {
        for (int64_t h = gg; h < nh; h += TG) {
            const int64_t q = heads[h];
            const uint32_t row = keys[q];
            const bool is_long = q + kLongSeg < n && keys[q + kLongSeg] == row;
            if (is_long) continue;
            for (int ch = sub; ch < D4; ch += LPR) {
                const int64_t off = (int64_t)row * D + 4 * ch;
                const float4 w = ld4n<(LPR >= 32)>(W + off);
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f), v = m;
                if (OPT == 1) { m = ld4n<(LPR >= 32)>(Mo + off); v = ld4n<(LPR >= 32)>(Vo + off); }
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                int cnt = 0;
                for (int64_t e = q; e < n && keys[e] == row; ++e) {
                    const int64_t o = perm[e];
                    const bool neg = SIGNED && o >= neg_start;
                    const float4 g = ld4n<(LPR >= 32)>(G + (neg ? o - neg_start : o) * D + 4 * ch);
                    if (neg) { acc.x -= g.x; acc.y -= g.y; acc.z -= g.z; acc.w -= g.w; }
                    else { acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w; }
                    cnt += (o < reg_limit) ? 1 : 0;
                }
                const float4 wn = upd_math<OPT>(w, m, v, acc, c * (float)cnt, hp);
                if (OPT == 1) { st4n<(LPR >= 32)>(Mo + off, m); st4n<(LPR >= 32)>(Vo + off, v); }
                st4n<(LPR >= 32)>(W + off, wn);
            }
        }
    }
    if (counters == nullptr) return;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q + kLongSeg < n; q += (int64_t)gridDim.x * kBlock) {
        const uint32_t row = keys[q];
        const uint32_t before = keys[q > 0 ? q - 1 : 0];
        const uint32_t far = keys[q + kLongSeg];
        if ((q > 0 && before == row) || far != row || row == ~0u) continue;       // (~0: the count path's filler behind the duplicate entries)
        int64_t lo = q + kLongSeg, hi = n;
        while (lo + 1 < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] == row) lo = mid; else hi = mid;
        }
        const int64_t len = hi - q;
        const unsigned np = (unsigned)((len + kPiece - 1) / kPiece);
        const unsigned base = atomicAdd(&counters[0], np);
        const unsigned li = atomicAdd(&counters[1], 1u);
        longs[li] = seg_long{q, len, (int64_t)base};
        for (unsigned k = 0; k < np; ++k) {
            const int64_t st = q + (int64_t)k * kPiece;
            pieces[base + k] = seg_piece{st, (hi - st) < kPiece ? (hi - st) : (int64_t)kPiece};
        }
    }
}
// The two tables of a fused step in ONE launch each (round 4): blockIdx.y = 0 the user table's duplicate segments, 1 the item table's.
// The device functions are the ones above with SIGNED = true for both sides -- the user side passes neg_start = n, so no occurrence is
// ever negated: the same sums in the same order as the two launches of round 3.  (Not the same BITS as the round-3 build on rows whose
// moments are non-zero: hipcc fuses the Adam update's multiply-adds differently in this instantiation -- 1 ulp on such rows, found by
// running both builds on one batch, tools/ab_step_builds.py; reruns of one build are bit-equal as before.)  At 65,536 triples the step's six
// small apply launches and three counter clears were ~45 us of its 230; the two sides also share the chip instead of queueing.
struct dup_side {
    float* W; float* M; float* V; const uint32_t* keys; const uint32_t* perm; int64_t n; const uint32_t* heads; const unsigned* nheads;
    const float* G; int64_t neg_start, reg_limit; const float* reg_coef; apply_hp hp;
    unsigned* counters; seg_long* longs; seg_piece* pieces; int* pcnt; float* partial;
    float* out = nullptr; const uint32_t* uidx = nullptr;     // row shard, item side: W = the received rows (read-only), out[uidx[position]] = the segment's sum
};
template <int LPR, int OPT>
__global__ __launch_bounds__(kBlock) void rowwise_apply_dups2_kernel(int D, dup_side a, dup_side b) {
    const dup_side& t = blockIdx.y ? b : a;
    rowwise_apply_dups_body<LPR, OPT, true>(t.W, t.M, t.V, D, t.keys, t.perm, t.n, t.heads, t.nheads, t.G, t.neg_start, t.reg_limit, t.reg_coef, t.hp,
                                            t.counters, t.longs, t.pieces);
}
template <int LPR>
__global__ __launch_bounds__(kBlock) void seg_piece_sum2_kernel(int D, dup_side a, dup_side b) {
    const dup_side& t = blockIdx.y ? b : a;
    if (t.counters == nullptr) return;
    seg_piece_sum_body<LPR, true>(D, t.perm, t.G, t.neg_start, t.reg_limit, nullptr, t.counters, t.pieces, t.partial, t.pcnt);
}
template <int LPR, int OPT>
__global__ __launch_bounds__(kBlock) void seg_long_finish2_kernel(int D, dup_side a, dup_side b) {
    const dup_side& t = blockIdx.y ? b : a;
    if (t.counters == nullptr) return;
    seg_long_finish_body<LPR, OPT>(t.W, t.M, t.V, D, t.keys, t.reg_coef, t.hp, t.counters, t.longs, t.partial, t.pcnt);
}

// ================================================================================================ round 6: the row-sharded step's own passes
// Keys of one requester-side sort for BOTH lists of a rank's routed triples (recv3 [Bl, 3] = {local user row, positive item, negative
// item}, item ids global): section A [0, Bl) = the user rows; section B [Bl, 3 Bl) = key_base + ((owner << lb) | local row) of [p | n],
// so that the distinct item rows come out grouped by owner, ascending inside an owner -- the order in which they are requested.
// Also unpacks the user column and clears the two head-list counters (cnt) of occ_flags_kernel.
__global__ __launch_bounds__(kBlock) void shard_keys_kernel(const int64_t* __restrict__ recv3, int64_t Bl, uint32_t G, unsigned lb,
                                                            uint32_t key_base, int64_t* __restrict__ u_loc, uint32_t* __restrict__ keys,
                                                            uint32_t* __restrict__ vals, unsigned* __restrict__ cnt) {
    if (blockIdx.x == 0 && threadIdx.x < 4) cnt[threadIdx.x] = 0u;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < Bl; t += (int64_t)gridDim.x * kBlock) {
        const int64_t u = recv3[3 * t];
        const uint32_t p = (uint32_t)recv3[3 * t + 1], n = (uint32_t)recv3[3 * t + 2];
        u_loc[t] = u;
        keys[t] = (uint32_t)u; vals[t] = (uint32_t)t;
        keys[Bl + t] = key_base + (((p % G) << lb) | (p / G)); vals[Bl + t] = (uint32_t)t;
        keys[2 * Bl + t] = key_base + (((n % G) << lb) | (n / G)); vals[2 * Bl + t] = (uint32_t)(Bl + t);
    }
}

struct head_flag_op {                          // 1 where a sorted position starts a new key (the scan's input, never materialised)
    const uint32_t* keys;
    __device__ uint32_t operator()(uint32_t q) const { return (q == 0u || keys[q] != keys[q - 1]) ? 1u : 0u; }
};

// Section B after the scan of its head flags: uidx[q] = dense index of q's segment (= the distinct item's slot in the request list and in
// both row buffers), umap[occurrence] = the same per occurrence ([0, Bl) positives, [Bl, 2 Bl) negatives), uniq_local[j] = the local row
// asked of the owner, starts[k] = first slot of owner k (no atomics), n_uniq.
__global__ __launch_bounds__(kBlock) void shard_emit_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                            const uint32_t* __restrict__ excl, int64_t n, int world, unsigned lb,
                                                            uint32_t key_base, uint32_t* __restrict__ uidx, int64_t* __restrict__ uniq_local,
                                                            int64_t* __restrict__ umap, int64_t* __restrict__ starts,
                                                            int64_t* __restrict__ n_uniq) {
    const uint32_t lmask = (1u << lb) - 1u;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += (int64_t)gridDim.x * kBlock) {
        const uint32_t key = keys[q] - key_base;
        const uint32_t prev = q > 0 ? keys[q - 1] - key_base : 0u;
        const bool head = q == 0 || prev != key;
        const uint32_t j = excl[q] + (head ? 1u : 0u) - 1u;
        uidx[q] = j;
        umap[perm[q]] = (int64_t)j;
        if (head) {
            uniq_local[j] = (int64_t)(key & lmask);
            const int owner = (int)(key >> lb), before = q > 0 ? (int)(prev >> lb) : -1;
            for (int k = before + 1; k <= owner; ++k) starts[k] = (int64_t)j;
        }
        if (q == n - 1) {
            n_uniq[0] = (int64_t)j + 1;
            for (int k = (int)(key >> lb) + 1; k <= world; ++k) starts[k] = (int64_t)j + 1;
        }
    }
}
__global__ void shard_counts_kernel(int64_t* __restrict__ starts_counts, int world) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int k = 0; k < world; ++k) starts_counts[k] = starts_counts[k + 1] - starts_counts[k];
}

// partials[block] = {sum_t ||U[u_loc[t]]||^2, sum_t nrm2[ip[t]]}: the EmbLoss norms of a rank's routed triples.  The item rows' squared
// norms were formed by their OWNERS while they gathered the rows (cdr_gather_rows_norms) and travelled beside them: 4 bytes per distinct
// row instead of a second pass over the received rows.
template <int LPR>
__global__ __launch_bounds__(kBlock) void shard_norms_kernel(const float* __restrict__ U, int D, const int64_t* __restrict__ uid,
                                                             const float* __restrict__ nrm2, const int64_t* __restrict__ ip, int64_t B,
                                                             double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    constexpr int UNR = 8;
    __shared__ double smem[2 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    double acc[2] = {0.0, 0.0};
    for (int64_t base = gg; base < B; base += TG * UNR) {
        int64_t iu[UNR], ii[UNR];
        float4 u[UNR];
        float pn[UNR];
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ii[r] = ip[tc];
        }
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            u[r] = (t < B && live) ? ld4n<(LPR >= 32)>(U + iu[r] * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            pn[r] = (t < B && sub == 0) ? nrm2[ii[r]] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const float su = group_sum<LPR>(dot4(u[r], u[r]));
            if (sub == 0) { acc[0] += (double)su; acc[1] += (double)pn[r]; }
        }
    }
    block_sum_d<2>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1];
    }
}

// The duplicate ITEM segments of a rank's routed triples: out[uidx[head]] = signed sum of GP over the segment's occurrences, in
// occurrence order (+ for positives, - for negatives) + c_i * #positives * (the received row) -- the one gradient row that goes home for
// that item.  Structure and summation order of rowwise_apply_dups_body (SU segments in flight; third and later occurrences eight at a
// time; long segments registered for the piece kernels); nothing is updated here.
template <int LPR>
__device__ __forceinline__ void segsum_dups_body(int D, const dup_side& t) {
    constexpr int GPB = kBlock / LPR;
    constexpr int SU = 4;
    const uint32_t* __restrict__ keys = t.keys; const uint32_t* __restrict__ perm = t.perm; const uint32_t* __restrict__ heads = t.heads;
    const float* __restrict__ G = t.G; const float* __restrict__ rows = t.W; float* __restrict__ out = t.out;
    const uint32_t* __restrict__ uidx = t.uidx;
    const int64_t n = t.n, neg_start = t.neg_start, reg_limit = t.reg_limit;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = t.reg_coef ? t.reg_coef[0] : 0.f;
    const int64_t nh = (int64_t)t.nheads[0];
    const bool live = sub < D4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t h0 = gg * SU; h0 < nh; h0 += TG * SU) {
        int64_t q[SU]; uint32_t row[SU], k2[SU]; bool ok[SU];
#pragma unroll
        for (int j = 0; j < SU; ++j) { ok[j] = h0 + j < nh; q[j] = ok[j] ? (int64_t)heads[h0 + j] : 0; }
        uint32_t far[SU], o0[SU], o1[SU], ju[SU];
#pragma unroll
        for (int j = 0; j < SU; ++j) {
            row[j] = keys[q[j]];
            k2[j] = q[j] + 2 < n ? keys[q[j] + 2] : ~0u;
            far[j] = q[j] + kLongSeg < n ? keys[q[j] + kLongSeg] : ~0u;
            o0[j] = perm[q[j]]; o1[j] = q[j] + 1 < n ? perm[q[j] + 1] : 0u;
            ju[j] = uidx[q[j]];
        }
        float4 w[SU], g0[SU], g1[SU];
#pragma unroll
        for (int j = 0; j < SU; ++j) {
            ok[j] = ok[j] && !(far[j] == row[j] && row[j] != ~0u && q[j] + kLongSeg < n);
            w[j] = g0[j] = g1[j] = z4;
            if (ok[j] && live) {
                if (c != 0.f) w[j] = ld4n<(LPR >= 32)>(rows + (int64_t)ju[j] * D + 4 * sub);
                const bool n0 = (int64_t)o0[j] >= neg_start, n1 = (int64_t)o1[j] >= neg_start;
                g0[j] = ld4n<(LPR >= 32)>(G + (n0 ? (int64_t)o0[j] - neg_start : (int64_t)o0[j]) * D + 4 * sub);
                g1[j] = ld4n<(LPR >= 32)>(G + (n1 ? (int64_t)o1[j] - neg_start : (int64_t)o1[j]) * D + 4 * sub);
            }
        }
#pragma unroll
        for (int j = 0; j < SU; ++j) {
            if (!ok[j]) continue;
            float4 acc = z4;
            int cnt = 0;
            {
                const bool n0 = (int64_t)o0[j] >= neg_start, n1 = (int64_t)o1[j] >= neg_start;
                if (n0) { acc.x -= g0[j].x; acc.y -= g0[j].y; acc.z -= g0[j].z; acc.w -= g0[j].w; }
                else { acc.x += g0[j].x; acc.y += g0[j].y; acc.z += g0[j].z; acc.w += g0[j].w; }
                cnt += ((int64_t)o0[j] < reg_limit) ? 1 : 0;
                if (n1) { acc.x -= g1[j].x; acc.y -= g1[j].y; acc.z -= g1[j].z; acc.w -= g1[j].w; }
                else { acc.x += g1[j].x; acc.y += g1[j].y; acc.z += g1[j].z; acc.w += g1[j].w; }
                cnt += ((int64_t)o1[j] < reg_limit) ? 1 : 0;
            }
            if (k2[j] == row[j]) {
                const int64_t end = seg_end_probe<LPR>(keys, q[j], n, row[j]);
                const uint32_t occ_l = seg_occ_fetch<LPR>(perm, q[j], end);
                for (int64_t e0 = q[j] + 2; e0 < end; e0 += 8) {
                    int64_t o[8]; float4 g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int64_t ov = seg_occ_at<LPR>(perm, occ_l, q[j], e0 + u < end ? e0 + u : q[j] + 2); o[u] = e0 + u < end ? ov : -1; }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        g[u] = (o[u] >= 0 && live) ? ld4n<(LPR >= 32)>(G + (o[u] >= neg_start ? o[u] - neg_start : o[u]) * D + 4 * sub) : z4;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (o[u] < 0) continue;
                        if (o[u] >= neg_start) { acc.x -= g[u].x; acc.y -= g[u].y; acc.z -= g[u].z; acc.w -= g[u].w; }
                        else { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
                        cnt += (o[u] < reg_limit) ? 1 : 0;
                    }
                }
            }
            const float rc = c * (float)cnt;
            if (live) st4n<(LPR >= 32)>(out + (int64_t)ju[j] * D + 4 * sub, make_float4(__builtin_fmaf(rc, w[j].x, acc.x), __builtin_fmaf(rc, w[j].y, acc.y),
                                                                          __builtin_fmaf(rc, w[j].z, acc.z), __builtin_fmaf(rc, w[j].w, acc.w)));
        }
    }
    if (t.counters == nullptr) return;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q + kLongSeg < n; q += (int64_t)gridDim.x * kBlock) {
        const uint32_t row = keys[q];
        const uint32_t before = keys[q > 0 ? q - 1 : 0];
        const uint32_t far = keys[q + kLongSeg];
        if ((q > 0 && before == row) || far != row) continue;
        int64_t lo = q + kLongSeg, hi = n;
        while (lo + 1 < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] == row) lo = mid; else hi = mid;
        }
        const int64_t len = hi - q;
        const unsigned np = (unsigned)((len + kPiece - 1) / kPiece);
        const unsigned base = atomicAdd(&t.counters[0], np);
        const unsigned li = atomicAdd(&t.counters[1], 1u);
        t.longs[li] = seg_long{q, len, (int64_t)base};
        for (unsigned k = 0; k < np; ++k) {
            const int64_t st = q + (int64_t)k * kPiece;
            t.pieces[base + k] = seg_piece{st, (hi - st) < kPiece ? (hi - st) : (int64_t)kPiece};
        }
    }
}
template <int LPR>
__device__ __forceinline__ void segsum_long_finish_body(int D, const dup_side& t) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = t.reg_coef ? t.reg_coef[0] : 0.f;
    const int64_t nl = t.counters[1];
    for (int64_t li = gg; li < nl; li += TG) {
        const seg_long sg = t.longs[li];
        const int64_t j = t.uidx[sg.head];
        const int64_t np = (sg.len + kPiece - 1) / kPiece;
        for (int ch = sub; ch < D4; ch += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnt = 0;
            for (int64_t k0 = 0; k0 < np; k0 += 8) {
                float4 g[8]; int cc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = k0 + u < np;
                    g[u] = in ? ld4n<(LPR >= 32)>(t.partial + (sg.base + k0 + u) * D + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    cc[u] = in ? t.pcnt[sg.base + k0 + u] : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + u < np) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; cnt += cc[u]; }
            }
            const float rc = c * (float)cnt;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c != 0.f) w = ld4n<(LPR >= 32)>(t.W + j * D + 4 * ch);
            st4n<(LPR >= 32)>(t.out + j * D + 4 * ch, make_float4(__builtin_fmaf(rc, w.x, acc.x), __builtin_fmaf(rc, w.y, acc.y),
                                                    __builtin_fmaf(rc, w.z, acc.z), __builtin_fmaf(rc, w.w, acc.w)));
        }
    }
}
// blockIdx.y = 0: the duplicate USER rows (updated in place, as in the one-GPU step); 1: the duplicate ITEM segments (summed for the way home)
template <int LPR, int OPT>
__global__ __launch_bounds__(kBlock) void shard_dups_kernel(int D, dup_side a, dup_side b) {
    if (blockIdx.y == 0)
        rowwise_apply_dups_body<LPR, OPT, true>(a.W, a.M, a.V, D, a.keys, a.perm, a.n, a.heads, a.nheads, a.G, a.neg_start, a.reg_limit, a.reg_coef, a.hp,
                                                a.counters, a.longs, a.pieces);
    else segsum_dups_body<LPR>(D, b);
}
template <int LPR, int OPT>
__global__ __launch_bounds__(kBlock) void shard_long_finish_kernel(int D, dup_side a, dup_side b) {
    if (blockIdx.y == 0) {
        if (a.counters) seg_long_finish_body<LPR, OPT>(a.W, a.M, a.V, D, a.keys, a.reg_coef, a.hp, a.counters, a.longs, a.partial, a.pcnt);
    } else if (b.counters) segsum_long_finish_body<LPR>(D, b);
}
// out9[6] = this rank's loss sum (out9[7..8] keep the all-reduced norm sums the caller put there) + both sides' long-segment counters cleared
__global__ __launch_bounds__(kBlock) void shard_sums2_kernel(const double* __restrict__ partials, int nblocks, float* __restrict__ out9,
                                                             unsigned* __restrict__ zero_a, unsigned* __restrict__ zero_b) {
    __shared__ double smem[3 * (kBlock / 64)];
    if (zero_a && threadIdx.x >= 64 && threadIdx.x < 68) zero_a[threadIdx.x - 64] = 0u;
    if (zero_b && threadIdx.x >= 128 && threadIdx.x < 132) zero_b[threadIdx.x - 128] = 0u;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2];
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) out9[6] = (float)acc[0];
}
// The owner's apply of ONE ascending duplicate-free run (what a single requester sends): position q updates row ids[q] with gradient row
// G[q] -- no keys, no permutation, no segments.  Two rows per lane group in flight, all four streams of a row (w, m, v, g) requested together.
template <int LPR, int OPT>
__global__ __launch_bounds__(kBlock) void apply_run_kernel(float* __restrict__ W, float* __restrict__ Mo, float* __restrict__ Vo, int D,
                                                           const int64_t* __restrict__ ids, int64_t n, const float* __restrict__ G, apply_hp hp) {
    HP_FROM_DEV(hp);
    constexpr int GPB = kBlock / LPR, UN = 2;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = gg; base < n; base += TG * UN) {
        int64_t off[UN]; float4 w[UN], m[UN], v[UN], g[UN]; bool ok[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t q = base + j * TG;
            ok[j] = q < n && live;
            off[j] = ids[q < n ? q : n - 1] * D + 4 * sub;
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t q = base + j * TG;
            w[j] = m[j] = v[j] = g[j] = z4;
            if (ok[j]) {
                w[j] = ld4n<(LPR >= 32)>(W + off[j]); g[j] = ld4n<(LPR >= 32)>(G + q * D + 4 * sub);
                if (OPT == 1) { m[j] = ld4n<(LPR >= 32)>(Mo + off[j]); v[j] = ld4n<(LPR >= 32)>(Vo + off[j]); }
            }
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            if (!ok[j]) continue;
            // (0 + g first: the segmented apply's accumulator starts at zero -- same bits, also for g = -0)
            const float4 acc = make_float4(0.f + g[j].x, 0.f + g[j].y, 0.f + g[j].z, 0.f + g[j].w);
            const float4 wn = upd_math<OPT>(w[j], m[j], v[j], acc, 0.f, hp);
            if (OPT == 1) { st4n<(LPR >= 32)>(Mo + off[j], m[j]); st4n<(LPR >= 32)>(Vo + off[j], v[j]); }
            st4n<(LPR >= 32)>(W + off[j], wn);
        }
    }
}

// keys / perm of a list that is ALREADY one ascending duplicate-free run (what one requester sends an owner): nothing to sort
__global__ __launch_bounds__(kBlock) void sorted_run_keys_kernel(const int64_t* __restrict__ ids, int64_t n, uint32_t* __restrict__ keys,
                                                                 uint32_t* __restrict__ perm) {
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += (int64_t)gridDim.x * kBlock) {
        keys[q] = (uint32_t)ids[q]; perm[q] = (uint32_t)q;
    }
}

// ================================================================================================ round 6: ids without a sort (medium batches)
// At 65,536 triples the step's id work -- make_keys + nine rocPRIM launches + occurrence flags, 72 us -- was a third of the step, to find
// the few hundred duplicate occurrences of a uniform batch.  With one uint32 counter per table row (all zero between steps):
//   batch_norms_count_kernel   the EmbLoss norms pass also counts every occurrence (atomicAdd, integer: order-independent) and fills the
//                              duplicate arrays with the sentinel key
//   count_flags_kernel         per triple: a row whose counter reads 1 occurs once (the same flag byte occ_flags_kernel derives from the
//                              sorted keys); every other occurrence is appended to a list as {key, occurrence}
//   count_sort_kernel          block 0 sorts that list by (key, occurrence) -- in LDS up to 16,384 entries, in global memory beyond (slow,
//                              correct: the host binding moves a stream with that many duplicates back to the sorted path) -- and writes
//                              it out in the sorted path's layout (users at keys[0..), items at keys[B..), heads of every run); the other
//                              blocks put the counters back to zero.
// Same flags, same segments in the same occurrence order as the sorted path: the forward-and-update kernel and the segmented applies
// behind it run unchanged, on identical operands -- bit-equal tables.  (Hashing the ids into a small table instead of one counter per row
// was measured first, tools/r06/mb_atomics.hip / profiles/r06_mb_atomics.txt: 33.9 us for the insert of 196,608 keys alone.)
constexpr int64_t kCountMinB = 16448, kCountMaxB = 131072;    // (profiles/r06_mb_idpath.json: -29 % / -22 % / -9 % of the step at 32,768 / 65,536 / 131,072 uniform triples)
constexpr int kCountLds = 16384;                      // duplicate occurrences block 0 sorts in LDS (128 KB of {key, occurrence} words)

template <int LPR, bool NORMS>
__global__ __launch_bounds__(kBlock) void batch_norms_count_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                   const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                                   const int64_t* __restrict__ nid, int64_t B, uint32_t* __restrict__ cu,
                                                                   uint32_t* __restrict__ ci, uint32_t* __restrict__ keysD,
                                                                   uint64_t* __restrict__ list_hdr, double* __restrict__ partials) {
    constexpr int GPB = kBlock / LPR;
    constexpr int UNR = 8;
    __shared__ double smem[2 * (kBlock / 64)];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < 3 * B; i += (int64_t)gridDim.x * kBlock) keysD[i] = ~0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) list_hdr[0] = 0ull;          // the users' share of the duplicate list (count_flags_kernel)
    double acc[2] = {0.0, 0.0};
    for (int64_t base = gg; base < B; base += TG * UNR) {
        int64_t iu[UNR], ip[UNR], in[UNR];
        float4 u[UNR], p[UNR];
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            const int64_t tc = t < B ? t : B - 1;
            iu[r] = uid[tc]; ip[r] = pid[tc]; in[r] = nid[tc];
        }
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int64_t t = base + (int64_t)r * TG;
            u[r] = p[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NORMS && t < B && live) { u[r] = ld4(U + iu[r] * D + 4 * sub); p[r] = ld4(I + ip[r] * D + 4 * sub); }
            if (t < B && sub == 0) { atomicAdd(cu + iu[r], 1u); atomicAdd(ci + ip[r], 1u); atomicAdd(ci + in[r], 1u); }
        }
        if (NORMS) {
#pragma unroll
            for (int r = 0; r < UNR; ++r) {
                const float su = group_sum<LPR>(dot4(u[r], u[r])), sp = group_sum<LPR>(dot4(p[r], p[r]));
                if (sub == 0) { acc[0] += (double)su; acc[1] += (double)sp; }
            }
        }
    }
    if (NORMS) {
        block_sum_d<2>(acc, smem);
        if (threadIdx.x == 0) {
            double* o = partials + (size_t)blockIdx.x * CDR_PARTIAL_STRIDE;
            o[0] = acc[0]; o[1] = acc[1];
        }
    }
}

// cnt[2] += duplicate occurrences (the list's length), cnt[3] = max(the largest counter seen): the host binding's statistics.
// list = {header: hdr[0] = the users' share of the list | 7 spare words | entries {key << 32 | occurrence}}
constexpr int kListHdr = 8;
__global__ __launch_bounds__(kBlock) void count_flags_kernel(const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                             const int64_t* __restrict__ nid, int64_t B, const uint32_t* __restrict__ cu,
                                                             const uint32_t* __restrict__ ci, uint32_t key_base, uint32_t* __restrict__ flags4,
                                                             uint64_t* __restrict__ list_hdr, unsigned* __restrict__ cnt) {
    constexpr int NW = kBlock / 64;
    __shared__ unsigned wtot[NW], wbase[NW], wmax[NW], wusr[NW];
    uint64_t* __restrict__ list = list_hdr + kListHdr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t t0 = (int64_t)blockIdx.x * kBlock; t0 < B; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t t = t0 + threadIdx.x;
        uint32_t u = 0, p = 0, n = 0, c0 = 1, c1 = 1, c2 = 1;
        if (t < B) {
            u = (uint32_t)uid[t]; p = (uint32_t)pid[t]; n = (uint32_t)nid[t];
            c0 = cu[u]; c1 = ci[p]; c2 = ci[n];
            flags4[t] = (c0 == 1u ? 1u : 0u) | (c1 == 1u ? 0x100u : 0u) | (c2 == 1u ? 0x10000u : 0u);
        }
        const unsigned mine = (c0 != 1u) + (c1 != 1u) + (c2 != 1u);
        unsigned incl = mine, mx = c0 > c1 ? (c0 > c2 ? c0 : c2) : (c1 > c2 ? c1 : c2), usr = c0 != 1u ? 1u : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned a = __shfl_up(incl, d, 64), b = __shfl_xor(mx, d, 64), c = __shfl_xor(usr, d, 64);
            if (lane >= d) incl += a;
            mx = mx > b ? mx : b;
            usr += c;
        }
        if (lane == 63) { wtot[wave] = incl; wmax[wave] = mx; wusr[wave] = usr; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0, m = 0, ut = 0;
            for (int w = 0; w < NW; ++w) { wbase[w] = tot; tot += wtot[w]; m = m > wmax[w] ? m : wmax[w]; ut += wusr[w]; }
            const unsigned b = tot ? atomicAdd(&cnt[2], tot) : 0u;
            for (int w = 0; w < NW; ++w) wbase[w] += b;
            if (m > 1u) atomicMax(&cnt[3], m);
            if (ut) atomicAdd((unsigned long long*)&list_hdr[0], (unsigned long long)ut);
        }
        __syncthreads();
        unsigned at = wbase[wave] + incl - mine;
        if (c0 != 1u) list[at++] = ((uint64_t)u << 32) | (uint32_t)t;
        if (c1 != 1u) list[at++] = ((uint64_t)(key_base + p) << 32) | (uint32_t)t;
        if (c2 != 1u) list[at++] = ((uint64_t)(key_base + n) << 32) | (uint32_t)(B + t);
        __syncthreads();
    }
}

// blocks [0, kRankBlocks): the duplicate list into sorted order.  Up to kCountLds entries by RANK: a block owns 64 entries, one WAVE per
// sixteenth of the list each, and counts for every entry the entries below it ((key, occurrence) words are distinct) while the whole list
// passes through LDS in tiles -- the n^2 compares spread over the chip (first cut: 256 entries per block, 1,741 entries on SEVEN CUs, 25 us;
// a one-workgroup bitonic network before that: 35 us at 2,048 entries, 250 us at 8,192); an entry with no smaller occurrence of its key
// heads its segment.  Beyond kCountLds entries block 0 alone sorts in global memory (slow, correct).
// blocks [kRankBlocks, ...): every counter this batch touched back to zero (the flags have been taken).
constexpr int kRankEPB = 64, kRankParts = 1024 / kRankEPB, kRankBlocks = kCountLds / kRankEPB, kRankTile = 2048;
__global__ __launch_bounds__(1024) void count_sort_kernel(const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                          const int64_t* __restrict__ nid, int64_t B, uint32_t* __restrict__ cu,
                                                          uint32_t* __restrict__ ci, uint32_t key_base, uint64_t* __restrict__ list_hdr,
                                                          uint32_t* __restrict__ keysD, uint32_t* __restrict__ permD,
                                                          uint32_t* __restrict__ headsA, uint32_t* __restrict__ headsB,
                                                          unsigned* __restrict__ cnt) {
    if ((int)blockIdx.x >= kRankBlocks) {
        const int64_t nth = (int64_t)(gridDim.x - kRankBlocks) * 1024;
        for (int64_t t = (int64_t)(blockIdx.x - kRankBlocks) * 1024 + threadIdx.x; t < B; t += nth) { cu[uid[t]] = 0u; ci[pid[t]] = 0u; ci[nid[t]] = 0u; }
        return;
    }
    __shared__ uint64_t tile[kRankTile];
    __shared__ unsigned part_rank[kRankParts][kRankEPB], part_dup[kRankParts][kRankEPB];
    __shared__ unsigned nA_s, hA_s, hB_s;
    uint64_t* __restrict__ list = list_hdr + kListHdr;
    const unsigned nd = cnt[2];
    if (nd == 0u) return;                             // cnt[0] = cnt[1] = 0 already (coef_finish_kernel)
    if (nd <= (unsigned)kCountLds) {
        const unsigned e0 = blockIdx.x * (unsigned)kRankEPB;
        if (e0 >= nd) return;
        const unsigned le = threadIdx.x % kRankEPB, part = threadIdx.x / kRankEPB;
        const bool have = e0 + le < nd;
        const uint64_t my = have ? list[e0 + le] : ~0ull;
        const uint32_t mykey = (uint32_t)(my >> 32);
        unsigned rank = 0, dup = 0;
        for (unsigned t0 = 0; t0 < nd; t0 += kRankTile) {
            for (unsigned i = threadIdx.x; i < (unsigned)kRankTile; i += 1024) tile[i] = t0 + i < nd ? list[t0 + i] : ~0ull;
            __syncthreads();
            const unsigned jb = part * (kRankTile / kRankParts);
#pragma unroll 8
            for (unsigned j = 0; j < (unsigned)(kRankTile / kRankParts); ++j) {
                const uint64_t x = tile[jb + j];
                const bool below = x < my;
                rank += below ? 1u : 0u;
                dup |= (below && (uint32_t)(x >> 32) == mykey) ? 1u : 0u;
            }
            __syncthreads();
        }
        part_rank[part][le] = rank; part_dup[part][le] = dup;
        if (threadIdx.x == 0) { hA_s = 0u; hB_s = 0u; }
        __syncthreads();
        // heads are numbered inside the workgroup first (LDS), one reservation per list and workgroup on the global counters: a returning
        // atomic per head on ONE word serialises at ~12 ns each (870 heads: 10 us of this launch's first 27)
        unsigned r = 0, nA = 0, slot = 0;
        bool head = false, isA = false;
        if (part == 0 && have) {
            unsigned dd = 0;
#pragma unroll
            for (int q = 0; q < kRankParts; ++q) { r += part_rank[q][le]; dd |= part_dup[q][le]; }
            head = dd == 0u;
            nA = (unsigned)list_hdr[0];
            isA = r < nA;
            if (isA) { keysD[r] = mykey; permD[r] = (uint32_t)my; }
            else { keysD[B + (r - nA)] = mykey; permD[B + (r - nA)] = (uint32_t)my; }
            if (head) slot = atomicAdd(isA ? &hA_s : &hB_s, 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) { nA_s = hA_s ? atomicAdd(&cnt[0], hA_s) : 0u; part_rank[0][0] = hB_s ? atomicAdd(&cnt[1], hB_s) : 0u; }
        __syncthreads();
        if (head) {
            if (isA) headsA[nA_s + slot] = r;
            else headsB[part_rank[0][0] + slot] = r - nA;
        }
        return;
    }
    if (blockIdx.x != 0) return;
    // ---- a long list (a skewed stream the host binding has not yet moved to the sorted path): bitonic network in global memory
    unsigned P = 1024;
    while (P < nd) P <<= 1;
    uint64_t* A = list;
    for (unsigned i = nd + threadIdx.x; i < P; i += 1024) list[i] = ~0ull;
    if (threadIdx.x == 0) { hA_s = 0u; hB_s = 0u; }
    __syncthreads();
    for (unsigned k = 2; k <= P; k <<= 1) {
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned i = threadIdx.x; i < P; i += 1024) {
                const unsigned l = i ^ j;
                if (l > i) {
                    const uint64_t a = A[i], b = A[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { A[i] = b; A[l] = a; }
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) nA_s = (unsigned)list_hdr[0];
    __syncthreads();
    const unsigned nA = nA_s;
    for (unsigned i = threadIdx.x; i < nd; i += 1024) {
        const uint64_t e = A[i];
        const uint32_t key = (uint32_t)(e >> 32), occ = (uint32_t)e;
        const bool head = i == 0u || (uint32_t)(A[i - 1] >> 32) != key;
        if (i < nA) {
            keysD[i] = key; permD[i] = occ;
            if (head) headsA[atomicAdd(&hA_s, 1u)] = i;
        } else {
            keysD[B + (i - nA)] = key; permD[B + (i - nA)] = occ;
            if (head) headsB[atomicAdd(&hB_s, 1u)] = i - nA;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { cnt[0] = hA_s; cnt[1] = hB_s; }
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

extern "C" int cdr_bpr_fwd_grad(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D,
                                const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, int64_t B_mean,
                                float gamma, float reg_weight, float* out6, float* GU, float* GP, int scatter) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out6 && GU && GP);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    if (B_mean <= 0) B_mean = B;
    const float invB = 1.0f / (float)B_mean;
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_for((B + kUnroll - 1) / kUnroll, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_GRAD, s);
        if (scatter) {
            DISPATCH_LPR(lpr, bpr_fwd_grad_kernel<L, true><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid,
                                                                                                 B, gamma, invB, GU, GP, ctx->partials));
        } else {
            DISPATCH_LPR(lpr, bpr_fwd_grad_kernel<L, false><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid,
                                                                                                  B, gamma, invB, GU, GP, ctx->partials));
        }
    }
    CDR_LAUNCH_CHECK();
    step_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B_mean, reg_weight, out6);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_point_fwd_grad(cdr_ctx* ctx, void* stream, int loss_kind, const float* user_tab, const float* item_tab, int D,
                                 const int64_t* uid, const int64_t* iid, const float* label, int64_t B, float reg_weight,
                                 float* out9, float* GU, float* GI) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && label && out9 && GU && GI);
    CDR_CHECK_ARG((loss_kind == CDR_LOSS_MSE || loss_kind == CDR_LOSS_BCE) && D > 0 && (D & 3) == 0 && D <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_for((B + kUnroll - 1) / kUnroll, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_POINT_FWD_GRAD, s);
        DISPATCH_LPR(lpr, point_fwd_grad_kernel<L><<<dim3(grid), dim3(kBlock), 0, s>>>(loss_kind, user_tab, item_tab, D, uid, iid, label, B,
                                                                                        1.0f / (float)B, GU, GI, ctx->partials));
    }
    CDR_LAUNCH_CHECK();
    step_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, out9);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_loss_finish_sums(void* stream, const float* sums3, int64_t B_mean, float reg_weight, float* out6) {
    CDR_CHECK_ARG(sums3 && out6 && B_mean > 0);
    finish_sums_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(sums3, B_mean, reg_weight, out6);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// rocPRIM's Onesweep with a configuration measured for this step's shape (tools/sort_tune.hip: 3,145,728 (key, index) pairs,
// 27 significant bits): 1,024-thread blocks x 8 items, 9-bit digits (27 bits = 3 passes instead of 4), wave-match ranking --
// 117 us against 161 us for the library's gfx950 default.  Used from 2^18 pairs up; smaller sorts keep the default.
using big_sort_config = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 9,
                                        rocprim::block_radix_rank_algorithm::match>,
    (size_t)1 << 17>;          // (the library merge-sorts up to 2^20 items: 0.167 ms for 1,048,576 pairs against 0.06 with Onesweep)
constexpr int64_t kBigSort = 1 << 18;

static inline hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                                    size_t n, unsigned bits, hipStream_t s) {
    if ((int64_t)n >= kBigSort) return rocprim::radix_sort_pairs<big_sort_config>(tmp, tmp_bytes, kin, kout, vin, vout, n, 0u, bits, s);
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0u, bits, s);
}

static inline unsigned bits_for(int64_t num_rows) {
    unsigned b = 1;
    while (b < 32 && ((int64_t)1 << b) < num_rows) ++b;
    return b;
}

extern "C" int cdr_sort_workspace_bytes(int64_t n, int64_t num_rows, size_t* bytes) {
    CDR_CHECK_ARG(bytes && n > 0 && num_rows > 0 && num_rows <= (int64_t)0xFFFFFFFFu && n <= (int64_t)0x7FFFFFFF);
    size_t tmp = 0;
    hipError_t e = sort_pairs(nullptr, tmp, nullptr, nullptr, nullptr, nullptr, (size_t)n, bits_for(num_rows), (hipStream_t)0);
    if (e != hipSuccess) { cdr_set_error("cdr_sort_workspace_bytes: %s", hipGetErrorString(e)); return (int)e; }
    tmp = (tmp + 255) & ~(size_t)255;
    *bytes = tmp + 2 * (((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255);
    return CDR_OK;
}

extern "C" int cdr_sort_ids(cdr_ctx* ctx, void* stream, const int64_t* ids0, int64_t n0, const int64_t* ids1, int64_t n1,
                            int64_t num_rows, uint32_t* keys_sorted, uint32_t* perm, void* workspace,
                            size_t workspace_bytes) {
    const int64_t n = n0 + n1;
    CDR_CHECK_ARG(ids0 && n0 > 0 && (n1 == 0 || ids1) && keys_sorted && perm && workspace);
    size_t need = 0;
    int rc = cdr_sort_workspace_bytes(n, num_rows, &need);
    if (rc) return rc;
    CDR_CHECK_ARG(workspace_bytes >= need);
    hipStream_t s = (hipStream_t)stream;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t* keys_in = (uint32_t*)workspace;
    uint32_t* vals_in = (uint32_t*)((char*)workspace + arr);
    void* tmp = (char*)workspace + 2 * arr;
    size_t tmp_bytes = workspace_bytes - 2 * arr;
    cdr_time_scope ts(ctx, CDR_TAG_SORT, s);
    make_keys_kernel<<<dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s>>>(ids0, n0, ids1, n1, keys_in, vals_in);
    CDR_LAUNCH_CHECK();
    CDR_HIP(sort_pairs(tmp, tmp_bytes, keys_in, keys_sorted, vals_in, perm, (size_t)n, bits_for(num_rows), s));
    return CDR_OK;
}

extern "C" int cdr_sort_ids_two_tables(cdr_ctx* ctx, void* stream, const int64_t* ids_a, int64_t n_a, int64_t rows_a,
                                       const int64_t* ids_b0, int64_t n_b0, const int64_t* ids_b1, int64_t n_b1, int64_t rows_b,
                                       uint32_t* keys_sorted, uint32_t* perm, uint32_t* key_base_out, void* workspace,
                                       size_t workspace_bytes) {
    const int64_t n = n_a + n_b0 + n_b1;
    CDR_CHECK_ARG(ids_a && n_a > 0 && ids_b0 && n_b0 > 0 && (n_b1 == 0 || ids_b1) && keys_sorted && perm && key_base_out && workspace);
    CDR_CHECK_ARG(rows_a > 0 && rows_b > 0);
    const unsigned hb = bits_for(rows_a) > bits_for(rows_b) ? bits_for(rows_a) : bits_for(rows_b);
    CDR_CHECK_ARG(hb < 31);
    const uint32_t key_base = 1u << hb;
    size_t need = 0;
    int rc = cdr_sort_workspace_bytes(n, (int64_t)key_base * 2, &need);
    if (rc) return rc;
    CDR_CHECK_ARG(workspace_bytes >= need);
    hipStream_t s = (hipStream_t)stream;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t* keys_in = (uint32_t*)workspace;
    uint32_t* vals_in = (uint32_t*)((char*)workspace + arr);
    void* tmp = (char*)workspace + 2 * arr;
    size_t tmp_bytes = workspace_bytes - 2 * arr;
    cdr_time_scope ts(ctx, CDR_TAG_SORT, s);
    make_keys2_kernel<<<dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s>>>(ids_a, n_a, ids_b0, n_b0, ids_b1, n_b1, key_base, keys_in, vals_in);
    CDR_LAUNCH_CHECK();
    CDR_HIP(sort_pairs(tmp, tmp_bytes, keys_in, keys_sorted, vals_in, perm, (size_t)n, hb + 1, s));
    *key_base_out = key_base;
    return CDR_OK;
}

extern "C" int cdr_rowwise_apply(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D,
                                 const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, const float* G,
                                 int64_t neg_start, int64_t reg_limit, const float* reg_coef, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int64_t step, const int64_t* occ_ids, uint32_t key_base) {
    CDR_CHECK_ARG(table && keys_sorted && perm && G && n > 0);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && exp_avg && exp_avg_sq && step > 0));
    hipStream_t s = (hipStream_t)stream;
    // keys of a two-table sort carry the table bit: instead of subtracting it per row in the kernels, the three table
    // pointers are moved back by key_base rows here (row r of the table is then addressed as key = key_base + r)
    if (key_base) {
        table -= (int64_t)key_base * D;
        if (exp_avg) exp_avg -= (int64_t)key_base * D;
        if (exp_avg_sq) exp_avg_sq -= (int64_t)key_base * D;
    }
    float step_size = lr, bc2_sqrt = 1.f;
    if (opt == 1) {
        cdr_adam_hp((double)step, lr, beta1, beta2, step_size, bc2_sqrt);       // (bc2_sqrt carries cdr_adam_hp's bc2: cdr_adam_math.h)
    }
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_for(n, kBlock / lpr);
    const bool is_signed = neg_start < n;
    const apply_hp hp{lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt};
    // long-segment scratch: counters | seg_long[long_cap] | seg_piece[piece_cap] | int pcnt[piece_cap] | float partial[piece_cap][D]
    const bool may_have_long = n > kLongSeg;
    unsigned* counters = nullptr; seg_long* longs = nullptr; seg_piece* pieces = nullptr; int* pcnt = nullptr; float* partial = nullptr;
    int64_t long_cap = 0, piece_cap = 0;
    if (may_have_long) {
        CDR_CHECK_ARG(ctx != nullptr);
        long_cap = n / (kLongSeg + 1) + 1;
        piece_cap = n / kPiece + long_cap + 1;
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_long = 256, o_piece = o_long + up(sizeof(seg_long) * long_cap), o_cnt = o_piece + up(sizeof(seg_piece) * piece_cap),
                     o_part = o_cnt + up(sizeof(int) * piece_cap), total = o_part + sizeof(float) * (size_t)piece_cap * D;
        void* base = nullptr;
        int rc = cdr_ctx_scratch(ctx, total, &base);
        if (rc != CDR_OK) return rc;
        counters = (unsigned*)base; longs = (seg_long*)((char*)base + o_long); pieces = (seg_piece*)((char*)base + o_piece);
        pcnt = (int*)((char*)base + o_cnt); partial = (float*)((char*)base + o_part);
        CDR_HIP(cdr_zero_u32(counters, 4, s));
    }
    cdr_time_scope ts(ctx, is_signed ? CDR_TAG_APPLY_SIGNED : CDR_TAG_APPLY_UNSIGNED, s);
#define APPLY_ARGS table, exp_avg, exp_avg_sq, D, keys_sorted, perm, n, G, neg_start, reg_limit, reg_coef, hp, occ_ids, counters, longs, pieces
    if (opt == 0 && !is_signed) { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 0, false><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
    else if (opt == 0) { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 0, true><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
    else if (!is_signed) { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 1, false><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
    else { DISPATCH_LPR(lpr, rowwise_apply_kernel<L, 1, true><<<dim3(grid), dim3(kBlock), 0, s>>>(APPLY_ARGS)); }
#undef APPLY_ARGS
    CDR_LAUNCH_CHECK();
    if (may_have_long) {
        // sized for the worst case, but a launch whose counters read 0 retires in a few microseconds
        const int gp = (int)(piece_cap < 2048 ? piece_cap : 2048);          // one workgroup per piece (looping past 2,048)
        if (is_signed) { DISPATCH_LPR(lpr, seg_piece_sum_kernel<L, true><<<dim3(gp), dim3(kBlock), 0, s>>>(D, perm, G, neg_start, reg_limit, occ_ids, counters, pieces, partial, pcnt)); }
        else { DISPATCH_LPR(lpr, seg_piece_sum_kernel<L, false><<<dim3(gp), dim3(kBlock), 0, s>>>(D, perm, G, neg_start, reg_limit, occ_ids, counters, pieces, partial, pcnt)); }
        CDR_LAUNCH_CHECK();
        const int gl = grid_for(long_cap < 4096 ? long_cap : 4096, kBlock / lpr);
        if (opt == 0) { DISPATCH_LPR(lpr, seg_long_finish_kernel<L, 0><<<dim3(gl), dim3(kBlock), 0, s>>>(table, exp_avg, exp_avg_sq, D, keys_sorted, reg_coef, hp, counters, longs, partial, pcnt)); }
        else { DISPATCH_LPR(lpr, seg_long_finish_kernel<L, 1><<<dim3(gl), dim3(kBlock), 0, s>>>(table, exp_avg, exp_avg_sq, D, keys_sorted, reg_coef, hp, counters, longs, partial, pcnt)); }
        CDR_LAUNCH_CHECK();
    }
    return CDR_OK;
}

// ------------------------------------------------------------------------------------------------ the fused step (round 3)
namespace {

// Both tables' duplicate-row applies of a fused step: one scratch request carved for the two sides, three launches with blockIdx.y =
// side instead of six (+ two counter clears, which the caller folds into step_finish_keep_kernel: dups_plan first, then that launch).
struct dup_host { float* table; float* m; float* v; const uint32_t* keys; const uint32_t* perm; int64_t n; const uint32_t* heads; const unsigned* nheads;
                  const float* G; int64_t neg_start, reg_limit; const float* reg_coef; apply_hp hp; uint32_t key_base;
                  float* out = nullptr; const uint32_t* uidx = nullptr; };
struct dups_plan { dup_side side[2]; int64_t long_cap[2], piece_cap[2]; };

static int dups_plan_make(cdr_ctx* ctx, int D, const dup_host (&h)[2], dups_plan& pl) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off[2][5], total = 0;
    for (int i = 0; i < 2; ++i) {
        pl.long_cap[i] = pl.piece_cap[i] = 0;
        if (h[i].n > kLongSeg) {
            pl.long_cap[i] = h[i].n / (kLongSeg + 1) + 1;
            pl.piece_cap[i] = h[i].n / kPiece + pl.long_cap[i] + 1;
            off[i][0] = total; total += 256;
            off[i][1] = total; total += up(sizeof(seg_long) * pl.long_cap[i]);
            off[i][2] = total; total += up(sizeof(seg_piece) * pl.piece_cap[i]);
            off[i][3] = total; total += up(sizeof(int) * pl.piece_cap[i]);
            off[i][4] = total; total += up(sizeof(float) * (size_t)pl.piece_cap[i] * D);
        }
    }
    char* base = nullptr;
    if (total) {
        void* b = nullptr;
        int rc = cdr_ctx_scratch(ctx, total, &b);
        if (rc != CDR_OK) return rc;
        base = (char*)b;
    }
    for (int i = 0; i < 2; ++i) {
        dup_side& t = pl.side[i];
        const int64_t kb = (int64_t)h[i].key_base * D;
        t.W = h[i].table - kb; t.M = h[i].m ? h[i].m - kb : nullptr; t.V = h[i].v ? h[i].v - kb : nullptr;
        t.keys = h[i].keys; t.perm = h[i].perm; t.n = h[i].n; t.heads = h[i].heads; t.nheads = h[i].nheads; t.G = h[i].G;
        t.neg_start = h[i].neg_start; t.reg_limit = h[i].reg_limit; t.reg_coef = h[i].reg_coef; t.hp = h[i].hp;
        t.counters = nullptr; t.longs = nullptr; t.pieces = nullptr; t.pcnt = nullptr; t.partial = nullptr;
        t.out = h[i].out; t.uidx = h[i].uidx;
        if (pl.long_cap[i]) {
            t.counters = (unsigned*)(base + off[i][0]); t.longs = (seg_long*)(base + off[i][1]); t.pieces = (seg_piece*)(base + off[i][2]);
            t.pcnt = (int*)(base + off[i][3]); t.partial = (float*)(base + off[i][4]);
        }
    }
    return CDR_OK;
}

static int apply_dups_pair(cdr_ctx* ctx, hipStream_t s, int opt, int D, const dups_plan& pl) {
    const int lpr = cdr_lpr_for(D);
    const int64_t nmax = pl.side[0].n > pl.side[1].n ? pl.side[0].n : pl.side[1].n;
    const int grid = grid_for(nmax / 16 + 1, kBlock / lpr);          // four segments per lane group and round (apply_dups)
    {
        cdr_time_scope ts(ctx, CDR_TAG_APPLY_SIGNED, s);
        if (opt == 0) { DISPATCH_LPR(lpr, rowwise_apply_dups2_kernel<L, 0><<<dim3(grid, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
        else { DISPATCH_LPR(lpr, rowwise_apply_dups2_kernel<L, 1><<<dim3(grid, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
    }
    CDR_LAUNCH_CHECK();
    if (pl.long_cap[0] || pl.long_cap[1]) {
        const int64_t pc = pl.piece_cap[0] > pl.piece_cap[1] ? pl.piece_cap[0] : pl.piece_cap[1];
        const int64_t lc = pl.long_cap[0] > pl.long_cap[1] ? pl.long_cap[0] : pl.long_cap[1];
        const int gp = (int)(pc < 2048 ? pc : 2048);                        // one workgroup per piece (looping past 2,048)
        DISPATCH_LPR(lpr, seg_piece_sum2_kernel<L><<<dim3(gp, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1]));
        CDR_LAUNCH_CHECK();
        const int gl = grid_for(lc < 4096 ? lc : 4096, kBlock / lpr);
        if (opt == 0) { DISPATCH_LPR(lpr, seg_long_finish2_kernel<L, 0><<<dim3(gl, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
        else { DISPATCH_LPR(lpr, seg_long_finish2_kernel<L, 1><<<dim3(gl, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
        CDR_LAUNCH_CHECK();
    }
    return CDR_OK;
}

static apply_hp make_hp(int opt, float lr, float beta1, float beta2, float eps, float wd, int64_t step) {
    float step_size = lr, bc2_sqrt = 1.f;
    if (opt == 1) {
        cdr_adam_hp((double)step, lr, beta1, beta2, step_size, bc2_sqrt);       // (bc2_sqrt carries cdr_adam_hp's bc2: cdr_adam_math.h)
    }
    return apply_hp{lr, beta1, beta2, eps, wd, step_size, bc2_sqrt};
}

}  // namespace

extern "C" int cdr_bpr_step_fused_heads_words(int64_t B, int64_t* words) {
    CDR_CHECK_ARG(words && B > 0);
    *words = 4 + (B / 2 + 1) + (B + 1);          // {counters[4] | heads of the user list | heads of the item list}
    return CDR_OK;
}

static int bpr_step_fused_impl(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                               float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D, const int64_t* uid,
                               const int64_t* pid, const int64_t* nid, int64_t B, float gamma, float reg_weight, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int64_t step_user, int64_t step_item,
                               int64_t* step_user_dev, int64_t* step_item_dev, float* hp_dev,
                               float* out9, float* GU, float* GP, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                               void* sort_ws, size_t sort_ws_bytes) {
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    unsigned* cnt = (unsigned*)heads;
    uint32_t* headsA = heads + 4;
    uint32_t* headsB = headsA + (B / 2 + 1);           // (cnt[0..3] are cleared by coef_finish_kernel)
    uint32_t key_base = 0;
    int rc;
    // ---- ids without a sort (medium batches, counters handed over by cdr_ctx_set_id_counters): see "round 6: ids without a sort"
    int64_t P = 1024;
    while (P < 3 * B) P <<= 1;
    const bool count_path = ctx->idc_user && ctx->idc_item && ctx->idc_user_rows == user_rows && ctx->idc_item_rows == item_rows &&
                            B >= kCountMinB && B <= kCountMaxB && ctx->idc_list_bytes >= (size_t)(P + kListHdr) * sizeof(uint64_t);
    if (count_path) {
        const unsigned hb = bits_for(user_rows) > bits_for(item_rows) ? bits_for(user_rows) : bits_for(item_rows);
        CDR_CHECK_ARG(hb < 31);
        key_base = 1u << hb;
        const int ngrid = grid_for((B + 7) / 8, kBlock / lpr);
        {
            cdr_time_scope ts(ctx, CDR_TAG_BATCH_NORMS, s);
            if (reg_weight != 0.f) {
                DISPATCH_LPR(lpr, batch_norms_count_kernel<L, true><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid, B, ctx->idc_user,
                                                                                                      ctx->idc_item, keys, (uint64_t*)ctx->idc_list, ctx->partials));
            } else {
                DISPATCH_LPR(lpr, batch_norms_count_kernel<L, false><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid, B, ctx->idc_user,
                                                                                                       ctx->idc_item, keys, (uint64_t*)ctx->idc_list, ctx->partials));
            }
        }
        CDR_LAUNCH_CHECK();
        coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, reg_weight != 0.f ? ngrid : 0, B, reg_weight, out9, 1, step_user_dev, step_item_dev, hp_dev,
                                                            lr, beta1, beta2, cnt);
        CDR_LAUNCH_CHECK();
        {
            cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
            count_flags_kernel<<<dim3(grid_for(B, kBlock)), dim3(kBlock), 0, s>>>(uid, pid, nid, B, ctx->idc_user, ctx->idc_item, key_base, (uint32_t*)flags,
                                                                                 (uint64_t*)ctx->idc_list, cnt);
            CDR_LAUNCH_CHECK();
            count_sort_kernel<<<dim3(kRankBlocks + 128), dim3(1024), 0, s>>>(uid, pid, nid, B, ctx->idc_user, ctx->idc_item, key_base,
                                                                                            (uint64_t*)ctx->idc_list, keys, perm, headsA, headsB, cnt);
        }
        CDR_LAUNCH_CHECK();
    } else {
    // ---- EmbLoss coefficients first (they do not need the sort): out9[4], out9[5]  (+ the device-resident update counts, when given)
    if (reg_weight != 0.f) {
        const int ngrid = grid_for((B + 7) / 8, kBlock / lpr);
        {
            cdr_time_scope ts(ctx, CDR_TAG_BATCH_NORMS, s);
            DISPATCH_LPR(lpr, batch_norms_kernel<L><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, B, ctx->partials));
        }
        CDR_LAUNCH_CHECK();
        coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, ngrid, B, reg_weight, out9, 1, step_user_dev, step_item_dev, hp_dev, lr, beta1, beta2,
                                                            (unsigned*)heads);
    } else {
        coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, 0, B, 0.f, out9, 1, step_user_dev, step_item_dev, hp_dev, lr, beta1, beta2,
                                                            (unsigned*)heads);
    }
    CDR_LAUNCH_CHECK();
    rc = cdr_sort_ids_two_tables(ctx, stream, uid, B, user_rows, pid, B, nid, B, item_rows, keys, perm, &key_base, sort_ws, sort_ws_bytes);
    if (rc) return rc;
    const int fgrid = grid_for(3 * B, kBlock * kFlagIT);
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(fgrid), dim3(kBlock), 0, s>>>(keys, perm, B, 3 * B, 4, flags, headsA, headsB, cnt, B <= kCountMaxB);
    }
    CDR_LAUNCH_CHECK();
    }
    apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    apply_hp hi = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_item);
    if (hp_dev && opt == 1) { hu.dev = hp_dev; hi.dev = hp_dev + 2; }          // the scalars coef_finish_kernel left on the device
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{item_tab, item_m, item_v};
    static const int un = [] { const char* e = getenv("CDR_FWD_APPLY_UN"); return (e && e[0] == '2') ? 2 : 1; }();   // A/B switch (tools/)
    const int grid = grid_for((B + un - 1) / un, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_APPLY, s);
#define FA_ARGS TU, TI, D, uid, pid, nid, (const uint32_t*)flags, B, gamma, 1.0f / (float)B, out9 + 4, hu, hi, GU, GP, ctx->partials
        if (opt == 0) { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 0, 1><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
        else if (un == 1) { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 1, 1><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
        else { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 1, 2><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
#undef FA_ARGS
    }
    CDR_LAUNCH_CHECK();
    const dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, B, headsA, cnt, GU, B, B, out9 + 4, hu, 0},
                               {item_tab, item_m, item_v, keys + B, perm + B, 2 * B, headsB, cnt + 1, GP, B, B, out9 + 5, hi, key_base}};
    dups_plan pl;
    rc = dups_plan_make(ctx, D, sides, pl);
    if (rc) return rc;
    step_finish_keep_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, out9, pl.side[0].counters, pl.side[1].counters);
    CDR_LAUNCH_CHECK();
    return apply_dups_pair(ctx, s, opt, D, pl);
}

extern "C" int cdr_id_count_workspace_bytes(int64_t B, size_t* bytes) {
    CDR_CHECK_ARG(bytes && B > 0 && 3 * B <= (int64_t)0x7FFFFFFF);
    int64_t P = 1024;
    while (P < 3 * B) P <<= 1;
    *bytes = (size_t)(P + 8) * sizeof(uint64_t);             // 8: the list's header words (kListHdr)
    return CDR_OK;
}

extern "C" int cdr_bpr_step_fused(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                                  float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D, const int64_t* uid,
                                  const int64_t* pid, const int64_t* nid, int64_t B, float gamma, float reg_weight, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int64_t step_user, int64_t step_item,
                                  float* out9, float* GU, float* GP, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                                  void* sort_ws, size_t sort_ws_bytes) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out9 && GU && GP && keys && perm && flags && heads && sort_ws);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && B > 0 && 3 * B <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && item_m && item_v && step_user > 0 && step_item > 0));
    CDR_CHECK_ARG(((uintptr_t)flags & 3) == 0);
    return bpr_step_fused_impl(ctx, stream, opt, user_tab, user_m, user_v, user_rows, item_tab, item_m, item_v, item_rows, D, uid, pid, nid, B, gamma,
                               reg_weight, lr, beta1, beta2, eps, weight_decay, step_user, step_item, nullptr, nullptr, nullptr, out9, GU, GP,
                               keys, perm, flags, heads, sort_ws, sort_ws_bytes);
}

// The same step with the tables' update counts in DEVICE memory (int64 each, advanced by the call's first finishing block) and the Adam
// scalars derived from them on the device (hp_dev: 4 floats of scratch owned by the caller): nothing about the update number is baked
// into the launches, so the call can be captured in a hipGraph and replayed.
extern "C" int cdr_bpr_step_fused_dev(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                                      float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D, const int64_t* uid,
                                      const int64_t* pid, const int64_t* nid, int64_t B, float gamma, float reg_weight, float lr,
                                      float beta1, float beta2, float eps, float weight_decay, int64_t* step_user_dev, int64_t* step_item_dev,
                                      float* hp_dev, float* out9, float* GU, float* GP, uint32_t* keys, uint32_t* perm, uint8_t* flags,
                                      uint32_t* heads, void* sort_ws, size_t sort_ws_bytes) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out9 && GU && GP && keys && perm && flags && heads && sort_ws);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && B > 0 && 3 * B <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(opt == 1 && user_m && user_v && item_m && item_v && step_user_dev && step_item_dev && hp_dev);
    CDR_CHECK_ARG(((uintptr_t)flags & 3) == 0);
    return bpr_step_fused_impl(ctx, stream, opt, user_tab, user_m, user_v, user_rows, item_tab, item_m, item_v, item_rows, D, uid, pid, nid, B, gamma,
                               reg_weight, lr, beta1, beta2, eps, weight_decay, 1, 1, step_user_dev, step_item_dev, hp_dev, out9, GU, GP,
                               keys, perm, flags, heads, sort_ws, sort_ws_bytes);
}


// ---- round 5: the pointwise step (EMCDR-MF / CMF rows: emcdr.py:111-122, cmf.py:75-99) as ONE call on the forward-and-update pass ---------
// Same structure as cdr_bpr_step_fused with two rows per batch row: EmbLoss norms of (U[uid], I[iid]) -> coefficients, ONE sort of both
// id lists + occurrence flags, point_fwd_apply_kernel (rows occurring once updated in place), segmented applies over the duplicate rows.
// Buffers as cdr_bpr_step_fused's with B = the number of (user, item, label) rows: keys / perm [2 B], flags [4 B], heads
// cdr_bpr_step_fused_heads_words(B) words, GU / GI [B, D], sort workspace cdr_sort_workspace_bytes(2 B, ...).
extern "C" int cdr_point_step_fused(cdr_ctx* ctx, void* stream, int loss_kind, int opt, float* user_tab, float* user_m, float* user_v,
                                    int64_t user_rows, float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D,
                                    const int64_t* uid, const int64_t* iid, const float* label, int64_t B, float reg_weight, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int64_t step_user, int64_t step_item,
                                    float* out9, float* GU, float* GI, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                                    void* sort_ws, size_t sort_ws_bytes) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && label && out9 && GU && GI && keys && perm && flags && heads && sort_ws);
    CDR_CHECK_ARG((loss_kind == CDR_LOSS_MSE || loss_kind == CDR_LOSS_BCE) && D > 0 && (D & 3) == 0 && D <= 256 && B > 0 && 2 * B <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && item_m && item_v && step_user > 0 && step_item > 0));
    CDR_CHECK_ARG(((uintptr_t)flags & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    if (reg_weight != 0.f) {
        const int ngrid = grid_for((B + 7) / 8, kBlock / lpr);
        {
            cdr_time_scope ts(ctx, CDR_TAG_BATCH_NORMS, s);
            DISPATCH_LPR(lpr, batch_norms_kernel<L><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, iid, B, ctx->partials));
        }
        CDR_LAUNCH_CHECK();
        coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, ngrid, B, reg_weight, out9, 1, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, (unsigned*)heads);
    } else {
        coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, 0, B, 0.f, out9, 1, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, (unsigned*)heads);
    }
    CDR_LAUNCH_CHECK();
    uint32_t key_base = 0;
    int rc = cdr_sort_ids_two_tables(ctx, stream, uid, B, user_rows, iid, B, nullptr, 0, item_rows, keys, perm, &key_base, sort_ws, sort_ws_bytes);
    if (rc) return rc;
    unsigned* cnt = (unsigned*)heads;
    uint32_t* headsA = heads + 4;
    uint32_t* headsB = headsA + (B / 2 + 1);
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(grid_for(2 * B, kBlock * kFlagIT)), dim3(kBlock), 0, s>>>(keys, perm, B, 2 * B, 4, flags, headsA, headsB, cnt);
    }
    CDR_LAUNCH_CHECK();
    const apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    const apply_hp hi = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_item);
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{item_tab, item_m, item_v};
    const int grid = grid_for(B, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_POINT_FWD_GRAD, s);
#define PA_ARGS loss_kind, TU, TI, D, uid, iid, label, (const uint32_t*)flags, B, 1.0f / (float)B, out9 + 4, hu, hi, GU, GI, ctx->partials
        if (opt == 0) { DISPATCH_LPR(lpr, point_fwd_apply_kernel<L, 0><<<dim3(grid), dim3(kBlock), 0, s>>>(PA_ARGS)); }
        else { DISPATCH_LPR(lpr, point_fwd_apply_kernel<L, 1><<<dim3(grid), dim3(kBlock), 0, s>>>(PA_ARGS)); }
#undef PA_ARGS
    }
    CDR_LAUNCH_CHECK();
    const dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, B, headsA, cnt, GU, B, B, out9 + 4, hu, 0},
                               {item_tab, item_m, item_v, keys + B, perm + B, B, headsB, cnt + 1, GI, B, B, out9 + 5, hi, key_base}};
    dups_plan pl;
    rc = dups_plan_make(ctx, D, sides, pl);
    if (rc) return rc;
    step_finish_keep_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, out9, pl.side[0].counters, pl.side[1].counters);
    CDR_LAUNCH_CHECK();
    return apply_dups_pair(ctx, s, opt, D, pl);
}

// The pointwise step per POSITIVE: uid [S] (the first S entries of recbole's tiled user column), iid [S + S k] = [positives | k-major
// negatives], label [S + S k].  Sizes from cdr_bpr_step_fused_kmajor_sizes(S, k); GU [S, D], GI [S + S k, D]; keys / perm [2 S + S k].
extern "C" int cdr_point_step_fused_kmajor(cdr_ctx* ctx, void* stream, int loss_kind, int opt, float* user_tab, float* user_m, float* user_v,
                                           int64_t user_rows, float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D,
                                           const int64_t* uid, const int64_t* iid, const float* label, int64_t S, int k, float reg_weight,
                                           float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step_user,
                                           int64_t step_item, float* out12, float* GU, float* GI, uint32_t* keys, uint32_t* perm,
                                           uint8_t* flags, uint32_t* heads, void* sort_ws, size_t sort_ws_bytes) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && label && out12 && GU && GI && keys && perm && flags && heads && sort_ws);
    CDR_CHECK_ARG((loss_kind == CDR_LOSS_MSE || loss_kind == CDR_LOSS_BCE) && D > 0 && (D & 3) == 0 && D <= 256 && S > 0 && k >= 1 && k <= 64);
    const int64_t nI = S + S * (int64_t)k, B = nI;
    CDR_CHECK_ARG(S + nI <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && item_m && item_v && step_user > 0 && step_item > 0));
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const int fstride = (2 + k + 3) & ~3;
    const int ngrid = reg_weight != 0.f ? grid_for(S, kBlock / lpr) : 0;
    if (reg_weight != 0.f) {
        cdr_time_scope ts(ctx, CDR_TAG_BATCH_NORMS, s);
        DISPATCH_LPR(lpr, point_norms_kmajor_kernel<L><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, iid, S, k, ctx->partials));
        CDR_LAUNCH_CHECK();
    }
    point_coef_kmajor_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, ngrid, B, k, reg_weight, out12, (unsigned*)heads);
    CDR_LAUNCH_CHECK();
    uint32_t key_base = 0;
    int rc = cdr_sort_ids_two_tables(ctx, stream, uid, S, user_rows, iid, S, iid + S, S * (int64_t)k, item_rows, keys, perm, &key_base, sort_ws, sort_ws_bytes);
    if (rc) return rc;
    unsigned* cnt = (unsigned*)heads;
    uint32_t* headsA = heads + 4;
    uint32_t* headsB = headsA + (S / 2 + 1);
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(grid_for(S + nI, kBlock * kFlagIT)), dim3(kBlock), 0, s>>>(keys, perm, S, S + nI, fstride, flags, headsA, headsB, cnt);
    }
    CDR_LAUNCH_CHECK();
    const apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    const apply_hp hi = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_item);
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{item_tab, item_m, item_v};
    const int grid = grid_for(S, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_POINT_FWD_GRAD, s);
#define PK_ARGS loss_kind, TU, TI, D, uid, iid, label, flags, fstride, S, k, 1.0f / (float)B, out12 + 4, hu, hi, GU, GI, ctx->partials
        if (opt == 0) { DISPATCH_LPR(lpr, point_fwd_apply_kmajor_kernel<L, 0, 2><<<dim3(grid), dim3(kBlock), 0, s>>>(PK_ARGS)); }
        else if (k <= 1) { DISPATCH_LPR(lpr, point_fwd_apply_kmajor_kernel<L, 1, 2><<<dim3(grid), dim3(kBlock), 0, s>>>(PK_ARGS)); }
        else { DISPATCH_LPR(lpr, point_fwd_apply_kmajor_kernel<L, 1, 3><<<dim3(grid), dim3(kBlock), 0, s>>>(PK_ARGS)); }
#undef PK_ARGS
    }
    CDR_LAUNCH_CHECK();
    // duplicate rows: users over GU [S, D] with (1 + k) c_u per list occurrence (out12[9]); items over GI [S + S k, D], every occurrence
    // with c_i (reg_limit = the whole list)
    const dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, S, headsA, cnt, GU, S, S, out12 + 9, hu, 0},
                               {item_tab, item_m, item_v, keys + S, perm + S, nI, headsB, cnt + 1, GI, nI, nI, out12 + 5, hi, key_base}};
    dups_plan pl;
    rc = dups_plan_make(ctx, D, sides, pl);
    if (rc) return rc;
    step_finish_keep_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, out12, pl.side[0].counters, pl.side[1].counters);
    CDR_LAUNCH_CHECK();
    return apply_dups_pair(ctx, s, opt, D, pl);
}

// ---- round 5: the fused single-occurrence update inside the two multi-GPU layouts -----------------------------------
// DIMENSION shard (cdr_dimshard.hip): the step is cut in two around the all-reduce of the partial scores.
//   cdr_bpr_step_presort    needs only the ids -> runs under the all-reduce: two-table sort + occurrence flags + duplicate-segment heads
//   cdr_bpr_step_from_diff  behind the all-reduce: EmbLoss coefficients from the all-reduced norms (diff[B], diff[B+1]), then ONE pass that
//                           re-gathers the column slices, forms g_t from the GIVEN score diff[t], updates every row that occurs once in
//                           place and writes gradient rows only for duplicate rows; the duplicate segments are applied as in the one-GPU step.
extern "C" int cdr_bpr_step_presort(cdr_ctx* ctx, void* stream, const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B,
                                    int64_t user_rows, int64_t item_rows, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                                    void* sort_ws, size_t sort_ws_bytes, uint32_t* key_base_out) {
    CDR_CHECK_ARG(ctx && uid && pid && nid && keys && perm && flags && heads && sort_ws && key_base_out && B > 0 && 3 * B <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(((uintptr_t)flags & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    int rc = cdr_sort_ids_two_tables(ctx, stream, uid, B, user_rows, pid, B, nid, B, item_rows, keys, perm, key_base_out, sort_ws, sort_ws_bytes);
    if (rc) return rc;
    CDR_HIP(cdr_zero_u32(heads, 4, s));                     // the two head lists' counters
    const int fgrid = grid_for(3 * B, kBlock * kFlagIT);
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(fgrid), dim3(kBlock), 0, s>>>(keys, perm, B, 3 * B, 4, flags, heads + 4, heads + 4 + (B / 2 + 1), (unsigned*)heads);
    }
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bpr_step_from_diff(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, float* item_tab,
                                      float* item_m, float* item_v, int Ds, const int64_t* uid, const int64_t* pid, const int64_t* nid,
                                      int64_t B, float gamma, float reg_weight, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, int64_t step_user, int64_t step_item, const float* diff, uint32_t key_base,
                                      float* out9, float* GU, float* GP, const uint32_t* keys, const uint32_t* perm, const uint8_t* flags,
                                      uint32_t* heads) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && diff && out9 && GU && GP && keys && perm && flags && heads);
    CDR_CHECK_ARG(Ds > 0 && (Ds & 3) == 0 && Ds <= 256 && B > 0 && 3 * B <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && item_m && item_v && step_user > 0 && step_item > 0));
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(Ds);
    coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, 0, B, reg_weight, out9, 1, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr, diff + B);
    CDR_LAUNCH_CHECK();
    const apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    const apply_hp hi = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_item);
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{item_tab, item_m, item_v};
    const int grid = grid_for(B, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_APPLY, s);
#define FA_ARGS TU, TI, Ds, uid, pid, nid, (const uint32_t*)flags, B, gamma, 1.0f / (float)B, out9 + 4, hu, hi, GU, GP, ctx->partials, diff
        if (opt == 0) { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 0, 1, true><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
        else { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 1, 1, true><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
#undef FA_ARGS
    }
    CDR_LAUNCH_CHECK();
    unsigned* cnt = (unsigned*)heads;
    const dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, B, heads + 4, cnt, GU, B, B, out9 + 4, hu, 0},
                               {item_tab, item_m, item_v, keys + B, perm + B, 2 * B, heads + 4 + (B / 2 + 1), cnt + 1, GP, B, B, out9 + 5, hi, key_base}};
    dups_plan pl;
    int rc = dups_plan_make(ctx, Ds, sides, pl);
    if (rc) return rc;
    step_finish_keep_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, out9, pl.side[0].counters, pl.side[1].counters, diff + B);
    CDR_LAUNCH_CHECK();
    return apply_dups_pair(ctx, s, opt, Ds, pl);
}

// The pointwise rows of the dimension layout (cdr_point_partial_dot -> all-reduce -> here), same cut as the BPR pair above:
//   cdr_point_step_presort   ids only (under the all-reduce): two-table sort of uid / iid + occurrence flags + duplicate-segment heads
//   cdr_point_step_from_dot  dot [B + 2] = all-reduced {<u, i> ..., sum u^2, sum i^2}
extern "C" int cdr_point_step_presort(cdr_ctx* ctx, void* stream, const int64_t* uid, const int64_t* iid, int64_t B, int64_t user_rows,
                                      int64_t item_rows, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads, void* sort_ws,
                                      size_t sort_ws_bytes, uint32_t* key_base_out) {
    CDR_CHECK_ARG(ctx && uid && iid && keys && perm && flags && heads && sort_ws && key_base_out && B > 0 && 2 * B <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(((uintptr_t)flags & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    int rc = cdr_sort_ids_two_tables(ctx, stream, uid, B, user_rows, iid, B, nullptr, 0, item_rows, keys, perm, key_base_out, sort_ws, sort_ws_bytes);
    if (rc) return rc;
    CDR_HIP(cdr_zero_u32(heads, 4, s));
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(grid_for(2 * B, kBlock * kFlagIT)), dim3(kBlock), 0, s>>>(keys, perm, B, 2 * B, 4, flags, heads + 4, heads + 4 + (B / 2 + 1), (unsigned*)heads);
    }
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_point_step_from_dot(cdr_ctx* ctx, void* stream, int loss_kind, int opt, float* user_tab, float* user_m, float* user_v,
                                       float* item_tab, float* item_m, float* item_v, int Ds, const int64_t* uid, const int64_t* iid,
                                       const float* label, int64_t B, float reg_weight, float lr, float beta1, float beta2, float eps,
                                       float weight_decay, int64_t step_user, int64_t step_item, const float* dot, uint32_t key_base,
                                       float* out9, float* GU, float* GI, const uint32_t* keys, const uint32_t* perm, const uint8_t* flags,
                                       uint32_t* heads) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && iid && label && dot && out9 && GU && GI && keys && perm && flags && heads);
    CDR_CHECK_ARG((loss_kind == CDR_LOSS_MSE || loss_kind == CDR_LOSS_BCE) && Ds > 0 && (Ds & 3) == 0 && Ds <= 256 && B > 0);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && item_m && item_v && step_user > 0 && step_item > 0));
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(Ds);
    coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, 0, B, reg_weight, out9, 1, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr, dot + B);
    CDR_LAUNCH_CHECK();
    const apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    const apply_hp hi = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_item);
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{item_tab, item_m, item_v};
    const int grid = grid_for(B, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_POINT_FWD_GRAD, s);
#define PA_ARGS loss_kind, TU, TI, Ds, uid, iid, label, (const uint32_t*)flags, B, 1.0f / (float)B, out9 + 4, hu, hi, GU, GI, ctx->partials, dot
        if (opt == 0) { DISPATCH_LPR(lpr, point_fwd_apply_kernel<L, 0, true><<<dim3(grid), dim3(kBlock), 0, s>>>(PA_ARGS)); }
        else { DISPATCH_LPR(lpr, point_fwd_apply_kernel<L, 1, true><<<dim3(grid), dim3(kBlock), 0, s>>>(PA_ARGS)); }
#undef PA_ARGS
    }
    CDR_LAUNCH_CHECK();
    unsigned* cnt = (unsigned*)heads;
    const dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, B, heads + 4, cnt, GU, B, B, out9 + 4, hu, 0},
                               {item_tab, item_m, item_v, keys + B, perm + B, B, heads + 4 + (B / 2 + 1), cnt + 1, GI, B, B, out9 + 5, hi, key_base}};
    dups_plan pl;
    int rc = dups_plan_make(ctx, Ds, sides, pl);
    if (rc) return rc;
    step_finish_keep_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, out9, pl.side[0].counters, pl.side[1].counters, dot + B);
    CDR_LAUNCH_CHECK();
    return apply_dups_pair(ctx, s, opt, Ds, pl);
}

// ROW shard (shard.ShardedBPRStep): after user-aligned routing the user rows of a rank's triples are its own, the item rows arrive in a
// compact buffer (`irows`, one row per distinct item, indexed by ip / in).
//   cdr_batch_norm_sums       sums3 = {0, sum ||U[u]||^2, sum ||irows[ip]||^2} of the rank's triples -> all-reduced by the caller (the EmbLoss
//                             coefficients need the GLOBAL norms before any row is updated) -> cdr_loss_finish_sums leaves them in out9[4..5]
//   cdr_bpr_shard_local_step  sort of the local user rows + flags, then the one-GPU forward-and-update kernel with the item "table" = irows
//                             and no item row ever flagged: user rows that occur once are updated in place, duplicate user rows go through
//                             GU and the segmented apply, GP[t] = g_t u_t for EVERY triple (summed per distinct item and sent home by the
//                             caller: cdr_segsum_rows).  out9[6..8] = this rank's {loss sum, sum u^2, sum p^2}; out9[4..5] untouched.
extern "C" int cdr_batch_norm_sums(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_rows, int D, const int64_t* uid,
                                   const int64_t* pid, int64_t B, float* sums3) {
    CDR_CHECK_ARG(ctx && user_tab && item_rows && uid && pid && sums3 && D > 0 && (D & 3) == 0 && D <= 256 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const int ngrid = grid_for((B + 7) / 8, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BATCH_NORMS, s);
        DISPATCH_LPR(lpr, batch_norms_kernel<L><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, item_rows, D, uid, pid, B, ctx->partials));
    }
    CDR_LAUNCH_CHECK();
    norm_sums_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, ngrid, sums3);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bpr_shard_local_step(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                                        const float* irows, int D, const int64_t* u_loc, const int64_t* ip, const int64_t* in, int64_t Bl,
                                        int64_t B_global, float gamma, float reg_weight, float lr, float beta1, float beta2, float eps,
                                        float weight_decay, int64_t step_user, float* out9, float* GU, float* GP, uint32_t* keys,
                                        uint32_t* perm, uint8_t* flags, uint32_t* heads, void* sort_ws, size_t sort_ws_bytes) {
    CDR_CHECK_ARG(ctx && user_tab && irows && u_loc && ip && in && out9 && GU && GP && keys && perm && flags && heads && sort_ws);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && Bl > 0 && Bl <= (int64_t)0x7FFFFFFF && B_global >= Bl);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && step_user > 0));
    CDR_CHECK_ARG(((uintptr_t)flags & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    int rc = cdr_sort_ids(ctx, stream, u_loc, Bl, nullptr, 0, user_rows, keys, perm, sort_ws, sort_ws_bytes);
    if (rc) return rc;
    CDR_HIP(cdr_zero_u32(heads, 4, s));
    unsigned* cnt = (unsigned*)heads;
    uint32_t* headsA = heads + 4;
    uint32_t* headsB = headsA + (Bl / 2 + 1);                       // stays empty: no item row is ever updated here
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(grid_for(Bl, kBlock * kFlagIT)), dim3(kBlock), 0, s>>>(keys, perm, Bl, Bl, 4, flags, headsA, headsB, cnt);
    }
    CDR_LAUNCH_CHECK();
    const apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    const apply_hp hi = make_hp(0, lr, beta1, beta2, eps, weight_decay, 1);                // (never used: the p / n flag bytes stay 0)
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{const_cast<float*>(irows), nullptr, nullptr};
    const int grid = grid_for(Bl, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_APPLY, s);
#define FA_ARGS TU, TI, D, u_loc, ip, in, (const uint32_t*)flags, Bl, gamma, 1.0f / (float)B_global, out9 + 4, hu, hi, GU, GP, ctx->partials
        if (opt == 0) { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 0, 1><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
        else { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 1, 1><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
#undef FA_ARGS
    }
    CDR_LAUNCH_CHECK();
    const dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, Bl, headsA, cnt, GU, Bl, Bl, out9 + 4, hu, 0},
                               {const_cast<float*>(irows), nullptr, nullptr, keys, perm, 0, headsB, cnt + 1, GP, 0, 0, out9 + 5, hi, 0}};
    dups_plan pl;
    rc = dups_plan_make(ctx, D, sides, pl);
    if (rc) return rc;
    shard_sums_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, out9, pl.side[0].counters);
    CDR_LAUNCH_CHECK();
    return apply_dups_pair(ctx, s, opt, D, pl);
}

// ---- round 6: the row-sharded step on its own passes --------------------------------------------------------------
//   cdr_bpr_shard_plan   ONE sort for both lists of a rank's routed triples (user rows | item keys grouped by owner) -> occurrence flags of
//                        both (user occurs once -> updated in place; item occurs once -> its gradient row goes straight into its send
//                        slot), the duplicate segments' heads, and the request list: distinct item rows per owner, occurrence -> slot map.
//   cdr_shard_norm_sums  EmbLoss norm sums from the local user rows and the owners' per-row squared norms.
//   cdr_bpr_shard_step   the forward-and-update pass (SH): user rows as in the one-GPU step; gradient rows of single item occurrences written
//                        once, into GS; GP only where a duplicate needs it; then ONE launch for the duplicate user rows (update) and the
//                        duplicate item segments (sum -> GS).
namespace {
struct shard_bits { unsigned lb, hb; uint32_t key_base; };
static int shard_key_bits(int64_t user_rows, int64_t item_local_rows, int world, shard_bits& kb) {
    if (user_rows <= 0 || item_local_rows <= 0 || world < 1 || world > 1024) return CDR_EINVAL;
    kb.lb = bits_for(item_local_rows);
    const unsigned ob = world > 1 ? bits_for(world) : 0u;
    const unsigned ub = bits_for(user_rows);
    kb.hb = ub > kb.lb + ob ? ub : kb.lb + ob;
    if (kb.hb >= 31) return CDR_EINVAL;
    kb.key_base = 1u << kb.hb;
    return CDR_OK;
}
}  // namespace

extern "C" int cdr_bpr_shard_plan_sizes(int64_t Bl, int64_t user_rows, int64_t item_local_rows, int world, int64_t* heads_words,
                                        size_t* ws_bytes) {
    CDR_CHECK_ARG(heads_words && ws_bytes && Bl > 0 && 3 * Bl <= (int64_t)0x7FFFFFFF);
    shard_bits kb;
    CDR_CHECK_ARG(shard_key_bits(user_rows, item_local_rows, world, kb) == CDR_OK);
    *heads_words = 4 + (Bl / 2 + 1) + (Bl + 1);
    size_t sort_need = 0;
    int rc = cdr_sort_workspace_bytes(3 * Bl, (int64_t)kb.key_base * 2, &sort_need);
    if (rc) return rc;
    size_t tmp = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tmp, rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), head_flag_op{nullptr}),
                                           (uint32_t*)nullptr, 0u, (size_t)(2 * Bl), rocprim::plus<uint32_t>());
    if (e != hipSuccess) { cdr_set_error("cdr_bpr_shard_plan_sizes: %s", hipGetErrorString(e)); return (int)e; }
    const size_t scan_need = (((size_t)(2 * Bl) * sizeof(uint32_t) + 255) & ~(size_t)255) + ((tmp + 255) & ~(size_t)255);
    *ws_bytes = sort_need > scan_need ? sort_need : scan_need;
    return CDR_OK;
}

extern "C" int cdr_bpr_shard_plan(cdr_ctx* ctx, void* stream, const int64_t* recv3, int64_t Bl, int64_t user_rows, int64_t item_local_rows,
                                  int world, int64_t* u_loc, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                                  uint32_t* uidx, int64_t* uniq_local, int64_t* umap, int64_t* counts, int64_t* n_uniq, void* ws,
                                  size_t ws_bytes) {
    CDR_CHECK_ARG(ctx && recv3 && u_loc && keys && perm && flags && heads && uidx && uniq_local && umap && counts && n_uniq && ws);
    CDR_CHECK_ARG(Bl > 0 && 3 * Bl <= (int64_t)0x7FFFFFFF && ((uintptr_t)flags & 3) == 0);
    shard_bits kb;
    CDR_CHECK_ARG(shard_key_bits(user_rows, item_local_rows, world, kb) == CDR_OK);
    int64_t hw = 0; size_t need = 0;
    int rc = cdr_bpr_shard_plan_sizes(Bl, user_rows, item_local_rows, world, &hw, &need);
    if (rc) return rc;
    CDR_CHECK_ARG(ws_bytes >= need);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = 3 * Bl;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t* keys_in = (uint32_t*)ws;
    uint32_t* vals_in = (uint32_t*)((char*)ws + arr);
    void* tmp = (char*)ws + 2 * arr;
    size_t tmp_bytes = ws_bytes - 2 * arr;
    unsigned* cnt = (unsigned*)heads;
    {
        cdr_time_scope ts(ctx, CDR_TAG_SORT, s);
        shard_keys_kernel<<<dim3(grid_for(Bl, kBlock)), dim3(kBlock), 0, s>>>(recv3, Bl, (uint32_t)world, kb.lb, kb.key_base, u_loc, keys_in, vals_in, cnt);
        CDR_LAUNCH_CHECK();
        CDR_HIP(sort_pairs(tmp, tmp_bytes, keys_in, keys, vals_in, perm, (size_t)n, kb.hb + 1, s));
    }
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(grid_for(n, kBlock * kFlagIT)), dim3(kBlock), 0, s>>>(keys, perm, Bl, n, 4, flags, heads + 4, heads + 4 + (Bl / 2 + 1), cnt);
    }
    CDR_LAUNCH_CHECK();
    // the request list from section B (the sort's inputs are dead: the scan reuses the workspace)
    const int64_t nB = 2 * Bl;
    const uint32_t* keysB = keys + Bl;
    const size_t arrB = ((size_t)nB * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t* excl = (uint32_t*)ws;
    void* stmp = (char*)ws + arrB;
    size_t stmp_bytes = ws_bytes - arrB;
    CDR_HIP(rocprim::exclusive_scan(stmp, stmp_bytes, rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), head_flag_op{keysB}),
                                    excl, 0u, (size_t)nB, rocprim::plus<uint32_t>(), s));
    shard_emit_kernel<<<dim3(grid_for(nB, kBlock)), dim3(kBlock), 0, s>>>(keysB, perm + Bl, excl, nB, world, kb.lb, kb.key_base, uidx, uniq_local, umap,
                                                                          counts, n_uniq);
    CDR_LAUNCH_CHECK();
    shard_counts_kernel<<<dim3(1), dim3(64), 0, s>>>(counts, world);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_shard_norm_sums(cdr_ctx* ctx, void* stream, const float* user_tab, int D, const int64_t* u_loc, const float* nrm2,
                                   const int64_t* ip, int64_t Bl, float* sums3) {
    CDR_CHECK_ARG(ctx && user_tab && u_loc && nrm2 && ip && sums3 && D > 0 && (D & 3) == 0 && D <= 256 && Bl > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const int ngrid = grid_for((Bl + 7) / 8, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BATCH_NORMS, s);
        DISPATCH_LPR(lpr, shard_norms_kernel<L><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, D, u_loc, nrm2, ip, Bl, ctx->partials));
    }
    CDR_LAUNCH_CHECK();
    norm_sums_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, ngrid, sums3);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_bpr_shard_step(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, const float* irows, int D,
                                  const int64_t* u_loc, const int64_t* umap, int64_t Bl, int64_t B_global, float gamma, float reg_weight,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step_user, float* out9, float* GU,
                                  float* GP, float* GS, const uint32_t* keys, const uint32_t* perm, const uint8_t* flags, uint32_t* heads,
                                  const uint32_t* uidx) {
    CDR_CHECK_ARG(ctx && user_tab && irows && u_loc && umap && out9 && GU && GP && GS && keys && perm && flags && heads && uidx);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && Bl > 0 && 3 * Bl <= (int64_t)0x7FFFFFFF && B_global >= Bl);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && step_user > 0));
    CDR_CHECK_ARG(((uintptr_t)flags & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    const apply_hp hi = make_hp(0, lr, beta1, beta2, eps, weight_decay, 1);                // (no item row is updated on this side)
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{const_cast<float*>(irows), nullptr, nullptr};
    const int grid = grid_for(Bl, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_APPLY, s);
#define FA_ARGS TU, TI, D, u_loc, umap, umap + Bl, (const uint32_t*)flags, Bl, gamma, 1.0f / (float)B_global, out9 + 4, hu, hi, GU, GP, ctx->partials, nullptr, GS
        if (opt == 0) { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 0, 1, false, true><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
        else { DISPATCH_LPR(lpr, bpr_fwd_apply_kernel<L, 1, 1, false, true><<<dim3(grid), dim3(kBlock), 0, s>>>(FA_ARGS)); }
#undef FA_ARGS
    }
    CDR_LAUNCH_CHECK();
    unsigned* cnt = (unsigned*)heads;
    dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, Bl, heads + 4, cnt, GU, Bl, Bl, out9 + 4, hu, 0},
                         {const_cast<float*>(irows), nullptr, nullptr, keys + Bl, perm + Bl, 2 * Bl, heads + 4 + (Bl / 2 + 1), cnt + 1, GP, Bl, Bl,
                          out9 + 5, hi, 0}};
    sides[1].out = GS; sides[1].uidx = uidx;
    dups_plan pl;
    int rc = dups_plan_make(ctx, D, sides, pl);
    if (rc) return rc;
    shard_sums2_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, out9, pl.side[0].counters, pl.side[1].counters);
    CDR_LAUNCH_CHECK();
    const int dgrid = grid_for((2 * Bl) / 16 + 1, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_APPLY_SIGNED, s);
        if (opt == 0) { DISPATCH_LPR(lpr, shard_dups_kernel<L, 0><<<dim3(dgrid, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
        else { DISPATCH_LPR(lpr, shard_dups_kernel<L, 1><<<dim3(dgrid, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
    }
    CDR_LAUNCH_CHECK();
    if (pl.long_cap[0] || pl.long_cap[1]) {
        const int64_t pc = pl.piece_cap[0] > pl.piece_cap[1] ? pl.piece_cap[0] : pl.piece_cap[1];
        const int64_t lc = pl.long_cap[0] > pl.long_cap[1] ? pl.long_cap[0] : pl.long_cap[1];
        const int gp = (int)(pc < 2048 ? pc : 2048);                        // one workgroup per piece (looping past 2,048)
        DISPATCH_LPR(lpr, seg_piece_sum2_kernel<L><<<dim3(gp, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1]));
        CDR_LAUNCH_CHECK();
        const int gl = grid_for(lc < 4096 ? lc : 4096, kBlock / lpr);
        if (opt == 0) { DISPATCH_LPR(lpr, shard_long_finish_kernel<L, 0><<<dim3(gl, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
        else { DISPATCH_LPR(lpr, shard_long_finish_kernel<L, 1><<<dim3(gl, 2), dim3(kBlock), 0, s>>>(D, pl.side[0], pl.side[1])); }
        CDR_LAUNCH_CHECK();
    }
    return CDR_OK;
}

// The owner's side: ids = `runs` ascending duplicate-free runs of local rows (one per requesting rank, in rank order), grads one summed
// gradient row per id (EmbLoss term inside).  One run needs no sort: the list IS the sorted segment list; several runs are radix-sorted
// as everywhere (stable: rows asked for by several ranks are summed in rank order).
extern "C" int cdr_shard_owner_apply(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int64_t table_rows,
                                     int D, const int64_t* ids, int64_t n, int runs, const float* grads, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, int64_t step, uint32_t* keys, uint32_t* perm, void* ws, size_t ws_bytes) {
    CDR_CHECK_ARG(ctx && table && ids && grads && keys && perm && n > 0 && runs >= 1 && table_rows > 0);
    hipStream_t s = (hipStream_t)stream;
    if (runs == 1 && D <= 256 && getenv("CDR_OWNER_RUN_GENERIC") == nullptr) {
        CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && (opt == 0 || (opt == 1 && exp_avg && exp_avg_sq && step > 0)));
        const apply_hp hp = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step);
        const int lpr = cdr_lpr_for(D);
        cdr_time_scope ts(ctx, CDR_TAG_APPLY_UNSIGNED, s);
        if (opt == 0) { DISPATCH_LPR(lpr, apply_run_kernel<L, 0><<<dim3(grid_for((n + 1) / 2, kBlock / lpr)), dim3(kBlock), 0, s>>>(table, exp_avg, exp_avg_sq, D, ids, n, grads, hp)); }
        else { DISPATCH_LPR(lpr, apply_run_kernel<L, 1><<<dim3(grid_for((n + 1) / 2, kBlock / lpr)), dim3(kBlock), 0, s>>>(table, exp_avg, exp_avg_sq, D, ids, n, grads, hp)); }
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    if (runs == 1) {
        sorted_run_keys_kernel<<<dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s>>>(ids, n, keys, perm);
        CDR_LAUNCH_CHECK();
    } else {
        int rc = cdr_sort_ids(ctx, stream, ids, n, nullptr, 0, table_rows, keys, perm, ws, ws_bytes);
        if (rc) return rc;
    }
    return cdr_rowwise_apply(ctx, stream, opt, table, exp_avg, exp_avg_sq, D, keys, perm, n, grads, n, 0, nullptr, lr, beta1, beta2, eps,
                             weight_decay, step, nullptr, 0);
}

// ---- the per-positive (k-major) form: uid / pid [S] (the first S entries of recbole's tiled [S k] columns), nid [S k] k-major
extern "C" int cdr_bpr_step_fused_kmajor_sizes(int64_t S, int k, int64_t* flag_bytes, int64_t* heads_words) {
    CDR_CHECK_ARG(flag_bytes && heads_words && S > 0 && k >= 1 && k <= 64);
    const int fstride = (2 + k + 3) & ~3;
    *flag_bytes = S * (int64_t)fstride;
    *heads_words = 4 + (S / 2 + 1) + ((S + S * (int64_t)k) / 2 + 1);
    return CDR_OK;
}

extern "C" int cdr_bpr_step_fused_kmajor(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v,
                                         int64_t user_rows, float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D,
                                         const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t S, int k, float gamma,
                                         float reg_weight, float lr, float beta1, float beta2, float eps, float weight_decay,
                                         int64_t step_user, int64_t step_item, float* out9, float* GU, float* GI, uint32_t* keys,
                                         uint32_t* perm, uint8_t* flags, uint32_t* heads, void* sort_ws, size_t sort_ws_bytes) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out9 && GU && GI && keys && perm && flags && heads && sort_ws);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && S > 0 && k >= 1 && k <= 64);
    const int64_t B = S * (int64_t)k, nI = S + B;
    CDR_CHECK_ARG(S + nI <= (int64_t)0x7FFFFFFF);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && item_m && item_v && step_user > 0 && step_item > 0));
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const int fstride = (2 + k + 3) & ~3;
    if (reg_weight != 0.f) {
        const int ngrid = grid_for((S + 7) / 8, kBlock / lpr);
        {
            cdr_time_scope ts(ctx, CDR_TAG_BATCH_NORMS, s);
            DISPATCH_LPR(lpr, batch_norms_kernel<L><<<dim3(ngrid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, S, ctx->partials));
        }
        CDR_LAUNCH_CHECK();
        coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, ngrid, B, reg_weight, out9, k, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, (unsigned*)heads);
    } else {
        coef_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, 0, B, 0.f, out9, k, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, (unsigned*)heads);
    }
    CDR_LAUNCH_CHECK();
    uint32_t key_base = 0;
    int rc = cdr_sort_ids_two_tables(ctx, stream, uid, S, user_rows, pid, S, nid, B, item_rows, keys, perm, &key_base, sort_ws, sort_ws_bytes);
    if (rc) return rc;
    unsigned* cnt = (unsigned*)heads;
    uint32_t* headsA = heads + 4;
    uint32_t* headsB = headsA + (S / 2 + 1);           // (cnt[0..3] were cleared by coef_finish_kernel)
    const int fgrid = grid_for(S + nI, kBlock * kFlagIT);
    {
        cdr_time_scope ts(ctx, CDR_TAG_OCC_FLAGS, s);
        occ_flags_kernel<<<dim3(fgrid), dim3(kBlock), 0, s>>>(keys, perm, S, S + nI, fstride, flags, headsA, headsB, cnt);
    }
    CDR_LAUNCH_CHECK();
    const apply_hp hu = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_user);
    const apply_hp hi = make_hp(opt, lr, beta1, beta2, eps, weight_decay, step_item);
    const tab_ptrs TU{user_tab, user_m, user_v}, TI{item_tab, item_m, item_v};
    const int grid = grid_for(S, kBlock / lpr);
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_APPLY, s);
#define FK_ARGS TU, TI, D, uid, pid, nid, flags, fstride, S, k, gamma, 1.0f / (float)B, out9 + 4, hu, hi, GU, GI, ctx->partials
        if (opt == 0) { DISPATCH_LPR(lpr, bpr_fwd_apply_kmajor_kernel<L, 0, 4><<<dim3(grid), dim3(kBlock), 0, s>>>(FK_ARGS)); }
        else if (k <= 2) { DISPATCH_LPR(lpr, bpr_fwd_apply_kmajor_kernel<L, 1, 2><<<dim3(grid), dim3(kBlock), 0, s>>>(FK_ARGS)); }
        else { DISPATCH_LPR(lpr, bpr_fwd_apply_kmajor_kernel<L, 1, 4><<<dim3(grid), dim3(kBlock), 0, s>>>(FK_ARGS)); }
#undef FK_ARGS
    }
    CDR_LAUNCH_CHECK();
    // duplicate rows: users over GU [S, D]; items over GI [S + B, D] (one row per occurrence of [pid | nid], signs folded in)
    const dup_host sides[2] = {{user_tab, user_m, user_v, keys, perm, S, headsA, cnt, GU, S, S, out9 + 4, hu, 0},
                               {item_tab, item_m, item_v, keys + S, perm + S, nI, headsB, cnt + 1, GI, nI, S, out9 + 5, hi, key_base}};
    dups_plan pl;
    rc = dups_plan_make(ctx, D, sides, pl);
    if (rc) return rc;
    step_finish_keep_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, reg_weight, out9, pl.side[0].counters, pl.side[1].counters);
    CDR_LAUNCH_CHECK();
    return apply_dups_pair(ctx, s, opt, D, pl);
}
