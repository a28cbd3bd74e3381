// Per-positive ("k-major") fused BPR step: the O(batch) training step of cdr_step.hip re-cut along the reference's own
// batch layout.  recbole's pairwise loader repeats every positive k times and lays the negatives out k-major
// (crossdomain_sampler.py:148-152: nid[j + m*S] is the m-th negative of positive j; uid / pid are S rows tiled k times;
// call sites emcdr.py:123-131,146-154).  Treating the B = S*k triples as independent re-reads the user row and the positive
// row k times and writes k gradient rows per positive.  Here
//
//   bpr_fwd_kmajor_kernel   one lane group per POSITIVE: u and p once, the k negatives, <u,p> once; writes ONE compact
//                           user-gradient row  GU[j] = sum_m g_m (p - n_m)  and, per item occurrence, an 8-byte record
//                           {user row, coefficient}: rec[j] = {u_j, sum_m g_m}, rec[S + j + m*S] = {u_j, -g_m} -- every
//                           item gradient row is coefficient * U[u_j], so none of them is written.
//   apply2_kernel<SRC=1>    item table: per distinct row, sum_occ coef * U[user row] rebuilt from the (pre-step) user table
//                           -- the 512-B read that used to fetch the stored gradient row fetches the user row instead --
//                           then EmbLoss term + SGD / Adam in place.  Runs BEFORE the user table's apply.
//   apply2_kernel<SRC=0>    user table: S gradient rows instead of B.
//
// Row traffic per positive at k = 4 (512-B rows): forward 6 reads + 1 write (was 12 + 8), user apply 1 + 6 (was 4 + 6).
// The Adam bias correction can come from a device counter (step_dev) so that the whole step -- forward, ONE-launch LDS
// sort (cdr_smallsort.hip), two applies -- is hipGraph-capturable; `small` drops the long-segment machinery (its memset and
// two extra launches) for batches whose worst-case segment a single lane group can walk.
#include "cdr_common.h"
#include "cdr_adam_math.h"
#include "cdr_ranksort.h"

namespace {

constexpr int kBlock = 256;

struct item_rec { int32_t urow; float coef; };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

inline int grid_for(int64_t units, int per_block) {
    int64_t g = (units + per_block - 1) / per_block;
    const int64_t cap = CDR_NUM_CU * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------ forward + compact grads
// KC negatives per positive are in flight together (k is walked in chunks of KC), UNR positives per lane group.
template <int LPR, int KC, int UNR>
__device__ __forceinline__ void bpr_fwd_kmajor_body(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                    const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                    const int64_t* __restrict__ nid, int64_t S, int k, float gamma, float invB,
                                                    float* __restrict__ GU, item_rec* __restrict__ rec, double* __restrict__ partials,
                                                    int bid, int nblk, double* smem /* [3 * kBlock / 64] */) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)bid * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)nblk * GPB;
    const int D4 = D >> 2;
    const bool live = sub < D4;
    double acc[3] = {0.0, 0.0, 0.0};
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = gg; base < S; base += TG * UNR) {
        int64_t jc[UNR], iu[UNR], ip[UNR];
        bool ok[UNR];
#pragma unroll
        for (int r = 0; r < UNR; ++r) {                       // ids first, then every row load (vmcnt is an in-order counter)
            const int64_t j = base + (int64_t)r * TG;
            ok[r] = j < S;
            jc[r] = ok[r] ? j : S - 1;
            iu[r] = uid[jc[r]]; ip[r] = pid[jc[r]];
        }
        float4 u[UNR], p[UNR], gu[UNR];
        float dp[UNR], gs[UNR];
#pragma unroll
        for (int r = 0; r < UNR; ++r) { gu[r] = z4; gs[r] = 0.f; dp[r] = 0.f; u[r] = p[r] = z4; }
        for (int m0 = 0; m0 < k; m0 += KC) {
            int64_t in[UNR][KC];
            float4 n[UNR][KC];
#pragma unroll
            for (int r = 0; r < UNR; ++r)
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const int m = m0 + c < k ? m0 + c : k - 1;
                    in[r][c] = nid[jc[r] + (int64_t)m * S];
                }
            if (m0 == 0) {
#pragma unroll
                for (int r = 0; r < UNR; ++r)
                    if (live) { u[r] = ld4(U + iu[r] * D + 4 * sub); p[r] = ld4(I + ip[r] * D + 4 * sub); }
            }
#pragma unroll
            for (int r = 0; r < UNR; ++r)
#pragma unroll
                for (int c = 0; c < KC; ++c) n[r][c] = live ? ld4(I + in[r][c] * D + 4 * sub) : z4;
            if (m0 == 0) {
#pragma unroll
                for (int r = 0; r < UNR; ++r) {
                    dp[r] = group_sum<LPR>(dot4(u[r], p[r]));
                    const float su = group_sum<LPR>(dot4(u[r], u[r]));
                    const float sp = group_sum<LPR>(dot4(p[r], p[r]));
                    if (ok[r] && sub == 0) { acc[1] += (double)k * (double)su; acc[2] += (double)k * (double)sp; }   // EmbLoss sees k copies
                }
            }
#pragma unroll
            for (int r = 0; r < UNR; ++r)
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const float dn = group_sum<LPR>(dot4(u[r], n[r][c]));
                    if (m0 + c < k && ok[r]) {
                        const float s = sigmoidf_(dp[r] - dn);
                        const float g = -invB * (s * (1.0f - s)) / (gamma + s);
                        gu[r].x += g * (p[r].x - n[r][c].x); gu[r].y += g * (p[r].y - n[r][c].y);
                        gu[r].z += g * (p[r].z - n[r][c].z); gu[r].w += g * (p[r].w - n[r][c].w);
                        gs[r] += g;
                        if (sub == 0) {
                            rec[S + jc[r] + (int64_t)(m0 + c) * S] = item_rec{(int32_t)iu[r], -g};
                            acc[0] += (double)(-logf(gamma + s));
                        }
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            if (!ok[r]) continue;
            if (live) st4(GU + jc[r] * D + 4 * sub, gu[r]);
            if (sub == 0) rec[jc[r]] = item_rec{(int32_t)iu[r], gs[r]};
        }
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)bid * CDR_PARTIAL_STRIDE;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

template <int LPR, int KC, int UNR>
__global__ __launch_bounds__(kBlock) void bpr_fwd_kmajor_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                                const int64_t* __restrict__ nid, int64_t S, int k, float gamma,
                                                                float invB, float* __restrict__ GU, item_rec* __restrict__ rec,
                                                                double* __restrict__ partials) {
    __shared__ double smem[3 * (kBlock / 64)];
    bpr_fwd_kmajor_body<LPR, KC, UNR>(U, I, D, uid, pid, nid, S, k, gamma, invB, GU, rec, partials, blockIdx.x, gridDim.x, smem);
}

// Small batches, launch 1 of 4: the forward blocks and the rank-count blocks of the id sort side by side (the sort needs only
// the ids).  Block 0 also advances the two tables' device update counters: nothing in this launch reads them.
template <int LPR, int KC, int UNR>
__global__ __launch_bounds__(kBlock) void small_fwd_count_kernel(const float* __restrict__ U, const float* __restrict__ I, int D,
                                                                 const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                                 const int64_t* __restrict__ nid, int64_t S, int k, float gamma,
                                                                 float invB, float* __restrict__ GU, item_rec* __restrict__ rec,
                                                                 double* __restrict__ partials, int fwd_blocks,
                                                                 ranksort::small_sort_args sa, uint32_t* __restrict__ rank,
                                                                 int64_t* bump_a, int64_t* bump_b) {
    __shared__ double smem[3 * (kBlock / 64)];
    __shared__ __attribute__((aligned(16))) uint32_t sh[512];
    if ((int)blockIdx.x < fwd_blocks) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (bump_a) bump_a[0] += 1;
            if (bump_b) bump_b[0] += 1;
        }
        bpr_fwd_kmajor_body<LPR, KC, UNR>(U, I, D, uid, pid, nid, S, k, gamma, invB, GU, rec, partials, blockIdx.x, fwd_blocks, smem);
    } else {
        ranksort::rank_count_body(sa, rank, (int)blockIdx.x - fwd_blocks, sh);
    }
}

// out9 as cdr_bpr_fwd_grad's, with the EmbLoss coefficients pre-multiplied by k (an occurrence in the S-row lists stands for
// k rows of the reference's batch); also advances the two tables' device update counters for the applies that follow.
__device__ __forceinline__ void kstep_finish_body(const double* __restrict__ partials, int nblocks, int64_t B, int k,
                                                  float reg_weight, float* __restrict__ out9, int64_t* bump_a, int64_t* bump_b,
                                                  double* smem) {
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const double* o = partials + (size_t)b * CDR_PARTIAL_STRIDE;
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2];
    }
    block_sum_d<3>(acc, smem);
    if (threadIdx.x == 0) {
        const float main_loss = (float)(acc[0] / (double)B);
        const float nu = (float)sqrt(acc[1]), ni = (float)sqrt(acc[2]);
        out9[1] = main_loss; out9[2] = nu; out9[3] = ni;
        out9[0] = main_loss + reg_weight * ((nu + ni) / (float)B);
        out9[4] = (reg_weight != 0.f && nu > 0.f) ? (float)k * (reg_weight / ((float)B * nu)) : 0.f;
        out9[5] = (reg_weight != 0.f && ni > 0.f) ? (float)k * (reg_weight / ((float)B * ni)) : 0.f;
        out9[6] = (float)acc[0]; out9[7] = (float)acc[1]; out9[8] = (float)acc[2];
        if (bump_a) bump_a[0] += 1;
        if (bump_b) bump_b[0] += 1;
    }
}

__global__ __launch_bounds__(kBlock) void kstep_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t B, int k,
                                                              float reg_weight, float* __restrict__ out9, int64_t* bump_a,
                                                              int64_t* bump_b) {
    __shared__ double smem[3 * (kBlock / 64)];
    kstep_finish_body(partials, nblocks, B, k, reg_weight, out9, bump_a, bump_b, smem);
}

// Small batches, launch 2 of 4: the rank sort's scatter blocks + one block that finishes the loss / EmbLoss coefficients.
__global__ __launch_bounds__(kBlock) void small_scatter_finish_kernel(ranksort::small_sort_args sa, uint32_t* __restrict__ rank,
                                                                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ perm_out,
                                                                      const double* __restrict__ partials, int nblocks, int64_t B, int k,
                                                                      float reg_weight, float* __restrict__ out9) {
    __shared__ double smem[3 * (kBlock / 64)];
    if ((int)blockIdx.x < sa.scatter_blocks) ranksort::rank_scatter_body(sa, rank, keys_out, perm_out, blockIdx.x);
    else kstep_finish_body(partials, nblocks, B, k, reg_weight, out9, nullptr, nullptr, smem);
}

// ------------------------------------------------------------------------------------------------ segmented apply, v2
constexpr int kLongSeg = 32;
constexpr int kPiece = 256;
struct seg_long { int64_t head, len, base; };
struct seg_piece { int64_t start; int64_t len; };
struct apply_hp { float lr, b1, b2, eps, wd, step_size, bc2_sqrt; };

template <int OPT>
__device__ __forceinline__ void apply_update(float* __restrict__ wp, float* __restrict__ mp, float* __restrict__ vp, float4 w,
                                             float4 acc, float rc, const apply_hp& h) {
    float4 gr = make_float4(acc.x + rc * w.x, acc.y + rc * w.y, acc.z + rc * w.z, acc.w + rc * w.w);
    float4 wn;
    if (OPT == 0) {
        if (h.wd != 0.f) { gr.x += h.wd * w.x; gr.y += h.wd * w.y; gr.z += h.wd * w.z; gr.w += h.wd * w.w; }
        wn = make_float4(w.x - h.lr * gr.x, w.y - h.lr * gr.y, w.z - h.lr * gr.z, w.w - h.lr * gr.w);
    } else {
        float4 m = ld4(mp), v = ld4(vp);
        if (h.wd != 0.f) { gr.x += h.wd * w.x; gr.y += h.wd * w.y; gr.z += h.wd * w.z; gr.w += h.wd * w.w; }
        m.x += (gr.x - m.x) * (1.0f - h.b1); m.y += (gr.y - m.y) * (1.0f - h.b1);
        m.z += (gr.z - m.z) * (1.0f - h.b1); m.w += (gr.w - m.w) * (1.0f - h.b1);
        v.x = h.b2 * v.x + (1.0f - h.b2) * gr.x * gr.x; v.y = h.b2 * v.y + (1.0f - h.b2) * gr.y * gr.y;
        v.z = h.b2 * v.z + (1.0f - h.b2) * gr.z * gr.z; v.w = h.b2 * v.w + (1.0f - h.b2) * gr.w * gr.w;
        st4(mp, m); st4(vp, v);
        wn = make_float4(w.x - cdr_adam_term(m.x, v.x, h.step_size, h.bc2_sqrt, h.eps), w.y - cdr_adam_term(m.y, v.y, h.step_size, h.bc2_sqrt, h.eps),
                         w.z - cdr_adam_term(m.z, v.z, h.step_size, h.bc2_sqrt, h.eps), w.w - cdr_adam_term(m.w, v.w, h.step_size, h.bc2_sqrt, h.eps));
    }
    st4(wp, wn);
}

__device__ __forceinline__ void hp_from_device(apply_hp& hp, const int64_t* step_dev) {
    if (step_dev) {                                           // capturable: the update count lives on the device
        const double st = (double)step_dev[0];
        cdr_adam_hp(st, hp.lr, hp.b1, hp.b2, hp.step_size, hp.bc2_sqrt);
    }
}

// gradient row of occurrence o, columns [4 ch, 4 ch + 4):  SRC 0: G[o]   SRC 1: rec[o].coef * Usrc[rec[o].urow]
template <int SRC>
__device__ __forceinline__ float4 occ_grad(const float* __restrict__ G, const item_rec* __restrict__ rec, const float* __restrict__ Usrc,
                                           int64_t o, int D, int ch) {
    if (SRC == 0) return ld4(G + o * D + 4 * ch);
    const item_rec r = rec[o];
    const float4 u = ld4(Usrc + (int64_t)r.urow * D + 4 * ch);
    return make_float4(r.coef * u.x, r.coef * u.y, r.coef * u.z, r.coef * u.w);
}

template <int LPR, int OPT, int SRC>
__global__ __launch_bounds__(kBlock) void apply2_kernel(float* __restrict__ W, float* __restrict__ Mo, float* __restrict__ Vo, int D,
                                                        const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm, int64_t n,
                                                        const float* __restrict__ G, const item_rec* __restrict__ rec,
                                                        const float* __restrict__ Usrc, int64_t reg_limit,
                                                        const float* __restrict__ reg_coef, apply_hp hp,
                                                        const int64_t* __restrict__ step_dev, unsigned* __restrict__ counters,
                                                        seg_long* __restrict__ longs, seg_piece* __restrict__ pieces) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    hp_from_device(hp, step_dev);
    const bool pieces_on = counters != nullptr;
    for (int64_t q = gg; q < n; q += TG) {
        const uint32_t row = keys[q];
        const uint32_t before = keys[q > 0 ? q - 1 : 0];
        const uint32_t far = keys[q + kLongSeg < n ? q + kLongSeg : n - 1];
        const bool head = !(q > 0 && before == row);
        const bool is_long = pieces_on && q + kLongSeg < n && far == row;
        if (head && !is_long) {
            for (int ch = sub; ch < D4; ch += LPR) {
                float* wp = W + (int64_t)row * D + 4 * ch;
                const float4 w = ld4(wp);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                int cnt = 0;
                for (int64_t e = q; e < n && keys[e] == row; ++e) {
                    const int64_t o = perm[e];
                    const float4 g = occ_grad<SRC>(G, rec, Usrc, o, D, ch);
                    acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
                    cnt += (o < reg_limit) ? 1 : 0;
                }
                apply_update<OPT>(wp, OPT ? Mo + (int64_t)row * D + 4 * ch : nullptr, OPT ? Vo + (int64_t)row * D + 4 * ch : nullptr,
                                  w, acc, c * (float)cnt, hp);
            }
        }
    }
    if (!pieces_on) return;
    // registration of the long segments, one THREAD per sorted position (kept out of the loop above: see cdr_step.hip)
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
        const unsigned base = atomicAdd(&counters[0], np);
        const unsigned li = atomicAdd(&counters[1], 1u);
        longs[li] = seg_long{q, len, (int64_t)base};
        for (unsigned kk = 0; kk < np; ++kk) {
            const int64_t st = q + (int64_t)kk * kPiece;
            pieces[base + kk] = seg_piece{st, (hi - st) < kPiece ? (hi - st) : (int64_t)kPiece};
        }
    }
}

template <int LPR, int SRC>
__global__ __launch_bounds__(kBlock) void piece_sum2_kernel(int D, const uint32_t* __restrict__ perm, const float* __restrict__ G,
                                                            const item_rec* __restrict__ rec, const float* __restrict__ Usrc,
                                                            int64_t reg_limit, const unsigned* __restrict__ counters,
                                                            const seg_piece* __restrict__ pieces, float* __restrict__ partial,
                                                            int* __restrict__ pcnt) {
    constexpr int GPB = kBlock / LPR;
    constexpr int UN = 4;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const int64_t np = counters[0];
    for (int64_t pi = gg; pi < np; pi += TG) {
        const seg_piece pc = pieces[pi];
        for (int ch = sub; ch < D4; ch += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnt = 0;
            for (int64_t e0 = 0; e0 < pc.len; e0 += UN) {
                int64_t o[UN]; float4 g[UN];
#pragma unroll
                for (int j = 0; j < UN; ++j) o[j] = (e0 + j < pc.len) ? (int64_t)perm[pc.start + e0 + j] : -1;
#pragma unroll
                for (int j = 0; j < UN; ++j) g[j] = o[j] >= 0 ? occ_grad<SRC>(G, rec, Usrc, o[j], D, ch) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    if (o[j] < 0) continue;
                    acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; acc.w += g[j].w;
                    cnt += (o[j] < reg_limit) ? 1 : 0;
                }
            }
            st4(partial + pi * D + 4 * ch, acc);
            if (ch == 0) pcnt[pi] = cnt;
        }
    }
}

template <int LPR, int OPT>
__global__ __launch_bounds__(kBlock) void long_finish2_kernel(float* __restrict__ W, float* __restrict__ Mo, float* __restrict__ Vo, int D,
                                                              const uint32_t* __restrict__ keys, const float* __restrict__ reg_coef,
                                                              apply_hp hp, const int64_t* __restrict__ step_dev,
                                                              const unsigned* __restrict__ counters, const seg_long* __restrict__ longs,
                                                              const float* __restrict__ partial, const int* __restrict__ pcnt) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    hp_from_device(hp, step_dev);
    const int64_t nl = counters[1];
    for (int64_t li = gg; li < nl; li += TG) {
        const seg_long sg = longs[li];
        const uint32_t row = keys[sg.head];
        const int64_t np = (sg.len + kPiece - 1) / kPiece;
        for (int ch = sub; ch < D4; ch += LPR) {
            float* wp = W + (int64_t)row * D + 4 * ch;
            const float4 w = ld4(wp);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnt = 0;
            for (int64_t kk = 0; kk < np; ++kk) {
                const float4 g = ld4(partial + (sg.base + kk) * D + 4 * ch);
                acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
                cnt += pcnt[sg.base + kk];
            }
            apply_update<OPT>(wp, OPT ? Mo + (int64_t)row * D + 4 * ch : nullptr, OPT ? Vo + (int64_t)row * D + 4 * ch : nullptr,
                              w, acc, c * (float)cnt, hp);
        }
    }
}

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

template <int SRC>
int apply2(cdr_ctx* ctx, hipStream_t s, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D, const uint32_t* keys,
           const uint32_t* perm, int64_t n, const float* G, const item_rec* rec, const float* Usrc, int64_t reg_limit,
           const float* reg_coef, float lr, float beta1, float beta2, float eps, float wd, int64_t step, const int64_t* step_dev,
           int small, int tag) {
    float step_size = lr, bc2_sqrt = 1.f;
    if (opt == 1 && !step_dev) {
        cdr_adam_hp((double)step, lr, beta1, beta2, step_size, bc2_sqrt);       // (bc2_sqrt carries cdr_adam_hp's bc2: cdr_adam_math.h)
    }
    const apply_hp hp{lr, beta1, beta2, eps, wd, step_size, bc2_sqrt};
    const int64_t* sd = opt == 1 ? step_dev : nullptr;
    const int lpr = cdr_lpr_for(D);
    const int grid = grid_for(n, kBlock / lpr);
    const bool may_have_long = !small && n > kLongSeg;
    unsigned* counters = nullptr; seg_long* longs = nullptr; seg_piece* pieces = nullptr; int* pcnt = nullptr; float* partial = nullptr;
    int64_t long_cap = 0, piece_cap = 0;
    if (may_have_long) {
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
    cdr_time_scope ts(ctx, tag, s);
#define A2 table, exp_avg, exp_avg_sq, D, keys, perm, n, G, rec, Usrc, reg_limit, reg_coef, hp, sd, counters, longs, pieces
    if (opt == 0) { DISPATCH_LPR(lpr, apply2_kernel<L, 0, SRC><<<dim3(grid), dim3(kBlock), 0, s>>>(A2)); }
    else { DISPATCH_LPR(lpr, apply2_kernel<L, 1, SRC><<<dim3(grid), dim3(kBlock), 0, s>>>(A2)); }
#undef A2
    CDR_LAUNCH_CHECK();
    if (may_have_long) {
        const int gp = grid_for(piece_cap < 16384 ? piece_cap : 16384, kBlock / lpr);
        DISPATCH_LPR(lpr, piece_sum2_kernel<L, SRC><<<dim3(gp), dim3(kBlock), 0, s>>>(D, perm, G, rec, Usrc, reg_limit, counters, pieces, partial, pcnt));
        CDR_LAUNCH_CHECK();
        const int gl = grid_for(long_cap < 4096 ? long_cap : 4096, kBlock / lpr);
        if (opt == 0) { DISPATCH_LPR(lpr, long_finish2_kernel<L, 0><<<dim3(gl), dim3(kBlock), 0, s>>>(table, exp_avg, exp_avg_sq, D, keys, reg_coef, hp, sd, counters, longs, partial, pcnt)); }
        else { DISPATCH_LPR(lpr, long_finish2_kernel<L, 1><<<dim3(gl), dim3(kBlock), 0, s>>>(table, exp_avg, exp_avg_sq, D, keys, reg_coef, hp, sd, counters, longs, partial, pcnt)); }
        CDR_LAUNCH_CHECK();
    }
    return CDR_OK;
}

}  // namespace

extern "C" int cdr_bpr_fwd_grad_kmajor(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D,
                                       const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t S, int k, float gamma,
                                       float reg_weight, float* out9, float* GU, void* item_rec_out, int64_t* bump_a, int64_t* bump_b) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out9 && GU && item_rec_out);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && S > 0 && k >= 1 && k <= 64);
    hipStream_t s = (hipStream_t)stream;
    const int64_t B = S * (int64_t)k;
    const float invB = 1.0f / (float)B;
    const int lpr = cdr_lpr_for(D);
    item_rec* rec = (item_rec*)item_rec_out;
    int grid;
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_KMAJOR, s);
        if (k == 1) {
            grid = grid_for((S + 3) / 4, kBlock / lpr);
            DISPATCH_LPR(lpr, bpr_fwd_kmajor_kernel<L, 1, 4><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid, S, k,
                                                                                                  gamma, invB, GU, rec, ctx->partials));
        } else if (k == 2) {
            grid = grid_for((S + 2) / 3, kBlock / lpr);
            DISPATCH_LPR(lpr, bpr_fwd_kmajor_kernel<L, 2, 3><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid, S, k,
                                                                                                  gamma, invB, GU, rec, ctx->partials));
        } else {
            grid = grid_for((S + 1) / 2, kBlock / lpr);
            DISPATCH_LPR(lpr, bpr_fwd_kmajor_kernel<L, 4, 2><<<dim3(grid), dim3(kBlock), 0, s>>>(user_tab, item_tab, D, uid, pid, nid, S, k,
                                                                                                  gamma, invB, GU, rec, ctx->partials));
        }
    }
    CDR_LAUNCH_CHECK();
    kstep_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, grid, B, k, reg_weight, out9, bump_a, bump_b);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_rowwise_apply_rows(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D,
                                      const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, const float* G, int64_t reg_limit,
                                      const float* reg_coef, float lr, float beta1, float beta2, float eps, float weight_decay,
                                      int64_t step, const int64_t* step_dev, int small) {
    CDR_CHECK_ARG(ctx && table && keys_sorted && perm && G && n > 0 && D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && exp_avg && exp_avg_sq && (step > 0 || step_dev)));
    return apply2<0>(ctx, (hipStream_t)stream, opt, table, exp_avg, exp_avg_sq, D, keys_sorted, perm, n, G, nullptr, nullptr, reg_limit,
                     reg_coef, lr, beta1, beta2, eps, weight_decay, step, step_dev, small, CDR_TAG_APPLY_UNSIGNED);
}

extern "C" int cdr_rowwise_apply_scaled(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D,
                                        const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, const void* item_rec_in,
                                        const float* src_table, int64_t reg_limit, const float* reg_coef, float lr, float beta1,
                                        float beta2, float eps, float weight_decay, int64_t step, const int64_t* step_dev, int small) {
    CDR_CHECK_ARG(ctx && table && keys_sorted && perm && item_rec_in && src_table && n > 0 && D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && exp_avg && exp_avg_sq && (step > 0 || step_dev)));
    return apply2<1>(ctx, (hipStream_t)stream, opt, table, exp_avg, exp_avg_sq, D, keys_sorted, perm, n, nullptr,
                     (const item_rec*)item_rec_in, src_table, reg_limit, reg_coef, lr, beta1, beta2, eps, weight_decay, step, step_dev,
                     small, CDR_TAG_APPLY_SIGNED);
}

// The whole small-batch step in FOUR launches (forward || rank count, scatter || finish, item apply, user apply), one host call.
extern "C" int cdr_bpr_step_small(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, float* item_tab,
                                  float* item_m, float* item_v, int D, const int64_t* uid, const int64_t* pid, const int64_t* nid,
                                  int64_t S, int k, float gamma, float reg_weight, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, int64_t* step_user_dev, int64_t* step_item_dev, float* out9, float* GU,
                                  void* item_rec_buf, uint32_t* keys, uint32_t* perm, uint32_t* rank_scratch, int64_t max_rows) {
    CDR_CHECK_ARG(ctx && user_tab && item_tab && uid && pid && nid && out9 && GU && item_rec_buf && keys && perm && rank_scratch);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256 && S > 0 && k >= 1 && k <= 64);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && user_m && user_v && item_m && item_v && step_user_dev && step_item_dev));
    const int64_t B = S * (int64_t)k;
    CDR_CHECK_ARG(S + B <= ranksort::kMaxSmall);
    hipStream_t s = (hipStream_t)stream;
    ranksort::small_sort_args sa;
    const int64_t* ids0[2] = {uid, pid};
    const int64_t* ids1[2] = {nullptr, nid};
    const int64_t n0[2] = {S, S}, n1[2] = {0, B}, off[2] = {0, S};
    if (!ranksort::plan(sa, 2, ids0, n0, ids1, n1, off, max_rows)) { cdr_set_error("cdr_bpr_step_small: bad id lists"); return CDR_EINVAL; }
    const float invB = 1.0f / (float)B;
    const int lpr = cdr_lpr_for(D);
    item_rec* rec = (item_rec*)item_rec_buf;
    int64_t* bu = opt == 1 ? step_user_dev : nullptr;
    int64_t* bi = opt == 1 ? step_item_dev : nullptr;
    int fb;
    {
        cdr_time_scope ts(ctx, CDR_TAG_BPR_FWD_KMAJOR, s);
#define SFC(KC_, UNR_) DISPATCH_LPR(lpr, small_fwd_count_kernel<L, KC_, UNR_><<<dim3(fb + sa.count_blocks), dim3(kBlock), 0, s>>>( \
        user_tab, item_tab, D, uid, pid, nid, S, k, gamma, invB, GU, rec, ctx->partials, fb, sa, rank_scratch, bu, bi))
        if (k == 1) { fb = grid_for((S + 3) / 4, kBlock / lpr); SFC(1, 4); }
        else if (k == 2) { fb = grid_for((S + 2) / 3, kBlock / lpr); SFC(2, 3); }
        else { fb = grid_for((S + 1) / 2, kBlock / lpr); SFC(4, 2); }
#undef SFC
    }
    CDR_LAUNCH_CHECK();
    small_scatter_finish_kernel<<<dim3(sa.scatter_blocks + 1), dim3(kBlock), 0, s>>>(sa, rank_scratch, keys, perm, ctx->partials, fb, B, k,
                                                                                     reg_weight, out9);
    CDR_LAUNCH_CHECK();
    // the item apply reads the PRE-step user rows: it runs first
    int rc = apply2<1>(ctx, s, opt, item_tab, item_m, item_v, D, keys + S, perm + S, S + B, nullptr, rec, user_tab, S, out9 + 5, lr, beta1,
                       beta2, eps, weight_decay, 0, bi, 1, CDR_TAG_APPLY_SIGNED);
    if (rc) return rc;
    return apply2<0>(ctx, s, opt, user_tab, user_m, user_v, D, keys, perm, S, GU, nullptr, nullptr, S, out9 + 4, lr, beta1, beta2, eps,
                     weight_decay, 0, bu, 1, CDR_TAG_APPLY_UNSIGNED);
}
