// Row movers and small fused elementwise kernels: embedding gather / dense scatter-add (K1), mapped-or-target select
// (K7), activation backward, column sums, MSE, exact dense Adam (K13).  All HBM-bound: 16 B per lane, whole rows per
// wave-instruction, grid capped at 8 blocks per CU with a grid stride.
#include "cdr_common.h"
#include "cdr_adam_math.h"
#include "cdr_produce.h"

namespace {

constexpr int kBlock = 256;

inline int grid_cap(int64_t blocks) {
    const int64_t cap = CDR_NUM_CU * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---- gather: out[r,:] = tab[ids[r],:]   (optionally ids[r] < n_overlap ? mapped[r,:] : tab[ids[r],:]) -----------
template <bool SELECT>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ tab, const float* __restrict__ mapped,
                                                             int D, const int64_t* __restrict__ ids, int64_t n,
                                                             int64_t n_overlap, float* __restrict__ out) {
    const int D4 = D >> 2;
    const int64_t total = n * D4;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    constexpr int UN = 4;           // ids of UN elements first, then UN independent row loads in flight, then the stores
    for (int64_t e0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; e0 < total; e0 += stride * UN) {
        int64_t r[UN], id[UN];
        int c[UN];
        float4 v[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t e = e0 + j * stride;
            r[j] = e < total ? e / D4 : 0;
            c[j] = e < total ? (int)(e - r[j] * D4) : 0;
            id[j] = ids[r[j]];
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const float* src = (SELECT && id[j] < n_overlap) ? mapped + r[j] * D : tab + id[j] * D;
            v[j] = ld4(src + 4 * c[j]);
        }
#pragma unroll
        for (int j = 0; j < UN; ++j)
            if (e0 + j * stride < total) st4(out + r[j] * D + 4 * c[j], v[j]);
    }
}

// gather + squared row norms: out[r,:] = tab[ids[r],:], nrm2[r] = ||tab[ids[r],:]||^2 (the row-sharded step's owner side: the requester
// needs the EmbLoss norm of the rows it asked for, and the owner has them in registers anyway).  One lane group per row, four rows in flight.
template <int LPR>
__global__ __launch_bounds__(kBlock) void gather_rows_norms_kernel(const float* __restrict__ tab, int D, const int64_t* __restrict__ ids,
                                                                   int64_t n, float* __restrict__ out, float* __restrict__ nrm2) {
    constexpr int GPB = kBlock / LPR, UN = 4;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const bool live = sub < (D >> 2);
    for (int64_t base = gg; base < n; base += TG * UN) {
        int64_t id[UN];
        float4 v[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) { const int64_t r = base + j * TG; id[j] = ids[r < n ? r : n - 1]; }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t r = base + j * TG;
            v[j] = (r < n && live) ? ld4n<(LPR >= 32)>(tab + id[j] * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t r = base + j * TG;
            const float s2 = group_sum<LPR>(dot4(v[j], v[j]));
            if (r < n) {
                if (live) st4n<(LPR >= 32)>(out + r * D + 4 * sub, v[j]);
                if (sub == 0) nrm2[r] = s2;
            }
        }
    }
}

template <bool SELECT>
__global__ __launch_bounds__(kBlock) void gather_rows_scalar_kernel(const float* __restrict__ tab, const float* __restrict__ mapped,
                                                                    int D, const int64_t* __restrict__ ids, int64_t n,
                                                                    int64_t n_overlap, float* __restrict__ out) {
    const int64_t total = n * D;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        const int64_t id = ids[r];
        out[e] = (SELECT && id < n_overlap) ? mapped[r * D + c] : tab[id * D + c];
    }
}

// ---- dense scatter-add: grad_tab[ids[r],:] += scale * src[r,:]  (fp32 atomics, like torch's embedding backward) --
__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(float* __restrict__ grad_tab, int D,
                                                                  const int64_t* __restrict__ ids, int64_t n,
                                                                  const float* __restrict__ src,
                                                                  const float* __restrict__ scale) {
    const float sc = scale ? scale[0] : 1.0f;
    const int64_t total = n * D;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        atomicAdd(grad_tab + ids[r] * D + c, sc * src[e]);
    }
}

// ---- several (table, id list) pairs in ONE launch (blockIdx.y = list): the small steps are launch-bound, and SSCDR's map phase reads
// four row sets per step (sscdr.py:161-172) ---------------------------------------------------------------------------------------
constexpr int kMultiLists = 4;
struct rows_multi {
    const float* tab[kMultiLists]; float* gtab[kMultiLists];
    const int64_t* ids[kMultiLists]; int64_t n[kMultiLists];
    float* out[kMultiLists]; const float* src[kMultiLists];
    int count, D;
};

__global__ __launch_bounds__(kBlock) void gather_rows_multi_kernel(rows_multi a, int64_t* __restrict__ bump) {
    const int l = blockIdx.y;
    const float* __restrict__ tab = a.tab[l];
    const int64_t* __restrict__ ids = a.ids[l];
    float* __restrict__ out = a.out[l];
    const int D = a.D;
    const int64_t total = a.n[l] * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        out[e] = tab[ids[r] * D + (e - r * D)];
    }
    // an optional device counter of the caller's, advanced here: the launch in front of this one read it (SSCDR's sampler call number)
    if (bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) bump[0] += 1;
}

__global__ __launch_bounds__(kBlock) void scatter_add_rows_multi_kernel(rows_multi a) {
    const int l = blockIdx.y;
    float* __restrict__ g = a.gtab[l];
    const int64_t* __restrict__ ids = a.ids[l];
    const float* __restrict__ src = a.src[l];
    const int D = a.D;
    const int64_t total = a.n[l] * D, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / D;
        atomicAdd(g + ids[r] * D + (e - r * D), src[e]);
    }
}

// ---- activation backward -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void act_bwd_kernel(int act, const float* __restrict__ y, const float* __restrict__ gy,
                                                         float* __restrict__ gx, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        const float yv = y[e], g = gy[e];
        float d;
        switch (act) {
            case CDR_ACT_TANH: d = 1.0f - yv * yv; break;
            case CDR_ACT_RELU: d = yv > 0.f ? 1.0f : 0.f; break;
            case CDR_ACT_SIGMOID: d = yv * (1.0f - yv); break;
            default: d = 1.0f;
        }
        gx[e] = g * d;
    }
}

// ---- column sums (bias gradients; batch reductions of per-row parameter-gradient partials): two passes, FIXED summation order.
// Pass 1: block = 64 columns x one slab of rows, the slab's rows strided over the 4 waves with four independent accumulators per
// lane (a wave's loads are 256-B row pieces: the chain of dependent adds, not bandwidth, bounded the old one-accumulator loop at
// 12 us for 4,096 x 64); the four waves' sums are added in wave order.  Pass 2: one thread per column adds the slabs in slab order.
constexpr int kColsumMaxSlabs = 128;
__global__ __launch_bounds__(kBlock) void colsum_slab_kernel(const float* __restrict__ X, int64_t M, int64_t N, int64_t rows_per_slab,
                                                             float* __restrict__ part) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < M ? r0 + rows_per_slab : M;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < N) {
        int64_t m = r0 + wave;
        for (; m + 12 < r1; m += 16) {
            s0 += X[m * N + col]; s1 += X[(m + 4) * N + col]; s2 += X[(m + 8) * N + col]; s3 += X[(m + 12) * N + col];
        }
        for (; m < r1; m += 4) s0 += X[m * N + col];
    }
    sm[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && col < N) part[(int64_t)blockIdx.y * N + col] = ((sm[0][lane] + sm[1][lane]) + sm[2][lane]) + sm[3][lane];
}

__global__ __launch_bounds__(kBlock) void colsum_finish_kernel(const float* __restrict__ part, int slabs, int64_t N, int accumulate,
                                                               float* __restrict__ out) {
    const int64_t col = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (col >= N) return;
    float s = 0.f;
    for (int b = 0; b < slabs; ++b) s += part[(int64_t)b * N + col];
    out[col] = accumulate ? out[col] + s : s;
}

// ---- MSE ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             int64_t n, double* __restrict__ partials) {
    __shared__ double smem[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        const float d = a[e] - b[e];
        acc[0] += (double)(d * d);
    }
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * CDR_PARTIAL_STRIDE] = acc[0];
}

__global__ __launch_bounds__(kBlock) void mse_finish_kernel(const double* __restrict__ partials, int nblocks, int64_t n,
                                                            float* __restrict__ out1) {
    __shared__ double smem[4];
    double acc[1] = {0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) acc[0] += partials[(size_t)b * CDR_PARTIAL_STRIDE];
    block_sum_d<1>(acc, smem);
    if (threadIdx.x == 0) out1[0] = (float)(acc[0] / (double)n);
}

__global__ __launch_bounds__(kBlock) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                         const float* __restrict__ grad_out, float* __restrict__ ga,
                                                         float* __restrict__ gb) {
    const float go = (grad_out ? grad_out[0] : 1.0f) * 2.0f / (float)n;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        const float g = go * (a[e] - b[e]);
        if (ga) ga[e] = g;
        if (gb) gb[e] = -g;
    }
}

// ---- exact dense Adam (torch.optim.Adam single-tensor path, amsgrad=False) ------------------------------------------
//   g += wd*p ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(kBlock) void adam_dense_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                            float lr, float b1, float b2, float eps, float wd,
                                                            float step_size, float bc2_sqrt) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        float mv = m[e], vv = v[e];
        p[e] = cdr_adam_elem(p[e], g[e], mv, vv, b1, b2, eps, wd, step_size, bc2_sqrt);   // (bc2_sqrt: cdr_adam_hp's bc2) the ONE element update
        m[e] = mv; v[e] = vv;
    }
}

// capturable variant: the step count lives on the device (hipGraph replays bake host scalars in)
__global__ __launch_bounds__(kBlock) void adam_dense_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                                float lr, float b1, float b2, float eps, float wd,
                                                                const int64_t* __restrict__ step_dev) {
    float step_size, bc2_sqrt;
    cdr_adam_hp((double)step_dev[0], lr, b1, b2, step_size, bc2_sqrt);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        float mv = m[e], vv = v[e];
        p[e] = cdr_adam_elem(p[e], g[e], mv, vv, b1, b2, eps, wd, step_size, bc2_sqrt);   // (bc2_sqrt: cdr_adam_hp's bc2) the ONE element update
        m[e] = mv; v[e] = vv;
    }
}

__global__ void inc_i64_kernel(int64_t* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

// ---- the same Adam over MANY parameters in one launch (drop-in models have 10-30 parameter tensors; one launch each plus
// one counter bump each was a quarter of the C3 step).  Pointers travel by value in the kernel arguments.
constexpr int kAdamMulti = 24;
struct adam_multi_args {
    float* p[kAdamMulti];
    const float* g[kAdamMulti];
    float* m[kAdamMulti];
    float* v[kAdamMulti];
    int64_t n[kAdamMulti];
    int64_t* step[kAdamMulti];
    int blk_start[kAdamMulti + 1];
    int count;
};

// (Tried in round 3: no separate launch for the bump -- every block signs in on the upper half of tensor 0's counter word with one
//  atomicAdd and the last one writes the counters.  Same-address atomics serialise: +6 us on C1's 650 blocks, +55 us on C4's 10 k,
//  against the 4.7 us launch it removed; numbers in profiles/r03_ab_adam_signin.txt.)
// The same one-workgroup launch keeps the epoch's loss total when asked to (loss_sum[0] += loss[0]: the trainer's per-step
// `total_loss += loss.item()` of recbole Trainer._train_epoch, without the host sync and without a launch of its own).
__global__ void inc_multi_kernel(adam_multi_args a, const float* __restrict__ loss, float* __restrict__ loss_sum) {
    if (blockIdx.x == 0 && (int)threadIdx.x < a.count) a.step[threadIdx.x][0] += 1;
    if (blockIdx.x == 0 && threadIdx.x == 63 && loss_sum) loss_sum[0] += loss[0];
}

// TICKET: no counter launch in front -- the kernel computes update number step + 1 itself, and the workgroup that signs in last
// (every workgroup has read the counters by then) stores the new counts and adds the step's loss to the epoch total.  Used for grids
// of up to kSignInMaxBlocks workgroups (fatter workgroups: 16 elements per thread); beyond that the same-address sign-ins cost more
// than the launch they replace (round 3: +6 us at 650 workgroups, +55 us at 10 k).
template <bool TICKET>
__device__ __forceinline__ void adam_multi_body(const adam_multi_args& a, float lr, float b1, float b2, float eps, float wd,
                                                unsigned* __restrict__ ticket, const float* __restrict__ loss,
                                                float* __restrict__ loss_sum, unsigned nblocks) {
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.blk_start[t + 1]) ++t;
    const int nb = a.blk_start[t + 1] - a.blk_start[t], lb = (int)blockIdx.x - a.blk_start[t];
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const int64_t n = a.n[t];
    // the two bias corrections (two fp64 pow each) once per block, not once per thread: they were a third of this kernel's time on
    // the 43 MB of C4's tables and most of it on the few KB of C3's tower weights.  Same expression, same value.
    __shared__ float hp[2];
    const int64_t stride = (int64_t)nb * kBlock;
    const bool vec16 = !(n & 3) && !(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15);
    // the one-launch form's fat workgroups make at most four trips per thread: ask for every trip's operands BEFORE waiting for the two
    // bias corrections (thread 0's four fp64 pow are a microsecond the loads can travel under, in a launch that is little more)
    const bool early = TICKET && vec16 && stride * 4 >= (n >> 2);
    const int64_t e0 = (int64_t)lb * kBlock + threadIdx.x;
    float4 pv4[4], mv4[4], vv4[4], gv4[4];
    if (early) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t e = e0 + r * stride;
            if (e < (n >> 2)) { pv4[r] = ld4(p + 4 * e); mv4[r] = ld4(m + 4 * e); vv4[r] = ld4(v + 4 * e); gv4[r] = ld4(g + 4 * e); }
        }
    }
    if (threadIdx.x == 0) {
        float ss, bc;
        cdr_adam_hp((double)(a.step[t][0] + (TICKET ? 1 : 0)), lr, b1, b2, ss, bc);
        hp[0] = ss; hp[1] = bc;
    }
    __syncthreads();
    const float step_size = hp[0], bc2_sqrt = hp[1];
    auto sign_out = [&]() {
        if (TICKET && cdr_sign_in_last_wide(ticket, nblocks)) {
            if ((int)threadIdx.x < a.count) a.step[threadIdx.x][0] += 1;
            if (threadIdx.x == 63 && loss_sum) loss_sum[0] += loss[0];
        }
    };
    if (early) {
        const int64_t n4 = n >> 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t e = e0 + r * stride;
            if (e < n4) {
                pv4[r].x = cdr_adam_elem(pv4[r].x, gv4[r].x, mv4[r].x, vv4[r].x, b1, b2, eps, wd, step_size, bc2_sqrt);
                pv4[r].y = cdr_adam_elem(pv4[r].y, gv4[r].y, mv4[r].y, vv4[r].y, b1, b2, eps, wd, step_size, bc2_sqrt);
                pv4[r].z = cdr_adam_elem(pv4[r].z, gv4[r].z, mv4[r].z, vv4[r].z, b1, b2, eps, wd, step_size, bc2_sqrt);
                pv4[r].w = cdr_adam_elem(pv4[r].w, gv4[r].w, mv4[r].w, vv4[r].w, b1, b2, eps, wd, step_size, bc2_sqrt);
                st4(p + 4 * e, pv4[r]); st4(m + 4 * e, mv4[r]); st4(v + 4 * e, vv4[r]);
            }
        }
        sign_out();
        return;
    }
    if (vec16) {           // 16-B requests
        const int64_t n4 = n >> 2;
        for (int64_t e = (int64_t)lb * kBlock + threadIdx.x; e < n4; e += stride) {
            float4 pv = ld4(p + 4 * e), mv = ld4(m + 4 * e), vv = ld4(v + 4 * e);
            const float4 gv = ld4(g + 4 * e);
            pv.x = cdr_adam_elem(pv.x, gv.x, mv.x, vv.x, b1, b2, eps, wd, step_size, bc2_sqrt);
            pv.y = cdr_adam_elem(pv.y, gv.y, mv.y, vv.y, b1, b2, eps, wd, step_size, bc2_sqrt);
            pv.z = cdr_adam_elem(pv.z, gv.z, mv.z, vv.z, b1, b2, eps, wd, step_size, bc2_sqrt);
            pv.w = cdr_adam_elem(pv.w, gv.w, mv.w, vv.w, b1, b2, eps, wd, step_size, bc2_sqrt);
            st4(p + 4 * e, pv); st4(m + 4 * e, mv); st4(v + 4 * e, vv);
        }
        sign_out();
        return;
    }
    for (int64_t e = (int64_t)lb * kBlock + threadIdx.x; e < n; e += stride) {
        float mv = m[e], vv = v[e];
        p[e] = cdr_adam_elem(p[e], g[e], mv, vv, b1, b2, eps, wd, step_size, bc2_sqrt);   // shared with the deferred per-row form
        m[e] = mv; v[e] = vv;
    }
    sign_out();
}

template <bool TICKET>
__global__ __launch_bounds__(kBlock) void adam_multi_dev_kernel(adam_multi_args a, float lr, float b1, float b2, float eps, float wd,
                                                                unsigned* __restrict__ ticket, const float* __restrict__ loss,
                                                                float* __restrict__ loss_sum) {
    adam_multi_body<TICKET>(a, lr, b1, b2, eps, wd, ticket, loss, loss_sum, gridDim.x);
}

// The one-launch dense Adam of step i with the loader's batch i + 1 produced in workgroups behind its own (round 6): the two are independent
// -- the producer overwrites the batch buffers, which the backward of step i (in front of this launch) was the last to read -- and both are
// latency-bound launches of 6-8 us at the reference's batch (profiles/r06_bench_c1_kernel_stats.csv): side by side they cost one of them.
struct produce_side { cdr_produce::batch_jobs jobs; int gx[CDR_BATCH_MAX_JOBS]; int n; };
__global__ __launch_bounds__(kBlock) void adam_multi_produce_kernel(adam_multi_args a, float lr, float b1, float b2, float eps, float wd,
                                                                    unsigned* __restrict__ ticket, const float* __restrict__ loss,
                                                                    float* __restrict__ loss_sum, unsigned n_adam, produce_side ps) {
    if (blockIdx.x < n_adam) { adam_multi_body<true>(a, lr, b1, b2, eps, wd, ticket, loss, loss_sum, n_adam); return; }
    unsigned b = blockIdx.x - n_adam;
    for (int j = 0; j < ps.n; ++j) {
        if (b < (unsigned)ps.gx[j]) { cdr_produce::batch_produce_body(ps.jobs.j[j], b, (unsigned)ps.gx[j]); return; }
        b -= (unsigned)ps.gx[j];
    }
}

// mode 0: out[0] = sum_i x[i * stride] * w[i] in index order (the weighted total of a few loss scalars); mode 1: out[i] = scale[0] * w[i]
// (its backward): one launch each where torch needs a multiply and a reduction
__global__ void scalar_mix_kernel(int mode, int n, const float* __restrict__ x, int64_t stride, const float* __restrict__ w,
                                  const float* __restrict__ scale, float* __restrict__ out) {
#pragma clang fp contract(off)
    if (mode == 0) {
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int i = 0; i < n; ++i) s = s + x[i * stride] * w[i];
            out[0] = s;
        }
    } else if ((int)threadIdx.x < n) {
        out[threadIdx.x] = scale[0] * w[threadIdx.x];
    }
}

}  // namespace

extern "C" int cdr_adam_dense_dev(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, const int64_t* step_dev) {
    CDR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_dev && n > 0);
    adam_dense_dev_kernel<<<dim3(grid_cap((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_dev);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_adam_multi_dev(void* stream, int count, float* const* params, const float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* numel, int64_t* const* step_dev, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, const float* loss, float* loss_sum, unsigned* ticket) {
    CDR_CHECK_ARG(count > 0 && params && grads && exp_avg && exp_avg_sq && numel && step_dev);
    CDR_CHECK_ARG((loss == nullptr) == (loss_sum == nullptr));
    hipStream_t s = (hipStream_t)stream;
    // one launch in all when the caller lends a sign-in word and the whole update fits kSignInMaxBlocks fat workgroups
    bool fat = ticket != nullptr && count <= kAdamMulti;
    if (fat) {
        int64_t blocks = 0;
        for (int i = 0; i < count; ++i) blocks += grid_cap((numel[i] + kBlock * 16 - 1) / (kBlock * 16));
        fat = blocks <= kSignInMaxBlocks;
    }
    const int per_thread = fat ? 16 : 4;
    for (int base = 0; base < count; base += kAdamMulti) {
        adam_multi_args a{};
        a.count = count - base < kAdamMulti ? count - base : kAdamMulti;
        int blocks = 0;
        for (int i = 0; i < a.count; ++i) {
            const int j = base + i;
            CDR_CHECK_ARG(params[j] && grads[j] && exp_avg[j] && exp_avg_sq[j] && step_dev[j] && numel[j] > 0);
            a.p[i] = params[j]; a.g[i] = grads[j]; a.m[i] = exp_avg[j]; a.v[i] = exp_avg_sq[j]; a.n[i] = numel[j]; a.step[i] = step_dev[j];
            a.blk_start[i] = blocks;
            blocks += grid_cap((numel[j] + kBlock * per_thread - 1) / (kBlock * per_thread));      // ~4 (16: one-launch form) elements per thread, capped per tensor
        }
        a.blk_start[a.count] = blocks;
        if (fat) {
            adam_multi_dev_kernel<true><<<dim3(blocks), dim3(kBlock), 0, s>>>(a, lr, beta1, beta2, eps, weight_decay, ticket, loss, loss_sum);
            CDR_LAUNCH_CHECK();
            continue;
        }
        inc_multi_kernel<<<dim3(1), dim3(64), 0, s>>>(a, base == 0 ? loss : nullptr, base == 0 ? loss_sum : nullptr);
        CDR_LAUNCH_CHECK();
        adam_multi_dev_kernel<false><<<dim3(blocks), dim3(kBlock), 0, s>>>(a, lr, beta1, beta2, eps, weight_decay, nullptr, nullptr, nullptr);
        CDR_LAUNCH_CHECK();
    }
    return CDR_OK;
}

// cdr_adam_multi_dev + cdr_batch_produce_jobs in ONE launch when the update fits the one-launch form (ticket given, <= 24 tensors, <= 512 fat
// workgroups); otherwise the two calls one after the other.  Same results as the two calls (the jobs' kernels are independent of the update).
extern "C" int cdr_adam_multi_dev_produce(void* stream, int count, float* const* params, const float* const* grads, float* const* exp_avg,
                                          float* const* exp_avg_sq, const int64_t* numel, int64_t* const* step_dev, float lr, float beta1,
                                          float beta2, float eps, float weight_decay, const float* loss, float* loss_sum, unsigned* ticket,
                                          const cdr_batch_job* jobs, int n_jobs) {
    CDR_CHECK_ARG(count > 0 && params && grads && exp_avg && exp_avg_sq && numel && step_dev);
    CDR_CHECK_ARG((loss == nullptr) == (loss_sum == nullptr));
    CDR_CHECK_ARG(jobs && n_jobs >= 1 && n_jobs <= CDR_BATCH_MAX_JOBS);
    hipStream_t s = (hipStream_t)stream;
    bool fat = ticket != nullptr && count <= kAdamMulti;
    int64_t blocks = 0;
    if (fat) {
        for (int i = 0; i < count; ++i) blocks += grid_cap((numel[i] + kBlock * 16 - 1) / (kBlock * 16));
        fat = blocks <= kSignInMaxBlocks;
    }
    if (!fat) {
        int rc = cdr_adam_multi_dev(stream, count, params, grads, exp_avg, exp_avg_sq, numel, step_dev, lr, beta1, beta2, eps, weight_decay, loss,
                                    loss_sum, ticket);
        if (rc) return rc;
        return cdr_batch_produce_jobs(stream, jobs, n_jobs);
    }
    adam_multi_args a{};
    a.count = count;
    int nb = 0;
    for (int i = 0; i < count; ++i) {
        CDR_CHECK_ARG(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && step_dev[i] && numel[i] > 0);
        a.p[i] = params[i]; a.g[i] = grads[i]; a.m[i] = exp_avg[i]; a.v[i] = exp_avg_sq[i]; a.n[i] = numel[i]; a.step[i] = step_dev[i];
        a.blk_start[i] = nb;
        nb += grid_cap((numel[i] + kBlock * 16 - 1) / (kBlock * 16));
    }
    a.blk_start[count] = nb;
    produce_side ps{};
    ps.n = n_jobs;
    int total = nb;
    for (int j = 0; j < n_jobs; ++j) {
        const cdr_batch_job& J = jobs[j];
        CDR_CHECK_ARG(J.users_all && J.cursor && J.out_users && J.S > 0 && J.k >= 0 && J.n_rows > 0);
        if (J.k > 0) CDR_CHECK_ARG(J.items_all && J.out_items && (J.pointwise || J.out_neg));
        ps.jobs.j[j] = J;
        ps.gx[j] = (int)cdr_produce::job_grid(J);
        total += ps.gx[j];
    }
    adam_multi_produce_kernel<<<dim3((unsigned)total), dim3(kBlock), 0, s>>>(a, lr, beta1, beta2, eps, weight_decay, ticket, loss, loss_sum, (unsigned)nb, ps);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_scalar_mix(void* stream, int mode, int n, const float* x, int64_t x_stride, const float* w, const float* scale, float* out) {
    CDR_CHECK_ARG(n > 0 && n <= 64 && w && out && (mode == 0 ? x != nullptr : scale != nullptr));
    scalar_mix_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(mode, n, x, x_stride, w, scale, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_inc_i64(void* stream, int64_t* counter) {
    CDR_CHECK_ARG(counter);
    inc_i64_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(counter);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_gather_rows(void* stream, const float* tab, int D, const int64_t* ids, int64_t n, float* out) {
    CDR_CHECK_ARG(tab && ids && out && D > 0 && n > 0);
    hipStream_t s = (hipStream_t)stream;
    if ((D & 3) == 0) {
        const int64_t total = n * (D >> 2);
        gather_rows_kernel<false><<<dim3(grid_cap((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(tab, nullptr, D, ids, n, 0, out);
    } else {
        const int64_t total = n * D;
        gather_rows_scalar_kernel<false><<<dim3(grid_cap((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(tab, nullptr, D, ids, n, 0, out);
    }
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_gather_rows_norms(void* stream, const float* tab, int D, const int64_t* ids, int64_t n, float* out, float* nrm2) {
    CDR_CHECK_ARG(tab && ids && out && nrm2 && D > 0 && (D & 3) == 0 && D <= 256 && n > 0);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const int64_t groups = (n + 3) / 4;
    switch (lpr) {
#define GRN_CASE(L) case L: gather_rows_norms_kernel<L><<<dim3(grid_cap((groups + (kBlock / L) - 1) / (kBlock / L))), dim3(kBlock), 0, s>>>(tab, D, ids, n, out, nrm2); break;
        GRN_CASE(1) GRN_CASE(2) GRN_CASE(4) GRN_CASE(8) GRN_CASE(16) GRN_CASE(32)
        default: gather_rows_norms_kernel<64><<<dim3(grid_cap((groups + 3) / 4)), dim3(kBlock), 0, s>>>(tab, D, ids, n, out, nrm2); break;
#undef GRN_CASE
    }
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_select_mapped(void* stream, const float* mapped, const float* tab, int D, const int64_t* ids, int64_t n,
                                 int64_t n_overlap, float* out) {
    CDR_CHECK_ARG(mapped && tab && ids && out && D > 0 && n > 0);
    hipStream_t s = (hipStream_t)stream;
    if ((D & 3) == 0) {
        const int64_t total = n * (D >> 2);
        gather_rows_kernel<true><<<dim3(grid_cap((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(tab, mapped, D, ids, n, n_overlap, out);
    } else {
        const int64_t total = n * D;
        gather_rows_scalar_kernel<true><<<dim3(grid_cap((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(tab, mapped, D, ids, n, n_overlap, out);
    }
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_scatter_add_rows(void* stream, float* grad_tab, int D, const int64_t* ids, int64_t n, const float* src,
                                    const float* scale) {
    CDR_CHECK_ARG(grad_tab && ids && src && D > 0 && n > 0);
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = n * D;
    scatter_add_rows_kernel<<<dim3(grid_cap((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(grad_tab, D, ids, n, src, scale);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_gather_rows_multi(void* stream, int count, const float* const* tabs, int D, const int64_t* const* ids,
                                     const int64_t* n, float* const* outs, int64_t* bump_counter) {
    CDR_CHECK_ARG(count >= 1 && count <= kMultiLists && tabs && ids && n && outs && D > 0);
    rows_multi a{};
    a.count = count; a.D = D;
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        CDR_CHECK_ARG(n[i] >= 0 && (n[i] == 0 || (tabs[i] && ids[i] && outs[i])));
        a.tab[i] = tabs[i]; a.ids[i] = ids[i]; a.n[i] = n[i]; a.out[i] = outs[i];
        if (n[i] > nmax) nmax = n[i];
    }
    gather_rows_multi_kernel<<<dim3(grid_cap((nmax * D + kBlock - 1) / kBlock), count), dim3(kBlock), 0, (hipStream_t)stream>>>(a, bump_counter);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_scatter_add_rows_multi(void* stream, int count, float* const* grad_tabs, int D, const int64_t* const* ids,
                                          const int64_t* n, const float* const* srcs) {
    CDR_CHECK_ARG(count >= 1 && count <= kMultiLists && grad_tabs && ids && n && srcs && D > 0);
    rows_multi a{};
    a.count = count; a.D = D;
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        CDR_CHECK_ARG(n[i] >= 0 && (n[i] == 0 || (grad_tabs[i] && ids[i] && srcs[i])));
        a.gtab[i] = grad_tabs[i]; a.ids[i] = ids[i]; a.n[i] = n[i]; a.src[i] = srcs[i];
        if (n[i] > nmax) nmax = n[i];
    }
    if (nmax == 0) return CDR_OK;
    scatter_add_rows_multi_kernel<<<dim3(grid_cap((nmax * D + kBlock - 1) / kBlock), count), dim3(kBlock), 0, (hipStream_t)stream>>>(a);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_act_bwd(void* stream, int act, const float* y, const float* gy, float* gx, int64_t n) {
    CDR_CHECK_ARG(y && gy && gx && n > 0);
    CDR_CHECK_ARG(act >= CDR_ACT_NONE && act <= CDR_ACT_SIGMOID);
    act_bwd_kernel<<<dim3(grid_cap((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(act, y, gy, gx, n);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_colsum(cdr_ctx* ctx, void* stream, const float* X, int64_t M, int64_t N, float* out, int accumulate) {
    CDR_CHECK_ARG(ctx && X && out && M > 0 && N > 0);
    int64_t rows_per_slab = 64;
    int64_t slabs = (M + rows_per_slab - 1) / rows_per_slab;
    if (slabs > kColsumMaxSlabs) { rows_per_slab = (M + kColsumMaxSlabs - 1) / kColsumMaxSlabs; slabs = (M + rows_per_slab - 1) / rows_per_slab; }
    void* part = nullptr;
    int rc = cdr_ctx_scratch(ctx, (size_t)slabs * (size_t)N * sizeof(float), &part);
    if (rc != CDR_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    colsum_slab_kernel<<<dim3((unsigned)((N + 63) / 64), (unsigned)slabs), dim3(kBlock), 0, s>>>(X, M, N, rows_per_slab, (float*)part);
    CDR_LAUNCH_CHECK();
    colsum_finish_kernel<<<dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>((const float*)part, (int)slabs, N, accumulate, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_mse_fwd(cdr_ctx* ctx, void* stream, const float* a, const float* b, int64_t n, float* out1) {
    CDR_CHECK_ARG(ctx && a && b && out1 && n > 0);
    hipStream_t s = (hipStream_t)stream;
    int64_t g = (n + kBlock * 4 - 1) / (kBlock * 4);
    if (g > CDR_MAX_PARTIAL_BLOCKS) g = CDR_MAX_PARTIAL_BLOCKS;
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    mse_partial_kernel<<<dim3((unsigned)g), dim3(kBlock), 0, s>>>(a, b, n, ctx->partials);
    CDR_LAUNCH_CHECK();
    mse_finish_kernel<<<dim3(1), dim3(kBlock), 0, s>>>(ctx->partials, (int)g, n, out1);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_mse_bwd(void* stream, const float* a, const float* b, int64_t n, const float* grad_out, float* ga,
                           float* gb) {
    CDR_CHECK_ARG(a && b && n > 0 && (ga || gb));
    mse_bwd_kernel<<<dim3(grid_cap((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(a, b, n, grad_out, ga, gb);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_adam_dense(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step) {
    CDR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0);
    float step_size, bc2_sqrt;
    cdr_adam_hp((double)step, lr, beta1, beta2, step_size, bc2_sqrt);
    adam_dense_kernel<<<dim3(grid_cap((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
