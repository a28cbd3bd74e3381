// EMCDR's OVERLAP phase ("overlapped-user transfer step": emcdr.py:156-168 calculate_map_loss, mapping :59-64,86-93) as TWO
// launches for the batches the reference's own loader produces.
//
// OverlapDataloader hands out slices of a shuffled arange(num_overlap) (data/dataloader.py:37-52, dataset.py:694-696): the ids
// of one batch are DISTINCT.  Every row is then touched by exactly one occurrence, so nothing has to be sorted or
// de-duplicated and the optimizer can run where the gradient is produced:
//
//   map_pipe_kernel     (the linear mapping, Ds == Dt in {64, 128}: EMCDR's default and BASELINE C5) one 512-thread workgroup per
//                       CU: four MFMA waves do forward / loss / weight-gradient tiles / dL/dS on 32 ids at a time out of LDS, four
//                       row waves keep the next block's six rows per id in flight with LDS DMA and apply SGD / Adam to the
//                       two table rows in place -- see the comment above the kernel.  3.3 TB/s at OB = 65,536.
//   map_pipe2_kernel    the same two wave groups for the reference's default mapping, Linear + Tanh + Linear with D and H in {64, 128}
//                       (gz and dL/dS share an LDS buffer; seven barriers per block): 1.6-1.8 TB/s.
//   map_step_kernel     (every other mapping shape: deeper MLPs, odd widths, Ds != Dt) a workgroup owns 32 ids at a time: gathers S[id]
//                       and T[id] into LDS, runs the mapping function
//                       (Linear, or Linear+Tanh ... Linear) on v_mfma_f32_32x32x2_f32 with the activations in LDS,
//                       d = mapped - T[id] (MSE partial), walks the mapping backwards to dL/dS[id], accumulates the mapping's
//                       weight gradients of ALL its ids in registers (gz^T x input, one accumulator per 32x32 tile), and
//                       applies SGD / Adam to the two table rows in place: 2 row reads, 4 moment reads, 6 writes per id --
//                       the 6,144 B/id of SURVEY 8d and not a byte more (round 1: ~20 launches through autograd, compact
//                       gradient rows written and re-read, a radix sort: 0.93 TB/s).
//   map_finish_kernel   adds the per-workgroup partials in workgroup order (loss, weight and bias gradients) and takes the
//                       exact dense Adam step on the mapping's parameters, one thread per element.
//
// Update counts live on the device (tables: read as count + 1 by launch 1, advanced by launch 2; mapping parameters: advanced
// by launch 1, read by launch 2), so the pair is hipGraph-capturable.  Batches with repeated ids take the general path
// (fused.FusedMapStep: gather -> mapping -> MSE -> sort -> row-wise applies).
#include <string.h>
#include <type_traits>
#include <cstdlib>
#include "cdr_common.h"
#include "cdr_adam_math.h"

// The rows of the two user tables are touched once per step: their LDS-DMA loads and their write-backs are issued non-temporal (rows of
// 512 bytes; replicated A/B on one MI355X: profiles/r06_ab_map_nt.txt).  -DCDR_NO_MAP_NT builds the plain-policy form.
#ifndef CDR_NO_MAP_NT
#define MAP_NT " nt"
#define MAP_ST4(p, v) st4n<true>((p), (v))
#else
#define MAP_NT ""
#define MAP_ST4(p, v) st4((p), (v))
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMapMaxL = CDR_MAP_MAX_LAYERS;
constexpr int kRows = 32;
constexpr int kSlots = 8;                      // weight-gradient tiles a wave may own (128 accumulator registers)

struct map_net {
    int L, vec, ntiles, nbias;
    int dims[kMapMaxL + 1];
    int act[kMapMaxL];                         // CDR_ACT_* after layer l
    int tile_off[kMapMaxL + 1];                // first weight-gradient tile of layer l
    int bias_off[kMapMaxL + 1];
    int buf_off[kMapMaxL + 1];                 // LDS float offset of layer l's INPUT buffer (buf_off[0] = X); [L] = mapped / gz
    const float* W[kMapMaxL];
    const float* b[kMapMaxL];
};
struct map_opt { float lr, b1, b2, eps, wd; int opt; };
struct map_params { float* W[kMapMaxL]; float* b[kMapMaxL]; float* mW[kMapMaxL]; float* vW[kMapMaxL]; float* mb[kMapMaxL]; float* vb[kMapMaxL];
                    int64_t* sW[kMapMaxL]; int64_t* sb[kMapMaxL]; };

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
__device__ __forceinline__ float4 ldw4(const float* p, bool vec) {
    return vec ? ld4(p) : make_float4(p[0], p[1], p[2], p[3]);
}
#define MFMA4(acc, a, b)                                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).x, (b).x, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).y, (b).y, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).z, (b).z, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).w, (b).w, acc, 0, 0, 0)
#define MF1(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void adam_hp(const map_opt& o, const int64_t* step_dev, int64_t plus, float& step_size, float& bc2_sqrt) {
    step_size = o.lr; bc2_sqrt = 1.f;
    if (o.opt == 1) {
        cdr_adam_hp((double)(step_dev[0] + plus), o.lr, o.b1, o.b2, step_size, bc2_sqrt);      // (bc2_sqrt carries cdr_adam_hp's bc2: cdr_adam_math.h)
    }
}

// one element of torch.optim.Adam / SGD (same arithmetic as cdr_step.hip's apply_update)
__device__ __forceinline__ float upd1(float w, float g, float& m, float& v, const map_opt& o, float step_size, float bc2_sqrt) {
    if (o.wd != 0.f) g += o.wd * w;
    if (o.opt == 0) return w - o.lr * g;
    m += (g - m) * (1.0f - o.b1);
    v = o.b2 * v + (1.0f - o.b2) * g * g;
    return w - cdr_adam_term(m, v, step_size, bc2_sqrt, o.eps);
}

// (round 4 had a second form of this update on v_rcp_f32 / v_sqrt_f32 for map_pipe_kernel alone; since round 5 EVERY Adam kernel of the
// library takes its update term from cdr_adam_term, so the pipe kernels call the same function as everything else)
__device__ __forceinline__ float updq(float w, float g, float& m, float& v, const map_opt& o, float step_size, float bc2) {
    return upd1(w, g, m, v, o, step_size, bc2);
}

// SLOTS: weight-gradient tiles per wave (4 -> 64 accumulator registers, 8 -> 128)
template <int SLOTS>
__global__ __launch_bounds__(256, 2) void map_step_kernel(map_net net, map_opt opt, float* __restrict__ S, float* __restrict__ mS,
                                                       float* __restrict__ vS, float* __restrict__ T, float* __restrict__ mT,
                                                       float* __restrict__ vT, const int64_t* __restrict__ idx, int64_t n,
                                                       const int64_t* __restrict__ step_s, const int64_t* __restrict__ step_t,
                                                       int gx_off, int t_off, float* __restrict__ wpart, double* __restrict__ lpart,
                                                       map_params bump) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float rowok[kRows];
    __shared__ int64_t rowid[kRows];
    __shared__ double red[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int L = net.L, Ds = net.dims[0], Dt = net.dims[L];
    const bool vec = net.vec != 0;
    float* X = smem + net.buf_off[0];
    float* GX = smem + gx_off;
    const int XS = Ds + 4, TS = Dt + 4;
    (void)t_off;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && t == 0) {                          // the mapping parameters' update counts: read by launch 2 only
        for (int l = 0; l < L; ++l) { if (bump.sW[l]) bump.sW[l][0] += 1; if (bump.sb[l]) bump.sb[l][0] += 1; }
    }
    float ss_s, bc_s, ss_t, bc_t;
    adam_hp(opt, step_s, 1, ss_s, bc_s);
    adam_hp(opt, step_t, 1, ss_t, bc_t);
    const float gscale = 2.0f / ((float)n * (float)Dt);        // d mean((a-b)^2) / da
    f32x16 wacc[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) wacc[q] = zero16();
    float bacc[2] = {0.f, 0.f};
    double lsum = 0.0;
    const int64_t nrb = (n + kRows - 1) / kRows;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        {   // ---- gather S[id] and T[id]: 8 threads per row
            const int row = t >> 3, c0 = t & 7;
            const int64_t g = rb * kRows + row;
            const bool valid = g < n;
            const int64_t id = idx[valid ? g : n - 1];
            for (int c = c0; c < (Ds >> 2); c += 8) st4(X + row * XS + 4 * c, ld4(S + id * Ds + 4 * c));
            // (the target rows are NOT staged: the last layer's epilogue and the apply read them straight from the table -- three
            //  16.9-KB buffers instead of four let THREE workgroups share a CU, and resident workgroups are what hides the ~20-us
            //  latency of a block's random-row traffic)
            if (c0 == 0) { rowok[row] = valid ? 1.f : 0.f; rowid[row] = id; }
        }
        lds_barrier();
        // (Tried and dropped, each measured with wall_clock64 stamps and at OB = 65,536: touching the next block's rows and moments
        //  one 4-byte load per line ahead of time -- the unloaded gather falls from 20 to 4 us but under load the touched lines are
        //  evicted again and HBM traffic grows 1.5x for no gain; holding the moments in registers from gather to apply -- 37 spills,
        //  slower; issuing all 16 moment loads of the apply at once -- 17-23 us against 12 for the chunk-by-chunk loop; three
        //  resident workgroups per CU -- no faster, a third more partials.)
        // ---- forward through the mapping (emcdr.py:86-93); the last layer's epilogue leaves gz = dL/d mapped in its buffer
        for (int l = 0; l < L; ++l) {
            const int din = net.dims[l], dout = net.dims[l + 1];
            const float* Xin = smem + net.buf_off[l];
            float* Xout = smem + net.buf_off[l + 1];
            const int IS = din + 4, OS = dout + 4;
            const int NT = (dout + 31) >> 5, KS = (din + 7) >> 3, KG = (KS + 3) >> 2;
            const bool last = l == L - 1;
            for (int job = wave; job < NT; job += 4) {
                const int ncol = job * 32 + li;
                const bool nv = ncol < dout;
                const float* wm = net.W[l] + (int64_t)(nv ? ncol : 0) * din;
                const float* xo = Xin + li * IS;
                f32x16 am = zero16();
                float4 nm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int k = 8 * j + 4 * lh; nm[j] = (nv && k < din) ? ldw4(wm + k, vec) : z4; }
                for (int g = 0; g < KG; ++g) {                       // four K steps per request, next group in flight (cdr_conet.hip)
                    float4 cm[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) cm[j] = nm[j];
                    if (g + 1 < KG) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const int k = 8 * (4 * (g + 1) + j) + 4 * lh; nm[j] = (nv && k < din) ? ldw4(wm + k, vec) : z4; }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 8 * (4 * g + j) + 4 * lh;
                        const float4 a0 = k < din ? ld4(xo + k) : z4;
                        MFMA4(am, a0, cm[j]);
                    }
                }
                if (nv) {
                    const float bv = net.b[l] ? net.b[l][ncol] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float v = am[r] + bv;
                        if (net.act[l] == CDR_ACT_TANH) v = tanhf(v);
                        if (last) {
                            const float d = rowok[row] != 0.f ? v - T[rowid[row] * Dt + ncol] : 0.f;   // nn.MSELoss (emcdr.py:81,162)
                            lsum += (double)d * (double)d;
                            v = gscale * d;
                        }
                        Xout[row * OS + ncol] = v;
                    }
                }
            }
            lds_barrier();
        }
        // ---- backward: weight gradients of every layer into the wave's register tiles, data gradient down to the source rows
        for (int l = L - 1; l >= 0; --l) {
            const int din = net.dims[l], dout = net.dims[l + 1];
            float* Ain = smem + net.buf_off[l];                      // layer input (post-activation of layer l-1)
            const float* Gz = smem + net.buf_off[l + 1];             // dL/d(pre-activation output of layer l)
            const int IS = din + 4, OS = dout + 4;
            const int MT = (dout + 31) >> 5, NTW = (din + 31) >> 5;
            // (1) dW_l += gz^T x input: this wave's tiles of layer l
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                const int tile = wave + 4 * q;
                if (tile >= net.tile_off[l] && tile < net.tile_off[l + 1]) {             // wave-uniform
                    const int loc = tile - net.tile_off[l], mt = loc / NTW, nt = loc - mt * NTW;
                    const int m = mt * 32 + li, nn = nt * 32 + li;
                    const bool mv = m < dout, nv = nn < din;
#pragma unroll
                    for (int s = 0; s < kRows / 8; ++s) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int row = 8 * s + 4 * lh + c;
                            const float a = mv ? Gz[row * OS + m] : 0.f;
                            const float b = nv ? Ain[row * IS + nn] : 0.f;
                            MF1(wacc[q], a, b);
                        }
                    }
                }
            }
            // bias gradients: column sums of gz over the 32 rows, one thread per column (fixed row order)
            if (net.b[l]) {                                          // thread t owns bias elements t and t + 256 of the flat list
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int c = t + 256 * sl - net.bias_off[l];
                    if (c >= 0 && c < dout) {
                        float sum = 0.f;
                        for (int row = 0; row < kRows; ++row) sum += Gz[row * OS + c];
                        bacc[sl] += sum;
                    }
                }
            }
            // (2) g_in = gz W_l ; for l > 0 folded with the previous layer's activation derivative, IN PLACE over its output
            float* Gout = l > 0 ? Ain : GX;
            const int GS = l > 0 ? IS : XS;
            const int NT = (din + 31) >> 5, KS = (dout + 7) >> 3;
            lds_barrier();                                            // every wave is done reading Ain as the dW operand
            for (int job = wave; job < NT; job += 4) {
                const int ncol = job * 32 + li;
                const bool nv = ncol < din;
                const float* w0 = net.W[l] + (nv ? ncol : 0);
                const float* ao = Gz + li * OS;
                f32x16 am = zero16();
                float4 nb = z4;
                if (nv && 4 * lh < dout) { const float* p = w0 + (int64_t)(4 * lh) * din; nb = make_float4(p[0], p[din], p[2 * din], p[3 * din]); }
                for (int s = 0; s < KS; ++s) {
                    const int k = 8 * s + 4 * lh;
                    const float4 cb = nb;
                    const int kn = k + 8;
                    nb = z4;
                    if (nv && kn < dout) { const float* p = w0 + (int64_t)kn * din; nb = make_float4(p[0], p[din], p[2 * din], p[3 * din]); }
                    const float4 a0 = k < dout ? ld4(ao + k) : z4;
                    MFMA4(am, a0, cb);
                }
                if (nv) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float v = am[r];
                        if (l > 0 && net.act[l - 1] == CDR_ACT_TANH) { const float a = Ain[row * IS + ncol]; v *= (1.0f - a * a); }
                        Gout[row * GS + ncol] = v;
                    }
                }
            }
            lds_barrier();
        }
        {   // ---- the two table rows of every id: SGD / Adam in place (w from LDS: the pre-step row gathered above)
            const int row = t >> 3, c0 = t & 7;
            const int64_t g = rb * kRows + row;
            if (g < n) {
                const int64_t id = idx[g];
                const float* gm = smem + net.buf_off[L];
                for (int c = c0; c < (Ds >> 2); c += 8) {
                    const int64_t o = id * Ds + 4 * c;
                    const float4 w = ld4(X + row * XS + 4 * c), gg = ld4(GX + row * XS + 4 * c);
                    float4 m = z4, v = z4;
                    if (opt.opt) { m = ld4(mS + o); v = ld4(vS + o); }
                    float4 wn;
                    wn.x = upd1(w.x, gg.x, m.x, v.x, opt, ss_s, bc_s); wn.y = upd1(w.y, gg.y, m.y, v.y, opt, ss_s, bc_s);
                    wn.z = upd1(w.z, gg.z, m.z, v.z, opt, ss_s, bc_s); wn.w = upd1(w.w, gg.w, m.w, v.w, opt, ss_s, bc_s);
                    MAP_ST4(S + o, wn);
                    if (opt.opt) { MAP_ST4(mS + o, m); MAP_ST4(vS + o, v); }
                }
                for (int c = c0; c < (Dt >> 2); c += 8) {
                    const int64_t o = id * Dt + 4 * c;
                    const float4 w = ld4(T + o), gn = ld4(gm + row * TS + 4 * c);                      // dL/dT[id] = -dL/d mapped
                    float4 m = z4, v = z4;
                    if (opt.opt) { m = ld4(mT + o); v = ld4(vT + o); }
                    float4 wn;
                    wn.x = upd1(w.x, -gn.x, m.x, v.x, opt, ss_t, bc_t); wn.y = upd1(w.y, -gn.y, m.y, v.y, opt, ss_t, bc_t);
                    wn.z = upd1(w.z, -gn.z, m.z, v.z, opt, ss_t, bc_t); wn.w = upd1(w.w, -gn.w, m.w, v.w, opt, ss_t, bc_t);
                    MAP_ST4(T + o, wn);
                    if (opt.opt) { MAP_ST4(mT + o, m); MAP_ST4(vT + o, v); }
                }
            }
        }
        lds_barrier();
    }
    // ---- this workgroup's partials: weight-gradient tiles (accumulator order), bias sums, loss
    const size_t pstride = (size_t)net.ntiles * 1024 + net.nbias;
    float* o = wpart + (size_t)blockIdx.x * pstride;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int tile = wave + 4 * q;
        if (tile < net.ntiles) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)tile * 1024 + r * 64 + lane] = wacc[q][r];
        }
    }
    if (t < net.nbias) o[(size_t)net.ntiles * 1024 + t] = bacc[0];
    if (t + 256 < net.nbias) o[(size_t)net.ntiles * 1024 + 256 + t] = bacc[1];
    lsum = wave_sum_d(lsum);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (t == 0) lpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- the linear mapping (emcdr.py:59-64: Linear(Ds, Dt, bias=False)) with Ds == Dt in {64, 128}: two wave groups ------------------
// map_step_kernel is bound by LATENCY, not bandwidth: per 32-id block a workgroup waits three times for random rows out of
// 25-GB tables (source rows; target rows in the epilogue; four moment rows in the apply) and its MFMA loops wait behind the apply's
// stores (vmcnt retires in order): ~50 us per block, two workgroups per CU to hide it, 1.8 TB/s.  Here ONE 512-thread workgroup
// per CU splits into
//   * four MFMA waves (one per SIMD): forward, loss + gz, weight-gradient tiles, dL/dS -- LDS in, LDS out, no table traffic, no
//     stores; their only global loads are the mapping's weights (L2);
//   * four row waves that keep the NEXT block's six rows per id in flight with global_load_lds (LDS DMA: no registers; counted by
//     THEIR vmcnt only) and run the optimizer: Adam on the target rows of block b while the MFMA waves do its backward, Adam on
//     the source rows of block b while they do the forward of block b + 1.  Each lane updates exactly the 16-byte chunks it
//     prefetched itself (same lane -> (row, chunk) map for the DMA and for the update), so the staging arrays need no
//     synchronisation between row waves and are refilled the moment the lane is done with them.
// The staged rows are XOR-swizzled through the DMA's SOURCE address (LDS DMA writes lane-linear, so padding is not available:
// physical chunk p of row r holds logical chunk p ^ (r mod chunks)): the MFMA operand reads (lane = row, same logical chunk) and
// the row waves' own reads are both conflict-free and no copy into a padded operand buffer is needed.
// LDS at D = 128: gz 16.9 KB + dL/dS 16.9 KB + source rows 2 x 16 KB (double-buffered: block b's are the update's `w` while
// b + 1's arrive) + T, mT, vT, mS, vS 5 x 16 KB = 145 KB.
#ifdef CDR_MAP_PROF
__device__ long long g_map_prof[64];
#define MP_STAMP(slot) do { if (blockIdx.x == 7 && k == 2 && (t == 0 || t == 256)) g_map_prof[(t ? 32 : 0) + (slot)] = wall_clock64(); } while (0)
#else
#define MP_STAMP(slot) do { } while (0)
#endif
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS DMA as an asm statement: through __builtin_amdgcn_global_load_lds hipcc treats every later ds_read of the wave as a possible
// reader of the in-flight DMA and drains it with vmcnt(0) first -- exactly the wait this kernel exists to avoid.  M0 (the
// wave-uniform LDS destination) is written and restored inside the statement.
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MAP_NT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

template <int NI>
__global__ __launch_bounds__(512, 1) void map_pipe_kernel(map_net net, map_opt opt, float* __restrict__ S, float* __restrict__ mS,
                                                         float* __restrict__ vS, float* __restrict__ T, float* __restrict__ mT,
                                                         float* __restrict__ vT, const int64_t* __restrict__ idx, int64_t n,
                                                         const int64_t* __restrict__ step_s, const int64_t* __restrict__ step_t,
                                                         float* __restrict__ wpart, double* __restrict__ lpart, map_params bump) {
    constexpr int D = 32 * NI;                 // Ds == Dt
    constexpr int LR = D / 4;                  // 16-byte chunks (= DMA lanes) per row
    constexpr int RPI = 64 / LR;               // rows per DMA instruction (1 KiB of LDS)
    constexpr int GS = D + 4;                  // padded row stride of the two MFMA-written buffers
    constexpr int SLOTS = NI * NI / 4 > 0 ? NI * NI / 4 : 1;       // weight-gradient tiles per MFMA wave
    constexpr int NJ = (NI + 3) / 4;           // 32-column jobs per MFMA wave
    constexpr int NQ = NI;                     // DMA instructions per row wave and array (4 row waves)
    // target rows updated before barrier M (the rest after it).  Stamps of one block (tools/prof_mapstep.py): whatever update arithmetic
    // a row wave has left after M runs against the dL/dS contraction of its SIMD (resident weights: back-to-back MFMAs) and takes 2.5-3 us
    // per quarter of the rows where a quarter takes 0.9 us before M, and barrier E then waits for it.  Halves: 0.1251 ms per launch at
    // OB = 65,536; three quarters before M: 0.1238; everything: 0.1223 (alternating processes on one box, profiles/r03_ab_map_qt.txt).
    constexpr int QT = NQ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float rok[3][kRows];
    __shared__ int64_t rid[3][kRows];
    __shared__ double red[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li0 = lane & 31, lh0 = lane >> 5;
    const bool rowwave = wave >= 4;
    float* GZ = smem;                          // [32][GS]  mapped -> gz = dL/d mapped
    float* GX = GZ + kRows * GS;               // [32][GS]  dL/dS
    float* SS = GX + kRows * GS;               // [2][32][D] source rows (swizzled)
    float* ST = SS + 2 * kRows * D;            // [32][D] each, swizzled alike
    float* SMT = ST + kRows * D;
    float* SVT = SMT + kRows * D;
    float* SMS = SVT + kRows * D;
    float* SVS = SMS + kRows * D;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool adam = opt.opt != 0;
    float ss_s, bc_s, ss_t, bc_t;
    adam_hp(opt, step_s, 1, ss_s, bc_s);
    adam_hp(opt, step_t, 1, ss_t, bc_t);
    const float gscale = 2.0f / ((float)n * (float)D);
    const float* __restrict__ W = net.W[0];
    if (blockIdx.x == 0 && t == 0 && bump.sW[0]) bump.sW[0][0] += 1;
    const int64_t nrb = (n + kRows - 1) / kRows;
    f32x16 wacc[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) wacc[q] = zero16();
    double lsum = 0.0;

    // (address arithmetic below is re-derived per phase from a laundered lane id: left alone, LICM hoists ~200 loop-invariant LDS /
    //  global offsets out of the block loop and spills them -- and a spill reload inside a row wave is a vector memory load whose
    //  wait drains the DMA queue)
    auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };
    // ---- row waves: lane <-> (row, physical chunk) of DMA instruction i = pw + 4 q, q < NQ
    const int pw = wave - 4;
    // byte offsets of this lane's NQ chunks inside a table, for the rows of one block: all ids are read from LDS first (one wait),
    // and the result serves every table staged for that block
    auto row_offsets = [&](const int64_t* ids, int64_t (&off)[NQ]) {
        const int ln = fresh(lane);
        int64_t idv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) idv[q] = ids[(pw + 4 * q) * RPI + ln / LR];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = (pw + 4 * q) * RPI + ln / LR, p = ln % LR;
            off[q] = (idv[q] * D + 4 * (p ^ (row & (LR - 1)))) * (int64_t)sizeof(float);
        }
    };
    // NQ DMA instructions of one table in ONE asm statement: M0 saved once, set per instruction, restored once
    auto stage = [&](const float* __restrict__ tab, unsigned dst_bytes, const int64_t (&off)[NQ]) {
        const char* base = reinterpret_cast<const char*>(tab);
        const unsigned d0 = __builtin_amdgcn_readfirstlane(dst_bytes + (unsigned)pw * 1024u);   // instruction i = pw + 4 q writes 1 KiB at i * 1 KiB
        unsigned keep;
        if constexpr (NQ == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off" MAP_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(base + off[0]), "v"(base + off[1]), "v"(base + off[2]), "v"(base + off[3]),
                           "s"(d0), "s"(d0 + 4096u), "s"(d0 + 8192u), "s"(d0 + 12288u)
                         : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MAP_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(base + off[0]), "v"(base + off[1]), "s"(d0), "s"(d0 + 4096u) : "memory");
    };
    auto lds_off = [](const float* p) {
        return (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p);
    };
    const unsigned bSS = lds_off(SS), bST = lds_off(ST), bSMT = lds_off(SMT), bSVT = lds_off(SVT), bSMS = lds_off(SMS), bSVS = lds_off(SVS);
    int64_t roff[NQ];
    // Adam / SGD on this lane's chunks of DMA instructions [Q0, Q1): g = sign * G[row][logical chunk].  Every operand of the
    // whole range is requested before the arithmetic starts (one wave per role and SIMD: nothing else hides the LDS latency)
    auto apply = [&](auto Q0c, auto Q1c, float* __restrict__ tab, float* __restrict__ mtab, float* __restrict__ vtab, const float* stW,
                     const float* stM, const float* stV, const float* G, float sign, const int64_t* ids, const float* ok,
                     float ssz, float bcs) {
        constexpr int Q0 = decltype(Q0c)::value, Q1 = decltype(Q1c)::value, NQQ = Q1 - Q0 > 0 ? Q1 - Q0 : 1;
        if (Q1 <= Q0) return;
        const int ln = fresh(lane);
        const float rbc = bcs;                       // cdr_adam_hp's bc2 as it is
        const int r0 = ln / LR, p = ln % LR;
        float4 w[NQQ], g[NQQ], m[NQQ], v[NQQ];
        int64_t o[NQQ];
        float okf[NQQ];
#pragma unroll
        for (int q = 0; q < Q1 - Q0; ++q) {
            const int row = (pw + 4 * (Q0 + q)) * RPI + r0, c = p ^ (row & (LR - 1)), so = (row * LR + p) * 4;
            w[q] = ld4(stW + so); g[q] = ld4(G + row * GS + 4 * c);
            m[q] = z4; v[q] = z4;
            if (adam) { m[q] = ld4(stM + so); v[q] = ld4(stV + so); }
            o[q] = ids[row] * D + 4 * c;
            okf[q] = ok[row];
        }
#pragma unroll
        for (int q = 0; q < Q1 - Q0; ++q) {
            float4 wn;
            wn.x = updq(w[q].x, sign * g[q].x, m[q].x, v[q].x, opt, ssz, rbc); wn.y = updq(w[q].y, sign * g[q].y, m[q].y, v[q].y, opt, ssz, rbc);
            wn.z = updq(w[q].z, sign * g[q].z, m[q].z, v[q].z, opt, ssz, rbc); wn.w = updq(w[q].w, sign * g[q].w, m[q].w, v[q].w, opt, ssz, rbc);
            if (okf[q] != 0.f) {
                MAP_ST4(tab + o[q], wn);
                if (adam) { MAP_ST4(mtab + o[q], m[q]); MAP_ST4(vtab + o[q], v[q]); }
            }
        }
    };
    using std::integral_constant;
#define IC(v) integral_constant<int, (v)>{}
    if (t < kRows) {                                                     // ids of this workgroup's first block
        const int64_t g = (int64_t)blockIdx.x * kRows + t;
        rid[0][t] = idx[g < n ? g : n - 1];
        rok[0][t] = g < n ? 1.f : 0.f;
    }
    __syncthreads();
    if (rowwave) {
        __builtin_amdgcn_s_setprio(3);
        row_offsets(rid[0], roff);
        stage(S, bSS, roff);
        stage(T, bST, roff);
        if (adam) { stage(mT, bSMT, roff); stage(vT, bSVT, roff); }
        vm_wait<0>();
    }
    lds_barrier();
    // weights that stay in registers across blocks: ALL of the dL/dS contraction's (it reads W down a column -- 4-byte strided
    // loads whose L2 latency under the step's HBM load no one-group look-ahead covers) and the first K group of the forward's
    float4 wf0[NJ][4], wb[NJ][D / 8];
    if (!rowwave) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int job = wave + 4 * jj;
#pragma unroll
            for (int j = 0; j < 4; ++j) wf0[jj][j] = job < NI ? ld4(W + (int64_t)(job * 32 + li0) * D + 8 * j + 4 * lh0) : z4;
#pragma unroll
            for (int s8 = 0; s8 < D / 8; ++s8) {
                wb[jj][s8] = z4;
                if (job < NI) {
                    const float* p = W + (int64_t)(8 * s8 + 4 * lh0) * D + job * 32 + li0;
                    wb[jj][s8] = make_float4(p[0], p[D], p[2 * D], p[3 * D]);
                }
            }
        }
    }
    int k = 0;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x, ++k) {
        const int par = k & 1, r3 = k % 3, r3n = (k + 1) % 3, r3p = (k + 2) % 3;
        const bool has_next = rb + gridDim.x < nrb;
        float* Xs = SS + par * kRows * D;                               // this block's source rows
        MP_STAMP(0);
        if (!rowwave) {
            // ---- forward: mapped = X W^T, one 32-column tile per job
            int64_t idn = 0; float okn = 0.f;
            if (t < kRows && has_next) { const int64_t g = (rb + gridDim.x) * kRows + t; idn = idx[g < n ? g : n - 1]; okn = g < n ? 1.f : 0.f; }
            f32x16 am[NJ];
            int li = fresh(li0), lh = fresh(lh0);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int job = wave + 4 * jj;
                am[jj] = zero16();
                if (job < NI) {
                    // K in groups of four steps: the weights of group g + 1 (L2) and the operand chunks of group g + 1 (LDS) are
                    // requested before group g's sixteen MFMAs; group 0's weights stay in registers across blocks (wf0)
                    const float* wm = W + (int64_t)(job * 32 + li) * D;
                    const float* xr = Xs + li * D;
                    const int sw = li & (LR - 1);
                    float4 nm[4], an[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { nm[j] = wf0[jj][j]; an[j] = ld4(xr + ((2 * j + lh) ^ sw) * 4); }
#pragma unroll
                    for (int g = 0; g < D / 32; ++g) {
                        float4 cm[4], ca[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { cm[j] = nm[j]; ca[j] = an[j]; }
                        if (g + 1 < D / 32) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                nm[j] = ld4(wm + 8 * (4 * (g + 1) + j) + 4 * lh);
                                an[j] = ld4(xr + ((2 * (4 * (g + 1) + j) + lh) ^ sw) * 4);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) { MFMA4(am[jj], ca[j], cm[j]); }
                    }
                }
            }
            if (t < kRows && has_next) { rid[r3n][t] = idn; rok[r3n][t] = okn; }
            MP_STAMP(1);
            lds_barrier();                                               // ---- X: the target rows of this block are in LDS
            MP_STAMP(2);
            li = fresh(li0); lh = fresh(lh0);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int job = wave + 4 * jj;
                if (job < NI) {
                    const int ncol = job * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float tv = ST[(row * LR + ((ncol >> 2) ^ (row & (LR - 1)))) * 4 + (ncol & 3)];
                        const float d = rok[r3][row] != 0.f ? am[jj][r] - tv : 0.f;              // nn.MSELoss (emcdr.py:81,162)
                        lsum += (double)d * (double)d;
                        GZ[row * GS + ncol] = gscale * d;
                    }
                }
            }
            MP_STAMP(3);
            lds_barrier();                                               // ---- F: gz is complete
            MP_STAMP(4);
            // ---- dW += gz^T x : this wave's tiles; the 32 operand words of tile q + 1 are requested before tile q's 16 MFMAs
            {
                float av[2][16], bv[2][16];
                auto fetch = [&](int q, float* a_, float* b_) {
                    const int tile = wave + 4 * q, mt = tile / NI, nt = tile - mt * NI;
                    const int l_i = fresh(li0), l_h = fresh(lh0);
                    const int m = mt * 32 + l_i, nn = nt * 32 + l_i;
                    // row = rc | (lh << 2) with rc = 8 (e >> 2) + (e & 3) a compile-time constant (bit 2 clear): the lane part of both
                    // addresses is formed once, each word costs one XOR (the swizzle) and a read with an immediate offset
                    const float* ga = GZ + (4 * l_h) * GS + m;
                    const float* xb = Xs + (4 * l_h) * D + (nn & 3);
                    const int cb = (nn >> 2) ^ (l_h << 2);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rc = 8 * (e >> 2) + (e & 3);
                        a_[e] = ga[rc * GS];
                        b_[e] = xb[rc * D + ((cb ^ (rc & (LR - 1))) << 2)];
                    }
                };
                if (wave < NI * NI) fetch(0, av[0], bv[0]);
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) {
                    if (wave + 4 * q < NI * NI) {
                        if (q + 1 < SLOTS && wave + 4 * (q + 1) < NI * NI) fetch(q + 1, av[(q + 1) & 1], bv[(q + 1) & 1]);
#pragma unroll
                        for (int e = 0; e < 16; ++e) MF1(wacc[q], av[q & 1][e], bv[q & 1][e]);
                    }
                }
            }
            MP_STAMP(5);
            lds_barrier();                                               // ---- M: the previous block's dL/dS has been consumed
            MP_STAMP(6);
            // ---- dL/dS = gz W  (K = the mapping's output dimension, in groups of four steps like the forward; W is read down a
            //      column here: four strided words per step)
            li = fresh(li0); lh = fresh(lh0);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int job = wave + 4 * jj;
                if (job < NI) {
                    const int ncol = job * 32 + li;
                    const float* ao = GZ + li * GS + 4 * lh;
                    f32x16 ag = zero16();
                    float4 an[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) an[j] = ld4(ao + 8 * j);
#pragma unroll
                    for (int g = 0; g < D / 32; ++g) {
                        float4 ca[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) ca[j] = an[j];
                        if (g + 1 < D / 32) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) an[j] = ld4(ao + 8 * (4 * (g + 1) + j));
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) { MFMA4(ag, ca[j], wb[jj][4 * g + j]); }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        GX[row * GS + ncol] = ag[r];
                    }
                }
            }
            MP_STAMP(7);
            lds_barrier();                                               // ---- E
            MP_STAMP(8);
        } else {
            vm_wait<0>();                                                // T, mT, vT of this block and mS, vS of the previous one
            MP_STAMP(1);
            if (k > 0) apply(IC(0), IC(NQ / 2), S, mS, vS, SS + (par ^ 1) * kRows * D, SMS, SVS, GX, 1.f, rid[r3p], rok[r3p], ss_s, bc_s);
            MP_STAMP(2);
            lds_barrier();                                               // ---- X
            if (k > 0) apply(IC(NQ / 2), IC(NQ), S, mS, vS, SS + (par ^ 1) * kRows * D, SMS, SVS, GX, 1.f, rid[r3p], rok[r3p], ss_s, bc_s);
            MP_STAMP(9);
            if (has_next) { row_offsets(rid[r3n], roff); stage(S, bSS + (unsigned)((par ^ 1) * kRows * D * 4), roff); }
            if (adam) { row_offsets(rid[r3], roff); stage(mS, bSMS, roff); stage(vS, bSVS, roff); }
            MP_STAMP(3);
            lds_barrier();                                               // ---- F
            MP_STAMP(4);
            apply(IC(0), IC(QT), T, mT, vT, ST, SMT, SVT, GZ, -1.f, rid[r3], rok[r3], ss_t, bc_t);   // dL/dT[id] = -dL/d mapped
            MP_STAMP(5);
            lds_barrier();                                               // ---- M
            apply(IC(QT), IC(NQ), T, mT, vT, ST, SMT, SVT, GZ, -1.f, rid[r3], rok[r3], ss_t, bc_t);
            MP_STAMP(10);
            if (has_next) {
                row_offsets(rid[r3n], roff);
                stage(T, bST, roff);
                if (adam) { stage(mT, bSMT, roff); stage(vT, bSVT, roff); vm_wait<5 * NQ>(); } else vm_wait<NQ>();
            }                                                            // (the next block's source rows have landed)
            MP_STAMP(7);
            lds_barrier();                                               // ---- E
            MP_STAMP(8);
        }
    }
    if (rowwave) {                                                       // the last block's source rows
        const int par = (k - 1) & 1, r3 = (k - 1) % 3;
        vm_wait<0>();
        apply(IC(0), IC(NQ), S, mS, vS, SS + par * kRows * D, SMS, SVS, GX, 1.f, rid[r3], rok[r3], ss_s, bc_s);
    } else {
        float* o = wpart + (size_t)blockIdx.x * ((size_t)net.ntiles * 1024 + net.nbias);
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int tile = wave + 4 * q;
            if (tile < NI * NI) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)tile * 1024 + r * 64 + lane] = wacc[q][r];
            }
        }
        lsum = wave_sum_d(lsum);
        if (lane == 0) red[wave] = lsum;
    }
    __syncthreads();
    if (t == 0) lpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- the same two wave groups for the reference's DEFAULT mapping: Linear(D, H) + Tanh + Linear(H, D) (emcdr.py:86-93;
// properties/model/EMCDR.yaml: non_linear, hidden 128), D and H in {64, 128} ----------------------------------------------------
// Twice the MFMA phases per block (two forwards, two weight-gradient phases, two data-gradient phases: seven barriers) around the same
// row-wave schedule.  LDS: one more operand buffer (the hidden activations) does not fit beside everything else, so gz = dL/d mapped and
// dL/dS SHARE a buffer -- gz is dead once the target rows are updated and the first layer's backward has run, dL/dS is dead once the
// source rows are updated, which the row waves finish before the next block's forward reaches its epilogue.  Weights come from L2
// (32 weight-gradient tiles take half of the register file).
template <int NI, int NHI>
__global__ __launch_bounds__(512, 1) void map_pipe2_kernel(map_net net, map_opt opt, float* __restrict__ S, float* __restrict__ mS,
                                                         float* __restrict__ vS, float* __restrict__ T, float* __restrict__ mT,
                                                         float* __restrict__ vT, const int64_t* __restrict__ idx, int64_t n,
                                                         const int64_t* __restrict__ step_s, const int64_t* __restrict__ step_t,
                                                         float* __restrict__ wpart, double* __restrict__ lpart, map_params bump) {
    constexpr int D = 32 * NI;                 // Ds == Dt
    constexpr int H = 32 * NHI;                // hidden width
    constexpr int HS = H + 4;
    constexpr int LR = D / 4;                  // 16-byte chunks (= DMA lanes) per row
    constexpr int RPI = 64 / LR;               // rows per DMA instruction (1 KiB of LDS)
    constexpr int GS = D + 4;                  // padded row stride of the two MFMA-written buffers
    constexpr int TL = NI * NHI;               // weight-gradient tiles per layer (layer 0: [H][D], layer 1: [D][H])
    constexpr int SL = TL / 4;                 // ... per MFMA wave and layer (TL is a multiple of 4)
    constexpr int NJD = (NI + 3) / 4, NJH = (NHI + 3) / 4;      // 32-column jobs per MFMA wave over D / over H
    constexpr int NQ = NI;                     // DMA instructions per row wave and array (4 row waves)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float rok[3][kRows];
    __shared__ int64_t rid[3][kRows];
    __shared__ double red[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li0 = lane & 31, lh0 = lane >> 5;
    const bool rowwave = wave >= 4;
    float* GZ = smem;                          // [32][GS]  mapped -> gz = dL/d mapped, later dL/dS (shared: see above)
    float* GX = GZ;
    float* HB = GZ + kRows * GS;               // [32][HS]  hidden activations -> dL/d(hidden pre-activation)
    float* SS = HB + kRows * HS;               // [2][32][D] source rows (swizzled)
    float* ST = SS + 2 * kRows * D;            // [32][D] each, swizzled alike
    float* SMT = ST + kRows * D;
    float* SVT = SMT + kRows * D;
    float* SMS = SVT + kRows * D;
    float* SVS = SMS + kRows * D;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool adam = opt.opt != 0;
    float ss_s, bc_s, ss_t, bc_t;
    adam_hp(opt, step_s, 1, ss_s, bc_s);
    adam_hp(opt, step_t, 1, ss_t, bc_t);
    const float gscale = 2.0f / ((float)n * (float)D);
    const float* __restrict__ W1 = net.W[0];   // [H][D]
    const float* __restrict__ W2 = net.W[1];   // [D][H]
    const float* __restrict__ B1 = net.b[0];
    const float* __restrict__ B2 = net.b[1];
    if (blockIdx.x == 0 && t == 0) {
        for (int l = 0; l < 2; ++l) { if (bump.sW[l]) bump.sW[l][0] += 1; if (bump.sb[l]) bump.sb[l][0] += 1; }
    }
    const int64_t nrb = (n + kRows - 1) / kRows;
    f32x16 wacc[2 * SL];                       // [0, SL): layer 0's tiles wave + 4 q ; [SL, 2 SL): layer 1's
#pragma unroll
    for (int q = 0; q < 2 * SL; ++q) wacc[q] = zero16();
    float bacc = 0.f;                          // MFMA-wave thread t owns flat bias element t (b1 then b2)
    double lsum = 0.0;

    // (address arithmetic below is re-derived per phase from a laundered lane id: left alone, LICM hoists ~200 loop-invariant LDS /
    //  global offsets out of the block loop and spills them -- and a spill reload inside a row wave is a vector memory load whose
    //  wait drains the DMA queue)
    auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };
    // ---- row waves: lane <-> (row, physical chunk) of DMA instruction i = pw + 4 q, q < NQ
    const int pw = wave - 4;
    // byte offsets of this lane's NQ chunks inside a table, for the rows of one block: all ids are read from LDS first (one wait),
    // and the result serves every table staged for that block
    auto row_offsets = [&](const int64_t* ids, int64_t (&off)[NQ]) {
        const int ln = fresh(lane);
        int64_t idv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) idv[q] = ids[(pw + 4 * q) * RPI + ln / LR];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = (pw + 4 * q) * RPI + ln / LR, p = ln % LR;
            off[q] = (idv[q] * D + 4 * (p ^ (row & (LR - 1)))) * (int64_t)sizeof(float);
        }
    };
    // NQ DMA instructions of one table in ONE asm statement: M0 saved once, set per instruction, restored once
    auto stage = [&](const float* __restrict__ tab, unsigned dst_bytes, const int64_t (&off)[NQ]) {
        const char* base = reinterpret_cast<const char*>(tab);
        const unsigned d0 = __builtin_amdgcn_readfirstlane(dst_bytes + (unsigned)pw * 1024u);   // instruction i = pw + 4 q writes 1 KiB at i * 1 KiB
        unsigned keep;
        if constexpr (NQ == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off" MAP_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(base + off[0]), "v"(base + off[1]), "v"(base + off[2]), "v"(base + off[3]),
                           "s"(d0), "s"(d0 + 4096u), "s"(d0 + 8192u), "s"(d0 + 12288u)
                         : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MAP_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(base + off[0]), "v"(base + off[1]), "s"(d0), "s"(d0 + 4096u) : "memory");
    };
    auto lds_off = [](const float* p) {
        return (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p);
    };
    const unsigned bSS = lds_off(SS), bST = lds_off(ST), bSMT = lds_off(SMT), bSVT = lds_off(SVT), bSMS = lds_off(SMS), bSVS = lds_off(SVS);
    int64_t roff[NQ];
    // Adam / SGD on this lane's chunks of DMA instructions [Q0, Q1): g = sign * G[row][logical chunk].  Every operand of the
    // whole range is requested before the arithmetic starts (one wave per role and SIMD: nothing else hides the LDS latency)
    auto apply = [&](auto Q0c, auto Q1c, float* __restrict__ tab, float* __restrict__ mtab, float* __restrict__ vtab, const float* stW,
                     const float* stM, const float* stV, const float* G, float sign, const int64_t* ids, const float* ok,
                     float ssz, float bcs) {
        constexpr int Q0 = decltype(Q0c)::value, Q1 = decltype(Q1c)::value, NQQ = Q1 - Q0 > 0 ? Q1 - Q0 : 1;
        if (Q1 <= Q0) return;
        const int ln = fresh(lane);
        const float rbc = bcs;                       // cdr_adam_hp's bc2 as it is
        const int r0 = ln / LR, p = ln % LR;
        float4 w[NQQ], g[NQQ], m[NQQ], v[NQQ];
        int64_t o[NQQ];
        float okf[NQQ];
#pragma unroll
        for (int q = 0; q < Q1 - Q0; ++q) {
            const int row = (pw + 4 * (Q0 + q)) * RPI + r0, c = p ^ (row & (LR - 1)), so = (row * LR + p) * 4;
            w[q] = ld4(stW + so); g[q] = ld4(G + row * GS + 4 * c);
            m[q] = z4; v[q] = z4;
            if (adam) { m[q] = ld4(stM + so); v[q] = ld4(stV + so); }
            o[q] = ids[row] * D + 4 * c;
            okf[q] = ok[row];
        }
#pragma unroll
        for (int q = 0; q < Q1 - Q0; ++q) {
            float4 wn;
            wn.x = updq(w[q].x, sign * g[q].x, m[q].x, v[q].x, opt, ssz, rbc); wn.y = updq(w[q].y, sign * g[q].y, m[q].y, v[q].y, opt, ssz, rbc);
            wn.z = updq(w[q].z, sign * g[q].z, m[q].z, v[q].z, opt, ssz, rbc); wn.w = updq(w[q].w, sign * g[q].w, m[q].w, v[q].w, opt, ssz, rbc);
            if (okf[q] != 0.f) {
                MAP_ST4(tab + o[q], wn);
                if (adam) { MAP_ST4(mtab + o[q], m[q]); MAP_ST4(vtab + o[q], v[q]); }
            }
        }
    };
    using std::integral_constant;
#define IC(v) integral_constant<int, (v)>{}
    if (t < kRows) {                                                     // ids of this workgroup's first block
        const int64_t g = (int64_t)blockIdx.x * kRows + t;
        rid[0][t] = idx[g < n ? g : n - 1];
        rok[0][t] = g < n ? 1.f : 0.f;
    }
    __syncthreads();
    if (rowwave) {
        __builtin_amdgcn_s_setprio(3);
        row_offsets(rid[0], roff);
        stage(S, bSS, roff);
        stage(T, bST, roff);
        if (adam) { stage(mT, bSMT, roff); stage(vT, bSVT, roff); }
        vm_wait<0>();
    }
    lds_barrier();
    int k = 0;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x, ++k) {
        const int par = k & 1, r3 = k % 3, r3n = (k + 1) % 3, r3p = (k + 2) % 3;
        const bool has_next = rb + gridDim.x < nrb;
        float* Xs = SS + par * kRows * D;                               // this block's source rows
        MP_STAMP(0);
        if (!rowwave) {
            int64_t idn = 0; float okn = 0.f;
            if (t < kRows && has_next) { const int64_t g = (rb + gridDim.x) * kRows + t; idn = idx[g < n ? g : n - 1]; okn = g < n ? 1.f : 0.f; }
            int li = fresh(li0), lh = fresh(lh0);
            // C[32 x 32-col tile] = A (LDS, one row per lane) x Wg[ncol][0..K) -- weight ROWS, K in groups of four steps, the next
            // group's weights (L2) and operand chunks (LDS) requested before the current group's sixteen MFMAs
            auto contract_rows = [&](auto Kc, const float* __restrict__ Wg, int ncol, auto a_chunk) {
                constexpr int K = decltype(Kc)::value;
                f32x16 acc = zero16();
                const float* wm = Wg + (int64_t)ncol * K + 4 * lh;
                float4 nm[4], an[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { nm[j] = ld4(wm + 8 * j); an[j] = a_chunk(2 * j + lh); }
#pragma unroll
                for (int g = 0; g < K / 32; ++g) {
                    float4 cm[4], ca[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { cm[j] = nm[j]; ca[j] = an[j]; }
                    if (g + 1 < K / 32) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { nm[j] = ld4(wm + 8 * (4 * (g + 1) + j)); an[j] = a_chunk(2 * (4 * (g + 1) + j) + lh); }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { MFMA4(acc, ca[j], cm[j]); }
                }
                return acc;
            };
            // C = A (LDS) x Wg[0..K)[ncol] -- weight COLUMNS (the data-gradient products): four strided words per K step
            auto contract_cols = [&](auto Kc, auto LDc, const float* __restrict__ Wg, int ncol, const float* ao) {
                constexpr int K = decltype(Kc)::value, LDW = decltype(LDc)::value;
                f32x16 acc = zero16();
                const float* w0 = Wg + (int64_t)(4 * lh) * LDW + ncol;
                float4 nb[4], an[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float* p = w0 + (int64_t)(8 * j) * LDW; nb[j] = make_float4(p[0], p[LDW], p[2 * LDW], p[3 * LDW]); an[j] = ld4(ao + 8 * j); }
#pragma unroll
                for (int g = 0; g < K / 32; ++g) {
                    float4 cb[4], ca[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { cb[j] = nb[j]; ca[j] = an[j]; }
                    if (g + 1 < K / 32) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int kk = 8 * (4 * (g + 1) + j);
                            const float* p = w0 + (int64_t)kk * LDW;
                            nb[j] = make_float4(p[0], p[LDW], p[2 * LDW], p[3 * LDW]); an[j] = ld4(ao + kk);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { MFMA4(acc, ca[j], cb[j]); }
                }
                return acc;
            };
            // ---- forward 0: HB = tanh(X W1^T + b1)
#pragma unroll
            for (int jj = 0; jj < NJH; ++jj) {
                const int job = wave + 4 * jj;
                if (job < NHI) {
                    const int ncol = job * 32 + li;
                    const float* xr = Xs + li * D;
                    const int sw = li & (LR - 1);
                    const f32x16 acc = contract_rows(IC(D), W1, ncol, [&](int ch) { return ld4(xr + ((ch ^ sw) << 2)); });
                    MP_STAMP(1);
                    const float bv = B1 ? B1[ncol] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) HB[((r & 3) + 8 * (r >> 2) + 4 * lh) * HS + ncol] = tanhf(acc[r] + bv);
                }
            }
            if (t < kRows && has_next) { rid[r3n][t] = idn; rok[r3n][t] = okn; }
            MP_STAMP(2);
            lds_barrier();                                               // ---- 1: hidden activations complete
            // ---- forward 1: mapped = HB W2^T + b2 (kept in registers across the barrier)
            li = fresh(li0); lh = fresh(lh0);
            f32x16 am[NJD];
#pragma unroll
            for (int jj = 0; jj < NJD; ++jj) {
                const int job = wave + 4 * jj;
                am[jj] = zero16();
                if (job < NI) {
                    const float* hr = HB + li * HS;
                    am[jj] = contract_rows(IC(H), W2, job * 32 + li, [&](int ch) { return ld4(hr + 4 * ch); });
                }
            }
            MP_STAMP(3);
            lds_barrier();                                               // ---- X: target rows landed; the previous dL/dS is consumed
            li = fresh(li0); lh = fresh(lh0);
#pragma unroll
            for (int jj = 0; jj < NJD; ++jj) {
                const int job = wave + 4 * jj;
                if (job < NI) {
                    const int ncol = job * 32 + li;
                    const float bv = B2 ? B2[ncol] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float tv = ST[(row * LR + ((ncol >> 2) ^ (row & (LR - 1)))) * 4 + (ncol & 3)];
                        const float d = rok[r3][row] != 0.f ? (am[jj][r] + bv) - tv : 0.f;       // nn.MSELoss (emcdr.py:81,162)
                        lsum += (double)d * (double)d;
                        GZ[row * GS + ncol] = gscale * d;
                    }
                }
            }
            MP_STAMP(4);
            lds_barrier();                                               // ---- F: gz complete
            // weight-gradient tiles of one layer: dW[m][nn] += sum_rows gz[row][m] in[row][nn]; the 32 operand words of a tile are
            // requested before its 16 MFMAs, the next tile's before them
            auto dw_layer = [&](auto QBc, auto NTWc, const float* gzb, int gzs, auto in_word) {
                constexpr int QB = decltype(QBc)::value, NTW = decltype(NTWc)::value;
                float av[2][16], bv[2][16];
                auto fetch = [&](int q, float* a_, float* b_) {
                    const int loc = wave + 4 * q, mt = loc / NTW, nt = loc - mt * NTW;
                    const int l_i = fresh(li0), l_h = fresh(lh0);
                    const float* ga = gzb + (4 * l_h) * gzs + mt * 32 + l_i;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rc = 8 * (e >> 2) + (e & 3);
                        a_[e] = ga[rc * gzs];
                        b_[e] = in_word(rc, l_h, nt * 32 + l_i);
                    }
                };
                fetch(0, av[0], bv[0]);
#pragma unroll
                for (int q = 0; q < SL; ++q) {
                    if (q + 1 < SL) fetch(q + 1, av[(q + 1) & 1], bv[(q + 1) & 1]);
#pragma unroll
                    for (int e = 0; e < 16; ++e) MF1(wacc[QB + q], av[q & 1][e], bv[q & 1][e]);
                }
            };
            // ---- layer 1: dW2 += gz^T HB ; db2 += column sums of gz
            dw_layer(IC(SL), IC(NHI), GZ, GS, [&](int rc, int l_h, int nn) { return HB[(rc + 4 * l_h) * HS + nn]; });
            if (B2 && t >= H && t < H + D) {
                float sum = 0.f;
                for (int row = 0; row < kRows; ++row) sum += GZ[row * GS + (t - H)];
                bacc += sum;
            }
            MP_STAMP(5);
            lds_barrier();                                               // ---- 4: HB may be overwritten
            // ---- dL/d hidden = gz W2, folded with tanh' = 1 - a^2, in place over HB
            li = fresh(li0); lh = fresh(lh0);
#pragma unroll
            for (int jj = 0; jj < NJH; ++jj) {
                const int job = wave + 4 * jj;
                if (job < NHI) {
                    const int ncol = job * 32 + li;
                    const f32x16 acc = contract_cols(IC(D), IC(H), W2, ncol, GZ + li * GS + 4 * lh);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ((r & 3) + 8 * (r >> 2) + 4 * lh) * HS + ncol;
                        const float a_ = HB[o];
                        HB[o] = acc[r] * (1.0f - a_ * a_);
                    }
                }
            }
            MP_STAMP(6);
            lds_barrier();                                               // ---- 5: gz0 complete
            // ---- layer 0: dW1 += gz0^T X ; db1 += column sums of gz0
            dw_layer(IC(0), IC(NI), HB, HS, [&](int rc, int l_h, int nn) {
                return Xs[(rc + 4 * l_h) * D + ((((nn >> 2) ^ (l_h << 2)) ^ (rc & (LR - 1))) << 2) + (nn & 3)];
            });
            if (B1 && t < H) {
                float sum = 0.f;
                for (int row = 0; row < kRows; ++row) sum += HB[row * HS + t];
                bacc += sum;
            }
            MP_STAMP(7);
            lds_barrier();                                               // ---- M: the target update has read gz for the last time
            // ---- dL/dS = gz0 W1 into the shared buffer
            li = fresh(li0); lh = fresh(lh0);
#pragma unroll
            for (int jj = 0; jj < NJD; ++jj) {
                const int job = wave + 4 * jj;
                if (job < NI) {
                    const int ncol = job * 32 + li;
                    const f32x16 acc = contract_cols(IC(H), IC(D), W1, ncol, HB + li * HS + 4 * lh);
#pragma unroll
                    for (int r = 0; r < 16; ++r) GX[((r & 3) + 8 * (r >> 2) + 4 * lh) * GS + ncol] = acc[r];
                }
            }
            MP_STAMP(8);
            lds_barrier();                                               // ---- E
            MP_STAMP(9);
        } else {
            vm_wait<0>();                                                // T, mT, vT of this block and mS, vS of the previous one
            MP_STAMP(1);
            if (k > 0) apply(IC(0), IC(NQ / 2), S, mS, vS, SS + (par ^ 1) * kRows * D, SMS, SVS, GX, 1.f, rid[r3p], rok[r3p], ss_s, bc_s);
            lds_barrier();                                               // ---- 1
            if (k > 0) apply(IC(NQ / 2), IC(NQ), S, mS, vS, SS + (par ^ 1) * kRows * D, SMS, SVS, GX, 1.f, rid[r3p], rok[r3p], ss_s, bc_s);
            if (has_next) { row_offsets(rid[r3n], roff); stage(S, bSS + (unsigned)((par ^ 1) * kRows * D * 4), roff); }
            if (adam) { row_offsets(rid[r3], roff); stage(mS, bSMS, roff); stage(vS, bSVS, roff); }
            MP_STAMP(2);
            lds_barrier();                                               // ---- X: dL/dS of the previous block is consumed
            MP_STAMP(3);
            lds_barrier();                                               // ---- F
            MP_STAMP(4);
            apply(IC(0), IC(NQ / 2), T, mT, vT, ST, SMT, SVT, GZ, -1.f, rid[r3], rok[r3], ss_t, bc_t);   // dL/dT[id] = -dL/d mapped
            lds_barrier();                                               // ---- 4
            apply(IC(NQ / 2), IC(NQ), T, mT, vT, ST, SMT, SVT, GZ, -1.f, rid[r3], rok[r3], ss_t, bc_t);
            if (has_next) {
                row_offsets(rid[r3n], roff);
                stage(T, bST, roff);
                if (adam) { stage(mT, bSMT, roff); stage(vT, bSVT, roff); }
            }
            lds_barrier();                                               // ---- 5
            MP_STAMP(5);
            lds_barrier();                                               // ---- M: gz has been read for the last time
            if (has_next) { if (adam) vm_wait<5 * NQ>(); else vm_wait<NQ>(); }   // the next block's source rows have landed
            MP_STAMP(7);
            lds_barrier();                                               // ---- E
            MP_STAMP(8);
        }
    }
    if (rowwave) {                                                       // the last block's source rows
        const int par = (k - 1) & 1, r3 = (k - 1) % 3;
        vm_wait<0>();
        apply(IC(0), IC(NQ), S, mS, vS, SS + par * kRows * D, SMS, SVS, GX, 1.f, rid[r3], rok[r3], ss_s, bc_s);
    } else {
        float* o = wpart + (size_t)blockIdx.x * ((size_t)net.ntiles * 1024 + net.nbias);
#pragma unroll
        for (int q = 0; q < 2 * SL; ++q) {
            const int tile = (q < SL ? 0 : TL) + wave + 4 * (q < SL ? q : q - SL);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)tile * 1024 + r * 64 + lane] = wacc[q][r];
        }
        if (t < net.nbias) o[(size_t)net.ntiles * 1024 + t] = bacc;
        lsum = wave_sum_d(lsum);
        if (lane == 0) red[wave] = lsum;
    }
    __syncthreads();
    if (t == 0) lpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- map_pipe2_kernel with (i) the weight-gradient accumulation in the ROW waves and (ii) every contraction's weight tile resident
// in registers one contraction ahead (round 3; the comment inside says what the stamps showed) -----------------------------------
template <int NI, int NHI>
__global__ __launch_bounds__(512, 1) void map_pipe3_kernel(map_net net, map_opt opt, float* __restrict__ S, float* __restrict__ mS,
                                                         float* __restrict__ vS, float* __restrict__ T, float* __restrict__ mT,
                                                         float* __restrict__ vT, const int64_t* __restrict__ idx, int64_t n,
                                                         const int64_t* __restrict__ step_s, const int64_t* __restrict__ step_t,
                                                         float* __restrict__ wpart, double* __restrict__ lpart, map_params bump) {
    constexpr int D = 32 * NI;                 // Ds == Dt
    constexpr int H = 32 * NHI;                // hidden width
    constexpr int HS = H + 4;
    constexpr int LR = D / 4;                  // 16-byte chunks (= DMA lanes) per row
    constexpr int RPI = 64 / LR;               // rows per DMA instruction (1 KiB of LDS)
    constexpr int GS = D + 4;                  // padded row stride of the two MFMA-written buffers
    constexpr int TL = NI * NHI;               // weight-gradient tiles per layer (layer 0: [H][D], layer 1: [D][H])
    constexpr int SL = TL / 4;                 // ... per MFMA wave and layer (TL is a multiple of 4)
    constexpr int NJD = (NI + 3) / 4, NJH = (NHI + 3) / 4;      // 32-column jobs per MFMA wave over D / over H
    constexpr int NQ = NI;                     // DMA instructions per row wave and array (4 row waves)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float rok[3][kRows];
    __shared__ int64_t rid[3][kRows];
    __shared__ double red[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li0 = lane & 31, lh0 = lane >> 5;
    const bool rowwave = wave >= 4;
    float* GZ = smem;                          // [32][GS]  mapped -> gz = dL/d mapped, later dL/dS (shared: see above)
    float* GX = GZ;
    float* HB = GZ + kRows * GS;               // [32][HS]  hidden activations -> dL/d(hidden pre-activation)
    float* SS = HB + kRows * HS;               // [2][32][D] source rows (swizzled)
    float* ST = SS + 2 * kRows * D;            // [32][D] each, swizzled alike
    float* SMT = ST + kRows * D;
    float* SVT = SMT + kRows * D;
    float* SMS = SVT + kRows * D;
    float* SVS = SMS + kRows * D;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool adam = opt.opt != 0;
    float ss_s, bc_s, ss_t, bc_t;
    adam_hp(opt, step_s, 1, ss_s, bc_s);
    adam_hp(opt, step_t, 1, ss_t, bc_t);
    const float gscale = 2.0f / ((float)n * (float)D);
    const float* __restrict__ W1 = net.W[0];   // [H][D]
    const float* __restrict__ W2 = net.W[1];   // [D][H]
    const float* __restrict__ B1 = net.b[0];
    const float* __restrict__ B2 = net.b[1];
    if (blockIdx.x == 0 && t == 0) {
        for (int l = 0; l < 2; ++l) { if (bump.sW[l]) bump.sW[l][0] += 1; if (bump.sb[l]) bump.sb[l][0] += 1; }
    }
    const int64_t nrb = (n + kRows - 1) / kRows;
    float bacc = 0.f;                          // MFMA-wave thread t owns flat bias element t (b1 then b2)
    double lsum = 0.0;

    // (address arithmetic below is re-derived per phase from a laundered lane id: left alone, LICM hoists ~200 loop-invariant LDS /
    //  global offsets out of the block loop and spills them -- and a spill reload inside a row wave is a vector memory load whose
    //  wait drains the DMA queue)
    auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };
    // ---- row waves: lane <-> (row, physical chunk) of DMA instruction i = pw + 4 q, q < NQ
    const int pw = wave - 4;
    // byte offsets of this lane's NQ chunks inside a table, for the rows of one block: all ids are read from LDS first (one wait),
    // and the result serves every table staged for that block
    auto row_offsets = [&](const int64_t* ids, int64_t (&off)[NQ]) {
        const int ln = fresh(lane);
        int64_t idv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) idv[q] = ids[(pw + 4 * q) * RPI + ln / LR];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = (pw + 4 * q) * RPI + ln / LR, p = ln % LR;
            off[q] = (idv[q] * D + 4 * (p ^ (row & (LR - 1)))) * (int64_t)sizeof(float);
        }
    };
    // NQ DMA instructions of one table in ONE asm statement: M0 saved once, set per instruction, restored once
    auto stage = [&](const float* __restrict__ tab, unsigned dst_bytes, const int64_t (&off)[NQ]) {
        const char* base = reinterpret_cast<const char*>(tab);
        const unsigned d0 = __builtin_amdgcn_readfirstlane(dst_bytes + (unsigned)pw * 1024u);   // instruction i = pw + 4 q writes 1 KiB at i * 1 KiB
        unsigned keep;
        if constexpr (NQ == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off" MAP_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(base + off[0]), "v"(base + off[1]), "v"(base + off[2]), "v"(base + off[3]),
                           "s"(d0), "s"(d0 + 4096u), "s"(d0 + 8192u), "s"(d0 + 12288u)
                         : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MAP_NT "\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MAP_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(base + off[0]), "v"(base + off[1]), "s"(d0), "s"(d0 + 4096u) : "memory");
    };
    auto lds_off = [](const float* p) {
        return (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p);
    };
    const unsigned bSS = lds_off(SS), bST = lds_off(ST), bSMT = lds_off(SMT), bSVT = lds_off(SVT), bSMS = lds_off(SMS), bSVS = lds_off(SVS);
    int64_t roff[NQ];
    // Adam / SGD on this lane's chunks of DMA instructions [Q0, Q1): g = sign * G[row][logical chunk].  Every operand of the
    // whole range is requested before the arithmetic starts (one wave per role and SIMD: nothing else hides the LDS latency)
    auto apply = [&](auto Q0c, auto Q1c, float* __restrict__ tab, float* __restrict__ mtab, float* __restrict__ vtab, const float* stW,
                     const float* stM, const float* stV, const float* G, float sign, const int64_t* ids, const float* ok,
                     float ssz, float bcs) {
        constexpr int Q0 = decltype(Q0c)::value, Q1 = decltype(Q1c)::value, NQQ = Q1 - Q0 > 0 ? Q1 - Q0 : 1;
        if (Q1 <= Q0) return;
        const int ln = fresh(lane);
        const float rbc = bcs;                       // cdr_adam_hp's bc2 as it is
        const int r0 = ln / LR, p = ln % LR;
        float4 w[NQQ], g[NQQ], m[NQQ], v[NQQ];
        int64_t o[NQQ];
        float okf[NQQ];
#pragma unroll
        for (int q = 0; q < Q1 - Q0; ++q) {
            const int row = (pw + 4 * (Q0 + q)) * RPI + r0, c = p ^ (row & (LR - 1)), so = (row * LR + p) * 4;
            w[q] = ld4(stW + so); g[q] = ld4(G + row * GS + 4 * c);
            m[q] = z4; v[q] = z4;
            if (adam) { m[q] = ld4(stM + so); v[q] = ld4(stV + so); }
            o[q] = ids[row] * D + 4 * c;
            okf[q] = ok[row];
        }
#pragma unroll
        for (int q = 0; q < Q1 - Q0; ++q) {
            float4 wn;
            wn.x = updq(w[q].x, sign * g[q].x, m[q].x, v[q].x, opt, ssz, rbc); wn.y = updq(w[q].y, sign * g[q].y, m[q].y, v[q].y, opt, ssz, rbc);
            wn.z = updq(w[q].z, sign * g[q].z, m[q].z, v[q].z, opt, ssz, rbc); wn.w = updq(w[q].w, sign * g[q].w, m[q].w, v[q].w, opt, ssz, rbc);
            if (okf[q] != 0.f) {
                MAP_ST4(tab + o[q], wn);
                if (adam) { MAP_ST4(mtab + o[q], m[q]); MAP_ST4(vtab + o[q], v[q]); }
            }
        }
    };
    using std::integral_constant;
#define IC(v) integral_constant<int, (v)>{}
    if (t < kRows) {                                                     // ids of this workgroup's first block
        const int64_t g = (int64_t)blockIdx.x * kRows + t;
        rid[0][t] = idx[g < n ? g : n - 1];
        rok[0][t] = g < n ? 1.f : 0.f;
    }
    __syncthreads();
    if (rowwave) {
        __builtin_amdgcn_s_setprio(3);
        row_offsets(rid[0], roff);
        stage(S, bSS, roff);
        stage(T, bST, roff);
        if (adam) { stage(mT, bSMT, roff); stage(vT, bSVT, roff); }
        vm_wait<0>();
    }
    lds_barrier();
    // ---- MFMA waves: weight tiles of their 32-column job, one contraction AHEAD.  In map_pipe2_kernel the weights stream from L2
    // one K group (0.45 us of MFMAs) ahead of their use, behind the row waves' DMA traffic in the CU's vector-memory path: stamps of
    // one block show the four 64-MFMA contractions at 4.5 / 6.2 / 6.5 / 12 us where the matrix pipe needs 1.8-2 us each.  With the
    // weight-gradient accumulators (128 registers) moved to the row waves, two 64-register tile buffers fit: the tile of the NEXT
    // contraction is requested before the current one starts and has a whole phase (and a barrier) to arrive.
    const int job = wave;                                                 // one 32-column tile per MFMA wave and contraction
    auto load_rows = [&](auto Kc, const float* __restrict__ Wg, int ncol, float4 (&T)[16]) {      // T[i] = Wg[ncol][8 i + 4 lh ..]
        constexpr int K = decltype(Kc)::value;
        const float* wm = Wg + (int64_t)ncol * K + 4 * fresh(lh0);
#pragma unroll
        for (int i = 0; i < K / 8; ++i) T[i] = ld4(wm + 8 * i);
    };
    auto load_cols = [&](auto Kc, auto LDc, const float* __restrict__ Wg, int ncol, float4 (&T)[16]) {   // T[i] = Wg[8 i + 4 lh + 0..3][ncol]
        constexpr int K = decltype(Kc)::value, LDW = decltype(LDc)::value;
        const float* w0 = Wg + (int64_t)(4 * fresh(lh0)) * LDW + ncol;
#pragma unroll
        for (int i = 0; i < K / 8; ++i) { const float* q_ = w0 + (int64_t)(8 * i) * LDW; T[i] = make_float4(q_[0], q_[LDW], q_[2 * LDW], q_[3 * LDW]); }
    };
    // 64 (K = 128) MFMAs on a resident tile; the A chunks of the next group of four K steps are read from LDS under the current group
    auto contract = [&](auto Kc, const float4 (&T)[16], auto a_chunk) {
        constexpr int K = decltype(Kc)::value;
        f32x16 acc = zero16();
        float4 an[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) an[j] = a_chunk(j);
#pragma unroll
        for (int g = 0; g < K / 32; ++g) {
            float4 ca[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ca[j] = an[j];
            if (g + 1 < K / 32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) an[j] = a_chunk(4 * (g + 1) + j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { MFMA4(acc, ca[j], T[4 * g + j]); }
        }
        return acc;
    };
    // weight-gradient tiles of one layer (ROW waves): dW[m][nn] += sum_rows gz[row][m] in[row][nn]
    auto dw_layer = [&](f32x16 (&wacc)[2 * SL], auto QBc, auto NTWc, const float* gzb, int gzs, auto in_word) {
        constexpr int QB = decltype(QBc)::value, NTW = decltype(NTWc)::value;
        float av[2][16], bv[2][16];
        auto fetch = [&](int q, float* a_, float* b_) {
            const int loc = pw + 4 * q, mt = loc / NTW, nt = loc - mt * NTW;
            const int l_i = fresh(li0), l_h = fresh(lh0);
            const float* ga = gzb + (4 * l_h) * gzs + mt * 32 + l_i;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rc = 8 * (e >> 2) + (e & 3);
                a_[e] = ga[rc * gzs];
                b_[e] = in_word(rc, l_h, nt * 32 + l_i);
            }
        };
        fetch(0, av[0], bv[0]);
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            if (q + 1 < SL) fetch(q + 1, av[(q + 1) & 1], bv[(q + 1) & 1]);
#pragma unroll
            for (int e = 0; e < 16; ++e) MF1(wacc[QB + q], av[q & 1][e], bv[q & 1][e]);
        }
    };
    // Each role runs its OWN block loop (same barriers per block in both): the row waves' 128 accumulator registers and the MFMA
    // waves' two 64-register weight tiles are then never live in the same code, which one loop with a branch inside made them
    // (239 registers spilled to scratch in that form).
    float* o = wpart + (size_t)blockIdx.x * ((size_t)net.ntiles * 1024 + net.nbias);
    if (!rowwave) {
        float4 TA[16], TB[16];
        if (job < NHI) load_rows(IC(D), W1, job * 32 + li0, TA);         // the first block's forward-0 tile
        int k = 0;
        for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x, ++k) {
            const int par = k & 1, r3 = k % 3, r3n = (k + 1) % 3, r3p = (k + 2) % 3;
            const bool has_next = rb + gridDim.x < nrb;
            float* Xs = SS + par * kRows * D;                               // this block's source rows
            MP_STAMP(0);
            int64_t idn = 0; float okn = 0.f;
            if (t < kRows && has_next) { const int64_t g = (rb + gridDim.x) * kRows + t; idn = idx[g < n ? g : n - 1]; okn = g < n ? 1.f : 0.f; }
            int li = fresh(li0), lh = fresh(lh0);
            // ---- forward 0: HB = tanh(X W1^T + b1)          [TA = W1 rows ; request TB = W2 rows]
            if (job < NI) load_rows(IC(H), W2, job * 32 + li, TB);
            __builtin_amdgcn_sched_barrier(0);
            if (job < NHI) {
                const int ncol = job * 32 + li;
                const float* xr = Xs + li * D;
                const int sw = li & (LR - 1);
                const f32x16 acc = contract(IC(D), TA, [&](int i) { return ld4(xr + (((2 * i + lh) ^ sw) << 2)); });
                MP_STAMP(1);
                const float bv = B1 ? B1[ncol] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) HB[((r & 3) + 8 * (r >> 2) + 4 * lh) * HS + ncol] = tanhf(acc[r] + bv);
            }
            if (t < kRows && has_next) { rid[r3n][t] = idn; rok[r3n][t] = okn; }
            MP_STAMP(2);
            lds_barrier();                                               // ---- 1: hidden activations complete
            // ---- forward 1: mapped = HB W2^T + b2            [TB = W2 rows ; request TA = W2 columns]
            li = fresh(li0); lh = fresh(lh0);
            if (job < NHI) load_cols(IC(D), IC(H), W2, job * 32 + li, TA);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 am = zero16();
            if (job < NI) {
                const float* hr = HB + li * HS;
                am = contract(IC(H), TB, [&](int i) { return ld4(hr + 4 * (2 * i + lh)); });
            }
            MP_STAMP(3);
            lds_barrier();                                               // ---- X: target rows landed; the previous dL/dS is consumed
            li = fresh(li0); lh = fresh(lh0);
            if (job < NI) {
                const int ncol = job * 32 + li;
                const float bv = B2 ? B2[ncol] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float tv = ST[(row * LR + ((ncol >> 2) ^ (row & (LR - 1)))) * 4 + (ncol & 3)];
                    const float d = rok[r3][row] != 0.f ? (am[r] + bv) - tv : 0.f;              // nn.MSELoss (emcdr.py:81,162)
                    lsum += (double)d * (double)d;
                    GZ[row * GS + ncol] = gscale * d;
                }
            }
            MP_STAMP(4);
            lds_barrier();                                               // ---- F: gz complete
            // ---- dL/d hidden = gz W2: the contraction now (the row waves are still reading HB for dW2), its in-place epilogue after 4
            //      [TA = W2 columns ; request TB = W1 columns]
            li = fresh(li0); lh = fresh(lh0);
            if (job < NI) load_cols(IC(H), IC(D), W1, job * 32 + li, TB);
            __builtin_amdgcn_sched_barrier(0);
            if (B2 && t >= H && t < H + D) {
                float sum = 0.f;
                for (int row = 0; row < kRows; ++row) sum += GZ[row * GS + (t - H)];
                bacc += sum;
            }
            f32x16 ah = zero16();
            if (job < NHI) {
                const float* ao = GZ + li * GS + 4 * lh;
                ah = contract(IC(D), TA, [&](int i) { return ld4(ao + 8 * i); });
            }
            MP_STAMP(5);
            lds_barrier();                                               // ---- 4: HB may be overwritten
            li = fresh(li0); lh = fresh(lh0);
            if (job < NHI) {
                const int ncol = job * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ((r & 3) + 8 * (r >> 2) + 4 * lh) * HS + ncol;
                    const float a_ = HB[o];
                    HB[o] = ah[r] * (1.0f - a_ * a_);                      // tanh' = 1 - a^2
                }
            }
            MP_STAMP(6);
            lds_barrier();                                               // ---- 5: gz0 complete
            // ---- dL/dS = gz0 W1: contraction now, written to the shared buffer after M  [TB = W1 columns ; request TA = W1 rows]
            li = fresh(li0); lh = fresh(lh0);
            if (job < NHI) load_rows(IC(D), W1, job * 32 + li, TA);     // the next block's forward 0 (or nobody's)
            __builtin_amdgcn_sched_barrier(0);
            if (B1 && t < H) {
                float sum = 0.f;
                for (int row = 0; row < kRows; ++row) sum += HB[row * HS + t];
                bacc += sum;
            }
            f32x16 ax = zero16();
            if (job < NI) {
                const float* ao = HB + li * HS + 4 * lh;
                ax = contract(IC(H), TB, [&](int i) { return ld4(ao + 8 * i); });
            }
            MP_STAMP(7);
            lds_barrier();                                               // ---- M: the target update has read gz for the last time
            li = fresh(li0); lh = fresh(lh0);
            if (job < NI) {
                const int ncol = job * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) GX[((r & 3) + 8 * (r >> 2) + 4 * lh) * GS + ncol] = ax[r];
            }
            MP_STAMP(8);
            lds_barrier();                                               // ---- E
            MP_STAMP(9);
        }
        if (t < net.nbias) o[(size_t)net.ntiles * 1024 + t] = bacc;
        lsum = wave_sum_d(lsum);
        if (lane == 0) red[wave] = lsum;
    } else {
        f32x16 wacc[2 * SL];                       // [0, SL): layer 0's tiles pw + 4 q ; [SL, 2 SL): layer 1's
#pragma unroll
        for (int q = 0; q < 2 * SL; ++q) wacc[q] = zero16();
        int k = 0;
        for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x, ++k) {
            const int par = k & 1, r3 = k % 3, r3n = (k + 1) % 3, r3p = (k + 2) % 3;
            const bool has_next = rb + gridDim.x < nrb;
            float* Xs = SS + par * kRows * D;                               // this block's source rows
            MP_STAMP(0);
            vm_wait<0>();                                                // T, mT, vT of this block and mS, vS of the previous one
            MP_STAMP(1);
            if (k > 0) apply(IC(0), IC(NQ / 2), S, mS, vS, SS + (par ^ 1) * kRows * D, SMS, SVS, GX, 1.f, rid[r3p], rok[r3p], ss_s, bc_s);
            lds_barrier();                                               // ---- 1
            if (k > 0) apply(IC(NQ / 2), IC(NQ), S, mS, vS, SS + (par ^ 1) * kRows * D, SMS, SVS, GX, 1.f, rid[r3p], rok[r3p], ss_s, bc_s);
            if (has_next) { row_offsets(rid[r3n], roff); stage(S, bSS + (unsigned)((par ^ 1) * kRows * D * 4), roff); }
            if (adam) { row_offsets(rid[r3], roff); stage(mS, bSMS, roff); stage(vS, bSVS, roff); }
            MP_STAMP(2);
            lds_barrier();                                               // ---- X: dL/dS of the previous block is consumed
            MP_STAMP(3);
            lds_barrier();                                               // ---- F
            MP_STAMP(4);
            apply(IC(0), IC(NQ / 2), T, mT, vT, ST, SMT, SVT, GZ, -1.f, rid[r3], rok[r3], ss_t, bc_t);   // dL/dT[id] = -dL/d mapped
            // ---- layer 1: dW2 += gz^T HB (HB is overwritten after barrier 4)
            dw_layer(wacc, IC(SL), IC(NHI), GZ, GS, [&](int rc, int l_h, int nn) { return HB[(rc + 4 * l_h) * HS + nn]; });
            lds_barrier();                                               // ---- 4
            apply(IC(NQ / 2), IC(NQ), T, mT, vT, ST, SMT, SVT, GZ, -1.f, rid[r3], rok[r3], ss_t, bc_t);
            if (has_next) {
                row_offsets(rid[r3n], roff);
                stage(T, bST, roff);
                if (adam) { stage(mT, bSMT, roff); stage(vT, bSVT, roff); }
            }
            lds_barrier();                                               // ---- 5
            MP_STAMP(5);
            // ---- layer 0: dW1 += gz0^T X
            dw_layer(wacc, IC(0), IC(NI), HB, HS, [&](int rc, int l_h, int nn) {
                return Xs[(rc + 4 * l_h) * D + ((((nn >> 2) ^ (l_h << 2)) ^ (rc & (LR - 1))) << 2) + (nn & 3)];
            });
            lds_barrier();                                               // ---- M: gz has been read for the last time
            if (has_next) { if (adam) vm_wait<5 * NQ>(); else vm_wait<NQ>(); }   // the next block's source rows have landed
            MP_STAMP(7);
            lds_barrier();                                               // ---- E
            MP_STAMP(8);
        }
        {                                                                // the last block's source rows, then this wave's weight-gradient tiles
            const int par = (k - 1) & 1, r3 = (k - 1) % 3;
            vm_wait<0>();
            apply(IC(0), IC(NQ), S, mS, vS, SS + par * kRows * D, SMS, SVS, GX, 1.f, rid[r3], rok[r3], ss_s, bc_s);
#pragma unroll
            for (int q = 0; q < 2 * SL; ++q) {
                const int tile = (q < SL ? 0 : TL) + pw + 4 * (q < SL ? q : q - SL);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)tile * 1024 + r * 64 + lane] = wacc[q][r];
            }
        }
    }
    __syncthreads();
    if (t == 0) lpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void map_finish_kernel(map_net net, map_opt opt, map_params P, const float* __restrict__ wpart,
                                                         const double* __restrict__ lpart, int nwg, int64_t n, float* __restrict__ loss_out,
                                                         int64_t* step_s, int64_t* step_t) {
    const size_t pstride = (size_t)net.ntiles * 1024 + net.nbias;
    if (blockIdx.x == gridDim.x - 1) {                          // last block: the loss (workgroup order) and the tables' counters
        __shared__ double red[4];
        double s[1] = {0.0};
        if (threadIdx.x < 256) for (int b = threadIdx.x; b < nwg; b += 256) s[0] += lpart[b];
        s[0] = wave_sum_d(s[0]);
        if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s[0];
        __syncthreads();
        if (threadIdx.x == 0) s[0] = (red[0] + red[1]) + (red[2] + red[3]);
        if (threadIdx.x == 0) {
            loss_out[0] = (float)(s[0] / ((double)n * (double)net.dims[net.L]));
            if (step_s) step_s[0] += 1;
            if (step_t) step_t[0] += 1;
        }
        return;
    }
    // 64 parameter elements per block x SIXTEEN partial sums each: thread (sub, el) adds the partials of workgroups sub, sub + 16,
    // ... (coalesced across el, independent across sub), then the sixteen sums are added in order through LDS -- deterministic, and
    // 16x shorter than one thread walking every workgroup's partial
    __shared__ float part[16][64];
    const int sub = threadIdx.x >> 6, el = threadIdx.x & 63;
    int64_t e = (int64_t)blockIdx.x * 64 + el;
    int l = 0, isb = 0;
    int64_t base = 0;
    bool found = false;
    for (int q = 0; q < net.L && !found; ++q) {
        const int64_t nw = (int64_t)net.dims[q] * net.dims[q + 1];
        if (e < base + nw) { l = q; isb = 0; e -= base; found = true; break; }
        base += nw;
        const int64_t nb = net.b[q] ? net.dims[q + 1] : 0;
        if (e < base + nb) { l = q; isb = 1; e -= base; found = true; break; }
        base += nb;
    }
    const int din = net.dims[l], dout = net.dims[l + 1];
    size_t off = 0;
    if (found) {
        if (!isb) {
            const int m = (int)(e / din), nn = (int)(e - (int64_t)m * din);
            const int NTW = (din + 31) >> 5;
            const int tile = net.tile_off[l] + (m >> 5) * NTW + (nn >> 5);
            const int mm = m & 31, h = (mm >> 2) & 1, r = (mm & 3) + 4 * (mm >> 3);
            off = (size_t)tile * 1024 + r * 64 + (nn & 31) + 32 * h;
        } else {
            off = (size_t)net.ntiles * 1024 + net.bias_off[l] + e;
        }
    }
    float g = 0.f;
    if (found)
        for (int b = sub; b < nwg; b += 16) g += wpart[(size_t)b * pstride + off];
    part[sub][el] = g;
    __syncthreads();
    if (!found || sub != 0) return;
    g = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) g += part[j][el];
    (void)dout;
    float* p = isb ? P.b[l] : P.W[l];
    float* mp = isb ? P.mb[l] : P.mW[l];
    float* vp = isb ? P.vb[l] : P.vW[l];
    const int64_t* sp = isb ? P.sb[l] : P.sW[l];
    float ss, bc;
    adam_hp(opt, sp, 0, ss, bc);                               // launch 1 already advanced the parameters' counters
    float m = 0.f, v = 0.f;
    if (opt.opt) { m = mp[e]; v = vp[e]; }
    p[e] = upd1(p[e], g, m, v, opt, ss, bc);
    if (opt.opt) { mp[e] = m; vp[e] = v; }
}

int fill(map_net& net, int L, const int* dims, const int* acts, const float* const* W, const float* const* b, size_t* lds_bytes,
         int* gx_off, int* t_off) {
    if (L < 1 || L > kMapMaxL || !dims || !W) return 0;
    memset(&net, 0, sizeof(net));
    net.L = L; net.vec = 1;
    int tiles = 0, nb = 0, off = 0;
    for (int l = 0; l <= L; ++l) { if (dims[l] <= 0 || (dims[l] & 3) || dims[l] > 1024) return 0; net.dims[l] = dims[l]; }
    for (int l = 0; l < L; ++l) {
        net.act[l] = acts ? acts[l] : CDR_ACT_NONE;
        if (net.act[l] != CDR_ACT_NONE && net.act[l] != CDR_ACT_TANH) return 0;
        net.W[l] = W[l]; net.b[l] = b ? b[l] : nullptr;
        if (!net.W[l]) return 0;
        if ((uintptr_t)net.W[l] & 15) net.vec = 0;
        net.tile_off[l] = tiles; tiles += ((dims[l + 1] + 31) / 32) * ((dims[l] + 31) / 32);
        net.bias_off[l] = nb; if (net.b[l]) nb += dims[l + 1];
    }
    if (net.act[L - 1] != CDR_ACT_NONE) return 0;               // the reference's mapping ends in a plain Linear (emcdr.py:86-93)
    net.tile_off[L] = tiles; net.bias_off[L] = nb;
    net.ntiles = tiles; net.nbias = nb;
    if (tiles > 4 * kSlots || nb > 512) return 0;
    for (int l = 0; l <= L; ++l) { net.buf_off[l] = off; off += kRows * (dims[l] + 4); }     // X, hidden activations, mapped / gz
    *gx_off = off; off += kRows * (dims[0] + 4);
    *t_off = off;                                             // (no staging buffer for the target rows)
    *lds_bytes = (size_t)off * sizeof(float);
    return *lds_bytes <= 150 * 1024;
}

inline int wg_count(int64_t n, int per_cu = 2) {
    int64_t g = (n + kRows - 1) / kRows;
    if (g > (int64_t)per_cu * CDR_NUM_CU) g = (int64_t)per_cu * CDR_NUM_CU;     // resident workgroups: each one walks several 32-id blocks
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

#ifdef CDR_MAP_PROF
extern "C" int cdr_map_prof_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_map_prof), sizeof(long long) * 64); }
#endif

extern "C" int cdr_map_step_plan(int L, const int* dims, const int* has_bias, int64_t n, size_t* workspace_bytes) {
    CDR_CHECK_ARG(dims && workspace_bytes && n > 0);
    map_net net;
    const float* W[kMapMaxL]; const float* b[kMapMaxL];
    for (int l = 0; l < kMapMaxL; ++l) { W[l] = (const float*)16; b[l] = (has_bias && l < L && has_bias[l]) ? (const float*)16 : nullptr; }
    size_t lds; int gx, to;
    if (!fill(net, L, dims, nullptr, W, b, &lds, &gx, &to)) { cdr_set_error("cdr_map_step_plan: unsupported mapping shape"); return CDR_EINVAL; }
    const int nwg = wg_count(n);
    *workspace_bytes = (size_t)nwg * ((size_t)net.ntiles * 1024 + net.nbias) * sizeof(float) + 256 + (size_t)nwg * sizeof(double);
    return CDR_OK;
}

extern "C" int cdr_map_step_unique(cdr_ctx* ctx, void* stream, int opt, float* src_tab, float* src_m, float* src_v, float* tgt_tab,
                                   float* tgt_m, float* tgt_v, const int64_t* idx, int64_t n, int L, const int* dims, const int* acts,
                                   float* const* W, float* const* bias, float* const* mW, float* const* vW, float* const* mb,
                                   float* const* vb, int64_t* const* step_W, int64_t* const* step_b, int64_t* step_src_dev,
                                   int64_t* step_tgt_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   float* loss_out, void* workspace, size_t workspace_bytes) {
    CDR_CHECK_ARG(ctx && src_tab && tgt_tab && idx && n > 0 && W && loss_out && workspace);
    CDR_CHECK_ARG(opt == 0 || (opt == 1 && src_m && src_v && tgt_m && tgt_v && mW && vW && step_W && step_src_dev && step_tgt_dev));
    map_net net;
    size_t lds; int gx, to;
    const float* Wc[kMapMaxL]; const float* bc[kMapMaxL];
    for (int l = 0; l < kMapMaxL; ++l) { Wc[l] = l < L ? W[l] : nullptr; bc[l] = (bias && l < L) ? bias[l] : nullptr; }
    if (!fill(net, L, dims, acts, Wc, bc, &lds, &gx, &to)) { cdr_set_error("cdr_map_step_unique: unsupported mapping shape"); return CDR_EINVAL; }
    map_params P;
    memset(&P, 0, sizeof(P));
    for (int l = 0; l < L; ++l) {
        P.W[l] = W[l]; P.b[l] = bias ? bias[l] : nullptr;
        if (opt == 1) {
            P.mW[l] = mW[l]; P.vW[l] = vW[l]; P.sW[l] = step_W[l];
            CDR_CHECK_ARG(P.mW[l] && P.vW[l] && P.sW[l]);
            if (P.b[l]) { CDR_CHECK_ARG(mb && vb && step_b && mb[l] && vb[l] && step_b[l]); P.mb[l] = mb[l]; P.vb[l] = vb[l]; P.sb[l] = step_b[l]; }
        }
    }
    // the two-wave-group kernel for the shape it is written for (linear mapping without bias, Ds == Dt in {64, 128}, 16-B aligned
    // weights); everything else takes the two-workgroups-per-CU kernel
    const int Dp = dims[0];
    const bool pipe = L == 1 && !net.b[0] && net.vec && dims[1] == Dp && (Dp == 64 || Dp == 128);
    // ... and its two-layer form for the default tanh MLP: D -> H -> D, D and H in {64, 128}, biases on both layers or on neither
    const int Hp = L == 2 ? dims[1] : 0;
    const bool pipe2 = L == 2 && net.vec && net.act[0] == CDR_ACT_TANH && dims[2] == Dp && (Dp == 64 || Dp == 128) &&
                       (Hp == 64 || Hp == 128) && ((net.b[0] != nullptr) == (net.b[1] != nullptr));
    const int nwg = (pipe || pipe2) ? wg_count(n, 1) : wg_count(n, 2);        // (three resident workgroups per CU measured no faster)
    const size_t wbytes = (size_t)nwg * ((size_t)net.ntiles * 1024 + net.nbias) * sizeof(float);
    const size_t woff = (wbytes + 255) & ~(size_t)255;
    CDR_CHECK_ARG(workspace_bytes >= woff + (size_t)nwg * sizeof(double));
    float* wpart = (float*)workspace;
    double* lpart = (double*)((char*)workspace + woff);
    const bool few = net.ntiles <= 16;
    const void* fn = few ? (const void*)map_step_kernel<4> : (const void*)map_step_kernel<8>;
    if (pipe) {
        lds = ((size_t)2 * kRows * (Dp + 4) + 7 * (size_t)kRows * Dp) * sizeof(float);
        fn = Dp == 128 ? (const void*)map_pipe_kernel<4> : (const void*)map_pipe_kernel<2>;
    }
    if (pipe2) {
        lds = ((size_t)kRows * (Dp + 4) + (size_t)kRows * (Hp + 4) + 7 * (size_t)kRows * Dp) * sizeof(float);
        fn = Dp == 128 ? (Hp == 128 ? (const void*)map_pipe2_kernel<4, 4> : (const void*)map_pipe2_kernel<4, 2>)
                       : (Hp == 128 ? (const void*)map_pipe2_kernel<2, 4> : (const void*)map_pipe2_kernel<2, 2>);
        const void* fn3 = Dp == 128 ? (Hp == 128 ? (const void*)map_pipe3_kernel<4, 4> : (const void*)map_pipe3_kernel<4, 2>)
                                    : (Hp == 128 ? (const void*)map_pipe3_kernel<2, 4> : (const void*)map_pipe3_kernel<2, 2>);
        if (lds > 64 * 1024) {
            hipError_t e3 = hipFuncSetAttribute(fn3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e3 != hipSuccess) { cdr_set_error("cdr_map_step_unique: %zu B of LDS refused: %s", lds, hipGetErrorString(e3)); return (int)e3; }
        }
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { cdr_set_error("cdr_map_step_unique: %zu B of LDS refused: %s", lds, hipGetErrorString(e)); return (int)e; }
    }
    const map_opt mo{lr, beta1, beta2, eps, weight_decay, opt};
    hipStream_t s = (hipStream_t)stream;
    {
        cdr_time_scope ts(ctx, CDR_TAG_MAP_STEP, s);
#define MS_ARGS net, mo, src_tab, src_m, src_v, tgt_tab, tgt_m, tgt_v, idx, n, step_src_dev, step_tgt_dev, gx, to, wpart, lpart, P
#define MP_ARGS net, mo, src_tab, src_m, src_v, tgt_tab, tgt_m, tgt_v, idx, n, step_src_dev, step_tgt_dev, wpart, lpart, P
        static const bool old_pipe2 = [] { const char* e = getenv("CDR_MAP_PIPE2"); return e && e[0] == '1'; }();     // A/B switch (tools/)
        if (pipe2 && !old_pipe2 && Dp == 128 && Hp == 128) map_pipe3_kernel<4, 4><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe2 && !old_pipe2 && Dp == 128) map_pipe3_kernel<4, 2><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe2 && !old_pipe2 && Hp == 128) map_pipe3_kernel<2, 4><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe2 && !old_pipe2) map_pipe3_kernel<2, 2><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe2 && Dp == 128 && Hp == 128) map_pipe2_kernel<4, 4><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe2 && Dp == 128) map_pipe2_kernel<4, 2><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe2 && Hp == 128) map_pipe2_kernel<2, 4><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe2) map_pipe2_kernel<2, 2><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe && Dp == 128) map_pipe_kernel<4><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (pipe) map_pipe_kernel<2><<<dim3(nwg), dim3(512), lds, s>>>(MP_ARGS);
        else if (few) map_step_kernel<4><<<dim3(nwg), dim3(256), lds, s>>>(MS_ARGS);
        else map_step_kernel<8><<<dim3(nwg), dim3(256), lds, s>>>(MS_ARGS);
#undef MP_ARGS
#undef MS_ARGS
    }
    CDR_LAUNCH_CHECK();
    int64_t elems = 0;
    for (int l = 0; l < L; ++l) elems += (int64_t)dims[l] * dims[l + 1] + (net.b[l] ? dims[l + 1] : 0);
    map_finish_kernel<<<dim3((unsigned)((elems + 63) / 64 + 1)), dim3(1024), 0, s>>>(net, mo, P, wpart, lpart, nwg, n, loss_out,
                                                                                      opt == 1 ? step_src_dev : nullptr,
                                                                                      opt == 1 ? step_tgt_dev : nullptr);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
