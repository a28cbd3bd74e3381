// Shared device/host helpers for libcdrhip (gfx950 only: wave = 64 lanes, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cdr_hip.h"

#define CDR_WAVE 64
#define CDR_NUM_CU 256
#define CDR_MAX_PARTIAL_BLOCKS 4096          // grid cap of every two-pass reduction
#define CDR_PARTIAL_STRIDE 8                 // doubles per block
#define CDR_TICKETS 1024

struct cdr_ctx {
    int device;
    double* partials;                        // [CDR_MAX_PARTIAL_BLOCKS][CDR_PARTIAL_STRIDE]
    unsigned* tickets;                       // [CDR_TICKETS] zero-initialised sign-in counters ('the block that signs in last finishes'); atomicInc wraps them back to 0
    void* scratch;                           // grow-on-demand device scratch (long-segment partial sums of cdr_rowwise_apply)
    size_t scratch_bytes;
    void* scrub_ptr;                         // cdr_ctx_scrub_next: a region the NEXT forward launch on this context zero-fills on the side
    size_t scrub_bytes;
    // cdr_ctx_set_id_counters: per-row occurrence counters of the NEXT fused BPR steps' two tables (all zero between steps) and the
    // duplicate list's workspace -- the count path of the medium-batch step (csrc/cdr_step.hip, "ids without a sort")
    uint32_t* idc_user; int64_t idc_user_rows;
    uint32_t* idc_item; int64_t idc_item_rows;
    void* idc_list; size_t idc_list_bytes;
    // cdr_conet_defer_finish: a training forward of the CoNet towers leaves the addition of its blocks' loss partials to the backward's
    // weight-gradient launch (one workgroup more there instead of a launch of its own); `pending` between the two calls
    // conet_fb_kernel's weight staging: per 16-byte chunk of the padded LDS weight area, the address it is copied from (built on the host,
    // re-sent only when a parameter moved; csrc/cdr_conet.hip: conet_dma_table)
    unsigned long long* conet_tab_dev; unsigned long long* conet_tab_pin; unsigned long long* conet_tab_shadow; int conet_tab_cap, conet_tab_n;
    int conet_defer, conet_pending, conet_fin_grid;
    int64_t conet_fin_ns, conet_fin_R;
    float* conet_fin_out;
    void* conet_fin_stream;
    // optional HIP-event brackets around the hot kernels, recorded on the launch stream (cdr_timing_*)
    int timing_cap, timing_n;
    hipEvent_t* ev0;
    hipEvent_t* ev1;
    int* tags;
};

// RAII bracket: records ev0 now and ev1 at scope exit on `s` when timing is enabled and a slot is free.
struct cdr_time_scope {
    cdr_ctx* c; hipStream_t s; int slot;
    cdr_time_scope(cdr_ctx* ctx, int tag, hipStream_t st) : c(ctx), s(st), slot(-1) {
        if (c && c->timing_cap > 0 && c->timing_n < c->timing_cap) {
            slot = c->timing_n++;
            c->tags[slot] = tag;
            (void)hipEventRecord(c->ev0[slot], s);
        }
    }
    ~cdr_time_scope() { if (slot >= 0) (void)hipEventRecord(c->ev1[slot], s); }
};

void cdr_set_error(const char* fmt, ...);
int cdr_ctx_scratch(cdr_ctx* ctx, size_t bytes, void** out);

#define CDR_CHECK_ARG(cond)                                                         \
    do {                                                                            \
        if (!(cond)) {                                                              \
            cdr_set_error("%s: invalid argument: %s", __func__, #cond);             \
            return CDR_EINVAL;                                                      \
        }                                                                           \
    } while (0)

#define CDR_HIP(expr)                                                               \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            cdr_set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(e_));  \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

#define CDR_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            cdr_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_));\
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

// ---- wave-level reductions -------------------------------------------------------------------------------------
// Sum over aligned sub-groups of LPR lanes (LPR a power of two, 1..64); every lane of the group gets the total.
// Butterfly without LDS: quad_perm / row_half_mirror / row_mirror DPP moves inside a 16-lane row, then gfx950's
// v_permlane16_swap / v_permlane32_swap across rows and wave halves (self-swap: result halves = the two partners).
// The obvious __shfl_xor chain compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt per step: ~140 LDS round trips per
// gather iteration, which is what bounded the gather kernels before (per-sample, not per-byte, limited).
template <int CTRL>
__device__ __forceinline__ float dpp_add_(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    if (LPR >= 2) v = dpp_add_<0xB1>(v);       // quad_perm [1,0,3,2]  : lane ^ 1
    if (LPR >= 4) v = dpp_add_<0x4E>(v);       // quad_perm [2,3,0,1]  : lane ^ 2
    if (LPR >= 8) v = dpp_add_<0x141>(v);      // row_half_mirror      : the other quad of the 8-lane half row
    if (LPR >= 16) v = dpp_add_<0x140>(v);     // row_mirror           : the other half of the 16-lane row
    if (LPR >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // rows (0,1) and (2,3)
    }
    if (LPR >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // wave halves
    }
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, CDR_WAVE);
    return v;
}

// Block-level sum of NV doubles; result valid in thread 0.  smem: NV * (blockDim/64) doubles.
template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_d(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += smem[w * NV + i];
            v[i] = s;
        }
    }
}

// Clears a few words on a stream with a KERNEL.  hipMemsetAsync is not used for this: issued inside a captured step it did not take
// effect on replay (cdr_row_flags: the list length it should have reset kept growing from replay to replay until the list overran;
// dropped from the capture or run out of order -- not determined; eager launches were always fine).
static __global__ void cdr_zero_u32_kernel(uint32_t* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t cdr_zero_u32(void* p, int64_t n_words, hipStream_t s) {
    int64_t g = (n_words + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    cdr_zero_u32_kernel<<<dim3((unsigned)g), dim3(256), 0, s>>>((uint32_t*)p, n_words);
    return hipGetLastError();
}

// ---- "the workgroup that signs in last finishes": a two-pass reduction without the second launch, for grids of a few hundred blocks
// (the finishing launch is ~4.7 us of a 40 us captured step at the reference's default batch of 2,048 rows).  Partials travel with
// system-scope stores and loads (written through / read past the per-XCD L2s, as in cdr_linear.hip): no release fence, which on this
// part is a write-back of the whole L2 per workgroup.  Tied to gfx950 for that reason.  ticket: a zero-initialised word (cdr_ctx::tickets);
// atomicInc wraps it back to 0, so the next launch on the stream finds it clean.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "cdr_common.h: the fence-free partial hand-over (cdr_sign_in_last) is only valid on gfx950; add __threadfence() pairs for another target"
#endif
__device__ __forceinline__ void cdr_store_sys(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double cdr_load_sys(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// call with the block's partials already stored by cdr_store_sys (any thread); true in EVERY thread of the block that signed in last
__device__ __forceinline__ bool cdr_sign_in_last(unsigned* ticket, unsigned nblocks) {
    __shared__ int cdr_last_flag_;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's partial stores have been acknowledged by memory ...
    __syncthreads();                                          // ... every thread's have
    if (threadIdx.x == 0) cdr_last_flag_ = atomicInc(ticket, nblocks - 1) == nblocks - 1 ? 1 : 0;
    __syncthreads();
    return cdr_last_flag_ != 0;
}
// The same for grids of a hundred workgroups and more: sign-ins on ONE word queue up at the memory side (measured: C1's pair loss kernel
// 11.6 us at 256 workgroups, and faster at 128 with a quarter of the lane groups idle), so the workgroups sign in on kSignGroups words a
// cache line apart (workgroup id modulo kSignGroups) and the last of each group signs in on the word in front of them: at most
// nblocks / kSignGroups + kSignGroups arrivals per address.  ticket: CDR_SIGNIN_WORDS zero-initialised words; every counter wraps to 0.
// Ordering as above, one hop longer: a group's last arrival happens after every member's stores were acknowledged, the finisher's after
// every group's.
constexpr unsigned kSignGroups = 8, kSignStride = 32;
__device__ __forceinline__ bool cdr_sign_in_last_wide(unsigned* ticket, unsigned nblocks) {
    __shared__ int cdr_last_flag_w_;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned bid = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned K = nblocks < kSignGroups ? nblocks : kSignGroups;
        const unsigned g = bid % K, ng = (nblocks - g + K - 1) / K;
        int last = 0;
        if (atomicInc(ticket + kSignStride * (1 + g), ng - 1) == ng - 1) last = atomicInc(ticket, K - 1) == K - 1 ? 1 : 0;
        cdr_last_flag_w_ = last;
    }
    __syncthreads();
    return cdr_last_flag_w_ != 0;
}
// Side job of a forward kernel: zero-fill n16 16-byte words (the dense gradient buffers its backward will scatter into -- a separate fill
// launch is ~5 us of a 40 us step; spread over a kernel that is waiting on its gathers anyway it is free).
__device__ __forceinline__ void cdr_scrub(uint4* __restrict__ z, int64_t n16) {
    if (!z) return;
    const int64_t nth = (int64_t)gridDim.x * gridDim.y * blockDim.x;
    const int64_t me = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = me; i < n16; i += nth) z[i] = zero;
}
// host side: hand the pending region to the launch being assembled (and clear it); `done` = false if this path cannot scrub in-kernel
static inline void cdr_take_scrub(cdr_ctx* ctx, uint4** z, int64_t* n16) {
    *z = (uint4*)ctx->scrub_ptr; *n16 = (int64_t)(ctx->scrub_bytes / 16);
    ctx->scrub_ptr = nullptr; ctx->scrub_bytes = 0;
}
constexpr int kSignInMaxBlocks = 512;        // beyond this the same-address atomics cost more than the launch they replace (profiles/r03_ab_adam_signin.txt)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Row traffic that is touched once per step (table rows, moments, gradient rows, staged rows of tables far larger than the caches) is
// marked non-temporal when a row is at least 512 bytes (LPR >= 32 lanes of one float4): replicated alternating A/B on one MI355X
// (profiles/r06_ab_nt.txt) -- the C5 step 3.975 -> 3.878 ms, its forward-and-update kernel 1.480 -> 1.445 ms (0.725 -> 0.743 of the HBM
// peak), the row-sharded step at one rank 6.32 -> 6.02 ms.  NOT for narrower rows: on the dimension layout's 64- and 128-byte column slices
// the same hint LOSES 7-12 % (tools/mb_dimshard.py, profiles/r06_mb_dimshard_nt_ab.json: a row that is read and written back relies on the
// line staying in L2 between the two).  -DCDR_NO_STREAM_NT builds the plain-policy library for that A/B.
#ifndef CDR_NO_STREAM_NT
typedef float cdr_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4n(const float* p) {
    if constexpr (NT) {
        const cdr_v4f v = __builtin_nontemporal_load(reinterpret_cast<const cdr_v4f*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return ld4(p);
    }
}
template <bool NT>
__device__ __forceinline__ void st4n(float* p, float4 v) {
    if constexpr (NT) {
        cdr_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
        __builtin_nontemporal_store(w, reinterpret_cast<cdr_v4f*>(p));
    } else {
        st4(p, v);
    }
}
#else
template <bool NT> __device__ __forceinline__ float4 ld4n(const float* p) { return ld4(p); }
template <bool NT> __device__ __forceinline__ void st4n(float* p, float4 v) { st4(p, v); }
#endif
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

static inline int cdr_lpr_for(int D) {        // lanes per row when a lane moves one float4
    int q = (D + 3) / 4, l = 1;
    while (l < q && l < 64) l <<= 1;
    return l;
}
