// The device side of the loader's batch production (cdr_batch_produce / cdr_batch_produce_jobs, csrc/cdr_sampler.hip) as inline device
// functions, so that the SAME body can also run in extra workgroups of the dense Adam launch (cdr_adam_multi_dev_produce, csrc/cdr_rows.hip:
// the production of batch i + 1 beside the optimizer of step i -- round 6).  Included by both translation units.
#pragma once
#include "cdr_common.h"

namespace cdr_produce {

constexpr int kBlock = 256;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// One uniform draw for element e of user u (counter-based: seed, element, attempt): the body of neg_sample_kernel, shared with the
// batch producer below.
__device__ __forceinline__ int64_t draw_uniform(int64_t u, int64_t e, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1,
                                                const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices, uint64_t seed,
                                                int max_tries, int* __restrict__ fail_flag) {
    const int64_t n0 = hi0 > lo0 ? hi0 - lo0 : 0, n1 = hi1 > lo1 ? hi1 - lo1 : 0, ncand = n0 + n1;
    const int64_t b = indptr ? indptr[u] : 0, en = indptr ? indptr[u + 1] : 0;
    for (int t = 0; t < max_tries; ++t) {
        const uint64_t r = mix64(seed + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xD1B54A32D192ED03ull);
        // unbiased enough for catalogue sizes << 2^64: multiply-high of a 64-bit uniform by ncand
        const int64_t c = (int64_t)__umul64hi(r, (uint64_t)ncand);
        const int64_t id = c < n0 ? lo0 + c : lo1 + (c - n0);
        int64_t l = b, h = en;                       // binary search in the user's sorted used items
        while (l < h) {
            const int64_t mid = (l + h) >> 1;
            if (indices[mid] < id) l = mid + 1; else h = mid;
        }
        if (!(l < en && indices[l] == id)) return id;
    }
    // max_tries rejections (a user whose history covers most of the range): the reference's loop would go on until a
    // draw is free, i.e. it returns a uniform draw over the FREE candidates -- pick the r-th free id directly by walking
    // the user's sorted history inside the range.  O(history) for this one element; never taken at ordinary densities.
    auto lower = [&](int64_t x) { int64_t l = b, h = en; while (l < h) { const int64_t m = (l + h) >> 1; if (indices[m] < x) l = m + 1; else h = m; } return l; };
    const int64_t a0 = n0 ? lower(lo0) : b, z0 = n0 ? lower(hi0) : b, a1 = n1 ? lower(lo1) : b, z1 = n1 ? lower(hi1) : b;
    const int64_t free0 = n0 - (z0 - a0), free1 = n1 - (z1 - a1);
    if (free0 + free1 > 0) {
        const uint64_t r = mix64(seed + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)max_tries * 0xD1B54A32D192ED03ull);
        int64_t k_ = (int64_t)__umul64hi(r, (uint64_t)(free0 + free1));
        int64_t id, p0, p1;
        if (k_ < free0) { id = lo0 + k_; p0 = a0; p1 = z0; } else { id = lo1 + (k_ - free0); p0 = a1; p1 = z1; }
        for (int64_t q = p0; q < p1 && indices[q] <= id; ++q) ++id;     // skip the used ids at or below the running candidate
        return id;
    }
    if (fail_flag) atomicExch(fail_flag, 1);             // every candidate is used: there is no valid answer (the loaders refuse such users)
    return n0 ? lo0 : lo1;
}

// Popularity-biased candidates (crossdomain_sampler.py:66-114): Walker alias table over the distinct items of the sampler's
// interactions, built on the host exactly as the reference does; a draw is "uniform column c, uniform p: p < prob[c] ? keys[c] :
// alias[c]".  Same rejection against the user's used items, same k-major layout.
__device__ __forceinline__ int64_t draw_alias(int64_t u, int64_t e, const int64_t* __restrict__ keys, const float* __restrict__ prob,
                                              const int64_t* __restrict__ alias, int64_t n_keys, const int64_t* __restrict__ indptr,
                                              const int64_t* __restrict__ indices, uint64_t seed, int max_tries,
                                              int* __restrict__ fail_flag) {
    const int64_t b = indptr ? indptr[u] : 0, en = indptr ? indptr[u + 1] : 0;
    for (int t = 0; t < max_tries; ++t) {
        const uint64_t r = mix64(seed + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xD1B54A32D192ED03ull);
        const int64_t c = (int64_t)__umul64hi(r, (uint64_t)n_keys);
        const float p = (float)(mix64(r ^ 0xA5A5A5A5A5A5A5A5ull) >> 40) * (1.0f / 16777216.0f);       // 24-bit uniform in [0,1)
        const int64_t al = alias[c];
        const int64_t id = (prob[c] > p || al < 0) ? keys[c] : al;      // al < 0: a column the construction left whole
        int64_t l = b, h = en;
        while (l < h) {
            const int64_t mid = (l + h) >> 1;
            if (indices[mid] < id) l = mid + 1; else h = mid;
        }
        if (!(l < en && indices[l] == id)) return id;
    }
    // max_tries rejections: scan the alias table's keys cyclically from a random column for one the user has not used
    // (valid; for this rare element the popularity weighting is given up -- the reference would keep drawing)
    const uint64_t r = mix64(seed + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)max_tries * 0xD1B54A32D192ED03ull);
    const int64_t c0 = (int64_t)__umul64hi(r, (uint64_t)n_keys);
    for (int64_t j = 0; j < n_keys; ++j) {
        const int64_t id = keys[(c0 + j) % n_keys];
        int64_t l = b, h = en;
        while (l < h) { const int64_t mid = (l + h) >> 1; if (indices[mid] < id) l = mid + 1; else h = mid; }
        if (!(l < en && indices[l] == id)) return id;
    }
    if (fail_flag) atomicExch(fail_flag, 1);
    return keys[0];
}


struct batch_jobs { cdr_batch_job j[CDR_BATCH_MAX_JOBS]; };

// One workgroup's share of job J: workgroup `bx` of `gx` (the grid row of the stand-alone launch).  See cdr_sampler.hip for the layouts.
__device__ __forceinline__ void batch_produce_body(const cdr_batch_job& J, unsigned bx, unsigned gx) {
    int64_t* __restrict__ cursor = J.cursor;
    const int64_t start = __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t draws = __hip_atomic_load(cursor + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t sd = J.seed + (uint64_t)draws * 0x85EBCA77C2B2AE63ull;
    const int k = J.k, pointwise = J.pointwise;
    const int64_t S = J.S;
    const int T = k == 0 ? 1 : (pointwise ? 1 + k : k);
    const int64_t total = S * (int64_t)T, stride = (int64_t)gx * kBlock;
    for (int64_t e = (int64_t)bx * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t j = e % S, m = e / S, row = start + j;
        const bool in = row < J.n_rows;
        const int64_t u = in ? J.users_all[row] : 0;
        J.out_users[e] = u;
        if (k == 0) continue;
        if (!pointwise || m == 0) J.out_items[e] = in ? J.items_all[row] : 0;
        if (pointwise && m == 0) continue;
        const int64_t es = pointwise ? e - S : e;                 // index in the k-major negative list [S k]
        int64_t neg = 0;
        if (in) neg = J.dist == 0 ? draw_uniform(u, es, J.lo0, J.hi0, J.lo1, J.hi1, J.used_indptr, J.used_indices, sd, 64, J.fail_flag)
                                  : draw_alias(u, es, J.keys, J.prob, J.alias, J.n_keys, J.used_indptr, J.used_indices, sd, 64, J.fail_flag);
        if (pointwise) J.out_items[e] = neg; else J.out_neg[e] = neg;
    }
    // every thread of this workgroup has read the cursor (first statements) -- sign in; the last workgroup of the job moves it on
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(cursor + 2);
        if (atomicInc(ticket, gx - 1) == gx - 1) {                             // wraps to 0 for the next launch
            __hip_atomic_store(cursor, start + S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cursor + 1, draws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// grid width of a job (workgroups of kBlock threads), as the stand-alone launch sizes it
static inline int64_t job_grid(const cdr_batch_job& J) {
    const int T = J.k == 0 ? 1 : (J.pointwise ? 1 + J.k : J.k);
    int64_t g = (J.S * T + kBlock - 1) / kBlock;
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    return g < 1 ? 1 : g;
}

}  // namespace cdr_produce
