// Owner routing for row-sharded tables (row r lives on rank r % G): a stable counting sort of ids by owner built on the
// same rocPRIM radix sort as cdr_sort_ids (over ceil(log2 G) bits: one onesweep pass), the bucket boundaries, and the
// small index movers the exchange needs (permute-with-divide, inverse permutation).  Index plumbing only.
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "cdr_common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ __launch_bounds__(kBlock) void owner_keys_kernel(const int64_t* __restrict__ ids0, int64_t n0,
                                                            const int64_t* __restrict__ ids1, int64_t n1, int64_t G,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t n = n0 + n1, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        const int64_t id = e < n0 ? ids0[e] : ids1[e - n0];
        keys[e] = (uint32_t)(id % G);
        vals[e] = (uint32_t)e;
    }
}

// starts[k] = first sorted position whose owner is >= k  (k = 0..G) ; counts[k] = starts[k+1] - starts[k]
__global__ __launch_bounds__(kBlock) void bucket_bounds_kernel(const uint32_t* __restrict__ keys, int64_t n, int G,
                                                               int64_t* __restrict__ starts) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e <= n; e += stride) {
        const int lo = (e == 0) ? 0 : (int)keys[e - 1] + 1;
        const int hi = (e == n) ? G : (int)keys[e];
        for (int k = lo; k <= hi; ++k) starts[k] = e;
    }
}

__global__ void counts_from_starts_kernel(const int64_t* __restrict__ starts, int G, int64_t* __restrict__ counts) {
    const int k = threadIdx.x;
    if (k < G) counts[k] = starts[k + 1] - starts[k];
}

__global__ __launch_bounds__(kBlock) void permute_i64_kernel(const int64_t* __restrict__ src0, int64_t n0,
                                                             const int64_t* __restrict__ src1, const uint32_t* __restrict__ perm,
                                                             int64_t n, int64_t divisor, int64_t flag_below,
                                                             int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
        const int64_t o = perm[q];
        int64_t v = o < n0 ? src0[o] : src1[o - n0];
        if (divisor > 1) v /= divisor;
        if (o < flag_below) v |= (int64_t)1 << 62;           // tags the occurrence (positive item) for the owner
        out[q] = v;
    }
}

__global__ __launch_bounds__(kBlock) void inverse_perm_kernel(const uint32_t* __restrict__ perm, int64_t n, int64_t* __restrict__ pos) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) pos[perm[q]] = q;
}

// send3[q] = {uid / G, pid, nid} of triple perm[q] (perm == nullptr: q itself): the routed triples in owner order, interleaved -- one
// all-to-all moves them (24 bytes each)
__global__ __launch_bounds__(kBlock) void pack_triples_kernel(const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                              const int64_t* __restrict__ nid, const uint32_t* __restrict__ perm,
                                                              int64_t n, int64_t G, int64_t* __restrict__ send3) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
        const int64_t o = perm ? (int64_t)perm[q] : q;
        send3[3 * q] = uid[o] / G; send3[3 * q + 1] = pid[o]; send3[3 * q + 2] = nid[o];
    }
}
__global__ void set_count_kernel(int64_t* __restrict__ counts, int64_t n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) counts[0] = n;
}

// The owner key has <= 10 significant bits (world <= 1024): one Onesweep pass of 9-bit digits (two past 512 ranks) whatever the item count.  rocPRIM's
// default configuration merge-sorts up to 2^20 items however few key bits there are: 0.6 ms for the 1,048,576 triples of a C5 domain step
// (block sort + ~48 merge launches, profiles/r05_force_shard_row_kernel_stats.csv before this) against one ~35 us pass.
using route_sort_config = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 9,
                                        rocprim::block_radix_rank_algorithm::match>,
    (size_t)1 << 12>;          // merge sort only below 4,096 items
template <class... A>
inline hipError_t route_sort(A&&... a) { return rocprim::radix_sort_pairs<route_sort_config>(static_cast<A&&>(a)...); }     // (the size argument is an in/out reference)

inline unsigned bits_for(int64_t v) {
    unsigned b = 1;
    while (b < 32 && ((int64_t)1 << b) < v) ++b;
    return b;
}

}  // namespace

extern "C" int cdr_route_by_owner(cdr_ctx* ctx, void* stream, const int64_t* ids0, int64_t n0, const int64_t* ids1, int64_t n1,
                                  int world, uint32_t* perm, int64_t* counts, void* workspace, size_t workspace_bytes) {
    (void)ctx;
    const int64_t n = n0 + n1;
    CDR_CHECK_ARG(ids0 && n0 > 0 && (n1 == 0 || ids1) && perm && counts && workspace && world >= 1 && world <= 1024);
    CDR_CHECK_ARG(n <= (int64_t)0x7FFFFFFF);
    hipStream_t s = (hipStream_t)stream;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    const size_t starts_bytes = (((size_t)world + 2) * sizeof(int64_t) + 255) & ~(size_t)255;
    size_t tmp_need = 0;
    CDR_HIP(route_sort(nullptr, tmp_need, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                      (uint32_t*)nullptr, (size_t)n, 0u, bits_for(world)));
    CDR_CHECK_ARG(workspace_bytes >= 3 * arr + starts_bytes + tmp_need);
    uint32_t* keys_in = (uint32_t*)workspace;
    uint32_t* vals_in = (uint32_t*)((char*)workspace + arr);
    uint32_t* keys_out = (uint32_t*)((char*)workspace + 2 * arr);
    int64_t* starts = (int64_t*)((char*)workspace + 3 * arr);
    void* tmp = (char*)workspace + 3 * arr + starts_bytes;
    size_t tmp_bytes = workspace_bytes - 3 * arr - starts_bytes;
    owner_keys_kernel<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(ids0, n0, ids1, n1, (int64_t)world, keys_in, vals_in);
    CDR_LAUNCH_CHECK();
    CDR_HIP(route_sort(tmp, tmp_bytes, (const uint32_t*)keys_in, keys_out, (const uint32_t*)vals_in, perm, (size_t)n,
                                      0u, bits_for(world), s));
    bucket_bounds_kernel<<<dim3(grid_for(n + 1)), dim3(kBlock), 0, s>>>(keys_out, n, world, starts);
    CDR_LAUNCH_CHECK();
    counts_from_starts_kernel<<<dim3(1), dim3(1024), 0, s>>>(starts, world, counts);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_route_workspace_bytes(int64_t n, int world, size_t* bytes) {
    CDR_CHECK_ARG(bytes && n > 0 && world >= 1 && world <= 1024);
    size_t tmp_need = 0;
    hipError_t e = route_sort(nullptr, tmp_need, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (size_t)n, 0u, bits_for(world));
    if (e != hipSuccess) { cdr_set_error("cdr_route_workspace_bytes: %s", hipGetErrorString(e)); return (int)e; }
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    const size_t starts_bytes = (((size_t)world + 2) * sizeof(int64_t) + 255) & ~(size_t)255;
    *bytes = 3 * arr + starts_bytes + ((tmp_need + 255) & ~(size_t)255);
    return CDR_OK;
}

// Stage 0 of the row-sharded step in one call: bucket the triples by the owner of their USER row (uid % world, stable) and write them
// interleaved, {uid / world, pid, nid}, in owner order; counts[k] = triples for owner k.  One owner: nothing to bucket, one pass.
extern "C" int cdr_route_triples_workspace_bytes(int64_t n, int world, size_t* bytes) {
    CDR_CHECK_ARG(bytes && n > 0 && world >= 1 && world <= 1024);
    if (world == 1) { *bytes = 256; return CDR_OK; }
    size_t r = 0;
    int rc = cdr_route_workspace_bytes(n, world, &r);
    if (rc) return rc;
    *bytes = r + (((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255);
    return CDR_OK;
}

extern "C" int cdr_route_triples(cdr_ctx* ctx, void* stream, const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t n, int world,
                                 int64_t* send3, int64_t* counts, void* workspace, size_t workspace_bytes) {
    CDR_CHECK_ARG(uid && pid && nid && send3 && counts && workspace && n > 0 && n <= (int64_t)0x7FFFFFFF && world >= 1 && world <= 1024);
    hipStream_t s = (hipStream_t)stream;
    if (world == 1) {
        pack_triples_kernel<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(uid, pid, nid, nullptr, n, 1, send3);
        CDR_LAUNCH_CHECK();
        set_count_kernel<<<dim3(1), dim3(64), 0, s>>>(counts, n);
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    size_t need = 0;
    int rc = cdr_route_triples_workspace_bytes(n, world, &need);
    if (rc) return rc;
    CDR_CHECK_ARG(workspace_bytes >= need);
    const size_t parr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t* perm = (uint32_t*)workspace;
    rc = cdr_route_by_owner(ctx, stream, uid, n, nullptr, 0, world, perm, counts, (char*)workspace + parr, workspace_bytes - parr);
    if (rc) return rc;
    pack_triples_kernel<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(uid, pid, nid, perm, n, (int64_t)world, send3);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_permute_i64(void* stream, const int64_t* src0, int64_t n0, const int64_t* src1, const uint32_t* perm,
                               int64_t n, int64_t divisor, int64_t flag_below, int64_t* out) {
    CDR_CHECK_ARG(src0 && perm && out && n > 0 && divisor >= 1);
    permute_i64_kernel<<<dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream>>>(src0, n0, src1, perm, n, divisor, flag_below, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_inverse_perm(void* stream, const uint32_t* perm, int64_t n, int64_t* pos) {
    CDR_CHECK_ARG(perm && pos && n > 0);
    inverse_perm_kernel<<<dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream>>>(perm, n, pos);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// ---- full-sort over row-sharded item tables: after the all-gather the scores sit shard-major, [G][U][Nl] with shard g
// holding the items g, g+G, g+2G, ... ; the reference's layout is [U][N] in item-id order.  One streaming pass:
// out[u][c] = gathered[c % G][u][c / G].  A thread produces 4 consecutive outputs (one 16-byte store); within a wave the
// reads fall into G streams of consecutive floats, so both sides stay full-line.
__global__ __launch_bounds__(kBlock) void interleave_shards_kernel(const float* __restrict__ g, int G, int64_t U, int64_t Nl,
                                                                   int64_t N, float* __restrict__ out) {
    const int64_t quads = (N + 3) / 4;
    const int64_t total = U * quads;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t u = t / quads, c0 = (t - u * quads) * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + j;
            v[j] = c < N ? g[((int64_t)(c % G) * U + u) * Nl + c / G] : 0.f;
        }
        float* o = out + u * N + c0;
        if (c0 + 3 < N && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (c0 + j < N) o[j] = v[j];
        }
    }
}

extern "C" int cdr_interleave_shards(void* stream, const float* gathered, int world, int64_t U, int64_t Nl, int64_t N, float* out) {
    CDR_CHECK_ARG(gathered && out && world >= 1 && U > 0 && Nl > 0 && N > 0 && N <= Nl * (int64_t)world);
    const int64_t total = U * ((N + 3) / 4);
    interleave_shards_kernel<<<dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream>>>(gathered, world, U, Nl, N, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// rows of a sharded table for a REPLICATED id list: out[r] = table[ids[r] / G] if ids[r] % G == rank else 0 -- summing
// this over the ranks (all-reduce) gives every rank the gathered rows exactly (x + 0 == x).
__global__ __launch_bounds__(kBlock) void gather_owned_rows_kernel(const float* __restrict__ tab, int D, const int64_t* __restrict__ ids,
                                                                   int64_t n, int64_t G, int64_t rank, float* __restrict__ out) {
    const int64_t total = n * D;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t r = t / D; const int d = (int)(t - r * D);
        const int64_t id = ids[r];
        out[t] = (id % G == rank) ? tab[(id / G) * D + d] : 0.f;
    }
}

extern "C" int cdr_gather_owned_rows(void* stream, const float* tab, int D, const int64_t* ids, int64_t n, int world, int rank,
                                     float* out) {
    CDR_CHECK_ARG(tab && ids && out && D > 0 && n > 0 && world >= 1 && rank >= 0 && rank < world);
    gather_owned_rows_kernel<<<dim3(grid_for(n * D)), dim3(kBlock), 0, (hipStream_t)stream>>>(tab, D, ids, n, world, rank, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// Block-partitioned tables (bitgcf_shard.BlockPartition): a rank owns the rows whose position in the all-gathered layout lies in
// [lo, lo + rows).  For a REPLICATED list of positions (every rank holds the whole batch's ids) the rows travel by reduction instead of
// by an id exchange: out[r] = shard[pos[r] - lo] if this rank owns pos[r], else 0 (pos < 0: nobody's, a padding slot) -- a reduce-scatter
// (sum) of this over the ranks hands every rank the rows of ITS slice of the batch exactly (x + 0 + ... + 0 == x); and the way back,
// grad_shard[pos[r] - lo] += src[r] for the owned positions of the all-gathered gradient rows (bitgcf.py:221-247 over row-sharded tables).
__global__ __launch_bounds__(kBlock) void gather_block_rows_kernel(const float* __restrict__ tab, int D, const int64_t* __restrict__ pos,
                                                                   int64_t n, int64_t lo, int64_t rows, float* __restrict__ out) {
    const int64_t total = n * D;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t r = t / D; const int d = (int)(t - r * D);
        const int64_t q = pos[r] - lo;
        out[t] = (pos[r] >= 0 && q >= 0 && q < rows) ? tab[q * D + d] : 0.f;
    }
}

__global__ __launch_bounds__(kBlock) void scatter_add_block_rows_kernel(float* __restrict__ grad, int D, const int64_t* __restrict__ pos,
                                                                        int64_t n, int64_t lo, int64_t rows, const float* __restrict__ src) {
    const int64_t total = n * D;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t r = t / D; const int d = (int)(t - r * D);
        const int64_t q = pos[r] - lo;
        if (pos[r] >= 0 && q >= 0 && q < rows) atomicAdd(grad + q * D + d, src[t]);
    }
}

extern "C" int cdr_gather_block_rows(void* stream, const float* shard, int D, const int64_t* pos, int64_t n, int64_t lo, int64_t rows,
                                     float* out) {
    CDR_CHECK_ARG(shard && pos && out && D > 0 && n > 0 && rows > 0);
    gather_block_rows_kernel<<<dim3(grid_for(n * D)), dim3(kBlock), 0, (hipStream_t)stream>>>(shard, D, pos, n, lo, rows, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_scatter_add_block_rows(void* stream, float* grad_shard, int D, const int64_t* pos, int64_t n, int64_t lo, int64_t rows,
                                          const float* src) {
    CDR_CHECK_ARG(grad_shard && pos && src && D > 0 && n > 0 && rows > 0);
    scatter_add_block_rows_kernel<<<dim3(grid_for(n * D)), dim3(kBlock), 0, (hipStream_t)stream>>>(grad_shard, D, pos, n, lo, rows, src);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

// Per-shard top-k lists -> the global top-k (sharded full-sort evaluation): every rank all-gathers its k best
// (value, LOCAL row) per user; one wave per user then selects k times over the world*k candidates, translating
// local row l of producer p to the global item id l * world + p.  Order: value descending, ties to the smaller item id --
// independent of the rank count.  Producers pad short lists with (-inf, -1); those never win.
__global__ __launch_bounds__(kBlock) void topk_merge_shards_kernel(const float* __restrict__ vals, const int64_t* __restrict__ lidx,
                                                                   int G, int64_t U, int k, int to_global, float* __restrict__ out_v,
                                                                   int64_t* __restrict__ out_i) {
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (u >= U) return;
    const int C = G * k;
    constexpr int64_t kNone = INT64_MAX;
    uint64_t used = 0;                                   // bit s: my candidate lane + 64 s is already in the output
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int64_t bi = kNone;
        int bslot = -1;
        for (int s = 0, c = lane; c < C; ++s, c += 64) {
            if ((used >> s) & 1) continue;
            const int p = c / k, j = c - p * k;
            const int64_t o = ((int64_t)p * U + u) * k + j;
            const int64_t li = lidx[o];
            if (li < 0) continue;
            const float v = vals[o];
            const int64_t gi = to_global ? li * G + p : li;        // to_global 0: the producers already sent output columns
            if (v > bv || (v == bv && gi < bi)) { bv = v; bi = gi; bslot = s; }
        }
        float wv = bv;
        int64_t wi = bi;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(wv, off, 64);
            const int64_t oi = __shfl_xor(wi, off, 64);
            if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
        }
        if (bslot >= 0 && bi == wi) used |= 1ull << bslot;          // item ids are unique over the candidates
        if (lane == 0) {
            out_v[u * k + r] = wi == kNone ? -INFINITY : wv;
            out_i[u * k + r] = wi == kNone ? -1 : wi;
        }
    }
}

extern "C" int cdr_topk_merge_shards(void* stream, const float* vals, const int64_t* local_idx, int world, int64_t U, int k,
                                     int local_to_global, float* out_vals, int64_t* out_idx) {
    CDR_CHECK_ARG(vals && local_idx && out_vals && out_idx && world >= 1 && U > 0 && k > 0);
    CDR_CHECK_ARG((int64_t)world * k <= 64 * 64);
    topk_merge_shards_kernel<<<dim3((unsigned)((U + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, (hipStream_t)stream>>>(
        vals, local_idx, world, U, k, local_to_global, out_vals, out_idx);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
