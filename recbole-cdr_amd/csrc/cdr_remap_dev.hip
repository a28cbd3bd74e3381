// The overlap id remap at scale, on the device (VERDICT r4 missing #3): CrossDomainDataset.calculate_user_item_from_both_domain
// + _remap_fields (recbole_cdr/data/dataset.py:344-445, :109-123) for ONE field of both domains, bit-exact with cdr_overlap_remap
// (csrc/cdr_remap.cpp, the host form: hash sets + std::sort of strings, single-threaded) and with the oracle.
//
// Formulation.  The reference builds three Python sets of tokens (overlap, target-only, source-only), sorts each in str order (== UTF-8
// byte order) and numbers them [1, OV) | [OV, OV + TO) | [OV + TO, total).  Sorting ALL occurrences of both domains together in byte
// order gives the same thing without a set or a hash: equal tokens are adjacent (a run = one distinct token), a run's class is the OR of
// the domain bits of its occurrences, and -- because the restriction of a total order to a subset is the subset's order -- the id of a
// run is its class's base + the number of runs of the SAME class before it.  That is K + 1 stable radix sorts (K = ceil(longest token /
// 8): one per 8-byte big-endian chunk from the last to the first, after one by length -- zero padding + the length as the least
// significant key IS byte order with "a prefix sorts first"), one adjacent-compare pass, two scans and two scatters: integer work
// at HBM speed, no string ever compared on the host.
//
//   valid occurrences -> list            (NaN tokens are left out: their id is -1)
//   LSD passes: key[q] = chunk_j(token(list[q])) ; stable sort (key, list)        rocPRIM radix_sort_pairs, 64-bit keys
//   head[q]   = token(list[q]) != token(list[q-1])  ->  run[q] = inclusive_scan(head) - 1
//   cls[run] |= domain bit of the occurrence (atomicOr) ; runhead[run] = list[q] at head positions
//   (a, b, c)[run] = exclusive scan of the class indicators (overlap, target-only, source-only)
//   id[run]   = overlap: 1 + a  |  target-only: OV + b  |  source-only: OV + TO + c ;  the overlap token '[PAD]' -> 0 (dataset.py:391)
//   out[list[q]] = id[run[q]]
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "cdr_common.h"

namespace {

constexpr int kBlock = 256;

struct tok_src {
    const uint8_t* bytes[2];
    const int64_t* off[2];
    const uint8_t* isnan[2];
    int64_t n_src, n;                     // occurrence e < n_src: source token e, else target token e - n_src
};

__device__ __forceinline__ void tok_of(const tok_src& t, int64_t e, const uint8_t*& p, int64_t& len) {
    const int side = e >= t.n_src;
    const int64_t i = side ? e - t.n_src : e;
    const int64_t a = t.off[side][i];
    p = t.bytes[side] + a;
    len = t.off[side][i + 1] - a;
}

inline int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > CDR_NUM_CU * 16) g = CDR_NUM_CU * 16;
    return (int)(g < 1 ? 1 : g);
}

// valid[e] = 1 for a real token; stats[1] = longest token (bytes)
__global__ __launch_bounds__(kBlock) void rd_valid_kernel(tok_src t, uint32_t* __restrict__ valid, unsigned long long* __restrict__ stats,
                                                          int64_t* __restrict__ src_ids, int64_t* __restrict__ tgt_ids) {
    unsigned long long mx = 0;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < t.n; e += (int64_t)gridDim.x * kBlock) {
        const int side = e >= t.n_src;
        const int64_t i = side ? e - t.n_src : e;
        const bool nan = t.isnan[side] && t.isnan[side][i];
        valid[e] = nan ? 0u : 1u;
        (side ? tgt_ids : src_ids)[i] = -1;
        if (!nan) { const unsigned long long l = (unsigned long long)(t.off[side][i + 1] - t.off[side][i]); mx = l > mx ? l : mx; }
    }
    for (int d = 32; d; d >>= 1) { const unsigned long long o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&stats[1], mx);
}

// list[pos[e]] = e for valid occurrences (pos = exclusive scan of valid); stats[0] = number of valid occurrences
__global__ __launch_bounds__(kBlock) void rd_compact_kernel(const uint32_t* __restrict__ valid, const uint32_t* __restrict__ pos, int64_t n,
                                                            uint32_t* __restrict__ list, unsigned long long* __restrict__ stats) {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
        if (valid[e]) list[pos[e]] = (uint32_t)e;
        if (e == n - 1) stats[0] = (unsigned long long)pos[e] + valid[e];
    }
}

// key[q] = bytes [8 j, 8 j + 8) of token list[q], big-endian, zero padded (j < 0: the token's length)
__global__ __launch_bounds__(kBlock) void rd_keys_kernel(tok_src t, const uint32_t* __restrict__ list, int64_t nv, int j, int shift,
                                                         uint64_t* __restrict__ keys) {
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < nv; q += (int64_t)gridDim.x * kBlock) {
        const uint8_t* p; int64_t len;
        tok_of(t, (int64_t)list[q], p, len);
        uint64_t k;
        if (j < 0) k = (uint64_t)len;
        else {
            k = 0;
            const int64_t b0 = 8 * (int64_t)j;
#pragma unroll
            for (int b = 0; b < 8; ++b) k = (k << 8) | (uint64_t)(b0 + b < len ? p[b0 + b] : (uint8_t)0);
            k >>= shift;                                     // the chunk's bytes that can be real (the longest token decides), right-aligned
        }
        keys[q] = k;
    }
}

// head[q] = 1 where the token at sorted position q differs from its predecessor's
__global__ __launch_bounds__(kBlock) void rd_heads_kernel(tok_src t, const uint32_t* __restrict__ list, int64_t nv, uint32_t* __restrict__ head) {
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < nv; q += (int64_t)gridDim.x * kBlock) {
        bool h = q == 0;
        if (!h) {
            const uint8_t *a, *b; int64_t la, lb;
            tok_of(t, (int64_t)list[q], a, la);
            tok_of(t, (int64_t)list[q - 1], b, lb);
            h = la != lb;
            for (int64_t i = 0; !h && i < la; ++i) h = a[i] != b[i];
        }
        head[q] = h ? 1u : 0u;
    }
}

// run[q] (inclusive scan of head, 1-based) -> class bits and the first occurrence of every run
__global__ __launch_bounds__(kBlock) void rd_class_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ head,
                                                          const uint32_t* __restrict__ run1, int64_t nv, int64_t n_src,
                                                          uint32_t* __restrict__ cls, uint32_t* __restrict__ runhead) {
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < nv; q += (int64_t)gridDim.x * kBlock) {
        const uint32_t r = run1[q] - 1u, e = list[q];
        atomicOr(&cls[r], (int64_t)e < n_src ? 1u : 2u);
        if (head[q]) runhead[r] = e;
    }
}

struct u3 { uint32_t a, b, c; };
struct u3_plus { __host__ __device__ u3 operator()(const u3& x, const u3& y) const { return u3{x.a + y.a, x.b + y.b, x.c + y.c}; } };

// indicator of each run's class: a = overlap (both domain bits), b = target-only, c = source-only; runs past the last one: zero
__global__ __launch_bounds__(kBlock) void rd_indicator_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ run1, int64_t nv,
                                                              u3* __restrict__ ind) {
    const int64_t nr = (int64_t)run1[nv - 1];
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < nv; r += (int64_t)gridDim.x * kBlock) {
        const uint32_t c = r < nr ? cls[r] : 0u;
        ind[r] = u3{c == 3u ? 1u : 0u, c == 2u ? 1u : 0u, c == 1u ? 1u : 0u};
    }
}

// counts4 = {OV (PAD counted), source-only, target-only, total}; id of every run
__global__ __launch_bounds__(kBlock) void rd_ids_kernel(tok_src t, const uint32_t* __restrict__ cls, const uint32_t* __restrict__ runhead,
                                                        const uint32_t* __restrict__ run1, int64_t nv, const u3* __restrict__ ex,
                                                        const u3* __restrict__ ind, int64_t* __restrict__ run_id, int64_t* __restrict__ counts4) {
    const int64_t nr = (int64_t)run1[nv - 1];
    const u3 le = ex[nr - 1], li = ind[nr - 1];
    const int64_t n_ov = 1 + (int64_t)le.a + li.a, n_to = (int64_t)le.b + li.b, n_so = (int64_t)le.c + li.c;
    if (blockIdx.x == 0 && threadIdx.x == 0) { counts4[0] = n_ov; counts4[1] = n_so; counts4[2] = n_to; counts4[3] = n_ov + n_so + n_to; }
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < nr; r += (int64_t)gridDim.x * kBlock) {
        const uint32_t c = cls[r];
        int64_t id = c == 3u ? 1 + (int64_t)ex[r].a : c == 2u ? n_ov + (int64_t)ex[r].b : n_ov + n_to + (int64_t)ex[r].c;
        if (c == 3u) {                                   // overlap_remap_dict['[PAD]'] = 0 is assigned after the zip (dataset.py:391)
            const uint8_t* p; int64_t len;
            tok_of(t, (int64_t)runhead[r], p, len);
            if (len == 5 && p[0] == '[' && p[1] == 'P' && p[2] == 'A' && p[3] == 'D' && p[4] == ']') id = 0;
        }
        run_id[r] = id;
    }
}

__global__ __launch_bounds__(kBlock) void rd_scatter_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ run1, int64_t nv,
                                                            int64_t n_src, const int64_t* __restrict__ run_id, int64_t* __restrict__ src_ids,
                                                            int64_t* __restrict__ tgt_ids) {
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < nv; q += (int64_t)gridDim.x * kBlock) {
        const int64_t e = (int64_t)list[q], id = run_id[run1[q] - 1u];
        if (e < n_src) src_ids[e] = id; else tgt_ids[e - n_src] = id;
    }
}

__global__ void rd_empty_counts_kernel(int64_t* __restrict__ counts4) { counts4[0] = 1; counts4[1] = 0; counts4[2] = 0; counts4[3] = 1; }

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// carve-up of the caller's workspace for n occurrences
struct rd_layout {
    size_t stats, valid, pos, listA, listB, keysA, keysB, cls, runhead, ind, ex, run_id, tmp, tmp_bytes, total;
};

hipError_t rd_plan(int64_t n, rd_layout& L) {
    size_t t_sort = 0, t_scan32 = 0, t_scan3 = 0;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    hipError_t e = rocprim::radix_sort_pairs(nullptr, t_sort, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, nn, 0u, 64u, (hipStream_t)0);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(nullptr, t_scan32, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, nn, rocprim::plus<uint32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(nullptr, t_scan3, (const u3*)nullptr, (u3*)nullptr, u3{0u, 0u, 0u}, nn, u3_plus(), (hipStream_t)0);
    if (e != hipSuccess) return e;
    size_t o = 0;
    auto take = [&](size_t b) { const size_t at = o; o += up256(b); return at; };
    L.stats = take(16);
    L.valid = take(4 * nn); L.pos = take(4 * nn);               // (later: head, run1)
    L.listA = take(4 * nn); L.listB = take(4 * nn);
    L.keysA = take(8 * nn); L.keysB = take(8 * nn);
    L.cls = take(4 * nn); L.runhead = take(4 * nn);
    L.ind = take(12 * nn); L.ex = take(12 * nn);
    L.run_id = take(8 * nn);
    L.tmp_bytes = t_sort > t_scan32 ? t_sort : t_scan32;
    if (t_scan3 > L.tmp_bytes) L.tmp_bytes = t_scan3;
    L.tmp = take(L.tmp_bytes);
    L.total = o;
    return hipSuccess;
}

}  // namespace

extern "C" int cdr_overlap_remap_dev_workspace_bytes(int64_t n_src, int64_t n_tgt, size_t* bytes) {
    CDR_CHECK_ARG(bytes && n_src >= 0 && n_tgt >= 0 && n_src + n_tgt < ((int64_t)1 << 32) - 1);
    rd_layout L;
    CDR_HIP(rd_plan(n_src + n_tgt, L));
    *bytes = L.total;
    return CDR_OK;
}

extern "C" int cdr_overlap_remap_dev(void* stream, const uint8_t* src_bytes, const int64_t* src_off, const uint8_t* src_isnan, int64_t n_src,
                                     const uint8_t* tgt_bytes, const int64_t* tgt_off, const uint8_t* tgt_isnan, int64_t n_tgt,
                                     int64_t* src_ids, int64_t* tgt_ids, int64_t* counts4, void* workspace, size_t workspace_bytes,
                                     int64_t* passes_out) {
    CDR_CHECK_ARG(counts4 && workspace && n_src >= 0 && n_tgt >= 0 && n_src + n_tgt < ((int64_t)1 << 32) - 1);
    CDR_CHECK_ARG((n_src == 0 || (src_off && src_ids)) && (n_tgt == 0 || (tgt_off && tgt_ids)));
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = n_src + n_tgt;
    if (passes_out) *passes_out = 0;
    if (n == 0) {
        rd_empty_counts_kernel<<<dim3(1), dim3(1), 0, s>>>(counts4);
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    rd_layout L;
    CDR_HIP(rd_plan(n, L));
    CDR_CHECK_ARG(workspace_bytes >= L.total && ((uintptr_t)workspace & 255) == 0);
    char* w = (char*)workspace;
    auto* stats = (unsigned long long*)(w + L.stats);
    auto* valid = (uint32_t*)(w + L.valid);
    auto* pos = (uint32_t*)(w + L.pos);
    uint32_t* list[2] = {(uint32_t*)(w + L.listA), (uint32_t*)(w + L.listB)};
    uint64_t* keys[2] = {(uint64_t*)(w + L.keysA), (uint64_t*)(w + L.keysB)};
    auto* cls = (uint32_t*)(w + L.cls);
    auto* runhead = (uint32_t*)(w + L.runhead);
    auto* ind = (u3*)(w + L.ind);
    auto* ex = (u3*)(w + L.ex);
    auto* run_id = (int64_t*)(w + L.run_id);
    void* tmp = w + L.tmp;
    size_t tb = L.tmp_bytes;
    const tok_src t{{src_bytes, tgt_bytes}, {src_off, tgt_off}, {src_isnan, tgt_isnan}, n_src, n};

    CDR_HIP(cdr_zero_u32(stats, 4, s));
    rd_valid_kernel<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(t, valid, stats, src_ids, tgt_ids);
    CDR_LAUNCH_CHECK();
    CDR_HIP(rocprim::exclusive_scan(tmp, tb, valid, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
    rd_compact_kernel<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(valid, pos, n, list[0], stats);
    CDR_LAUNCH_CHECK();
    unsigned long long h[2] = {0, 0};
    CDR_HIP(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, s));
    CDR_HIP(hipStreamSynchronize(s));                      // the one host wait: how many tokens, how long the longest (= how many passes)
    const int64_t nv = (int64_t)h[0];
    const int K = (int)((h[1] + 7) / 8);
    if (nv == 0) {
        rd_empty_counts_kernel<<<dim3(1), dim3(1), 0, s>>>(counts4);
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    // ---- LSD: length first (least significant), then the 8-byte chunks from the last to the first
    int cur = 0;
    unsigned len_bits = 1;
    while (len_bits < 64 && (h[1] >> len_bits)) ++len_bits;
    for (int j = -1, left = K + 1; left > 0; --left, j = left - 1) {          // j = -1, K - 1, K - 2, ..., 0
        const int64_t real = j < 0 ? 0 : ((int64_t)h[1] - 8 * (int64_t)j >= 8 ? 8 : (int64_t)h[1] - 8 * (int64_t)j);
        rd_keys_kernel<<<dim3(grid_for(nv)), dim3(kBlock), 0, s>>>(t, list[cur], nv, j, (int)(64 - 8 * real), keys[0]);
        CDR_LAUNCH_CHECK();
        tb = L.tmp_bytes;
        // only the key bits that can differ are sorted on: the length needs len_bits; chunk j holds min(8, longest - 8 j) real bytes
        // (tokens of up to 9 bytes: 1 + 1 + 8 digit passes of 8 bits instead of 1 + 8 + 8)
        const unsigned hi = j < 0 ? len_bits : (unsigned)(8 * real);          // (rd_keys_kernel right-aligns the chunk's real bytes: bits [0, hi))
        CDR_HIP(rocprim::radix_sort_pairs(tmp, tb, (const uint64_t*)keys[0], keys[1], (const uint32_t*)list[cur], list[cur ^ 1], (size_t)nv, 0u,
                                          hi, s));
        cur ^= 1;
    }
    if (passes_out) *passes_out = K + 1;
    uint32_t* L_ = list[cur];
    uint32_t* head = valid;                                 // (the compaction's arrays are free again)
    uint32_t* run1 = pos;
    rd_heads_kernel<<<dim3(grid_for(nv)), dim3(kBlock), 0, s>>>(t, L_, nv, head);
    CDR_LAUNCH_CHECK();
    tb = L.tmp_bytes;
    CDR_HIP(rocprim::inclusive_scan(tmp, tb, head, run1, (size_t)nv, rocprim::plus<uint32_t>(), s));
    CDR_HIP(cdr_zero_u32(cls, nv, s));
    rd_class_kernel<<<dim3(grid_for(nv)), dim3(kBlock), 0, s>>>(L_, head, run1, nv, n_src, cls, runhead);
    CDR_LAUNCH_CHECK();
    rd_indicator_kernel<<<dim3(grid_for(nv)), dim3(kBlock), 0, s>>>(cls, run1, nv, ind);
    CDR_LAUNCH_CHECK();
    tb = L.tmp_bytes;
    CDR_HIP(rocprim::exclusive_scan(tmp, tb, ind, ex, u3{0u, 0u, 0u}, (size_t)nv, u3_plus(), s));
    rd_ids_kernel<<<dim3(grid_for(nv)), dim3(kBlock), 0, s>>>(t, cls, runhead, run1, nv, ex, ind, run_id, counts4);
    CDR_LAUNCH_CHECK();
    rd_scatter_kernel<<<dim3(grid_for(nv)), dim3(kBlock), 0, s>>>(L_, run1, nv, n_src, run_id, src_ids, tgt_ids);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
