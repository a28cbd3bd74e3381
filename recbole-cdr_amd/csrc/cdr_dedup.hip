// Id de-duplication for the row-sharded step ("dedup ids -> all-to-all(v) of id lists -> owners gather rows ->
// all-to-all(v) of rows ... hence overlap + dedup are mandatory"): a rank requests every distinct item row ONCE per step and
// sends back ONE gradient row per distinct item, already summed over its occurrences.  With skewed (real) id streams that is
// the difference between 2 rows per triple and ~0.5 (Zipf(1.05): 27 % of the positives are distinct).
//
//   cdr_dedup_sorted   occurrences sorted by key = (owner << local_bits) | local_row (cdr_sort_ids on those keys):
//                      segment heads -> dense unique index (exclusive scan), unique local rows, occurrence -> unique map,
//                      unique count per owner.
//   cdr_segsum_rows    out[j] = sum over the occurrences of unique j of (+/-) G[occurrence]  (+ coef * #positives * row_j,
//                      the EmbLoss term, so that the owner needs no per-occurrence tags), summed in occurrence order;
//                      segments longer than 32 are cut into pieces of 256 that are summed in parallel and combined in piece
//                      order -- the same fixed-order scheme as cdr_rowwise_apply.
#include <cstring>
#include <string.h>
#include <rocprim/device/device_scan.hpp>
#include "cdr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kLongSeg = 32;
constexpr int kPiece = 256;

inline int grid_cap(int64_t g) {
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ __launch_bounds__(kBlock) void head_flags_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) flags[q] = (q == 0 || keys[q] != keys[q - 1]) ? 1u : 0u;
}

// uidx[q] = (exclusive scan of the head flags)[q] + flag[q] - 1 = dense index of q's segment
__global__ __launch_bounds__(kBlock) void dedup_emit_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                            const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl,
                                                            int64_t n, int world, unsigned local_bits, uint32_t* __restrict__ uidx,
                                                            int64_t* __restrict__ uniq_local, int64_t* __restrict__ umap,
                                                            int64_t* __restrict__ counts, int64_t* __restrict__ n_uniq) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const uint32_t lmask = (1u << local_bits) - 1u;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
        const uint32_t j = excl[q] + flags[q] - 1u;
        uidx[q] = j;
        umap[perm[q]] = (int64_t)j;
        if (flags[q]) {
            const uint32_t key = keys[q];
            uniq_local[j] = (int64_t)(key & lmask);
            // owner boundaries of the (sorted) uniques: starts[k] = first unique index whose owner is >= k  (no atomics: two
            // million increments of eight counters serialise for tens of milliseconds)
            const int owner = (int)(key >> local_bits);
            const int prev = q > 0 ? (int)(keys[q - 1] >> local_bits) : -1;
            for (int k = prev + 1; k <= owner; ++k) counts[k] = (int64_t)j;
        }
        if (q == n - 1) {
            n_uniq[0] = (int64_t)j + 1;
            for (int k = (int)(keys[q] >> local_bits) + 1; k <= world; ++k) counts[k] = (int64_t)j + 1;
        }
    }
}

// counts[k] <- starts[k+1] - starts[k]  (in place; one thread, world <= 1024)
__global__ void counts_from_starts_kernel(int64_t* __restrict__ starts_counts, int world) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int k = 0; k < world; ++k) starts_counts[k] = starts_counts[k + 1] - starts_counts[k];
}

struct seg_long { int64_t head, len, base; };
struct seg_piece { int64_t start; int64_t len; };

__device__ __forceinline__ void add_signed(float4& acc, const float4 g, bool neg) {
    if (neg) { acc.x -= g.x; acc.y -= g.y; acc.z -= g.z; acc.w -= g.w; }
    else { acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w; }
}

// heads of short segments sum them; heads of long ones only register pieces (one thread per position, after the main loop)
template <int LPR>
__global__ __launch_bounds__(kBlock) void segsum_head_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                             const uint32_t* __restrict__ uidx, int64_t n, const float* __restrict__ G,
                                                             int64_t neg_start, int D, const float* __restrict__ rows,
                                                             const float* __restrict__ reg_coef, float* __restrict__ out,
                                                             unsigned* __restrict__ counters, seg_long* __restrict__ longs,
                                                             seg_piece* __restrict__ pieces) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    for (int64_t q = gg; q < n; q += TG) {
        const uint32_t key = keys[q];
        const uint32_t before = keys[q > 0 ? q - 1 : 0];
        const uint32_t far = keys[q + kLongSeg < n ? q + kLongSeg : n - 1];
        if ((q > 0 && before == key) || (q + kLongSeg < n && far == key)) continue;
        const int64_t j = uidx[q];
        for (int ch = sub; ch < D4; ch += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnt = 0;
            for (int64_t e = q; e < n && keys[e] == key; ++e) {
                const int64_t o = perm[e];
                const bool neg = o >= neg_start;
                add_signed(acc, ld4n<(LPR >= 32)>(G + (neg ? o - neg_start : o) * D + 4 * ch), neg);
                cnt += neg ? 0 : 1;
            }
            if (c != 0.f && cnt) {
                const float4 w = ld4n<(LPR >= 32)>(rows + j * D + 4 * ch);
                const float rc = c * (float)cnt;
                acc.x += rc * w.x; acc.y += rc * w.y; acc.z += rc * w.z; acc.w += rc * w.w;
            }
            st4n<(LPR >= 32)>(out + j * D + 4 * ch, acc);
        }
    }
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q + kLongSeg < n; q += (int64_t)gridDim.x * kBlock) {
        const uint32_t key = keys[q];
        if ((q > 0 && keys[q - 1] == key) || keys[q + kLongSeg] != key) continue;
        int64_t lo = q + kLongSeg, hi = n;
        while (lo + 1 < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] == key) lo = mid; else hi = mid;
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

template <int LPR>
__global__ __launch_bounds__(kBlock) void segsum_piece_kernel(int D, const uint32_t* __restrict__ perm, const float* __restrict__ G,
                                                              int64_t neg_start, const unsigned* __restrict__ counters,
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
                for (int j = 0; j < UN; ++j)
                    g[j] = o[j] >= 0 ? ld4n<(LPR >= 32)>(G + (o[j] >= neg_start ? o[j] - neg_start : o[j]) * D + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    if (o[j] < 0) continue;
                    add_signed(acc, g[j], o[j] >= neg_start);
                    cnt += o[j] >= neg_start ? 0 : 1;
                }
            }
            st4n<(LPR >= 32)>(partial + pi * D + 4 * ch, acc);
            if (ch == 0) pcnt[pi] = cnt;
        }
    }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void segsum_long_finish_kernel(int D, const uint32_t* __restrict__ uidx,
                                                                    const float* __restrict__ rows, const float* __restrict__ reg_coef,
                                                                    const unsigned* __restrict__ counters, const seg_long* __restrict__ longs,
                                                                    const float* __restrict__ partial, const int* __restrict__ pcnt,
                                                                    float* __restrict__ out) {
    constexpr int GPB = kBlock / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D4 = D >> 2;
    const float c = reg_coef ? reg_coef[0] : 0.f;
    const int64_t nl = counters[1];
    for (int64_t li = gg; li < nl; li += TG) {
        const seg_long sg = longs[li];
        const int64_t j = uidx[sg.head];
        const int64_t np = (sg.len + kPiece - 1) / kPiece;
        for (int ch = sub; ch < D4; ch += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnt = 0;
            for (int64_t k = 0; k < np; ++k) {
                const float4 g = ld4n<(LPR >= 32)>(partial + (sg.base + k) * D + 4 * ch);
                acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
                cnt += pcnt[sg.base + k];
            }
            if (c != 0.f && cnt) {
                const float4 w = ld4n<(LPR >= 32)>(rows + j * D + 4 * ch);
                const float rc = c * (float)cnt;
                acc.x += rc * w.x; acc.y += rc * w.y; acc.z += rc * w.z; acc.w += rc * w.w;
            }
            st4n<(LPR >= 32)>(out + j * D + 4 * ch, acc);
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

}  // namespace

extern "C" int cdr_dedup_workspace_bytes(int64_t n, size_t* bytes) {
    CDR_CHECK_ARG(bytes && n > 0);
    size_t tmp = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>());
    if (e != hipSuccess) { cdr_set_error("cdr_dedup_workspace_bytes: %s", hipGetErrorString(e)); return (int)e; }
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    *bytes = 2 * arr + ((tmp + 255) & ~(size_t)255);
    return CDR_OK;
}

extern "C" int cdr_dedup_sorted(void* stream, const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, int world,
                                int local_bits, uint32_t* uniq_index, int64_t* uniq_local, int64_t* occ_to_uniq, int64_t* counts,
                                int64_t* n_uniq, void* workspace, size_t workspace_bytes) {
    CDR_CHECK_ARG(keys_sorted && perm && uniq_index && uniq_local && occ_to_uniq && counts && n_uniq && workspace && n > 0);
    CDR_CHECK_ARG(world >= 1 && local_bits >= 1 && local_bits < 32);
    size_t need = 0;
    int rc = cdr_dedup_workspace_bytes(n, &need);
    if (rc) return rc;
    CDR_CHECK_ARG(workspace_bytes >= need);
    hipStream_t s = (hipStream_t)stream;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t* flags = (uint32_t*)workspace;
    uint32_t* excl = (uint32_t*)((char*)workspace + arr);
    void* tmp = (char*)workspace + 2 * arr;
    size_t tmp_bytes = workspace_bytes - 2 * arr;
    head_flags_kernel<<<dim3(grid_cap((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(keys_sorted, n, flags);
    CDR_LAUNCH_CHECK();
    CDR_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, (const uint32_t*)flags, excl, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
    dedup_emit_kernel<<<dim3(grid_cap((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(keys_sorted, perm, flags, excl, n, world,
                                                                                       (unsigned)local_bits, uniq_index, uniq_local,
                                                                                       occ_to_uniq, counts, n_uniq);
    CDR_LAUNCH_CHECK();
    counts_from_starts_kernel<<<dim3(1), dim3(64), 0, s>>>(counts, world);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_segsum_rows(cdr_ctx* ctx, void* stream, const uint32_t* keys_sorted, const uint32_t* perm, const uint32_t* uniq_index,
                               int64_t n, const float* G, int64_t neg_start, int D, const float* rows, const float* reg_coef,
                               float* out) {
    CDR_CHECK_ARG(ctx && keys_sorted && perm && uniq_index && G && out && n > 0 && D > 0 && (D & 3) == 0);
    CDR_CHECK_ARG(reg_coef == nullptr || rows != nullptr);
    hipStream_t s = (hipStream_t)stream;
    const int lpr = cdr_lpr_for(D);
    const int64_t long_cap = n / (kLongSeg + 1) + 1, piece_cap = n / kPiece + long_cap + 1;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_long = 256, o_piece = o_long + up(sizeof(seg_long) * long_cap), o_cnt = o_piece + up(sizeof(seg_piece) * piece_cap),
                 o_part = o_cnt + up(sizeof(int) * piece_cap), total = o_part + sizeof(float) * (size_t)piece_cap * D;
    void* base = nullptr;
    int rc = cdr_ctx_scratch(ctx, total, &base);
    if (rc != CDR_OK) return rc;
    unsigned* counters = (unsigned*)base;
    seg_long* longs = (seg_long*)((char*)base + o_long);
    seg_piece* pieces = (seg_piece*)((char*)base + o_piece);
    int* pcnt = (int*)((char*)base + o_cnt);
    float* partial = (float*)((char*)base + o_part);
    CDR_HIP(cdr_zero_u32(counters, 4, s));
    const int64_t groups = (n + (kBlock / lpr) - 1) / (kBlock / lpr);
    DISPATCH_LPR(lpr, segsum_head_kernel<L><<<dim3(grid_cap(groups)), dim3(kBlock), 0, s>>>(keys_sorted, perm, uniq_index, n, G, neg_start, D, rows,
                                                                                          reg_coef, out, counters, longs, pieces));
    CDR_LAUNCH_CHECK();
    if (n > kLongSeg) {
        DISPATCH_LPR(lpr, segsum_piece_kernel<L><<<dim3(grid_cap((piece_cap < 16384 ? piece_cap : 16384) / (kBlock / lpr) + 1)), dim3(kBlock), 0, s>>>(
                              D, perm, G, neg_start, counters, pieces, partial, pcnt));
        CDR_LAUNCH_CHECK();
        DISPATCH_LPR(lpr, segsum_long_finish_kernel<L><<<dim3(grid_cap((long_cap < 4096 ? long_cap : 4096) / (kBlock / lpr) + 1)), dim3(kBlock), 0, s>>>(
                              D, uniq_index, rows, reg_coef, counters, longs, partial, pcnt, out));
        CDR_LAUNCH_CHECK();
    }
    return CDR_OK;
}
