// Rank sort of small id lists, device side (see cdr_smallsort.hip for the design note).  Bodies are device functions so that
// cdr_kstep.hip can run them as extra blocks of its own launches (horizontal fusion of the small-batch step).
#pragma once
#include "cdr_common.h"

namespace ranksort {

constexpr int kMaxSeg = 4;
constexpr int64_t kMaxSmall = 16384;
constexpr int kTile = 256;

struct small_seg { const int64_t* ids0; int64_t n0; const int64_t* ids1; int64_t n1; int64_t out_off; int blk0, jchunks, jc, sblk0, shift; };
struct small_sort_args { small_seg seg[kMaxSeg]; int nseg, count_blocks, scatter_blocks; };

// element i's rank inside its list = #{ j : (id_j, j) < (id_i, i) }, this block's share: 256 elements x one chunk of j's
__device__ __forceinline__ void rank_count_body(const small_sort_args& a, uint32_t* __restrict__ rank, int bid, uint32_t* sh /* [512] */) {
    int s = 0;
    for (int q = 1; q < a.nseg; ++q) if (bid >= a.seg[q].blk0) s = q;
    const small_seg sg = a.seg[s];
    const int local = bid - sg.blk0;
    const int tile = local / sg.jchunks, chunk = local - tile * sg.jchunks;
    const int n = (int)(sg.n0 + sg.n1), n0 = (int)sg.n0;
    const int i = tile * kTile + threadIdx.x;
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const uint32_t ki = (uint32_t)(ic < n0 ? sg.ids0[ic] : sg.ids1[ic - n0]);
    // the chunk's keys through LDS (padded with the largest key at positions >= n, which can never count), then four
    // comparisons per 16-B broadcast read: a scalar load per j with a full wait made the loop latency-bound
    const int j0 = chunk * sg.jc;
    uint32_t cnt = 0;
    if (sg.shift) {
        // id and occurrence fit one 32-bit word (id << shift | occurrence): ONE unsigned compare per pair, 2 VALU instructions
        for (int t = threadIdx.x; t < sg.jc; t += kTile) {
            const int j = j0 + t;
            sh[t] = j < n ? (((uint32_t)(j < n0 ? sg.ids0[j] : sg.ids1[j - n0]) << sg.shift) | (uint32_t)j) : 0xFFFFFFFFu;
        }
        __syncthreads();
        const uint32_t kc = (ki << sg.shift) | (uint32_t)ic;
        for (int jj = 0; jj < sg.jc; jj += 4) {
            const uint4 k4 = *reinterpret_cast<const uint4*>(sh + jj);
            cnt += (k4.x < kc) ? 1u : 0u;
            cnt += (k4.y < kc) ? 1u : 0u;
            cnt += (k4.z < kc) ? 1u : 0u;
            cnt += (k4.w < kc) ? 1u : 0u;
        }
    } else {
        for (int t = threadIdx.x; t < sg.jc; t += kTile) {
            const int j = j0 + t;
            sh[t] = j < n ? (uint32_t)(j < n0 ? sg.ids0[j] : sg.ids1[j - n0]) : 0xFFFFFFFFu;
        }
        __syncthreads();
        const int before = i - j0;                                   // positions jj < before are occurrences earlier than i
        for (int jj = 0; jj < sg.jc; jj += 4) {
            const uint4 k4 = *reinterpret_cast<const uint4*>(sh + jj);
            cnt += (k4.x < ki || (k4.x == ki && jj + 0 < before)) ? 1u : 0u;
            cnt += (k4.y < ki || (k4.y == ki && jj + 1 < before)) ? 1u : 0u;
            cnt += (k4.z < ki || (k4.z == ki && jj + 2 < before)) ? 1u : 0u;
            cnt += (k4.w < ki || (k4.w == ki && jj + 3 < before)) ? 1u : 0u;
        }
    }
    if (valid && cnt) atomicAdd(&rank[sg.out_off + i], cnt);
}

// keys_out[rank] = id, perm_out[rank] = occurrence; re-zeroes the scratch.  One block per 256 elements (bid over all lists).
__device__ __forceinline__ void rank_scatter_body(const small_sort_args& a, uint32_t* __restrict__ rank, uint32_t* __restrict__ keys_out,
                                                  uint32_t* __restrict__ perm_out, int bid) {
    int s = 0;
    for (int q = 1; q < a.nseg; ++q) if (bid >= a.seg[q].sblk0) s = q;
    const small_seg sg = a.seg[s];
    const int n = (int)(sg.n0 + sg.n1), n0 = (int)sg.n0;
    const int i = (bid - sg.sblk0) * kTile + threadIdx.x;
    if (i < n) {
        const uint32_t r = rank[sg.out_off + i];
        rank[sg.out_off + i] = 0;                                   // ready for the next call
        keys_out[sg.out_off + r] = (uint32_t)(i < n0 ? sg.ids0[i] : sg.ids1[i - n0]);
        perm_out[sg.out_off + r] = (uint32_t)i;
    }
}

// host: fill the argument block; returns 0 on bad arguments
inline int bits_of(int64_t x) { int b = 0; while (b < 63 && ((int64_t)1 << b) <= x) ++b; return b; }

// max_id: an upper bound of every id (0 = unknown).  When id and occurrence index fit 32 bits together the count loop compares
// ONE composite word per pair (3x fewer instructions).
inline int plan(small_sort_args& a, int nseg, const int64_t* const* ids0, const int64_t* n0, const int64_t* const* ids1,
                const int64_t* n1, const int64_t* out_off, int64_t max_id = 0) {
    if (nseg < 1 || nseg > kMaxSeg || !ids0 || !n0 || !out_off) return 0;
    a = small_sort_args{};
    a.nseg = nseg;
    int blocks = 0, sblocks = 0;
    for (int s = 0; s < nseg; ++s) {
        const int64_t m1 = (n1 && ids1 && ids1[s]) ? n1[s] : 0;
        if (!ids0[s] || n0[s] <= 0 || m1 < 0 || n0[s] + m1 > kMaxSmall || out_off[s] < 0) return 0;
        const int64_t n = n0[s] + m1;
        const int jc = n <= 1024 ? 64 : n <= 8192 ? 128 : 512;      // ~1-2 k blocks at the largest sizes, >= 1 wave of work each
        const int tiles = (int)((n + kTile - 1) / kTile), jchunks = (int)((n + jc - 1) / jc);
        const int ib = bits_of(n - 1) > 0 ? bits_of(n - 1) : 1;     // strictly below 32 bits in total: 0xFFFFFFFF stays the padding key
        const int shift = (max_id > 0 && bits_of(max_id) + ib < 32) ? ib : 0;
        a.seg[s] = small_seg{ids0[s], n0[s], m1 ? ids1[s] : nullptr, m1, out_off[s], blocks, jchunks, jc, sblocks, shift};
        blocks += tiles * jchunks;
        sblocks += tiles;
    }
    a.count_blocks = blocks;
    a.scatter_blocks = sblocks;
    return 1;
}

}  // namespace ranksort
