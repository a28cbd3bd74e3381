// Ordered dense backward (cdr_ordered_bwd): the drop-in losses' dense gradients as occurrence-order sums, one launch, no float
// atomics, no sort.  See include/cdr_hip.h for the term a list entry contributes.  functional.set_deterministic(True) runs it for lists
// of up to 8,192 entries (four times the reference's batch, overall.yaml:19); beyond that the id sort + segmented scatter is faster.
//
// One WAVE per list entry (512-thread workgroups: 8 entries; three of them on a CU, six waves per SIMD).  The workgroup stages the low words of the
// list's ids in LDS (the entry's own index operands are requested first and travel meanwhile).  A wave runs over the list in trips of
// 512 positions: lane l reads the ids at base + 64 m + l, m < 8, so the wave ballot of slice m IS the hit map of 64 consecutive
// positions in list order; a trip without a hit is 8 LDS reads, 8 compares, their OR and one scalar branch.  Hits arrive in ascending
// position: a hit below the entry's own position means an earlier occurrence owns the row and the wave leaves before it has added
// anything; the first occurrence queues the later ones and adds them in list order to a zero, 2 x (64 / lanes-per-row) at a time --
// every lane group of the wave fetches a different occurrence's operands (index loads together, then row loads together), the terms go
// round by lane shuffles and every lane adds them in the same order -- and writes the row with one plain store.
//
// What bounds it (in-kernel clock stamps, profiles/r05_ab_ordered_bwd.txt; one list of 4,096 uniform ids over 3,707 rows, D = 64,
// 18.8 us): descriptor + staging 2.1 us, the scan 5 us on average and 12 us for the slowest wave -- N^2 / 16 lane compares per SIMD
// cycle is nothing, the instruction stream around them (and, for two thirds of the waves at this density, the walk over a trip's hits)
// is what the SIMDs issue -- and 2.6 us for the first occurrences' dependent index -> row loads at the end.  A row with c occurrences
// costs its first occurrence c / 8 rounds of two memory latencies (D <= 64): one item holding 13 % of a 4,096-entry list, 280 us.
#include "cdr_common.h"

namespace {

constexpr int kBlock = 512;        // 8 waves: one wave per list entry; 80 registers: three workgroups (six waves per SIMD) on a CU
constexpr int kEPB = kBlock / 64;   // list entries per workgroup
constexpr int kIT = 512;            // list positions a wave tests per trip
constexpr int kSL = 2;              // occurrences per lane group and round (4: 115 registers, one workgroup per CU less -- slower, r05_ab_ordered_ksl.txt)
constexpr int kQMax = 4 * kSL;      // later occurrences a wave adds per round, at most

struct ord_args {
    cdr_ord_list l[CDR_ORD_MAX_LISTS];
    int total[CDR_ORD_MAX_LISTS];
};

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// a_j * (x - y) + c * r, the expression of the atomic kernels (bpr_bwd_dense_kernel: g * (p - n) + cu * u); go and c are the segment's
__device__ __forceinline__ float4 term_of(const cdr_ord_seg& S, float go, float c, float cf, float4 x, float4 y, float4 r) {
    float4 t = zero4();
    if (S.X) {
        if (S.Y) { x.x -= y.x; x.y -= y.y; x.z -= y.z; x.w -= y.w; }
        const float aj = S.coef ? S.sign * (go * cf) : S.sign;
        t.x = aj * x.x; t.y = aj * x.y; t.z = aj * x.z; t.w = aj * x.w;
    }
    if (c != 0.f) { t.x = t.x + c * r.x; t.y = t.y + c * r.y; t.z = t.z + c * r.z; t.w = t.w + c * r.w; }
    return t;
}

__global__ __launch_bounds__(kBlock) void ordered_bwd_kernel(ord_args a, int D) {
    extern __shared__ uint32_t lo[];
    __shared__ int wide;                                          // some id of the list does not fit 32 bits: confirm hits on 64
    __shared__ cdr_ord_list L;                                    // the list's descriptor: indexed by lane-dependent segment numbers below
    __shared__ float seg_go[CDR_ORD_MAX_SEGS], seg_c[CDR_ORD_MAX_SEGS];      // per segment: go, and the reg coefficient c (0: no reg term)
    __shared__ int seg_end[CDR_ORD_MAX_SEGS];                     // list position one past the segment's last entry
    __shared__ int qpos[kEPB][kQMax];                             // per wave: list positions of later occurrences, ascending
    constexpr int SU = 4;                                         // staging: ids a thread has in flight
    const int N = a.total[blockIdx.y];
    if ((int)blockIdx.x * kEPB >= N) return;                      // uniform: this list is shorter than the launch's longest
    const int Npad = (N + kIT - 1) / kIT * kIT;
    {   // the descriptor into LDS: five waves copy a part each (scalar loads from the kernel arguments, LDS stores by one lane)
        const int cw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if ((threadIdx.x & 63) == 0) {
            const cdr_ord_list& src = a.l[blockIdx.y];
            if (cw == 0) { L.g = src.g; L.g_stride = src.g_stride; L.nseg = src.nseg; L.accumulate = src.accumulate; wide = 0; }
            else if (cw <= CDR_ORD_MAX_SEGS) L.seg[cw - 1] = src.seg[cw - 1];
        }
    }
    __syncthreads();
    // one wave per list entry.  A row is moved by LPRe lanes (a float4 each), so the wave holds G = 64 / LPRe lane groups: the scan uses
    // all 64 lanes, the adding of later occurrences lets every lane group fetch a different occurrence's operands.
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform, and known to be
    const int D4 = D >> 2;
    const int LPRe = D4 <= 16 ? 16 : D4 <= 32 ? 32 : 64;
    const int G = 64 / LPRe, QC = kSL * G;                          // occurrences added per round: 8 (D <= 64), 4 (D <= 128), 2
    const int sub = lane % LPRe, h = lane / LPRe;
    const int e = (int)blockIdx.x * kEPB + wv;
    const bool live = sub < D4;
    const bool mine = e < N;                                      // (waves past the end still help staging the list)

    // the entry's own index operands first: they travel while the list is staged
    int s0 = 0, j0 = mine ? e : 0;
    while (s0 + 1 < L.nseg && j0 >= (int)L.seg[s0].n) { j0 -= (int)L.seg[s0].n; ++s0; }
    float coef0 = 0.f;
    int64_t xi0 = 0, yi0 = 0, key_v = 0;
    if (mine) {
        const cdr_ord_seg& S = L.seg[s0];
        key_v = S.ids[j0];
        if (S.coef) coef0 = S.coef[j0];
        if (S.X) { xi0 = S.xid ? S.xid[j0] : (int64_t)j0; if (S.Y) yi0 = S.yid[j0]; }
    }
    float my_go = 1.0f, my_nrm = 0.f;
    if ((int)threadIdx.x < L.nseg) {
        const cdr_ord_seg& S = L.seg[threadIdx.x];
        if (S.go) my_go = S.go[0];
        if (S.R && S.reg_weight != 0.f) my_nrm = S.norm[0];
    }
    {
        int off = 0;
        bool w = false;
        for (int s = 0; s < L.nseg; ++s) {
            const int n = (int)L.seg[s].n;
            const int64_t* __restrict__ ids = L.seg[s].ids;
            for (int jj0 = threadIdx.x; jj0 < n; jj0 += kBlock * SU) {      // SU loads on their way per thread, then the LDS stores
                int64_t v[SU];
#pragma unroll
                for (int q = 0; q < SU; ++q) { const int j = jj0 + q * kBlock; v[q] = j < n ? ids[j] : 0; }
#pragma unroll
                for (int q = 0; q < SU; ++q) {
                    const int j = jj0 + q * kBlock;
                    if (j < n) { lo[off + j] = (uint32_t)v[q]; w |= ((uint64_t)v[q] >> 32) != 0; }
                }
            }
            off += n;
            if (threadIdx.x == 0) seg_end[s] = off;
        }
        if (threadIdx.x == 0) for (int s = L.nseg; s < CDR_ORD_MAX_SEGS; ++s) seg_end[s] = 0x7fffffff;
        for (int j = N + threadIdx.x; j < Npad; j += kBlock) lo[j] = 0xffffffffu;      // a padding hit fails the j < N test
        if (w) wide = 1;
        if ((int)threadIdx.x < L.nseg) {
            const cdr_ord_seg& S = L.seg[threadIdx.x];
            const float go = my_go * S.go_scale;
            float c = 0.f;
            if (S.R && S.reg_weight != 0.f && my_nrm > 0.f) c = go * S.reg_weight / ((float)S.B * my_nrm);
            seg_go[threadIdx.x] = go; seg_c[threadIdx.x] = c;
        }
    }
    __syncthreads();
    if (!mine) return;                                            // wave-uniform
    const bool confirm = wide != 0;
    const int e0 = seg_end[0], e1 = seg_end[1], e2 = seg_end[2];

    // ... then its rows, which arrive under the scan
    const int64_t key = ((int64_t)__builtin_amdgcn_readfirstlane((int)(key_v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)key_v);
    float4 x0 = zero4(), y0 = zero4(), r0 = zero4();
    {
        const cdr_ord_seg& S = L.seg[s0];
        if (S.X && live) {
            x0 = ld4(S.X + xi0 * S.x_stride + 4 * sub);
            if (S.Y) y0 = ld4(S.Y + yi0 * S.x_stride + 4 * sub);
        }
        if (S.R && live && seg_c[s0] != 0.f) r0 = ld4(S.R + key * S.r_stride + 4 * sub);
    }
    const uint32_t klo = (uint32_t)key;
    float4 acc = zero4();                                         // every lane group keeps the same sum
    bool self_done = false;
    int nq = 0;

    // Trips of kIT = 512 list positions: lane l tests the ids at base + 64 m + l, m < 8, so the wave ballot of slice m IS the hit map of
    // 64 consecutive positions in list order.  Almost every trip has no hit and is 8 LDS reads, 8 compares, their OR and one branch.
    // A trip with hits walks the bits of its maps: a position below the entry's own ends the wave (an earlier occurrence owns the
    // row), its own position is skipped, later ones are queued.  The queue is added in ONE place, when it holds QC positions or at the
    // extra ``tail`` trip: the entry's own term first (always the first of its row's sum), then the later occurrences in list order --
    // lane group h fetches occurrences h and G + h (index loads together, then row loads together), the terms are handed
    // round with lane shuffles and added by every lane in list order.
    const int trips = Npad / kIT;
    for (int it = 0; it <= trips; ++it) {
        uint64_t M[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) M[m] = 0ull;
        // the tight part: run over trips without a hit -- reads, compares, their OR, one scalar branch, nothing else
        for (; it < trips; ++it) {
            const uint32_t* p = lo + it * kIT + lane;
            uint64_t any = 0ull;
#pragma unroll
            for (int m = 0; m < 8; ++m) { M[m] = __ballot(p[64 * m] == klo); any |= M[m]; }
            if (any != 0ull) break;
        }
        const bool tail = it == trips;
        const int base = it * kIT;
        if (!tail) {
            uint64_t any;
            // a trip with hits: drop the entry's own position; the LOWEST remaining hit decides for most waves -- below the entry's
            // own position an earlier occurrence owns the row and the wave is done
            if (it == e / kIT) {
                const uint64_t own = 1ull << (e & 63);
#pragma unroll
                for (int m = 0; m < 8; ++m) if (m == ((e % kIT) >> 6)) M[m] &= ~own;
            }
            any = 0ull;
            int first = -1;
#pragma unroll
            for (int m = 7; m >= 0; --m) { any |= M[m]; if (M[m] != 0ull) first = base + 64 * m + __builtin_ctzll(M[m]); }
            if (any == 0ull || first >= N) continue;              // (ascending positions: a first hit in the padding means no hit)
            if (first < e && !confirm) return;
        }
        int w = 0;
        uint64_t cur = M[0];                                      // everything below is wave-uniform
        for (;;) {
            bool have = false;
            int j = 0;
            for (;;) {                                            // the next hit of this trip, if any
                if (cur != 0ull) {
                    const int b = __builtin_ctzll(cur);
                    cur &= cur - 1ull;
                    j = base + 64 * w + b;
                    if (j != e && j < N) { have = true; break; }  // the entry's own position is not a hit; padding neither
                } else if (w < 7) {
                    ++w;
                    cur = w == 1 ? M[1] : w == 2 ? M[2] : w == 3 ? M[3] : w == 4 ? M[4] : w == 5 ? M[5] : w == 6 ? M[6] : M[7];
                } else {
                    break;
                }
            }
            if (have) {
                if (j < e) {                                      // an earlier entry with this id (low word): it owns the row
                    bool same = true;
                    if (confirm) {
                        const int s = (j >= e0) + (j >= e1) + (j >= e2);
                        same = L.seg[s].ids[j - (s == 0 ? 0 : s == 1 ? e0 : s == 2 ? e1 : e2)] == key;
                    }
                    if (same) return;                             // wave-uniform
                    continue;
                }
                if (lane == 0) qpos[wv][nq] = j;
                ++nq;
            }
            if (nq == QC || (tail && !have)) {
                if (!self_done) {
                    const float4 t = term_of(L.seg[s0], seg_go[s0], seg_c[s0], coef0, x0, y0, r0);
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;       // lane groups > 0 hold the same operands: same sum
                    self_done = true;
                }
                if (nq > 0) {
                    int qs[kSL], qj[kSL];
                    float qcf[kSL];
                    int64_t qxi[kSL], qyi[kSL];
                    bool ok[kSL];
#pragma unroll
                    for (int i = 0; i < kSL; ++i) {                 // index loads: this lane group's (up to) four occurrences
                        const int z = i * G + h;
                        ok[i] = z < nq;
                        const int p = ok[i] ? qpos[wv][z] : 0;
                        qs[i] = (p >= e0) + (p >= e1) + (p >= e2);
                        qj[i] = p - (qs[i] == 0 ? 0 : qs[i] == 1 ? e0 : qs[i] == 2 ? e1 : e2);
                        qcf[i] = 0.f; qxi[i] = qyi[i] = 0;
                        if (ok[i]) {
                            const cdr_ord_seg& S = L.seg[qs[i]];
                            if (confirm && S.ids[qj[i]] != key) ok[i] = false;           // equal low words only: not this row
                            if (S.coef) qcf[i] = S.coef[qj[i]];
                            if (S.X) { qxi[i] = S.xid ? S.xid[qj[i]] : (int64_t)qj[i]; if (S.Y) qyi[i] = S.yid[qj[i]]; }
                        }
                    }
                    float4 t[kSL];
                    {
                        float4 x[kSL], y[kSL], r[kSL];
#pragma unroll
                        for (int i = 0; i < kSL; ++i) {             // row loads
                            x[i] = y[i] = r[i] = zero4();
                            if (ok[i] && live) {
                                const cdr_ord_seg& S = L.seg[qs[i]];
                                if (S.X) { x[i] = ld4(S.X + qxi[i] * S.x_stride + 4 * sub); if (S.Y) y[i] = ld4(S.Y + qyi[i] * S.x_stride + 4 * sub); }
                                if (seg_c[qs[i]] != 0.f) r[i] = ld4(S.R + key * S.r_stride + 4 * sub);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < kSL; ++i)
                            t[i] = ok[i] ? term_of(L.seg[qs[i]], seg_go[qs[i]], seg_c[qs[i]], qcf[i], x[i], y[i], r[i]) : zero4();
                    }
#pragma unroll
                    for (int i = 0; i < kSL; ++i) {                 // list order: occurrence i G + g comes from lane group g
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (g < G && i * G + g < nq) {        // uniform
                                const int src = sub + g * LPRe;
                                const float tx = __shfl(t[i].x, src, 64), ty = __shfl(t[i].y, src, 64);
                                const float tz = __shfl(t[i].z, src, 64), tw = __shfl(t[i].w, src, 64);
                                const bool okg = __shfl((int)ok[i], src, 64) != 0;
                                if (okg) { acc.x += tx; acc.y += ty; acc.z += tz; acc.w += tw; }
                            }
                        }
                    }
                    nq = 0;
                }
            }
            if (!have) break;
        }
    }
    if (live && h == 0) {
        float* row = L.g + key * L.g_stride + 4 * sub;
        if (L.accumulate) { const float4 p = ld4(row); acc.x = p.x + acc.x; acc.y = p.y + acc.y; acc.z = p.z + acc.z; acc.w = p.w + acc.w; }
        st4(row, acc);
    }
}

}  // namespace

extern "C" int cdr_ordered_bwd(void* stream, int D, const cdr_ord_list* lists, int nlists) {
    CDR_CHECK_ARG(lists && nlists >= 1 && nlists <= CDR_ORD_MAX_LISTS);
    CDR_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 256);
    ord_args a{};
    int64_t nmax = 0;
    int used = 0;
    for (int i = 0; i < nlists; ++i) {
        const cdr_ord_list& L = lists[i];
        CDR_CHECK_ARG(L.g && L.g_stride >= D && L.nseg >= 1 && L.nseg <= CDR_ORD_MAX_SEGS);
        int64_t tot = 0;
        for (int s = 0; s < L.nseg; ++s) {
            const cdr_ord_seg& S = L.seg[s];
            CDR_CHECK_ARG(S.n >= 0 && (S.n == 0 || S.ids));
            CDR_CHECK_ARG(!S.X || S.x_stride >= D);
            CDR_CHECK_ARG(!S.Y || (S.X && S.yid));
            CDR_CHECK_ARG(!S.R || (S.r_stride >= D && S.norm && S.B > 0));
            tot += S.n;
        }
        CDR_CHECK_ARG(tot <= CDR_ORD_MAX_TOTAL);
        if (tot == 0) continue;
        a.l[used] = L;
        a.total[used] = (int)tot;
        ++used;
        if (tot > nmax) nmax = tot;
    }
    if (used == 0) return CDR_OK;
    const int grid = (int)((nmax + kEPB - 1) / kEPB);
    const size_t lds = (size_t)((nmax + kIT - 1) / kIT * kIT) * sizeof(uint32_t);
    if (lds > 48 * 1024) {        // past the default dynamic-LDS allowance: gfx950 has 160 KB per CU, ask for what the longest list needs
        CDR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ordered_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(CDR_ORD_MAX_TOTAL * 4 + 1024)));
    }
    ordered_bwd_kernel<<<dim3(grid, used), dim3(kBlock), lds, (hipStream_t)stream>>>(a, D);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
