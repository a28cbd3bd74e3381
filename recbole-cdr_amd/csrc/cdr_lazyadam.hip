// Exact dense Adam, evaluated lazily per row.
//
// The reference's optimizer is torch.optim.Adam over whole embedding tables (recbole Trainer): every row moves
// every step through its momentum, an O(table) sweep -- 74 M elements, 2.1 GB, 0.38 ms per step at BASELINE C3, where the
// batch touches 8 k of 290 k rows.  But a row that receives NO gradient in step tau evolves by a closed recurrence of its own
// (m, v, w) and the step number only:
//        m <- m + (0 - m)(1 - b1) ;  v <- b2 v ;  w <- w - step_size(tau) * m / (sqrt(v) / bc2_sqrt(tau) + eps)
// so those updates can be POSTPONED until the row is next needed and then replayed, in order, in registers: same operations,
// same order, same operands => bit-identical to the dense sweep (cdr_adam_math.h keeps the arithmetic shared and
// contraction-free; the per-step scalars come from one table filled by the same expression the dense kernel evaluates).
//
//   last[row]           the update number the row's (w, m, v) currently reflect
//   lz_prepare_kernel   before the forward pass: every distinct row of the batch replays updates last+1 .. t-1 (t = the update
//                       about to happen), so the gather reads exactly what the dense optimizer would have left there
//   lz_apply_kernel     after the backward pass: update t with the row's summed gradient (occurrence order: deterministic)
//   lz_flush_kernel     brings EVERY row to the current update (evaluation, state_dict, checkpoint): the postponed sweeps,
//                       paid once per use instead of once per step
// The per-update scalars (step size, sqrt of the second bias correction) live in a RING of hp_capacity entries indexed by
// update number & (hp_capacity - 1): a row may be at most hp_capacity - 1 updates behind, which the host guarantees by flushing
// every table at least that often (lazyadam.DeferredRowAdam) -- no bound on the number of updates of a run.
// Memory per step: 3 row reads + 3 row writes per touched row (twice: prepare + apply) instead of 7 x the table.
#include "cdr_common.h"
#include "cdr_adam_math.h"
#include "cdr_produce.h"
#include "cdr_ranksort.h"
#include <stdlib.h>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxTab = 4;

struct lz_table {
    float* W; float* M; float* V; int32_t* last;
    const uint32_t* keys; const uint32_t* perm; int64_t n;          // sorted ids of the batch's occurrences for this table
    const float* G; int64_t ldg;                                    // gradient of occurrence o: G[o * ldg .. + D)
};
struct lz_args { lz_table t[kMaxTab]; int count, D; float lr, b1, b2, eps, wd; int64_t hp_mask; };   // hp is a ring of hp_mask + 1 entries

inline int grid_for(int64_t units, int per_block) {
    int64_t g = (units + per_block - 1) / per_block;
    const int64_t cap = CDR_NUM_CU * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

// replay updates (from, to] of one row chunk without gradient
__device__ __forceinline__ void replay(float4& w, float4& m, float4& v, int64_t from, int64_t to, const float2* __restrict__ hp,
                                       const lz_args& a) {
    // a chunk whose moments are all zero (a row that never received a gradient) is a fixed point of the gradient-free update when there
    // is no weight decay: m and v stay 0 and w moves by step_size * 0 / eps = 0 -- nothing to replay, bit for bit.  (Without this the
    // periodic flush replays tens of thousands of updates for every untouched row: 2.9 s per 32,768 steps at C3's table sizes.)
    if (a.wd == 0.f && m.x == 0.f && m.y == 0.f && m.z == 0.f && m.w == 0.f && v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f) return;
    // The per-update scalars are read EIGHT updates at a time, the next eight requested before this eight are applied: with real
    // (non-repeating) batches a row is tens to hundreds of updates behind, and one dependent L2 round trip per update (~0.3 us) was
    // most of the replay (round 4: lz_prepare_kernel 58 us at C3 with the arithmetic already on v_sqrt / v_rcp).  Entries past `to`
    // are read (any ring slot is readable) and not used.
    constexpr int CH = 8;
    float2 hn[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) hn[j] = hp[(from + 1 + j) & a.hp_mask];
    for (int64_t tau = from + 1; tau <= to; tau += CH) {
        float2 h[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) h[j] = hn[j];
        if (tau + CH <= to) {
#pragma unroll
            for (int j = 0; j < CH; ++j) hn[j] = hp[(tau + CH + j) & a.hp_mask];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (tau + j > to) break;
            w.x = cdr_adam_elem(w.x, 0.f, m.x, v.x, a.b1, a.b2, a.eps, a.wd, h[j].x, h[j].y);
            w.y = cdr_adam_elem(w.y, 0.f, m.y, v.y, a.b1, a.b2, a.eps, a.wd, h[j].x, h[j].y);
            w.z = cdr_adam_elem(w.z, 0.f, m.z, v.z, a.b1, a.b2, a.eps, a.wd, h[j].x, h[j].y);
            w.w = cdr_adam_elem(w.w, 0.f, m.w, v.w, a.b1, a.b2, a.eps, a.wd, h[j].x, h[j].y);
        }
    }
}

// counters[0] = updates completed (t_done), counters[1] = the update in progress (t_cur), written here for lz_apply_kernel
template <int LPR>
__global__ __launch_bounds__(kBlock) void lz_prepare_kernel(lz_args a, float2* __restrict__ hp, int64_t* __restrict__ counters) {
    constexpr int GPB = kBlock / LPR;
    const lz_table tb = a.t[blockIdx.y];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D = a.D, D4 = D >> 2;
    const int64_t t = counters[0] + 1;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float ss, bc;
        cdr_adam_hp((double)t, a.lr, a.b1, a.b2, ss, bc);           // read by lz_apply_kernel (next launch) and by later replays
        hp[t & a.hp_mask] = make_float2(ss, bc);
        counters[1] = t;
    }
    for (int64_t q = gg; q < tb.n; q += TG) {
        const uint32_t row = tb.keys[q];
        if (q > 0 && tb.keys[q - 1] == row) continue;                // one lane group per DISTINCT row
        const int64_t from = tb.last[row];
        if (from >= t - 1) continue;
        // (Tried: one or two elements per lane instead of a float4 and four updates' scalars per request, to shorten the dependent chain
        //  of the most-postponed row: 17.5 -> 18.7 us.  The launch moves ~60 MB of 512-byte row segments: it is at the HBM rate.)
        for (int ch = sub; ch < D4; ch += LPR) {
            const int64_t o = (int64_t)row * D + 4 * ch;
            float4 w = ld4(tb.W + o), m = ld4(tb.M + o), v = ld4(tb.V + o);
            replay(w, m, v, from, t - 1, hp, a);
            st4(tb.W + o, w); st4(tb.M + o, m); st4(tb.V + o, v);
        }
        if (sub == 0) tb.last[row] = (int32_t)(t - 1);
    }
}

// ---- the same, ONE element per lane (round 4) ------------------------------------------------------------------------------------
// With real batches the kernel's time is the serial replay of its most-postponed row: a C3 user row comes up every ~60 steps on
// average and the worst of a batch's ~1,600 is ~400 updates behind.  A lane that owns a float4 issues 4 x 16 instruction slots per
// update (256 cycles per update on its SIMD): 400 updates = 100 k cycles = 45 us, whatever the rest of the chip does.  One element
// per lane makes that 64 cycles per update -- the chip-wide VALU work is unchanged, the critical path a quarter.  Updates are applied
// eight at a time in straight-line code (only m' = m - m (1 - b1) and v' = b2 v are carried from update to update; the eight
// sqrt / rcp chains behind them overlap), with the gradient-free arithmetic written out (the terms that multiply g = 0 add +0).
__device__ __forceinline__ float lz_elem_nograd(float pv, float& m, float& v, float b1, float b2, float eps, float step_size, float bc2) {
#pragma clang fp contract(off)
    // cdr_adam_elem with gv = 0, wd = 0:  (0 - m) = -m exactly;  ((1 - b2) * 0) * 0 = +0 and b2 v + 0 = b2 v exactly (v >= 0)
    const float mv = m + (0.f - m) * (1.0f - b1);
    const float vv = b2 * v;
    m = mv; v = vv;
    return pv - cdr_adam_term(mv, vv, step_size, bc2, eps);
}

__global__ __launch_bounds__(kBlock) void lz_prepare1_kernel(lz_args a, float2* __restrict__ hp, int64_t* __restrict__ counters, int lanes_per_row) {
    const lz_table tb = a.t[blockIdx.y];
    const int rows_per_block = kBlock / lanes_per_row;
    const int sub = threadIdx.x % lanes_per_row;
    const int64_t TG = (int64_t)gridDim.x * rows_per_block;
    const int D = a.D;
    const int64_t t = counters[0] + 1;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float ss, bc;
        cdr_adam_hp((double)t, a.lr, a.b1, a.b2, ss, bc);
        hp[t & a.hp_mask] = make_float2(ss, bc);
        counters[1] = t;
    }
    // (Tried on the C3 stream, 41.7 us as written: scalar loop control via readfirstlane 46.5 us; eight rows in flight per lane group
    //  46.9 us; the four tables interleaved over a 1-D grid 58.3 us.  ~20 us of it is the replay's VALU time at full chip occupancy.
    //  Round 5: s_setprio 1..3 by the row's lag, so that the most-postponed rows' waves win their SIMD's issue slots: 42.05 -> 42.27 us.
    //  The launch is the sum of the replays, not the longest of them.)
    // A row wider than 64 elements is spread over several WAVES, which all read last[row] and of which one moves it: every iteration
    // of the (block-uniform) loop therefore has a barrier between the reads and that write.  (Without it a wave that starts its
    // iteration after its sibling has finished finds last[row] already advanced and leaves its part of the row unreplayed -- which is
    // what the first version of this kernel did: rare, timing-dependent, found by comparing the trained state of two invocations.)
    const bool wide = lanes_per_row > 64;
    const int64_t to = t - 1;
    for (int64_t base = (int64_t)blockIdx.x * rows_per_block; base < tb.n; base += TG) {
        const int grp = threadIdx.x / lanes_per_row;                     // (192 lanes per row leave 64 threads of the block without a row)
        const int64_t q = base + grp;
        uint32_t row = 0;
        int64_t from = to;
        if (grp < rows_per_block && q < tb.n) {
            row = tb.keys[q];
            if (!(q > 0 && tb.keys[q - 1] == row)) from = tb.last[row];   // one lane group per DISTINCT row
        }
        if (wide) __syncthreads();
        if (from >= to) continue;
        if (sub < D) {
            const int64_t o = (int64_t)row * D + sub;
            float w = tb.W[o], m = tb.M[o], v = tb.V[o];
            if (a.wd != 0.f || m != 0.f || v != 0.f) {               // (zero moments without weight decay: a fixed point, as in replay())
                constexpr int CH = 8;
                int64_t tau = from + 1;
                if (a.wd == 0.f) {
                    for (; tau + CH - 1 <= to; tau += CH) {          // eight updates, straight line
                        float2 h[CH];
#pragma unroll
                        for (int j = 0; j < CH; ++j) h[j] = hp[(tau + j) & a.hp_mask];
#pragma unroll
                        for (int j = 0; j < CH; ++j) w = lz_elem_nograd(w, m, v, a.b1, a.b2, a.eps, h[j].x, h[j].y);
                    }
                    for (; tau <= to; ++tau) { const float2 h = hp[tau & a.hp_mask]; w = lz_elem_nograd(w, m, v, a.b1, a.b2, a.eps, h.x, h.y); }
                } else {
                    for (; tau <= to; ++tau) { const float2 h = hp[tau & a.hp_mask]; w = cdr_adam_elem(w, 0.f, m, v, a.b1, a.b2, a.eps, a.wd, h.x, h.y); }
                }
                tb.W[o] = w; tb.M[o] = m; tb.V[o] = v;
            }
        }
        if (sub == 0) tb.last[row] = (int32_t)to;
    }
}

// ---- the same, TWO elements per lane, the ring scalars staged in LDS (round 6) -----------------------------------------------------
// lz_prepare1_kernel's inner loop is 135 instructions per eight updates of one element, of which 47 fetch the eight ring entries (64-bit
// index arithmetic per lane and eight global loads) and 16 are the quarter-rate v_sqrt / v_rcp: ~90 SIMD cycles per update of 64 elements.
// Here (a) a workgroup copies the window of the ring its rows need -- the last max-lag entries, the same for every row up to where each
// starts -- into LDS once, and a chunk's eight entries are eight ds_read_b64 off ONE address register with immediate offsets; (b) a lane
// owns two neighbouring elements as a 2-vector, so every add / multiply of the update is one v_pk_*_f32 for both (the sqrt / rcp stay one
// per element): ~53 cycles per update of 64 elements.  A D = 128 row is one wave again -- no barrier for the row's `last`, no second wave
// to race with.  Element by element the operations are lz_elem_nograd's, in its order, contraction off: bit-identical rows.
typedef float lz_f2 __attribute__((ext_vector_type(2)));
constexpr int kWin = 1024;                                              // ring entries staged per workgroup (8 KB); a longer lag starts from global memory

__device__ __forceinline__ lz_f2 lz_elem2_nograd(lz_f2 pv, lz_f2& m, lz_f2& v, float b1, float b2, float eps, float step_size, float bc2) {
#pragma clang fp contract(off)
#ifdef CDR_ADAM_IEEE
    lz_f2 r;
    float mx = m.x, my = m.y, vx = v.x, vy = v.y;
    r.x = lz_elem_nograd(pv.x, mx, vx, b1, b2, eps, step_size, bc2);
    r.y = lz_elem_nograd(pv.y, my, vy, b1, b2, eps, step_size, bc2);
    m.x = mx; m.y = my; v.x = vx; v.y = vy;
    return r;
#else
    const lz_f2 mv = m + (-m) * (1.0f - b1);                             // (0 - m) and -m differ for m = +-0 only, and m + (+-0)(1 - b1) is m's +0 either way
    const lz_f2 vv = b2 * v;
    m = mv; v = vv;
    lz_f2 sq, rc;
    sq.x = __builtin_amdgcn_sqrtf(vv.x); sq.y = __builtin_amdgcn_sqrtf(vv.y);
    const lz_f2 denom = sq * bc2 + eps;                                  // cdr_adam_term, both elements at once
    rc.x = __builtin_amdgcn_rcpf(denom.x); rc.y = __builtin_amdgcn_rcpf(denom.y);
    return pv - step_size * (mv * rc);
#endif
}

// CLAIM (cdr_lazy_adam_prepare_sort_small): the batch's ids come UNSORTED, as the lists the small rank sort takes, and a row is taken by
// whichever of its occurrences exchanges `to` into last[row] first (the winner gets the row's old update number, the others read `to`
// and have nothing to do: same rows replayed over the same updates whoever wins) -- so the replay no longer waits for the id sort, and
// the sort's counting pass runs as further workgroups of THIS launch (ranksort::rank_count_body; its scatter is the next launch, needed
// by the row update only).  The launch is 1-D then: workgroup b < n_prep is workgroup b % gx of table b / gx.
// SWEEP (inside CLAIM): the launch's FIRST workgroups are not the batch's: every update, a window of rows[y] / sweep_period rows of each
// table (the window moves by its own length per update and wraps) is claimed like a batch row and brought up to date.  No row is then
// ever more than ~sweep_period updates behind, which bounds the launch's tail -- a C3 user row comes up every ~95 updates on average,
// the most-postponed one of a batch was 300-700 behind, and its serial replay (114 cycles per update), started wherever the dispatcher
// happened to put it, was up to ~19 us of the launch; the window's rows, all ~sweep_period behind, start first.  The arithmetic is
// conserved (a postponed update is replayed once, by whoever claims the row first); a window row that is also in the batch goes to one of
// the two claimants as any row of the batch does.
struct lz_claim { ranksort::small_sort_args sa; uint32_t* rank; int seg_of[kMaxTab]; unsigned gx, n_prep;
                  unsigned n_sweep, sweep_start[kMaxTab + 1]; int64_t rows[kMaxTab], chunk[kMaxTab]; int period; };
template <bool CLAIM>
__global__ __launch_bounds__(kBlock) void lz_prepare2_kernel(lz_args a, float2* __restrict__ hp, int64_t* __restrict__ counters, int lanes_per_row,
                                                             lz_claim cl) {
    // Dependent memory round trips of a workgroup: {keys, the newest kBlock ring entries} -> {last, W, M, V} -> replay -> stores.  (As first
    // written -- keys -> last -> window of the longest lag -> rows -- a launch over rows that were all less than 48 updates behind still took
    // 22 us: four trips of latency per workgroup and ~4.6 rounds of workgroups per CU.)  The row is requested together with its `last`,
    // before it is known whether it has anything to replay (duplicate occurrences excepted: they are known from the keys).
    // C3 in steady state (rocprofv3, 400 steps): lz_prepare1_kernel 47.8 us, this kernel 36.3 us -- against ~18-26 us of VALU issue time
    // for the step's ~4-6 x 10^5 postponed row-updates (per update of 64 element pairs: 12-13 full-rate instructions and 2 x 2 quarter-rate
    // v_sqrt / v_rcp = ~114 cycles).  Tried and not kept: a window per WAVE and no workgroup barrier for rows of one wave (a lane group
    // on a duplicate occurrence leaves at once instead of waiting at the barriers): 37.9 us -- four times the ring requests.
    __shared__ __attribute__((aligned(16))) float2 win[kWin];           // win[p] = the scalars of update to - kWin + 1 + p
    __shared__ int lag_max[2];
    __shared__ int from_sh[CLAIM ? kBlock : 1];
    if (CLAIM && blockIdx.x >= cl.n_sweep + cl.n_prep) {
        ranksort::rank_count_body(cl.sa, cl.rank, (int)(blockIdx.x - cl.n_sweep - cl.n_prep), reinterpret_cast<uint32_t*>(win));
        return;
    }
    const bool sweep = CLAIM && blockIdx.x < cl.n_sweep;
    unsigned bx = blockIdx.x, by = blockIdx.y, gx = gridDim.x;
    if (CLAIM) {
        if (sweep) {
            by = 0;
            for (int y = 1; y < a.count; ++y) if (blockIdx.x >= cl.sweep_start[y]) by = y;
            bx = blockIdx.x - cl.sweep_start[by]; gx = cl.sweep_start[by + 1] - cl.sweep_start[by];
        } else {
            const unsigned b = blockIdx.x - cl.n_sweep;
            bx = b % cl.gx; by = b / cl.gx; gx = cl.gx;
        }
    }
    lz_table tb = a.t[by];
    const ranksort::small_seg sg = cl.sa.seg[CLAIM ? cl.seg_of[by] : 0];
    if (CLAIM) tb.n = sweep ? cl.chunk[by] : sg.n0 + sg.n1;
    int64_t sweep_first = 0;
    auto id_at = [&](int64_t q) { return sweep ? (uint32_t)(sweep_first + q) : (uint32_t)(q < sg.n0 ? sg.ids0[q] : sg.ids1[q - sg.n0]); };
    auto id_ok = [&](int64_t q) { return !sweep || sweep_first + q < cl.rows[by]; };
    const int rows_per_block = kBlock / lanes_per_row;
    const int sub = threadIdx.x % lanes_per_row;
    const int grp = threadIdx.x / lanes_per_row;
    const int64_t TG = (int64_t)gx * rows_per_block;
    const int D = a.D, D2 = D >> 1;
    const int64_t t = counters[0] + 1;
    if (bx == 0 && by == 0 && threadIdx.x == 0) {
        float ss, bc;
        cdr_adam_hp((double)t, a.lr, a.b1, a.b2, ss, bc);
        hp[t & a.hp_mask] = make_float2(ss, bc);
        counters[1] = t;
    }
    const int64_t to = t - 1;
    const int64_t w0 = to - kWin + 1;
    if (sweep) sweep_first = (to % cl.period) * cl.chunk[by];
    int64_t base = (int64_t)bx * rows_per_block;
    // first trip's keys, then the window's newest entries
    uint32_t row = 0; bool mine = false;
    if (grp < rows_per_block && base + grp < tb.n) {
        const int64_t q = base + grp;
        if (CLAIM) { row = id_at(q); mine = id_ok(q); }                  // (every occurrence tries; one gets the row)
        else {
            row = tb.keys[q];
            mine = !(q > 0 && tb.keys[q - 1] == row);                   // one lane group per DISTINCT row
        }
    }
    {
        const int p = kWin - kBlock + (int)threadIdx.x;
        if (w0 + p >= 1) win[p] = hp[(w0 + p) & a.hp_mask];
    }
    if (threadIdx.x < 2) lag_max[threadIdx.x] = 0;
    int staged = kBlock;                                                // ring entries in `win`, counted from the newest (block-uniform)
    __syncthreads();
    for (int trip = 0; base < tb.n; base += TG, ++trip) {               // block-uniform: barriers inside
        int64_t from = to;
        lz_f2 w = {0.f, 0.f}, m = {0.f, 0.f}, v = {0.f, 0.f};
        const int64_t o = (int64_t)row * D + 2 * sub;
        if (CLAIM) {
            if (sub == 0 && grp < rows_per_block) {
                const int old = mine ? atomicExch(&tb.last[row], (int32_t)to) : (int)to;
                from_sh[grp] = old;
                if (old < (int)to) atomicMax(&lag_max[trip & 1], (int)to - old);
            }
        } else {
            if (mine) {
                from = tb.last[row];
                if (sub < D2) { w = *(const lz_f2*)(tb.W + o); m = *(const lz_f2*)(tb.M + o); v = *(const lz_f2*)(tb.V + o); }
            }
            if (sub == 0 && from < to) atomicMax(&lag_max[trip & 1], (int)(to - from));
        }
        if (threadIdx.x == 0) lag_max[(trip + 1) & 1] = 0;              // last read before the barrier that ended the previous trip
        __syncthreads();                                                 // (also: every wave of a wide row has read `last` before one moves it)
        if (CLAIM) {
            from = grp < rows_per_block ? (int64_t)from_sh[grp] : to;
            if (from < to && sub < D2) { w = *(const lz_f2*)(tb.W + o); m = *(const lz_f2*)(tb.M + o); v = *(const lz_f2*)(tb.V + o); }
        }
        int need = lag_max[trip & 1];
        need = need < kWin ? need : kWin;
        if (need > staged) {
            for (int p = kWin - need + (int)threadIdx.x; p < kWin - staged; p += kBlock) win[p] = hp[(w0 + p) & a.hp_mask];
            staged = need;
            __syncthreads();
        }
        if (from < to) {
            // (zero moments without weight decay: a fixed point, as in replay(); a pair with one such element replays it to itself)
            if (sub < D2 && (a.wd != 0.f || m.x != 0.f || m.y != 0.f || v.x != 0.f || v.y != 0.f)) {
                int64_t tau = from + 1;
                if (a.wd == 0.f) {
                    for (; tau < to - kWin + 1; ++tau) {                 // postponed further than the window holds
                        const float2 h = hp[tau & a.hp_mask];
                        w = lz_elem2_nograd(w, m, v, a.b1, a.b2, a.eps, h.x, h.y);
                    }
                    constexpr int CH = 8;
                    int p = (int)(tau - w0);
                    for (; p + CH <= kWin; p += CH) {                    // eight updates, straight line
                        float2 h[CH];
#pragma unroll
                        for (int j = 0; j < CH; ++j) h[j] = win[p + j];
#pragma unroll
                        for (int j = 0; j < CH; ++j) w = lz_elem2_nograd(w, m, v, a.b1, a.b2, a.eps, h[j].x, h[j].y);
                    }
                    for (; p < kWin; ++p) { const float2 h = win[p]; w = lz_elem2_nograd(w, m, v, a.b1, a.b2, a.eps, h.x, h.y); }
                } else {
                    for (; tau <= to; ++tau) {
                        const float2 h = hp[tau & a.hp_mask];
                        float mx = m.x, my = m.y, vx = v.x, vy = v.y;
                        w.x = cdr_adam_elem(w.x, 0.f, mx, vx, a.b1, a.b2, a.eps, a.wd, h.x, h.y);
                        w.y = cdr_adam_elem(w.y, 0.f, my, vy, a.b1, a.b2, a.eps, a.wd, h.x, h.y);
                        m.x = mx; m.y = my; v.x = vx; v.y = vy;
                    }
                }
                *(lz_f2*)(tb.W + o) = w; *(lz_f2*)(tb.M + o) = m; *(lz_f2*)(tb.V + o) = v;
            }
            if (!CLAIM && sub == 0) tb.last[row] = (int32_t)to;
        }
        // next trip's keys
        row = 0; mine = false;
        if (grp < rows_per_block && base + TG + grp < tb.n) {
            const int64_t q = base + TG + grp;
            if (CLAIM) { row = id_at(q); mine = id_ok(q); }
            else {
                row = tb.keys[q];
                mine = !(q > 0 && tb.keys[q - 1] == row);
            }
        }
        __syncthreads();
    }
}

// PROD (cdr_lazy_adam_apply_produce): the loader's next batch is produced by workgroups behind the update's own in grid row 0 -- the two
// are independent (the update reads the sorted ids and the gradient rows, never the batch buffers) and the producer alone is a 10 us launch
struct lz_prod { cdr_produce::batch_jobs jobs; int gx[CDR_BATCH_MAX_JOBS]; int n; unsigned n_apply; };
template <int LPR, bool PROD>
__global__ __launch_bounds__(kBlock) void lz_apply_kernel(lz_args a, const float2* __restrict__ hp, int64_t* __restrict__ counters, lz_prod ps) {
    constexpr int GPB = kBlock / LPR;
    if (PROD && blockIdx.x >= ps.n_apply) {
        if (blockIdx.y == 0) {
            unsigned b = blockIdx.x - ps.n_apply;
            for (int j = 0; j < ps.n; ++j) {
                if (b < (unsigned)ps.gx[j]) { cdr_produce::batch_produce_body(ps.jobs.j[j], b, (unsigned)ps.gx[j]); return; }
                b -= (unsigned)ps.gx[j];
            }
        }
        return;
    }
    const lz_table tb = a.t[blockIdx.y];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)(PROD ? ps.n_apply : gridDim.x) * GPB;
    const int D = a.D, D4 = D >> 2;
    const int64_t t = counters[1];
    const float2 h = hp[t & a.hp_mask];
    // lanes of this group inside its wave, for the group-wide ballot / shuffles below
    const int gbase = (threadIdx.x & 63) / LPR * LPR;
    const unsigned long long gmask = LPR >= 64 ? ~0ull : ((1ull << LPR) - 1ull);
    for (int64_t q = gg; q < tb.n; q += TG) {
        const uint32_t row = tb.keys[q];
        if (q > 0 && tb.keys[q - 1] == row) continue;
        for (int c0 = 0; c0 < D4; c0 += LPR) {                 // every lane of the group runs every pass (the ballot and shuffles below)
            const int ch = c0 + sub < D4 ? c0 + sub : 0;
            const bool act = c0 + sub < D4;
            const int64_t o = (int64_t)row * D + 4 * ch;
            float4 w = ld4(tb.W + o), m = ld4(tb.M + o), v = ld4(tb.V + o);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            // the row's occurrences, summed in occurrence order.  Walking them one by one is a chain of two dependent global loads per
            // occurrence (key, then permutation entry, then the gradient row: five occurrences of a user = ten round trips, most of this
            // kernel's 18 us); here the group reads LPR keys at once, then its permutation entries at once, then up to eight gradient
            // rows at once -- three round trips for a row with up to eight occurrences.
            for (int64_t e = q;;) {
                const uint32_t kk = e + sub < tb.n ? tb.keys[e + sub] : ~row;
                const unsigned long long hit = (__ballot(kk == row) >> gbase) & gmask;
                const int run = hit == gmask ? LPR : __builtin_ctzll(~hit);          // leading lanes of the group that still see this row
                const uint32_t pe = sub < run ? tb.perm[e + sub] : 0u;
                for (int i0 = 0; i0 < run; i0 += 8) {
                    float4 x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t pi = (uint32_t)__shfl((int)pe, gbase + (i0 + i < run ? i0 + i : 0));
                        x[i] = i0 + i < run ? ld4(tb.G + (int64_t)pi * tb.ldg + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i0 + i < run) { g.x += x[i].x; g.y += x[i].y; g.z += x[i].z; g.w += x[i].w; }
                }
                if (run < LPR) break;
                e += run;
            }
            w.x = cdr_adam_elem(w.x, g.x, m.x, v.x, a.b1, a.b2, a.eps, a.wd, h.x, h.y);
            w.y = cdr_adam_elem(w.y, g.y, m.y, v.y, a.b1, a.b2, a.eps, a.wd, h.x, h.y);
            w.z = cdr_adam_elem(w.z, g.z, m.z, v.z, a.b1, a.b2, a.eps, a.wd, h.x, h.y);
            w.w = cdr_adam_elem(w.w, g.w, m.w, v.w, a.b1, a.b2, a.eps, a.wd, h.x, h.y);
            if (act) { st4(tb.W + o, w); st4(tb.M + o, m); st4(tb.V + o, v); }
        }
        if (sub == 0) tb.last[row] = (int32_t)t;
    }
    // the update is complete once this launch retires: counters[0] is read by the NEXT prepare / flush only (this launch reads [1])
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) counters[0] = t;
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void lz_flush_kernel(lz_args a, int64_t rows, const float2* __restrict__ hp,
                                                          const int64_t* __restrict__ counters) {
    constexpr int GPB = kBlock / LPR;
    const lz_table tb = a.t[0];
    const int sub = threadIdx.x % LPR;
    const int64_t gg = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    const int64_t TG = (int64_t)gridDim.x * GPB;
    const int D = a.D, D4 = D >> 2;
    const int64_t t = counters[0];
    for (int64_t row = gg; row < rows; row += TG) {
        const int64_t from = tb.last[row];
        if (from >= t) continue;
        for (int ch = sub; ch < D4; ch += LPR) {
            const int64_t o = row * D + 4 * ch;
            float4 w = ld4(tb.W + o), m = ld4(tb.M + o), v = ld4(tb.V + o);
            replay(w, m, v, from, t, hp, a);
            st4(tb.W + o, w); st4(tb.M + o, m); st4(tb.V + o, v);
        }
        if (sub == 0) tb.last[row] = (int32_t)t;
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

int fill(lz_args& a, int count, int D, float* const* W, float* const* M, float* const* V, int32_t* const* last,
         const uint32_t* const* keys, const uint32_t* const* perm, const int64_t* n, const float* const* G, const int64_t* ldg,
         float lr, float b1, float b2, float eps, float wd, bool need_grads, int64_t* nmax, int64_t hp_capacity) {
    if (count < 1 || count > kMaxTab || D <= 0 || (D & 3) || !W || !M || !V || !last) return 0;
    if (hp_capacity < 2 || (hp_capacity & (hp_capacity - 1))) return 0;               // a power of two: the ring index is tau & mask
    a = lz_args{};
    a.count = count; a.D = D; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd; a.hp_mask = hp_capacity - 1;
    *nmax = 0;
    for (int i = 0; i < count; ++i) {
        if (!W[i] || !M[i] || !V[i] || !last[i]) return 0;
        a.t[i].W = W[i]; a.t[i].M = M[i]; a.t[i].V = V[i]; a.t[i].last = last[i];
        if (keys) {
            if (!keys[i] || !n || n[i] <= 0) return 0;
            a.t[i].keys = keys[i]; a.t[i].n = n[i];
            if (n[i] > *nmax) *nmax = n[i];
        }
        if (need_grads) {
            if (!perm || !perm[i] || !G || !G[i] || !ldg || ldg[i] < D || (ldg[i] & 3)) return 0;
            a.t[i].perm = perm[i]; a.t[i].G = G[i]; a.t[i].ldg = ldg[i];
        }
    }
    return 1;
}

// CDR_LZ_PREPARE=1: the one-element-per-lane kernel of rounds 4-5 (A/B runs)
inline bool prepare_one_per_lane() {
    static const bool v = [] { const char* e = getenv("CDR_LZ_PREPARE"); return e && e[0] == '1'; }();
    return v;
}

}  // namespace

extern "C" int cdr_lazy_adam_prepare(void* stream, int count, int D, float* const* W, float* const* M, float* const* V,
                                     int32_t* const* last, const uint32_t* const* keys_sorted, const int64_t* n, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, void* hp_table, int64_t hp_capacity,
                                     int64_t* counters, int64_t step_host) {
    CDR_CHECK_ARG(hp_table && counters && step_host >= 1);
    lz_args a; int64_t nmax;
    if (!fill(a, count, D, W, M, V, last, keys_sorted, nullptr, n, nullptr, nullptr, lr, beta1, beta2, eps, weight_decay, false, &nmax, hp_capacity)) {
        cdr_set_error("cdr_lazy_adam_prepare: bad table description"); return CDR_EINVAL;
    }
    if (D % 2 == 0 && D <= 2 * kBlock && !prepare_one_per_lane()) {
        // two elements per lane, ring scalars out of LDS (see lz_prepare2_kernel): a row takes D / 2 lanes, rounded up to a power of two
        // inside a wave or to whole waves
        const int D2 = D / 2;
        int lanes = 1;
        if (D2 <= 64) { while (lanes < D2) lanes *= 2; } else lanes = (D2 + 63) / 64 * 64;
        const int rpb = kBlock / lanes;
        int64_t g = (nmax + rpb - 1) / rpb;
        if (g > CDR_NUM_CU * 32) g = CDR_NUM_CU * 32;
        lz_prepare2_kernel<false><<<dim3((unsigned)g, count), dim3(kBlock), 0, (hipStream_t)stream>>>(a, (float2*)hp_table, counters, lanes, lz_claim{});
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    if (D <= kBlock) {
        // one element per lane: a row takes D lanes rounded up to whole waves (see lz_prepare1_kernel)
        const int lanes = (D + 63) / 64 * 64;
        int64_t g = (nmax + kBlock / lanes - 1) / (kBlock / lanes);
        if (g > CDR_NUM_CU * 32) g = CDR_NUM_CU * 32;
        lz_prepare1_kernel<<<dim3((unsigned)g, count), dim3(kBlock), 0, (hipStream_t)stream>>>(a, (float2*)hp_table, counters, lanes);
        CDR_LAUNCH_CHECK();
        return CDR_OK;
    }
    const int lpr = cdr_lpr_for(D);
    const dim3 grid(grid_for(nmax, kBlock / lpr), count);
    DISPATCH_LPR(lpr, lz_prepare_kernel<L><<<grid, dim3(kBlock), 0, (hipStream_t)stream>>>(a, (float2*)hp_table, counters));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

namespace {
__global__ __launch_bounds__(ranksort::kTile) void lz_rank_scatter_kernel(ranksort::small_sort_args a, uint32_t* __restrict__ rank,
                                                                          uint32_t* __restrict__ keys_out, uint32_t* __restrict__ perm_out) {
    ranksort::rank_scatter_body(a, rank, keys_out, perm_out, blockIdx.x);
}
}  // namespace

// cdr_sort_ids_small(lists) + cdr_lazy_adam_prepare(tables) in TWO launches instead of three, with the replay no longer behind the sort:
// {replay of every table's rows (rows claimed through `last`, see lz_prepare2_kernel<true>) + the sort's counting pass} -> {the sort's scatter}.
// table_list[i] = the list table i's rows are named by.  Same rows, same sorted keys / positions as the two calls.
extern "C" int cdr_lazy_adam_prepare_sort_small(void* stream, int count, int D, float* const* W, float* const* M, float* const* V,
                                                int32_t* const* last, const int* table_list, int nseg, const int64_t* const* ids0,
                                                const int64_t* n0, const int64_t* const* ids1, const int64_t* n1, const int64_t* out_off,
                                                uint32_t* keys_out, uint32_t* perm_out, uint32_t* rank_scratch, int64_t max_id, float lr,
                                                float beta1, float beta2, float eps, float weight_decay, void* hp_table, int64_t hp_capacity,
                                                int64_t* counters, int64_t step_host, const int64_t* table_rows, int sweep_period) {
    CDR_CHECK_ARG(hp_table && counters && step_host >= 1 && table_list && keys_out && perm_out && rank_scratch);
    CDR_CHECK_ARG(sweep_period == 0 || (table_rows && sweep_period >= 2 && sweep_period < hp_capacity));
    CDR_CHECK_ARG(D % 2 == 0 && D <= 2 * kBlock);
    lz_args a; int64_t nmax = 0;
    if (!fill(a, count, D, W, M, V, last, nullptr, nullptr, nullptr, nullptr, nullptr, lr, beta1, beta2, eps, weight_decay, false, &nmax, hp_capacity)) {
        cdr_set_error("cdr_lazy_adam_prepare_sort_small: bad table description"); return CDR_EINVAL;
    }
    lz_claim cl{};
    if (!ranksort::plan(cl.sa, nseg, ids0, n0, ids1, n1, out_off, max_id)) { cdr_set_error("cdr_lazy_adam_prepare_sort_small: bad list description"); return CDR_EINVAL; }
    cl.rank = rank_scratch;
    for (int i = 0; i < count; ++i) {
        CDR_CHECK_ARG(table_list[i] >= 0 && table_list[i] < nseg);
        cl.seg_of[i] = table_list[i];
        const int64_t n = cl.sa.seg[table_list[i]].n0 + cl.sa.seg[table_list[i]].n1;
        if (n > nmax) nmax = n;
    }
    const int D2 = D / 2;
    int lanes = 1;
    if (D2 <= 64) { while (lanes < D2) lanes *= 2; } else lanes = (D2 + 63) / 64 * 64;
    const int rpb = kBlock / lanes;
    int64_t g = (nmax + rpb - 1) / rpb;
    if (g > CDR_NUM_CU * 32) g = CDR_NUM_CU * 32;
    cl.gx = (unsigned)g;
    cl.n_prep = (unsigned)(g * count);
    cl.period = sweep_period > 0 ? sweep_period : 1;
    for (int i = 0; i < count; ++i) {
        cl.sweep_start[i] = cl.n_sweep;
        if (sweep_period > 0) {
            CDR_CHECK_ARG(table_rows[i] > 0);
            cl.rows[i] = table_rows[i];
            cl.chunk[i] = (table_rows[i] + sweep_period - 1) / sweep_period;
            cl.n_sweep += (unsigned)((cl.chunk[i] + rpb - 1) / rpb);
        }
    }
    cl.sweep_start[count] = cl.n_sweep;
    hipStream_t s = (hipStream_t)stream;
    lz_prepare2_kernel<true><<<dim3(cl.n_sweep + cl.n_prep + (unsigned)cl.sa.count_blocks), dim3(kBlock), 0, s>>>(a, (float2*)hp_table, counters, lanes, cl);
    CDR_LAUNCH_CHECK();
    lz_rank_scatter_kernel<<<dim3(cl.sa.scatter_blocks), dim3(ranksort::kTile), 0, s>>>(cl.sa, rank_scratch, keys_out, perm_out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_lazy_adam_apply(void* stream, int count, int D, float* const* W, float* const* M, float* const* V,
                                   int32_t* const* last, const uint32_t* const* keys_sorted, const uint32_t* const* perm,
                                   const int64_t* n, const float* const* G, const int64_t* ldg, float lr, float beta1, float beta2,
                                   float eps, float weight_decay, const void* hp_table, int64_t hp_capacity, int64_t* counters) {
    CDR_CHECK_ARG(hp_table && counters);
    lz_args a; int64_t nmax;
    if (!fill(a, count, D, W, M, V, last, keys_sorted, perm, n, G, ldg, lr, beta1, beta2, eps, weight_decay, true, &nmax, hp_capacity)) {
        cdr_set_error("cdr_lazy_adam_apply: bad table description"); return CDR_EINVAL;
    }
    const int lpr = cdr_lpr_for(D);
    const dim3 grid(grid_for(nmax, kBlock / lpr), count);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_LPR(lpr, lz_apply_kernel<L, false><<<grid, dim3(kBlock), 0, s>>>(a, (const float2*)hp_table, counters, lz_prod{}));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_lazy_adam_apply_produce(void* stream, int count, int D, float* const* W, float* const* M, float* const* V, int32_t* const* last,
                                           const uint32_t* const* keys_sorted, const uint32_t* const* perm, const int64_t* n, const float* const* G,
                                           const int64_t* ldg, float lr, float beta1, float beta2, float eps, float weight_decay,
                                           const void* hp_table, int64_t hp_capacity, int64_t* counters, const cdr_batch_job* jobs, int n_jobs) {
    CDR_CHECK_ARG(hp_table && counters && jobs && n_jobs >= 1 && n_jobs <= CDR_BATCH_MAX_JOBS);
    lz_args a; int64_t nmax;
    if (!fill(a, count, D, W, M, V, last, keys_sorted, perm, n, G, ldg, lr, beta1, beta2, eps, weight_decay, true, &nmax, hp_capacity)) {
        cdr_set_error("cdr_lazy_adam_apply_produce: bad table description"); return CDR_EINVAL;
    }
    lz_prod ps{};
    ps.n = n_jobs;
    const int lpr = cdr_lpr_for(D);
    ps.n_apply = (unsigned)grid_for(nmax, kBlock / lpr);
    int64_t total = ps.n_apply;
    for (int j = 0; j < n_jobs; ++j) {
        const cdr_batch_job& J = jobs[j];
        CDR_CHECK_ARG(J.users_all && J.cursor && J.out_users && J.S > 0 && J.k >= 0 && J.n_rows > 0);
        if (J.k > 0) CDR_CHECK_ARG(J.items_all && J.out_items && (J.pointwise || J.out_neg));
        ps.jobs.j[j] = J;
        ps.gx[j] = (int)cdr_produce::job_grid(J);
        total += ps.gx[j];
    }
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_LPR(lpr, lz_apply_kernel<L, true><<<dim3((unsigned)total, count), dim3(kBlock), 0, s>>>(a, (const float2*)hp_table, counters, ps));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_lazy_adam_flush(void* stream, int D, float* W, float* M, float* V, int32_t* last, int64_t rows, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, const void* hp_table,
                                   int64_t hp_capacity, const int64_t* counters) {
    CDR_CHECK_ARG(hp_table && counters && rows > 0);
    lz_args a; int64_t nmax;
    float* Wp[1] = {W}; float* Mp[1] = {M}; float* Vp[1] = {V}; int32_t* lp[1] = {last};
    if (!fill(a, 1, D, Wp, Mp, Vp, lp, nullptr, nullptr, nullptr, nullptr, nullptr, lr, beta1, beta2, eps, weight_decay, false, &nmax, hp_capacity)) {
        cdr_set_error("cdr_lazy_adam_flush: bad table description"); return CDR_EINVAL;
    }
    const int lpr = cdr_lpr_for(D);
    DISPATCH_LPR(lpr, lz_flush_kernel<L><<<dim3(grid_for(rows, kBlock / lpr)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        a, rows, (const float2*)hp_table, counters));
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
