// Negative sampler on device: recbole's `sample_by_user_ids` as used by the cross-domain loaders
// (recbole_cdr/sampler/crossdomain_sampler.py:139-175,187-221): for every (user j, slot m) draw an item uniformly from the
// domain's candidate id ranges and redraw while it is one of the user's used (interacted) items; output laid out k-major,
// neg[j + m*S] = m-th negative of positive j (crossdomain_sampler.py:148-152).
//   candidates = [lo0, hi0) U [lo1, hi1)   (source domain: [1,OI) U [OI+TOI,total) ; target domain: [1,n_items) U {})
//   used items = CSR over users, column ids sorted ascending per user (binary search)
// Counter-based RNG (splitmix64 of seed, element, attempt): reproducible per seed, no state.  The reference draws from
// numpy's global generator, so individual draws are not comparable -- the distribution and the constraints are.
#include "cdr_common.h"
#include "cdr_produce.h"

namespace {

constexpr int kBlock = 256;
using cdr_produce::mix64;
using cdr_produce::draw_uniform;
using cdr_produce::draw_alias;

__global__ __launch_bounds__(kBlock) void neg_sample_kernel(const int64_t* __restrict__ users, int64_t S, int k, int64_t lo0,
                                                            int64_t hi0, int64_t lo1, int64_t hi1,
                                                            const int64_t* __restrict__ indptr,
                                                            const int64_t* __restrict__ indices, uint64_t seed,
                                                            int max_tries, int64_t* __restrict__ out,
                                                            int* __restrict__ fail_flag) {
    const int64_t total = S * (int64_t)k, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride)
        out[e] = draw_uniform(users[e % S], e, lo0, hi0, lo1, hi1, indptr, indices, seed, max_tries, fail_flag);
}

__global__ __launch_bounds__(kBlock) void neg_sample_alias_kernel(const int64_t* __restrict__ users, int64_t S, int k,
                                                                  const int64_t* __restrict__ keys, const float* __restrict__ prob,
                                                                  const int64_t* __restrict__ alias, int64_t n_keys,
                                                                  const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                                  uint64_t seed, int max_tries, int64_t* __restrict__ out,
                                                                  int* __restrict__ fail_flag) {
    const int64_t total = S * (int64_t)k, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride)
        out[e] = draw_alias(users[e % S], e, keys, prob, alias, n_keys, indptr, indices, seed, max_tries, fail_flag);
}

// ---- the loader's batch on the device, in one launch (recbole TrainDataLoader._next_batch_data + _neg_sampling as the cross-domain
// loaders use them, recbole_cdr/data/dataloader.py:114-162; crossdomain_sampler.py:139-175): rows [start, start + S) of the epoch's
// (shuffled) interaction columns, tiled and joined with their sampled negatives in recbole's layouts --
//   pairwise  (k >= 1): out_users[j + m S] = u_j, out_items[j + m S] = i_j, out_neg[j + m S] = m-th negative of u_j          (m < k)
//   pointwise (k >= 1): out_users[j + t S] = u_j (t <= k), out_items[j] = i_j, out_items[S + j + m S] = m-th negative of u_j  (m < k)
//   k == 0            : out_users[j] = users_all[start + j]   (OverlapDataloader's slice of the shuffled id list, dataloader.py:37-52)
// `start` and the number of draws made so far live ON THE DEVICE (cursor[0], cursor[1]) and the workgroup that signs in last on
// cursor[2] advances them (start += S, draws += 1) -- every workgroup reads them before it signs in -- so the launch can sit inside
// a captured training step: each hipGraph replay produces the NEXT batch with fresh negatives and no host involvement.  The draw
// for element e uses seed + draws * 0x85EBCA77C2B2AE63 in the formulas of neg_sample_kernel / neg_sample_alias_kernel.
// Rows past n_rows (the host never asks for them) come out as PAD id 0.
using cdr_produce::batch_jobs;

__global__ __launch_bounds__(kBlock) void batch_produce_kernel(batch_jobs a) {
    cdr_produce::batch_produce_body(a.j[blockIdx.y], blockIdx.x, gridDim.x);     // one grid row per loader (BOTH state: target and source in one launch)
}

// SSCDR's in-loss sampler (sscdr.py:89-118) on the device: for every overlapped id one INTERACTED source-domain entity, uniform over
// the id's interaction list (repeats count, as np.random.choice over the Python list does; an empty list is [0], :98-99,110-111),
// and one NON-interacted one, uniform over the candidate ids [lo0, hi0) U [lo1, hi1) (= range(overlapped) + range(target_num,
// total), 0 included) and redrawn while it is in the list.  hist = CSR over ids, entries sorted ascending per id (repeats kept).
// Counter-based RNG keyed by (seed, call number, element, attempt); the call number may live on the device (calls_dev: read here,
// bumped by cdr_inc_i64 behind the launch) so that a hipGraph replay of the loss draws fresh ids every time.
__global__ __launch_bounds__(kBlock) void sscdr_pair_sample_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t lo0, int64_t hi0,
                                                                   int64_t lo1, int64_t hi1, const int64_t* __restrict__ indptr,
                                                                   const int64_t* __restrict__ indices, uint64_t seed,
                                                                   const int64_t* __restrict__ calls_dev, int max_tries,
                                                                   int64_t* __restrict__ pos_out, int64_t* __restrict__ neg_out,
                                                                   int* __restrict__ fail_flag) {
    const int64_t n0 = hi0 > lo0 ? hi0 - lo0 : 0, n1 = hi1 > lo1 ? hi1 - lo1 : 0, ncand = n0 + n1;
    const uint64_t sd = seed + (calls_dev ? (uint64_t)calls_dev[0] * 0x85EBCA77C2B2AE63ull : 0ull);
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
        const int64_t id = ids[e];
        const int64_t b = indptr[id], en = indptr[id + 1];
        const bool empty = en == b;                          // the reference appends 0 to an empty list: interacted = 0, and 0 is "used"
        const uint64_t r0 = mix64(sd + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
        pos_out[e] = empty ? 0 : indices[b + (int64_t)__umul64hi(r0, (uint64_t)(en - b))];
        auto used = [&](int64_t x) {
            if (empty) return x == 0;
            int64_t l = b, h = en;
            while (l < h) { const int64_t m = (l + h) >> 1; if (indices[m] < x) l = m + 1; else h = m; }
            return l < en && indices[l] == x;
        };
        int64_t pick = -1;
        for (int t = 0; t < max_tries; ++t) {
            const uint64_t r = mix64(sd + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xD1B54A32D192ED03ull);
            const int64_t c = (int64_t)__umul64hi(r, (uint64_t)ncand);
            const int64_t x = c < n0 ? lo0 + c : lo1 + (c - n0);
            if (!used(x)) { pick = x; break; }
        }
        if (pick < 0) {
            // max_tries rejections: the r-th FREE candidate directly (what the redraw loop converges to), walking the DISTINCT used ids
            auto lower = [&](int64_t x) { int64_t l = b, h = en; while (l < h) { const int64_t m = (l + h) >> 1; if (indices[m] < x) l = m + 1; else h = m; } return l; };
            auto distinct_in = [&](int64_t lo, int64_t hi) {     // distinct used ids inside [lo, hi)
                if (hi <= lo) return (int64_t)0;
                if (empty) return (int64_t)((lo <= 0 && 0 < hi) ? 1 : 0);
                int64_t c = 0, last = -1;
                for (int64_t q = lower(lo); q < en && indices[q] < hi; ++q) { if (indices[q] != last) { ++c; last = indices[q]; } }
                return c;
            };
            const int64_t free0 = n0 - distinct_in(lo0, hi0), free1 = n1 - distinct_in(lo1, hi1);
            if (free0 + free1 > 0) {
                const uint64_t r = mix64(sd + (uint64_t)(e + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)max_tries * 0xD1B54A32D192ED03ull);
                int64_t k_ = (int64_t)__umul64hi(r, (uint64_t)(free0 + free1));
                const bool first = k_ < free0;
                int64_t x = first ? lo0 + k_ : lo1 + (k_ - free0);
                const int64_t lo = first ? lo0 : lo1, hi = first ? hi0 : hi1;
                if (empty) { if (lo <= 0 && 0 <= x) ++x; }
                else {
                    int64_t last = -1;
                    for (int64_t q = lower(lo); q < en && indices[q] < hi && indices[q] <= x; ++q) { if (indices[q] != last) { ++x; last = indices[q]; } }
                }
                pick = x;
            } else {
                pick = n0 ? lo0 : lo1;
                if (fail_flag) atomicExch(fail_flag, 1);
            }
        }
        neg_out[e] = pick;
    }
}

}  // namespace

extern "C" int cdr_sscdr_pair_sample(void* stream, const int64_t* ids, int64_t n, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1,
                                     const int64_t* hist_indptr, const int64_t* hist_indices, uint64_t seed, const int64_t* calls_dev,
                                     int64_t* pos_out, int64_t* neg_out, int* fail_flag) {
    CDR_CHECK_ARG(ids && n > 0 && hist_indptr && hist_indices && pos_out && neg_out);
    CDR_CHECK_ARG((hi0 > lo0) || (hi1 > lo1));
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    sscdr_pair_sample_kernel<<<dim3((unsigned)g), dim3(kBlock), 0, (hipStream_t)stream>>>(ids, n, lo0, hi0, lo1, hi1, hist_indptr, hist_indices,
                                                                                        seed, calls_dev, 64, pos_out, neg_out, fail_flag);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_neg_sample_alias(void* stream, const int64_t* users, int64_t S, int k, const int64_t* keys, const float* prob,
                                    const int64_t* alias, int64_t n_keys, const int64_t* used_indptr, const int64_t* used_indices,
                                    uint64_t seed, int64_t* out, int* fail_flag) {
    CDR_CHECK_ARG(users && out && keys && prob && alias && S > 0 && k > 0 && n_keys > 0);
    CDR_CHECK_ARG((used_indptr == nullptr) == (used_indices == nullptr));
    int64_t g = (S * k + kBlock - 1) / kBlock;
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    neg_sample_alias_kernel<<<dim3((unsigned)g), dim3(kBlock), 0, (hipStream_t)stream>>>(users, S, k, keys, prob, alias, n_keys, used_indptr,
                                                                                       used_indices, seed, 64, out, fail_flag);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_neg_sample_uniform(void* stream, const int64_t* users, int64_t S, int k, int64_t lo0, int64_t hi0,
                                      int64_t lo1, int64_t hi1, const int64_t* used_indptr, const int64_t* used_indices,
                                      uint64_t seed, int64_t* out, int* fail_flag) {
    CDR_CHECK_ARG(users && out && S > 0 && k > 0);
    CDR_CHECK_ARG((hi0 > lo0) || (hi1 > lo1));
    CDR_CHECK_ARG((used_indptr == nullptr) == (used_indices == nullptr));
    int64_t g = (S * k + kBlock - 1) / kBlock;
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    neg_sample_kernel<<<dim3((unsigned)g), dim3(kBlock), 0, (hipStream_t)stream>>>(users, S, k, lo0, hi0, lo1, hi1, used_indptr,
                                                                                 used_indices, seed, 64, out, fail_flag);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_batch_produce_jobs(void* stream, const cdr_batch_job* jobs, int n_jobs) {
    CDR_CHECK_ARG(jobs && n_jobs >= 1 && n_jobs <= CDR_BATCH_MAX_JOBS);
    batch_jobs a{};
    int64_t g = 1;
    for (int i = 0; i < n_jobs; ++i) {
        const cdr_batch_job& J = jobs[i];
        CDR_CHECK_ARG(J.users_all && J.cursor && J.out_users && J.S > 0 && J.k >= 0 && J.n_rows > 0);
        if (J.k > 0) {
            CDR_CHECK_ARG(J.items_all && J.out_items && (J.pointwise || J.out_neg));
            CDR_CHECK_ARG((J.used_indptr == nullptr) == (J.used_indices == nullptr));
            if (J.dist == 0) CDR_CHECK_ARG((J.hi0 > J.lo0) || (J.hi1 > J.lo1));
            else CDR_CHECK_ARG(J.dist == 1 && J.keys && J.prob && J.alias && J.n_keys > 0);
        }
        const int T = J.k == 0 ? 1 : (J.pointwise ? 1 + J.k : J.k);
        const int64_t gi = (J.S * T + kBlock - 1) / kBlock;
        if (gi > g) g = gi;
        a.j[i] = J;
    }
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    batch_produce_kernel<<<dim3((unsigned)g, (unsigned)n_jobs), dim3(kBlock), 0, (hipStream_t)stream>>>(a);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}

extern "C" int cdr_batch_produce(void* stream, const int64_t* users_all, const int64_t* items_all, int64_t n_rows, int64_t* cursor,
                                 int64_t S, int k, int pointwise, int dist, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1,
                                 const int64_t* keys, const float* prob, const int64_t* alias, int64_t n_keys,
                                 const int64_t* used_indptr, const int64_t* used_indices, uint64_t seed, int64_t* out_users,
                                 int64_t* out_items, int64_t* out_neg, int* fail_flag) {
    cdr_batch_job J{};
    J.users_all = users_all; J.items_all = items_all; J.n_rows = n_rows; J.cursor = cursor; J.S = S; J.k = k; J.pointwise = pointwise;
    J.dist = dist; J.lo0 = lo0; J.hi0 = hi0; J.lo1 = lo1; J.hi1 = hi1; J.keys = keys; J.prob = prob; J.alias = alias; J.n_keys = n_keys;
    J.used_indptr = used_indptr; J.used_indices = used_indices; J.seed = seed; J.out_users = out_users; J.out_items = out_items;
    J.out_neg = out_neg; J.fail_flag = fail_flag;
    return cdr_batch_produce_jobs(stream, &J, 1);
}
