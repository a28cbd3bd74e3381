// Small-batch id sort (the reference's default train_batch_size is 2,048 rows: properties/overall.yaml:19).
//
// cdr_sort_ids (rocPRIM radix / merge sort) is a chain of small launches: ~20 us at 2,048 ids, 0.17 ms at a million,
// most of a small step.  A single-workgroup LDS bitonic network (first attempt) is bound by ONE CU's VALU rate: 38 us
// for 4,096 ids.  At these sizes the whole chip is idle, so the sort here is a RANK sort spread over all CUs:
//
//   rank_count_kernel    element i's rank = #{ j : (id_j, j) < (id_i, i) }; a block owns 256 elements x a chunk of j's; the
//                        j's keys sit in LDS, four comparisons per 16-B broadcast read, ~6 VALU instructions each;
//                        n^2 comparisons = 17 M at n = 4,096 = ~2 us of the chip.  Partial counts are added with integer
//                        atomics (order-independent: the result is deterministic).
//   rank_scatter_kernel  keys_out[rank] = id, perm_out[rank] = occurrence; re-zeroes the rank scratch for the next call.
//
// The composite (id, occurrence) order is total, so the result -- keys ascending, equal ids in occurrence order -- is exactly
// the stable radix sort's, and the row-wise apply kernels consume it unchanged.  Up to 4 lists per launch (e.g. the user ids
// and the positive+negative item ids of a BPR step; the user and item ids of a CoNet step), each <= 16,384 ids.
#include "cdr_ranksort.h"

namespace {

__global__ __launch_bounds__(ranksort::kTile) void rank_count_kernel(ranksort::small_sort_args a, uint32_t* __restrict__ rank) {
    __shared__ __attribute__((aligned(16))) uint32_t sh[512];
    ranksort::rank_count_body(a, rank, blockIdx.x, sh);
}

__global__ __launch_bounds__(ranksort::kTile) void rank_scatter_kernel(ranksort::small_sort_args a, uint32_t* __restrict__ rank,
                                                                       uint32_t* __restrict__ keys_out, uint32_t* __restrict__ perm_out) {
    ranksort::rank_scatter_body(a, rank, keys_out, perm_out, blockIdx.x);
}

}  // namespace

extern "C" int cdr_sort_ids_small(void* stream, int nseg, const int64_t* const* ids0, const int64_t* n0, const int64_t* const* ids1,
                                  const int64_t* n1, const int64_t* out_off, uint32_t* keys_out, uint32_t* perm_out,
                                  uint32_t* rank_scratch, int64_t max_id) {
    CDR_CHECK_ARG(keys_out && perm_out && rank_scratch);
    ranksort::small_sort_args a;
    if (!ranksort::plan(a, nseg, ids0, n0, ids1, n1, out_off, max_id)) { cdr_set_error("cdr_sort_ids_small: bad list description"); return CDR_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    rank_count_kernel<<<dim3(a.count_blocks), dim3(ranksort::kTile), 0, st>>>(a, rank_scratch);
    CDR_LAUNCH_CHECK();
    rank_scatter_kernel<<<dim3(a.scatter_blocks), dim3(ranksort::kTile), 0, st>>>(a, rank_scratch, keys_out, perm_out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
