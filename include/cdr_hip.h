/*
 * cdr_hip.h -- C ABI of libcdrhip.so: the MI355X (gfx950) cross-domain recommendation hot path.
 *
 * Drop-in boundary.  The reference (RUCAIBox/RecBole-CDR, 100 % Python) has no FFI; its hot path is the Python class
 * contract CrossDomainRecommender.calculate_loss / predict / full_sort_predict (recbole_cdr/model/
 * crossdomain_recommender.py:14-51) driven by CrossDomainTrainer.fit (recbole_cdr/trainer/trainer.py:43-76).  Every
 * entry point below replaces the stock torch ops the reference issues at the cited file:line; the Python host
 * (recbole-cdr_amd/) binds them with ctypes on torch tensors' data_ptr() and presents the reference's class contract.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.  All tensor pointers are DEVICE pointers owned by the
 *     caller (torch allocations), fp32 row-major contiguous unless a leading dimension is passed; ids are int64
 *     (torch.LongTensor, the dtype recbole's Interaction hands over).
 *   - Every function enqueues on `stream` (a hipStream_t passed as void*) and never synchronises the device.
 *   - Return value: 0 on success, otherwise a negative CDR_E* code or a positive hipError_t; cdr_last_error() returns a
 *     thread-local message.  No exception crosses the boundary.
 *   - The library allocates nothing persistent except what a cdr_ctx owns (reduction scratch).
 *   - Reductions are two-pass with a fixed order (per-block partials -> single finishing block, fp64 accumulate):
 *     results are run-to-run reproducible; only the dense scatter-add (*_bwd_dense, cdr_scatter_add_rows) uses
 *     fp32 atomics, as torch's own embedding backward does.
 */
#ifndef CDR_HIP_H
#define CDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDR_OK 0
#define CDR_EINVAL (-1)   /* bad argument (null pointer, unsupported size)            */
#define CDR_ENOMEM (-2)   /* scratch allocation failed                                */
#define CDR_ENODEV (-3)   /* no gfx950 device / kernel image not loadable             */

typedef struct cdr_ctx cdr_ctx;

/* ---- context ------------------------------------------------------------------------------------------------ */
int cdr_ctx_create(int device, cdr_ctx** out);      /* allocates the reduction scratch on `device`           */
int cdr_ctx_destroy(cdr_ctx* ctx);
const char* cdr_last_error(void);
/* The NEXT forward launch made with this context (cdr_bpr_fwd / cdr_point_fwd / cdr_point_fwd_pair) also zero-fills [ptr, ptr + bytes)
 * (16-byte aligned, a multiple of 16 bytes): the dense [rows, D] gradient buffers the step's backward scatters into -- what
 * autograd's zeros_like + embedding backward do in the reference (emcdr.py:123-131 under loss.backward()) -- cleared under the
 * forward's gathers instead of by a fill launch of their own.  One pending region per context; consumed by that launch. */
int cdr_ctx_scrub_next(cdr_ctx* ctx, void* ptr, size_t bytes);
/* Ids without a sort (round 6; the medium batches of SURVEY 8d's grid, emcdr.py:110-131 at B = 16,448 ... 131,072 triples): hand the context
 * one uint32 occurrence counter per row of the two tables the NEXT cdr_bpr_step_fused / _dev calls train (ALL ZERO on entry; every step leaves
 * them all zero) and a list workspace of cdr_id_count_workspace_bytes(B) bytes.  The step then derives its single-occurrence flags from the
 * counters and sorts only the duplicate occurrences (a few hundred of 196,608 at uniform ids) -- same flags, same duplicate segments in the
 * same occurrence order, bit-equal tables; heads[2] / heads[3] of the step's `heads` buffer report the duplicate occurrences and the largest
 * counter of the batch (the sorted path reports heads[2] only), so that a host can move a heavily skewed stream back to the sorted path
 * (NULL counters).  Row counts are checked against the step's own; a mismatch, a batch outside the range or a short workspace take the sorted path. */
int cdr_ctx_set_id_counters(cdr_ctx* ctx, uint32_t* user_counts, int64_t user_rows, uint32_t* item_counts, int64_t item_rows,
                            void* list_ws, size_t list_ws_bytes);
int cdr_id_count_workspace_bytes(int64_t B, size_t* bytes);
#define CDR_ABI_VERSION 59
int cdr_abi_version(void);                          /* == CDR_ABI_VERSION of the header the library was built from; bumped on any signature change */

/* Optional measurement aid: HIP-event brackets around the hot kernels, recorded on the stream each kernel is launched
 * on.  cdr_timing_enable(ctx, capacity) arms `capacity` slots (0 disarms); every instrumented launch made with this
 * ctx takes the next free slot; cdr_timing_collect waits for the recorded events, returns (tag, milliseconds) per slot
 * in launch order and re-arms.  Tags: */
#define CDR_TAG_BPR_FWD 1
#define CDR_TAG_POINT_FWD 2
#define CDR_TAG_BPR_FWD_GRAD 3
#define CDR_TAG_APPLY_UNSIGNED 4     /* rowwise_apply_kernel<.., SIGNED=false>: user table */
#define CDR_TAG_APPLY_SIGNED 5       /* rowwise_apply_kernel<.., SIGNED=true >: item table (pos + neg occurrences) */
#define CDR_TAG_SORT 6
#define CDR_TAG_POINT_FWD_GRAD 7
#define CDR_TAG_BPR_PARTIAL_DIFF 8
#define CDR_TAG_BPR_GRAD_FROM_DIFF 9
#define CDR_TAG_POINT_PARTIAL_DOT 10
#define CDR_TAG_POINT_GRAD_FROM_DOT 11
#define CDR_TAG_CONET_FWD 12           /* conet_fwd_kernel: gather + every cross unit + output unit + BCE */
#define CDR_TAG_CONET_BWD 13           /* conet_bwd_kernel: data gradients of the towers */
#define CDR_TAG_BPR_FWD_KMAJOR 15      /* bpr_fwd_kmajor_kernel: one lane group per positive */
#define CDR_TAG_MAP_STEP 16             /* map_step_kernel: the OVERLAP step of distinct ids in one pass */
#define CDR_TAG_CONET_WGRAD 14         /* conet_wgrad_kernel: weight gradients, one wave per (tile, row chunk) */
#define CDR_TAG_OCC_FLAGS 17            /* occ_flags_kernel: single-occurrence flags + duplicate-segment heads */
#define CDR_TAG_BPR_FWD_APPLY 18        /* bpr_fwd_apply_kernel: forward + optimizer on the single-occurrence rows */
#define CDR_TAG_BATCH_NORMS 19          /* batch_norms_kernel: EmbLoss norms of the batch's user and positive rows */
#define CDR_TAG_CONET_FB 20             /* conet_fb_kernel: conet_fwd_kernel's and conet_bwd_kernel's passes over a row block in one launch */
int cdr_timing_enable(cdr_ctx* ctx, int capacity);
int cdr_timing_collect(cdr_ctx* ctx, int* tags, float* ms, int max_n, int* n_out);

/* ---- K1+K2+K3+K5: negative-sampled pairwise (BPR) loss -------------------------------------------------------
 * replaces emcdr.py:98-108,119-131,142-154 (source/target_forward x2, BPRLoss, EmbLoss re-gather).
 *   score_pos[b] = <U[uid[b]], I[pid[b]]>, score_neg[b] = <U[uid[b]], I[nid[b]]>
 *   out[1] = mean_b -log(gamma + sigmoid(score_pos - score_neg))          (recbole BPRLoss, gamma = 1e-10)
 *   out[2] = ||U[uid]||_F over the whole batch, repeats included ; out[3] = ||I[pid]||_F   (recbole EmbLoss)
 *   out[0] = out[1] + reg_weight * (out[2] + out[3]) / B
 *   gcoef[b] (optional) = d out[1] / d(score_pos[b] - score_neg[b])  -- consumed by the backward / fused step.
 */
int cdr_bpr_fwd(cdr_ctx* ctx, void* stream,
                const float* user_tab, const float* item_tab, int D,
                const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B,
                float gamma, float reg_weight,
                float* out4, float* gcoef /* [B] or NULL */);

/* dense backward: grad_user_tab / grad_item_tab are full-size [rows, D] buffers the caller zeroed (what autograd
 * hands torch.optim for a sparse=False nn.Embedding).  grad_out = upstream d/d out[0] (device scalar). */
int cdr_bpr_bwd_dense(cdr_ctx* ctx, void* stream,
                      const float* user_tab, const float* item_tab, int D,
                      const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B,
                      const float* gcoef, const float* out4, float reg_weight, const float* grad_out,
                      float* grad_user_tab, float* grad_item_tab);

/* ---- K1+K2+K4+K5: pointwise loss (MSE on the raw dot, or BCE on sigmoid(dot)) ---------------------------------
 * replaces emcdr.py:111-122,134-145 (MF) ; cmf.py:75-99 ; bitgcf.py:221-247 (tables = propagated embeddings,
 * reg tables = ego embeddings).  reg_user_tab/reg_item_tab may alias user_tab/item_tab.
 *   out[1] = loss_kind==MSE ? mean (dot-label)^2 : BCELoss(sigmoid(dot), label) with torch's -100 log clamp
 *   out[2], out[3] = EmbLoss norms of reg_user_tab[uid], reg_item_tab[iid] ; out[0] = out[1] + reg_weight*(..)/B
 *   scores[b] (optional) = dot (MSE) or sigmoid(dot) (BCE) ; gcoef[b] (optional) = d out[1] / d dot[b]
 */
#define CDR_LOSS_MSE 0
#define CDR_LOSS_BCE 1
int cdr_point_fwd(cdr_ctx* ctx, void* stream, int loss_kind,
                  const float* user_tab, const float* item_tab,
                  const float* reg_user_tab, const float* reg_item_tab, int D,
                  const int64_t* uid, const int64_t* iid, const float* label, int64_t B,
                  float reg_weight, float* out4, float* gcoef, float* scores);

int cdr_point_bwd_dense(cdr_ctx* ctx, void* stream,
                        const float* user_tab, const float* item_tab,
                        const float* reg_user_tab, const float* reg_item_tab, int D,
                        const int64_t* uid, const int64_t* iid, int64_t B,
                        const float* gcoef, const float* out4, float reg_weight, const float* grad_out,
                        float* grad_user_tab, float* grad_item_tab,
                        float* grad_reg_user_tab, float* grad_reg_item_tab);
/* Two pointwise batches in ONE launch each way (CMF's two domains on shared tables, cmf.py:81-99; BiTGCF's two stacks): every pointer
 * argument is a host array of two (batch 0, batch 1) with cdr_point_fwd / cdr_point_bwd_dense's meaning, same arithmetic per batch.
 * D % 4 == 0.  total (may be NULL): total[0] = w[0] * out4[0][0] + w[1] * out4[1][0], written by the same finishing block.      */
int cdr_point_fwd_pair(cdr_ctx* ctx, void* stream, int loss_kind, const float* const* user_tab, const float* const* item_tab,
                       const float* const* reg_user_tab, const float* const* reg_item_tab, int D, const int64_t* const* uid,
                       const int64_t* const* iid, const float* const* label, const int64_t* B, const float* reg_weight,
                       float* const* out4, float* const* gcoef, float* const* scores, const float* w, float* total);
/* ... with reg tables reg_D <= D floats wide (reg_D % 4 == 0; reg tables required when reg_D < D): BiTGCF's whole loss value in one launch,
 * BCE on rows of the propagated stacks + reg_weight x EmbLoss of the batch's EGO rows (bitgcf.py:222-240). */
int cdr_point_fwd_pair_ex(cdr_ctx* ctx, void* stream, int loss_kind, const float* const* user_tab, const float* const* item_tab,
                          const float* const* reg_user_tab, const float* const* reg_item_tab, int D, int reg_D, const int64_t* const* uid,
                          const int64_t* const* iid, const float* const* label, const int64_t* B, const float* reg_weight,
                          float* const* out4, float* const* gcoef, float* const* scores, const float* w, float* total);
int cdr_point_bwd_dense_pair(cdr_ctx* ctx, void* stream, const float* const* user_tab, const float* const* item_tab,
                             const float* const* reg_user_tab, const float* const* reg_item_tab, int D, const int64_t* const* uid,
                             const int64_t* const* iid, const int64_t* B, const float* const* gcoef, const float* const* out4,
                             const float* reg_weight, const float* const* grad_out, const float* grad_scale,
                             float* const* grad_user_tab, float* const* grad_item_tab, float* const* grad_reg_user_tab,
                             float* const* grad_reg_item_tab);    /* grad_scale (host [2], NULL = 1): batch d's d loss = grad_out[d][0] * grad_scale[d] */

/* ---- K1: row gather / dense scatter-add ---------------------------------------------------------------------
 * replaces nn.Embedding(idx) (emcdr.py:99-100,159-160 ; conet.py:106-109 ; sscdr.py:138-140 ...) and its dense
 * backward.  out[r,:] = tab[ids[r],:] ;  grad_tab[ids[r],:] += scale * src[r,:]  (scale: device scalar or NULL=1) */
int cdr_gather_rows(void* stream, const float* tab, int D, const int64_t* ids, int64_t n, float* out);
int cdr_scatter_add_rows(void* stream, float* grad_tab, int D, const int64_t* ids, int64_t n,
                         const float* src, const float* scale);
/* the two above for up to four (table, id list) pairs in ONE launch (host arrays of device pointers); bump_counter: optional device
 * int64 advanced by one in the gather's launch (a call counter the launch in front of it read: sscdr.py:166's sampler) */
int cdr_gather_rows_multi(void* stream, int count, const float* const* tabs, int D, const int64_t* const* ids, const int64_t* n,
                          float* const* outs, int64_t* bump_counter);
int cdr_scatter_add_rows_multi(void* stream, int count, float* const* grad_tabs, int D, const int64_t* const* ids, const int64_t* n,
                               const float* const* srcs);

/* K7: mapped-or-target select (emcdr.py:195-197,201-203,222-224 ; sscdr.py:214-216,242-244):
 *   out[r,:] = ids[r] < n_overlap ? mapped[r,:] : tab[ids[r],:]                                               */
int cdr_select_mapped(void* stream, const float* mapped, const float* tab, int D,
                      const int64_t* ids, int64_t n, int64_t n_overlap, float* out);

/* ---- K6/K8/K9: fp32 MFMA contraction with fused epilogue -----------------------------------------------------
 * C[M,N] = epi( op(A) x op(B) ), exact fp32 (v_mfma_f32_32x32x2_f32).  NT form (transB=1) is the all-items scoring
 * matmul user_e x all_item_e^T (emcdr.py:232 ; cmf.py:111 ; bitgcf.py:271 ; sscdr.py:256) and nn.Linear
 * (y = x W^T + b: emcdr.py:86-93 mapping, conet.py:118-137 cross units).
 *   transA: A is stored [K,M] (lda) ; transB: B is stored [N,K] (ldb), else [K,N].
 *   epilogue: v = acc (+ bias[n]) ; v = act(v) ; if accumulate: v += C
 */
#define CDR_ACT_NONE 0
#define CDR_ACT_TANH 1
#define CDR_ACT_RELU 2
#define CDR_ACT_SIGMOID 3
int cdr_gemm_f32(void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                 const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, const float* bias, int act, int accumulate);

/* K9 drop-in: scores[u, 0:n0] = <user_e[u], slab0[j]>, scores[u, n0:n0+n1] = <user_e[u], slab1[j]> -- the item
 * operand given as <= 2 row ranges so the reference's torch.cat copy (emcdr.py:212-214,228-230) never happens. */
int cdr_fullsort_scores_f32(void* stream, const float* user_e, int64_t U, int D,
                            const float* slab0, int64_t n0, const float* slab1, int64_t n1,
                            float* scores /* [U, n0+n1] */);

/* Mask + top-k fused after the scoring contraction (recbole Trainer._full_sort_batch_eval: scores[:,0] =
 * -inf, scores[history_index] = -inf, then Collector's torch.topk): for every user the k largest scores over the columns
 * [0, n0+n1) of cdr_fullsort_scores_f32's matrix, WITHOUT materialising it when U > 32 and D is 64 or 128 (the score tile
 * stays in LDS; only k entries per user and item stripe reach HBM).  hist_indptr [U+1] / hist_cols: the columns to skip
 * per user (ascending inside each user's range), or both NULL.  out_vals descending; out_idx = column index, -1 (with
 * -inf) when fewer than k columns are left.  Ties between equal scores resolve to the one met first.  k <= 64.          */
int cdr_fullsort_topk_workspace_bytes(int64_t U, int D, int64_t n0, int64_t n1, int k, size_t* bytes);
int cdr_fullsort_topk_f32(void* stream, const float* user_e, int64_t U, int D,
                          const float* slab0, int64_t n0, const float* slab1, int64_t n1, int k,
                          const int64_t* hist_indptr, const int64_t* hist_cols, int exclude_first_col,
                          float* out_vals /* [U,k] */, int64_t* out_idx /* [U,k] */, void* workspace, size_t workspace_bytes);

/* SSCDR scoring (sscdr.py:253-259): scores = -(((-2 <u,i>) + |u|^2) + |i|^2) on already-normalised rows.
 * norm_scratch: caller-owned [U + N] floats (row norms are recomputed into it).                               */
int cdr_fullsort_neg_sqdist_f32(void* stream, const float* user_e, int64_t U, int D,
                                const float* items, int64_t N, float* norm_scratch, float* scores);

/* ---- elementwise helpers used by the model mirrors --------------------------------------------------------- */
/* d/dx of act given y = act(x): gx = gy * act'(y)  (tanh: 1-y^2, relu: y>0, sigmoid: y(1-y)) ; in place allowed */
int cdr_act_bwd(void* stream, int act, const float* y, const float* gy, float* gx, int64_t n);
/* column sums: out[n] (+)= sum_m X[m,n]  (bias gradients, batch reductions of per-row parameter-gradient partials): two launches,
 * fixed summation order (slab partials in the context's scratch, added in slab order): bit-reproducible */
int cdr_colsum(cdr_ctx* ctx, void* stream, const float* X, int64_t M, int64_t N, float* out, int accumulate);
/* dW [out, in] = gz^T x and db [out] = column sums of gz (db may be NULL) in ONE launch, for the backward of nn.Linear on small
 * batches (gz [rows, out], x [rows, in] row-major; emcdr.py:86-93, sscdr.py:60-66): fixed-order sums, no float atomics */
int cdr_linear_wgrad_small_workspace(int64_t rows, int dout, int din, size_t* bytes);   /* 0 up to 512 rows */
int cdr_linear_wgrad_small(cdr_ctx* ctx, void* stream, const float* gz, const float* y_out /* or NULL */, int act, const float* x,
                           int64_t rows, int dout, int din, float* dW, float* db, void* workspace, size_t workspace_bytes);
/* (y_out given: gz is the OUTPUT gradient and the kernel forms gz (.) act'(y_out) itself -- no cdr_act_bwd launch in front) */
/* the forward and the input gradient of the same small layers, one wave per 32 x 32 output tile, operands from global memory:
 *   w_is_k_major = 0: C [M, N] = act(A [M, K] W^T + bias), W [N, K] row-major (y = act(x W^T + b));
 *   w_is_k_major = 1: C [M, N] = A [M, K] W, W [K, N] row-major (dx = gz W); bias NULL / act CDR_ACT_NONE there.  K % 4 == 0. */
int cdr_linear_small(void* stream, int w_is_k_major, const float* A, int64_t lda, const float* W, int64_t ldw, int64_t M, int N, int K,
                     const float* bias, int act, float* C, int64_t ldc, const float* a_out /* or NULL: A := A (.) a_act'(a_out), same shape */,
                     int a_act);
/* mean-squared error over all elements + its gradient: out[0] = mean((a-b)^2) ; ga = 2(a-b)/n * grad_out       */
int cdr_mse_fwd(cdr_ctx* ctx, void* stream, const float* a, const float* b, int64_t n, float* out1);
int cdr_mse_bwd(void* stream, const float* a, const float* b, int64_t n, const float* grad_out,
                float* ga /* or NULL */, float* gb /* or NULL */);

/* ---- K13: exact dense Adam (torch.optim.Adam semantics, amsgrad=False, maximize=False) ---------------------- */
int cdr_adam_dense(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step);

/* hipGraph-capturable form: the step count is read from device memory (cdr_inc_i64 bumps it inside the graph) */
int cdr_adam_dense_dev(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, const int64_t* step_dev);
int cdr_inc_i64(void* stream, int64_t* counter);
/* A few loss scalars combined on the device in ONE launch (cmf.py:97-99: alpha * L_s + (1 - alpha) * L_t): mode 0: out[0] = sum_i
 * x[i * x_stride] * w[i], i < n <= 64, added in index order; mode 1 (its backward): out[i] = scale[0] * w[i].                    */
int cdr_scalar_mix(void* stream, int mode, int n, const float* x, int64_t x_stride, const float* w, const float* scale, float* out);
/* the same update for `count` parameter tensors in one launch (+ one launch that bumps their device step counters first):
 * host arrays of device pointers, one entry per tensor.  loss / loss_sum (both or neither): the counter launch also does
 * loss_sum[0] += loss[0] -- recbole Trainer._train_epoch's `total_loss += loss.item()` (recbole_cdr/trainer/trainer.py:59-73 runs
 * that loop) kept on the device, no launch and no host sync of its own. */
int cdr_adam_multi_dev(void* stream, int count, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, int64_t* const* step_dev,
                       float lr, float beta1, float beta2, float eps, float weight_decay, const float* loss, float* loss_sum,
                       unsigned* ticket);
#define CDR_SIGNIN_WORDS 288
/* ticket (optional): CDR_SIGNIN_WORDS zero-initialised device words owned by the caller (a two-level sign-in: eight group counters a cache
 * line apart behind one word; one word was enough up to ABI 53).  With it, updates of up to 512 fat workgroups run as ONE
 * launch: the kernel evaluates update number step + 1 itself and its last workgroup to finish stores the counters (and the loss total). */

/* cdr_gemm_f32 with a per-output-row scale and a pre-activation accumulate -- the CoNet cross unit
 * (conet.py:127-135):  first  C = s W^T + b           (cdr_gemm_f32, act none)
 *                      then   C = relu(C + m (.) (t H^T))   (rowscale = m, act relu, accumulate = 2)
 *   v = rowscale[m] * acc ; accumulate 0: act(v + bias) ; 1: act(v + bias) + C ; 2: act((C + v) + bias)            */
int cdr_gemm_f32_ex(void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                    const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, const float* bias, const float* rowscale, int act, int accumulate);

/* ---- CoNet helpers (conet.py:105-242) ------------------------------------------------------------------------- */
/* out[r*ldo + c] = tab[ids[r]*D + c]  /  grad_tab[ids[r]*D + c] += src[r*lds + c]  (the [u ; i] concatenated input) */
int cdr_gather_rows_ld(void* stream, const float* tab, int D, const int64_t* ids, int64_t n, float* out, int64_t ldo);
int cdr_scatter_add_rows_ld(void* stream, float* grad_tab, int D, const int64_t* ids, int64_t n, const float* src, int64_t lds);
/* the deterministic form: ids sorted by cdr_sort_ids / cdr_sort_ids_small (keys_sorted, perm); every distinct row of the ZEROED
 * grad_tab receives the sum of its occurrences' rows src[perm[e] * lds ..] in occurrence order -- no float atomics */
int cdr_scatter_rows_sorted(void* stream, float* grad_tab, int D, const uint32_t* keys_sorted, const uint32_t* perm, int64_t n,
                            const float* src, int64_t lds);
/* ---- Ordered dense backward: the drop-in losses' DENSE gradients without float atomics and without a sort (round 5) -------------------
 * replaces the scatter half of torch's embedding backward behind emcdr.py:111-154 (BPR / MF), cmf.py:81-99, bitgcf.py:221-247 and the
 * plain nn.Embedding gathers of the other models, at the reference's own batch sizes (overall.yaml:19 train_batch_size = 2,048).
 * A LIST is every occurrence that adds into one gradient buffer, in a fixed order: up to CDR_ORD_MAX_SEGS segments laid end to end
 * (e.g. the item list of a BPR batch = its positives, then its negatives).  Entry j of a segment contributes
 *     a_j * (X[xj] - Y[yid[j]]) + c * R[ids[j]]        a_j = sign * go * coef[j]   (coef NULL: a_j = sign, no go)
 *                                                      xj  = xid ? xid[j] : j      (X NULL: no first term; Y NULL: no subtraction)
 *                                                      c   = go * reg_weight / (B * norm[0])   (R NULL, reg_weight 0 or norm 0: none)
 *     go  = (go_ptr ? go_ptr[0] : 1) * go_scale
 * to row ids[j] of g.  One lane group per entry: the list's ids sit in LDS, the group scans them once in order; an entry with an equal id
 * EARLIER in the list is not the row's first occurrence and leaves; the first occurrence adds the terms of every later equal entry in
 * list order to a zero and writes the row with ONE plain store (accumulate: adds the sum to the row's earlier content instead, still
 * one read-modify-write by one group).  g must be zero where no entry points (the forward launch clears it);
 * two lists of one call must not point at the same rows.  Result: bit-reproducible from run to run, the occurrence-order sum that
 * cdr_scatter_rows_sorted gives, in one launch and O(total^2 / lanes) compares -- meant for total <= CDR_ORD_MAX_TOTAL per list
 * (CDR_EINVAL beyond; callers keep the sorted or the atomic form there).  D % 4 == 0, D <= 256, every row stride in floats. */
#define CDR_ORD_MAX_SEGS 4
#define CDR_ORD_MAX_LISTS 4
#define CDR_ORD_MAX_TOTAL 16384
typedef struct cdr_ord_seg {
    const int64_t* ids; int64_t n;
    const float* coef; float sign;
    const float* go; float go_scale;
    const float* X; const int64_t* xid; const float* Y; const int64_t* yid; int64_t x_stride;
    const float* R; int64_t r_stride; const float* norm; float reg_weight; int64_t B;
} cdr_ord_seg;
typedef struct cdr_ord_list {
    float* g; int64_t g_stride; int32_t nseg; int32_t accumulate;     /* accumulate != 0: row = row + sum (g holds earlier gradients) */
    cdr_ord_seg seg[CDR_ORD_MAX_SEGS];
} cdr_ord_list;
int cdr_ordered_bwd(void* stream, int D, const cdr_ord_list* lists, int nlists);
int cdr_overlap_mask(void* stream, const int64_t* ids, int64_t n, int64_t n_overlap, float* out);   /* id < n ? 1 : 0 */
int cdr_rowscale(void* stream, const float* x, const float* scale, int64_t M, int64_t N, float* out);
int cdr_bcast_add_act(void* stream, const float* P, const float* q, int64_t N, int64_t H, int act, float* out);
/* nn.BCELoss on probabilities (log clamp -100, mean) and its gradient (p-y)/max(p(1-p),1e-12)/n * grad_out */
int cdr_bce_prob_fwd(cdr_ctx* ctx, void* stream, const float* p, const float* y, int64_t n, float* out1);
int cdr_bce_prob_bwd(void* stream, const float* p, const float* y, int64_t n, const float* grad_out, float* gp);
/* torch.norm(W) (Frobenius, conet.py:198-201) and d/dW = grad_out * W / norm */
int cdr_frobenius_fwd(cdr_ctx* ctx, void* stream, const float* x, int64_t n, float* out1);
int cdr_frobenius_bwd(void* stream, const float* x, int64_t n, const float* norm, const float* grad_out, float* gx, int accumulate);

/* ---- per-positive ("k-major") fused BPR step + one-launch small sort + capturable applies --------------------------------
 * recbole's pairwise batch (crossdomain_sampler.py:148-152; emcdr.py:123-131,146-154): S positives tiled k times, negatives
 * k-major: uid[j + m S] = uid[j], pid[j + m S] = pid[j], nid[j + m S] = m-th negative of positive j; B = S k rows.
 *   cdr_bpr_fwd_grad_kmajor : same loss as cdr_bpr_fwd_grad over the B rows (BPRLoss mean over B, EmbLoss over the B repeated
 *       rows), reading uid[0..S), pid[0..S), nid[0..B).  u_j and p_j are gathered ONCE per positive.  Writes GU[S, D]:
 *       GU[j] = sum_m g_{j,m} (p_j - n_{j,m}), and item_rec[S + B] 8-byte records {int32 user row, float coefficient} in the
 *       order of the item occurrence list [pid[0..S) | nid[0..B)]: {uid_j, sum_m g_{j,m}} and {uid_j, -g_{j,m}} -- the gradient
 *       row of an item occurrence is coefficient * user_tab[user row].  out9 as cdr_bpr_fwd_grad's with out9[4], out9[5]
 *       pre-multiplied by k (an S-list occurrence stands for k batch rows).  bump_a / bump_b (optional device int64
 *       counters) are incremented: the two tables' Adam update counts for the applies below.
 *   cdr_sort_ids_small      : up to 4 id lists of <= 16384 ids each (list s = ids0[s][0..n0[s]) ++ ids1[s][0..n1[s]), ids1
 *       optional) as a rank sort over the whole chip (rank = #{(id, occurrence) smaller}, two launches, integer atomics only:
 *       the result equals the stable radix sort of cdr_sort_ids); list s lands at keys_out / perm_out + out_off[s].
 *   cdr_rowwise_apply_rows  : cdr_rowwise_apply for unsigned gradient rows G[n, D] (no negative block), plus: step_dev (optional
 *       device int64: the Adam update count is read on the device -> hipGraph-capturable) and small (1: no long-segment
 *       pass -- no memset, no extra launches; a segment is walked by its head's lane group whatever its length).
 *   cdr_rowwise_apply_scaled: the item table of the k-major step: occurrence o contributes item_rec[o].coef *
 *       src_table[item_rec[o].urow] (src_table = the user table BEFORE its own apply); reg_limit = S.                       */
int cdr_bpr_fwd_grad_kmajor(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D, const int64_t* uid,
                            const int64_t* pid, const int64_t* nid, int64_t S, int k, float gamma, float reg_weight, float* out9,
                            float* GU, void* item_rec, int64_t* bump_a, int64_t* bump_b);
int cdr_sort_ids_small(void* stream, int nseg, const int64_t* const* ids0, const int64_t* n0, const int64_t* const* ids1,
                       const int64_t* n1, const int64_t* out_off, uint32_t* keys_out, uint32_t* perm_out,
                       uint32_t* rank_scratch /* one uint32 per id, ZERO before the first call; left zero by every call */,
                       int64_t max_id /* an upper bound of the ids, or 0: when id and occurrence index fit 32 bits together the
                                         comparison loop works on one composite word (3x fewer instructions) */);
int cdr_rowwise_apply_rows(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D,
                           const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, const float* G, int64_t reg_limit,
                           const float* reg_coef, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                           const int64_t* step_dev, int small);
/* The whole k-major step for small batches (S + S k <= 16384 ids) in FOUR launches behind one call: {forward blocks || rank-count
 * blocks}, {rank-scatter blocks || loss finish}, item apply, user apply.  Same results as cdr_bpr_fwd_grad_kmajor +
 * cdr_sort_ids_small + cdr_rowwise_apply_scaled + cdr_rowwise_apply_rows (small = 1).  Adam: the update counts are the device
 * counters step_*_dev, incremented by the first launch -- the call is hipGraph-capturable.  keys / perm: uint32 [2 S + S k];
 * rank_scratch: uint32 [2 S + S k], zero before the first call.                                                                  */
int cdr_bpr_step_small(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, float* item_tab,
                       float* item_m, float* item_v, int D, const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t S,
                       int k, float gamma, float reg_weight, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t* step_user_dev, int64_t* step_item_dev, float* out9, float* GU, void* item_rec, uint32_t* keys,
                       uint32_t* perm, uint32_t* rank_scratch, int64_t max_rows /* >= rows of both tables, or 0 */);
int cdr_rowwise_apply_scaled(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D,
                             const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, const void* item_rec,
                             const float* src_table, int64_t reg_limit, const float* reg_coef, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, const int64_t* step_dev, int small);

/* ---- EMCDR OVERLAP phase for batches of DISTINCT ids (emcdr.py:156-168 + mapping :59-64,86-93) in two launches ------------------
 * The reference's OverlapDataloader yields slices of a shuffled arange(num_overlap) (data/dataloader.py:37-52): ids are distinct
 * within a batch, so each row is updated by exactly one occurrence and nothing is sorted.  Launch 1: gather S[idx], T[idx] ->
 * mapping (L layers W_l [d_{l+1}, d_l], optional bias, act[l] in {CDR_ACT_NONE, CDR_ACT_TANH}; the last layer has no activation)
 * -> loss = mean((mapping(S[idx]) - T[idx])^2) -> backward -> SGD / Adam on the two rows of every id in place (lazy row-wise
 * Adam, as cdr_rowwise_apply) + per-workgroup partial sums of the mapping's gradients.  Launch 2: partials added in workgroup
 * order, loss_out[0], exact dense Adam on the mapping parameters.  Device update counters: step_src_dev / step_tgt_dev hold the
 * tables' counts BEFORE the step and are advanced by it; step_W / step_b (one int64 per parameter tensor) likewise.
 * PRECONDITION: idx[0..n) pairwise distinct (repeated ids: use the general path).
 * Launch 1 is one of three kernels by shape (same arguments, same results): the linear mapping and the Linear-Tanh-Linear MLP with
 * widths in {64, 128} run as MFMA waves + row waves with LDS-DMA prefetch (csrc/cdr_mapstep.hip); those two take the Adam STEP's two
 * divisions and square root on v_rcp_f32 / v_sqrt_f32 (1 ulp each: the weight differs from the IEEE form by <= 4e-7 x lr; exp_avg and
 * exp_avg_sq are bit-identical); every other shape runs the single-group kernel with IEEE arithmetic throughout.                 */
#define CDR_MAP_MAX_LAYERS 4
int cdr_map_step_plan(int L, const int* dims, const int* has_bias, int64_t n, size_t* workspace_bytes);
int cdr_map_step_unique(cdr_ctx* ctx, void* stream, int opt, float* src_tab, float* src_m, float* src_v, float* tgt_tab,
                        float* tgt_m, float* tgt_v, const int64_t* idx, int64_t n, int L, const int* dims, const int* acts,
                        float* const* W, float* const* bias, float* const* mW, float* const* vW, float* const* mb,
                        float* const* vb, int64_t* const* step_W, int64_t* const* step_b, int64_t* step_src_dev,
                        int64_t* step_tgt_dev, float lr, float beta1, float beta2, float eps, float weight_decay, float* loss_out,
                        void* workspace, size_t workspace_bytes);

/* ---- exact dense Adam evaluated lazily per row (the reference's torch.optim.Adam over whole tables, recbole Trainer) -----------
 * A row without a gradient in update tau evolves by a recurrence of its own (w, m, v) and tau only, so those updates are postponed
 * and replayed, in order, when the row is next needed: bit-identical to the dense sweep of cdr_adam_multi_dev (shared,
 * contraction-free arithmetic), at 3 row reads + 3 row writes per touched row instead of 7 x the table per step.
 *   state per table: W, M, V [rows, D] and last [rows] (int32, zero-initialised: the update each row reflects);
 *   shared: hp_table (float2 [capacity]: step_size, sqrt(bias correction 2) per update, filled on the device),
 *           counters (int64 [2], zero-initialised: [0] updates completed, [1] update in progress).
 *   cdr_lazy_adam_prepare : BEFORE the forward pass: every distinct row of keys_sorted (per table; cdr_sort_ids / _small) replays its
 *                           postponed updates up to the one before the coming update.  step_host = the coming update's number.
 *                           hp_table: float2 [hp_capacity], a RING indexed by update number & (hp_capacity - 1) (hp_capacity a
 *                           power of two): no row may fall hp_capacity updates behind -- the caller flushes at least that often.
 *   cdr_lazy_adam_apply   : AFTER the backward pass: the coming update with gradient sum_occurrences G[perm[e] * ldg .. + D)
 *                           (occurrence order), then counters[0] advances.
 *   cdr_lazy_adam_flush   : every row of one table replays up to counters[0] (evaluation, state_dict, checkpoint).              */
int cdr_lazy_adam_prepare(void* stream, int count, int D, float* const* W, float* const* M, float* const* V, int32_t* const* last,
                          const uint32_t* const* keys_sorted, const int64_t* n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, void* hp_table, int64_t hp_capacity, int64_t* counters, int64_t step_host);
/* cdr_sort_ids_small(lists) + cdr_lazy_adam_prepare(tables) in two launches: {the replay of every table's rows + the sort's counting pass}
 * -> {the sort's scatter}.  The ids are read UNSORTED (the lists cdr_sort_ids_small takes; table_list[i] = the list that names table i's
 * rows) and a row is taken by whichever of its occurrences exchanges the new update number into last[row] first -- same rows replayed
 * over the same updates, same keys_out / perm_out as the two calls; the replay no longer waits for the sort.  D even, <= 512.          */
int cdr_lazy_adam_prepare_sort_small(void* stream, int count, int D, float* const* W, float* const* M, float* const* V, int32_t* const* last,
                                     const int* table_list, int nseg, const int64_t* const* ids0, const int64_t* n0,
                                     const int64_t* const* ids1, const int64_t* n1, const int64_t* out_off, uint32_t* keys_out,
                                     uint32_t* perm_out, uint32_t* rank_scratch, int64_t max_id, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, void* hp_table, int64_t hp_capacity, int64_t* counters, int64_t step_host,
                                     const int64_t* table_rows /* [count]: rows of every table (sweep_period > 0) */,
                                     int sweep_period /* 0: off.  P >= 2: every call also brings a window of rows / P rows of each table up
                                        to date (the window moves on by its length per update), so that no row is ever more than ~P updates
                                        behind: bounds the replay launch's tail; results unchanged (a postponed update is replayed once) */);
int cdr_lazy_adam_apply(void* stream, int count, int D, float* const* W, float* const* M, float* const* V, int32_t* const* last,
                        const uint32_t* const* keys_sorted, const uint32_t* const* perm, const int64_t* n, const float* const* G,
                        const int64_t* ldg, float lr, float beta1, float beta2, float eps, float weight_decay, const void* hp_table,
                        int64_t hp_capacity, int64_t* counters);
int cdr_lazy_adam_flush(void* stream, int D, float* W, float* M, float* V, int32_t* last, int64_t rows, float lr, float beta1,
                        float beta2, float eps, float weight_decay, const void* hp_table, int64_t hp_capacity, const int64_t* counters);

/* ---- CoNet towers fused (conet.py:105-203: source_forward + target_forward + BCELoss x2 + reg) -------------------------
 * One stack of R rows -- rows [0, n_source) are the source batch (user_s, item_s, label_s), the rest the target batch (the two
 * batches stay separate tensors, as calculate_loss receives them; the kernel stacks them) -- runs BOTH towers through the
 * L cross units  s' = relu(s Ws^T + bs + m (.) (t H^T)),  t' = relu(t Wt^T + bt + m (.) (s H^T))  (conet.py:118-137;
 * m = 1 where the user -- or item, overlap_users = 0 -- id is < n_overlap, PAD id 0 included), the output unit of the tower
 * a row belongs to (sigmoid(Linear(d_L, 1)), conet.py:140,179) and nn.BCELoss against label[r]:
 *   out[1] = BCE(source rows), out[2] = BCE(target rows), out[4 + l] = ||H_l||_F, out[3] = sum_l out[4 + l],
 *   out[0] = (out[1] + out[2]) + out[3]                                                         (conet.py:195-201)
 * dims[0..L] = {2 D, mlp_hidden_size...}, every entry a multiple of 4 and <= CDR_CONET_MAX_LAYERS layers; params = host array
 * of 5 L + 4 device pointers {Ws_l [d_{l+1}, d_l], bs_l, Wt_l, bt_l, H_l} for l < L, then {wo_s [1, d_L], bo_s, wo_t, bo_t}.
 * The forward keeps what the backward needs: x0 [R, 4 D] (the gathered [s | t] inputs), acts [R, act_width] (post-ReLU
 * outputs of every layer, act_width = 2 sum_l d_{l+1}), prob [R], maskf [R].
 * cdr_conet_bwd: gx0 [R, 4 D] = d loss / d x0 (columns [0,D) -> source user table row user[r], [D,2D) -> source item table,
 * [2D,3D) -> target user table, [3D,4D) -> target item table: scatter with cdr_scatter_add_rows_ld); grads = host array of
 * 5 L + 4 device pointers laid out as params, every entry overwritten (d||H_l||_F included); gz [R, act_width] is scratch.
 * No float atomics: partial sums are added in a fixed order.  cdr_conet_plan gives act_width and the workspace size.
 * Training steps: cdr_conet_fwd with gz, gx0 and the workspace given (all three or none) also runs the DATA backward of every row
 * block in the same launch, for a unit upstream gradient, and reports it in *data_gradients_done (0: this shape keeps the two-launch
 * route); cdr_conet_bwd is then called with that flag, skips its data pass and applies grad_out (any value) to the results.
 * cdr_conet_defer_finish(ctx, 1): from now on a TRAINING forward on this context does not add its blocks' loss partials itself -- `out`
 * is complete only after the cdr_conet_bwd that follows on the same stream (its weight-gradient launch gets one workgroup more instead
 * of the forward's finishing launch).  For a caller that differentiates every loss at once and reads it afterwards (a captured step:
 * graph_step.GraphedTrainStep); the drop-in default (0) keeps `out` valid when cdr_conet_fwd's launches retire, as conet.py:186-203's
 * caller expects (recbole's Trainer checks the loss for NaN before backward()).  A deferred forward that is followed by another forward
 * instead of its backward is finished first.                                                                                        */
#define CDR_CONET_MAX_LAYERS 8
int cdr_conet_plan(int L, const int* dims, int64_t R, int* act_width, size_t* workspace_bytes);
int cdr_conet_defer_finish(cdr_ctx* ctx, int on);
int cdr_conet_fwd(cdr_ctx* ctx, void* stream, const float* su_tab, const float* si_tab, const float* tu_tab, const float* ti_tab,
                  int D, const int64_t* user_s /* [n_source] */, const int64_t* user_t /* [R - n_source] */, const int64_t* item_s,
                  const int64_t* item_t, int64_t R, int64_t n_source, int64_t n_overlap, int overlap_users, int L, const int* dims,
                  const float* const* params, const float* label_s, const float* label_t, float* x0, float* acts, float* prob,
                  float* maskf, float* label_cat /* [R]: the stacked labels, for cdr_conet_bwd */,
                  int64_t* ids_cat /* [2 R]: the stacked user ids, then the stacked item ids */, float* out /* [4 + L] */,
                  float* gz, float* gx0, void* workspace, size_t workspace_bytes, int* data_gradients_done);
int cdr_conet_bwd(cdr_ctx* ctx, void* stream, int64_t R, int64_t n_source, int L, const int* dims, const float* const* params,
                  const float* label, const float* x0, const float* acts, const float* prob, const float* maskf,
                  const float* out, const float* grad_out /* device scalar or NULL = 1 */, float* gz, float* gx0,
                  float* const* grads, void* workspace, size_t workspace_bytes, int data_gradients_done);

/* ---- CoNet full-sort scoring, all users x all items in one launch (conet.py:222-242: the reference's per-user Python loop over the
 * target tower without cross terms) --------------------------------------------------------------------------------------
 * The first layer is separable, W1 [u ; i] + b1 = Q[u] + P[i] with Q = user_e W1[:, :D]^T + b1 [U, h1] and P = items W1[:, D:]^T
 * [N, h1] (two cdr_gemm_f32_ex calls); this entry does the rest for every (u, i):
 *      out[u, i] = sigmoid(wo . relu(W_T ... relu(W_1 relu(P[i] + Q[u]) + b_1) ... + b_T) + bo)
 * W[t]: [tail_dims[t], d_in] row-major (d_in = h1 for t = 0, tail_dims[t - 1] after), b[t]: [tail_dims[t]]; wo: [tail_dims[last]].
 * Range: h1 <= 64, 1 <= n_tail <= 3, every tail width <= 32 (cdr_conet_fullsort_supported returns 1 / 0; outside it the call
 * returns CDR_EINVAL and the host keeps its contraction-per-layer path).  No intermediate touches memory.                       */
int cdr_conet_fullsort_supported(int h1, int n_tail, const int* tail_dims);
int cdr_conet_fullsort(void* stream, const float* P, int64_t ldp, const float* Q, int64_t ldq, int64_t U, int64_t N, int h1,
                       int n_tail, const int* tail_dims, const float* const* W, const float* const* b, const float* wo,
                       const float* bo, float* out, int64_t ldo);
/* The same scores with Q formed INSIDE the launch, Q[u] = W1u user_table[uid[u]] + b1 (W1u = the first D columns of the first layer's weight,
 * row stride ldw1): one launch per call instead of gather + contraction + scoring.  For the few-users call of recbole's evaluation loop
 * (conet.py:222-242 runs once per eval batch -- ONE user at the default eval_batch_size over a large catalogue); correct for any U, but every
 * workgroup re-forms its users' Q rows, so throughput-sized user batches belong to cdr_conet_fullsort. */
int cdr_conet_fullsort_users(void* stream, const float* P, int64_t ldp, const float* user_table, int64_t ldu, const int64_t* uid,
                             const float* W1u, int64_t ldw1, const float* b1, int D, int64_t U, int64_t N, int h1, int n_tail,
                             const int* tail_dims, const float* const* W, const float* const* b, const float* wo, const float* bo,
                             float* out, int64_t ldo);

/* ---- SSCDR helpers (sscdr.py:120-187) -------------------------------------------------------------------------- */
/* embedding_normalize: len = sum x^2, y = x / (len > 1 ? len : 1)  -- the squared-length quirk is kept */
int cdr_sqnorm_normalize_fwd(void* stream, const float* x, int64_t rows, int D, float* y, float* len_out);
int cdr_sqnorm_normalize_bwd(void* stream, const float* x, const float* len, const float* gy, int64_t rows, int D, float* gx);
/* SSCDR's whole map-phase loss in one pass (sscdr.py:161-172): mapped3 [3 n, D] = the mapping applied to [source rows of the overlapped
 * ids ; rows of the sampled interacted ids ; rows of the sampled non-interacted ids], target_rows [n, D] the ids' target rows.
 *   out3 = {MSE(mapped3[:n], target_rows) + lambda * triplet(normalize(target_rows), normalize(mapped3[n:2n]), normalize(mapped3[2n:])),
 *           the MSE, the triplet term};  g_mapped3 / g_target_rows = d out3[0] / d inputs for a unit upstream gradient
 * (cdr_scale2_unless_one applies any other: x *= s, y *= s unless the device scalar s is exactly 1).                                  */
int cdr_sscdr_map_loss(cdr_ctx* ctx, void* stream, const float* mapped3, const float* target_rows, int64_t n, int D, float margin,
                       float eps, float lambda, float* out3, float* g_mapped3, float* g_target_rows);
int cdr_scale2_unless_one(void* stream, const float* scale_dev, float* x, int64_t nx, float* y, int64_t ny);
/* nn.TripletMarginLoss(margin, p=2, eps): mean_r max(||a-p+eps|| - ||a-n+eps|| + margin, 0) */
int cdr_triplet_fwd(cdr_ctx* ctx, void* stream, const float* a, const float* p, const float* n, int64_t rows, int D,
                    float margin, float eps, float* out1, float* dap, float* dan);
int cdr_triplet_bwd(void* stream, const float* a, const float* p, const float* n, int64_t rows, int D, float margin, float eps,
                    const float* dap, const float* dan, const float* grad_out, float* ga, float* gp, float* gn);

/* ---- BiTGCF (bitgcf.py:130-250) --------------------------------------------------------------------------------
 * CSR adjacency (int64 indptr [n+1], int64 indices < 2^31, fp32 values) of the normalised bipartite graph (symmetric).
 *   cdr_spmm_csr_f32    out = A x E                                  (torch.sparse.mm, bitgcf.py:131)
 *   cdr_graph_layer_fwd side = A x E ; new = E + (side + E (.) side)   (bitgcf.py:130-135, dropout = identity)
 *   cdr_graph_layer_bwd gE = gnew (.) (1 + side) + A x (gnew (.) (1 + E))     (tmp: [n, D] scratch)
 *   cdr_transfer_*      rows < n_overlap: ((lam*s + (1-lam)*t) + (ds*s + dt*t)/(ds+dt+1e-7)) / 2, others pass through
 *   cdr_l2_normalize_*  F.normalize(p=2, dim=1, eps=1e-12), output written with leading dimension ldo
 *   cdr_copy_cols / cdr_colblock_mean_*   concat / mean of the layer outputs (bitgcf.py:191-198)
 *   cdr_embloss_*       recbole EmbLoss of the EGO rows: out3 = {(||U_b|| + ||I_b||)/B, ||U_b||, ||I_b||}            */
int cdr_spmm_csr_f32(void* stream, const int64_t* indptr, const int64_t* indices, const float* values, int64_t n_rows,
                     const float* E, int D, float* out);
/* row_flags (uint8 [n_rows], NULL = every row): the rows of the LAST propagation layer that the loss gathers -- nothing else reads
 * that layer's output (bitgcf.py:200-205,222-240 index it with the batch's user / item ids only).  Forward: only flagged rows of
 * side / new are computed (the others stay unwritten).  Backward: the caller guarantees gnew == 0 outside the flagged rows; the
 * non-zeros whose column is not flagged are skipped (exact zeros of the sum: gE is bit-identical to the unflagged call), tmp is not
 * used (may be NULL).  cdr_row_flags builds the set from id lists (rows offsets[i] + ids[i][k]) into one work buffer that starts with
 * the byte flags and also holds the set as a bit map (the backward's column probes, copied into LDS) and as a compacted row list
 * (the forward walks it); row_flags arguments take that buffer.                                                                  */
int cdr_graph_layer_fwd(void* stream, const int64_t* indptr, const int64_t* indices, const float* values, int64_t n_rows,
                        const float* E, int D, float* side_out, float* new_out, const uint8_t* row_flags);
int cdr_graph_layer_bwd(void* stream, const int64_t* indptr, const int64_t* indices, const float* values, int64_t n_rows,
                        const float* E, const float* side, const float* gnew, int D, float* tmp, float* gE, const uint8_t* row_flags);
int cdr_row_flags_layout(int64_t rows, size_t* bytes);      /* size of the work buffer below (16-B aligned device memory) */
int cdr_row_flags(void* stream, int n_lists, const int64_t* const* ids, const int64_t* counts, const int64_t* offsets, int64_t rows,
                  uint8_t* work, size_t work_bytes);
/* Row-sharded BiTGCF (BASELINE configs[3]; SURVEY 8e: E and the CSR sharded by destination row, per-layer all-gather of E): the rank
 * holds n_rows rows of the CSR whose column indices address the ALL-GATHERED embedding buffer; the row's own value comes from the
 * local slice.  Forward: side = A_rows E_gathered, new = E_rows + side + E_rows (.) side (bitgcf.py:130-135).  Backward (the
 * adjacency is symmetric): tmp_rows = gnew_rows (.) (1 + E_rows) [cdr_mul_one_plus] is all-gathered by the host, then
 * gE_rows = gnew_rows (.) (1 + side_rows) + A_rows tmp_gathered.                                                                 */
int cdr_graph_layer_fwd_rows(void* stream, const int64_t* indptr, const int64_t* indices, const float* values, int64_t n_rows,
                             const float* E_gathered, const float* E_rows, int D, float* side_out, float* new_out);
int cdr_mul_one_plus(void* stream, const float* g, const float* x, int64_t n, float* out);
int cdr_graph_layer_bwd_rows(void* stream, const int64_t* indptr, const int64_t* indices, const float* values, int64_t n_rows,
                             const float* tmp_gathered, const float* gnew_rows, const float* side_rows, int D, float* gE_rows);
int cdr_transfer_fwd(void* stream, const float* S, const float* T, const float* deg_s, const float* deg_t, int64_t rows, int D,
                     int64_t n_overlap, float lam_s, float lam_t, float* S_out, float* T_out);
int cdr_transfer_bwd(void* stream, const float* gS_out, const float* gT_out, const float* deg_s, const float* deg_t,
                     int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t, float* gS, float* gT);
/* the same with nn.Dropout of the layer output (bitgcf.py:134) folded in: the forward reads S / T through the mask, the backward writes
 * gS / gT through it.  Mask = cdr_dropout's (host `seed`) or cdr_dropout_dev's (`seed_dev` != NULL) with salt_s / salt_t for the two
 * domains; elem0 = element offset of this row block inside the [n, D] layer output (so that the user and the item block, launched
 * separately, draw the masks one call over the whole output would).                                                              */
int cdr_transfer_drop_fwd(void* stream, const float* S, const float* T, const float* deg_s, const float* deg_t, int64_t rows, int D,
                          int64_t n_overlap, float lam_s, float lam_t, float p, uint64_t seed, const int64_t* seed_dev,
                          uint64_t salt_s, uint64_t salt_t, int64_t elem0, float* S_out, float* T_out);
int cdr_transfer_drop_bwd(void* stream, const float* gS_out, const float* gT_out, const float* deg_s, const float* deg_t,
                          int64_t rows, int D, int64_t n_overlap, float lam_s, float lam_t, float p, uint64_t seed,
                          const int64_t* seed_dev, uint64_t salt_s, uint64_t salt_t, int64_t elem0, float* gS, float* gT);
/* One launch per layer and direction for everything between the graph layer and the layer stack (bitgcf.py:134,137-172,190-199), users
 * and items, both domains: [dropout of the layer output, p = 0: none] -> transfer on the overlapped rows -> L2-normalised copy into
 * the stack's column block (leading dimension ldc / ldg); one wave per row of the stacked [users ; items] table.  Same arithmetic in
 * the same order as cdr_dropout(_dev) / cdr_transfer_* / cdr_l2_normalize_* run one after the other (equal to the last bit or two: FMA
 * contraction; identical dropout masks).  The backward adds the gradient
 * arriving from the layer above (gS_prev / gT_prev, both NULL for the top layer) before the transfer's backward.
 * row_flags (NULL = all rows; see cdr_graph_layer_fwd): forward writes zeros into the stack block of unflagged rows and nothing else
 * for them; backward (top layer only: gS_prev == NULL) writes zero gradient rows for them without reading their saved tensors.   */
int cdr_bitgcf_mix_fwd(void* stream, const float* newS, const float* newT, const float* deg_su, const float* deg_tu,
                       const float* deg_si, const float* deg_ti, int64_t nu, int64_t ni, int D, int64_t OU, int64_t OI, float lam_s,
                       float lam_t, float p, uint64_t seed, const int64_t* seed_dev, uint64_t salt_s, uint64_t salt_t, float* S2,
                       float* T2, float* catS_block, float* catT_block, int64_t ldc, float* nS, float* nT, const uint8_t* row_flags);
int cdr_bitgcf_mix_bwd(void* stream, const float* S2, const float* T2, const float* nS, const float* nT, const float* gcatS_block,
                       const float* gcatT_block, int64_t ldg, const float* gS_prev, const float* gT_prev, const float* deg_su,
                       const float* deg_tu, const float* deg_si, const float* deg_ti, int64_t nu, int64_t ni, int D, int64_t OU,
                       int64_t OI, float lam_s, float lam_t, float p, uint64_t seed, const int64_t* seed_dev, uint64_t salt_s,
                       uint64_t salt_t, float* gnS, float* gnT, const uint8_t* row_flags);
int cdr_l2_normalize_fwd(void* stream, const float* x, int64_t rows, int D, float* y, int64_t ldo, float* norm_out);
int cdr_l2_normalize_bwd(void* stream, const float* x, const float* norm, const float* gy, int64_t ldg, int64_t rows, int D,
                         float* gx, int accumulate);
int cdr_copy_cols(void* stream, const float* src, int64_t lds, int64_t rows, int D, float* dst, int64_t ldo, int accumulate);
/* BiTGCF's ego layer in one launch (bitgcf.py:175-178,190): S = [su ; si], T = [tu ; ti] (contiguous [nu + ni, D]) and the same rows into
 * column block 0 of the two layer stacks (leading dimension ldc); and its backward's last step: gS += block 0 of gcatS, likewise gT. */
int cdr_bitgcf_stack(void* stream, const float* su, const float* si, const float* tu, const float* ti, int64_t nu, int64_t ni, int D,
                     float* S, float* T, float* catS, float* catT, int64_t ldc);
int cdr_bitgcf_unstack_bwd(void* stream, const float* gcatS, const float* gcatT, int64_t ldg, int64_t n, int D, float* gS, float* gT);
int cdr_colblock_mean_fwd(void* stream, const float* cat, int64_t rows, int D, int nb, float* out);
int cdr_colblock_mean_bwd(void* stream, const float* gout, int64_t rows, int D, int nb, float* gcat);
/* nn.Dropout(p), training mode: out = mask ? x/(1-p) : 0, counter-based mask from `seed` (same call with the same seed on
 * the upstream gradient is the backward); in place allowed (bitgcf.py:66,134). */
int cdr_dropout(void* stream, const float* x, int64_t n, float p, uint64_t seed, float* out);
/* hipGraph-capturable form: the seed is a device counter (bump it with cdr_inc_i64 once per step, inside the captured step), `salt`
 * separates the masks drawn within one step (layer, domain).  A host seed would be baked into the graph and repeat one mask forever. */
int cdr_dropout_dev(void* stream, const float* x, int64_t n, float p, const int64_t* seed_dev, uint64_t salt, float* out);
int cdr_embloss_fwd(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_tab, int D,
                    const int64_t* uid, const int64_t* iid, int64_t B, float* out3);
int cdr_embloss_bwd_dense(void* stream, const float* user_tab, const float* item_tab, int D, const int64_t* uid,
                          const int64_t* iid, int64_t B, const float* out3, const float* grad_out,
                          float* grad_user_tab, float* grad_item_tab);
/* Two EmbLoss backward passes in one launch (host arrays of two, as cdr_point_fwd_pair): norms[d] -> {||U_b||_F, ||I_b||_F} on the device
 * (out3 + 1 of cdr_embloss_fwd, out4 + 2 of cdr_point_fwd_pair_ex), row coefficient scale[d] * grad_out[d][0] / (B ||rows||)
 * (grad_out NULL or grad_out[d] NULL = 1); gradients are ADDED (fp32 atomics) into the given tables. */
int cdr_embloss_bwd_dense_pair(void* stream, const float* const* user_tab, const float* const* item_tab, int D, const int64_t* const* uid,
                               const int64_t* const* iid, const int64_t* B, const float* const* norms, const float* const* grad_out,
                               const float* scale, float* const* grad_user_tab, float* const* grad_item_tab);

/* ---- integer paths (bit-exact) --------------------------------------------------------------------------------
 * cdr_overlap_remap (HOST function, no GPU): CrossDomainDataset.calculate_user_item_from_both_domain + _remap_fields for
 * one field (dataset.py:344-445,109-123).  Tokens are UTF-8 bytes: token i = bytes[off[i], off[i+1]); isnan[i] != 0 marks
 * a NaN token (dropped; its output id is -1).  Output: the remapped id of every occurrence and
 * counts4 = {num_overlap (PAD included), num_source_only, num_target_only, num_total}.
 * cdr_revoke_map (device): iid < overlap_item_num ? iid : iid - target_only_item_num  (dataloader.py:244-245). */
int cdr_overlap_remap(const char* src_bytes, const int64_t* src_off, const uint8_t* src_isnan, int64_t n_src,
                      const char* tgt_bytes, const int64_t* tgt_off, const uint8_t* tgt_isnan, int64_t n_tgt,
                      int64_t* src_ids, int64_t* tgt_ids, int64_t* counts4);
/* The same remap ON THE DEVICE, for fields of 10^7..10^9 token occurrences (dataset.py:344-445 + :109-123 as above, bit-exact
 * with cdr_overlap_remap): all occurrences of both domains are sorted together in byte order by K + 1 stable radix sorts (K = ceil(longest
 * token / 8)), a run of equal tokens is one distinct token, its class the OR of its occurrences' domains, its id the class base + the number
 * of runs of the same class before it.  Every pointer is DEVICE memory (bytes uint8, offsets int64 [n + 1], isnan uint8 [n] or NULL);
 * ids come back as int64 per occurrence (-1 for NaN), counts4 = {OV (PAD counted), source-only, target-only, total} on the device.
 * workspace: cdr_overlap_remap_dev_workspace_bytes(n_src, n_tgt), 256-byte aligned.  n_src + n_tgt < 2^32 - 1.  One host wait inside
 * (token count and longest token decide the number of passes; *passes_out, optional, reports it): not capturable, ingest runs once. */
int cdr_overlap_remap_dev_workspace_bytes(int64_t n_src, int64_t n_tgt, size_t* bytes);
int cdr_overlap_remap_dev(void* stream, const uint8_t* src_bytes, const int64_t* src_off, const uint8_t* src_isnan, int64_t n_src,
                          const uint8_t* tgt_bytes, const int64_t* tgt_off, const uint8_t* tgt_isnan, int64_t n_tgt, int64_t* src_ids,
                          int64_t* tgt_ids, int64_t* counts4, void* workspace, size_t workspace_bytes, int64_t* passes_out);
int cdr_revoke_map(void* stream, const int64_t* ids, int64_t n, int64_t overlap_item_num,
                   int64_t target_only_item_num, int64_t* out);

/* ---- fused row-wise training step (tables too large for dense gradients / dense Adam: BASELINE config C5) -----
 * Replaces, for one BPR step, loss.backward() + optimizer.step() of the reference's loop (trainer.py:59-73 ->
 * recbole Trainer._train_epoch) without ever materialising a table-sized gradient:
 *   cdr_bpr_fwd_grad : cdr_bpr_fwd + the compact gradient rows  GU[b] = g_b (I[pid_b]-I[nid_b]),  GP[b] = g_b U[uid_b]
 *                      out9 = {total, bpr, ||U_b||, ||I_b||, c_u, c_i, sum_loss, sum_u2, sum_p2},
 *                      c = reg_weight / (B_mean * norm).  B_mean (<=0 -> B) is the batch the mean / EmbLoss are
 *                      taken over: the GLOBAL batch when the step is row-sharded over several GPUs -- the three raw
 *                      sums are then all-reduced and cdr_loss_finish_sums recomputes out[0..5] from them.
 *   cdr_sort_ids     : stable radix sort of (row id, occurrence index) over the significant key bits; ids1 (optional)
 *                      is appended after ids0 (items: ids0 = pid, ids1 = nid -> occurrences [0,B) positive, [B,2B) negative)
 *   cdr_rowwise_apply: per distinct row r (segment of keys_sorted):
 *                        grad = sum_{o in seg, o <  neg_start} G[o] - sum_{o in seg, o >= neg_start} G[o - neg_start]
 *                             + reg_coef[0] * #{o in seg : o < reg_limit} * table[r]
 *                      then opt 0: SGD  table[r] -= lr * (grad + wd * table[r])
 *                           opt 1: Adam (torch.optim.Adam arithmetic, per-row lazy) with exp_avg / exp_avg_sq rows.
 *                      Summation follows occurrence order (the sort is stable): bit-reproducible.  Segments longer than
 *                      32 occurrences (skewed id streams) are cut into pieces of 256 that are summed in parallel and
 *                      combined in piece order -- still a fixed order; the partial sums live in ctx-owned scratch, so a
 *                      ctx must not be used from two streams at once (the host binding keeps one per stream).
 */
int cdr_bpr_fwd_grad(cdr_ctx* ctx, void* stream,
                     const float* user_tab, const float* item_tab, int D,
                     const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, int64_t B_mean,
                     float gamma, float reg_weight, float* out9, float* GU /* [B,D] */, float* GP /* [B,D] */,
                     int scatter /* != 0 (row-sharded step): GP[pid[b]] = g u, GP[nid[b]] = -g u instead of GP[b] = g u */);
int cdr_loss_finish_sums(void* stream, const float* sums3, int64_t B_mean, float reg_weight, float* out6);
/* The same step with the optimizer of every row that occurs ONCE in the batch applied by the forward kernel itself (round 3;
 * replaces cdr_bpr_fwd_grad + cdr_sort_ids_two_tables + 2 x cdr_rowwise_apply for emcdr.py:123-131,146-154 + Adam): the EmbLoss
 * norms of the batch are gathered first (the gradient coefficient reg_weight / (B ||rows||) is needed before the first row is
 * updated), the ids are sorted, one pass over the sorted keys marks the single occurrences and compacts the heads of the
 * duplicate segments; the forward kernel reads the two moments of a single row next to the row and writes row + moments back
 * (its GU[b] / GP[b] is never written); duplicate rows go through the segmented apply as before.  Per row the arithmetic is
 * that of cdr_rowwise_apply.  keys / perm: uint32 [3B]; flags: uint8 [4B], 4-byte aligned ({user, positive, negative, -} per
 * triple); heads: uint32 [cdr_bpr_step_fused_heads_words(B)]; sort_ws as for cdr_sort_ids_two_tables(3B).  step_user /
 * step_item: the tables' update counts INCLUDING this update (Adam bias correction).  out9 as cdr_bpr_fwd_grad's.             */
int cdr_bpr_step_fused_heads_words(int64_t B, int64_t* words);
int cdr_bpr_step_fused(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                       float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D, const int64_t* uid,
                       const int64_t* pid, const int64_t* nid, int64_t B, float gamma, float reg_weight, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int64_t step_user, int64_t step_item, float* out9,
                       float* GU /* [B,D] */, float* GP /* [B,D] */, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                       void* sort_ws, size_t sort_ws_bytes);
/* The same step (Adam) with the tables' update counts in DEVICE memory: *step_user_dev / *step_item_dev hold the counts BEFORE the call
 * and are advanced by it; hp_dev: 4 floats of caller-owned device scratch for the Adam scalars derived from them.  Nothing about the update
 * number is baked into the launches: the call can be captured in a hipGraph and replayed (emcdr.py:110-154 under recbole's step loop,
 * one graph launch per batch). */
int cdr_bpr_step_fused_dev(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                           float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D, const int64_t* uid,
                           const int64_t* pid, const int64_t* nid, int64_t B, float gamma, float reg_weight, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int64_t* step_user_dev, int64_t* step_item_dev, float* hp_dev,
                           float* out9, float* GU, float* GP, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                           void* sort_ws, size_t sort_ws_bytes);

/* The pointwise step -- (user, item, label) rows of EMCDR's default MF model (emcdr.py:111-122: MSE on the raw dot) and CMF (cmf.py:75-99:
 * BCE on sigmoid(dot)) with reg_weight * EmbLoss(U[uid], I[iid]) -- as ONE call on the forward-and-update pass: rows that occur once in the
 * batch are updated by the kernel that gathers them, duplicate rows through GU / GI and the segmented applies.  Replaces
 * cdr_point_fwd_grad -> cdr_sort_ids_two_tables -> cdr_rowwise_apply x 2.  loss_kind: CDR_LOSS_MSE | CDR_LOSS_BCE.  Buffers as
 * cdr_bpr_step_fused's for B rows: keys / perm [2 B], flags [4 B] (4-byte aligned), heads cdr_bpr_step_fused_heads_words(B) words,
 * GU / GI [B, D], sort workspace cdr_sort_workspace_bytes(2 B, 2 * next power of two of the larger row count). */
int cdr_point_step_fused(cdr_ctx* ctx, void* stream, int loss_kind, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                         float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D, const int64_t* uid, const int64_t* iid,
                         const float* label, int64_t B, float reg_weight, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int64_t step_user, int64_t step_item, float* out9, float* GU, float* GI, uint32_t* keys, uint32_t* perm,
                         uint8_t* flags, uint32_t* heads, void* sort_ws, size_t sort_ws_bytes);

/* The same step per POSITIVE, for recbole's pointwise batch layout (TrainDataLoader._neg_sampling under data/dataloader.py:114-162): uid [S]
 * (the first S entries of the user column tiled 1 + k times), iid [S + S k] = [positives | k-major negatives], label [S + S k].  In that
 * layout every user row occurs 1 + k times, so the per-row call above finds no user that occurs once; here the user row of a positive is
 * gathered once, its gradient over the 1 + k rows summed in registers, and a user that occurs in one positive is updated in place.  Sizes
 * of flags / heads from cdr_bpr_step_fused_kmajor_sizes(S, k); GU [S, D], GI [S + S k, D]; keys / perm [2 S + S k]; out12: 12 floats
 * (out12[9] = the user coefficient per list occurrence, scratch). */
int cdr_point_step_fused_kmajor(cdr_ctx* ctx, void* stream, int loss_kind, int opt, float* user_tab, float* user_m, float* user_v,
                                int64_t user_rows, float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D,
                                const int64_t* uid, const int64_t* iid, const float* label, int64_t S, int k, float reg_weight, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int64_t step_user, int64_t step_item, float* out12,
                                float* GU, float* GI, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads, void* sort_ws,
                                size_t sort_ws_bytes);

/* ---- the fused single-occurrence update inside the two multi-GPU layouts (reference math emcdr.py:110-154 on sharded tables:
 * the reference itself is single-device, parity = the one-GPU result).
 * DIMENSION shard -- the step cut in two around the all-reduce of the partial scores (cdr_bpr_partial_diff):
 *   cdr_bpr_step_presort    ids only (runs under the all-reduce): two-table sort + occurrence flags + duplicate-segment heads;
 *                           *key_base_out = the table bit of the item keys (host value; hand it to cdr_bpr_step_from_diff)
 *   cdr_bpr_step_from_diff  diff [B + 2] = all-reduced {x_t ..., sum u^2, sum p^2}; every row occurring once in the GLOBAL batch is updated
 *                           by the pass that re-gathers this rank's column slices, duplicate rows through GU / GP + the segmented apply
 * buffers (keys, perm [3 B]; flags [4 B] 4-byte aligned; heads cdr_bpr_step_fused_heads_words(B) words) as cdr_bpr_step_fused's. */
int cdr_bpr_step_presort(cdr_ctx* ctx, void* stream, const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B,
                         int64_t user_rows, int64_t item_rows, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                         void* sort_ws, size_t sort_ws_bytes, uint32_t* key_base_out);
int cdr_bpr_step_from_diff(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, float* item_tab,
                           float* item_m, float* item_v, int Ds, const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B,
                           float gamma, float reg_weight, float lr, float beta1, float beta2, float eps, float weight_decay,
                           int64_t step_user, int64_t step_item, const float* diff, uint32_t key_base, float* out9, float* GU, float* GP,
                           const uint32_t* keys, const uint32_t* perm, const uint8_t* flags, uint32_t* heads);
/* The pointwise rows (EMCDR-MF / CMF) of the dimension layout, cut the same way around the all-reduce of cdr_point_partial_dot's output:
 * cdr_point_step_presort (ids only) and cdr_point_step_from_dot (dot [B + 2] = all-reduced {<u, i> ..., sum u^2, sum i^2}); buffers as
 * cdr_point_step_fused's. */
int cdr_point_step_presort(cdr_ctx* ctx, void* stream, const int64_t* uid, const int64_t* iid, int64_t B, int64_t user_rows, int64_t item_rows,
                           uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads, void* sort_ws, size_t sort_ws_bytes,
                           uint32_t* key_base_out);
int cdr_point_step_from_dot(cdr_ctx* ctx, void* stream, int loss_kind, int opt, float* user_tab, float* user_m, float* user_v, float* item_tab,
                            float* item_m, float* item_v, int Ds, const int64_t* uid, const int64_t* iid, const float* label, int64_t B,
                            float reg_weight, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step_user,
                            int64_t step_item, const float* dot, uint32_t key_base, float* out9, float* GU, float* GI, const uint32_t* keys,
                            const uint32_t* perm, const uint8_t* flags, uint32_t* heads);
/* ROW shard -- a rank's triples after user-aligned routing: local user rows u_loc, the item rows it asked for in `irows` (one row per
 * distinct item, indexed by ip / in).
 *   cdr_batch_norm_sums       sums3 = {0, sum ||U[u]||^2, sum ||irows[ip]||^2}: all-reduce over the ranks, then cdr_loss_finish_sums puts the
 *                             EmbLoss coefficients of the GLOBAL batch into out9[4..5] BEFORE any row moves
 *   cdr_bpr_shard_local_step  sort + flags of the local user rows, one forward-and-update pass (user rows occurring once updated in place;
 *                             GU for duplicate user rows; GP[t] = g_t u_t for every triple, never an item update), segmented apply of
 *                             the duplicate user rows.  out9[4..5] in; out9[6..8] = this rank's {loss sum, sum u^2, sum p^2} out.
 *                             B_global: the divisor of the mean (triples of all ranks).  flags [4 Bl] must be zero-initialised once. */
int cdr_batch_norm_sums(cdr_ctx* ctx, void* stream, const float* user_tab, const float* item_rows, int D, const int64_t* uid,
                        const int64_t* pid, int64_t B, float* sums3);
int cdr_bpr_shard_local_step(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                             const float* irows, int D, const int64_t* u_loc, const int64_t* ip, const int64_t* in, int64_t Bl,
                             int64_t B_global, float gamma, float reg_weight, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step_user, float* out9, float* GU, float* GP, uint32_t* keys, uint32_t* perm,
                             uint8_t* flags, uint32_t* heads, void* sort_ws, size_t sort_ws_bytes);
/* Round 6 -- the row shard's step on its own passes (north_star's layout; parity = the one-GPU result of emcdr.py:110-154, the reference has no
 * multi-GPU code).  Per rank and domain step:
 *   cdr_route_triples      stage 0: triples bucketed by the owner of their USER row (uid % world, stable), written interleaved {uid / world, pid,
 *                          nid} in owner order into send3 [n, 3]; counts [world] (device).  Workspace: cdr_route_triples_workspace_bytes.
 *   cdr_bpr_shard_plan     after the all-to-all of the triples (recv3 [Bl, 3]): ONE radix sort of {user rows | item keys (owner << bits) | local
 *                          row} -> u_loc [Bl]; keys / perm [3 Bl] (section A = users, B = [positives | negatives]); flags [4 Bl] (4-byte aligned;
 *                          byte 0 / 1 / 2 of triple t = its user / positive / negative is the ONLY occurrence of its row in this rank's lists);
 *                          heads (cdr_bpr_shard_plan_sizes words: counters[4] | duplicate user segments | duplicate item segments); the request
 *                          list: uniq_local [<= 2 Bl] (distinct item rows, grouped by owner, ascending local row), uidx [2 Bl] (sorted position ->
 *                          slot), umap [2 Bl] (occurrence -> slot: ip = umap, in = umap + Bl), counts [world + 1] (slots per owner), n_uniq [1].
 *   cdr_gather_rows_norms  the owner: out[r] = tab[ids[r]], nrm2[r] = ||tab[ids[r]]||^2 -- the squared norms travel beside the rows (4 B each).
 *   cdr_shard_norm_sums    sums3 = {0, sum_t ||U[u_loc[t]]||^2, sum_t nrm2[ip[t]]} -> all-reduce -> cdr_loss_finish_sums: the EmbLoss coefficients
 *                          of the GLOBAL batch in out9[4..5] before any row moves.
 *   cdr_bpr_shard_step     one forward-and-update pass: user rows occurring once updated in place (GU for the others); an item occurrence that
 *                          is the only one of its row has its finished gradient row (g u + c_i p | -g u) written straight into its send slot
 *                          GS[slot] (GS [n_uniq, D]); GP[t] = g_t u_t only where a duplicate needs it; then one launch for the duplicate user
 *                          rows (segmented apply) and the duplicate item segments (GS[slot] = signed sum in occurrence order + c_i #pos row).
 *                          out9[4..5] in; out9[6] = this rank's loss sum out (out9[7..8], the all-reduced norm sums, stay).
 *   cdr_shard_owner_apply  the owner: ids = `runs` ascending duplicate-free runs (one per requesting rank), grads one summed row per id ->
 *                          row-wise optimizer.  One run is applied as it stands (no sort); several are radix-sorted first (stable: rank order).
 *                          keys / perm [n]; ws as cdr_sort_workspace_bytes(n, table_rows) (unused for one run). */
int cdr_route_triples_workspace_bytes(int64_t n, int world, size_t* bytes);
int cdr_route_triples(cdr_ctx* ctx, void* stream, const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t n, int world,
                      int64_t* send3, int64_t* counts, void* workspace, size_t workspace_bytes);
int cdr_bpr_shard_plan_sizes(int64_t Bl, int64_t user_rows, int64_t item_local_rows, int world, int64_t* heads_words, size_t* ws_bytes);
int cdr_bpr_shard_plan(cdr_ctx* ctx, void* stream, const int64_t* recv3, int64_t Bl, int64_t user_rows, int64_t item_local_rows, int world,
                       int64_t* u_loc, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads, uint32_t* uidx, int64_t* uniq_local,
                       int64_t* umap, int64_t* counts, int64_t* n_uniq, void* ws, size_t ws_bytes);
int cdr_gather_rows_norms(void* stream, const float* tab, int D, const int64_t* ids, int64_t n, float* out, float* nrm2);
int cdr_shard_norm_sums(cdr_ctx* ctx, void* stream, const float* user_tab, int D, const int64_t* u_loc, const float* nrm2, const int64_t* ip,
                        int64_t Bl, float* sums3);
int cdr_bpr_shard_step(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, const float* irows, int D,
                       const int64_t* u_loc, const int64_t* umap, int64_t Bl, int64_t B_global, float gamma, float reg_weight, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int64_t step_user, float* out9, float* GU, float* GP, float* GS,
                       const uint32_t* keys, const uint32_t* perm, const uint8_t* flags, uint32_t* heads, const uint32_t* uidx);
int cdr_shard_owner_apply(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int64_t table_rows, int D,
                          const int64_t* ids, int64_t n, int runs, const float* grads, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int64_t step, uint32_t* keys, uint32_t* perm, void* ws, size_t ws_bytes);
/* ... and on recbole's pairwise batch layout (S positives tiled k times, k-major negatives: crossdomain_sampler.py:148-152): one
 * lane group per positive, u and p gathered once; uid / pid [S], nid [S k]; the loss, the per-row gradients and the update are
 * those of cdr_bpr_step_fused on the B = S k tiled rows.  GU [S, D]; GI [S + S k, D] (gradient rows of the duplicate item
 * occurrences, indexed like the item list [pid | nid]); keys / perm uint32 [2 S + S k]; flags / heads sized by
 * cdr_bpr_step_fused_kmajor_sizes; sort_ws as for cdr_sort_ids_two_tables(2 S + S k).                                        */
int cdr_bpr_step_fused_kmajor_sizes(int64_t S, int k, int64_t* flag_bytes, int64_t* heads_words);
int cdr_bpr_step_fused_kmajor(cdr_ctx* ctx, void* stream, int opt, float* user_tab, float* user_m, float* user_v, int64_t user_rows,
                              float* item_tab, float* item_m, float* item_v, int64_t item_rows, int D, const int64_t* uid,
                              const int64_t* pid, const int64_t* nid, int64_t S, int k, float gamma, float reg_weight, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step_user, int64_t step_item,
                              float* out9, float* GU, float* GI, uint32_t* keys, uint32_t* perm, uint8_t* flags, uint32_t* heads,
                              void* sort_ws, size_t sort_ws_bytes);
/* pointwise form of the same step (EMCDR's default MF latent factor model, emcdr.py:111-122: MSE(dot, label) +
 * reg_weight * EmbLoss(u_rows, i_rows); CDR_LOSS_BCE = BCE on sigmoid(dot) as in cmf.py:75-99): GU[b] = g_b i_b, GI[b] = g_b u_b,
 * out9 as above; apply both tables with cdr_sort_ids + cdr_rowwise_apply (neg_start = n, reg_limit = n).                 */
int cdr_point_fwd_grad(cdr_ctx* ctx, void* stream, int loss_kind, const float* user_tab, const float* item_tab, int D,
                       const int64_t* uid, const int64_t* iid, const float* label, int64_t B, float reg_weight,
                       float* out9, float* GU /* [B,D] */, float* GI /* [B,D] */);
int cdr_sort_workspace_bytes(int64_t n, int64_t num_rows, size_t* bytes);
int cdr_sort_ids(cdr_ctx* ctx, void* stream, const int64_t* ids0, int64_t n0, const int64_t* ids1, int64_t n1, int64_t num_rows,
                 uint32_t* keys_sorted /* [n0+n1] */, uint32_t* perm /* [n0+n1] */,
                 void* workspace, size_t workspace_bytes);
#define CDR_OPT_SGD 0
#define CDR_OPT_ADAM 1
int cdr_rowwise_apply(cdr_ctx* ctx, void* stream, int opt, float* table, float* exp_avg, float* exp_avg_sq, int D,
                      const uint32_t* keys_sorted, const uint32_t* perm, int64_t n,
                      const float* G, int64_t neg_start, int64_t reg_limit, const float* reg_coef,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                      const int64_t* occ_ids /* NULL, or per-occurrence ids whose bit 62 marks the EmbLoss occurrences */,
                      uint32_t key_base /* 0, or the table bit cdr_sort_ids_two_tables put on this table's keys */);
/* Both tables of a step in one sort (the radix sort's cost is a fixed ~0.16 ms at these sizes): table a's n_a keys come out
 * first (keys_sorted[0, n_a), perm = occurrence index), table b's n_b0 + n_b1 behind them with *key_base_out added to the
 * key and perm = occurrence index inside b's own list (ids_b0 ++ ids_b1).  Size the workspace with
 * cdr_sort_workspace_bytes(n_a + n_b0 + n_b1, 2^(1 + ceil(log2(max(rows_a, rows_b))))).                               */
int cdr_sort_ids_two_tables(cdr_ctx* ctx, void* stream, const int64_t* ids_a, int64_t n_a, int64_t rows_a,
                            const int64_t* ids_b0, int64_t n_b0, const int64_t* ids_b1, int64_t n_b1, int64_t rows_b,
                            uint32_t* keys_sorted, uint32_t* perm, uint32_t* key_base_out, void* workspace, size_t workspace_bytes);

/* ---- negative sampler on device (recbole_cdr/sampler/crossdomain_sampler.py:139-175,212-221) ----------
 * out[j + m*S] (k-major) = m-th negative of users[j]: uniform over [lo0,hi0) U [lo1,hi1), redrawn while it is one of the
 * user's used items (CSR over users, ascending column ids; NULL = no rejection).  *fail_flag is set if 64 redraws did
 * not find a free item for some element (the reference raises for such users at construction, :241-247).             */
int cdr_neg_sample_uniform(void* stream, const int64_t* users, int64_t S, int k, int64_t lo0, int64_t hi0,
                           int64_t lo1, int64_t hi1, const int64_t* used_indptr, const int64_t* used_indices,
                           uint64_t seed, int64_t* out, int* fail_flag);
/* SSCDR's in-loss sampler (sscdr.py:89-118) on the device: per overlapped id one interacted source-domain entity (uniform over the
 * id's interaction list, repeats included; an empty list counts as [0]) and one non-interacted candidate of [lo0, hi0) U [lo1, hi1)
 * (redrawn while it is in the list; after 64 rejections the r-th free candidate is taken directly).  hist = CSR over ids, entries
 * ascending per id.  calls_dev (optional device int64): the call number mixed into the RNG key, so that a captured launch draws
 * fresh ids on every replay when the caller bumps it (cdr_inc_i64) behind the launch.                                           */
int cdr_sscdr_pair_sample(void* stream, const int64_t* ids, int64_t n, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1,
                          const int64_t* hist_indptr, const int64_t* hist_indices, uint64_t seed, const int64_t* calls_dev,
                          int64_t* pos_out, int64_t* neg_out, int* fail_flag);
/* popularity-biased variant (crossdomain_sampler.py:66-114, distribution == 'popularity'): candidates drawn through a Walker
 * alias table over the n_keys distinct items of the sampler's interactions (keys / prob / alias as the reference builds
 * them; alias holds item ids), then the same rejection and layout. */
int cdr_neg_sample_alias(void* stream, const int64_t* users, int64_t S, int k, const int64_t* keys, const float* prob,
                         const int64_t* alias, int64_t n_keys, const int64_t* used_indptr, const int64_t* used_indices,
                         uint64_t seed, int64_t* out, int* fail_flag);

/* ---- the loader's batch on the device, in one launch -----------------------------------------------------
 * replaces recbole's TrainDataLoader._next_batch_data + _neg_sampling as the cross-domain loaders drive them
 * (recbole_cdr/data/dataloader.py:114-162: slice of the epoch's shuffled interactions -> repeat -> sample_by_user_ids ->
 * join, crossdomain_sampler.py:139-175) and OverlapDataloader's slice (dataloader.py:37-52).  Rows [start, start + S) of the
 * two int64 columns users_all / items_all [n_rows]:
 *   pairwise  (pointwise = 0, k >= 1): out_users[j + m S] = u_j, out_items[j + m S] = i_j, out_neg[j + m S] = m-th negative (m < k)
 *   pointwise (pointwise = 1, k >= 1): out_users[j + t S] = u_j (t <= k); out_items[j] = i_j, out_items[S + j + m S] = m-th negative
 *   k == 0                           : out_users[j] = users_all[start + j]  (items_all / out_items / out_neg may be NULL)
 * cursor: DEVICE int64 [4] = {start, draws, sign-in word (0 between launches), spare}.  The launch reads start and draws, and
 * its last workgroup to finish stores start + S and draws + 1, so the call can be captured in a hipGraph: every replay yields the
 * next batch with fresh negatives.  dist = 0: uniform candidates [lo0,hi0) U [lo1,hi1) (cdr_neg_sample_uniform's draw with
 * seed + draws * 0x85EBCA77C2B2AE63); dist = 1: the alias table (cdr_neg_sample_alias's draw).  Rows past n_rows yield PAD 0.   */
/* The same arguments as a struct, and several loaders in ONE launch (the BOTH state produces the target and the source batch
 * together, recbole_cdr/data/dataloader.py:156-161): job i is served by grid row i.                                          */
typedef struct cdr_batch_job {
    const int64_t* users_all; const int64_t* items_all; int64_t n_rows; int64_t* cursor; int64_t S;
    int32_t k, pointwise, dist, reserved;
    int64_t lo0, hi0, lo1, hi1;
    const int64_t* keys; const float* prob; const int64_t* alias; int64_t n_keys;
    const int64_t* used_indptr; const int64_t* used_indices; uint64_t seed;
    int64_t* out_users; int64_t* out_items; int64_t* out_neg; int* fail_flag;
} cdr_batch_job;
#define CDR_BATCH_MAX_JOBS 4
int cdr_batch_produce_jobs(void* stream, const cdr_batch_job* jobs, int n_jobs);
/* cdr_adam_multi_dev(...) and cdr_batch_produce_jobs(jobs) in ONE launch (round 6): the production of the loader's batch i + 1 in
 * workgroups behind the optimizer's of step i -- the producer overwrites the batch buffers, which the backward of step i was the last
 * to read; both are latency-bound launches of 6-8 us at the reference's batch (properties/overall.yaml:19), side by side they cost one of
 * them.  Falls back to the two calls when the update does not fit the one-launch form (no ticket, > 24 tensors, > 512 fat workgroups). */
int cdr_adam_multi_dev_produce(void* stream, int count, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, int64_t* const* step_dev, float lr, float beta1,
                               float beta2, float eps, float weight_decay, const float* loss, float* loss_sum, unsigned* ticket,
                               const cdr_batch_job* jobs, int n_jobs);
/* cdr_lazy_adam_apply(...) and cdr_batch_produce_jobs(jobs) in ONE launch: the deferred Adam's row update of step i with the loader's batch
 * i + 1 produced in workgroups behind its own (the update reads the sorted ids and the gradient rows, never the batch buffers).  Same
 * results as the two calls.                                                                                                            */
int cdr_lazy_adam_apply_produce(void* stream, int count, int D, float* const* W, float* const* M, float* const* V, int32_t* const* last,
                                const uint32_t* const* keys_sorted, const uint32_t* const* perm, const int64_t* n, const float* const* G,
                                const int64_t* ldg, float lr, float beta1, float beta2, float eps, float weight_decay, const void* hp_table,
                                int64_t hp_capacity, int64_t* counters, const cdr_batch_job* jobs, int n_jobs);
int cdr_batch_produce(void* stream, const int64_t* users_all, const int64_t* items_all, int64_t n_rows, int64_t* cursor,
                      int64_t S, int k, int pointwise, int dist, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1,
                      const int64_t* keys, const float* prob, const int64_t* alias, int64_t n_keys,
                      const int64_t* used_indptr, const int64_t* used_indices, uint64_t seed, int64_t* out_users,
                      int64_t* out_items, int64_t* out_neg, int* fail_flag);

/* ---- owner routing for row-sharded tables (row r lives on rank r % world) -- index plumbing of shard.py --------
 * cdr_route_by_owner: stable counting sort of the ids (ids1 appended after ids0) by owner = id % world.
 *     perm[q] = occurrence index at sorted position q ; counts[k] = number of ids owned by rank k (device int64 [world])
 * cdr_permute_i64:    out[q] = src[perm[q]] / divisor (| 1<<62 for tagged occurrences)  (src = src0 ++ src1 ; divisor = world
 *                     turns ids into local rows)
 * cdr_inverse_perm:   pos[perm[q]] = q                                                                              */
int cdr_route_workspace_bytes(int64_t n, int world, size_t* bytes);
int cdr_route_by_owner(cdr_ctx* ctx, void* stream, const int64_t* ids0, int64_t n0, const int64_t* ids1, int64_t n1,
                       int world, uint32_t* perm, int64_t* counts, void* workspace, size_t workspace_bytes);
int cdr_permute_i64(void* stream, const int64_t* src0, int64_t n0, const int64_t* src1, const uint32_t* perm,
                    int64_t n, int64_t divisor, int64_t flag_below /* occurrences o < flag_below get bit 62 set */, int64_t* out);
int cdr_inverse_perm(void* stream, const uint32_t* perm, int64_t n, int64_t* pos);

/* ---- id de-duplication for the row-sharded step (a rank requests each distinct row once and returns one summed
 * gradient row per distinct id).  Input: the step's item occurrences sorted by key = (owner << local_bits) | local_row with
 * cdr_sort_ids (keys_sorted, perm).
 *   cdr_dedup_sorted: uniq_index[q] = dense index of sorted position q's segment ; uniq_local[j] = local row of unique j (grouped
 *                     by owner, ascending) ; occ_to_uniq[o] = unique index of occurrence o ; counts[k] = uniques owned by rank k ;
 *                     n_uniq[0] = number of uniques (all on the device).
 *   cdr_segsum_rows : out[j,:] = sum_{o in unique j, o < neg_start} G[o] - sum_{o >= neg_start} G[o - neg_start]
 *                               + reg_coef[0] * #{o < neg_start} * rows[j,:]      (occurrence order: bit-reproducible)       */
int cdr_dedup_workspace_bytes(int64_t n, size_t* bytes);
int cdr_dedup_sorted(void* stream, const uint32_t* keys_sorted, const uint32_t* perm, int64_t n, int world, int local_bits,
                     uint32_t* uniq_index /* [n] */, int64_t* uniq_local /* [n] */, int64_t* occ_to_uniq /* [n] */,
                     int64_t* counts /* [world + 1]: one scratch slot */, int64_t* n_uniq /* [1] */, void* workspace, size_t workspace_bytes);
int cdr_segsum_rows(cdr_ctx* ctx, void* stream, const uint32_t* keys_sorted, const uint32_t* perm, const uint32_t* uniq_index,
                    int64_t n, const float* G, int64_t neg_start, int D, const float* rows /* [n_uniq, D] or NULL */,
                    const float* reg_coef /* device scalar or NULL */, float* out /* [n_uniq, D] */);

/* ---- full-sort over row-sharded tables (SURVEY 8e "Full-sort": each rank scores its item shard, all-gather) -----------
 * cdr_interleave_shards: the all-gathered scores are shard-major [world][U][Nl]; the reference's full_sort_predict
 *     layout (emcdr.py:208-233 -> score.view(-1)) is [U][N] in item-id order: out[u][c] = gathered[c % world][u][c / world].
 * cdr_gather_owned_rows: out[r,:] = ids[r] % world == rank ? shard[ids[r] / world,:] : 0 -- all-reduce(sum) of this over the
 *     ranks replicates the eval users' rows exactly.
 * cdr_topk_merge_shards: vals / local_idx are the all-gathered per-shard top-k lists [world][U][k] (local row l of shard p is
 *     item l * world + p; short lists padded with (-inf, -1)); out = the k best per user, value descending, ties to the
 *     smaller item id -- the sharded form of cdr_fullsort_topk_f32's output.  world * k <= 4096.                       */
int cdr_interleave_shards(void* stream, const float* gathered, int world, int64_t U, int64_t Nl, int64_t N, float* out);
int cdr_gather_owned_rows(void* stream, const float* shard, int D, const int64_t* ids, int64_t n, int world, int rank,
                          float* out);
/* Block-partitioned tables (BiTGCF's row shard, BASELINE configs[3]; reference math bitgcf.py:221-247: the batch loss gathers
 * user / item rows of the propagated tables): the rank owns positions [lo, lo + rows) of the all-gathered layout.
 * cdr_gather_block_rows: out[r,:] = owned(pos[r]) ? shard[pos[r] - lo,:] : 0 (pos < 0 = padding slot) -- a reduce-scatter(sum) over the
 *     ranks of this, taken over the REPLICATED batch's positions, hands every rank the rows of its slice of the batch exactly.
 * cdr_scatter_add_block_rows: grad_shard[pos[r] - lo,:] += src[r,:] for the owned positions (fp32 atomics, as cdr_scatter_add_rows). */
int cdr_gather_block_rows(void* stream, const float* shard, int D, const int64_t* pos, int64_t n, int64_t lo, int64_t rows, float* out);
int cdr_scatter_add_block_rows(void* stream, float* grad_shard, int D, const int64_t* pos, int64_t n, int64_t lo, int64_t rows,
                               const float* src);
int cdr_topk_merge_shards(void* stream, const float* vals, const int64_t* local_idx, int world, int64_t U, int k,
                          int local_to_global /* 1: entry l of shard p is item l * world + p; 0: entries are output columns already */,
                          float* out_vals, int64_t* out_idx);

/* ---- BPR step over DIMENSION-sharded tables (the alternative to the row exchange above) ---------------------
 * Rank r holds columns [r*Ds, (r+1)*Ds) of every row (Ds = D / world) and walks the GLOBAL batch (ids all-gathered, 24 B per
 * triple); the only data-path collective is ONE all-reduce(sum) of diff[0 .. B+2):
 *   cdr_bpr_partial_diff   : diff[t] = <u,p> - <u,n> over this rank's columns (emcdr.py:98-108 split over columns);
 *                            diff[B] = sum_t |u_t|^2, diff[B+1] = sum_t |p_t|^2 over them (EmbLoss norms)
 *   cdr_bpr_grad_from_diff : from the all-reduced diff: s = sigmoid(diff[t]), g = -(1/B) s (1-s) / (gamma + s);
 *                            GU[t,:] = g (p - n), GP[t,:] = g u on this rank's columns; out9 laid out as cdr_bpr_fwd_grad's
 *                            (the same on every rank).  cdr_sort_ids_two_tables + cdr_rowwise_apply then run on the
 *                            [rows, Ds] tables exactly as in the single-GPU step.                                      */
/* The ids cross the links as int32 (12 B per triple; every table here has < 2^31 rows): cdr_ids_pack32 narrows the rank's three
 * id arrays into out32[3][Bl] (bad_flag[0] = 1 if an id does not fit); after ONE all-gather the buffer is rank-major
 * [world][3][Bl] and cdr_ids_unpack32 widens it to the field-major int64 [3][world * Bl] the kernels read.               */
int cdr_ids_pack32(void* stream, const int64_t* uid, const int64_t* pid, const int64_t* nid /* or NULL */,
                   const float* label /* pointwise rows: fp32 bit pattern in the third slot; NULL iff nid is given */, int64_t Bl,
                   int32_t* out32, int* bad_flag);
int cdr_ids_unpack32(void* stream, const int32_t* gathered, int world, int64_t Bl, int64_t* out64 /* [3][world*Bl]; [2][..] used if label_out */,
                     float* label_out /* [world*Bl] or NULL */);
int cdr_bpr_partial_diff(cdr_ctx* ctx, void* stream, const float* user_cols, const float* item_cols, int Ds,
                         const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, float* diff /* [B + 2] */);
int cdr_bpr_grad_from_diff(cdr_ctx* ctx, void* stream, const float* user_cols, const float* item_cols, int Ds,
                           const int64_t* uid, const int64_t* pid, const int64_t* nid, int64_t B, float gamma, float reg_weight,
                           const float* diff /* [B + 2], all-reduced */, float* out9, float* GU, float* GP);
/* Pointwise rows (user, item, label) in the same layout -- EMCDR's MF latent factor model (emcdr.py:111-122, MSE on the dot) and
 * BCE on sigmoid(dot) (cmf.py:75-99): dot[t] = <u,i> over this rank's columns, dot[B], dot[B+1] = its share of the EmbLoss norms
 * -> all-reduce -> GU[t,:] = g i, GI[t,:] = g u with cdr_point_fwd_grad's loss arithmetic and out9 layout.                        */
int cdr_point_partial_dot(cdr_ctx* ctx, void* stream, const float* user_cols, const float* item_cols, int Ds, const int64_t* uid,
                          const int64_t* iid, int64_t B, float* dot /* [B + 2] */);
int cdr_point_grad_from_dot(cdr_ctx* ctx, void* stream, int loss_kind, const float* user_cols, const float* item_cols, int Ds,
                            const int64_t* uid, const int64_t* iid, const float* label, int64_t B, float reg_weight,
                            const float* dot /* [B + 2], all-reduced */, float* out9, float* GU, float* GI);

/* ---- row-wise fusions of the five remaining models (their dense layers run on cdr_gemm_f32_ex) ------------------
 * DTCDR (dtcdr.py:112-126): out[r, 0..D) = maximum(A[ids[r]], B[ids[r]]) written with row stride ldo (the [user ; item] operand of
 * the NeuMF tower is two such calls into one [n, 2D] buffer); the backward routes g to the larger operand, half to each on a tie
 * (torch.maximum), scatter-added into dense [rows, D] gradients (either may be NULL).                                          */
int cdr_gather_max2(void* stream, const float* A, const float* B, int D, const int64_t* ids, int64_t n, float* out, int64_t ldo);
int cdr_gather_max2_bwd(void* stream, const float* A, const float* B, int D, const int64_t* ids, int64_t n, const float* g,
                        int64_t ldg, float* gA /* or NULL */, float* gB /* or NULL */);
/* DeepAPF (deepapf.py:69-152).  s / o / t = gathered share / domain-only / other-side rows [B, D].
 *   cdr_apf_prod    : X[b] = s[b] (.) t[b], X[B + b] = o[b] (.) t[b]     -- the attention MLP's [2B, D] operand (:77-78)
 *   cdr_apf_combine : a[2B] = the MLP's scores; the share score is replaced by -1e31 where ids[b] > n_overlap (STRICTLY greater,
 *                     :75,80), alpha = softmax over the pair, e = alpha_s s + alpha_o o, p = sigmoid(wp . (e (.) t))  (:82-88)
 *   the backwards return per-row gradients; gwp_rows [B, D] is reduced over the batch with cdr_colsum (fixed order).            */
int cdr_apf_prod(void* stream, const float* s, const float* o, const float* t, int64_t B, int D, float* X);
int cdr_apf_prod_bwd(void* stream, const float* s, const float* o, const float* t, const float* gX, int64_t B, int D,
                     float* gs, float* go, float* gt);
int cdr_apf_combine(void* stream, const float* a, const float* s, const float* o, const float* t, const float* wp,
                    const int64_t* ids, int64_t n_overlap, int64_t B, int D, float* p, float* alpha_s);
int cdr_apf_combine_bwd(void* stream, const float* s, const float* o, const float* t, const float* wp, const float* p,
                        const float* alpha_s, const float* gp, int64_t B, int D, float* ga /* [2B] */, float* gs, float* go,
                        float* gt, float* gwp_rows);
/* DCDCSR (dcdcsr.py:167-172): y = (x - mean) / (max - mean) per row, mean = (max + min) / 2; stats [n, 2] = (mean, max) or NULL.
 * The backward spreads the amax / amin gradients evenly over tied elements, as torch does.                                      */
int cdr_maxmin_norm(void* stream, const float* x, int64_t n, int D, float* y, float* stats /* or NULL */);
int cdr_maxmin_norm_bwd(void* stream, const float* x, const float* gy, int64_t n, int D, float* gx);
/* NATR phase 2 (natr.py:112-156) after the transfer layer: per batch row
 *   score[l] = bu + wu . relu(pu (.) He[l]) + (mask[l] ? 0 : -10000) ; att = softmax_l(score) ; su = sum_l att[l] He[l]
 *   b_s = bd + wd . relu(su (.) qi), b_p = bd + wd . relu(pu (.) qi) ; beta = e^b_s / (e^b_s + e^b_p)
 *   p = sigmoid((beta su + (1 - beta) pu) . qi)
 * He [B, L, D] = transfer_layer(source rows of the history), pu [B, D] the side the history belongs to, qi [B, D] the other side.
 * Saved for the backward: att [B, L], su [B, D], beta [B], p [B].  The backward returns gHe, gpu, gqi and per-row partials of the
 * four parameter gradients (gwu_rows, gwd_rows [B, D]; gb_rows [B, 2] = d bu, d bd) for cdr_colsum.                               */
#define CDR_ROWMODEL_MAX_DIM 256
#define CDR_NATR_MAX_HIST 256
int cdr_natr_att_fwd(void* stream, const float* He, const float* pu, const float* qi, const float* mask, const float* wu,
                     const float* bu, const float* wd, const float* bd, int64_t B, int L, int D, float* att, float* su,
                     float* beta, float* p);
int cdr_natr_att_bwd(void* stream, const float* He, const float* pu, const float* qi, const float* mask, const float* wu,
                     const float* bu, const float* wd, const float* bd, int64_t B, int L, int D, const float* att,
                     const float* su, const float* beta, const float* p, const float* gp, float* gHe, float* gpu, float* gqi,
                     float* gwu_rows, float* gwd_rows, float* gb_rows);

/* ---- (10) the exchanges of the multi-GPU path (SURVEY 8b family (10), 8e) ---------------------------------------------------
 * For a host that is not Python: what shard.ShardedBPRStep / ShardedFullSort do through torch.distributed, over one RCCL communicator
 * per process (one process per GPU; RCCL is bound at run time, so a torch process shares the RCCL torch has loaded).  Reference math
 * these exchanges serve: emcdr.py:110-154 on row-sharded tables (ids travel to the owner of their row, rows come back, gradient rows
 * go home), the OVERLAP transfer step emcdr.py:156-168, the sharded full-sort emcdr.py:208-233.  Counts are HOST arrays [world]; the
 * send buffer is ordered by destination rank, the receive buffer by source rank.  Everything is enqueued on `stream`.
 *   cdr_comm_unique_id    rank 0 makes the 128-byte id and hands it to the other ranks by the host's own means
 *   cdr_comm_init         collective over the `world` processes, on the calling thread's current HIP device
 *   cdr_a2a_ids           int64 ids, send_counts[p] of them to rank p     cdr_a2a_rows   fp32 rows of width D, send_rows[p] to rank p
 *   cdr_allgather_scores  n floats of every rank -> [world * n] in rank order (the per-shard scores / top-k candidates of full-sort)
 *   cdr_allreduce_sum_f32 in place (loss sums, the dimension shard's partial scores)
 * Errors: CDR_ENODEV when librccl cannot be loaded, 1000 + ncclResult_t when RCCL reports one (cdr_last_error has its text). */
typedef struct cdr_comm cdr_comm;
#define CDR_COMM_ID_BYTES 128
int cdr_comm_unique_id(void* id_out /* CDR_COMM_ID_BYTES */);
int cdr_comm_init(cdr_comm** comm, int rank, int world, const void* unique_id);
int cdr_comm_destroy(cdr_comm* comm);
int cdr_comm_info(const cdr_comm* comm, int* rank, int* world);
int cdr_a2a_ids(cdr_comm* comm, void* stream, const int64_t* send, const int64_t* send_counts, int64_t* recv,
                const int64_t* recv_counts);
int cdr_a2a_rows(cdr_comm* comm, void* stream, const float* send, const int64_t* send_rows, float* recv, const int64_t* recv_rows,
                 int D);
/* The offset / element-count arithmetic of one all-to-all(v), host only (needs neither RCCL nor a device): what cdr_a2a_ids / _rows
 * hand to ncclSend / ncclRecv for each peer.  counts[p] rows of `unit` elements of `elem` bytes; any output pointer may be NULL.
 * CDR_EINVAL on a negative count.  (Serves the same reference lines as the exchanges above; lets a host size its buffers.) */
int cdr_a2a_plan(int world, const int64_t* send_counts, const int64_t* recv_counts, int64_t unit, int64_t elem,
                 int64_t* send_off_bytes, int64_t* recv_off_bytes, int64_t* send_elems, int64_t* recv_elems,
                 int64_t* send_total_rows, int64_t* recv_total_rows);
int cdr_allgather_scores(cdr_comm* comm, void* stream, const float* send, int64_t n, float* recv);
int cdr_allreduce_sum_f32(cdr_comm* comm, void* stream, float* buf, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* CDR_HIP_H */
