"""ctypes binding of libcdrhip.so (include/cdr_hip.h) on torch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the current HIP stream; every arithmetic kernel is in the
library.  ``data_ptr()`` of a contiguous fp32 / int64 CUDA tensor is handed over as a raw device pointer together
with ``torch.cuda.current_stream().cuda_stream`` so that the kernels are ordered with torch's own work.
"""
import contextlib
import ctypes
import gc
import os
import subprocess
import threading

import torch


@contextlib.contextmanager
def capturing(graph, stream):
    """``torch.cuda.graph(graph, stream=stream)`` with Python's cyclic collector held off.  Since torch 2.9 ``torch.cuda.graph`` no longer
    runs ``gc.collect()`` on entry, so a dead cycle that owns an older ``CUDAGraph`` (a previous phase's captured step) can be collected
    in the middle of a capture: its ``hipGraphExecDestroy`` is refused under the global capture mode, the C++ destructor throws and the
    process aborts ('Fatal Python error: Aborted ... Garbage-collecting', seen in the GPU suite).  Collect first, keep the collector off
    until the capture has ended."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream):
            yield
    finally:
        if was:
            gc.enable()


_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get('CDR_LIB_PATH') or os.path.join(_HERE, 'lib', 'libcdrhip.so')   # env: A/B builds only
ABI_VERSION = 59
SIGNIN_WORDS = 288          # CDR_SIGNIN_WORDS: the sign-in words cdr_adam_multi_dev's ``ticket`` points at

CDR_LOSS_MSE, CDR_LOSS_BCE = 0, 1
ACT_NONE, ACT_TANH, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3


class NativeLibraryError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def build(force=False):
    """Compile csrc/*.hip for gfx950 into lib/libcdrhip.so (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, 'csrc')
    cmd = ['make', '-C', csrc, '-j', str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.run(['make', '-C', csrc, 'clean'], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise NativeLibraryError('building libcdrhip.so failed:\n' + r.stdout[-4000:])
    return _LIB_PATH


_c_i64, _c_int, _c_f32, _c_ptr = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes, mirrors include/cdr_hip.h one to one
_SIGNATURES = {
    'cdr_ctx_create': [_c_int, ctypes.POINTER(_c_ptr)],
    'cdr_ctx_destroy': [_c_ptr],
    'cdr_abi_version': [],
    'cdr_ctx_scrub_next': [_c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_ctx_set_id_counters': [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, ctypes.c_size_t],
    'cdr_id_count_workspace_bytes': [_c_i64, ctypes.POINTER(ctypes.c_size_t)],
    'cdr_bpr_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_ptr, _c_ptr],
    'cdr_bpr_bwd_dense': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_f32,
                          _c_ptr, _c_ptr, _c_ptr],
    'cdr_point_fwd': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                      _c_f32, _c_ptr, _c_ptr, _c_ptr],
    'cdr_point_bwd_dense': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr,
                            _c_f32, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_gather_rows': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr],
    'cdr_gather_rows_multi': [_c_ptr, _c_int, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_scatter_add_rows_multi': [_c_ptr, _c_int, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr],
    'cdr_sscdr_map_loss': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_f32, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_ptr],
    'cdr_scale2_unless_one': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64],
    'cdr_scatter_add_rows': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr],
    'cdr_select_mapped': [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_i64, _c_ptr],
    'cdr_gemm_f32': [_c_ptr, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr,
                     _c_int, _c_int],
    'cdr_fullsort_scores_f32': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr],
    'cdr_fullsort_neg_sqdist_f32': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr],
    'cdr_act_bwd': [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64],
    'cdr_colsum': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_int],
    'cdr_linear_wgrad_small': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_linear_wgrad_small_workspace': [_c_i64, _c_int, _c_int, ctypes.POINTER(ctypes.c_size_t)],
    'cdr_linear_small': [_c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_int, _c_int, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_int],
    'cdr_mse_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_mse_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr],
    'cdr_bpr_fwd_grad': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_f32, _c_f32, _c_ptr,
                         _c_ptr, _c_ptr, _c_int],
    'cdr_loss_finish_sums': [_c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr],
    'cdr_bpr_step_fused_heads_words': [_c_i64, ctypes.POINTER(_c_i64)],
    'cdr_bpr_step_fused_kmajor_sizes': [_c_i64, _c_int, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)],
    'cdr_bpr_step_fused_kmajor': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int,
                                  _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_i64,
                                  _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_bpr_step_fused': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int,
                           _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_i64,
                           _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_bpr_step_fused_dev': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int,
                               _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_ptr,
                               _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_bpr_step_presort': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                             ctypes.c_size_t, _c_ptr],
    'cdr_bpr_step_from_diff': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                               _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_i64, _c_ptr, ctypes.c_uint32, _c_ptr, _c_ptr,
                               _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_batch_norm_sums': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_bpr_shard_local_step': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                 _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                 _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_route_triples_workspace_bytes': [_c_i64, _c_int, ctypes.POINTER(ctypes.c_size_t)],
    'cdr_route_triples': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_bpr_shard_plan_sizes': [_c_i64, _c_i64, _c_i64, _c_int, ctypes.POINTER(_c_i64), ctypes.POINTER(ctypes.c_size_t)],
    'cdr_bpr_shard_plan': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                           _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_gather_rows_norms': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr],
    'cdr_shard_norm_sums': [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_bpr_shard_step': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_f32, _c_f32, _c_f32,
                           _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_shard_owner_apply': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_i64, _c_int, _c_ptr, _c_f32, _c_f32, _c_f32,
                              _c_f32, _c_f32, _c_i64, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_overlap_remap_dev_workspace_bytes': [_c_i64, _c_i64, ctypes.POINTER(ctypes.c_size_t)],
    'cdr_overlap_remap_dev': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                              ctypes.c_size_t, _c_ptr],
    'cdr_point_step_fused': [_c_ptr, _c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr,
                             _c_ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                             _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_point_step_fused_kmajor': [_c_ptr, _c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr,
                                    _c_ptr, _c_ptr, _c_i64, _c_int, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                    _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_point_step_presort': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t, _c_ptr],
    'cdr_point_step_from_dot': [_c_ptr, _c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_i64, _c_ptr, ctypes.c_uint32, _c_ptr, _c_ptr, _c_ptr,
                                _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_sort_workspace_bytes': [_c_i64, _c_i64, ctypes.POINTER(ctypes.c_size_t)],
    'cdr_timing_enable': [_c_ptr, _c_int],
    'cdr_timing_collect': [_c_ptr, ctypes.POINTER(_c_int), ctypes.POINTER(_c_f32), _c_int, ctypes.POINTER(_c_int)],
    'cdr_sort_ids': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_rowwise_apply': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64,
                          _c_ptr, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_ptr, ctypes.c_uint32],
    'cdr_sort_ids_two_tables': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                _c_ptr, ctypes.c_size_t],
    'cdr_gemm_f32_ex': [_c_ptr, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr,
                        _c_ptr, _c_int, _c_int],
    'cdr_gather_rows_ld': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64],
    'cdr_scatter_add_rows_ld': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64],
    'cdr_scatter_rows_sorted': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64],
    'cdr_ordered_bwd': [_c_ptr, _c_int, _c_ptr, _c_int],
    'cdr_overlap_mask': [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr],
    'cdr_rowscale': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr],
    'cdr_bcast_add_act': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr],
    'cdr_bce_prob_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_bce_prob_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr],
    'cdr_frobenius_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_frobenius_bwd': [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_int],
    'cdr_sqnorm_normalize_fwd': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr],
    'cdr_sqnorm_normalize_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr],
    'cdr_triplet_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_ptr],
    'cdr_triplet_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                        _c_ptr],
    'cdr_spmm_csr_f32': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int, _c_ptr],
    'cdr_graph_layer_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr],
    'cdr_graph_layer_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr],
    'cdr_row_flags': [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, ctypes.c_size_t],
    'cdr_row_flags_layout': [_c_i64, _c_ptr],
    'cdr_graph_layer_fwd_rows': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr],
    'cdr_mul_one_plus': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_graph_layer_bwd_rows': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr],
    'cdr_transfer_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_i64, _c_f32, _c_f32, _c_ptr, _c_ptr],
    'cdr_transfer_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_i64, _c_f32, _c_f32, _c_ptr, _c_ptr],
    'cdr_bitgcf_mix_fwd': [_c_ptr] * 7 + [_c_i64, _c_i64, _c_int, _c_i64, _c_i64, _c_f32, _c_f32, _c_f32, ctypes.c_uint64, _c_ptr, ctypes.c_uint64,
                           ctypes.c_uint64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr],
    'cdr_bitgcf_mix_bwd': [_c_ptr] * 7 + [_c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_i64, _c_i64, _c_f32, _c_f32,
                           _c_f32, ctypes.c_uint64, _c_ptr, ctypes.c_uint64, ctypes.c_uint64, _c_ptr, _c_ptr, _c_ptr],
    'cdr_transfer_drop_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_i64, _c_f32, _c_f32, _c_f32, ctypes.c_uint64, _c_ptr,
                              ctypes.c_uint64, ctypes.c_uint64, _c_i64, _c_ptr, _c_ptr],
    'cdr_transfer_drop_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_i64, _c_f32, _c_f32, _c_f32, ctypes.c_uint64, _c_ptr,
                              ctypes.c_uint64, ctypes.c_uint64, _c_i64, _c_ptr, _c_ptr],
    'cdr_l2_normalize_fwd': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_i64, _c_ptr],
    'cdr_l2_normalize_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_int],
    'cdr_copy_cols': [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_i64, _c_int],
    'cdr_bitgcf_stack': [_c_ptr] * 5 + [_c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64],
    'cdr_bitgcf_unstack_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr],
    'cdr_colblock_mean_fwd': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_int, _c_ptr],
    'cdr_colblock_mean_bwd': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_int, _c_ptr],
    'cdr_dropout_dev': [_c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr, ctypes.c_uint64, _c_ptr],
    'cdr_gather_max2': [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64],
    'cdr_gather_max2_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr],
    'cdr_apf_prod': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr],
    'cdr_apf_prod_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr],
    'cdr_apf_combine': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr],
    'cdr_apf_combine_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr,
                            _c_ptr, _c_ptr],
    'cdr_maxmin_norm': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr],
    'cdr_maxmin_norm_bwd': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr],
    'cdr_natr_att_fwd': [_c_ptr] * 9 + [_c_i64, _c_int, _c_int] + [_c_ptr] * 4,
    'cdr_natr_att_bwd': [_c_ptr] * 9 + [_c_i64, _c_int, _c_int] + [_c_ptr] * 11,
    'cdr_dropout': [_c_ptr, _c_ptr, _c_i64, _c_f32, ctypes.c_uint64, _c_ptr],
    'cdr_embloss_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_embloss_bwd_dense': [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_route_workspace_bytes': [_c_i64, _c_int, ctypes.POINTER(ctypes.c_size_t)],
    'cdr_route_by_owner': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_permute_i64': [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr],
    'cdr_inverse_perm': [_c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_fullsort_topk_workspace_bytes': [_c_i64, _c_int, _c_i64, _c_i64, _c_int, _c_ptr],
    'cdr_fullsort_topk_f32': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_int, _c_ptr,
                              _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_dedup_workspace_bytes': [_c_i64, _c_ptr],
    'cdr_dedup_sorted': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_segsum_rows': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr],
    'cdr_interleave_shards': [_c_ptr, _c_ptr, _c_int, _c_i64, _c_i64, _c_i64, _c_ptr],
    'cdr_gather_owned_rows': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_int, _c_int, _c_ptr],
    'cdr_gather_block_rows': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr],
    'cdr_comm_unique_id': [_c_ptr],
    'cdr_comm_init': [_c_ptr, _c_int, _c_int, _c_ptr],
    'cdr_comm_destroy': [_c_ptr],
    'cdr_comm_info': [_c_ptr, _c_ptr, _c_ptr],
    'cdr_a2a_ids': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_a2a_rows': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int],
    'cdr_a2a_plan': [_c_int, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_allgather_scores': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_allreduce_sum_f32': [_c_ptr, _c_ptr, _c_ptr, _c_i64],
    'cdr_scatter_add_block_rows': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr],
    'cdr_ids_pack32': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr],
    'cdr_ids_unpack32': [_c_ptr, _c_ptr, _c_int, _c_i64, _c_ptr, _c_ptr],
    'cdr_point_partial_dot': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_point_grad_from_dot': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr, _c_ptr,
                                _c_ptr, _c_ptr],
    'cdr_bpr_partial_diff': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    'cdr_bpr_grad_from_diff': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_ptr,
                               _c_ptr, _c_ptr, _c_ptr],
    'cdr_topk_merge_shards': [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_int, _c_int, _c_ptr, _c_ptr],
    'cdr_adam_dense_dev': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_ptr],
    'cdr_inc_i64': [_c_ptr, _c_ptr],
    'cdr_point_fwd_pair': [_c_ptr, _c_ptr, _c_int] + [_c_ptr] * 4 + [_c_int] + [_c_ptr] * 10,
    'cdr_point_bwd_dense_pair': [_c_ptr, _c_ptr] + [_c_ptr] * 4 + [_c_int] + [_c_ptr] * 12,
    'cdr_point_fwd_pair_ex': [_c_ptr, _c_ptr, _c_int] + [_c_ptr] * 4 + [_c_int, _c_int] + [_c_ptr] * 10,
    'cdr_embloss_bwd_dense_pair': [_c_ptr, _c_ptr, _c_ptr, _c_int] + [_c_ptr] * 8,
    'cdr_scalar_mix': [_c_ptr, _c_int, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr],
    'cdr_point_fwd_grad': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr, _c_ptr, _c_ptr],
    'cdr_adam_multi_dev': [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_ptr],
    'cdr_adam_multi_dev_produce': [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_ptr,
                                   _c_ptr, _c_int],
    'cdr_neg_sample_alias': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, ctypes.c_uint64, _c_ptr, _c_ptr],
    'cdr_neg_sample_uniform': [_c_ptr, _c_ptr, _c_i64, _c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, ctypes.c_uint64,
                               _c_ptr, _c_ptr],
    'cdr_batch_produce': [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr,
                          _c_ptr, _c_i64, _c_ptr, _c_ptr, ctypes.c_uint64, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_batch_produce_jobs': [_c_ptr, _c_ptr, _c_int],
    'cdr_sscdr_pair_sample': [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, ctypes.c_uint64, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    'cdr_bpr_fwd_grad_kmajor': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_f32, _c_f32, _c_ptr, _c_ptr,
                                _c_ptr, _c_ptr, _c_ptr],
    'cdr_sort_ids_small': [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64],
    'cdr_rowwise_apply_rows': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_f32,
                               _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_ptr, _c_int],
    'cdr_rowwise_apply_scaled': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_ptr,
                                 _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64, _c_ptr, _c_int],
    'cdr_bpr_step_small': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int,
                           _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64],
    'cdr_map_step_plan': [_c_int, _c_ptr, _c_ptr, _c_i64, ctypes.POINTER(ctypes.c_size_t)],
    'cdr_map_step_unique': [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr,
                            _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f32, _c_f32, _c_f32, _c_f32,
                            _c_f32, _c_ptr, _c_ptr, ctypes.c_size_t],
    'cdr_lazy_adam_prepare': [_c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32,
                              _c_ptr, _c_i64, _c_ptr, _c_i64],
    'cdr_lazy_adam_prepare_sort_small': [_c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                         _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_int],
    'cdr_lazy_adam_apply': [_c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f32, _c_f32,
                            _c_f32, _c_f32, _c_f32, _c_ptr, _c_i64, _c_ptr],
    'cdr_lazy_adam_apply_produce': [_c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f32, _c_f32,
                            _c_f32, _c_f32, _c_f32, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_int],
    'cdr_lazy_adam_flush': [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_ptr, _c_i64, _c_ptr],
    'cdr_conet_plan': [_c_int, _c_ptr, _c_i64, ctypes.POINTER(_c_int), ctypes.POINTER(ctypes.c_size_t)],
    'cdr_conet_defer_finish': [_c_ptr, _c_int],
    'cdr_conet_fwd': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int,
                      _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                      _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t, ctypes.POINTER(_c_int)],
    'cdr_conet_bwd': [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                      _c_ptr, _c_ptr, _c_ptr, ctypes.c_size_t, _c_int],
    'cdr_conet_fullsort_supported': [_c_int, _c_int, _c_ptr],
    'cdr_conet_fullsort': [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64],
    'cdr_conet_fullsort_users': [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int, _c_i64, _c_i64, _c_int, _c_int, _c_ptr, _c_ptr,
                                 _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64],
    'cdr_overlap_remap': [ctypes.c_char_p, _c_ptr, _c_ptr, _c_i64, ctypes.c_char_p, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr],
    'cdr_revoke_map': [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr],
    'cdr_adam_dense': [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _c_i64],
}

class BatchJob(ctypes.Structure):
    """``cdr_batch_job`` of include/cdr_hip.h."""
    _fields_ = [('users_all', _c_ptr), ('items_all', _c_ptr), ('n_rows', _c_i64), ('cursor', _c_ptr), ('S', _c_i64),
                ('k', ctypes.c_int32), ('pointwise', ctypes.c_int32), ('dist', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('lo0', _c_i64), ('hi0', _c_i64), ('lo1', _c_i64), ('hi1', _c_i64),
                ('keys', _c_ptr), ('prob', _c_ptr), ('alias', _c_ptr), ('n_keys', _c_i64),
                ('used_indptr', _c_ptr), ('used_indices', _c_ptr), ('seed', ctypes.c_uint64),
                ('out_users', _c_ptr), ('out_items', _c_ptr), ('out_neg', _c_ptr), ('fail_flag', _c_ptr)]


ORD_MAX_SEGS, ORD_MAX_LISTS, ORD_MAX_TOTAL = 4, 4, 16384


class OrdSeg(ctypes.Structure):
    """``cdr_ord_seg`` of include/cdr_hip.h."""
    _fields_ = [('ids', _c_ptr), ('n', _c_i64), ('coef', _c_ptr), ('sign', _c_f32), ('go', _c_ptr), ('go_scale', _c_f32),
                ('X', _c_ptr), ('xid', _c_ptr), ('Y', _c_ptr), ('yid', _c_ptr), ('x_stride', _c_i64),
                ('R', _c_ptr), ('r_stride', _c_i64), ('norm', _c_ptr), ('reg_weight', _c_f32), ('B', _c_i64)]


class OrdList(ctypes.Structure):
    """``cdr_ord_list`` of include/cdr_hip.h."""
    _fields_ = [('g', _c_ptr), ('g_stride', _c_i64), ('nseg', ctypes.c_int32), ('accumulate', ctypes.c_int32), ('seg', OrdSeg * ORD_MAX_SEGS)]


_lib = None
_lib_lock = threading.Lock()
_ctx = {}


def exported_symbols():
    return sorted(list(_SIGNATURES) + ['cdr_last_error'])


def load():
    """dlopen the library (no GPU needed for that) and type every entry point.  Raises loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(_LIB_PATH):
            raise NativeLibraryError(
                f'{_LIB_PATH} is missing: build it with `make -C recbole-cdr_amd/csrc` (or __graft_entry__.build()). '
                'There is no CPU / eager fallback for this path.')
        lib = ctypes.CDLL(_LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export what the header declares
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        lib.cdr_last_error.argtypes = []
        lib.cdr_last_error.restype = ctypes.c_char_p
        if lib.cdr_abi_version() != ABI_VERSION:
            raise NativeLibraryError(f'libcdrhip ABI {lib.cdr_abi_version()} != binding ABI {ABI_VERSION}; rebuild')
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load().cdr_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what} failed (code {rc}): {msg}')


def _dev_index(device):
    idx = torch.device(device).index
    return torch.cuda.current_device() if idx is None else idx


def ctx(device):
    """The cdr_ctx (reduction partials, long-segment scratch, timing ring) of (device, CURRENT stream), created on first
    use.  One per stream: two steps pipelined on two HIP streams (shard.run_pipelined) must not share reduction scratch."""
    idx = _dev_index(device)
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    if key not in _ctx:
        h = _c_ptr()
        _check(load().cdr_ctx_create(idx, ctypes.byref(h)), 'cdr_ctx_create')
        _ctx[key] = h
        if _timing_cap.get(idx, 0):
            _check(load().cdr_timing_enable(h, _timing_cap[idx]), 'cdr_timing_enable')
    return _ctx[key]


def stream():
    return _c_ptr(torch.cuda.current_stream().cuda_stream)


_alive = []      # tensors whose pointers were taken for the call being assembled (see ptr / call)


def ptr(t, dtype=None):
    """Raw device pointer of a contiguous CUDA tensor (None -> NULL).  The tensor is kept referenced until the native
    call it is an argument of has been enqueued: a temporary (``x.contiguous()`` of a strided view, ``t.to(dev)``) would
    otherwise be returned to the caching allocator before the launch and could be handed to the NEXT temporary."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeLibraryError('libcdrhip works on ROCm device tensors only (got a CPU tensor); there is no CPU path')
    if not t.is_contiguous():
        raise ValueError('tensor must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'expected {dtype}, got {t.dtype}')
    _alive.append(t)
    return _c_ptr(t.data_ptr())


def f32(t):
    return ptr(t, torch.float32)


def i64(t):
    return ptr(t, torch.int64)


def raw(t):
    return ptr(t)


def call(name, *args):
    try:
        rc = getattr(load(), name)(*args)
    finally:
        _alive.clear()           # enqueued: stream order now protects the buffers
    _check(rc, name)


TAGS = {1: 'bpr_fwd_kernel', 2: 'point_fwd_kernel', 3: 'bpr_fwd_grad_kernel', 4: 'rowwise_apply_kernel(users)',
        5: 'rowwise_apply_kernel(items)', 6: 'sort_ids', 7: 'point_fwd_grad_kernel',
        8: 'bpr_partial_diff_kernel', 9: 'bpr_grad_from_diff_kernel',
        10: 'point_partial_dot_kernel', 11: 'point_grad_from_dot_kernel',
        12: 'conet_fwd_kernel', 13: 'conet_bwd_kernel', 14: 'conet_wgrad_kernel', 15: 'bpr_fwd_kmajor_kernel', 16: 'map_step_kernel',
        17: 'occ_flags_kernel', 18: 'bpr_fwd_apply_kernel', 19: 'batch_norms_kernel', 20: 'conet_fb_kernel'}


_timing_cap = {}     # device index -> ring capacity requested for every context (= stream) of that device


def timing_enable(device, capacity):
    """HIP-event brackets around the hot kernels on every stream of ``device`` (capacity 0 switches them off)."""
    idx = _dev_index(device)
    _timing_cap[idx] = int(capacity)
    ctx(device)                                    # make sure the current stream has a context
    for (d, _s), h in _ctx.items():
        if d == idx:
            call('cdr_timing_enable', h, int(capacity))


def timing_collect(device, max_n=65536):
    """-> list of (kernel name, milliseconds), per stream in launch order, measured by HIP events on the launch stream."""
    idx = _dev_index(device)
    res = []
    for (d, _s), h in _ctx.items():
        if d != idx:
            continue
        tags = (_c_int * max_n)()
        ms = (_c_f32 * max_n)()
        n = _c_int(0)
        call('cdr_timing_collect', h, tags, ms, max_n, ctypes.byref(n))
        res += [(TAGS.get(tags[i], str(tags[i])), float(ms[i])) for i in range(n.value)]
    return res
