"""ctypes binding of libsamble_hip.so (include/samble.h).

There is no CPU or eager fallback: if the library is missing or a call fails, a
`SambleError` is raised.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
(or `make -C samble_amd/csrc`).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsamble_hip.so")


class SambleError(RuntimeError):
    pass


_SIGNATURES = {
    "samble_version": (c_char_p, []),
    "samble_last_error": (c_char_p, []),
    "samble_abi_version": (c_int, []),
    "samble_knn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "samble_knn_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_proj_workspace_bytes": (c_size_t, [c_int, c_int]),
    "samble_proj_fwd_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                    c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "samble_proj_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p,
                                    c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "samble_proj_fwd_tri_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                    c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "samble_proj_bwd_tri_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p,
                                    c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_proj_fwd_tri_workspace_bytes": (c_size_t, []),
    "samble_proj_w_image_bytes": (c_size_t, []),
    "samble_inverse_neighbors_workspace_bytes": (c_size_t, [c_int, c_int]),
    "samble_inverse_neighbors": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                         c_void_p]),
    "samble_proj_bwd_tri_workspace_bytes": (c_size_t, [c_int, c_int]),
    "samble_attn_fwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                    c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "samble_attn_colsum_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int, c_int,
                                       c_int, c_void_p, c_void_p]),
    "samble_stat_score_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "samble_score_workspace_bytes": (c_size_t, [c_int, c_int]),
    "samble_sparse_score_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    "samble_select_chain_supported": (c_int, [c_int, c_int, c_int]),
    "samble_select_chain_workspace_bytes": (c_size_t, [c_int, c_int]),
    "samble_sparse_score_map_quantiles_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_uint,
                                                      c_void_p, c_void_p]),
    "samble_bin_plan_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_int,
                                    c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_uint, c_void_p, c_void_p]),
    "samble_select_chain_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_uint, c_void_p, c_void_p]),
    "samble_select_chain_status_async": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "samble_edge_glue_partials_bytes": (c_size_t, []),
    "samble_edge_glue_constants_bytes": (c_size_t, []),
    "samble_edge_glue_statistics_bytes": (c_size_t, []),
    "samble_edge_glue_pooled_bytes": (c_size_t, []),
    "samble_edge_bn1_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                    c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "samble_edge_bn2_out_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "samble_edge_bwd_pre_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "samble_edge_bwd_post_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "samble_interp_blend_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                            c_void_p]),
    "samble_interp_blend_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "samble_interp_blend_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "samble_linear_image_bytes": (c_size_t, [c_int]),
    "samble_linear_weight_images_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "samble_linear_weight_images_t_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "samble_linear_two_plane_build": (c_int, []),
    "samble_linear_fwd_tri_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                          c_int64, c_int64, c_void_p]),
    "samble_linear_sign_bytes": (c_size_t, [c_int, c_int, c_int]),
    "samble_linear_chain_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64,
                                        c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "samble_linear_amax_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "samble_linear_amax_fwd_tri_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
    "samble_linear_dx_tri_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                         c_void_p, c_void_p]),
    "samble_linear_dw_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "samble_linear_dw_tri_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "samble_linear_dw_t_tri_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "samble_linear_weight_images_pair_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                     c_void_p]),
    "samble_linear_fwd_cm_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int64,
                                         c_void_p]),
    "samble_linear_dw_cm_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    "samble_bn_train_workspace_bytes": (c_size_t, [c_int, c_int]),
    "samble_bn_train_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "samble_bn_train_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "samble_bn_train_stats_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_bn_train_apply_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "samble_bn_train_bwd_sums_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "samble_bn_train_bwd_apply_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "samble_amax_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "samble_amax_bwd_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                    c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_zscore_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "samble_quantiles_workspace_bytes": (c_size_t, []),
    "samble_batch_quantiles_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_blend_boundaries_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_int, c_void_p]),
    "samble_bin_assign_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "samble_alloc_counts_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "samble_bin_select_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_float, c_void_p, c_void_p]),
    "samble_bin_select_seeded_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64, c_int, c_int, c_int,
                                             c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "samble_exp1_noise_f32": (c_int, [c_uint64, c_uint64, c_int, c_int, c_void_p, c_void_p]),
    "samble_attn_heads_fwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                          c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int64,
                                          c_void_p, c_void_p]),
    "samble_attn_heads_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "samble_attn_heads_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                          c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int64,
                                          c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                          c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t,
                                          c_void_p]),
    "samble_gather_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "samble_gather_points_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "samble_n2p_attn_fwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "samble_n2p_attn_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "samble_n2p_attn_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_segment_sum_rows_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "samble_segment_sum_rows_pair_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int64,
                                                 c_void_p, c_void_p, c_void_p]),
    "samble_attn_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "samble_attn_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                    c_int, c_void_p, c_size_t, c_void_p]),
    "samble_edge_partial_count": (c_int, []),
    "samble_edge_gather_sums_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "samble_edge_mlp_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "samble_edge_mlp_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "samble_group_gather_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "samble_fps_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "samble_timing_select": (c_int, [c_uint64]),
    "samble_timing_read": (c_int, [c_int, POINTER(c_float), POINTER(c_float), POINTER(c_int)]),
    "samble_attn_map_row_stride": (c_int, [c_int, c_int]),
    "samble_tri_image_bytes": (c_size_t, [c_int, c_int, c_int]),
    "samble_tri_split_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "samble_proj_fwd_split_tri_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                              c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_tri_k_logit_form": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "samble_tri_split_qkv_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "samble_attn_rows_bwd_tri_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "samble_attn_rows_bwd_tri_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                             c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int64,
                                             c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int,
                                             c_void_p, c_size_t, c_void_p]),
    "samble_attn_rows_fwd_tri_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                             c_void_p, c_void_p]),
    "samble_nn_masks_bytes": (c_size_t, [c_int, c_int]),
    "samble_nn_prepare": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_attn_stats_nl_tri_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "samble_attn_rows_fwd_recompute_tri_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                       c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "samble_attn_stats_tri_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "samble_attn_stats_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                      c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "samble_sparse_score_map_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "samble_attn_rows_fwd_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_int,
                                         c_int, c_int, c_int, c_void_p, c_void_p]),
    "samble_attn_rows_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                         c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                         c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)
ABI_VERSION = 6  # include/samble.h SAMBLE_ABI_VERSION this binding's argument table was written against

_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library once; raise SambleError (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process (the one
    # that owns the tensors' device context), so it has to be loaded before our library binds to
    # that SONAME.  Loaded the other way round, /opt/rocm's copy wins and sees no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise SambleError(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run "
            "`python -c \"import __graft_entry__ as g; g.build()\"` or `make -C samble_amd/csrc`. "
            "samble_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    # a library built from another revision of include/samble.h would take shifted arguments: refuse it before any call
    try:
        lib.samble_abi_version.restype = c_int
        built = int(lib.samble_abi_version())
    except AttributeError:
        built = None
    if built != ABI_VERSION:
        raise SambleError(f"{LIB_PATH} implements ABI version {built}, this binding is written against {ABI_VERSION}: "
                          "rebuild it (`make -C samble_amd/csrc`)")
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise SambleError(f"{LIB_PATH} does not export {name}: the library is stale, rebuild it") from None
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point and turn a failure into SambleError."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.samble_last_error().decode(errors="replace")
        raise SambleError(f"{name} failed with code {rc}: {msg}")


def query(name: str, *args) -> int:
    return int(getattr(load(), name)(*args))


# ids of the measurement hook (include/samble.h SAMBLE_T_*)
TIMED_KERNELS = {
    "attn_stats": 1, "attn_rows": 2, "bwd_dv": 3, "knn": 4, "attn_fwd": 5, "bwd_dq": 6, "bwd_dk": 7, "proj_fwd": 8,
    "proj_dx": 9, "proj_dw": 10, "tri_split": 11, "knn_prep": 12, "sparse_score": 13, "quantiles": 14, "bin_assign": 15,
    "alloc_counts": 16, "bin_select": 17, "bwd_prep": 18, "gather": 19, "bwd_rows_f32": 22, "nn_prepare": 23,
    "edge_fwd": 24, "edge_bwd": 25, "n2p_fwd": 26, "n2p_bwd": 27, "inv_nn": 28, "seg_sum": 29, "edge_sums": 30,
    "knn_small": 31, "lin_fwd": 32, "lin_dx": 33, "lin_dw": 34, "lin_amax": 35, "lin_amax_bwd": 36, "bn_fwd": 37, "lin_chain": 38, "bn_bwd": 39,
}


def timing_select(names) -> None:
    """Record HIP events around the library's launches of the named kernels (empty = off)."""
    mask = 0
    for n in names:
        mask |= 1 << TIMED_KERNELS[n]
    call("samble_timing_select", mask)


def timing_read(name):
    """(mean ms, median ms, launches seen) of one selected kernel; None if it never ran."""
    mean, med, cnt = c_float(), c_float(), c_int()
    rc = load().samble_timing_read(TIMED_KERNELS[name], ctypes.byref(mean), ctypes.byref(med), ctypes.byref(cnt))
    if rc != 0:
        return None
    return float(mean.value), float(med.value), int(cnt.value)
