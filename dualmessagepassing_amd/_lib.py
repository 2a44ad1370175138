"""ctypes binding of libdmp_hip.so (the C ABI declared in include/dmp_hip.h).

The product path has no CPU fallback: if the shared library cannot be built or
loaded, or a tensor is not on an AMD GPU, the ops raise.
"""
import ctypes
import os
import threading

from . import _build

c_i64 = ctypes.c_int64
c_double = ctypes.c_double
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_ptr = ctypes.c_void_p
c_size = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/dmp_hip.h one to one.
SIGNATURES = {
    "dmp_abi_version": (c_int, []),
    "dmp_last_hip_error": (ctypes.c_char_p, []),
    "dmp_csr_workspace_words": (c_size, [c_i64, c_i64]),
    "dmp_csr_build": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_incidence_build": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "dmp_degree_coef": (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_collate": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                            c_ptr, c_ptr, c_ptr]),
    "dmp_add_reversed_edges": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64,
                                       c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_line_graph_count": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_line_graph_fill": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_first_edge_of_id": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "dmp_dedupe_table_words": (c_size, [c_i64]),
    "dmp_dedupe_first": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "dmp_subiso_node_weights": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_dual_subisomorphisms": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                         c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_subiso_enumerate": (c_i64, [c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                     c_i64]),
    "dmp_subiso_count_batch": (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                       c_ptr, c_int]),
    "dmp_random_walks": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, ctypes.c_uint64, c_ptr, c_ptr, c_ptr]),
    "dmp_sample_in_edges": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, ctypes.c_uint64, c_ptr, c_ptr]),
    "dmp_pool_index": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                               c_ptr, c_ptr]),
    "dmp_pattern_edge_active": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_subiso_edge_weights": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                        c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "dmp_class_tiles_workspace_words": (c_size, [c_i64, c_int]),
    "dmp_class_tiles": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_csr_keep_scratch_words": (c_i64, [c_i64]),
    "dmp_csr_keep": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_class_tiles_gated": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_scan_workspace_words": (c_size, [c_i64]),
    "dmp_exclusive_scan_i64": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "dmp_seg_sum": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr]),
    "dmp_seg_sum2": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32, c_f32, c_ptr, c_i64,
                             c_int, c_ptr]),
    "dmp_small_gemm_jobs": (c_int, [c_ptr, c_int, c_ptr]),
    "dmp_pool_index_jobs": (c_int, [c_ptr, c_int, c_ptr]),
    "dmp_collate_jobs": (c_int, [c_ptr, c_int, c_ptr]),
    "dmp_csr_build_graphs_max_nodes": (c_int, []),
    "dmp_csr_build_graphs": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                     c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_pool_weight_sums": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_gate_compact_hist_nodes": (c_int, []),
    "dmp_gate_compact": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                 c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_out_degrees": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "dmp_seg_sum2_graphs_max_nodes": (c_int, []),
    "dmp_seg_sum2_graphs": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32,
                                    c_ptr, c_i64, c_ptr]),
    "dmp_seg_sum2_graphs_masked": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32,
                                           c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    "dmp_seg_sum2_rows": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_f32, c_f32, c_ptr, c_i64, c_ptr]),
    "dmp_incidence_keep": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_seg_sum2_tiled": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32,
                                   c_ptr, c_i64, c_ptr]),
    "dmp_gather_rows": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "dmp_gather_select": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_f32, c_f32,
                                  c_ptr, c_i64, c_ptr]),
    "dmp_colsum_partial_rows": (c_i64, [c_i64, c_int]),
    "dmp_gate_residual": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "dmp_scale_rows_colsum": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_relu_bwd_colsum": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_f32, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_edge_combine_bwd_g_colsum": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_colsum_partials": (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr]),
    "dmp_reduce_partials": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_int, c_ptr]),
    "dmp_add_bias_relu": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_f32, c_ptr, c_i64, c_ptr]),
    "dmp_heads_forward": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_f32, c_ptr]),
    "dmp_heads_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_f32, c_ptr]),
    "dmp_fold_layers": (c_int, [c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "dmp_unfold_layers": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "dmp_smallk_embed_gate": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "dmp_smallk_embed_live": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "dmp_smallk_atb_blocks": (c_i64, [c_i64]),
    "dmp_smallk_embed_cols": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr]),
    "dmp_smallk_atb_cols_masked": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "dmp_smallk_atb_cols_rows": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "dmp_smallk_atb": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "dmp_smallk_atb_jobs": (c_int, [c_ptr, c_int, c_int, c_int, c_ptr]),
    "dmp_l0_pack_jobs": (c_int, [c_ptr, c_int, c_ptr]),
    "dmp_count_loss": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_f32, c_ptr, c_ptr, c_ptr]),
    "dmp_reduce_partials_multi": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    "dmp_l0_pack": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr]),
    "dmp_l0_edge_fwd_masked": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int,
                                       c_f32, c_ptr, c_i64, c_ptr]),
    "dmp_l0_bwd_w_blocks": (c_i64, [c_i64]),
    "dmp_l0_node_pack_rows": (c_i64, []),
    "dmp_l0_node_fwd": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_f32, c_ptr,
                                c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    "dmp_bn_partial_rows": (c_i64, [c_i64, c_int]),
    "dmp_bn_train_fwd": (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_f32, c_f32, c_ptr, c_ptr, c_int, c_f32, c_ptr, c_ptr, c_ptr,
                                 c_i64, c_ptr]),
    "dmp_bn_train_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_ptr, c_int, c_f32, c_ptr, c_ptr, c_ptr, c_i64,
                                 c_ptr]),
    "dmp_bn_train_fwd_rows": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_int, c_ptr, c_ptr, c_f32, c_f32, c_ptr, c_ptr, c_int, c_f32, c_ptr, c_ptr,
                                      c_ptr, c_i64, c_ptr]),
    "dmp_bn_train_bwd_rows": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_int, c_ptr, c_int, c_f32, c_ptr, c_ptr, c_ptr,
                                      c_i64, c_ptr]),
    "dmp_l0_bwd_w_masked": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "dmp_scalar_filter_gates": (c_int, [c_ptr, c_int, c_i64, c_ptr, c_i64, c_ptr]),
    "dmp_csr_pair_workspace_words": (ctypes.c_size_t, [c_i64]),
    "dmp_csr_build_pair": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                   c_ptr, c_ptr]),
    "dmp_heads_blend": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr]),
    "dmp_concat_pairs": (c_int, [c_ptr, c_int, c_ptr]),
    "dmp_table_rows": (c_int, [c_ptr, c_int, c_ptr]),
    "dmp_len_masks": (c_int, [c_ptr, c_int, c_i64, c_ptr]),
    "dmp_pack_segments": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr]),
    "dmp_adamw_step": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_double, c_double, c_double, c_double,
                               c_double, c_i64, c_ptr]),
    "dmp_adamw_step_skip": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_double, c_double, c_double, c_double,
                                    c_double, c_i64, c_ptr, c_ptr, c_int, c_ptr]),
    "dmp_adamw_step_dev": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_double, c_double, c_double, c_double,
                                   c_ptr, c_ptr, c_int, c_ptr]),
    "dmp_adamw_step_segments": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_double, c_double,
                                        c_double, c_double, c_ptr, c_int, c_ptr]),
    "dmp_adamw_step_guarded": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_double, c_double, c_double, c_double,
                                       c_ptr, c_ptr, c_int, c_ptr, c_int, c_ptr]),
    "dmp_edge_select_build": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_edge_select_nodes": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_edge_fwd_fused": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                                   c_i64, c_int, c_f32, c_ptr, c_i64, c_ptr]),
    "dmp_edge_fwd_typed": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                   c_ptr, c_i64, c_i64, c_int, c_f32, c_ptr, c_i64, c_ptr]),
    "dmp_bwd_z_typed_arow": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_f32,
                                     c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    "dmp_mask_slots": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_pool_relu_bwd_blocks": (c_i64, [c_i64, c_int]),
    "dmp_pool_relu_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_f32, c_ptr, c_i64,
                                  c_ptr, c_ptr, c_ptr]),
    "dmp_relu_bwd_gathered_colsum": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_f32, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_dev_set_mfma_variant": (None, [c_int]),
    "dmp_dev_set_exact_fp32": (None, [c_int]),
    "dmp_dev_get_exact_fp32": (c_int, []),
    "dmp_rel_gemm": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64,
                             c_int, c_ptr, c_i64, c_ptr]),
    "dmp_rel_atb_blocks": (c_i64, [c_int]),
    "dmp_rel_atb": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_ptr, c_ptr]),
    "dmp_atb_typed_blocks_h": (c_i64, [c_i64, c_int]),
    "dmp_atb_rows_blocks_h": (c_i64, [c_i64, c_int, c_int, c_int]),
    "dmp_atb_rows_plain": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr]),
    "dmp_atb_rows_masked": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "dmp_atb_jobs_blocks_h": (c_i64, [c_i64, c_int, c_int]),
    "dmp_atb_rows_jobs_h": (c_int, [c_ptr, c_int, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "dmp_atb_tile_jobs_blocks": (c_i64, [c_i64, c_int, c_int]),
    "dmp_mfma_partial_rows_h": (c_i64, [c_i64, c_int]),
    "dmp_gemm_k64": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_i64, c_ptr]),
    "dmp_atb_typed": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr]),
    "dmp_out_fwd_fused_rows": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr,
                                       c_i64, c_ptr]),
    "dmp_bwd_h1_fused_rows": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_f32, c_ptr,
                                      c_i64, c_ptr, c_ptr, c_ptr]),
    "dmp_kept_rows_scratch_words": (c_i64, [c_i64]),
    "dmp_kept_rows": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_l0_edge_fwd_rows": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int,
                                     c_f32, c_ptr, c_i64, c_ptr]),
    "dmp_l0_bwd_w_rows": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "dmp_typed_partial_rows": (c_i64, [c_i64, c_int]),
    "dmp_out_fwd_typed": (c_int, [c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr]),
    "dmp_bwd_h1_typed": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_f32, c_ptr, c_i64,
                                 c_ptr, c_ptr, c_ptr]),
    "dmp_kept_rows_jobs": (c_int, [c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_row_mask_bits_jobs": (c_int, [c_ptr, c_int, c_ptr]),
    "dmp_atb2_blocks": (c_i64, [c_i64, c_int]),
    "dmp_atb2_jobs": (c_int, [c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr]),
    "dmp_bwd_z_w": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_f32, c_f32,
                            c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_out_fwd_typed_codes": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr]),
    "dmp_bwd_h1_w_blocks": (c_i64, [c_i64]),
    "dmp_bwd_h1_w": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_int, c_f32, c_ptr, c_i64,
                             c_ptr, c_ptr, c_ptr, c_ptr]),
    "dmp_row_mask_bits": (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    "dmp_row_mask_rows": (c_int, [c_ptr, c_i64, c_int, c_i64, c_ptr, c_ptr]),
    "dmp_bwd_z_fused": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                                c_f32, c_f32, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "dmp_gemm_k128": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_i64, c_int, c_ptr]),
    "dmp_edge_combine": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                 c_int, c_int, c_f32, c_ptr, c_i64, c_ptr]),
    "dmp_relu_bwd_g_colsum": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_f32, c_ptr, c_i64,
                                      c_ptr, c_ptr]),
    "dmp_edge_combine_bwd_g": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "dmp_compgcn_agg": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int,
                                c_ptr, c_i64, c_ptr]),
    "dmp_compgcn_agg_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                                    c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
}

ABI_VERSION = 84
# ``_lib.VALIDATE = True``: index builds read back the kernels' status word (one host sync each) and raise on an edge endpoint
# or a lookup index outside its range -- otherwise such an entry is dropped from the CSR and gathers read row 0 (validate
# datasets once with harness.validate_samples, or run a debugging pass with this attribute set)
VALIDATE = False
ERRORS = {-1: "DMP_ERR_BAD_ARG", -2: "DMP_ERR_UNSUPPORTED", -3: "DMP_ERR_HIP"}


class DmpError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


def load(build_if_missing=True):
    """Load (building first if needed) libdmp_hip.so; raises if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB_PATH
        if build_if_missing:
            try:
                path = _build.build_lib()
            except Exception as e:
                if not os.path.exists(path):
                    raise
                # A failed build leaves the library of an EARLIER source state in place.  It is loaded only when its content
                # hash still matches the sources (a box without a compiler that was handed a current library); a STALE one is
                # refused -- it once hid a compile error behind green tests -- unless DMP_ALLOW_STALE_LIB=1 asks for it.
                stale = _build._stale()
                if stale and os.environ.get("DMP_ALLOW_STALE_LIB") != "1":
                    raise DmpError("building libdmp_hip.so failed (%s) and the existing %s is STALE against the sources: refusing "
                                   "to run old kernels (DMP_ALLOW_STALE_LIB=1 overrides)" % (str(e).splitlines()[0][:300], path)) from e
                import sys
                print("dualmessagepassing_amd: building libdmp_hip.so failed (%s); loading the existing %s, which %s" %
                      (str(e).splitlines()[0][:200], path, "is STALE against the sources (DMP_ALLOW_STALE_LIB=1)" if stale else "matches the sources"),
                      file=sys.stderr)
        if not os.path.exists(path):
            raise DmpError("libdmp_hip.so is missing (%s); run __graft_entry__.build()" % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if lib.dmp_abi_version() != ABI_VERSION:
            raise DmpError("libdmp_hip.so ABI %d != binding ABI %d" % (lib.dmp_abi_version(), ABI_VERSION))
        if os.environ.get("DMP_EXACT_FP32") == "1":   # development switch: f32-input MFMA instead of the bf16x6 products
            lib.dmp_dev_set_exact_fp32(1)
        _lib = lib
    return _lib


def check(code, what):
    if code == 0:
        return
    msg = ERRORS.get(code, "error %d" % code)
    if code == -3:
        msg += ": " + (load().dmp_last_hip_error() or b"").decode(errors="replace")
    raise DmpError("%s failed: %s" % (what, msg))


def ptr(t):
    """Device pointer of a torch tensor (or NULL for None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    """Raw ``hipStream_t`` of torch's current stream on the current device (what every launch is ordered on)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


_CAPTURE_PINS = []         # pins of recordings nobody owns (a bare torch.cuda.graph around library calls)
_PIN_OWNER = []            # the innermost recording's own list (dp.StepGraph): its pins die with the recording


class capture_pins:
    """``with capture_pins() as pins:`` around a stream capture -- tensors pinned inside (``pin_for_capture``) go to ``pins``,
    which the owner of the recording keeps for as long as it keeps the recording."""

    def __enter__(self):
        self.pins = []
        _PIN_OWNER.append(self.pins)
        return self.pins

    def __exit__(self, *exc):
        _PIN_OWNER.pop()
        return False


def pin_for_capture(*tensors):
    """A tensor that was allocated OUTSIDE a stream capture (a memoised index array) and is about to be read by kernels
    being recorded must outlive the recording: the memo that owns it may evict it later.  Inside a capture this keeps a
    reference in the recording's own pin list (``capture_pins``; ``dp.StepGraph`` drops it with the recording) or, for a
    capture nobody registered, for the life of the process; outside a capture it does nothing."""
    import torch
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        (_PIN_OWNER[-1] if _PIN_OWNER else _CAPTURE_PINS).extend(t for t in tensors if t is not None)


def require_gpu(*tensors):
    """Every tensor of a launch must live on the GPU that is CURRENT: the library launches on the current device's
    stream (``stream_ptr``) and never switches devices itself, so a tensor of another GPU would be handed to the wrong
    device's stream.  Module ``forward``s enter the device of their inputs (``on_input_device``); anything that reaches
    an op with a foreign tensor raises here instead of faulting on the GPU."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise DmpError(
                "dualmessagepassing_amd ops run on an AMD GPU only (tensor on %s); "
                "there is no CPU fallback in the product path" % (t.device,))
        if cur is None:
            import torch
            cur = torch._C._cuda_getDevice()
        if t.device.index != cur:
            raise DmpError("tensor on %s but the current device is cuda:%d: launches go to the current device's stream "
                           "(use torch.cuda.set_device / `with torch.cuda.device(...)`)" % (t.device, cur))


def _first_cuda_index(objs):
    for o in objs:
        dev = getattr(o, "device", None)          # tensors and graph objects both carry one
        if dev is not None and getattr(dev, "type", None) == "cuda":
            return dev.index
    return None


def on_input_device(fn):
    """Decorator for module ``forward``s: run with the GPU of the first CUDA tensor / graph argument as the current
    device (the autograd engine does the same for the backward of what was recorded)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kw):
        import torch
        idx = _first_cuda_index(args) if args else None
        if idx is None or idx == torch._C._cuda_getDevice():
            return fn(self, *args, **kw)
        with torch.cuda.device(idx):
            return fn(self, *args, **kw)
    return wrapped


# ----------------------------------------------------------------------------- per-launch timing
class KernelTimer:
    """HIP-event timing of individual kernel launches on the launch stream (bench.py's
    ``roofline`` numbers).  Disabled by default: zero cost in the product path."""

    def __init__(self):
        self.enabled = False
        self.only = None   # if set: time only launches whose name starts with this prefix
        self.records = {}  # name -> list of (start_event, end_event, algorithmic_bytes)

    def reset(self):
        self.records = {}

    def summary(self):
        """name -> dict(launches, avg_us, bytes, gbps, interrupted, avg_us_all); synchronises the device.
        A launch whose event pair spans more than 10x the median of its name (and at least 1 ms more) was interrupted by something
        else on the box (seen: one 20 ms stall in 60 launches of a 27 us kernel): such launches are left out of ``avg_us`` and
        counted in ``interrupted``; ``avg_us_all`` is the plain mean over every launch."""
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            nbytes = sum(r[2] for r in recs) / len(recs)
            kept = uninterrupted(ms)
            avg = sum(kept) / len(kept)
            out[name] = {"launches": len(recs), "avg_us": avg * 1e3, "bytes": nbytes,
                         "gbps": nbytes / (avg * 1e-3) / 1e9 if avg > 0 else 0.0,
                         "interrupted": len(ms) - len(kept), "avg_us_all": sum(ms) / len(ms) * 1e3}
        return out


def uninterrupted(ms):
    """The launch times (ms) without the launches something else on the box stalled: more than 10x the median AND at least 1 ms
    above it (a kernel's own spread never looks like that)."""
    med = sorted(ms)[len(ms) // 2]
    return [t for t in ms if not (t > 10.0 * med and t > med + 1.0)] or list(ms)


timer = KernelTimer()


class timed:
    """``with timed(name_format, format_args, nbytes): <one kernel launch>`` -- the name is only formatted
    when the timer is on (the product path pays one attribute test per launch)."""
    __slots__ = ("fmt", "args", "nbytes", "name", "on", "a", "b")

    def __init__(self, fmt, args, nbytes):
        self.fmt, self.args, self.nbytes = fmt, args, nbytes

    def __enter__(self):
        self.on = False
        if timer.enabled:
            self.name = self.fmt % self.args
            self.on = timer.only is None or self.name.startswith(timer.only)
            if self.on:
                import torch
                self.a = torch.cuda.Event(enable_timing=True)
                self.b = torch.cuda.Event(enable_timing=True)
                self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            timer.records.setdefault(self.name, []).append((self.a, self.b, self.nbytes))
        return False
