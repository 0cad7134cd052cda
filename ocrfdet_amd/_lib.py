"""Loader for ``libocrf_hip.so`` — the C-ABI HIP library behind every op in this package.

There is no CPU or PyTorch fallback anywhere in ``ocrfdet_amd``: if the library cannot be loaded
the ops raise.  ``build()`` compiles it in-tree with hipcc for gfx950 (cross-compiles without a
GPU); the built ``.so`` is git-ignored but travels with the source tree.
"""
import contextlib
import ctypes
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# OCRF_HIP_SO: another build of the same library (A/B of kernel versions in diagnostics); default: the in-tree one
_SO = os.environ.get("OCRF_HIP_SO") or os.path.join(_CSRC, "libocrf_hip.so")
_LIB = None


class OcrfHipError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile every ``csrc/*.hip`` into ``csrc/libocrf_hip.so`` (``--offload-arch=gfx950``)."""
    args = ["make", "-C", _CSRC, "-j4", "libocrf_hip.so"]
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(args, stdout=out)
    return _SO


def lib():
    """The loaded library.  Raises ``OcrfHipError`` if it is missing — never falls back."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise OcrfHipError(
                f"{_SO} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ocrfdet_amd/csrc` (hipcc, gfx950). ocrfdet_amd has no fallback path.")
        try:
            _LIB = ctypes.CDLL(_SO)
        except OSError as e:  # pragma: no cover
            raise OcrfHipError(f"cannot load {_SO}: {e}") from e
        _declare(_LIB)
        _declare_step(_LIB)
    rec = getattr(_RECORDER, 'rec', None)
    if rec is not None:
        return _RecordingLib(_LIB, rec)
    return _LIB


def _declare(L):
    c_int, c_size_t, c_void_p, c_float = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float
    L.ocrf_version.restype = ctypes.c_char_p
    L.ocrf_version.argtypes = []
    L.bev_pool_v2.restype = None
    L.bev_pool_v2.argtypes = [c_int, c_int] + [c_void_p] * 8
    L.bev_pool_v2_grad.restype = None
    L.bev_pool_v2_grad.argtypes = [c_int, c_int] + [c_void_p] * 10
    c_long = ctypes.c_long
    L.ocrf_bev_pool_v2.restype = c_int
    L.ocrf_bev_pool_v2.argtypes = [c_int, c_int, c_int, c_long] + [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p]
    L.ocrf_bev_pool_v2_workspace_bytes.restype = c_size_t
    L.ocrf_bev_pool_v2_workspace_bytes.argtypes = [c_int, c_int, c_long]
    L.ocrf_bev_pool_v2_nchw.restype = c_int
    L.ocrf_bev_pool_v2_nchw.argtypes = ([c_int, c_int, c_int] + [c_void_p] * 8 + [c_int] * 5 +
                                        [c_void_p, c_size_t, c_void_p])
    L.ocrf_bev_pool_v2_nchw_dyn.restype = c_int
    L.ocrf_bev_pool_v2_nchw_dyn.argtypes = ([c_int, c_int, c_int] + [c_void_p] * 9 + [c_int] * 5 +
                                            [c_void_p, c_size_t, c_void_p])
    L.ocrf_bev_pool_v2_nchw_workspace_bytes.restype = c_size_t
    L.ocrf_bev_pool_v2_nchw_workspace_bytes.argtypes = [c_int] * 7
    L.ocrf_bev_pool_plan_bytes.restype = c_size_t
    L.ocrf_bev_pool_plan_bytes.argtypes = [c_int] * 6
    L.ocrf_bev_pool_plan_build.restype = c_int
    L.ocrf_bev_pool_plan_build.argtypes = [c_int] * 3 + [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_size_t, c_void_p]
    L.ocrf_bev_pool_planned_workspace_bytes.restype = c_size_t
    L.ocrf_bev_pool_planned_workspace_bytes.argtypes = [c_int, c_int]
    L.ocrf_bev_pool_v2_nchw_planned.restype = c_int
    L.ocrf_bev_pool_v2_nchw_planned.argtypes = ([c_int] * 2 + [c_void_p] * 6 + [c_int] * 5 +
                                                [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p])
    L.ocrf_bev_pool_mfma_panel_rows.restype = c_int
    L.ocrf_bev_pool_mfma_panel_rows.argtypes = []
    L.ocrf_diag_pool_mfma_stamps.restype = c_int
    L.ocrf_diag_pool_mfma_stamps.argtypes = [c_void_p]
    L.ocrf_bev_pool_mfma_tile_side.restype = c_int
    L.ocrf_bev_pool_mfma_tile_side.argtypes = []
    L.ocrf_bev_pool_mfma_slab_bytes.restype = c_size_t
    L.ocrf_bev_pool_mfma_slab_bytes.argtypes = [c_int, c_int]
    L.ocrf_bev_pool_v2_nchw_mfma.restype = c_int
    L.ocrf_bev_pool_mfma_max_unit_panels.restype = c_int
    L.ocrf_diag_pool_panel_stamps.restype = c_int
    L.ocrf_diag_pool_panel_stamps.argtypes = [c_void_p]
    L.ocrf_bev_pool_cell_weights.restype = c_int
    L.ocrf_bev_pool_cell_weights.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.ocrf_bev_pool_v2_nchw_panel.restype = c_int
    L.ocrf_bev_pool_v2_nchw_panel.argtypes = [c_int, c_int] + [c_void_p] * 10 + [c_int] * 5 + [c_void_p] * 2 + [c_size_t, c_void_p]
    L.ocrf_bev_pool_mfma_max_unit_panels.argtypes = []
    L.ocrf_bev_pool_v2_nchw_mfma.argtypes = [c_int, c_int] + [c_void_p] * 10 + [c_int] * 5 + [c_void_p] * 3
    L.ocrf_tune_set.restype = c_int
    L.ocrf_tune_set.argtypes = [c_int, c_int]
    L.ocrf_bev_pool_max_units.restype = c_int
    L.ocrf_bev_pool_max_units.argtypes = [c_int] * 6
    L.ocrf_diag_bev_pool_stamps.restype = c_int
    L.ocrf_diag_bev_pool_stamps.argtypes = [c_int] * 2 + [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 3
    L.ocrf_bev_pool_v2_check_intervals.restype = c_int
    L.ocrf_bev_pool_v2_check_intervals.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    L.ocrf_bev_pool_v2_grad.restype = c_int
    L.ocrf_bev_pool_v2_grad.argtypes = [c_int, c_int] + [c_void_p] * 10 + [c_void_p]
    L.ocrf_rasterize_forward.restype = c_int
    L.ocrf_rasterize_forward.argtypes = ([c_int] * 4 + [c_void_p] * 4 + [c_float] + [c_void_p] * 4 +
                                         [c_int] + [c_void_p] * 7 + [c_void_p, c_size_t, c_void_p])
    L.ocrf_rasterize_forward_sets.restype = c_int
    L.ocrf_rasterize_forward_sets.argtypes = ([c_int] * 5 + [c_void_p] * 4 + [c_float] + [c_void_p] * 4 +
                                              [c_int] + [c_void_p] * 7 + [c_void_p, c_size_t, c_void_p])
    L.ocrf_raster_plan_count_workspace_bytes.restype = c_size_t
    L.ocrf_raster_plan_count_workspace_bytes.argtypes = [c_int]
    L.ocrf_raster_plan_count.restype = c_int
    L.ocrf_raster_plan_count.argtypes = [c_int] * 4 + [c_void_p] * 2 + [c_float] + [c_void_p] * 3 + [c_size_t, c_void_p]
    L.ocrf_raster_plan_build_workspace_bytes.restype = c_size_t
    L.ocrf_raster_plan_build_workspace_bytes.argtypes = [c_int, c_int, c_long]
    L.ocrf_raster_plan_bytes.restype = c_size_t
    L.ocrf_raster_plan_bytes.argtypes = [c_int, c_int, c_long]
    L.ocrf_raster_plan_build.restype = c_int
    L.ocrf_raster_plan_build.argtypes = ([c_int] * 4 + [c_void_p] * 2 + [c_float, c_long] +
                                         [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p])
    L.ocrf_rasterize_planned_workspace_bytes.restype = c_size_t
    L.ocrf_rasterize_planned_workspace_bytes.argtypes = [c_long, c_int]
    L.ocrf_rasterize_planned.restype = c_int
    L.ocrf_rasterize_planned.argtypes = ([c_void_p, c_size_t, c_int, c_int, c_long] + [c_int] * 4 + [c_void_p] * 4 +
                                         [c_float] + [c_void_p] * 2 + [c_int] + [c_void_p] * 6 + [c_size_t, c_int] +
                                         [c_void_p] * 2 + [c_size_t, c_int, c_void_p, c_int, c_void_p, c_int] +
                                         [c_void_p, c_size_t, c_int, c_int, c_long, c_void_p, c_void_p])
    L.ocrf_rasterize_planned_bins_workspace_bytes.restype = c_size_t
    L.ocrf_rasterize_planned_bins_workspace_bytes.argtypes = [c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_long]
    L.ocrf_raster_plan_bins_bytes.restype = c_size_t
    L.ocrf_raster_plan_bins_bytes.argtypes = [c_int] * 5 + [c_long]
    L.ocrf_raster_plan_bins_workspace_bytes.restype = c_size_t
    L.ocrf_raster_plan_bins_workspace_bytes.argtypes = [c_int] * 6 + [c_long]
    L.ocrf_raster_plan_bins_build.restype = c_int
    L.ocrf_raster_plan_bins_build.argtypes = ([c_void_p, c_size_t, c_int, c_int, c_long, c_int, c_int, c_float, c_int, c_int,
                                               c_long, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p])
    L.ocrf_stream_write_value32.restype = c_int
    L.ocrf_stream_write_value32.argtypes = [c_void_p, c_int, c_void_p]
    L.ocrf_lss_prepare.restype = c_int
    L.ocrf_lss_prepare.argtypes = ([c_int] * 5 + [c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 6 +
                                   [c_void_p, c_size_t, c_void_p])
    L.ocrf_lss_prepare_workspace_bytes.restype = c_size_t
    L.ocrf_lss_prepare_workspace_bytes.argtypes = [c_int] * 8
    L.ocrf_ht_prepare.restype = c_int
    L.ocrf_ht_prepare.argtypes = ([c_int] * 7 + [c_void_p] * 3 + [c_float] * 4 + [c_void_p] * 6 +
                                  [c_void_p, c_size_t, c_void_p])
    L.ocrf_geometry_blocks.restype = c_int
    L.ocrf_geometry_blocks.argtypes = [c_int, c_int] + [c_void_p] * 7 + [c_int, c_int, c_float, c_float] + [c_void_p] * 4
    L.ocrf_ht_prepare_workspace_bytes.restype = c_size_t
    L.ocrf_ht_prepare_workspace_bytes.argtypes = [c_int, c_int]
    L.ocrf_rasterize_backward.restype = c_int
    L.ocrf_rasterize_backward.argtypes = ([c_int] * 4 + [c_void_p] * 4 + [c_float] + [c_void_p] * 13 +
                                          [c_void_p, c_size_t, c_void_p])
    L.ocrf_rasterize_backward_cov3d.restype = c_int
    L.ocrf_rasterize_backward_cov3d.argtypes = [c_int] * 4 + [c_void_p] * 16 + [c_size_t, c_void_p]
    L.ocrf_rasterize_backward_workspace_bytes.restype = c_size_t
    L.ocrf_rasterize_backward_workspace_bytes.argtypes = [c_int, c_int]
    L.ocrf_rasterize_workspace_bytes.restype = c_size_t
    L.ocrf_rasterize_workspace_bytes.argtypes = [c_int, c_int]
    L.ocrf_hoa_channel_stats.restype = c_int
    L.ocrf_hoa_channel_stats.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.ocrf_hoa_opacity_mask_gate.restype = c_int
    L.ocrf_hoa_opacity_mask_gate.argtypes = [c_void_p] * 4 + [c_int] * 5 + [c_void_p] * 3
    L.ocrf_hoa_height_attention.restype = c_int
    L.ocrf_hoa_height_attention.argtypes = ([c_void_p] + [c_int] * 5 + [c_void_p] * 4 +
                                            [c_void_p, c_size_t, c_void_p])
    L.ocrf_hoa_height_attention_workspace_bytes.restype = c_size_t
    L.ocrf_hoa_height_attention_workspace_bytes.argtypes = [c_int, c_int]
    L.ocrf_hoa_unet_block.restype = c_int
    L.ocrf_hoa_unet_block.argtypes = ([c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p])
    L.ocrf_hoa_v2b_weights_len.restype = c_int
    L.ocrf_hoa_v2b_weights_len.argtypes = []
    L.ocrf_hoa_v2b_workspace_bytes.restype = c_size_t
    L.ocrf_hoa_v2b_workspace_bytes.argtypes = [c_int] * 3
    L.ocrf_hoa_v2b_forward.restype = c_int
    L.ocrf_hoa_v2b_forward.argtypes = [c_void_p] * 3 + [c_int] * 3 + [c_void_p, c_size_t, c_void_p, c_void_p]
    L.ocrf_hoa_unet_tiles.restype = c_int
    L.ocrf_hoa_unet_tiles.argtypes = [c_int, c_int]
    L.ocrf_hoa_height_gate_from_tiles.restype = c_int
    L.ocrf_hoa_height_gate_from_tiles.argtypes = [c_int] * 4 + [c_void_p] * 5
    L.ocrf_hoa_gated_conv1x1.restype = c_int
    L.ocrf_hoa_gated_conv1x1.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p] * 4
    L.ocrf_hoa1_forward.restype = c_int
    L.ocrf_hoa1_forward.argtypes = [c_void_p] * 3 + [c_int] * 3 + [c_float] + [c_void_p] * 3
    L.ocrf_hoa1_weights_len.restype = c_int
    L.ocrf_hoa1_weights_len.argtypes = []
    L.ocrf_ht_project.restype = c_int
    L.ocrf_ht_project.argtypes = [c_int] * 4 + [c_void_p] * 3 + [c_float] * 4 + [c_void_p] * 4
    L.ocrf_prefilter.restype = c_int
    L.ocrf_prefilter.argtypes = [c_void_p] + [c_int] * 4 + [c_float] * 2 + [c_void_p] * 5
    L.ocrf_pillar_sample_mean.restype = c_int
    L.ocrf_pillar_sample_mean.argtypes = [c_void_p] * 4 + [c_int] * 6 + [c_void_p]
    L.ocrf_retain_valid_pixels.restype = c_int
    L.ocrf_retain_valid_pixels.argtypes = [c_void_p] * 5 + [c_int] * 6 + [c_void_p]
    L.ocrf_gauss_heads.restype = c_int
    L.ocrf_gauss_heads.argtypes = [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 5
    L.ocrf_gauss_heads_params_len.restype = c_int
    L.ocrf_gauss_heads_params_len.argtypes = [c_int, c_int]
    L.ocrf_sh_to_rgb.restype = c_int
    L.ocrf_sh_to_rgb.argtypes = [c_int] * 3 + [c_void_p] * 6
    L.ocrf_sh_to_rgb_backward.restype = c_int
    L.ocrf_sh_to_rgb_backward.argtypes = [c_int] * 3 + [c_void_p] * 8
    L.ocrf_gauss_heads_backward_workspace_bytes.restype = c_size_t
    L.ocrf_gauss_heads_backward_workspace_bytes.argtypes = [c_int] * 4
    L.ocrf_gauss_heads_backward.restype = c_int
    L.ocrf_gauss_heads_backward.argtypes = [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 7 + [c_size_t, c_void_p]
    L.ocrf_nerf_alpha.restype = c_int
    L.ocrf_nerf_alpha.argtypes = [c_void_p] * 4 + [c_int] * 3 + [c_void_p]
    L.ocrf_nerf_render.restype = c_int
    L.ocrf_nerf_render.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 3
    L.ocrf_nerf_render_params_len.restype = c_int
    L.ocrf_nerf_render_params_len.argtypes = []
    L.ocrf_dual_feat_fusion.restype = c_int
    L.ocrf_dual_feat_fusion.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_void_p]
    L.ocrf_dual_feat_fusion_plus.restype = c_int
    L.ocrf_dual_feat_fusion_plus.argtypes = [c_void_p] * 7 + [c_int] * 4 + [c_void_p]
    L.ocrf_plane_stats_pair.restype = c_int
    L.ocrf_plane_stats_pair.argtypes = [c_void_p] * 2 + [c_int] * 5 + [c_void_p] * 3
    L.ocrf_plane_bias_act_stats.restype = c_int
    L.ocrf_plane_bias_act_stats.argtypes = [c_void_p] * 2 + [c_int] * 8 + [c_void_p] * 3
    L.ocrf_channel_mlp.restype = c_int
    L.ocrf_channel_mlp.argtypes = [c_void_p] * 2 + [c_int] * 3 + [c_float] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 2
    L.ocrf_scaled_channel_stats.restype = c_int
    L.ocrf_scaled_channel_stats.argtypes = [c_void_p] * 2 + [c_int] * 3 + [c_void_p] * 2
    L.ocrf_cbam_tail.restype = c_int
    L.ocrf_cbam_tail.argtypes = [c_void_p] * 4 + [c_int] + [c_void_p] * 2 + [c_float] + [c_int] * 4 + [c_void_p] * 3
    L.ocrf_hoa_dw3x3.restype = c_int
    L.ocrf_hoa_dw3x3.argtypes = [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 2
    L.ocrf_hoa_dw3x3_wgrad_bands.restype = c_int
    L.ocrf_hoa_dw3x3_wgrad_bands.argtypes = [c_int]
    L.ocrf_hoa_dw3x3_wgrad.restype = c_int
    L.ocrf_hoa_dw3x3_wgrad.argtypes = [c_void_p] * 2 + [c_int] * 4 + [c_void_p] * 2
    L.ocrf_diag_stamp.restype = c_int
    L.ocrf_diag_stamp.argtypes = [c_void_p, c_void_p]
    L.ocrf_diag_plan_stats.restype = c_int
    L.ocrf_diag_plan_stats.argtypes = [c_void_p]
    L.ocrf_diag_plan_resident.restype = c_int
    L.ocrf_diag_plan_resident.argtypes = []
    L.ocrf_diag_where.restype = c_int
    L.ocrf_diag_where.argtypes = [c_int, c_void_p, c_int, c_void_p]
    L.ocrf_kernel_name.restype = ctypes.c_char_p
    L.ocrf_kernel_name.argtypes = [c_int]
    L.ocrf_graph_node_census.restype = c_int
    L.ocrf_graph_node_census.argtypes = [c_void_p] + [ctypes.POINTER(c_int)] * 4
    L.ocrf_stream_create.restype = c_int
    L.ocrf_stream_create.argtypes = [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]
    L.ocrf_stream_destroy.restype = c_int
    L.ocrf_stream_destroy.argtypes = [c_void_p]
    L.ocrf_timer_create.restype = c_int
    L.ocrf_timer_create.argtypes = [c_int, ctypes.POINTER(c_void_p)]
    L.ocrf_timer_arm.restype = c_int
    L.ocrf_timer_arm.argtypes = [c_void_p, c_int]
    L.ocrf_timer_read.restype = c_int
    L.ocrf_timer_read.argtypes = [c_void_p, ctypes.POINTER(c_float), c_int, ctypes.POINTER(c_int)]
    L.ocrf_timer_destroy.restype = c_int
    L.ocrf_timer_destroy.argtypes = [c_void_p]


def _declare_step(L):
    c_int, c_void_p = ctypes.c_int, ctypes.c_void_p
    L.ocrf_step_fn_id.restype = c_int
    L.ocrf_step_fn_id.argtypes = [ctypes.c_char_p]
    L.ocrf_step_fn_args.restype = c_int
    L.ocrf_step_fn_args.argtypes = [c_int]
    L.ocrf_step_create.restype = c_int
    L.ocrf_step_create.argtypes = [ctypes.POINTER(c_void_p)]
    L.ocrf_step_destroy.restype = None
    L.ocrf_step_destroy.argtypes = [c_void_p]
    L.ocrf_step_add_call.restype = c_int
    L.ocrf_step_add_call.argtypes = [c_void_p, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_uint64)]
    L.ocrf_step_add_fork.restype = c_int
    L.ocrf_step_add_fork.argtypes = [c_void_p, c_int, c_int]
    L.ocrf_step_add_join.restype = c_int
    L.ocrf_step_add_join.argtypes = [c_void_p, c_int, c_int]
    L.ocrf_step_size.restype = c_int
    L.ocrf_step_size.argtypes = [c_void_p]
    L.ocrf_step_run.restype = c_int
    L.ocrf_step_run.argtypes = [c_void_p, ctypes.POINTER(c_void_p), c_int]
    L.ocrf_hotpath_step.restype = c_int
    L.ocrf_hotpath_step.argtypes = [c_void_p, c_void_p, c_void_p]


# ---------------------------------------------------------------------------------------------------------------
# One host call per step (csrc/step.hip): the library calls of a step are recorded once, while the step runs eagerly,
# and replayed from C afterwards.
# ---------------------------------------------------------------------------------------------------------------
import threading  # noqa: E402

_RECORDER = threading.local()      # .rec: the StepRecorder of THIS thread (another thread calling lib()
                                   # while a step is recorded must not get its calls spliced into that step)
# entry points that enqueue nothing (sizes, lengths, knobs, timers): passed through while a step is being recorded
_QUERY_SUFFIXES = ('_bytes', '_len', '_rows', '_side', '_panels', '_tiles', '_bands', '_resident', '_version', '_name',
                   '_max_units')
_QUERY_PREFIXES = ('ocrf_timer_', 'ocrf_tune_', 'ocrf_step_', 'ocrf_diag_', 'ocrf_kernel_')


class StepNotRecordable(OcrfHipError):
    pass


class StepRecorder:
    """Records the library calls of ONE step while it is issued eagerly: ``with recorder: step()``.  ``streams``: the
    torch streams of the step in slot order (slot 0 = the caller's); a call on any other stream, or of an entry point
    ``ocrf_step_fn_id`` does not know, makes the step not recordable (``ok`` False — keep issuing it call by call).
    The owner marks the fork / join points itself (``fork`` / ``join``: it is the one that issues the torch events).
    ``build()`` -> a ``CompiledStep``.  Everything a recorded call points at has to stay alive and in place."""

    def __init__(self, streams):
        self.slots = {int(st.cuda_stream): i for i, st in enumerate(streams)}
        self.items = []
        self.names = []                  # every enqueueing entry point the step called, recordable or not
        self.ok = True
        self.why = None

    def __enter__(self):
        if getattr(_RECORDER, 'rec', None) is not None:
            raise OcrfHipError('a step is already being recorded')
        lib()
        _RECORDER.rec = self
        return self

    def __exit__(self, *exc):
        _RECORDER.rec = None
        return False

    def fail(self, why):
        if self.ok:
            self.ok, self.why = False, why

    def fork(self, frm, to):
        self.items.append(('fork', frm, to))

    def join(self, frm, to):
        self.items.append(('join', frm, to))

    def on_call(self, name, fn, args):
        L = _LIB
        self.names.append(name)
        fid = L.ocrf_step_fn_id(name.encode())
        if fid < 0:
            return self.fail(f'{name} cannot be held by a step object')
        types = fn.argtypes
        if types is None or len(types) != len(args) or L.ocrf_step_fn_args(fid) != len(args) - 1:
            return self.fail(f'{name}: argument list does not match its declaration')
        raw = args[-1]
        raw = raw.value if isinstance(raw, ctypes._SimpleCData) else raw
        slot = self.slots.get(int(raw or 0))
        if slot is None:
            return self.fail(f'{name} was issued on a stream the step does not know')
        words = []
        for a, t in zip(args[:-1], types[:-1]):
            v = a.value if isinstance(a, ctypes._SimpleCData) else a
            if t is ctypes.c_float:
                import struct
                words.append(struct.unpack('<I', struct.pack('<f', float(v)))[0])
            else:
                words.append((0 if v is None else int(v)) & 0xFFFFFFFFFFFFFFFF)
        self.items.append(('call', fid, slot, words))

    def build(self):
        if not self.ok:
            raise StepNotRecordable(self.why)
        return CompiledStep(self.items)


class _RecordingLib:
    """What ``lib()`` hands out while a step is recorded: every enqueueing entry point is logged, then called."""

    def __init__(self, L, rec):
        self._L, self._rec = L, rec

    def __getattr__(self, name):
        fn = getattr(self._L, name)
        if (not name.startswith('ocrf_') or name.endswith(_QUERY_SUFFIXES) or name.startswith(_QUERY_PREFIXES)):
            return fn
        rec = self._rec

        def logged(*args):
            rec.on_call(name, fn, args)
            return fn(*args)
        return logged


class CompiledStep:
    """The calls of one step inside the library (``ocrf_step``): ``run(main, side, ...)`` issues all of them with one
    host call."""

    def __init__(self, items):
        L = lib()
        self._h = ctypes.c_void_p()
        check(L.ocrf_step_create(ctypes.byref(self._h)), 'ocrf_step_create')
        self.n_calls = 0
        for it in items:
            if it[0] == 'call':
                _, fid, slot, words = it
                arr = (ctypes.c_uint64 * len(words))(*words)
                check(L.ocrf_step_add_call(self._h, fid, slot, len(words), arr), 'ocrf_step_add_call')
                self.n_calls += 1
            elif it[0] == 'fork':
                check(L.ocrf_step_add_fork(self._h, it[1], it[2]), 'ocrf_step_add_fork')
            else:
                check(L.ocrf_step_add_join(self._h, it[1], it[2]), 'ocrf_step_add_join')
        self._run2 = L.ocrf_hotpath_step
        self._runn = L.ocrf_step_run

    def run(self, *raw_streams):
        """``raw_streams``: hipStream_t values (ints) in slot order."""
        if len(raw_streams) == 2:
            err = self._run2(self._h, raw_streams[0], raw_streams[1])
        else:
            arr = (ctypes.c_void_p * len(raw_streams))(*raw_streams)
            err = self._runn(self._h, arr, len(raw_streams))
        if err != 0:
            raise OcrfHipError(f'ocrf_hotpath_step failed with hipError_t {err}')

    def __del__(self):
        try:
            if self._h:
                _LIB.ocrf_step_destroy(self._h)
                self._h = None
        except Exception:
            pass


def check(err, what):
    if err != 0:
        raise OcrfHipError(f"{what} failed with hipError_t {err}")


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on ``device`` as an integer for ctypes."""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else device
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_NO_SWITCH = contextlib.nullcontext()


def on_device(device):
    """``torch.cuda.device(device)`` when the current device differs, else a no-op: the context manager costs
    ~10 us per call, more than the launch it guards (the reference's OptionalCUDAGuard, bev_pool.cpp:42)."""
    idx = device.index if isinstance(device, torch.device) else device
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(idx)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise OcrfHipError(
                "ocrfdet_amd ops run on the GPU only (HIP kernels, no CPU fallback); got a "
                f"{t.device} tensor")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else 0)


def graph_census(cuda_graph):
    """{kernel, memset, memcpy, other} node counts of a ``torch.cuda.CUDAGraph`` built with ``keep_graph=True``."""
    k, ms, mc, o = (ctypes.c_int(0) for _ in range(4))
    check(lib().ocrf_graph_node_census(ctypes.c_void_p(int(cuda_graph.raw_cuda_graph())), ctypes.byref(k), ctypes.byref(ms),
                                       ctypes.byref(mc), ctypes.byref(o)), 'ocrf_graph_node_census')
    return dict(kernel=k.value, memset=ms.value, memcpy=mc.value, other=o.value)


def masked_stream(device, cu_bits=None, priority=0):
    """A torch stream object over a HIP stream limited to the compute units whose bits are set in ``cu_bits`` (an
    iterable of CU indices; None: every CU, with HIP ``priority``).  The HIP stream lives for the process."""
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        if cu_bits is None:
            check(lib().ocrf_stream_create(None, 0, int(priority), ctypes.byref(h)), 'ocrf_stream_create')
        else:
            bits = sorted(set(int(b) for b in cu_bits))
            n_words = (max(bits) // 32) + 1
            words = (ctypes.c_uint32 * n_words)()
            for b in bits:
                words[b // 32] |= 1 << (b % 32)
            check(lib().ocrf_stream_create(ctypes.cast(words, c_void_p_t), n_words, 0, ctypes.byref(h)), 'ocrf_stream_create')
    return torch.cuda.ExternalStream(h.value, device=device)


c_void_p_t = ctypes.c_void_p


def diag_stamp(stamps, index):
    """Diagnostic: store the device clock (100 MHz) into ``stamps[index]`` (int64, cuda) when torch's current
    stream reaches this point — works inside a captured graph (``ocrf_diag_stamp``)."""
    require_cuda(stamps)
    check(lib().ocrf_diag_stamp(ctypes.c_void_p(stamps.data_ptr() + 8 * index), stream_ptr(stamps.device)),
          'ocrf_diag_stamp')


class Workspace:
    """Per-device grow-only scratch buffers (torch uint8 tensors, one per tag), so that steady-state calls
    allocate nothing.  Reuse is stream-ordered: a tag is meant for one stream at a time (the side streams of
    ``HotPath`` / ``_core_fused`` use tags of their own).

    Two hazards of replacing a buffer when a larger request arrives are handled here:
    * a captured hipGraph has the raw pointers of the buffers it used baked in — ``hold(device)`` returns
      strong references to every buffer of the device; whoever owns a graph keeps that object for the
      graph's lifetime (``GraphedNeck``), so a later, larger eager request gets a NEW buffer while the
      graph keeps replaying on its own, still allocated one;
    * the caching allocator may hand a freed block to the next allocation on the allocating stream while
      another stream that used the buffer is still running — every stream that asked for the tag is
      recorded on the old buffer (``Tensor.record_stream``) before it is dropped."""

    def __init__(self):
        self._bufs = {}
        self._users = {}                       # key -> raw hipStream_t values that asked for the tag
        self._generation = {}                  # device index -> number of buffers REPLACED so far

    def get(self, device, nbytes, tag="default"):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, tag)
        buf = self._bufs.get(key)
        if _raw_stream is not None:
            users = self._users.get(key)
            if users is None:
                users = self._users[key] = set()
            users.add(_raw_stream(idx))
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                # whoever baked the old buffer's address into a recorded step (HotPath._compiled, ShardedHotPath._segments)
                # sees the generation move and records anew instead of replaying into freed memory
                self._generation[idx] = self._generation.get(idx, 0) + 1
                for raw in self._users.get(key, ()):
                    if raw:                    # 0 = the legacy default stream, the allocator's own
                        buf.record_stream(torch.cuda.ExternalStream(raw, device=buf.device))
                self._users[key] = {_raw_stream(idx)} if _raw_stream is not None else set()
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf

    def generation(self, device):
        """Number of scratch buffers of ``device`` replaced so far: a recorded step (raw pointers baked in) is valid
        while this has not moved since it was recorded."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return self._generation.get(idx, 0)

    def hold(self, device):
        """Strong references to every scratch buffer of ``device`` as of now (see the class docstring)."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return [b for (d, _), b in self._bufs.items() if d == idx]


workspace = Workspace()


K_BEV_POOL_FWD, K_BEV_POOL_FIXUP, K_BEV_POOL_INTERVAL, K_BEV_POOL_GRAD, K_BEV_POOL_NCHW = 1, 2, 3, 4, 5
K_BEV_POOL_MFMA, K_BEV_POOL_PANEL, K_BEV_POOL_CELL_WEIGHTS = 6, 7, 8
K_RASTER_PREPROCESS, K_RASTER_BLEND, K_RASTER_GATHER = 10, 11, 12
K_RASTER_SCAN, K_RASTER_BLEND_BWD, K_RASTER_PRE_BWD = 13, 15, 16
K_RASTER_PLAN_UPDATE, K_RASTER_BLEND_SORTED, K_RASTER_BLEND_SECOND = 17, 18, 19
K_LSS_KEYS, K_RADIX_HIST, K_SCAN, K_RADIX_SCATTER, K_LSS_BOUNDS, K_LSS_EMIT, K_HT_COUNT, K_HT_EMIT = range(40, 48)
K_HOA_STATS, K_HOA_MASK_GATE, K_HOA_HEIGHT_MAX, K_HOA_HEIGHT_GATE = 20, 21, 22, 23
K_HOA_UNET_BLOCK, K_HOA_OUT_CONV, K_HOA1_ATTN, K_HOA1_KV = 24, 25, 26, 29


class KernelTimer:
    """Device duration of every launch of ONE kernel of the library while armed (HIP events on
    the launch stream, ``ocrf_timer_*`` in include/ocrf_hip.h).  Used by bench.py's roofline leg."""

    def __init__(self, kernel_id, capacity):
        self.kernel_id, self.capacity = kernel_id, int(capacity)
        self._h = ctypes.c_void_p()
        check(lib().ocrf_timer_create(self.capacity, ctypes.byref(self._h)), 'ocrf_timer_create')

    @property
    def kernel_name(self):
        return lib().ocrf_kernel_name(self.kernel_id).decode()

    def arm(self):
        """Several timers (different kernel ids) may be armed at once."""
        check(lib().ocrf_timer_arm(self._h, self.kernel_id), 'ocrf_timer_arm')

    def disarm(self):
        check(lib().ocrf_timer_arm(None, 0), 'ocrf_timer_arm')

    @staticmethod
    def disarm_all():
        check(lib().ocrf_timer_arm(None, 0), 'ocrf_timer_arm')

    def mean_ms(self):
        v = self.read_ms()
        self._count = len(v)
        return sum(v) / len(v) if v else None

    def count(self):
        return getattr(self, '_count', 0)

    def read_ms(self):
        buf = (ctypes.c_float * self.capacity)()
        n = ctypes.c_int(0)
        check(lib().ocrf_timer_read(self._h, buf, self.capacity, ctypes.byref(n)), 'ocrf_timer_read')
        return [float(buf[i]) for i in range(n.value)]

    def close(self):
        if self._h:
            lib().ocrf_timer_destroy(self._h)
            self._h = ctypes.c_void_p()
