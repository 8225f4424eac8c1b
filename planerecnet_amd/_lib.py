"""ctypes binding of libprn_hip.so (C ABI declared in include/prn.h).

There is deliberately NO fallback: if the library is missing or a symbol is absent, importing this
module raises, and every op in planerecnet_amd.ops raises on non-device tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRN_LIB") or os.path.join(_HERE, "libprn_hip.so")   # PRN_LIB: A/B builds in tuning runs

c_int, c_float, c_void_p, c_i64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64


class GemmOpts(ctypes.Structure):
    """mirror of prn_gemm_opts: per-call execution options (which matrix pipe, piece format, thresholds, weight-gradient launch size).
    The library keeps no such state; this value travels inside every descriptor / as an argument."""
    _fields_ = [("split_mode", ctypes.c_int32), ("split_kind", ctypes.c_int32), ("split_products", ctypes.c_int32), ("split_min_tiles", ctypes.c_int32),
                ("split_min_gflop", ctypes.c_float), ("wgrad_wgs", ctypes.c_int32), ("wgrad_target", ctypes.c_int32), ("wgrad_split", ctypes.c_int32)]

    def key(self):
        return (self.split_mode, self.split_kind, self.split_products, self.split_min_tiles, self.split_min_gflop, self.wgrad_wgs, self.wgrad_target, self.wgrad_split)


SPLIT_OFF, SPLIT_PLAN, SPLIT_ALWAYS = 0, 1, 2
PIECES_BF16, PIECES_F16 = 0, 16


class ConvDesc(ctypes.Structure):
    """mirror of prn_conv_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "C", "H", "W", "M", "KH", "KW", "stride", "pad", "Ho", "Wo", "in_mode", "dil", "epilogue", "ystride", "yH", "yW", "reserved")] + [
                    ("opts", GemmOpts)]


class Ragged(ctypes.Structure):
    """mirror of prn_ragged"""
    _fields_ = [("nseg", ctypes.c_int32), ("H", ctypes.c_int32 * 6), ("W", ctypes.c_int32 * 6)]


class DcnDesc(ctypes.Structure):
    """mirror of prn_dcn_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "C", "H", "W", "M", "stride", "pad", "Ho", "Wo", "raw")] + [("max_offset", ctypes.c_float),
                                                                                                              ("epilogue", ctypes.c_int32),
                                                                                                              ("opts", GemmOpts)]


class BottleneckDesc(ctypes.Structure):
    """mirror of prn_bottleneck_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "C", "H", "W", "planes", "stride", "dcn", "downsample", "flags")] + [
        ("eps", ctypes.c_float * 4), ("momentum", ctypes.c_float * 4), ("max_offset", ctypes.c_float), ("reserved", ctypes.c_int32), ("opts", GemmOpts)]


class BottleneckParams(ctypes.Structure):
    """mirror of prn_bottleneck_params: device addresses (or None) of a block's parameters and derived operands"""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "w1", "w2", "w3", "wd", "b2", "w1_img", "w3_img", "wd_img", "u2", "u2_img", "w1_t", "w2_t", "w3_t", "wd_t", "w1_t_img", "w3_t_img", "wd_t_img", "ut2", "ut2_img",
        "w27", "b27", "w27_t", "w2_cols_t", "w2_cols_t_img")] + [(n, ctypes.c_void_p * 4) for n in ("gamma", "beta", "running_mean", "running_var")]


IN_ZERO, IN_REFLECT, IN_UP2_REFLECT, IN_DILATED, IN_UP2_PHASE, IN_EMBED1 = 0, 1, 2, 3, 4, 5
EPI_NONE, EPI_RELU, EPI_SIGMOID = 0, 1, 2
BN_SPLITS = 32

P = c_void_p
_DP = ctypes.POINTER(ConvDesc)
_OP = ctypes.POINTER(GemmOpts)
SIGNATURES = {
    "prn_version": (c_int, []),
    "prn_last_error": (ctypes.c_char_p, []),
    "prn_conv2d_fwd_ws_bytes": (c_i64, [_DP]),
    "prn_conv2d_kernel_kind": (c_int, [_DP]),
    "prn_conv2d_fwd_partials": (c_int, [_DP, ctypes.POINTER(c_i64)]),
    "prn_gemm_opts_default": (None, [_OP]),
    "prn_gemm_pipe": (c_int, [c_int, c_int, c_int, c_int, c_int, _OP]),
    "prn_split_images_bytes": (c_i64, [c_int, c_int, c_int]),
    "prn_split_prepare": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "prn_split_prepare_batched": (c_int, [P, c_int, c_i64, c_i64, c_int, P]),
    "prn_conv2d_fwd": (c_int, [_DP, P, P, P, P, P, P, P]),
    "prn_conv2d_fwd_ragged": (c_int, [_DP, P, P, P, P, P, P, P]),
    "prn_conv2d_wgrad_ragged_ws_bytes": (c_i64, [_DP, P]),
    "prn_conv2d_wgrad_ragged": (c_int, [_DP, P, P, P, P, P, P]),
    "prn_gn_relu_fwd_ragged": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P, c_int, c_float, P]),
    "prn_gn_relu_bwd_ragged": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, P, c_int, P]),
    "prn_conv2d_fwd_phase": (c_int, [_DP, P, P, P, P, P, P, P, c_int]),
    "prn_conv2d_fwd_counted": (c_int, [_DP, P, P, P, P, P, P, P, P, P, c_int]),
    "prn_conv2d_wgrad_phase": (c_int, [_DP, P, P, P, P, P, c_int]),
    "prn_weight_flip_transpose": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "prn_weight_flip_transpose_batched": (c_int, [P, c_int, c_i64, P]),
    "prn_winograd_tiles": (c_i64, [c_int, c_int, c_int]),
    "prn_winograd_weights_batched": (c_int, [P, c_int, c_i64, P]),
    "prn_winograd_input": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "prn_gemm_batched_ws_bytes": (c_i64, [c_int, c_int, c_int, c_int, _OP]),
    "prn_gemm_batched": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, _OP, P]),
    "prn_winograd_output": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "prn_winograd_output_bn_fwd": (c_int, [P] * 8 + [c_int] * 4 + [c_float, c_float, c_int, P]),
    "prn_winograd_output_bn_bwd": (c_int, [P] * 8 + [c_int] * 5 + [P]),
    "prn_winograd_wgrad_ws_bytes": (c_i64, [c_int] * 5 + [_OP]),
    "prn_winograd_dy": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "prn_gemm_batched_nt_splits": (c_int, [c_int] * 4 + [_OP]),
    "prn_gemm_batched_nt_kind": (c_int, [c_int] * 4 + [_OP]),
    "prn_conv2d_wgrad_kernel_kind": (c_int, [_DP, c_int]),
    "prn_gemm_batched_nt": (c_int, [c_int, c_int, c_int, c_int, P, P, P, _OP, P]),
    "prn_winograd_dw": (c_int, [P, P, c_int, c_int, c_int, P]),
    "prn_conv3x3_winograd_wgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, _OP, P, c_int]),
    "prn_winograd_tiles_ragged": (c_i64, [P, c_int]),
    "prn_conv3x3_winograd_ragged_ws_bytes": (c_i64, [P, c_int, c_int, c_int, _OP]),
    "prn_conv3x3_winograd_ragged": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, _OP, P]),
    "prn_winograd_wgrad_ragged_ws_bytes": (c_i64, [P, c_int, c_int, c_int, _OP]),
    "prn_conv3x3_winograd_wgrad_ragged": (c_int, [P, P, P, P, P, c_int, c_int, c_int, _OP, P]),
    "prn_conv3x3_winograd_wgrad_v": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, _OP, P]),
    "prn_conv3x3_winograd_ws_bytes": (c_i64, [c_int] * 5 + [_OP]),
    "prn_conv3x3_winograd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _OP, P]),
    "prn_conv2d_wgrad_ws_bytes": (c_i64, [_DP]),
    "prn_conv2d_wgrad_grouped_ws_bytes": (c_i64, [_DP, c_int]),
    "prn_conv2d_wgrad_grouped": (c_int, [_DP, c_int, P, P, P, P, P]),
    "prn_conv2d_wgrad": (c_int, [_DP, P, P, P, P, P]),
    "prn_pad_fold": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "prn_pad_fold_pitched": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "prn_up2_phase_weights": (c_int, [P, P, c_int, c_int, P]),
    "prn_up2_dgrad_weights": (c_int, [P, P, c_int, c_int, P]),
    "prn_up2_wgrad_combine": (c_int, [P, P, c_int, c_int, P]),
    "prn_space_to_depth2": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "prn_replicate_fold": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "prn_channel_sum": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "prn_dcn_sample": (c_int, [P, P, P] + [c_int] * 7 + [c_float, P]),
    "prn_dcn_sample_bwd_ws_bytes": (c_i64, [c_int] * 6),
    "prn_dcn_sample_bwd": (c_int, [P, P, P, P, P, P] + [c_int] * 7 + [c_float, P]),
    "prn_dcnv2_table_bytes": (c_i64, [P]),
    "prn_dcnv2_table": (c_int, [P, P, P, P, P]),
    "prn_dcnv2_fwd_ws_bytes": (c_i64, [P]),
    "prn_dcnv2_fwd": (c_int, [P] * 8),
    "prn_dcnv2_fwd_phase": (c_int, [P] * 8 + [c_int]),
    "prn_dcnv2_bwd_weight_ws_bytes": (c_i64, [P]),
    "prn_dcnv2_bwd_weight": (c_int, [P] * 7),
    "prn_dcnv2_bwd_weight_phase": (c_int, [P] * 7 + [c_int]),
    "prn_dcnv2_bwd_ws_bytes": (c_i64, [P]),
    "prn_dcnv2_bwd_input": (c_int, [P] * 9),
    "prn_dcnv2_bwd_input_phase": (c_int, [P] * 9 + [c_int]),
    "prn_bn_kernel_kind": (c_int, [c_int, c_int]),
    "prn_dcnv2_bwd_offset_mask": (c_int, [P] * 8),
    "prn_plane_prior_ws_bytes": (c_i64, [c_int] * 6 + [_OP]),
    "prn_plane_prior_fwd": (c_int, [P] * 7 + [c_int] * 6 + [_OP, P]),
    "prn_plane_prior_fwd_phase": (c_int, [P] * 7 + [c_int] * 6 + [_OP, P, c_int]),
    "prn_plane_prior_wgrad_ws_bytes": (c_i64, [c_int] * 5 + [_OP]),
    "prn_plane_prior_wgrad": (c_int, [P] * 4 + [c_int] * 5 + [_OP, P]),
    "prn_fpn_level_ws_bytes": (c_i64, [c_int] * 8 + [_OP]),
    "prn_fpn_level_fwd": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P, P] + [c_int] * 6 + [_OP, P]),
    "prn_frame_to_input": (c_int, [P] + [c_int] * 6 + [P, P, c_int, P, P, P]),
    "prn_mask_loss_ws_floats": (c_int, [c_int]),
    "prn_mask_loss_fwd": (c_int, [P] * 9 + [c_int, c_int, c_int, c_float, c_float, P]),
    "prn_mask_loss_bwd": (c_int, [P] * 8 + [c_int, c_int, P]),
    "prn_vnl_triplets": (c_int, [P] * 12 + [c_int, c_int, c_int, c_float, P]),
    "prn_vnl_scatter": (c_int, [P] * 5 + [c_int, c_int, P]),
    "prn_depth_metrics_ws_doubles": (c_int, []),
    "prn_depth_metrics": (c_int, [P, P, P, P, c_i64, c_float, c_float, P]),
    "prn_vnl_trim_key": (c_int, [P, P, P, P, c_int, P]),
    "prn_vnl_trim_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P]),
    "prn_vnl_trim_ws_bytes": (c_i64, [c_int]),
    "prn_vnl_trim_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P, P]),
    "prn_loss_ws_doubles": (c_int, [c_int]),
    "prn_focal_sum_fwd": (c_int, [P, P, P, P, c_i64, c_int, c_float, c_float, P]),
    "prn_focal_sum_bwd": (c_int, [P, P, P, P, c_i64, c_int, c_float, c_float, P]),
    "prn_rmse_log_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_float, c_float, P]),
    "prn_rmse_log_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_float, c_float, P]),
    "prn_gt_segments": (c_i64, [c_int, c_int]),
    "prn_gt_mask_stats": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P]),
    "prn_gt_quarter_masks": (c_int, [P, P, c_int, c_int, c_int, P]),
    "prn_gt_sample_triplets": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, ctypes.c_uint64, ctypes.c_uint64, c_i64, P, P]),
    "prn_adam_chunk_elems": (c_int, []),
    "prn_adam_step": (c_int, [P, c_int, c_int, P, P, P, P, P, P, P, P, P, ctypes.c_double, ctypes.c_double, c_float, P]),
    "prn_adam_step_masked": (c_int, [P, c_int, c_int, P, P, P, P, P, P, P, P, P, ctypes.c_double, ctypes.c_double, c_float, P, P, P]),
    "prn_pairwise_iou_ws_bytes": (c_i64, [c_int, c_int, c_i64]),
    "prn_pairwise_iou": (c_int, [P, P, P, P, c_int, c_int, c_i64, P, P, P, P]),
    "prn_mask_boxes": (c_int, [P, c_int, c_int, c_int, P, P]),
    "prn_mask_stats": (c_int, [P, c_int, c_i64, c_float, P, P, P]),
    "prn_sigmoid_point_nms": (c_int, [P, P, c_int, c_int, c_int, c_i64, P]),
    "prn_matrix_nms": (c_int, [P, P, P, c_int, c_float, c_int, P, P, P]),
    "prn_bn_stats": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P]),
    "prn_bn_apply": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "prn_bn_train_fwd": (c_int, [P] * 9 + [c_int, c_int, c_int, c_float, c_float, c_int, P]),
    "prn_bn_bwd": (c_int, [P] * 11 + [c_int] * 5 + [P]),
    "prn_bn_train_fwd_into": (c_int, [P] * 6 + [c_i64] + [P] * 3 + [c_int, c_int, c_int, c_float, c_float, c_int, P]),
    "prn_bn_bwd_from": (c_int, [P, c_i64] + [P] * 10 + [c_int] * 5 + [P]),
    "prn_bn_train_fwd_partials": (c_int, [P, c_int, c_i64] + [P] * 8 + [c_int, c_int, c_int, c_float, c_float, c_int, P]),
    "prn_bn_bwd_partials": (c_int, [P, c_int, c_i64] + [P] * 9 + [c_int] * 5 + [P]),
    "prn_bn_train_fwd_winograd": (c_int, [P, c_int, c_i64] + [P] * 9 + [c_int] * 4 + [c_float, c_float, c_int, P]),
    "prn_bn_bwd_winograd": (c_int, [P, c_int, c_i64] + [P] * 10 + [c_int] * 6 + [P]),
    "prn_gn_relu_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "prn_gn_relu_bwd": (c_int, [P] * 8 + [c_int] * 4 + [P]),
    "prn_resize_bilinear_fwd": (c_int, [P, P] + [c_int] * 5 + [P]),
    "prn_resize_bilinear_bwd": (c_int, [P, P] + [c_int] * 5 + [P]),
    "prn_resize_bilinear_add_fwd": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "prn_resize_bilinear_bwd_add": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "prn_maxpool3s2_fwd": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "prn_maxpool3s2_bwd": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "prn_sum_rows": (c_int, [P, P, c_int, c_int, c_int, P]),
    "prn_lava_gt_weights": (c_int, [P, P, c_int, c_int, c_int, c_float, P]),
    "prn_debug_skip_launches": (c_int, [c_int]),
    "prn_bottleneck_plan_bytes": (c_i64, []),
    "prn_bottleneck_params_bytes": (c_i64, []),
    "prn_bottleneck_plan": (c_int, [ctypes.POINTER(BottleneckDesc), P]),
    "prn_bottleneck_plan_info": (c_int, [P, ctypes.POINTER(c_i64), c_int]),
    "prn_bottleneck_train_fwd": (c_int, [P, ctypes.POINTER(BottleneckParams), P, P, P, P, P]),
    "prn_bottleneck_train_bwd": (c_int, [P, ctypes.POINTER(BottleneckParams), P, P, P, P, c_int, P, P, P, P, P]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
if lib.prn_bottleneck_params_bytes() != ctypes.sizeof(BottleneckParams):
    raise ImportError("libprn_hip.so and planerecnet_amd._lib disagree about prn_bottleneck_params (rebuild the library)")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib.prn_last_error().decode()}")
