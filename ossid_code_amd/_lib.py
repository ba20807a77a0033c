"""ctypes binding of libossid_hip.so (include/ossid_hip.h). There is no CPU fallback: if the library is
missing, or a call returns a non-zero status, the caller gets an exception."""
import ctypes as C
import os

import torch

from ._build import LIB_PATH

_lib = None

OSSID_OK = 0

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class PN2Weights(C.Structure):
    """struct ossid_pn2_weights (include/ossid_hip.h)."""
    _fields_ = [("blob", _vp), ("w_off", C.c_int64 * 12), ("b_off", C.c_int64 * 12), ("wxyz2_off", C.c_int64),
                ("npoint1", C.c_int32), ("npoint2", C.c_int32), ("radius1", _f), ("radius2", _f)]


class ConvDesc(C.Structure):
    """struct ossid_conv_desc (include/ossid_hip.h)."""
    _fields_ = [(n, _vp) for n in ("x", "wpk", "bias", "pre_scale", "pre_shift", "post_scale", "post_shift", "out")] + \
               [(n, C.c_int32) for n in ("batch", "height", "width", "cin", "cout", "taps", "act", "pre_relu",
                                         "src_height", "src_width", "in_channel_stride", "out_channel_stride",
                                         "out_channel_offset", "pre_batch_stride")] + [("in_batch_stride", C.c_int64)] + \
               [("scratch", _vp), ("scratch_bytes", C.c_int64), ("exact", C.c_int32)]


class WgradDesc(C.Structure):
    """struct ossid_wgrad_desc (include/ossid_hip.h)."""
    _fields_ = [(n, _vp) for n in ("x", "dy", "pre_scale", "pre_shift", "dw", "workspace")] + [("workspace_bytes", _sz)] + \
               [(n, C.c_int32) for n in ("batch", "height", "width", "cin", "cout", "taps", "pre_relu", "accumulate",
                                         "in_channel_stride", "dy_channel_stride", "src_height", "src_width")] + \
               [(n, _vp) for n in ("dy_add", "dy_add_scale", "dy_add_shift")]


class ChanOpDesc(C.Structure):
    """struct ossid_chan_op_desc (include/ossid_hip.h)."""
    _fields_ = [(n, _vp) for n in ("g", "x", "out", "alpha", "beta", "kappa", "mask_scale", "mask_shift", "partials",
                                   "sums", "pivot")] + [("n_rows", C.c_int64)] + \
               [(n, C.c_int32) for n in ("channels", "g_stride", "x_stride", "out_stride", "mask_mode", "accumulate",
                                         "sum_mode", "sums_row_stride", "defer_finalize")]


class PackRow(C.Structure):
    """struct ossid_pack_row (include/ossid_hip.h)."""
    _fields_ = [("w", _vp), ("wpk", _vp), ("first_block", C.c_int64), ("cout", C.c_int32), ("cin", C.c_int32),
                ("taps", C.c_int32), ("kind", C.c_int32)]


SEQ_MAX_INT, SEQ_MAX_FP = 24, 8


class SeqOp(C.Structure):
    """struct ossid_seq_op (include/ossid_hip.h)."""
    _fields_ = [("fn", _vp), ("event", _vp), ("slot", C.c_int32), ("wait_for", C.c_int32), ("n_int", C.c_int32),
                ("n_fp", C.c_int32), ("iarg", C.c_uint64 * SEQ_MAX_INT), ("fparg", C.c_uint64 * SEQ_MAX_FP)]


_PROTOS = {
    "ossid_abi_version": (_i, [C.c_char_p, _i]),
    "ossid_conv_wino_split_bf16": (_i, []),
    "ossid_conv_split_bf16": (_i, []),
    "ossid_conv_wgrad_split_bf16": (_i, []),
    "ossid_seg_tail_split_bf16": (_i, []),
    "ossid_zephyr_prep_frame_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ossid_zephyr_prep_frame_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ossid_zephyr_prep_model": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "ossid_zephyr_project_uv": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp]),
    "ossid_zephyr_inconst_count": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp]),
    "ossid_zephyr_featurize": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _f, _f, _f, _f, _i, _vp, _vp, _vp]),
    "ossid_pose_errors": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ossid_dtoid_prep_sample": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "ossid_mask_bbox_heatmap": (_i, [_vp, _i, _i, _i, _i, C.c_double, C.c_double, _vp, _vp, _vp]),
    "ossid_render_depth_points": (_i, [_vp, _vp, _i, _f, _f, _f, _f, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_visib_mask_iou": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "ossid_pn2_fps": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_pn2_ball_query": (_i, [_vp, _i, _i, _i, _vp, _i, _f, _i, _vp, _vp]),
    "ossid_pn2_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ossid_pn2_score": (_i, [_vp, _i, _i, C.POINTER(PN2Weights), _vp, _sz, _vp] + [_vp] * 7 + [_vp, _vp]),
    "ossid_pn2_stage_names": (C.c_char_p, []),
    "ossid_event_create": (_i, [C.POINTER(_vp)]),
    "ossid_event_destroy": (_i, [_vp]),
    "ossid_event_record": (_i, [_vp, _vp]),
    "ossid_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(_f)]),
    "ossid_pn2_kernel_names": (C.c_char_p, []),
    "ossid_dw_xcorr_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "ossid_dw_xcorr_nhwc_bcast": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ossid_dw_xcorr_bwd_x": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ossid_dw_xcorr_bwd_k": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ossid_conv_packed_floats": (_sz, [_i, _i, _i]),
    "ossid_conv_pack_weights": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ossid_conv_pack_weights_form": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ossid_conv_packed_floats_form": (C.c_size_t, [_i, _i, _i, _i]),
    "ossid_conv_nhwc_fwd": (_i, [_vp, _vp]),
    "ossid_seg_tail_packed_floats": (_sz, []),
    "ossid_seg_tail_pack_weights": (_i, [_vp, _vp, _vp]),
    "ossid_seg_tail_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_conv3x3_wgrad_splits": (_i, [_i, _i, _i, _i, _i]),
    "ossid_conv3x3_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ossid_conv3x3_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i, _vp]),
    "ossid_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ossid_conv_wgrad": (_i, [_vp, _vp]),
    "ossid_conv_wgrad_group_workspace_bytes": (_sz, [_vp, _i]),
    "ossid_conv_wgrad_group": (_i, [_vp, _i, _vp, _sz, _vp]),
    "ossid_conv_pack_weights_dgrad": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ossid_conv_wino_packed_floats": (C.c_size_t, [_i, _i]),
    "ossid_conv_pack_weights_wino": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ossid_conv3x3_wino_fwd": (_i, [_vp, _vp]),
    "ossid_conv3x3_wino_workspace_bytes": (_sz, [_vp]),
    "ossid_conv3x3_wino_pair_workspace_bytes": (_sz, [_vp, _vp]),
    "ossid_conv3x3_wino_fwd_pair": (_i, [_vp, _vp, _vp]),
    "ossid_fill_zero": (_i, [_vp, _sz, _vp]),
    "ossid_seg_bce_iou_workspace_bytes": (_sz, [_i]),
    "ossid_seg_bce_iou_fwd": (_i, [_vp, _vp, _i, C.c_longlong, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ossid_conv3x3_c1_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ossid_conv3x3_c1_dgrad": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_conv3x3_c1_wgrad_workspace_bytes": (_sz, []),
    "ossid_conv3x3_c1_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ossid_stem_weight_relayout": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ossid_chan_op_partials": (_i, [C.c_longlong, _i]),
    "ossid_chan_op": (_i, [_vp, _vp]),
    "ossid_bn_fold_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, C.c_double, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_bn_fold_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, C.c_double, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ossid_conv_pack_weights_table": (_i, [_vp, _i, C.c_longlong, _vp]),
    "ossid_colsum_finalize": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "ossid_avgpool2_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "ossid_upsample_nearest_bwd_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ossid_focal_smoothl1_loss_workspace_floats": (_sz, [_i, _i]),
    "ossid_focal_smoothl1_loss_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_focal_smoothl1_loss_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_im2col_stem": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ossid_conv1x1_c1_fwd": (_i, [_vp, C.c_longlong, _i, _vp, _vp, _i, _vp, _vp]),
    "ossid_conv1x1_c1_bwd_workspace_floats": (_sz, [C.c_longlong, _i]),
    "ossid_conv1x1_c1_bwd": (_i, [_vp, _vp, C.c_longlong, _i, _vp, _vp, _vp, _vp, _vp]),
    "ossid_spatial_mean": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ossid_small_matmul": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ossid_detect_post_workspace_bytes": (_sz, [_i, _i]),
    "ossid_detect_post": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _f, _f, _f, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_detect_emit": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, C.c_longlong, _vp, C.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_dense_fwd1_stats_partials": (_i, [C.c_longlong]),
    "ossid_dense_fwd1_stats": (_i, [_vp, _i, _i, _vp, _vp, _vp, C.c_longlong, _vp, _vp, _vp, _vp]),
    "ossid_bn_fold_fwd_tail": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, C.c_double, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_bn_fold_fwd_rows": (_i, [_vp, _vp, _i, _i, C.c_double, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_dense_dgrad3_mask_partials": (_i, [_i, _i, _i]),
    "ossid_dense_dgrad3_mask": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ossid_dense_dgrad1_acc_partials": (_i, [C.c_longlong]),
    "ossid_dense_dgrad1_acc": (_i, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ossid_dense_fused_available": (_i, []),
    "ossid_dense_table_bytes": (_sz, [_i]),
    "ossid_dense_entry": (_i, [_vp, _i, _i, C.c_longlong, _i, _vp, _vp, _vp]),
    "ossid_dense_layer": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ossid_stem_conv_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ossid_stem_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "ossid_stem_conv_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _i, _vp]),
    "ossid_stem_tail_nhwc": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ossid_stem_tail_pool_nhwc": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "ossid_bn_relu_avgpool2_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "ossid_maxpool_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ossid_dw_add_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ossid_dw_add_stats_partials": (_i, [_i, _i, _i, _i]),
    "ossid_dw_add_stats_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ossid_stem_pool_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_stem_pool_bwd_partials": (_i, [_i, _i, _i, _i]),
    "ossid_stem_pool_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_dw_bwd_k_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "ossid_dw_bwd_k_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_resample_taps_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "ossid_maxpool_idx_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ossid_maxpool_bwd_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ossid_topk_workspace_bytes": (_sz, [_i, _i]),
    "ossid_topk": (_i, [_vp, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ossid_nms_workspace_bytes": (_sz, [_i]),
    "ossid_nms": (_i, [_vp, _i, _f, _vp, _sz, _vp, _vp, _vp]),
    "ossid_decode_clip_boxes": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "ossid_gather_rows": (_i, [_vp, _i, C.c_longlong, _vp, _i, _i, _vp, _vp]),
    "ossid_dot_expand": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ossid_bias_elu_affine_slice": (_i, [_vp, C.c_longlong, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ossid_bcast_sub_epilogue": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "ossid_amsgrad_step": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp]),
    "ossid_seq_replay": (_i, [C.POINTER(SeqOp), _i, C.POINTER(_vp), _i, C.POINTER(_i)]),
    "ossid_seq_release": (_i, [C.POINTER(SeqOp), _i]),
    "ossid_seq_probe": (_i, [C.c_int32, _f, _vp, C.c_double, C.c_int64, C.c_int32, _f, _sz, C.c_int32, C.c_int32, C.c_int32,
                             C.c_double, C.c_int32, C.c_int64, C.POINTER(C.c_double), _vp]),
}


ABI_VERSION = 6      # OSSID_ABI_VERSION of include/ossid_hip.h: the struct layouts below (tests/test_abi.py compares the two)


def exported_symbols():
    """Names include/ossid_hip.h declares (kept in step by tests/test_abi.py)."""
    return sorted(_PROTOS)


def lib():
    """The loaded library; raises if it has not been built (python -m ossid_code_amd._build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libossid_hip.so is missing (%s): build it with `python -m ossid_code_amd._build` or "
                "__graft_entry__.build(); there is no CPU fallback for the OSSID hot path" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name, None)
            if fn is None:
                continue  # a later round's symbol not built yet; calling it raises in check()
            fn.restype, fn.argtypes = res, args
        have = handle.ossid_abi_version(None, 0)
        if have != ABI_VERSION:
            raise RuntimeError("libossid_hip.so has ABI version %d, this binding expects %d (include/ossid_hip.h, "
                               "OSSID_ABI_VERSION): rebuild the library with `python -m ossid_code_amd._build`" % (have, ABI_VERSION))
        _lib = handle
    return _lib


def fn(name):
    f = getattr(lib(), name, None)
    if f is None:
        raise RuntimeError("libossid_hip.so does not export %s -- rebuild the library" % name)
    g = _counting(f, _MFMA_RULES[name], name) if (_MFMA_COUNT is not None and name in _MFMA_RULES) else f
    if _REC is not None:
        return _REC.wrap(name, f, g)
    return g


# ---- recorded launch sequences ---------------------------------------------------------------------------------------------
# The finetune step is ~2 000 small launches and the single host thread that enqueues all of its HIP streams is on the
# critical path. A Seq is the list of C-ABI launches one fixed-shape piece of the step makes -- a dense block's
# forward, a template encoder's backward -- recorded ONCE while the ordinary Python code runs (`with record(seq):` -- every
# `fn(name)(...)` call is executed AND stored with its argument values, descriptors included) and replayed afterwards by a
# loop that does nothing but call the same entry points with the same arguments and the current stream handles: no
# descriptor building, no tensor allocation, no autograd bookkeeping per launch. Requirements, all the caller's: every
# device address in the arguments is stable across steps (persistent buffers owned by the plan that owns the Seq; Seq.keep
# holds them), and the recorded region contains NO torch kernels (they would run at record time only). Launches carry a
# stream SLOT (0 = the stream current at replay, 1 = the weight-gradient side stream); "wait" entries order the slots.
_REC = None
# entry points that enqueue work (their last argument is the stream) and may appear in a sequence
RECORDABLE = frozenset((
    "ossid_conv_nhwc_fwd", "ossid_conv3x3_wino_fwd", "ossid_chan_op", "ossid_bn_fold_fwd", "ossid_bn_fold_bwd",
    "ossid_colsum_finalize", "ossid_conv_wgrad", "ossid_conv_wgrad_group", "ossid_avgpool2_nhwc",
    "ossid_upsample_nearest_bwd_nhwc", "ossid_maxpool_idx_nhwc", "ossid_maxpool_bwd_nhwc", "ossid_dw_add_nhwc",
    "ossid_dw_bwd_k_nhwc", "ossid_im2col_stem", "ossid_conv_pack_weights", "ossid_conv_pack_weights_dgrad", "ossid_conv_pack_weights_form",
    "ossid_conv_pack_weights_wino", "ossid_fill_zero", "ossid_resample_taps_nhwc",
    "ossid_stem_weight_relayout", "ossid_stem_conv_fwd", "ossid_stem_conv_wgrad",
    "ossid_dw_add_stats_nhwc", "ossid_stem_pool_fwd", "ossid_stem_pool_bwd", "ossid_dense_dgrad1_acc", "ossid_dense_fwd1_stats", "ossid_bn_fold_fwd_rows", "ossid_bn_fold_fwd_tail", "ossid_dense_dgrad3_mask"))
# entry points that only compute sizes / return static data: called through, never stored
_QUERIES = frozenset((
    "ossid_conv_packed_floats", "ossid_conv_packed_floats_form", "ossid_conv_wino_packed_floats", "ossid_chan_op_partials", "ossid_conv_wgrad_workspace_bytes",
    "ossid_conv_wgrad_group_workspace_bytes", "ossid_dw_bwd_k_workspace_floats",
    "ossid_conv3x3_wgrad_splits", "ossid_abi_version", "ossid_conv3x3_wino_workspace_bytes",
    "ossid_conv3x3_wino_pair_workspace_bytes", "ossid_stem_conv_wgrad_workspace_bytes",
    "ossid_dw_add_stats_partials", "ossid_stem_pool_bwd_partials", "ossid_dense_dgrad1_acc_partials", "ossid_dense_fwd1_stats_partials", "ossid_dense_dgrad3_mask_partials", "ossid_conv_split_bf16"))


SEQ_C = os.environ.get("OSSID_SEQ_C", "1") != "0"      # replay through ossid_seq_replay (0: the Python loop, for A/B runs)
_U32 = C.c_uint32


class Seq:
    """A recorded launch sequence (see above). ops: (callable, args without the stream, slot, name) or ("wait", waiter, signal)."""

    def __init__(self):
        self.ops, self.keep, self.uses_side = [], [], False
        self._compiled, self._compiled_len = None, -1
        self.scratch = {}          # (purpose, device, stream slot) -> the sequence's OWN scratch buffer (record.scratch)

    def __len__(self):
        return len(self.ops)

    def run(self, streams):
        """streams: torch.cuda.Stream per slot (slot 0 first). Enqueues every recorded launch: through the C-side loop
        (ossid_seq_replay, csrc/seq.hip) unless bench's flop counter is listening or OSSID_SEQ_C=0."""
        raw = [st.cuda_stream for st in streams]
        cnt = _MFMA_COUNT
        if cnt is None and SEQ_C:
            c = self._compiled if self._compiled_len == len(self.ops) else self._compile()
            if c is not None:
                ns = len(raw)
                rc = c[2](c[0], c[1], (_vp * ns)(*raw), ns, c[3])
                if rc:
                    k = c[3][0]
                    raise RuntimeError("%s failed with status %d (replayed sequence, op %d)"
                                       % (self.ops[k][3] if 0 <= k < len(self.ops) and self.ops[k][0] != "wait" else "stream wait", rc, k))
                return
        for op in self.ops:
            f = op[0]
            if f == "wait":
                streams[op[1]].wait_stream(streams[op[2]])
                continue
            rc = f(*op[1], raw[op[2]])
            if rc:
                raise RuntimeError("%s failed with status %d (replayed sequence)" % (op[3], rc))
            if cnt is not None and op[3] in _MFMA_RULES:
                cnt._add(op[3], _MFMA_RULES[op[3]](op[1]), op[1])

    def _compile(self):
        """The ops as an ossid_seq_op array (built at the first replay, again if launches were recorded since). The recorded
        argument objects stay referenced by self.ops: descriptor structs are read through their addresses at every replay.
        None if a launch does not fit the C-side call shape (the Python loop replays it)."""
        self._release()
        n = len(self.ops)
        arr = (SeqOp * max(n, 1))()
        ok = True
        for k, op in enumerate(self.ops):
            o = arr[k]
            if op[0] == "wait":
                o.fn, o.slot, o.wait_for = None, op[1], op[2]
                continue
            types = _PROTOS[op[3]][1][:-1]
            if len(types) != len(op[1]):
                ok = False
                break
            ni = nf = 0
            for a, t in zip(op[1], types):
                if t is _f or t is C.c_double:
                    if nf >= SEQ_MAX_FP:
                        ok = False
                        break
                    o.fparg[nf] = (_U32.from_buffer_copy(C.c_float(a)).value if t is _f
                                   else C.c_uint64.from_buffer_copy(C.c_double(a)).value)
                    nf += 1
                else:
                    if ni >= SEQ_MAX_INT:
                        ok = False
                        break
                    if isinstance(a, int):
                        v = a
                    elif a is None:
                        v = 0
                    else:
                        v = C.cast(a, _vp).value or 0
                    o.iarg[ni] = v & 0xFFFFFFFFFFFFFFFF
                    ni += 1
            if not ok:
                break
            o.fn = C.cast(op[0], _vp).value
            o.slot, o.n_int, o.n_fp = op[2], ni, nf
        self._compiled_len = n
        if not ok:
            self._compiled = None
            return None
        h = lib()
        self._compiled = (arr, n, h.ossid_seq_replay, (_i * 1)(-1), h.ossid_seq_release)
        return self._compiled

    def _release(self):
        c, self._compiled = self._compiled, None
        if c is not None:
            c[4](c[0], c[1])       # the stream-order ops' events

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


class record:
    """`with record(seq):` -- see Seq. Nested use is an error; `slot(k)` switches the stream slot of the launches inside."""

    def __init__(self, seq):
        self.seq, self.cur_slot = seq, 0

    def __enter__(self):
        global _REC
        if _REC is not None:
            raise RuntimeError("nested launch-sequence recording")
        _REC = self
        return self

    def __exit__(self, *a):
        global _REC
        _REC = None
        return False

    def wrap(self, name, f, run_now):
        """f: the raw entry point (stored); run_now: what executes it at record time (f, or f under bench's flop counter)."""
        if name in _QUERIES:
            return f
        if name not in RECORDABLE:
            raise RuntimeError("%s was called inside a recorded launch sequence but is not recordable" % name)

        def call(*args):
            self.seq.ops.append((f, args[:-1], self.cur_slot, name))
            return run_now(*args)
        return call

    def wait(self, waiter, signal):
        self.seq.ops.append(("wait", waiter, signal))

    def keep(self, *tensors):
        self.seq.keep.extend(t for t in tensors if t is not None)

    def scratch(self, name, nbytes, device):
        """Scratch memory owned by the sequence being recorded, per purpose and stream slot (launches of one slot are
        ordered among themselves at every replay, whatever streams the replay runs on). A request larger than the buffer
        gets a new one; launches recorded earlier keep the old one alive through `keep`."""
        key = (name, str(device), self.cur_slot)
        t = self.seq.scratch.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
            self.seq.scratch[key] = t
            self.seq.keep.append(t)
        return t


def recording():
    return _REC


# ---- executed matrix-core work, counted where the launches are issued ------------------------------------------------------
# `with count_mfma() as c:` around an EAGER pass (no hipGraph replay: a replay issues no host calls) sums, per launch of a
# matrix-core kernel of this library, the multiply-adds the launch asks the MFMA pipe for on real (unpadded) operands:
# direct / phase convolutions 2*B*px*Cout*Cin*taps, the Winograd form 16 multiplies per 2x2 output tile and channel pair
# (not the 36 of the convolution it replaces), weight gradients 2*B*px*Cout*Cin*taps, the decoder tail's first convolution
# with its merged kernel rows. Two sums: `.flops` = those multiply-adds as f32 arithmetic, `.pipe` = the same launches in
# units of the f32 MFMA pipe's TIME: a layer on v_mfma_f32_32x32x2_f32 counts 1:1, a layer on the split-bf16 form (three
# v_mfma_f32_32x32x16_bf16 products per f32 product, each 16x the f32 instruction's rate: csrc/wino.hip) counts 3/16 -- what it
# occupies the pipe for. bench.py divides `.pipe` by the measured time and the f32 matrix peak for `roofline.frac_mfma`: the
# pipe's own busy fraction, which cannot exceed 1 -- unlike a fraction on the reference's nominal flop count.
_MFMA_COUNT = None
SPLIT_BF16_PIPE_WEIGHT = 3.0 / 16.0


def _wgrad_pipe(d):
    """f32-pipe-equivalent flops of one weight gradient."""
    return _wgrad_flops_d(d) * (SPLIT_BF16_PIPE_WEIGHT if lib().ossid_conv_wgrad_split_bf16() else 1.0)


def _pipe_flops(name, args, flops):
    if name in ("ossid_conv3x3_wino_fwd", "ossid_conv3x3_wino_fwd_pair"):
        return flops * (SPLIT_BF16_PIPE_WEIGHT if lib().ossid_conv_wino_split_bf16() else 1.0)
    if name == "ossid_conv_nhwc_fwd":
        form = (args[0]._obj.exact if args else 0) if lib().ossid_conv_split_bf16() else 1
        return flops * (SPLIT_BF16_PIPE_WEIGHT if form == 0 else (2.0 * SPLIT_BF16_PIPE_WEIGHT if form == 2 else 1.0))
    if name in ("ossid_dense_entry", "ossid_dense_layer"):
        return flops * SPLIT_BF16_PIPE_WEIGHT
    if name == "ossid_seg_tail_fwd":
        return flops * (SPLIT_BF16_PIPE_WEIGHT if lib().ossid_seg_tail_split_bf16() else 1.0)
    if name == "ossid_conv_wgrad" and args:
        return _wgrad_pipe(args[0]._obj)
    if name == "ossid_conv_wgrad_group" and args:
        return sum(_wgrad_pipe(args[0][i]) for i in range(args[1]))
    return flops


def _conv_flops(args):
    d = args[0]._obj
    px = d.height * d.width * (4 if d.taps == 4 else 1)        # taps 4: four 2x2 phase convolutions of the SOURCE image
    return 2.0 * d.batch * px * d.cout * d.cin * d.taps


def _wino_flops_d(d):
    return 2.0 * d.batch * ((d.height + 1) // 2) * ((d.width + 1) // 2) * 16 * d.cout * d.cin


def _wgrad_flops_d(d):
    return 2.0 * d.batch * d.height * d.width * d.cout * d.cin * d.taps


_MFMA_RULES = {
    "ossid_conv_nhwc_fwd": _conv_flops,
    "ossid_conv3x3_wino_fwd": lambda a: _wino_flops_d(a[0]._obj),
    "ossid_conv3x3_wino_fwd_pair": lambda a: _wino_flops_d(a[0]._obj) + _wino_flops_d(a[1]._obj),
    "ossid_conv_wgrad": lambda a: _wgrad_flops_d(a[0]._obj),
    "ossid_conv_wgrad_group": lambda a: sum(_wgrad_flops_d(a[0][i]) for i in range(a[1])),
    # (x, B, Hs, Ws, C, H, W, ...): 32 -> 16 on the up-sampled image, kernel rows merged 3 -> 2 away from the border
    "ossid_seg_tail_fwd": lambda a: 2.0 * a[1] * a[5] * a[6] * 16 * a[4] * 6,
    # (buf, ctot, c0, pixels, nlayers, ...): every layer's 1x1 share of the block's c0 input channels
    "ossid_dense_entry": lambda a: 2.0 * a[3] * 128 * a[2] * a[4],
    # (y, buf, B, H, W, ctot, c0, layer, nlayers, ...): the 3x3 plus the 32-channel shares of the later layers
    "ossid_dense_layer": lambda a: 2.0 * a[2] * a[3] * a[4] * 128 * 32 * (9 + a[8] - 1 - a[7]),
}


def _counting(f, rule, name):
    def call(*args):
        if _MFMA_COUNT is not None:
            _MFMA_COUNT._add(name, rule(args), args)
        return f(*args)
    return call


class count_mfma:
    """Context manager: .flops / .pipe / .launches of the matrix-core launches issued inside (see above)."""

    def __enter__(self):
        global _MFMA_COUNT
        self.flops, self.pipe, self.launches, self._outer = 0.0, 0.0, 0, _MFMA_COUNT
        _MFMA_COUNT = self
        return self

    def __exit__(self, *a):
        global _MFMA_COUNT
        _MFMA_COUNT = self._outer
        return False

    def _add(self, name, flops, args=None):
        self.flops += flops
        self.pipe += _pipe_flops(name, args, flops)
        self.launches += 1

    @staticmethod
    def add(flops):
        """For matrix-core work this library does not launch itself (a library GEMM on the product path)."""
        if _MFMA_COUNT is not None:
            _MFMA_COUNT._add("", flops)


def check(rc, what):
    if rc != OSSID_OK:
        raise RuntimeError("%s failed with status %d" % (what, rc))


def ptr(t):
    """Device (or host) address of a tensor / None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("non-contiguous tensor passed across the C ABI")
    return t.data_ptr()


def stream():
    """Raw handle of torch's current HIP stream on the current device (the private fast path: ~0.3 us instead of ~4 us
    through torch.cuda.current_stream() -- the eager finetune step makes ~2000 launches)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()


def on_device(dev):
    """`with on_device(t.device):` -- torch.cuda.device(dev), or nothing at all when dev is already current (one process
    per GPU: always, after start-up)."""
    idx = dev.index if isinstance(dev, torch.device) else dev
    if idx is None or torch._C._cuda_getDevice() == idx:
        return _NULL_CTX
    return torch.cuda.device(dev)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the OSSID hot path runs on the GPU only (got a %s tensor)" % t.device)


PN2_NSTAGES = 9


class StageEvents:
    """OSSID_PN2_NSTAGES+1 HIP events for ossid_pn2_score's stage_events_host argument."""

    def __init__(self):
        self.n = PN2_NSTAGES + 1
        self.arr = (_vp * self.n)()
        for i in range(self.n):
            e = _vp()
            check(fn("ossid_event_create")(C.byref(e)), "ossid_event_create")
            self.arr[i] = e.value

    def elapsed_ms(self):
        """Per-stage milliseconds of the last recorded call (waits for it to finish)."""
        out = []
        for i in range(self.n - 1):
            ms = _f()
            check(fn("ossid_event_elapsed_ms")(self.arr[i], self.arr[i + 1], C.byref(ms)), "ossid_event_elapsed_ms")
            out.append(ms.value)
        return out

    def close(self):
        for i in range(self.n):
            if self.arr[i]:
                fn("ossid_event_destroy")(self.arr[i])
                self.arr[i] = None

    @staticmethod
    def names():
        return fn("ossid_pn2_stage_names")().decode().split(",")
