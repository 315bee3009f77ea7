"""ctypes binding of ``libadellhip.so`` (C ABI: ``include/adell_hip.h``).

The library is the product: there is no CPU or eager-torch fallback. A missing
library, or a kernel call on a non-CUDA tensor, raises immediately.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADELL_HIP_LIBRARY: another build of the same library (tools/ab_lib.py compares two builds)
LIB_PATH = os.environ.get("ADELL_HIP_LIBRARY") or os.path.join(_HERE, "libadellhip.so")

_lib = None

OK = 0
E_BADARG, E_UNSUPPORTED, E_HIP, E_NOMEM = -1, -2, -3, -4

ACT_IDS = {
    "identity": 0, "swish": 1, "silu": 1, "relu": 2, "leaky_relu": 3, "prelu": 4,
    "gelu": 5, "sigmoid": 6, "tanh": 7, "elu": 8,
}


class AdellHipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "D", "H", "W", "C0", "C1", "Cout", "KD", "KH", "KW", "SD", "SH", "SW",
        "PD", "PH", "PW", "Do", "Ho", "Wo")]


class NormActDesc(ctypes.Structure):
    _fields_ = [
        ("N", ctypes.c_int64), ("V", ctypes.c_int64), ("C", ctypes.c_int32),
        ("stats_per_item", ctypes.c_int32), ("act", ctypes.c_int32),
        ("act_w_n", ctypes.c_int32), ("act_p", ctypes.c_float),
        ("drop_p", ctypes.c_float), ("seed", ctypes.c_uint64),
        ("rng_offset", ctypes.c_uint32),
    ]


class AdnSite(ctypes.Structure):
    """adell_adn_site (include/adell_hip.h): a norm -> dropout -> activation site whose backward
    is fused into the backward-data kernel that produces the gradient of its output."""
    _fields_ = [("y", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
                ("keep_mask", ctypes.c_void_p), ("drop_p", ctypes.c_float),
                ("act_p", ctypes.c_float), ("act", ctypes.c_int32)]


_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> (restype, argtypes); every symbol include/adell_hip.h declares.
SIGNATURES = {
    "adell_abi_version": (_i, []),
    "adell_last_error": (ctypes.c_char_p, []),
    "adell_pack_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "adell_conv3d_fwd_ntiles": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_plan_epoch": (_l, []),
    "adell_conv3d_fwd": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "adell_conv3d_bwd_data": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "adell_pack_weight_f16x3_bytes": (_l, [_i, _i, _i, _i]),
    "adell_pack_weight_f16x3": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "adell_conv3d_fwd_ntiles_f16x3": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_fwd_ntiles_f16x3_ws": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_fwd_f16x3": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 8 + [_i, _vp, _vp]),
    "adell_conv3d_bwd_data_f16x3": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 7),
    "adell_pack_weight_f16x3_multi": (_i, [_vp, _i, _l, _vp]),
    "adell_conv3d_bwd_data_s2_f16x3": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 6),
    "adell_conv3d_bwd_data_s2_f16x3_add": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 7),
    "adell_conv3d_fwd_s2_fused_applicable": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_fwd_s2_fused_ntiles": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_fwd_s2_fused": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 6 + [_i, _vp, _vp]),
    "adell_conv3d_bwd_data_s2_fused_applicable": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_bwd_data_s2_fused": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 7),
    "adell_conv3d_splitk_workspace": (_l, [ctypes.POINTER(ConvDesc), _i]),
    "adell_conv3d_fwd_f16x3_ws": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 8 + [_i, _vp, _vp,
                                                                         ctypes.c_size_t, _vp]),
    "adell_conv3d_bwd_data_f16x3_ws": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 7 + [ctypes.c_size_t, _vp]),
    "adell_conv3d_bwd_data_f16x3_add": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 7
                                        + [ctypes.c_size_t, _vp]),
    "adell_conv3d_bwd_weight_workspace": (_l, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_bwd_weight": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_conv3d_bwd_weight_f16x3_workspace": (_l, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_bwd_weight_f16x3": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_bias_grad_workspace": (_l, [_l, _i]),
    "adell_bias_grad": (_i, [_vp, _l, _i, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_convtranspose3d_k2s2_bwd_weight_workspace": (_l, [_i, _i, _i, _i, _i, _i]),
    "adell_convtranspose3d_k2s2_bwd_weight": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_convtranspose3d_fwd": (_i, [_i] * 9 + [_vp] * 5),
    "adell_convtranspose3d_bwd_data": (_i, [_i] * 9 + [_vp] * 4),
    "adell_convtranspose3d_bwd_weight_workspace": (_l, [_i] * 9),
    "adell_convtranspose3d_bwd_weight": (_i, [_i] * 9 + [_vp] * 4 + [ctypes.c_size_t, _vp]),
    "adell_convtranspose3d_k2s2_fwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "adell_convtranspose3d_k2s2_bwd_data": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "adell_stats_finalize_workspace": (_l, [_i, _i, _i]),
    "adell_stats_finalize": (_i, [_vp, _i, _i, _i, _l, _f, _i, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_bn_running_update": (_i, [_vp] * 5 + [_i, _l, _f, _f, _vp]),
    "adell_channel_partials_ntiles": (_i, [_l]),
    "adell_channel_partials": (_i, [_vp, _i, _l, _i, _vp, _vp]),
    "adell_norm_act_fwd": (_i, [ctypes.POINTER(NormActDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "adell_norm_act_mask_bytes": (_l, [ctypes.POINTER(NormActDesc)]),
    "adell_norm_act_fwd_split": (_i, [ctypes.POINTER(NormActDesc)] + [_vp] * 7 + [_i, _vp, _vp]),
    "adell_split_rows_from_f32": (_i, [_vp, _i, _l, _i, _vp, _vp, _vp]),
    "adell_split_rows_to_f32": (_i, [_vp, _i, _l, _i, _vp, _vp, _vp]),
    "adell_conv3d_f16x3_rows_ok": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_bwd_weight_f16x3_rows_ok": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_bwd_weight_f16x3_rows": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 10
                                           + [ctypes.c_size_t, _vp]),
    "adell_conv3d_fwd_f16x3_rows": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 10 + [_i, _vp, _vp]),
    "adell_norm_act_fwd_mask": (_i, [ctypes.POINTER(NormActDesc)] + [_vp] * 9),
    "adell_norm_act_bwd_from_dt": (_i, [ctypes.POINTER(NormActDesc)] + [_vp] * 5 + [_i, _i, _i, _vp, _vp,
                                                                              ctypes.c_size_t, _vp]),
    "adell_conv3d_bwd_data_f16x3_adn_ntiles": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv3d_bwd_data_f16x3_adn": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 7
                                        + [ctypes.POINTER(AdnSite), ctypes.POINTER(AdnSite), _vp, _i, _vp]),
    "adell_norm_act_bwd_workspace": (_l, [ctypes.POINTER(NormActDesc)]),
    "adell_norm_act_bwd": (_i, [ctypes.POINTER(NormActDesc)] + [_vp] * 11 + [ctypes.c_size_t, _vp]),
    "adell_norm_act_bwd_lowrank": (_i, [ctypes.POINTER(NormActDesc), _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                        _vp, ctypes.c_size_t, _vp]),
    "adell_dice_focal_workspace": (_l, [_i, _l]),
    "adell_dice_focal_fwd": (_i, [_vp, _vp, _i, _l, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_dice_focal_bwd": (_i, [_vp, _vp, _i, _l, _f, _f, _f, _f, _f, _vp, _f, _f, _vp, _vp]),
    "adell_dice_focal_bwd_dev": (_i, [_vp, _vp, _i, _l, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "adell_class_sums_workspace": (_l, [_i, _l, _i]),
    "adell_class_sums_fwd": (_i, [_vp, _vp, _i, _l, _i, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_class_sums_bwd": (_i, [_vp, _vp, _i, _l, _i, _vp, _vp]),
    "adell_sgd_step": (_i, [_vp, _vp, _vp, _l, _f, _f, _f, _i, _i, _f, _vp]),
    "adell_adamw_step": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _l, _f, _vp]),
    "adell_adam_step": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _l, _f, _vp]),
    "adell_ema_update": (_i, [_vp, _vp, _l, _f, _vp]),
    "adell_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _vp]),
    "adell_layernorm_bwd_workspace": (_l, [_l, _i]),
    "adell_layernorm_bwd": (_i, [_vp] * 8 + [_l, _i, _vp, ctypes.c_size_t, _vp]),
    "adell_add_bcast": (_i, [_vp, _vp, _vp, _l, _l, _vp]),
    "adell_sum_bcast": (_i, [_vp, _vp, _l, _l, _vp]),
    "adell_attention_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, ctypes.c_ulonglong,
                                 ctypes.c_uint, _vp, _vp, _vp]),
    "adell_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f,
                                 ctypes.c_ulonglong, ctypes.c_uint, _vp, _vp, _vp, _vp]),
    "adell_attention_strided_ok": (_i, [_i, _i, _i]),
    "adell_attention_fwd_strided": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _f,
                                         ctypes.c_ulonglong, ctypes.c_uint, _vp, _vp, _vp]),
    "adell_attention_bwd_strided": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                         _vp, _f, _f, ctypes.c_ulonglong, ctypes.c_uint, _vp, _vp,
                                         _vp, _vp]),
    "adell_copy_channels": (_i, [_vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "adell_interp_nearest_fwd": (_i, [_vp, _vp] + [_i] * 8 + [_vp]),
    "adell_interp_nearest_bwd": (_i, [_vp, _vp] + [_i] * 8 + [_vp]),
    "adell_interp_linear_fwd": (_i, [_vp, _vp] + [_i] * 8 + [_f] * 3 + [_i, _vp]),
    "adell_interp_linear_bwd": (_i, [_vp, _vp] + [_i] * 8 + [_f] * 3 + [_i, _vp]),
    "adell_fold_x_taps": (_i, [_vp, _vp] + [_i] * 8 + [_vp]),
    "adell_scale_bc": (_i, [_vp, _vp, _vp, _i, _l, _i, _vp]),
    "adell_scale_bc_dscale_workspace_floats": (_l, [_i, _l, _i]),
    "adell_scale_bc_dscale": (_i, [_vp, _vp, _vp, _i, _l, _i, _vp, _vp]),
    "adell_cse_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _vp]),
    "adell_cse_apply_bwd_workspace_floats": (_l, [_i, _l, _i]),
    "adell_cse_apply_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _vp, _vp]),
    "adell_bcast_nc": (_i, [_vp, _vp, _i, _l, _i, _f, _vp]),
    "adell_maxpool3d_fwd": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "adell_maxpool3d_bwd": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "adell_dwconv3d_fwd": (_i, [_i] * 8 + [_vp] * 5),
    "adell_dwconv3d_bwd_data": (_i, [_i] * 8 + [_vp] * 4),
    "adell_dwconv3d_bwd_weight_workspace_floats": (_l, [_i] * 8),
    "adell_dwconv3d_bwd_weight": (_i, [_i] * 8 + [_vp] * 6),
    "adell_vicreg_scratch_floats": (_l, [_i, _i]),
    "adell_vicreg_fwd": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "adell_vicreg_bwd": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "adell_convt_k2_applicable": (_i, [_i] * 6),
    "adell_convt_k2_fwd": (_i, [_i] * 6 + [_vp] * 5),
    "adell_convt_k2_bwd_data": (_i, [_i] * 6 + [_vp] * 4),
    "adell_convt_k2_wgrad_workspace": (_l, [_i] * 6),
    "adell_convt_k2_bwd_weight": (_i, [_i] * 6 + [_vp] * 5 + [ctypes.c_size_t, _vp]),
    "adell_convt_k221_applicable": (_i, [_i] * 6),
    "adell_convt_k221_fwd": (_i, [_i] * 6 + [_vp] * 5),
    "adell_convt_k221_bwd_data": (_i, [_i] * 6 + [_vp] * 4),
    "adell_convt_k221_wgrad_workspace": (_l, [_i] * 6),
    "adell_convt_k221_bwd_weight": (_i, [_i] * 6 + [_vp] * 5 + [ctypes.c_size_t, _vp]),
    "adell_conv_cinfold_applicable": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv_cinfold_ntiles": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv_cinfold_fwd": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 5 + [_i, _vp]),
    "adell_conv_cinfold_fwd_f16x3": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 5 + [_i, _vp]),
    "adell_conv_cinfold_wgrad_workspace": (_l, [ctypes.POINTER(ConvDesc)]),
    "adell_conv_cinfold_bwd_weight": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 5
                                      + [ctypes.c_size_t, _vp]),
    "adell_conv_cinfold_bwd_weight_f16x3": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 5
                                            + [ctypes.c_size_t, _vp]),
    "adell_conv_cinfold_dx_applicable": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv_cinfold_bwd_data": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 4),
    "adell_conv_cinfold_bwd_data_f16x3": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 4),
    "adell_pair_loss_scratch_floats": (_l, [_i, _i]),
    "adell_pair_loss_fwd": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "adell_pair_loss_bwd": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "adell_loco_loss_workspace": (_l, [_i, _l, _i]),
    "adell_loco_loss_fwd": (_i, [_vp, _vp, _i, _l, _i, _f, _f, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_loco_loss_bwd": (_i, [_vp, _vp, _vp, _i, _l, _i, _f, _f, _vp, _vp, _vp]),
    "adell_prelu_wgrad_workspace": (_l, [ctypes.POINTER(NormActDesc)]),
    "adell_prelu_wgrad": (_i, [ctypes.POINTER(NormActDesc)] + [_vp] * 8 + [ctypes.c_size_t, _vp]),
    "adell_convtranspose3d_fwd_f16x3": (_i, [_i] * 9 + [_vp] * 7),
    "adell_convtranspose3d_bwd_data_f16x3": (_i, [_i] * 9 + [_vp] * 6),
    "adell_conv1_small_applicable": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv1_small_fwd": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 6),
    "adell_conv1_small_bwd_data": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 5),
    "adell_conv1_small_wgrad_workspace": (_l, [ctypes.POINTER(ConvDesc)]),
    "adell_conv1_small_bwd_weight": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 6 + [ctypes.c_size_t, _vp]),
    "adell_conv_cin_small_applicable": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv_cin_small_ntiles": (_i, [ctypes.POINTER(ConvDesc)]),
    "adell_conv_cin_small_fwd": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 5 + [_i, _vp]),
    "adell_conv_cin_small_bwd_data": (_i, [ctypes.POINTER(ConvDesc)] + [_vp] * 4),
    "adell_multi_copy": (_i, [_vp, _i, _vp, _vp]),
    "adell_item_stats_workspace": (_l, [_i, _l]),
    "adell_item_stats": (_i, [_vp, _i, _l, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_aug_intensity": (_i, [_vp, _vp, _i, _l, _vp, ctypes.c_uint64, ctypes.c_uint32, _vp]),
    "adell_affine_sample": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "adell_axis_filter": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "adell_bias_field": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "adell_axis_lut_sample": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "adell_dw_mfma_ok": (_i, [_i] * 8 + [_vp, _vp]),
    "adell_dw_dense_ok": (_i, [_i] * 8 + [_vp, _vp]),
    "adell_dw_wgrad_mfma_ok": (_i, [_i] * 8 + [_vp, _vp]),
    "adell_wgrad_zring_plan": (_i, [_i] * 16 + [_vp]),
    "adell_rowscale_fwd": (_i, [_vp] * 5 + [_i, _i, _vp]),
    "adell_rowscale_bwd": (_i, [_vp] * 8 + [_i, _i, _vp]),
    "adell_window_ndhwc": (_i, [_vp, _vp] + [_i] * 11 + [_vp]),
    "adell_gibbs_workspace": (_l, [_i, _i, _i, _i, _i]),
    "adell_gibbs_lowpass": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_gather_nd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "adell_layernorm_rows_fwd": (_i, [_vp, _l, _i, _i, _l, _l, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "adell_layernorm_rows_bwd_workspace": (_l, [_l, _i]),
    "adell_layernorm_rows_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _l, _l, _vp, _l, _l,
                                      _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "adell_winattn_fwd": (_i, [_vp, _vp, _vp, _l, _l, _vp, _vp, _i, _l, _i, _i, _i, _i, _f, _f,
                               ctypes.c_ulong, ctypes.c_uint, _vp, _vp, _vp]),
    "adell_winattn_bwd": (_i, [_vp, _vp, _vp, _l, _l, _vp, _vp, _i, _vp, _vp, _vp, _l, _i, _i, _i,
                               _i, _f, _f, ctypes.c_ulong, ctypes.c_uint, _vp, _vp, _vp, _vp, _vp]),
    "adell_gemm_f32_workspace_floats": (_l, [_i, _i, _i]),
    "adell_gemm_f32": (_i, [_i, _i, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp, _vp, _l, _vp, _vp]),
    "adell_absmax_f32": (_i, [_vp, _l, _vp, _vp]),
    "adell_gemm_f16x3_applicable": (_i, [_i, _i, _i, _vp, _l, _i, _vp, _l, _i]),
    "adell_gemm_f16x3_workspace_floats": (_l, [_i, _i, _i]),
    "adell_gemm_f16x3": (_i, [_i, _i, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp, _vp, _l, _vp, _vp,
                              _vp, _vp]),
    "adell_gemm_f16x3_act": (_i, [_i, _i, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp, _vp, _l, _vp,
                                  _vp, _vp, _i, _f, _vp, _vp, _vp]),
    "adell_optim_step": (_i, [_i, _vp, _vp, _vp, _vp, _l, _f, _f, _f, ctypes.POINTER(ctypes.c_float),
                              _vp]),
    "adell_seg_loss_workspace": (_l, [_i, _l, _i]),
    "adell_seg_loss_fwd": (_i, [_i, _vp, _vp, _vp, _i, _l, _i, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp,
                                ctypes.c_size_t, _vp]),
    "adell_seg_loss_bwd": (_i, [_i, _vp, _vp, _vp, _i, _l, _i, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp,
                                _vp]),
    "adell_channel_softmax_fwd": (_i, [_vp, _vp, _l, _i, _vp]),
    "adell_channel_softmax_bwd": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "adell_channel_max_fwd": (_i, [_vp, _vp, _vp, _i, _l, _i, _vp]),
    "adell_channel_max_bwd": (_i, [_vp, _vp, _vp, _i, _l, _i, _vp]),
    "adell_debug_force_conv_cfg": (None, [_i]),
    "adell_set_tuning": (_i, [ctypes.c_char_p, _i]),
    "adell_rng_advance": (_i, [ctypes.c_uint32, _i, _vp]),
    "adell_get_tuning": (_i, [ctypes.c_char_p]),
}


def lib():
    """Load (once) and return the ctypes handle; raises if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AdellHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C adell_mri_amd/csrc`). adell_mri_amd has no fallback path."
            )
        # torch ships its own libamdhip64; it must be the HIP runtime already loaded when
        # libadellhip.so is resolved, or the process ends up with two runtimes (and this
        # library with one that sees no device and none of torch's allocations).
        import torch  # noqa: F401

        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc):
    if rc != OK:
        msg = lib().adell_last_error().decode("utf-8", "replace")
        kind = {E_BADARG: "bad argument", E_UNSUPPORTED: "unsupported", E_HIP: "HIP error",
                E_NOMEM: "out of memory"}.get(rc, f"error {rc}")
        raise AdellHipError(f"libadellhip: {kind}: {msg}")
    return rc


class tuning:
    """``with _lib.tuning(igemm_nospec=1): ...`` -- set launch-plan switches (adell_set_tuning)
    for the duration of a block; used by the A/B tests of the kernel instances."""

    def __init__(self, **switches):
        self.switches = switches
        self.old = {}

    def __enter__(self):
        h = lib()
        for k, v in self.switches.items():
            self.old[k] = h.adell_get_tuning(k.encode())
            check(h.adell_set_tuning(k.encode(), int(v)))
        return self

    def __exit__(self, *exc):
        h = lib()
        for k, v in self.old.items():
            h.adell_set_tuning(k.encode(), int(v))
        return False
