"""ctypes binding of libtcvom_hip.so (the C ABI declared in include/tcvom_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is
absent, importing this module raises.  (`__graft_entry__.build()` / `make -C
tcvom_amd/csrc` produces the library; it cross-compiles without a GPU.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The 16-bit storage type of activations and packed weights: one build of the library per type, same sources, same ABI (csrc/common.h).
#   TCVOM_DTYPE=bf16  libtcvom_hip.so      (default, round 5) bf16: the type BASELINE.json's north star names for the GCA+TAM window (fp32's
#                     exponent range, no loss scale); the stem + layer1 run the doubled-tap high-precision forward (gca_net.HP_LAYERS) that
#                     keeps the alpha-matte error under the 1e-4 bound.
#   TCVOM_DTYPE=fp16  libtcvom_hip_f16.so  IEEE fp16: 3 more mantissa bits at the same MFMA rate -- the alpha-matte error of 16-bit storage
#                     falls ~35x (tests/test_bf16_noise_floor.py), no high-precision stem (0.8 ms faster per 1080p step); the backward runs
#                     under an internal loss scale with an overflow guard (ops.LOSS_SCALE, ops.LossScaler).  BASELINE config 5 (FBA+TAM) names
#                     this type: `bench.py --config fba` selects it.
DTYPE_NAME = os.environ.get('TCVOM_DTYPE', 'bf16').lower()
if DTYPE_NAME in ('f16', 'half', 'float16'):
    DTYPE_NAME = 'fp16'
if DTYPE_NAME not in ('bf16', 'fp16'):
    raise ImportError('TCVOM_DTYPE must be bf16 or fp16, got %r' % DTYPE_NAME)
_DEFAULT_LIB = os.path.join(_HERE, 'lib', 'libtcvom_hip_f16.so' if DTYPE_NAME == 'fp16' else 'libtcvom_hip.so')
LIB_PATH = os.environ.get('TCVOM_LIB') or _DEFAULT_LIB      # TCVOM_LIB: a study build of the same library (kernel ablations)
if os.environ.get('TCVOM_HIP_LIB'):          # kernel A/B work: load an alternative build of the same ABI
    LIB_PATH = os.environ['TCVOM_HIP_LIB']

MAX_TAPS = 32


class ConvDesc(C.Structure):
    """struct tcvom_conv_desc (include/tcvom_hip.h)."""
    _fields_ = [
        ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('C', C.c_int32),
        ('OH', C.c_int32), ('OW', C.c_int32), ('K', C.c_int32),
        ('PH', C.c_int32), ('PW', C.c_int32),
        ('in_step', C.c_int32), ('out_step', C.c_int32), ('out_off_h', C.c_int32), ('out_off_w', C.c_int32),
        ('ntaps', C.c_int32),
        ('tap_dh', C.c_int32 * MAX_TAPS), ('tap_dw', C.c_int32 * MAX_TAPS), ('tap_w', C.c_int32 * MAX_TAPS),
        ('wt', C.c_int32), ('ldo', C.c_int32), ('act', C.c_int32), ('out_fp32', C.c_int32),
        ('stats_group_offset', C.c_int32), ('batch', C.c_int32), ('w_layout', C.c_int32), ('in_f16', C.c_int32),
        ('in_bstride', C.c_int64), ('w_bstride', C.c_int64), ('out_bstride', C.c_int64), ('vec_bstride', C.c_int64),
        ('stats_bstride', C.c_int64),
    ]


class SnScratch(C.Structure):
    """struct tcvom_sn_scratch."""
    _fields_ = [('tvec', C.c_void_p), ('svec', C.c_void_p), ('sigma', C.c_void_p), ('uhist', C.c_void_p),
                ('vhist', C.c_void_p), ('sum_h', C.c_int64), ('sum_wd', C.c_int64), ('num_layers', C.c_int32)]


class BnSync(C.Structure):
    """struct tcvom_bn_sync: one in-kernel SyncBatchNorm exchange (tcvom_amd/mailbox.py fills it)."""
    _fields_ = [('peers', C.c_void_p), ('world', C.c_int32), ('rank', C.c_int32), ('seq', C.c_uint32), ('ring', C.c_int32),
                ('capacity', C.c_int64), ('timeout_ticks', C.c_int64), ('status', C.c_void_p), ('wait_ticks', C.c_void_p)]


class SnDot(C.Structure):
    """struct tcvom_sn_dot: <dy, y> of a SpectralNorm'd conv as a by-product of its BatchNorm backward."""
    _fields_ = [('out', C.c_void_p), ('frame_stride', C.c_int64), ('eps', C.c_float), ('training', C.c_int32), ('scale', C.c_float)]


class TcvomError(RuntimeError):
    pass


def _load():
    # torch first: its bundled HIP runtime must be the one this library binds to (loading the library before torch
    # would resolve libamdhip64 from /opt/rocm and leave two runtimes in the process, the second one without a device)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'tcvom_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C tcvom_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.' % LIB_PATH)
    return C.CDLL(LIB_PATH)


_lib = _load()
_lib.tcvom_last_error.restype = C.c_char_p

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
DP = C.POINTER(ConvDesc)
SP = C.POINTER(SnScratch)
YP = C.POINTER(BnSync)
TP = C.POINTER(SnDot)

# name -> argtypes (all return int except the explicitly listed ones)
_PROTOS = {
    'tcvom_conv_igemm': [vp, vp, vp, vp, vp, vp, vp, DP, vp],
    'tcvom_conv_stats_groups': [DP, i32],
    'tcvom_conv_igemm_phases': [vp, vp, vp, vp, vp, DP, i32, vp],
    'tcvom_gemm_pair': [vp, vp, vp, vp, vp, DP, i64, vp],
    'tcvom_wgrad_igemm_phases': [vp, vp, vp, DP, i32, i32, vp],
    'tcvom_wgrad_igemm': [vp, vp, vp, DP, i32, vp],
    'tcvom_wgrad_igemm_batched': [vp, vp, vp, i32, DP, i32, i32, vp],
    'tcvom_bn_finalize': [vp, i32, i32, i64, i64, vp, vp, vp, vp, f32, f32, vp, vp, vp, i32, i64, vp],
    'tcvom_bn_finalize_scratch_doubles': [i32],
    'tcvom_bn_ema_update': [vp, vp, vp, i32, f32, f32, i64, vp],
    'tcvom_bn_eval_coeffs': [i32, vp, vp, vp, vp, f32, vp, vp, vp],
    'tcvom_bn_apply': [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, vp],
    'tcvom_bn_apply_mask': [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, vp],
    'tcvom_bn_apply_f16': [vp, vp, vp, i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, vp],
    'tcvom_bn_bwd_reduce_mask': [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, i32, i32, vp],
    'tcvom_bn_bwd_apply_mask': [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i64, i32, i32, vp],
    'tcvom_bn_bwd_reduce3': [vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, vp],
    'tcvom_bn_bwd_apply3': [vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i64, vp],
    'tcvom_bn_bwd_groups': [i64, i32],
    'tcvom_bn_bwd_groups_n': [i64, i32, i32],
    'tcvom_bn_bwd_reduce': [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, vp],
    'tcvom_bn_bwd_reduce_ranged': [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, i32, i32, vp],
    'tcvom_bn_bwd_finalize': [vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, i32, i32, i64, TP, vp],
    'tcvom_bn_ema_multi': [vp, i32, vp, vp, vp],
    'tcvom_bn_reduce_sums': [vp, i32, i32, vp, vp, i32, vp],
    'tcvom_bn_finalize_sums': [vp, i32, i64, i64, vp, vp, f32, vp, vp, i32, i64, vp],
    'tcvom_bn_bwd_finalize_sums': [vp, vp, i32, i64, vp, vp, vp, vp, vp, i32, i32, i64, TP, vp],
    'tcvom_bn_bwd_apply': [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i64, vp],
    'tcvom_bn_bwd_apply_ranged': [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i64, i32, i32, vp],
    'tcvom_bn_finalize_sync': [vp, i32, i32, i64, i64, vp, vp, f32, vp, vp, vp, i32, i64, YP, vp],
    'tcvom_bn_bwd_finalize_sync': [vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, i32, i32, i64, YP, TP, vp],
    'tcvom_mbox_alloc': [i64, C.POINTER(C.c_void_p), vp],
    'tcvom_mbox_open': [vp, C.POINTER(C.c_void_p)],
    'tcvom_mbox_close': [vp],
    'tcvom_mbox_device': [vp, C.POINTER(C.c_int32)],
    'tcvom_mbox_free': [vp],
    'tcvom_sn_power_iteration': [vp, SP, vp, i32, vp, i32, vp, i32, i32, i32, vp],
    'tcvom_sn_pack': [vp, SP, vp, i32, i32, vp, vp, i64, i64, vp],
    'tcvom_sn_apply_blocks': [i32, i32, i32, i32, i64],
    'tcvom_gca_dv': [vp, vp, vp, i32, i32, i64, i32, vp],
    'tcvom_gca_pv': [vp, vp, vp, i32, i32, i64, i32, vp],
    'tcvom_gca_dq_dk': [vp, vp, vp, vp, i32, i32, i64, i32, vp],
    'tcvom_sn_backward': [vp, SP, vp, i32, vp, i32, vp, vp, i64, vp, i32, vp, f32, vp, vp, vp],
    'tcvom_avgpool2': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_avgpool2_f16': [vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_upsample2': [vp, vp, i32, i32, i32, i32, f32, vp],
    'tcvom_sumpool2': [vp, vp, i32, i32, i32, i32, f32, vp],
    'tcvom_reflect_pad1': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_reflect_pad1_bwd': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_add': [vp, vp, vp, vp, i64, vp],
    'tcvom_colsum': [vp, vp, i64, i32, i32, vp],
    'tcvom_transpose_bf16': [vp, vp, i32, i32, i64, i64, i32, i64, i64, vp],
    'tcvom_head_conv_fwd': [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_head_conv_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_maxpool2_idx': [vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_unpool2': [vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_pick2': [vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_relu_bwd': [vp, vp, vp, i64, f32, vp],
    'tcvom_gn_finalize': [vp, i32, i32, i64, i32, vp, vp, f32, vp, vp, vp, i32, i64, vp],
    'tcvom_gn_bwd_finalize': [vp, i32, i32, i64, i32, vp, vp, vp, vp, vp, vp, i32, i64, vp],
    'tcvom_ws_stats': [vp, vp, i32, vp],
    'tcvom_ws_backward': [vp, vp, i32, vp, vp],
    'tcvom_maxpool3s2': [vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_maxpool3s2_bwd': [vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_adaptive_avgpool': [vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_adaptive_avgpool_multi': [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_adaptive_avgpool_multi_ws': [vp, vp, vp, i32, vp, i32, i32, i32, i32, vp],
    'tcvom_adaptive_avgpool_scratch_floats': [vp, i32, i32, i32, i32, i32],
    'tcvom_adaptive_avgpool_bwd': [vp, vp, i32, vp, i32, i32, i32, i32, vp],
    'tcvom_adaptive_avgpool_bwd_add': [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_up2_concat': [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_bilinear': [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_bilinear_up2_bwd': [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_bilinear_small_bwd': [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_fba_head_fwd': [vp, vp, vp, vp, vp, i32, i64, i64, i64, vp],
    'tcvom_fba_head_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, i64, vp],
    'tcvom_fba_input': [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, vp],
    'tcvom_matting_metrics': [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    'tcvom_fba_point_fwd': [vp, vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    'tcvom_fba_point_bwd': [vp, vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, i32, i32, i32, vp],
    'tcvom_excl_abs': [vp, vp, i32, i32, i32, vp],
    'tcvom_excl_terms': [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_avgpool2_f32': [vp, vp, i64, i32, i32, vp],
    'tcvom_excl_bwd': [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    'tcvom_lap_down': [vp, vp, i64, i32, i32, vp],
    'tcvom_lap_resid': [vp, vp, vp, vp, i64, i32, i32, vp],
    'tcvom_fba_loss_finish': [vp, i32, f32, f32, f32, f32, f32, vp, vp],
    'tcvom_fba_loss_coefs': [vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, vp, vp],
    'tcvom_lap_bwd_coarse': [vp, vp, vp, vp, i64, i32, i32, vp],
    'tcvom_lap_bwd_fine': [vp, vp, vp, vp, i64, i32, i32, vp],
    'tcvom_crop_resize_u8': [vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_count_unknown': [vp, i32, i64, vp, vp],
    'tcvom_flow_crop_resize': [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_pad_bottom_right': [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp],
    'tcvom_unfold': [vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_fold': [vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_dim_losses_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, vp],
    'tcvom_dim_losses_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, vp],
    'tcvom_tam_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_tam_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_gca_prepare': [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_row_softmax': [vp, vp, i32, i32, i64, i64, vp],
    'tcvom_row_softmax_bwd': [vp, vp, vp, vp, i32, i32, i64, i64, i32, vp],
    'tcvom_gca_value_patches': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_gca_value_patches_bwd': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_gca_fold': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_gca_unfold': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_gca_patches_bwd': [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_preprocess': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp],
    'tcvom_preprocess_clips': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(i32), f32, i32, vp],
    'tcvom_preprocess_clips_f16': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(i32), f32, i32, vp],
    'tcvom_masked_l1_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, vp],
    'tcvom_masked_l1_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i64, i64, i64, i64, vp],
    'tcvom_avgpool8': [vp, vp, i64, i32, i32, vp],
    'tcvom_att_bce': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, i64, i32, vp],
    'tcvom_att_bce_bwd': [vp, vp, vp, f32, vp, i64, i32, vp],
    'tcvom_adam_mt_guarded': [vp, vp, i32, f32, f32, f32, f32, f32, i64, f32, vp, vp],
    'tcvom_overflow_sink': [vp],
    'tcvom_loss_finalize': [vp, vp, f32, i32, f32, i32, i32, vp],
    'tcvom_adam_mt': [vp, vp, i32, f32, f32, f32, f32, f32, i64, f32, vp],
    'tcvom_abi_version': [],
    'tcvom_act_dtype': [],
    'tcvom_conv_trace_read': [vp, i32],
    'tcvom_gca_dp_softmax_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i32, vp],
    'tcvom_gca_scores_softmax_ok': [i32, i32, i64, i32],
    'tcvom_gca_scores_softmax': [vp, vp, vp, vp, vp, i32, i32, i64, i32, vp],
    'tcvom_gca_scores_exp': [vp, vp, vp, vp, vp, i32, i32, i64, i32, vp],
    'tcvom_gca_softmax_rescale': [vp, vp, i32, i64, i32, vp],
    'tcvom_rowdot_bf16': [vp, vp, i32, vp, i64, i32, vp],
    'tcvom_gca_fold_f32': [vp, vp, i32, i32, i32, i32, vp],
    'tcvom_wgrad_ws_multi': [vp, vp, vp, i32, DP, i32, vp],
    'tcvom_index_pool_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_index_pool_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_index_up_fwd': [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_index_up_bwd': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    'tcvom_conv5x5_c1': [vp, vp, vp, i32, i32, i32, i32, vp],
    'tcvom_conv5x5_c1_wgrad': [vp, vp, vp, i32, i32, i32, vp],
    'tcvom_dw3x3_stats_groups': [i64, i32],
    'tcvom_dw3x3': [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_dw3x3_wgrad': [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    'tcvom_wgrad_ws_max_problems': [],
    'tcvom_wgrad_ws_hetero': [vp, vp, vp, i32, DP, i32, vp, vp],
    'tcvom_wgrad_igemm_hetero_plan': [DP, vp, vp, i32, i32, i32, vp, i32],
    'tcvom_wgrad_igemm_hetero': [vp, vp, vp, i32, vp, vp, i32, i32, i32, vp],
    'tcvom_wgrad_igemm_hetero_max_problems': [],
    'tcvom_wgrad_ws_max_geometries': [],
}
# entry points that return a count, not a status
_PLAIN = {'tcvom_conv_stats_groups', 'tcvom_bn_bwd_groups', 'tcvom_bn_bwd_groups_n', 'tcvom_abi_version', 'tcvom_act_dtype', 'tcvom_bn_finalize_scratch_doubles',
          'tcvom_wgrad_ws_max_problems', 'tcvom_wgrad_ws_max_geometries', 'tcvom_dw3x3_stats_groups', 'tcvom_gca_scores_softmax_ok', 'tcvom_sn_apply_blocks',
          'tcvom_adaptive_avgpool_scratch_floats', 'tcvom_wgrad_igemm_hetero_plan', 'tcvom_wgrad_igemm_hetero_max_problems'}

# entry points that return a string
_STRING = {'tcvom_conv_igemm_variant': [DP, i32], 'tcvom_wgrad_igemm_variant': [DP]}

EXPORTS = sorted(list(_PROTOS) + list(_STRING) + ['tcvom_last_error'])


def _bind():
    fns = {}
    for name, argtypes in _PROTOS.items():
        try:
            fn = getattr(_lib, name)
        except AttributeError as e:
            raise ImportError('tcvom_amd: %s does not export %s (stale build?)' % (LIB_PATH, name)) from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
        fns[name] = fn
    for name, argtypes in _STRING.items():
        try:
            fn = getattr(_lib, name)
        except AttributeError as e:
            raise ImportError('tcvom_amd: %s does not export %s (stale build?)' % (LIB_PATH, name)) from e
        fn.argtypes = argtypes
        fn.restype = C.c_char_p
        fns[name] = fn
    return fns


_FNS = _bind()


def _act_dtype():
    import torch
    kind = _FNS['tcvom_act_dtype']()
    if kind != (1 if DTYPE_NAME == 'fp16' else 0):
        raise ImportError('tcvom_amd: %s stores %s but TCVOM_DTYPE=%s was asked for' % (LIB_PATH, 'fp16' if kind else 'bf16', DTYPE_NAME))
    return torch.float16 if kind else torch.bfloat16


ACT_DTYPE = _act_dtype()         # torch dtype of every activation / packed-weight tensor handed to the library


def last_error():
    return (_lib.tcvom_last_error() or b'').decode()


PROFILE = None   # bench.py sets this to a list to bracket every igemm launch with HIP events


def _profiled(name, args):
    import torch
    if name == 'tcvom_wgrad_ws_multi':
        d, n, nb = args[4][0], 1, args[3]
        info = {'P': d.N * d.PH * d.PW, 'K': d.K, 'C': d.C, 'ntaps': 9, 'tap_w': [0] * 9, 'batch': nb, 'phases': 1}
    elif name == 'tcvom_wgrad_ws_hetero':
        # problems of several geometries in one launch: explicit totals (the per-shape formula of the readers does not apply)
        nb, descs, gidx = args[3], args[4], C.cast(args[6], C.POINTER(C.c_int32))
        gs = [descs[gidx[i]] for i in range(nb)]
        d = descs[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _FNS[name](*args)
        e1.record()
        PROFILE.append((name, {'P': sum(g.N * g.PH * g.PW for g in gs) // nb, 'K': d.K, 'C': d.C, 'ntaps': 9, 'tap_w': [0] * 9, 'batch': nb,
                               'phases': 1, 'variant': _FNS['tcvom_wgrad_igemm_variant'](C.byref(d)).decode() + '+%dgeo' % args[5],
                               'gflop': sum(2.0 * g.N * g.PH * g.PW * g.K * 9 * g.C for g in gs) / 1e9,
                               'algo_bytes': sum(2 * g.N * g.H * g.W * (g.C + g.K) + 4 * g.K * g.C * 9 for g in gs)}, e0, e1))
        return rc
    elif name == 'tcvom_wgrad_igemm_batched':
        arr, n, nb = args[4], args[5], args[3]
        d = arr[0]
        # algorithmic taps: distinct offsets (the hi + residual weight pair of a high-precision layer is ONE tap of the layer)
        taps = sum(len({(arr[i].tap_dh[t], arr[i].tap_dw[t]) for t in range(arr[i].ntaps) if arr[i].tap_w[t] >= 0}) for i in range(n))
        info = {'P': d.N * d.PH * d.PW, 'K': d.K, 'C': d.C, 'ntaps': taps, 'tap_w': [0] * taps, 'batch': nb, 'phases': n}
    elif name.endswith('_phases'):
        arr = args[5 if name == 'tcvom_conv_igemm_phases' else 3]
        n = args[6 if name == 'tcvom_conv_igemm_phases' else 4]
        d = arr[0]
        # algorithmic taps: distinct offsets (the hi + residual weight pair of a high-precision layer is ONE tap of the layer)
        taps = sum(len({(arr[i].tap_dh[t], arr[i].tap_dw[t]) for t in range(arr[i].ntaps) if arr[i].tap_w[t] >= 0}) for i in range(n))
        # (frame-batched launches: `batch` frames of P pixels each, ops._set_frames)
        info = {'P': d.N * d.PH * d.PW, 'K': d.K, 'C': d.C, 'ntaps': taps, 'tap_w': [0] * taps, 'batch': max(int(d.batch), 1), 'phases': n}
    elif name == 'tcvom_gca_dp_softmax_bwd':
        # dP[i][j] = sum_v dO[i][v] V[j][v] with the softmax backward in its epilogue: an N x N x DV GEMM per frame on gemm_nt256<2,4>
        N_, DV_, ld_, B_ = args[8:12]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _FNS[name](*args)
        e1.record()
        # algorithmic bytes: dO, V, P read; T (and T^T, P^T when asked for) written (16-bit)
        nbytes = B_ * 2 * (2 * N_ * DV_ + (4 if args[6] else 2) * N_ * ld_)
        PROFILE.append((name, {'P': N_, 'K': N_, 'C': DV_, 'ntaps': 1, 'tap_w': [0], 'batch': B_, 'phases': 1, 'variant': 'gemm_nt256',
                               'algo_bytes': nbytes}, e0, e1))
        return rc
    elif name in ('tcvom_gca_scores_exp', 'tcvom_gca_dv', 'tcvom_gca_pv', 'tcvom_gca_dq_dk'):
        # the other GEMMs of GuidedCxtAtten on gemm_nt256: (rows P, columns K, reduction C, products), algorithmic bytes = every
        # operand once (16-bit N x N matrices, fp32 gradients)
        if name == 'tcvom_gca_scores_exp':              # S' = c_j <G_i, G_j>, exp + tile statistics in the epilogue (EPI 3)
            N_, D_, ld_, B_ = args[5:9]
            dims, nbytes = (N_, N_, D_, B_), B_ * 2 * (N_ * D_ + N_ * ld_)
        elif name in ('tcvom_gca_dv', 'tcvom_gca_pv'):   # dV = P^T dO (P, dO k-major) / O = P V (V k-major)
            N_, DV_, ld_, B_ = args[3:7]
            dims, nbytes = (N_, DV_, N_, B_), B_ * (2 * N_ * ld_ + 2 * DV_ * N_ + 4 * N_ * DV_)
        else:                                           # dWq = T G and M' = T^T G: two products, T read twice
            N_, D_, ld_, B_ = args[4:8]
            dims, nbytes = (N_, D_, N_, 2 * B_), B_ * (4 * N_ * ld_ + 2 * D_ * ld_ + 8 * N_ * D_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _FNS[name](*args)
        e1.record()
        PROFILE.append((name, {'P': dims[0], 'K': dims[1], 'C': dims[2], 'ntaps': 1, 'tap_w': [0], 'batch': dims[3], 'phases': 1,
                               'variant': 'gemm_nt256', 'algo_bytes': nbytes}, e0, e1))
        return rc
    elif name == 'tcvom_gemm_pair':
        n = 1
        d = args[5]._obj
        info = {'P': d.N * d.PH * d.PW, 'K': d.K, 'C': d.C, 'ntaps': 1, 'tap_w': [0], 'batch': 2 * max(int(d.batch), 1), 'phases': 1}
    else:
        n = 1
        d = args[7 if name == 'tcvom_conv_igemm' else 3]._obj
        info = {'P': d.N * d.PH * d.PW, 'K': d.K, 'C': d.C, 'ntaps': d.ntaps, 'tap_w': list(d.tap_w), 'batch': d.batch, 'phases': 1}
    if name.startswith('tcvom_conv_igemm') or name == 'tcvom_gemm_pair':
        info['variant'] = _FNS['tcvom_conv_igemm_variant'](C.byref(d), n).decode()
    else:
        info['variant'] = _FNS['tcvom_wgrad_igemm_variant'](C.byref(d)).decode()
    # algorithmic bytes of the launch: every operand once -- input pixels x channels, the weights, the output (16-bit, or fp32 output)
    osz = 4 if (name.startswith('tcvom_conv_igemm') or name == 'tcvom_gemm_pair') and int(d.out_fp32) else 2
    if name.startswith('tcvom_wgrad'):
        info['algo_bytes'] = info['batch'] * 2 * (d.N * d.H * d.W * d.C + info['P'] * d.K) + 4 * d.K * d.C * max(info['ntaps'], 1)
    else:
        info['algo_bytes'] = info['batch'] * (2 * d.N * d.H * d.W * d.C + osz * info['P'] * d.K * info['phases']) + 2 * d.K * d.C * max(info['ntaps'], 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = _FNS[name](*args)
    e1.record()
    PROFILE.append((name, info, e0, e1))
    return rc


def _profiled_tam(name, args):
    """The Temporal Attention Module launches, bracketed by HIP events; `bytes` = algorithmic HBM bytes of the call: q, v (or dout),
    k_b, k_f read and out (or dq, dk_b, dk_f) written once as bf16 [B, H, W, C], the unknown mask once, plus the 2 x w^2 fp32
    attention logits of every pixel (forward: written; backward: their gradients read, p / ds scratch written and read)."""
    import torch
    if name == 'tcvom_tam_fwd':
        B, H, W, Cc, win = args[9:14]
        nbytes = 5 * B * H * W * Cc * 2 + B * H * W + 2 * win * win * B * H * W * 4
    else:
        B, H, W, Cc, win = args[13:18]
        nbytes = 7 * B * H * W * Cc * 2 + B * H * W + 2 * win * win * B * H * W * 4 * 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = _FNS[name](*args)
    e1.record()
    PROFILE.append((name, {'variant': name[6:], 'bytes': nbytes, 'P': 0, 'K': 0, 'C': 0, 'ntaps': 0, 'tap_w': [], 'batch': 1}, e0, e1))
    return rc


def call_with_info(name, info, *args):
    """`call` for launches whose algorithmic work the generic readers of bench.py's event-instrumented step cannot derive from the
    arguments (descriptor tables in device memory): `info` = {'variant', 'gflop', 'algo_bytes', ...} is recorded with the events."""
    if PROFILE is None:
        return call(name, *args)
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = _FNS[name](*args)
    e1.record()
    PROFILE.append((name, dict({'P': 0, 'K': 0, 'C': 0, 'ntaps': 1, 'tap_w': [0], 'batch': 1, 'phases': 1}, **info), e0, e1))
    if rc != 0:
        raise TcvomError('%s failed (%d): %s' % (name, rc, last_error()))
    return rc


def call(name, *args):
    """Invoke a status-returning entry point; raise TcvomError on failure."""
    if PROFILE is not None and name in ('tcvom_tam_fwd', 'tcvom_tam_bwd'):
        rc = _profiled_tam(name, args)
    elif PROFILE is not None and name in ('tcvom_conv_igemm', 'tcvom_wgrad_igemm', 'tcvom_conv_igemm_phases', 'tcvom_wgrad_igemm_phases',
                                        'tcvom_wgrad_igemm_batched', 'tcvom_wgrad_ws_multi', 'tcvom_wgrad_ws_hetero', 'tcvom_gemm_pair', 'tcvom_gca_dp_softmax_bwd',
                                        'tcvom_gca_scores_exp', 'tcvom_gca_dv', 'tcvom_gca_pv', 'tcvom_gca_dq_dk'):
        rc = _profiled(name, args)
    else:
        rc = _FNS[name](*args)
    if name in _PLAIN:
        return rc
    if rc != 0:
        raise TcvomError('%s failed (%d): %s' % (name, rc, last_error()))
    return rc


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (or NULL for None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
