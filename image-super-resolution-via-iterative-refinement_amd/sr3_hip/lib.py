"""ctypes binding of libsr3_mi355x.so (see include/sr3_mi355x.h).

There is no fallback: if the shared library is missing or an entry point fails, this module
raises.  PyTorch is used by callers only for device memory / streams; nothing here touches torch.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SR3_LIBRARY: load another build of the same ABI (in-run A/B of kernel changes); default: the in-tree library
LIB_PATH = os.environ.get('SR3_LIBRARY') or os.path.join(_HERE, 'libsr3_mi355x.so')


class Sr3Error(RuntimeError):
    pass


class UnetDesc(C.Structure):
    _fields_ = [('variant', C.c_int), ('in_channel', C.c_int), ('out_channel', C.c_int),
                ('inner_channel', C.c_int), ('norm_groups', C.c_int), ('n_mults', C.c_int),
                ('channel_mults', C.c_int * 8), ('n_attn_res', C.c_int), ('attn_res', C.c_int * 8),
                ('res_blocks', C.c_int), ('image_size', C.c_int)]


class ParamInfo(C.Structure):
    _fields_ = [('name', C.c_char * 128), ('ndim', C.c_int), ('shape', C.c_int * 4), ('pack', C.c_int),
                ('offset', C.c_size_t), ('numel', C.c_size_t)]


class OpInfo(C.Structure):
    _fields_ = [('kind', C.c_int), ('tile_cfg', C.c_int), ('ksplit', C.c_int), ('ksize', C.c_int), ('stride', C.c_int),
                ('upsample', C.c_int), ('cin', C.c_int), ('cout', C.c_int), ('h_out', C.c_int), ('w_out', C.c_int),
                ('fused_res_conv_cin', C.c_int), ('fused_output_stats', C.c_int), ('flops', C.c_double)]


_P = C.c_void_p
_I = C.c_int
_Z = C.c_size_t

# name -> (restype, argtypes): every symbol include/sr3_mi355x.h and include/sr3_io_mi355x.h declare
_F = C.c_float
_PI = C.POINTER(C.c_int)
SIGNATURES = {
    'sr3_tensor2img': (_I, [_P, _I, _I, _I, _I, _F, _F, _I, _I, _I, _P, _PI, _PI, _PI, _P]),
    'sr3_sse_u8': (_I, [_P, _P, _I, _Z, _P, _P]),
    'sr3_ssim_scratch_bytes': (_Z, [_I, _I, _I, _I]),
    'sr3_ssim_u8': (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, _P, _P]),
    'sr3_eval_scratch_bytes': (_Z, [_I, _I, _I, _I]),
    'sr3_eval_psnr_ssim_f32': (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _P, _Z, _P, _P, _P]),
    'sr3_attention_bwd_f32': (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    'sr3_conv_wgrad_scratch_bytes': (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    'sr3_conv_wgrad_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _Z, _P]),
    'sr3_attention_bwd_scratch_bytes': (_Z, [_I, _I, _I]),
    'sr3_attention_bwd_ex_f32': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _Z, _P]),
    'sr3_resize_scratch_bytes': (_Z, [_I, _I, _I, _I, _I, _I]),
    'sr3_resize_u8': (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P]),
    'sr3_images_u8_to_f32': (_I, [_P, _I, _I, _I, _I, _P, _F, _F, _P, _P]),
    'sr3_version': (_I, []),
    'sr3_selftest_split3': (_I, [_P, C.POINTER(C.c_int), _P]),
    'sr3_last_error': (C.c_char_p, []),
    'sr3_plan_create': (_I, [C.POINTER(UnetDesc), C.POINTER(_P)]),
    'sr3_plan_destroy': (None, [_P]),
    'sr3_plan_num_params': (_I, [_P]),
    'sr3_plan_param_info': (_I, [_P, _I, C.POINTER(ParamInfo)]),
    'sr3_plan_param_floats': (_Z, [_P]),
    'sr3_plan_op_info': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(OpInfo)]),
    'sr3_plan_op_side': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'sr3_plan_num_ops': (_I, [_P, _I]),
    'sr3_plan_forward_flops': (C.c_double, [_P, _I]),
    'sr3_plan_set_option': (_I, [_P, C.c_char_p, _I]),
    'sr3_plan_num_taps': (_I, [_P]),
    'sr3_plan_tap_info': (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_Z), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    'sr3_workspace_bytes': (_Z, [_P, _I]),
    'sr3_plan_derived_bytes': (_Z, [_P]),
    'sr3_plan_bind_derived': (_I, [_P, _P, _Z]),
    'sr3_plan_prepare_derived': (_I, [_P, _P, _P]),
    'sr3_plan_invalidate_derived': (_I, [_P]),
    'sr3_unet_forward': (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _P]),
    'sr3_unet_forward_profile': (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P, _I, _P, _I, _P, _P, _P, _P]),
    'sr3_p_sample_step': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'sr3_p_sample_step_ex': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'sr3_step_decrement': (_I, [_P, _P]),
    'sr3_reverse_step': (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    'sr3_q_sample': (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    'sr3_train_workspace_bytes': (_Z, [_P, _I, _I]),
    'sr3_train_step': (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, C.c_float, C.c_float, C.c_uint,
                            _I, _P, _P, _I, _P]),
    'sr3_adam_step': (_I, [_P, _P, _P, _P, _Z, C.c_float, C.c_float, C.c_float, C.c_float, _I, _P]),
    'sr3_conv_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P,
                          _I, _I, _P, _Z, _P]),
    'sr3_block_conv_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I,
                                _I, _P, _Z, _P]),
    'sr3_conv_dropout_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P,
                                  _I, _I, _P, _Z, C.c_uint, C.c_float, _P]),
    'sr3_dropout_threshold': (C.c_uint, [C.c_float, C.POINTER(C.c_float)]),
    'sr3_conv_scratch_bytes': (_Z, [_I, _I, _I, _I, _I, _I, _I, _I]),
    'sr3_groupnorm_stats_f32': (_I, [_P, _I, _I, _I, _P, _P]),
    'sr3_groupnorm_stats_slices': (_I, [_I, _I, _I]),
    'sr3_conv_stats_slices': (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    'sr3_groupnorm_fold_f32': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, C.c_float, _P, _P]),
    'sr3_attention_f32': (_I, [_P, _I, _I, _I, _P, _P]),
    'sr3_attention_ex_f32': (_I, [_P, _I, _I, _I, _P, _I, _P]),
    'sr3_film_embed_f32': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    'sr3_conv_in_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    'sr3_conv_out_f32': (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
}

_lib = None


def load():
    """dlopen the engine; raises Sr3Error (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Sr3Error('libsr3_mi355x.so is not built (%s); run `python __graft_entry__.py build` '
                       'or csrc/build.sh -- there is no CPU / eager fallback' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.sr3_version() != 1:
        raise Sr3Error('ABI version mismatch: %d' % lib.sr3_version())
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().sr3_last_error()
        raise Sr3Error('libsr3_mi355x error %d: %s' % (rc, (msg or b'').decode()))


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
