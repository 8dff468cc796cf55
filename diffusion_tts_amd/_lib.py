"""ctypes binding of libdts_hip.so (the C ABI declared in include/dts.h).

There is no CPU fallback: if the HIP library has not been built, or a kernel call fails, this raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must come first: torch ships its own libamdhip64; loading ours before it would give the
#                              process two HIP runtimes (kernels would then launch on a runtime with no device context)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DTS_LIB_PATH') or os.path.join(_HERE, 'libdts_hip.so')     # override: A/B kernel tuning only

DTS_F32, DTS_BF16, DTS_F16, DTS_F16X3 = 0, 1, 2, 3

_p, _i, _f, _d, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64


class ConvArgs(C.Structure):
    _fields_ = [('x1', _p), ('c1', C.c_int32), ('x2', _p), ('c2', C.c_int32), ('w', _p), ('bias', _p),
                ('bias_nc', _p), ('ld_bias_nc', C.c_int32), ('residual', _p), ('out', _p),
                ('n', C.c_int32), ('hin', C.c_int32), ('win', C.c_int32), ('cout', C.c_int32),
                ('ksize', C.c_int32), ('up', C.c_int32), ('out_scale', C.c_float), ('dtype', C.c_int32),
                ('workspace', _p), ('workspace_bytes', C.c_int64), ('stats_out', _p), ('stats_written', C.c_int32),
                ('ev_start', _p), ('ev_stop', _p), ('gn_coef', _p), ('gn_silu', C.c_int32), ('acc_scale', C.c_float), ('out_split2', C.c_int32),
                ('skip_c', C.c_int32), ('skip_x', _p), ('skip_w', _p), ('skip_acc_scale', C.c_float), ('skip_up', C.c_int32)]


# name -> argtypes (every function returns int status except the three noted below)
SIGNATURES = {
    'dts_nchw_to_nhwc': [_p, _p, _i, _i, _i, _i, _i, _p],
    'dts_nhwc_to_nchw': [_p, _i, _p, _i, _i, _i, _i, _p],
    'dts_nchw_to_nhwc_pad': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'dts_pack_conv_weight': [_p, _p, _i, _i, _i, _i, _i, _p, _p],
    'dts_conv2d': [C.POINTER(ConvArgs), _p],
    'dts_conv_in3': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'dts_conv_out3': [_p, _i, _p, _p, _p, _i, _i, _i, _i, _p],
    'dts_gn_coef': [_p, _i, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _i, _p, _p, _p],
    'dts_gn_coef_strips': [_p, _i, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _i, _p, _p],
    'dts_gn_apply': [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    'dts_gn_apply_x3': [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'dts_gn_fused': [_p, _i, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _i, _p, _i, _p],
    'dts_resample2x': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'dts_attention': [_p, _p, _i, _i, _i, _i, _i, _f, _p],
    'dts_linear': [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'dts_pos_embedding': [_p, _p, _p, _i, _i, _i, _p],
    'dts_edm_precond_in': [_p, _p, _i, _f, _p, _p, _i, _i, _p],
    'dts_edm_precond_out': [_p, _p, _p, _p, _i, _i, _p],
    'dts_split3_f16': [_p, _i, _p, _i, _p, _i64, _p],
    'dts_split2_f16': [_p, _i, _p, _i64, _p],
    'dts_attention_x3': [_p, _p, _i, _i, _i, _i, _i, _f, _p],
    'dts_cast_from_f32': [_p, _p, _i, _i64, _p],
    'dts_cast_to_f32': [_p, _i, _p, _i64, _p],
    'dts_heun_xhat': [_p, _i, _i, _p, _i, _d, _p, _i, _i, _p],
    'dts_heun_euler': [_p, _p, _d, _d, _p, _p, _i64, _p],
    'dts_heun_correct': [_p, _p, _p, _d, _d, _p, _i64, _p],
    'dts_quantize_u8': [_p, _i, _p, _i64, _p],
    'dts_brightness': [_p, _p, _i, _i, _p],
    'dts_u8_to_unit_f32': [_p, _p, _i64, _p],
    'dts_resample_u8': [_p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p],
    'dts_lut_u8_f32': [_p, _p, _p, _i, _i, _i, _p],
    'dts_cosine_rows': [_p, _p, _i, _p, _i, _i, _p],
    'dts_attnpool_tokens': [_p, _p, _p, _i, _i, _i, _i, _p],
    'dts_take_token': [_p, _i, _p, _i, _i, _i, _i, _p],
    'dts_softmax_gather': [_p, _p, _p, _i, _i, _p],
    'dts_candidate_noise': [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    'dts_candidate_noise_sd': [_p, _p, _p, _p, _p, _i, _i, _i64, _p],
    'dts_cfg_combine': [_p, _p, _f, _p, _i, _i64, _p],
    'dts_ddim_candidates': [_p, _p, _p, _p, _p, _i, _f, _f, _f, _i, _i64, _p],
}
OTHER = {'dts_version': ([], _i), 'dts_conv_fuses_gn': ([C.POINTER(ConvArgs)], _i), 'dts_conv_kernel': ([C.POINTER(ConvArgs)], _i), 'dts_conv_folds_skip': ([C.POINTER(ConvArgs)], _i), 'dts_set_tuning': ([_i, _i], _i), 'dts_get_tuning': ([_i], _i), 'dts_last_error': ([], C.c_char_p), 'dts_gn_ws_floats': ([_i, _i], _i64)}

_lib = None
ABI_VERSION = 111              # include/dts.h DTS_ABI_VERSION this binding was written against (ConvArgs = 208 bytes)


def load():
    """Loads the library once; raises if it is missing (build it with `python -m diffusion_tts_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
                           f'Build it with `python -m diffusion_tts_amd.build` or __graft_entry__.build().')
    lib = C.CDLL(LIB_PATH)
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argt, _i
    for name, (argt, rest) in OTHER.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argt, rest
    got = lib.dts_version()
    if got != ABI_VERSION or C.sizeof(ConvArgs) != 208:
        raise RuntimeError(f'{LIB_PATH} has ABI version {got}, this binding needs {ABI_VERSION}: rebuild with `python -m diffusion_tts_amd.build --force`')
    _lib = lib
    return lib


def check(status, what=''):
    if status != 0:
        msg = load().dts_last_error().decode(errors='replace')
        raise RuntimeError(f'libdts_hip {what} failed ({status}): {msg}')


KNOBS = {'att_xcd': 0, 'att_qt': 1, 'conv_tile': 2, 'conv_splits': 3, 'conv_variant': 4, 'gn_fuse': 5, 'att_db': 6, 'conv_stages': 7, 'conv_waves': 8, 'conv_half_round': 9, 'conv_epi32': 10, 'conv_skip_fold': 11}


def set_tuning(name, value):
    """Measurement aid (tools/*_bench.py): switch a kernel variant / block order at run time; -1 restores the default."""
    check(load().dts_set_tuning(KNOBS[name], int(value)), 'dts_set_tuning')
