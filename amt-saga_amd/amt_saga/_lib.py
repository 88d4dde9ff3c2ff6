"""ctypes binding of the C ABI declared in include/amt_saga.h.

The HIP extension is mandatory: there is no CPU fallback in the product path.
``load()`` raises RuntimeError if libamt_saga_hip.so has not been built
(``python amt-saga_amd/build.py`` or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMT_LIB_PATH overrides the in-tree library (diagnostic builds, packaged installs)
LIB_PATH = os.environ.get('AMT_LIB_PATH') or os.path.join(os.path.dirname(_HERE), 'lib', 'libamt_saga_hip.so')

AMT_OK = 0
AMT_E_INVALID, AMT_E_SHAPE, AMT_E_HIP, AMT_E_NOMEM, AMT_E_UNSUPPORTED, AMT_E_ATTRIB = \
    -1, -2, -3, -4, -5, -6

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
vp = C.c_void_p


class SubtractArgs(C.Structure):
    _fields_ = [('resid', vp), ('resid_max', vp), ('guess', vp), ('guess_max', vp),
                ('guess_index', vp), ('guess_frames', vp), ('offset_frames', vp),
                ('new_max', vp), ('resid_stride', C.c_size_t), ('guess_stride', C.c_size_t),
                ('B', C.c_int32), ('T', C.c_int32), ('ldf', C.c_int32), ('F', C.c_int32),
                ('guess_frames_all', C.c_int32), ('normalize', C.c_int32),
                ('relu', C.c_int32), ('overkill_factor', C.c_float)]


class CqtArgs(C.Structure):
    _fields_ = [('wave', vp), ('src_frame', vp), ('bin0', vp), ('phase_inc', vp),
                ('length', vp), ('ref', vp), ('coef', vp), ('out', vp), ('wave_stride', C.c_size_t),
                ('B', C.c_int32), ('L', C.c_int32), ('hop', C.c_int32), ('frames', C.c_int32),
                ('n_bins', C.c_int32), ('n_table', C.c_int32)]


class RdcnnDesc(C.Structure):
    _fields_ = [('n_towers', C.c_int32), ('in_h', C.c_int32 * 2), ('in_w', C.c_int32 * 2),
                ('kh', C.c_int32 * 2), ('kw', C.c_int32 * 2),
                ('pool_h', C.c_int32 * 2), ('pool_w', C.c_int32 * 2),
                ('conv_layers', C.c_int32), ('feature_expand_frequency', C.c_int32),
                ('pool_layer_frequency', C.c_int32), ('residual_frequency', C.c_int32),
                ('dense_units', C.c_int32), ('output_classes', C.c_int32),
                ('out_lo', C.c_float), ('out_hi', C.c_float)]


# name -> (restype, argtypes); every symbol include/amt_saga.h declares
PROTOTYPES = {
    'amt_version': (C.c_int, []),
    'amt_strerror': (C.c_char_p, [C.c_int]),
    'amt_last_hip_error': (C.c_char_p, []),
    'amt_device_info': (C.c_int, [c_int32_p, c_int32_p, C.c_char_p, C.c_int]),
    'amt_stft_plan_create': (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int]),
    'amt_stft_plan_destroy': (C.c_int, [vp]),
    'amt_stft_frames': (C.c_int, [vp, C.c_int]),
    'amt_stft_mag': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, vp, vp, C.c_int,
                               C.c_int, C.c_size_t, vp]),
    'amt_istft': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, vp,
                            C.c_size_t, vp]),
    'amt_window_max': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp]),
    'amt_subtract': (C.c_int, [C.POINTER(SubtractArgs), vp]),
    'amt_subtract_span': (C.c_int, [C.POINTER(SubtractArgs), vp, C.c_int, vp]),
    'amt_compress_bands': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp,
                                     C.c_int, vp, vp, vp, C.c_int, vp]),
    'amt_compress_bands_fmax': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp,
                                     C.c_int, vp, vp, vp, C.c_int, vp, vp]),
    'amt_short_window': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                   vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp]),
    'amt_gather_frames': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, C.c_int,
                                    C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_size_t, vp]),
    'amt_amplitude_to_db': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_float,
                                      C.c_float, vp, vp]),
    'amt_db_to_amplitude': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, vp]),
    'amt_spectral_flatness': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_float, vp, vp]),
    'amt_cqt_slices': (C.c_int, [C.POINTER(CqtArgs), vp]),
    'amt_cqt_slices_complex': (C.c_int, [C.POINTER(CqtArgs), vp, vp]),
    'amt_cqt_coef': (C.c_int, [vp, vp, C.c_int, vp, vp]),
    'amt_cqt_window_max_workspace': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'amt_cqt_window_max': (C.c_int, [vp, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_size_t, vp]),
    'amt_cqt_mfma_table_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'amt_cqt_mfma_table': (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    'amt_cqt_window_max_mfma': (C.c_int, [vp, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp]),
    'amt_round_clamp': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'amt_argmax_rows': (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    'amt_resize_table': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'amt_note_select': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, vp, vp, vp]),
    'amt_pack_events': (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    'amt_affine_i32': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'amt_synth_windows': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_size_t, vp, vp]),
    'amt_sf2_synth_windows': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp,
                                        C.c_size_t, vp, vp]),
    'amt_synth_windows_timbres': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int, vp, C.c_size_t, vp, vp]),
    'amt_guess_notes': (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                  C.c_float, vp, vp]),
    'amt_rdcnn_create': (C.c_int, [C.POINTER(vp), C.POINTER(RdcnnDesc), vp, C.c_size_t]),
    'amt_rdcnn_destroy': (C.c_int, [vp]),
    'amt_rdcnn_param_count': (C.c_size_t, [C.POINTER(RdcnnDesc)]),
    'amt_rdcnn_workspace_bytes': (C.c_size_t, [vp, C.c_int]),
    'amt_rdcnn_forward': (C.c_int, [vp, C.POINTER(vp), C.c_int, vp, vp, vp, C.c_size_t, vp]),
    'amt_rdcnn_flops_per_window': (C.c_double, [vp]),
    'amt_rdcnn_set_mode': (C.c_int, [vp, C.c_int]),
    'amt_trainer_create': (C.c_int, [C.POINTER(vp), C.POINTER(RdcnnDesc), vp, C.c_size_t, C.c_float, C.c_float,
                                     C.c_float]),
    'amt_trainer_destroy': (C.c_int, [vp]),
    'amt_trainer_step': (C.c_int, [vp, C.POINTER(vp), vp, C.c_int, C.c_int, c_float_p, vp, vp]),
    'amt_trainer_get_weights': (C.c_int, [vp, vp, C.c_size_t]),
    'amt_trainer_get_grads': (C.c_int, [vp, vp, C.c_size_t]),
    'amt_fftconv_create': (C.c_int, [C.POINTER(vp), vp, vp, vp, vp, vp]),
    'amt_fftconv_destroy': (C.c_int, [vp]),
    'amt_fftconv_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'amt_fftconv_run': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_size_t, C.c_int, vp]),
    'amt_fftpk_create': (C.c_int, [C.POINTER(vp), vp, vp, vp, vp, vp]),
    'amt_fftpk_destroy': (C.c_int, [vp]),
    'amt_fftpk_workspace_bytes': (C.c_size_t, [C.c_int]),
    'amt_fftpk_run': (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]),
    'amt_probe_mfma_f16': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]),
    'amt_rdcnn_profile': (C.c_int, [vp, C.c_int]),
    'amt_rdcnn_profile_read': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, c_int32_p, C.c_int]),
}

_lib = None


def load():
    """Load the shared library once; fail loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'amt_saga: HIP extension not built (%s missing). Run '
            '`python amt-saga_amd/build.py`; there is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    """Map a status code to the exception the reference raises for it."""
    if status == AMT_OK:
        return
    lib = load()
    msg = lib.amt_strerror(status).decode()
    if status == AMT_E_HIP:
        raise RuntimeError('amt_saga: %s: %s' % (msg, lib.amt_last_hip_error().decode()))
    if status in (AMT_E_NOMEM,):
        raise MemoryError('amt_saga: ' + msg)
    raise ValueError(msg)
