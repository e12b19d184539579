"""ctypes binding of libfasterseg_hip.so (the C ABI declared in include/fasterseg_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module raises, and
every operator in fasterseg_amd.* fails with it.  The library handle is process-local and loaded lazily, so modules
stay picklable (the reference evaluator pickles the network into spawned workers, tools/engine/evaluator.py:128-157).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfasterseg_hip.so")

EXPECTED_ABI = 211          # FS_ABI_VERSION of include/fasterseg_hip.h these bindings were written against
FS_F32, FS_BF16 = 0, 1
FS_CONV_RELU, FS_CONV_TRANSPOSED, FS_CONV_ACCUM, FS_CONV_RELU_TAIL = 1, 2, 4, 8

c_int, c_ll, c_float, c_vp = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cin", "Cout", "R", "S", "stride", "pad", "Ho", "Wo",
                                     "x_cs", "y_cs", "dtype", "flags", "w_os", "w_ts", "vr_H", "vr_W", "vr_relu", "bn_groups",
                                     "n_seg", "n_jump", "k_seg", "k_jump", "g_jump")]


class ZoomDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cin", "Cmid", "Cout", "h", "w", "Ho", "Wo", "x_cs", "y_cs", "dtype", "down", "up")]


class LogitsDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("N", "h", "w", "C", "cs", "H", "W", "dtype")]


class ResizeDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("N", "Hi", "Wi", "Ho", "Wo", "C", "x_cs", "y_cs", "dtype", "relu", "out_nchw")]


class CensusEntry(ctypes.Structure):
    _fields_ = [("family", c_int), ("desc", ConvDesc), ("count", c_ll), ("ms", ctypes.c_double)]


class KernelTime(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 56), ("count", c_ll), ("ms", ctypes.c_double), ("bytes", ctypes.c_double)]


class SgdTensor(ctypes.Structure):
    _fields_ = [("p", c_vp), ("g_off", c_ll), ("numel", c_ll), ("I", c_int), ("taps", c_int), ("pack_fwd", c_vp), ("pack_flip", c_vp)]


# name -> argtypes (restype is int status unless listed in _SPECIAL); order = include/fasterseg_hip.h
SIGNATURES = {
    "fs_pack_weight": [c_vp, c_vp, c_ll, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_vp],
    "fs_unpack_weight_grad": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_ll, c_ll, c_int],
    "fs_conv2d_fwd": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "fs_conv2d_fwd_ws": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll],
    "fs_conv2d_wgrad": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp],
    "fs_conv2d_wgrad_strided": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_ll, c_ll, c_ll],
    "fs_factorized_reduce_fwd": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp],
    "fs_factorized_reduce_wgrad": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp],
    "fs_time_op": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_int, c_int, ctypes.POINTER(ctypes.c_float)],
    "fs_conv2d_wgrad_ws": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_ll, c_ll, c_ll, c_vp, c_ll],
    "fs_pack_weight_frag": [c_vp, c_vp, c_ll, c_ll, c_int, c_int, c_int, c_vp],
    "fs_conv3x3_s1_fwd": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "fs_zoom_cell_fwd": [c_vp, ctypes.POINTER(ZoomDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "fs_conv_stem_fwd": [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int],
    "fs_bilinear_fwd": [c_vp, ctypes.POINTER(ResizeDesc), c_vp, c_vp],
    "fs_bilinear_argmax": [c_vp, ctypes.POINTER(ResizeDesc), c_vp, c_vp],
    "fs_hist_info": [c_vp, c_vp, c_vp, c_int, c_ll, c_int, c_vp, c_vp],
    "fs_bilinear_bwd": [c_vp, ctypes.POINTER(ResizeDesc), c_vp, c_vp, c_vp],
    "fs_bilinear_bwd_nchw": [c_vp, ctypes.POINTER(ResizeDesc), c_vp, c_vp, c_vp],
    "fs_bn_finalize": [c_vp, c_int, c_ll, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "fs_bn_train_apply": [c_vp, c_ll, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                          c_int, c_int],
    "fs_affine_act": [c_vp, c_ll, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int],
    "fs_channel_stats": [c_vp, c_ll, c_int, c_vp, c_int, c_int, c_vp],
    "fs_channel_stats_g": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_int, c_vp],
    "fs_channel_stats_ws": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_ll],
    "fs_bn_bwd_reduce_ws": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_ll],
    "fs_bn_train_apply_g": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp,
                            c_int, c_int, c_int],
    "fs_bn_bwd_reduce_g": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "fs_bn_bwd_apply_g": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_ll, c_int,
                          c_int, c_vp, c_int, c_vp, c_vp, c_vp],
    "fs_bn_group_fwd": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp,
                        c_int, c_int, c_int],
    "fs_bn_group_bwd": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp,
                        c_vp, c_vp],
    "fs_bn_bwd_reduce": [c_vp, c_ll, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp],
    "fs_bn_bwd_apply": [c_vp, c_ll, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_ll, c_int,
                        c_int, c_vp, c_int, c_vp, c_vp],
    "fs_bn_act_train_fwd": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                            c_int, c_int, c_int, c_vp, c_ll],
    "fs_bn_act_train_bwd": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp,
                            c_int, c_vp, c_vp, c_vp, c_ll],
    "fs_conv_bn_act_train_fwd": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_float,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_ll],
    "fs_conv_bn_act_train_bwd": [c_vp, ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp,
                                 c_vp, c_vp, c_vp, c_ll, c_ll, c_ll, c_vp, c_int, c_int, c_int, c_vp, c_ll],
    "fs_nchw_to_nhwc": [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int],
    "fs_nhwc_to_nchw": [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp],
    "fs_copy_channels": [c_vp, c_ll, c_int, c_vp, c_int, c_vp, c_int, c_int],
    "fs_axpy_channels": [c_vp, c_ll, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int],
    "fs_dot": [c_vp, c_ll, c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp],
    "fs_weighted_sum": [c_vp, c_ll, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int],
    "fs_weighted_sum_bwd": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int],
    "fs_weighted_sum_dots": [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp],
    "fs_ohem_ce_fwd": [c_vp, c_vp, c_vp, c_ll, c_int, c_ll, c_int, c_vp, c_vp, c_vp],
    "fs_ohem_ce_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_ll, c_vp],
    "fs_kl_distill_fwd": [c_vp, c_vp, c_vp, c_ll, c_int, c_ll, c_vp, c_vp, c_vp],
    "fs_ohem_ce_up_fwd": [c_vp, ctypes.POINTER(LogitsDesc), c_vp, c_vp, c_int, c_vp, c_vp, c_vp],
    "fs_ohem_ce_up_bwd": [c_vp, ctypes.POINTER(LogitsDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll],
    "fs_kl_distill_up_fwd": [c_vp, ctypes.POINTER(LogitsDesc), c_vp, ctypes.POINTER(LogitsDesc), c_vp, c_vp, c_vp, c_vp],
    "fs_kl_distill_up_bwd": [c_vp, ctypes.POINTER(LogitsDesc), c_vp, ctypes.POINTER(LogitsDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                             c_ll],
    "fs_kl_distill_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_ll, c_vp],
    "fs_exec_program": [c_vp, c_vp, c_ll, c_vp, c_vp, c_int],
    "fs_exec_program_streams": [c_vp, c_int, c_vp, c_ll, c_vp, c_vp, c_int],
    "fs_exec_program_group": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int],
    "fs_sgd_momentum_multi": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_float, c_float, c_float, c_int, c_int],
}
_SPECIAL = {
    "fs_last_error": ([], ctypes.c_char_p),
    "fs_version": ([], c_int),
    "fs_struct_size": ([c_int], c_int),
    "fs_packed_weight_elems": ([c_int, c_int, c_int, c_int], c_ll),
    "fs_debug_force_conv_cfg": ([c_int], None),
    "fs_debug_stem_mfma": ([c_int], None),
    "fs_packed_weight_frag_elems": ([c_int, c_int, c_int], c_ll),
    "fs_sgd_chunk_elems": ([], c_int),
    "fs_sgd_tensor_chunks": ([c_ll, c_int, c_int, c_int], c_ll),
    "fs_loss_up_workspace_bytes": ([ctypes.POINTER(LogitsDesc)], c_ll),
    "fs_zoom_cell_supported": ([ctypes.POINTER(ZoomDesc)], c_int),
    "fs_workspace_counter_bytes": ([], c_ll),
    "fs_set_deterministic": ([c_int], None),
    "fs_get_deterministic": ([], c_int),
    "fs_set_fp32_split": ([c_int], None),
    "fs_get_fp32_split": ([], c_int),
    "fs_census_enable": ([c_int], None),
    "fs_census_read": ([c_vp, c_int], c_int),
    "fs_census_read_kernels": ([c_vp, c_int], c_int),
    "fs_census_tag": ([c_int], None),
    "fs_census_read_tags": ([c_int, c_vp, c_vp], c_int),
    "fs_event_create": ([], c_vp),
    "fs_event_destroy": ([c_vp], None),
}
ALL_SYMBOLS = sorted(list(SIGNATURES) + list(_SPECIAL))

_lib = None


class FasterSegHipError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libfasterseg_hip.so is not built (%s missing): run `python -m fasterseg_amd.build` "
                              "— there is no CPU/eager fallback" % LIB_PATH)
        import torch  # noqa: F401  torch first: its bundled libamdhip64 must be the process's HIP runtime, not a second copy loaded for us
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        for name, (argtypes, restype) in _SPECIAL.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = restype
        got = handle.fs_version()
        if got != EXPECTED_ABI:
            raise ImportError("libfasterseg_hip.so has ABI %d, these bindings expect %d: rebuild with `python -m fasterseg_amd.build "
                              "--force`" % (got, EXPECTED_ABI))
        for which, struct in enumerate((ConvDesc, ResizeDesc, ZoomDesc, SgdTensor, LogitsDesc)):
            if handle.fs_struct_size(which) != ctypes.sizeof(struct):
                raise ImportError("libfasterseg_hip.so: sizeof(%s) is %d in the library, %d in the bindings - stale build" % (
                    struct.__name__, handle.fs_struct_size(which), ctypes.sizeof(struct)))
        _lib = handle
    return _lib


_fns = {}


def call(name, *args):
    """Invoke a status-returning entry point; raise FasterSegHipError(message) on failure."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    status = fn(*args)
    if status != 0:
        msg = lib().fs_last_error()
        raise FasterSegHipError("%s failed (status %d): %s" % (name, status, msg.decode() if msg else "?"))
