"""ctypes binding of libb2f.so (include/b2f.h).  There is no CPU fallback: if the HIP
library is missing or no GPU is present, calls fail loudly."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# B2F_LIB: profiling builds of the same library (tools/build_variant.py), never a different backend
SO_PATH = os.environ.get("B2F_LIB") or os.path.join(_HERE, "libb2f.so")
_lib = None

c_float_p = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/b2f.h declares
SIGNATURES = {
    "b2f_last_error": (C.c_char_p, []),
    "b2f_version": (C.c_int, []),
    "b2f_init": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "b2f_init_ex": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "b2f_destroy": (None, [C.c_void_p]),
    "b2f_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                           C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "b2f_param_count": (C.c_longlong, [C.c_int]),
    "b2f_random_weights": (C.c_int, [C.c_ulonglong, C.c_int, C.c_float, c_float_p, C.c_longlong]),
    "b2f_set_weights": (C.c_int, [C.c_void_p, c_float_p, C.c_longlong]),
    "b2f_get_weights": (C.c_int, [C.c_void_p, c_float_p, C.c_longlong]),
    "b2f_weights_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]),
    "b2f_commit_weights": (C.c_int, [C.c_void_p]),
    "b2f_load_t7": (C.c_int, [C.c_char_p, c_float_p, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "b2f_load_t7_ex": (C.c_int, [C.c_char_p, C.c_char_p, c_float_p, C.c_longlong, C.POINTER(C.c_longlong), C.c_char_p, C.c_int]),
    "b2f_compute_flow": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p, C.c_int, C.c_int,
                                   C.POINTER(C.c_double), C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]),
    "b2f_compute_flow_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, C.c_int, C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]),
    "b2f_compute_flow_batch_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte),
                                            C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_double),
                                            C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]),
    "b2f_forward_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "b2f_forward": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, C.POINTER(c_float_p), C.c_int]),
    "b2f_output_shapes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.c_int]),
    "b2f_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "b2f_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "b2f_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong),
                                   C.c_int, C.POINTER(C.c_int)]),
    "b2f_profile_reset": (C.c_int, [C.c_void_p]),
    "b2f_synchronize": (C.c_int, [C.c_void_p]),
    "b2f_init_multi": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "b2f_destroy_multi": (None, [C.c_void_p]),
    "b2f_multi_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "b2f_multi_context": (C.c_void_p, [C.c_void_p, C.c_int]),
    "b2f_multi_rebroadcast": (C.c_int, [C.c_void_p]),
    "b2f_multi_weights_checksum": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]),
    "b2f_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "b2f_multi_compute_flow_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, C.c_int, C.c_int,
                                               C.POINTER(C.c_double), C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]),
    "b2f_multi_compute_flow_batch_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte),
                                                  C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]),
    "b2f_op_costvol": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, c_float_p]),
    "b2f_op_warp_bhwd": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, c_float_p]),
    "b2f_op_warp_bhwd_backward": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, c_float_p, c_float_p]),
    "b2f_op_costvol_backward": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, c_float_p, c_float_p]),
    "b2f_op_warp_costvol": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, C.c_float, C.c_int,
                                      C.c_int, C.c_int, C.c_int, c_float_p]),
    "b2f_op_conv3x3": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p,
                                 C.c_int, C.c_int, C.c_int, c_float_p]),
    "b2f_op_conv_head16": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "b2f_op_upsample_flow2x": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, c_float_p]),
    "b2f_op_image_scale": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, C.c_int]),
}


class B2FError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise B2FError("%s is missing: build it with `python -m back2future_amd.build` "
                           "(there is no CPU fallback for the computeFlow path)" % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise B2FError(lib().b2f_last_error().decode("utf-8", "replace"))


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
