"""ctypes binding of libipx.so -- one prototype per declaration of include/ipx.h.

The library is the product's only compute path: importing this module raises if it has not
been built (``python -m image_processing_amd.build``); there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPX_LIB") or os.path.join(_HERE, "lib", "libipx.so")   # IPX_LIB: an instrumented build (tools/)


class IpxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libipx status %d: %s" % (status, message))
        self.status = status


class Params(C.Structure):
    """ipx_params of include/ipx.h (the settings of ip.py:167-177 plus the hard-coded constants)."""
    _fields_ = [
        ("consecutive_threshold", C.c_int32),
        ("sliding_window_radius", C.c_int32),
        ("min_cluster_size", C.c_int32),
        ("max_cluster_size", C.c_int32),
        ("min_valid_frames", C.c_int32),
        ("circularity_size_threshold", C.c_int32),
        ("estimate_threshold", C.c_int32),
        ("reserved0", C.c_int32),
        ("brightness_threshold", C.c_double),
        ("flow_threshold", C.c_double),
        ("static_threshold", C.c_double),
        ("min_circularity", C.c_double),
        ("pixel_std_coeff", C.c_double),
    ]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "%s is missing: build it with `python -m image_processing_amd.build` (hipcc, gfx950). "
        "image_processing_amd has no CPU fallback." % LIB_PATH)

# One HIP runtime per process.  libipx.so names libamdhip64 by soname; torch ships its own copy and loads it by path.
# If libipx.so came first it would bring in /opt/rocm's copy, torch would add its own, and the second runtime to
# initialise finds no device ("no ROCm-capable device is detected").  With torch imported first the soname resolves
# to the copy already in the process.  (A caller without torch, e.g. tests/c_abi/group_example.c, simply uses
# /opt/rocm's runtime.)
try:
    import torch  # noqa: F401
except ImportError:                                      # pragma: no cover - the Python host needs torch anyway
    pass

lib = C.CDLL(LIB_PATH)

_p = C.c_void_p
_i = C.c_int
_ctx = C.c_void_p

PROTOTYPES = {
    # name: (restype, argtypes)
    "ipx_last_error": (C.c_char_p, []),
    "ipx_version": (_i, []),
    "ipx_device_count": (_i, [C.POINTER(_i)]),
    "ipx_default_params": (None, [C.POINTER(Params)]),
    "ipx_words_per_row": (_i, [_i]),
    "ipx_ctx_create": (_i, [_i, C.POINTER(_ctx)]),
    "ipx_ctx_destroy": (_i, [_ctx]),
    "ipx_set_params": (_i, [_ctx, C.POINTER(Params)]),
    "ipx_set_stream": (_i, [_ctx, _p]),
    "ipx_synchronize": (_i, [_ctx]),
    "ipx_timer_begin": (_i, [_ctx]),
    "ipx_timer_end": (_i, [_ctx, C.POINTER(C.c_float)]),
    "ipx_profile_enable": (_i, [_ctx, _i]),
    "ipx_profile_reset": (_i, [_ctx]),
    "ipx_profile_kernel_classes": (_i, []),
    "ipx_profile_read": (_i, [_ctx, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                              C.POINTER(C.c_double)]),
    "ipx_profile_read_flow_level": (_i, [_ctx, _i, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ipx_host_register": (_i, [_p, C.c_size_t]),
    "ipx_host_unregister": (_i, [_p]),
    "ipx_upload_frames": (_i, [_ctx, _p, C.c_size_t, _p]),
    "ipx_bgr2gray": (_i, [_ctx, _p, C.c_size_t, _p]),
    "ipx_unpack_dib": (_i, [_ctx, _p, C.c_size_t, _i, _i, _i, _i, _i, _i, _p]),
    "ipx_flow_metric": (_i, [_ctx, _p, _i, _i, _i, _p, _p, _p]),
    "ipx_reserve_flow": (_i, [_ctx, _i, _i, _i]),
    "ipx_flow_status": (_i, [_ctx, C.POINTER(_i), _i]),
    "ipx_debug_fi_spin_limit": (_i, [_ctx, C.c_uint]),
    "ipx_farneback_pair": (_i, [_ctx, _p, _p, _i, _i, _p]),
    "ipx_background": (_i, [_ctx, _p, _i, _i, _i, _i, _p]),
    "ipx_damaged_mask": (_i, [_ctx, _p, _i, _i, _p, _i, _p, _p]),
    "ipx_detect_raw": (_i, [_ctx, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "ipx_remove_bright_regions": (_i, [_ctx, _p, _p, _i, _i, _i, _p]),
    "ipx_noalloc_scratch_bytes": (C.c_size_t, []),
    "ipx_remove_bright_regions_noalloc": (_i, [_ctx, _p, _p, _i, _i, _i, _p]),
    "ipx_persistent_mask": (_i, [_ctx, _p, _p, _i, _i, _i, _p]),
    "ipx_or_planes": (_i, [_ctx, _p, _i, _i, _i, _p]),
    "ipx_static_maps": (_i, [_ctx, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "ipx_static_from_maps": (_i, [_ctx, _p, _p, _i, _i, _p]),
    "ipx_finalize": (_i, [_ctx, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "ipx_dense_counts": (_i, [_ctx, _p, _i, _i, _i, _p]),
    "ipx_cluster_stats": (_i, [_ctx, _p, _p, _p, _i, _i, _i, _p, _i, _p, _p]),
    "ipx_detect_chunk": (_i, [_ctx, _p, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p]),
    "ipx_group_create": (_i, [_p, _i, _i, C.POINTER(_p)]),
    "ipx_group_destroy": (_i, [_p]),
    "ipx_group_set_params": (_i, [_p, C.POINTER(Params)]),
    "ipx_group_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i)]),
    "ipx_group_detect_chunk": (_i, [_p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _p]),
    "ipx_group_submit": (_i, [_p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _p]),
    "ipx_group_wait": (_i, [_p]),
    "ipx_pack_mask": (_i, [_ctx, _p, _i, _i, _i, _p]),
    "ipx_unpack_mask": (_i, [_ctx, _p, _i, _i, _i, _p]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library out of step
    _fn.restype = _res
    _fn.argtypes = _args


def check(status):
    if status != 0:
        raise IpxError(status, (lib.ipx_last_error() or b"").decode(errors="replace"))


def default_params():
    p = Params()
    lib.ipx_default_params(C.byref(p))
    return p
