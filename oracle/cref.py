"""ctypes bindings to oracle/_build/liboracle.so (the C half of the CPU restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/oracle.c.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


_lib = _load()

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")

_lib.orc_find_background_f32.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_double, _f32p]
_lib.orc_blur_f64.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, C.c_int, _f64p]
_lib.orc_boxsum_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
_lib.orc_ccl8.argtypes = [_u8p, C.c_int, C.c_int, _i32p]
_lib.orc_ccl8.restype = C.c_int
_lib.orc_outer_contour.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int)]
_lib.orc_outer_contour.restype = C.c_int
_lib.orc_gaussian_kernel_f32.argtypes = [C.c_int, C.c_double, _f32p]
_lib.orc_gaussian_blur_f32.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_double, _f32p]
_lib.orc_resize_linear_f32.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int]
_lib.orc_polyexp.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_double, _f32p]
_lib.orc_update_matrices.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p]
_lib.orc_blur_solve.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p]
_lib.orc_fb_levels.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
_lib.orc_fb_levels.restype = C.c_int
_lib.orc_fb_level_geom.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
_lib.orc_farneback.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_double, _f32p]
_lib.orc_flow_magnitude.argtypes = [_f32p, C.c_size_t, _f32p]
_lib.orc_bgr2gray_u8.argtypes = [_u8p, C.c_size_t, _u8p]


def find_background_f32(frames, coeff=1.0):
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    n, h, w = frames.shape
    bg = np.empty((h, w), np.float32)
    _lib.orc_find_background_f32(frames, n, h, w, float(coeff), bg)
    return bg


def blur_f64(src, ksize):
    src = np.ascontiguousarray(src, dtype=np.float64)
    h, w = src.shape
    dst = np.empty_like(src)
    _lib.orc_blur_f64(src, h, w, int(ksize[1]), int(ksize[0]), dst)
    return dst


def boxsum_u8(src, ksize=(30, 30)):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.empty((h, w), np.int32)
    _lib.orc_boxsum_u8(src, h, w, int(ksize[1]), int(ksize[0]), dst)
    return dst


def ccl8(mask):
    m = np.ascontiguousarray(mask != 0, dtype=np.uint8)
    h, w = m.shape
    labels = np.empty((h, w), np.int32)
    n = _lib.orc_ccl8(m, h, w, labels)
    return n, labels


def outer_contour(img, x0, y0, want_points=False):
    img = np.ascontiguousarray(img != 0, dtype=np.uint8)
    h, w = img.shape
    na, nd = C.c_int(0), C.c_int(0)
    if want_points:
        cap = 8 * int(img.sum()) + 8
        pts = np.zeros((cap, 2), np.int32)
        n = _lib.orc_outer_contour(img, h, w, int(x0), int(y0), pts.ctypes.data, cap,
                                   C.byref(na), C.byref(nd))
        return pts[:n].copy(), na.value, nd.value
    n = _lib.orc_outer_contour(img, h, w, int(x0), int(y0), None, 0, C.byref(na), C.byref(nd))
    return n, na.value, nd.value


def gaussian_kernel_f32(n, sigma):
    k = np.empty(n, np.float32)
    _lib.orc_gaussian_kernel_f32(n, float(sigma), k)
    return k


def gaussian_blur_f32(src, n, sigma):
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w = src.shape
    dst = np.empty_like(src)
    _lib.orc_gaussian_blur_f32(src, h, w, int(n), float(sigma), dst)
    return dst


def resize_linear_f32(src, dh, dw):
    src = np.ascontiguousarray(src, dtype=np.float32)
    cn = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.float32)
    _lib.orc_resize_linear_f32(src, sh, sw, cn, dst, dh, dw)
    return dst


def polyexp(src, n=5, sigma=2.0):
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w = src.shape
    dst = np.empty((h, w, 5), np.float32)
    _lib.orc_polyexp(src, h, w, int(n), float(sigma), dst)
    return dst


def update_matrices(R0, R1, flow):
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    M = np.empty((h, w, 5), np.float32)
    _lib.orc_update_matrices(R0, R1, flow, h, w, M)
    return M


def blur_solve(M, block_size=15):
    M = np.ascontiguousarray(M, np.float32)
    h, w = M.shape[:2]
    flow = np.empty((h, w, 2), np.float32)
    _lib.orc_blur_solve(M, h, w, int(block_size), flow)
    return flow


def fb_levels(h, w, pyr_scale=0.5, levels=3):
    return _lib.orc_fb_levels(h, w, float(pyr_scale), int(levels))


def fb_level_geom(h, w, pyr_scale, k):
    lh, lw, sg, sz = C.c_int(), C.c_int(), C.c_double(), C.c_int()
    _lib.orc_fb_level_geom(h, w, float(pyr_scale), k, C.byref(lh), C.byref(lw), C.byref(sg), C.byref(sz))
    return lh.value, lw.value, sg.value, sz.value


def farneback(prev, nxt, pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=2.0):
    prev = np.ascontiguousarray(prev, dtype=np.float32)
    nxt = np.ascontiguousarray(nxt, dtype=np.float32)
    h, w = prev.shape
    flow = np.empty((h, w, 2), np.float32)
    _lib.orc_farneback(prev, nxt, h, w, float(pyr_scale), int(levels), int(winsize), int(iterations),
                       int(poly_n), float(poly_sigma), flow)
    return flow


def flow_magnitude(flow):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    mag = np.empty((h, w), np.float32)
    _lib.orc_flow_magnitude(flow, h * w, mag)
    return mag


def bgr2gray_u8(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    g = np.empty((h, w), np.uint8)
    _lib.orc_bgr2gray_u8(bgr, h * w, g)
    return g
