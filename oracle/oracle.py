"""ctypes loader of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by anything under sdf_amd/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

MODE_FAUX, MODE_EXHAUSTIVE, MODE_TUNED, MODE_FAUX_FAITHFUL = 0, 1, 2, 3

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `make -C oracle`")
        _lib = C.CDLL(LIB_PATH)
        for name in ("oracle_nearest", "oracle_knn", "oracle_knn_tuned", "oracle_eval_prim", "oracle_bounds", "oracle_render",
                     "oracle_hardware_threads"):
            getattr(_lib, name).restype = C.c_int
    return _lib


EXT_SMOOTH_UNION, EXT_BOX_FRAME, EXT_CLONE_REPAIR = 1, 2, 4


class extensions:
    """with O.extensions(O.EXT_SMOOTH_UNION | ...): the oracle evaluates the opt-in extensions."""

    def __init__(self, mask):
        self.mask = mask

    def __enter__(self):
        lib().oracle_set_extensions.restype = C.c_uint32
        self.old = lib().oracle_set_extensions(C.c_uint32(self.mask))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_extensions(C.c_uint32(self.old))
        return False


def _tree_args(buf):
    n = buf.buffer_len
    return (C.c_void_p(buf.downtree_buffer.ctypes.data if n else None),
            C.c_void_p(buf.uptree_buffer.ctypes.data if n else None), C.c_uint32(n))


def _pts(p):
    return np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 4)


def nearest(buf, pts_xyzw, mode=MODE_FAUX, threads=1):
    p = _pts(pts_xyzw)
    n = p.shape[0]
    dist = np.empty(n, np.float32)
    prim = np.empty(n, np.uint32)
    rc = lib().oracle_nearest(*_tree_args(buf), C.c_void_p(p.ctypes.data), C.c_size_t(n), C.c_int(mode),
                              C.c_int(threads), C.c_void_p(dist.ctypes.data), C.c_void_p(prim.ctypes.data))
    if rc:
        raise RuntimeError(f"oracle_nearest failed: {rc}")
    return dist, prim


def knn(buf, pts_xyzw, k, threads=1, tuned=False):
    """k nearest: exhaustive top-k (the contract), or tuned=True the pruned walk (same results on
    consistent buffers; the CPU baseline of the kNN leg)."""
    p = _pts(pts_xyzw)
    n = p.shape[0]
    dist = np.empty((n, k), np.float32)
    prim = np.empty((n, k), np.uint32)
    fn = lib().oracle_knn_tuned if tuned else lib().oracle_knn
    rc = fn(*_tree_args(buf), C.c_void_p(p.ctypes.data), C.c_size_t(n), C.c_uint32(k),
                          C.c_int(threads), C.c_void_p(dist.ctypes.data), C.c_void_p(prim.ctypes.data))
    if rc:
        raise RuntimeError(f"oracle_knn failed: {rc}")
    return dist, prim


def eval_prim(buf, prim, pts_xyzw):
    p = _pts(pts_xyzw)
    out = np.empty(p.shape[0], np.float32)
    rc = lib().oracle_eval_prim(*_tree_args(buf), C.c_uint32(prim), C.c_void_p(p.ctypes.data),
                                C.c_size_t(p.shape[0]), C.c_void_p(out.ctypes.data))
    if rc:
        raise RuntimeError(f"oracle_eval_prim failed: {rc}")
    return out


def bounds(buf, node, pts_xyzw):
    p = _pts(pts_xyzw)
    lo = np.empty(p.shape[0], np.float32)
    hi = np.empty(p.shape[0], np.float32)
    rc = lib().oracle_bounds(C.c_void_p(buf.downtree_buffer.ctypes.data), C.c_uint32(buf.buffer_len),
                             C.c_uint32(node), C.c_void_p(p.ctypes.data), C.c_size_t(p.shape[0]),
                             C.c_void_p(lo.ctypes.data), C.c_void_p(hi.ctypes.data))
    if rc:
        raise RuntimeError(f"oracle_bounds failed: {rc}")
    return lo, hi


def render(buf, cam, width, height, mp, mode=MODE_TUNED, threads=1, pix_first=0, pix_count=None, pix_stride=1):
    """cam / mp: the ctypes PODs of sdf_amd._ffi (plain data, same layout as include/sdfgpu.h)."""
    if pix_count is None:
        pix_count = (width * height - pix_first + pix_stride - 1) // pix_stride
    hit = np.empty(pix_count, np.uint8)
    depth = np.empty(pix_count, np.float32)
    normal = np.empty((pix_count, 3), np.float32)
    prim = np.empty(pix_count, np.uint32)
    steps = np.empty(pix_count, np.uint32)
    rc = lib().oracle_render(*_tree_args(buf), C.byref(cam), C.c_uint32(width), C.c_uint32(height), C.byref(mp),
                             C.c_int(mode), C.c_int(threads), C.c_size_t(pix_first), C.c_size_t(pix_stride), C.c_size_t(pix_count),
                             C.c_void_p(hit.ctypes.data), C.c_void_p(depth.ctypes.data),
                             C.c_void_p(normal.ctypes.data), C.c_void_p(prim.ctypes.data),
                             C.c_void_p(steps.ctypes.data))
    if rc:
        raise RuntimeError(f"oracle_render failed: {rc}")
    return {"hit": hit, "depth": depth, "normal": normal, "prim": prim, "steps": steps}


def hardware_threads():
    return lib().oracle_hardware_threads()
