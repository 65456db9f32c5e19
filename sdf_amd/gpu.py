"""Thin Python view of the C ABI in include/sdfgpu.h (one context per GPU).

Everything here ends in a call into libsdfgpu.so; there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .scene import SdfTreeBuffer


class SdfGpuError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"sdfgpu status {status}: {message}")
        self.status = status
        self.message = message


def camera_look_at(eye, target, up, fov_y, width, height):
    """Pinhole camera POD (geometry of src/main__.rs:261-264)."""
    cam = _ffi.Camera()
    e, t, u = (np.asarray(v, dtype=np.float32) for v in (eye, target, up))
    _ffi.load_library().sdfgpu_camera_look_at(_ffi.fptr(e), _ffi.fptr(t), _ffi.fptr(u), fov_y, width, height, C.byref(cam))
    return cam


def march_params(max_steps=128, eps_hit=1e-3, t_min=0.0, t_max=1e3, normal_h=1e-4):
    return _ffi.MarchParams(max_steps, eps_hit, t_min, t_max, normal_h, 0)


def compact_host(buf, bounds=_ffi.BOUNDS_AUTO, layout=_ffi.LAYOUT_GLAM_SSE2, extensions=0):
    """Validation + compaction only (no GPU): returns (status, TreeInfo, message)."""
    lib = _ffi.load_library()
    info = _ffi.TreeInfo()
    err = C.create_string_buffer(512)
    rc = lib.sdfgpu_compact_host_ext(
        buf.downtree_buffer.ctypes.data if buf.buffer_len else None,
        buf.uptree_buffer.ctypes.data if buf.buffer_len else None,
        buf.buffer_len, layout, bounds, extensions, C.byref(info), err, 512,
    )
    return rc, info, err.value.decode()


class SdfGpu:
    """sdfgpu_ctx: owns the resident tree of one GPU."""

    def __init__(self, device=0):
        self._lib = _ffi.load_library()
        h = C.c_void_p()
        rc = self._lib.sdfgpu_create(device, C.byref(h))
        if rc != 0:
            raise SdfGpuError(rc, f"sdfgpu_create(device={device}) failed: " + self._lib.sdfgpu_status_string(rc).decode())
        self._h = h
        self.device = device
        self._comms = []   # live communicators made on this context: closed before it (sdfgpu.h: destroy them first)

    def close(self):
        if getattr(self, "_h", None):
            for m in list(getattr(self, "_comms", [])):
                m.close()
            self._lib.sdfgpu_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise SdfGpuError(rc, self._lib.sdfgpu_last_error(self._h).decode())

    def sync(self):
        self._check(self._lib.sdfgpu_sync(self._h))

    # ---- tree -----------------------------------------------------------
    def upload_tree(self, buf: SdfTreeBuffer, bounds=_ffi.BOUNDS_AUTO, layout=_ffi.LAYOUT_GLAM_SSE2, extensions=0):
        """extensions: EXT_* bits (0 = evaluate the buffer as the crate is written)."""
        n = buf.buffer_len
        self._check(
            self._lib.sdfgpu_upload_tree_ext(
                self._h,
                buf.downtree_buffer.ctypes.data if n else None,
                buf.uptree_buffer.ctypes.data if n else None,
                n, layout, bounds, extensions,
            )
        )
        return self.info()

    def info(self):
        info = _ffi.TreeInfo()
        self._check(self._lib.sdfgpu_tree_info_get(self._h, C.byref(info)))
        return info

    # ---- nearest ----------------------------------------------------------
    def nearest(self, pts_xyzw):
        """faux_shader::nearest_neighbor for an (n,4) float32 array -> (dist, prim)."""
        p = np.ascontiguousarray(pts_xyzw, dtype=np.float32).reshape(-1, 4)
        n = p.shape[0]
        dist = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        self._check(self._lib.sdfgpu_nearest(self._h, p.ctypes.data, n, dist.ctypes.data, prim.ctypes.data))
        return dist, prim

    def nearest_dev(self, d_pts, n, d_dist, d_prim, stream=0):
        self._check(self._lib.sdfgpu_nearest_dev(self._h, d_pts, n, d_dist, d_prim, stream))

    def nearest_dev_static(self, d_pts, n, d_dist, d_prim, stream=0):
        """Allocation-free variant (scratch in the library's static __device__ array)."""
        self._check(self._lib.sdfgpu_nearest_dev_static(self._h, d_pts, n, d_dist, d_prim, stream))

    def nearest_stats_dev(self, d_pts, n):
        st = _ffi.Stats()
        self._check(self._lib.sdfgpu_nearest_stats_dev(self._h, d_pts, n, C.byref(st)))
        return st.as_dict()

    def knn_stats_dev(self, d_pts, n, k):
        st = _ffi.Stats()
        self._check(self._lib.sdfgpu_knn_stats_dev(self._h, d_pts, n, k, C.byref(st)))
        return st.as_dict()

    def set_option(self, option, value):
        self._check(self._lib.sdfgpu_set_option(self._h, option, float(value)))

    def set_reorder(self, enabled):
        self.set_option(_ffi.OPT_REORDER, 1 if enabled else 0)

    def set_compaction(self, enabled):
        self.set_option(_ffi.OPT_COMPACTION, 1 if enabled else 0)

    def set_tile_order(self, enabled):
        self.set_option(_ffi.OPT_TILE_ORDER, 1 if enabled else 0)

    def set_candidate_grid(self, enabled):
        """Whether the NEXT upload also builds the nearest-candidate grid (default on)."""
        self.set_option(_ffi.OPT_CANDIDATE_GRID, 1 if enabled else 0)

    def set_skip_extension(self, value):
        """0 (default): no widening of the search radius; < 0: half the mean primitive size."""
        self.set_option(_ffi.OPT_SKIP_EXTENSION, value)

    def knn(self, pts_xyzw, k):
        p = np.ascontiguousarray(pts_xyzw, dtype=np.float32).reshape(-1, 4)
        n = p.shape[0]
        dist = np.empty((n, k), np.float32)
        prim = np.empty((n, k), np.uint32)
        self._check(self._lib.sdfgpu_knn(self._h, p.ctypes.data, n, k, dist.ctypes.data, prim.ctypes.data))
        return dist, prim

    def knn_dev(self, d_pts, n, k, d_dist, d_prim, stream=0):
        self._check(self._lib.sdfgpu_knn_dev(self._h, d_pts, n, k, d_dist, d_prim, stream))

    def box_bounds(self, box_block, pts_xyzw):
        """MINDIST / MAXDIST (faux_shader.rs:39-58) of one SdfBoundingBoxBlock at the points."""
        b = np.ascontiguousarray(np.array([box_block], dtype=_ffi.BBOX_DTYPE))
        p = np.ascontiguousarray(pts_xyzw, dtype=np.float32).reshape(-1, 4)
        lo = np.empty(p.shape[0], np.float32)
        hi = np.empty(p.shape[0], np.float32)
        self._check(self._lib.sdfgpu_box_bounds(self._h, b.ctypes.data, p.ctypes.data, p.shape[0], lo.ctypes.data, hi.ctypes.data))
        return lo, hi

    # ---- sphere trace -----------------------------------------------------
    def render(self, cam, width, height, mp):
        np_ = width * height
        hit = np.empty(np_, np.uint8)
        depth = np.empty(np_, np.float32)
        normal = np.empty((np_, 3), np.float32)
        prim = np.empty(np_, np.uint32)
        self._check(
            self._lib.sdfgpu_render(
                self._h, C.byref(cam), width, height, C.byref(mp),
                hit.ctypes.data, depth.ctypes.data, normal.ctypes.data, prim.ctypes.data,
            )
        )
        return {"hit": hit, "depth": depth, "normal": normal, "prim": prim}

    def render_dev(self, cam, width, height, mp, d_hit, d_depth, d_normal, d_prim, stream=0):
        self._check(
            self._lib.sdfgpu_render_dev(
                self._h, C.byref(cam), width, height, C.byref(mp), d_hit, d_depth, d_normal, d_prim, stream
            )
        )

    def render_tiles_dev(self, cam, width, height, mp, tile_first, tile_stride, tile_count,
                         d_hit, d_depth, d_normal, d_prim, stream=0):
        self._check(
            self._lib.sdfgpu_render_tiles_dev(
                self._h, C.byref(cam), width, height, C.byref(mp), tile_first, tile_stride, tile_count,
                d_hit, d_depth, d_normal, d_prim, stream,
            )
        )

    def untile_dev(self, width, height, tile_first, tile_stride, tile_count,
                   t_hit, t_depth, t_normal, t_prim, d_hit, d_depth, d_normal, d_prim, stream=0):
        self._check(
            self._lib.sdfgpu_untile_dev(
                self._h, width, height, tile_first, tile_stride, tile_count,
                t_hit, t_depth, t_normal, t_prim, d_hit, d_depth, d_normal, d_prim, stream,
            )
        )

    def untile_gathered_dev(self, width, height, world, rank, d_gathered, d_mine, d_hit, d_depth, d_normal, d_prim, stream=0):
        """Whole-frame assembly in one launch from the ranks' packed tile buffers (sdfgpu_untile_gathered_dev)."""
        self._check(self._lib.sdfgpu_untile_gathered_dev(self._h, width, height, world, rank, d_gathered, d_mine,
                                                         d_hit, d_depth, d_normal, d_prim, stream))

    def render_stats(self, cam, width, height, mp):
        st = _ffi.Stats()
        self._check(self._lib.sdfgpu_render_stats(self._h, C.byref(cam), width, height, C.byref(mp), C.byref(st)))
        return st.as_dict()


class SdfComm:
    """sdfgpu_comm: an RCCL communicator of the library (one rank per process / GPU) for framebuffer
    assembly.  `id_bytes` comes from SdfComm.make_id(gpu) on one rank and is shipped to the others."""

    @staticmethod
    def make_id(gpu):
        buf = C.create_string_buffer(_ffi.COMM_ID_BYTES)
        gpu._check(gpu._lib.sdfgpu_comm_get_id(gpu._h, buf))
        return buf.raw

    def __init__(self, gpu, id_bytes, rank, world):
        self._gpu = gpu
        h = C.c_void_p()
        gpu._check(gpu._lib.sdfgpu_comm_create(gpu._h, id_bytes, rank, world, C.byref(h)))
        self._h = h
        self.rank, self.world = rank, world
        gpu._comms.append(self)

    def render_frame(self, cam, width, height, mp, root, d_hit, d_depth, d_normal, d_prim, stream=0):
        g = self._gpu
        g._check(g._lib.sdfgpu_render_frame_comm(g._h, self._h, C.byref(cam), width, height, C.byref(mp), root,
                                                 d_hit, d_depth, d_normal, d_prim, stream))

    @classmethod
    def create_all(cls, gpus):
        """One process, several GPUs: a clique over the contexts (sdfgpu_comm_create_all, ncclCommInitAll);
        returns one SdfComm per context, rank i = gpus[i]."""
        n = len(gpus)
        ctxs = (C.c_void_p * n)(*[g._h for g in gpus])
        out = (C.c_void_p * n)()
        gpus[0]._check(gpus[0]._lib.sdfgpu_comm_create_all(ctxs, n, out))
        comms = []
        for i, g in enumerate(gpus):
            m = cls.__new__(cls)
            m._gpu, m._h, m.rank, m.world = g, C.c_void_p(out[i]), i, n
            g._comms.append(m)
            comms.append(m)
        return comms

    @staticmethod
    def render_frame_all(comms, cam, width, height, mp, root, d_hit, d_depth, d_normal, d_prim, streams=None):
        """sdfgpu_render_frame_all: every rank of an in-process clique from this one thread."""
        n = len(comms)
        hs = (C.c_void_p * n)(*[m._h for m in comms])
        st = (C.c_void_p * n)(*streams) if streams is not None else None
        g = comms[root]._gpu
        g._check(g._lib.sdfgpu_render_frame_all(hs, n, C.byref(cam), width, height, C.byref(mp), root,
                                                d_hit, d_depth, d_normal, d_prim, st))

    def close(self):
        if getattr(self, "_h", None):
            self._gpu._lib.sdfgpu_comm_destroy(self._h)
            self._h = None
        if getattr(self, "_gpu", None) is not None and self in self._gpu._comms:
            self._gpu._comms.remove(self)

    __del__ = close


def query_slice(n, rank, world):
    """(first, count) of slice `rank` of a batch of n queries cut into `world` contiguous slices (sdfgpu_query_slice)."""
    lo, cnt = C.c_size_t(), C.c_size_t()
    _ffi.load_library().sdfgpu_query_slice(n, rank, world, C.byref(lo), C.byref(cnt))
    return lo.value, cnt.value


def _sharded(gpus, pts_xyzw, k):
    p = np.ascontiguousarray(pts_xyzw, dtype=np.float32).reshape(-1, 4)
    n = p.shape[0]
    ctxs = (C.c_void_p * len(gpus))(*[g._h for g in gpus])
    dist = np.empty((n, k) if k else n, np.float32)
    prim = np.empty((n, k) if k else n, np.uint32)
    lib = gpus[0]._lib
    if k:
        gpus[0]._check(lib.sdfgpu_knn_sharded(ctxs, len(gpus), p.ctypes.data, n, k, dist.ctypes.data, prim.ctypes.data))
    else:
        gpus[0]._check(lib.sdfgpu_nearest_sharded(ctxs, len(gpus), p.ctypes.data, n, dist.ctypes.data, prim.ctypes.data))
    return dist, prim


def nearest_sharded(gpus, pts_xyzw):
    """One host batch cut into contiguous slices over several contexts (sdfgpu_nearest_sharded)."""
    return _sharded(gpus, pts_xyzw, 0)


def knn_sharded(gpus, pts_xyzw, k):
    return _sharded(gpus, pts_xyzw, k)
