"""Python mirror of the crate's scene-construction API over libsdfgpu.so's host side.

    crate (sdf/src/node.rs:371-414)                     here
    SdfBuilder::primitive(SdfSphere { radius })         SdfBuilder.primitive(SdfSphere(radius))
        .transform(Transform::from_translation(..))         .transform(Transform.from_translation(..))
        .operation(SdfUnion { smooth_radius: 0.0 })         .operation(SdfUnion(0.0))
        .with(builder)                                      .with_(builder)
        .finalize()                                         .finalize()        -> SdfNode
    node.expanded()              (node.rs:209)          node.expanded()         -> ExpandedSdfNode
    expanded.make_buffer()       (node.rs:75)           expanded.make_buffer()  -> SdfTreeBuffer

All geometry is computed by the C++ host library (sdf_amd/csrc/host); this file
only holds handles.  A builder is consumed by the call that uses it, exactly
like the by-value `self` of the Rust methods.
"""
import ctypes as C
import math

import numpy as np

from . import _ffi


class SdfError(RuntimeError):
    pass


def _host_err():
    return _ffi.load_library().sdfhost_last_error().decode()


def _opts(opts):
    if opts is None:
        return None
    o = _ffi.HostOptions()
    o.pca_trans_inverse = 1 if opts.get("pca_trans_inverse") else 0
    o.sibling_other_box = 1 if opts.get("sibling_other_box") else 0
    o.merge_pad_rel = float(opts.get("merge_pad_rel", -1.0))
    return C.byref(o)


class Quat:
    """glam Quat (x, y, z, w)."""

    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.xyzw = np.array([x, y, z, w], dtype=np.float32)

    @staticmethod
    def identity():
        return Quat()

    @staticmethod
    def from_euler_xyz(a, b, c):
        """Quat::from_euler(EulerRot::XYZ, a, b, c) (node.rs:482-487)."""
        out = np.zeros(4, dtype=np.float32)
        _ffi.load_library().sdfhost_quat_from_euler_xyz(a, b, c, _ffi.fptr(out))
        return Quat(*out)


class Transform:
    """bevy Transform: scale, then rotation, then translation."""

    def __init__(self, translation=(0, 0, 0), rotation=None, scale=(1, 1, 1)):
        self.translation = np.array(translation, dtype=np.float32)
        self.rotation = rotation if rotation is not None else Quat.identity()
        self.scale = np.array(scale, dtype=np.float32)

    @staticmethod
    def from_translation(t):
        return Transform(translation=t)

    @staticmethod
    def from_xyz(x, y, z):
        return Transform(translation=(x, y, z))

    @staticmethod
    def from_rotation(q):
        return Transform(rotation=q)

    @staticmethod
    def from_scale(s):
        return Transform(scale=s)

    def compute_matrix(self):
        """Mat4::from_scale_rotation_translation as a 4x4 float64 array (row, col)."""
        x, y, z, w = (float(v) for v in self.rotation.xyzw)
        r = np.array(
            [
                [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
            ]
        )
        m = np.eye(4)
        m[:3, :3] = r * self.scale.astype(np.float64)[None, :]
        m[:3, 3] = self.translation
        return m


class SdfSphere:
    """elements.rs:85-118 (primitive op 0)."""

    def __init__(self, radius):
        self.radius = float(radius)

    def distance_to(self, p):
        return math.sqrt(sum(float(v) ** 2 for v in p)) - self.radius


class SdfBox:
    """Build-defined solid box (primitive op 2, SURVEY C-7); half extents."""

    def __init__(self, half):
        self.half = tuple(float(v) for v in half)

    def distance_to(self, p):
        q = [abs(float(p[i])) - self.half[i] for i in range(3)]
        return math.sqrt(sum(max(v, 0.0) ** 2 for v in q)) + min(max(q), 0.0)


class SdfBoxFrame:
    """elements.rs:120-152 (primitive op 5; private and unfinished in the crate: it has no distance
    formula there, so distance_to is the trait default |p|)."""

    def __init__(self, dimension, thickness):
        self.dimension = tuple(float(v) for v in dimension)
        self.thickness = float(thickness)

    def distance_to(self, p):
        return math.sqrt(sum(float(v) ** 2 for v in p))


class SdfUnion:
    """elements.rs:157-212 (operation op 0)."""

    def __init__(self, smooth_radius=0.0):
        self.smooth_radius = float(smooth_radius)


class SdfCaaClone:
    """elements.rs:215-262 (operation op 1, one slot): lattice repetition of its child."""

    def __init__(self, displacement, neg_limit, pos_limit):
        self.displacement = np.asarray(displacement, np.float32).reshape(3)
        self.neg_limit = np.asarray(neg_limit, np.float32).reshape(3)
        self.pos_limit = np.asarray(pos_limit, np.float32).reshape(3)


class SdfTreeBuffer:
    """component.rs:80-96: downtree_buffer / uptree_buffer / buffer_len."""

    def __init__(self, downtree, uptree):
        assert downtree.dtype == _ffi.DT_DTYPE and uptree.dtype == _ffi.UT_DTYPE
        assert len(downtree) == len(uptree)
        self.downtree_buffer = np.ascontiguousarray(downtree)
        self.uptree_buffer = np.ascontiguousarray(uptree)
        self.buffer_len = len(downtree)

    @staticmethod
    def make_empty():
        return SdfTreeBuffer(np.zeros(0, _ffi.DT_DTYPE), np.zeros(0, _ffi.UT_DTYPE))

    @property
    def n_prims(self):
        return int(self.downtree_buffer["is_primitive"].sum())

    def save(self, path):
        np.savez_compressed(path, dt=self.downtree_buffer.view(np.uint8), ut=self.uptree_buffer.view(np.uint8))

    @staticmethod
    def load(path):
        z = np.load(path)
        return SdfTreeBuffer(z["dt"].view(_ffi.DT_DTYPE), z["ut"].view(_ffi.UT_DTYPE))


class ExpandedSdfNode:
    """node.rs:20-139."""

    def __init__(self, handle, opts):
        self._h, self._opts = handle, opts

    def make_buffer(self):
        lib = _ffi.load_library()
        b = lib.sdfhost_make_buffer(self._h, _opts(self._opts))
        if not b:
            raise SdfError(_host_err())
        n = lib.sdfhost_buffer_len(b)
        if n == 0:
            lib.sdfhost_buffer_free(b)
            return SdfTreeBuffer.make_empty()
        # The arrays view the builder's own memory (170 MB for 100k primitives: copying it cost as much as
        # building it); the C buffer lives as long as either array, or any view of one, does.
        owner = _HostBuffer(lib, b)

        def view(ptr, nbytes, dtype):
            raw = (C.c_uint8 * nbytes).from_address(ptr)
            raw._owner = owner   # numpy keeps `raw` as the array's base, `raw` keeps the C buffer
            return np.frombuffer(raw, dtype=np.uint8).view(dtype)

        dt = view(C.cast(lib.sdfhost_buffer_downtree(b), C.c_void_p).value, n * 624, _ffi.DT_DTYPE)
        ut = view(C.cast(lib.sdfhost_buffer_uptree(b), C.c_void_p).value, n * 224, _ffi.UT_DTYPE)
        return SdfTreeBuffer(dt, ut)

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.load_library().sdfhost_expanded_free(self._h)
            self._h = None


class _HostBuffer:
    """Owner of one sdfhost_buffer (freed when the last array viewing it goes away)."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    def __del__(self):
        if self._h:
            self._lib.sdfhost_buffer_free(self._h)
            self._h = None


class SdfNode:
    """node.rs:141-235 (finalized)."""

    def __init__(self, handle, opts):
        self._h, self._opts = handle, opts

    @property
    def num_slots(self):
        return _ffi.load_library().sdfhost_node_num_slots(self._h)

    @property
    def bbox(self):
        out = np.zeros(1, _ffi.BBOX_DTYPE)
        if _ffi.load_library().sdfhost_node_bbox(self._h, out.ctypes.data) != 0:
            raise SdfError(_host_err())
        return out[0]

    def nearest_neighbor(self, point):
        """node.rs:331-368 (recursive n-ary query, host).  Returns (distance, primitive ordinal)."""
        p = np.ascontiguousarray(point, dtype=np.float32).reshape(3)
        d = C.c_float()
        i = C.c_uint32()
        if _ffi.load_library().sdfhost_node_nearest_neighbor(self._h, _ffi.fptr(p), C.byref(d), C.byref(i)) != 0:
            raise SdfError(_host_err())
        return float(np.float32(d.value)), int(i.value)

    def expanded(self):
        e = _ffi.load_library().sdfhost_expanded_new(self._h, _opts(self._opts))
        if not e:
            raise SdfError(_host_err())
        return ExpandedSdfNode(e, self._opts)

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.load_library().sdfhost_node_free(self._h)
            self._h = None


class SdfBuilder:
    """node.rs:371-414.  Every method consumes the builder it is called on."""

    def __init__(self, handle, opts):
        self._h, self._opts = handle, opts

    def _take(self):
        if not self._h:
            raise SdfError("builder already consumed")
        h, self._h = self._h, None
        return h

    @staticmethod
    def primitive(prim, opts=None):
        lib = _ffi.load_library()
        if isinstance(prim, SdfSphere):
            h = lib.sdfhost_primitive_sphere(prim.radius, _opts(opts))
        elif isinstance(prim, SdfBox):
            h = lib.sdfhost_primitive_box(*prim.half, _opts(opts))
        elif isinstance(prim, SdfBoxFrame):
            h = lib.sdfhost_primitive_box_frame(*prim.dimension, prim.thickness, _opts(opts))
        else:
            raise SdfError("not a primitive")  # node.rs:377 assert
        if not h:
            raise SdfError(_host_err())
        return SdfBuilder(h, opts)

    def operation(self, op):
        if isinstance(op, SdfCaaClone):
            h = _ffi.load_library().sdfhost_operation_caa_clone(
                self._take(), _ffi.fptr(op.displacement), _ffi.fptr(op.neg_limit), _ffi.fptr(op.pos_limit))
        elif isinstance(op, SdfUnion):
            h = _ffi.load_library().sdfhost_operation_union(self._take(), op.smooth_radius)
        else:
            raise SdfError("unsupported operation")
        if not h:
            raise SdfError(_host_err())
        return SdfBuilder(h, self._opts)

    def with_(self, node):
        h = _ffi.load_library().sdfhost_with(self._take(), node._take())
        if not h:
            raise SdfError(_host_err())
        return SdfBuilder(h, self._opts)

    def transform(self, t):
        h = _ffi.load_library().sdfhost_transform(
            self._take(), _ffi.fptr(t.translation), _ffi.fptr(t.rotation.xyzw), _ffi.fptr(t.scale)
        )
        if not h:
            raise SdfError(_host_err())
        return SdfBuilder(h, self._opts)

    def finalize(self):
        h = _ffi.load_library().sdfhost_finalize(self._take())
        if not h:
            raise SdfError(_host_err())
        return SdfNode(h, self._opts)

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.load_library().sdfhost_builder_free(self._h)
            self._h = None


def default_extent(n):
    return float(_ffi.load_library().sdfhost_default_extent(n))


def random_scene(n, seed, extent=0.0, opts=None):
    """Synthetic scene of SURVEY 8d: one flat SdfUnion over n random
    sphere/box primitives (recipe of node.rs:471-491 / src/main.rs:13-31)."""
    h = _ffi.load_library().sdfhost_random_scene(n, seed, float(extent), _opts(opts))
    if not h:
        raise SdfError(_host_err())
    return SdfNode(h, opts)


def random_scene_buffer(n, seed, extent=0.0, opts=None):
    return random_scene(n, seed, extent, opts).expanded().make_buffer()
