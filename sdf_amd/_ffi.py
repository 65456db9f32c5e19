"""ctypes binding of libsdfgpu.so (include/sdfgpu.h + include/sdfhost.h).

The library is the product; there is no Python or CPU fallback.  If the shared
object is missing this module raises at import of the symbol table.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdfgpu.so")

# ---- record dtypes: byte-exact mirrors of include/sdf_blocks.h -------------
f4, u4, u1 = np.float32, np.uint32, np.uint8
OPSPEC_DTYPE = np.dtype(
    [("mat4s", f4, (2, 16)), ("vec4s", f4, (3, 4)), ("floats", f4, (2,)), ("_pad", u1, (8,))]
)
BBOX_DTYPE = np.dtype(
    [("matrix", f4, (16,)), ("scale", f4, (4,)), ("full_inverse", f4, (16,)), ("trans_inverse", f4, (16,))]
)
DT_DTYPE = np.dtype(
    [
        ("op_code", u4),
        ("is_primitive", u1),
        ("parent_is_union", u1),
        ("_pad0", u1, (2,)),
        ("len", u4),
        ("level", u4),
        ("op_specific", OPSPEC_DTYPE),
        ("bounding_box", BBOX_DTYPE),
        ("other_box", BBOX_DTYPE),
    ]
)
UT_DTYPE = np.dtype(
    [
        ("op_code", u4),
        ("parent_is_union", u1),
        ("_pad0", u1, (11,)),
        ("op_specific", OPSPEC_DTYPE),
        ("level", u4),
        ("_pad1", u1, (12,)),
    ]
)
assert OPSPEC_DTYPE.itemsize == 192 and BBOX_DTYPE.itemsize == 208
assert DT_DTYPE.itemsize == 624 and UT_DTYPE.itemsize == 224

PRIM_SPHERE, PRIM_BOX, PRIM_BOX_FRAME, OP_UNION, OP_CAA_CLONE = 0, 2, 5, 0, 1
NO_PRIM = 0xFFFFFFFF

BOUNDS_AUTO, BOUNDS_AABB, BOUNDS_OBB, BOUNDS_NONE, BOUNDS_BVH = 0, 1, 2, 3, 4
LAYOUT_GLAM_SSE2 = 0
EXT_SMOOTH_UNION, EXT_BOX_FRAME, EXT_CLONE_REPAIR = 1, 2, 4
WALK_HIERARCHY, WALK_CHAIN, WALK_GENERAL_PRUNED, WALK_GENERAL_EXHAUSTIVE = 0, 1, 2, 3

OK = 0
ERR_INVALID_ARG, ERR_HIP, ERR_NO_TREE, ERR_LAYOUT, ERR_BAD_TREE = -1, -2, -3, -4, -5
ERR_UNSUPPORTED_OP, ERR_TOO_DEEP, ERR_NON_FINITE, ERR_POINT_W, ERR_TOO_LARGE, ERR_COMM = -6, -7, -8, -9, -10, -11
COMM_ID_BYTES = 128
MAX_K = 16
STATIC_MAX_POINTS = 262144
OPT_REORDER, OPT_SKIP_EXTENSION, OPT_COMPACTION, OPT_TILE_ORDER, OPT_STATS_ASM_WALK, OPT_QUERY_PIPELINE, OPT_CANDIDATE_GRID = 0, 1, 2, 3, 4, 5, 6


class TreeInfo(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32),
        ("n_prims", C.c_uint32),
        ("n_inner", C.c_uint32),
        ("max_level", C.c_uint32),
        ("bounds", C.c_uint32),
        ("identity_inner", C.c_uint32),
        ("rigid_leaves", C.c_uint32),
        ("stack_depth", C.c_uint32),
        ("device_bytes", C.c_uint64),
        ("scene_lo", C.c_float * 3),
        ("scene_hi", C.c_float * 3),
        ("extensions", C.c_uint32),
        ("grid_cells", C.c_uint32),
        ("walk", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("eye", C.c_float * 3),
        ("origin", C.c_float * 3),
        ("du", C.c_float * 3),
        ("dv", C.c_float * 3),
    ]


class MarchParams(C.Structure):
    _fields_ = [
        ("max_steps", C.c_uint32),
        ("eps_hit", C.c_float),
        ("t_min", C.c_float),
        ("t_max", C.c_float),
        ("normal_h", C.c_float),
        ("flags", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("queries", C.c_uint64),
        ("inner_visits", C.c_uint64),
        ("sphere_evals", C.c_uint64),
        ("box_evals", C.c_uint64),
        ("wave_inner", C.c_uint64),
        ("wave_leaf", C.c_uint64),
        ("steps", C.c_uint64),
        ("rays", C.c_uint64),
        ("wave_steps", C.c_uint64),
        ("wave_walks", C.c_uint64),
        ("inner_needed", C.c_uint64),
        ("sphere_needed", C.c_uint64),
        ("box_needed", C.c_uint64),
        ("list_steps", C.c_uint64),
        ("wave_lists", C.c_uint64),
        ("wave_list_evals", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class HostOptions(C.Structure):
    _fields_ = [
        ("pca_trans_inverse", C.c_uint32),
        ("sibling_other_box", C.c_uint32),
        ("merge_pad_rel", C.c_float),
        ("reserved", C.c_uint32),
    ]


_vp, _u32, _u64, _f, _sz, _i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t, C.c_int
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/*.h declares is listed here
GPU_SYMBOLS = {
    "sdfgpu_create": (_i, [_i, C.POINTER(_vp)]),
    "sdfgpu_destroy": (None, [_vp]),
    "sdfgpu_last_error": (C.c_char_p, [_vp]),
    "sdfgpu_status_string": (C.c_char_p, [_i]),
    "sdfgpu_sync": (_i, [_vp]),
    "sdfgpu_upload_tree": (_i, [_vp, _vp, _vp, _u32, _u32, _u32]),
    "sdfgpu_compact_host": (_i, [_vp, _vp, _u32, _u32, _u32, C.POINTER(TreeInfo), C.c_char_p, _sz]),
    "sdfgpu_upload_tree_ext": (_i, [_vp, _vp, _vp, _u32, _u32, _u32, _u32]),
    "sdfgpu_compact_host_ext": (_i, [_vp, _vp, _u32, _u32, _u32, _u32, C.POINTER(TreeInfo), C.c_char_p, _sz]),
    "sdfgpu_tree_info_get": (_i, [_vp, C.POINTER(TreeInfo)]),
    "sdfgpu_nearest": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "sdfgpu_nearest_dev": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "sdfgpu_nearest_dev_static": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "sdfgpu_set_option": (_i, [_vp, _i, C.c_double]),
    "sdfgpu_knn": (_i, [_vp, _vp, _sz, _u32, _vp, _vp]),
    "sdfgpu_knn_dev": (_i, [_vp, _vp, _sz, _u32, _vp, _vp, _vp]),
    "sdfgpu_box_bounds": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "sdfgpu_render": (_i, [_vp, C.POINTER(Camera), _u32, _u32, C.POINTER(MarchParams), _vp, _vp, _vp, _vp]),
    "sdfgpu_render_dev": (_i, [_vp, C.POINTER(Camera), _u32, _u32, C.POINTER(MarchParams), _vp, _vp, _vp, _vp, _vp]),
    "sdfgpu_render_tiles_dev": (
        _i,
        [_vp, C.POINTER(Camera), _u32, _u32, C.POINTER(MarchParams), _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp],
    ),
    "sdfgpu_untile_dev": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdfgpu_packed_share_bytes": (C.c_size_t, [_u32, _u32, _u32]),
    "sdfgpu_untile_gathered_dev": (_i, [_vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdfgpu_comm_get_id": (_i, [_vp, _vp]),
    "sdfgpu_comm_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "sdfgpu_comm_destroy": (None, [_vp]),
    "sdfgpu_render_frame_comm": (_i, [_vp, _vp, C.POINTER(Camera), _u32, _u32, C.POINTER(MarchParams), _i, _vp, _vp, _vp, _vp, _vp]),
    "sdfgpu_comm_create_all": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp)]),
    "sdfgpu_render_frame_all": (_i, [C.POINTER(_vp), _i, C.POINTER(Camera), _u32, _u32, C.POINTER(MarchParams), _i, _vp, _vp, _vp, _vp,
                                     C.POINTER(_vp)]),
    "sdfgpu_query_slice": (None, [_sz, _u32, _u32, C.POINTER(_sz), C.POINTER(_sz)]),
    "sdfgpu_nearest_sharded": (_i, [C.POINTER(_vp), _i, _vp, _sz, _vp, _vp]),
    "sdfgpu_knn_sharded": (_i, [C.POINTER(_vp), _i, _vp, _sz, _u32, _vp, _vp]),
    "sdfgpu_render_stats": (_i, [_vp, C.POINTER(Camera), _u32, _u32, C.POINTER(MarchParams), C.POINTER(Stats)]),
    "sdfgpu_nearest_stats_dev": (_i, [_vp, _vp, _sz, C.POINTER(Stats)]),
    "sdfgpu_knn_stats_dev": (_i, [_vp, _vp, _sz, _u32, C.POINTER(Stats)]),
    "sdfgpu_camera_look_at": (None, [_fp, _fp, _fp, _f, _u32, _u32, C.POINTER(Camera)]),
}

HOST_SYMBOLS = {
    "sdfhost_last_error": (C.c_char_p, []),
    "sdfhost_primitive_sphere": (_vp, [_f, C.POINTER(HostOptions)]),
    "sdfhost_primitive_box": (_vp, [_f, _f, _f, C.POINTER(HostOptions)]),
    "sdfhost_primitive_box_frame": (_vp, [_f, _f, _f, _f, C.POINTER(HostOptions)]),
    "sdfhost_operation_union": (_vp, [_vp, _f]),
    "sdfhost_operation_caa_clone": (_vp, [_vp, _fp, _fp, _fp]),
    "sdfhost_with": (_vp, [_vp, _vp]),
    "sdfhost_transform": (_vp, [_vp, _fp, _fp, _fp]),
    "sdfhost_finalize": (_vp, [_vp]),
    "sdfhost_builder_free": (None, [_vp]),
    "sdfhost_quat_from_euler_xyz": (None, [_f, _f, _f, _fp]),
    "sdfhost_random_scene": (_vp, [_u32, _u64, _f, C.POINTER(HostOptions)]),
    "sdfhost_default_extent": (_f, [_u32]),
    "sdfhost_node_num_slots": (_u32, [_vp]),
    "sdfhost_node_bbox": (_i, [_vp, _vp]),
    "sdfhost_node_nearest_neighbor": (_i, [_vp, _fp, _fp, C.POINTER(_u32)]),
    "sdfhost_node_free": (None, [_vp]),
    "sdfhost_expanded_new": (_vp, [_vp, C.POINTER(HostOptions)]),
    "sdfhost_expanded_free": (None, [_vp]),
    "sdfhost_make_buffer": (_vp, [_vp, C.POINTER(HostOptions)]),
    "sdfhost_buffer_len": (_u32, [_vp]),
    "sdfhost_buffer_downtree": (_vp, [_vp]),
    "sdfhost_buffer_uptree": (_vp, [_vp]),
    "sdfhost_buffer_free": (None, [_vp]),
    "sdfhost_bbox_distance_to": (_f, [_vp, _fp]),
    "sdfhost_bbox_max_distance": (_f, [_vp, _fp]),
}

_lib = None


def load_library():
    """Load libsdfgpu.so and type every exported symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C sdf_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no fallback path."
        )
    # PyTorch-ROCm ships its own libamdhip64.so.7.  Two HIP runtimes in one
    # process cannot both own the GPU ("No HIP GPUs are available" from whichever
    # initialises second), so when torch is installed it is imported first and
    # libsdfgpu.so binds to the runtime torch already loaded (same soname).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for table in (GPU_SYMBOLS, HOST_SYMBOLS):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def fptr(a):
    return a.ctypes.data_as(_fp)
