"""sdf_amd -- MI355X-native nearest-primitive / sphere-trace path of the sdf crate.

Host side mirrors the crate's scene API (scene.py over sdf_amd/csrc/host);
the queries run as hand-written gfx950 kernels behind the C ABI of
include/sdfgpu.h (gpu.py).  Importing the package loads libsdfgpu.so and fails
loudly if it has not been built.
"""
from . import _ffi
from ._ffi import (BOUNDS_AABB, BOUNDS_AUTO, BOUNDS_BVH, BOUNDS_NONE, BOUNDS_OBB, EXT_BOX_FRAME,
                   EXT_CLONE_REPAIR, EXT_SMOOTH_UNION, NO_PRIM,
                   WALK_CHAIN, WALK_GENERAL_EXHAUSTIVE, WALK_GENERAL_PRUNED, WALK_HIERARCHY,
                   Camera, MarchParams, load_library)
from .gpu import (SdfComm, SdfGpu, SdfGpuError, camera_look_at, compact_host, knn_sharded, march_params, nearest_sharded,
                  query_slice)
from .scene import (ExpandedSdfNode, Quat, SdfBox, SdfBoxFrame, SdfBuilder, SdfCaaClone, SdfError, SdfNode,
                    SdfSphere, SdfTreeBuffer, SdfUnion, Transform, default_extent,
                    random_scene, random_scene_buffer)

load_library()
