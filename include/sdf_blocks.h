/*
 * sdf_blocks.h -- C mirror of the reference crate's `#[repr(C)]` node records.
 *
 * This is the data format of the drop-in boundary: a Rust caller passes
 * `SdfTreeBuffer::downtree_buffer.as_ptr()` / `uptree_buffer.as_ptr()` and
 * `buffer_len`; this header describes those bytes exactly.
 *
 * Reference: sdf/src/component.rs:11-17  (SdfOpSpecificBlock)
 *            sdf/src/component.rs:27-38  (SdfOperationBlock)
 *            sdf/src/component.rs:53-60  (SdfOperationUptreeBlock)
 *            sdf/src/component.rs:62-69  (SdfBoundingBoxBlock)
 *            sdf/src/component.rs:80-86  (SdfTreeBuffer, not repr(C): passed as
 *                                         two pointers + a length)
 *
 * Layout assumes glam built with SSE2 (the bevy default on x86-64): Vec4 and
 * Mat4 are 16-byte aligned, Mat4 is column-major (translation in floats
 * 12..14).  Sizes: 192 / 208 / 624 / 224 bytes.  A producer built with glam's
 * `scalar-math` feature has 4-byte alignment and different sizes; such
 * buffers are rejected by sdfgpu_upload_tree (layout enum in sdfgpu.h).
 */
#ifndef SDF_BLOCKS_H
#define SDF_BLOCKS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define SDF_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define SDF_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
/* 16-byte alignment of the TYPE (C11 _Alignas cannot be put on a struct tag) */
#define SDF_ALIGN16 __attribute__((aligned(16)))

#ifdef __cplusplus
extern "C" {
#endif

/* glam::Vec4 (SSE2): 4 x f32, align 16. */
typedef struct SdfVec4 {
    float v[4]; /* x y z w */
} SDF_ALIGN16 SdfVec4;

/* glam::Mat4 (SSE2): 4 columns of Vec4, column-major: m[4*col + row]. */
typedef struct SdfMat4 {
    float m[16];
} SDF_ALIGN16 SdfMat4;

/* component.rs:11-17 */
typedef struct SdfOpSpecificBlock {
    SdfMat4 mat4s[2];  /* @0   */
    SdfVec4 vec4s[3];  /* @128 */
    float floats[2];   /* @176 */
    /* 8 bytes tail padding -> 192 */
} SdfOpSpecificBlock;

/* component.rs:62-69 */
typedef struct SdfBoundingBoxBlock {
    SdfMat4 matrix;        /* @0   unit cube -> frame of the tested point */
    SdfVec4 scale;         /* @64  half extents, w = 0 */
    SdfMat4 full_inverse;  /* @80  matrix^-1 */
    SdfMat4 trans_inverse; /* @144 un-scaled inverse (point re-basing) */
} SdfBoundingBoxBlock;

/* component.rs:27-38 -- one per node, pre-order ("downtree") */
typedef struct SdfOperationBlock {
    uint32_t op_code;           /* @0  */
    uint8_t is_primitive;       /* @4  Rust bool: 0 or 1 */
    uint8_t parent_is_union;    /* @5  */
    uint8_t _pad0[2];           /* @6  */
    uint32_t len;               /* @8  number of descendants in pre-order */
    uint32_t level;             /* @12 root = 1 */
    SdfOpSpecificBlock op_specific;   /* @16  */
    SdfBoundingBoxBlock bounding_box; /* @208 */
    SdfBoundingBoxBlock other_box;    /* @416 */
} SdfOperationBlock;

/* component.rs:53-60 -- one per node, post-order ("uptree") */
typedef struct SdfOperationUptreeBlock {
    uint32_t op_code;           /* @0 */
    uint8_t parent_is_union;    /* @4 */
    uint8_t _pad0[11];          /* @5 */
    SdfOpSpecificBlock op_specific; /* @16 */
    uint32_t level;             /* @208 */
    uint8_t _pad1[12];          /* @212 */
} SdfOperationUptreeBlock;

SDF_STATIC_ASSERT(sizeof(SdfVec4) == 16, "Vec4 size");
SDF_STATIC_ASSERT(sizeof(SdfMat4) == 64, "Mat4 size");
SDF_STATIC_ASSERT(sizeof(SdfOpSpecificBlock) == 192, "SdfOpSpecificBlock size");
SDF_STATIC_ASSERT(offsetof(SdfOpSpecificBlock, vec4s) == 128, "vec4s offset");
SDF_STATIC_ASSERT(offsetof(SdfOpSpecificBlock, floats) == 176, "floats offset");
SDF_STATIC_ASSERT(sizeof(SdfBoundingBoxBlock) == 208, "SdfBoundingBoxBlock size");
SDF_STATIC_ASSERT(offsetof(SdfBoundingBoxBlock, scale) == 64, "scale offset");
SDF_STATIC_ASSERT(offsetof(SdfBoundingBoxBlock, full_inverse) == 80, "full_inverse offset");
SDF_STATIC_ASSERT(offsetof(SdfBoundingBoxBlock, trans_inverse) == 144, "trans_inverse offset");
SDF_STATIC_ASSERT(sizeof(SdfOperationBlock) == 624, "SdfOperationBlock size");
SDF_STATIC_ASSERT(offsetof(SdfOperationBlock, is_primitive) == 4, "is_primitive offset");
SDF_STATIC_ASSERT(offsetof(SdfOperationBlock, parent_is_union) == 5, "parent_is_union offset");
SDF_STATIC_ASSERT(offsetof(SdfOperationBlock, len) == 8, "len offset");
SDF_STATIC_ASSERT(offsetof(SdfOperationBlock, level) == 12, "level offset");
SDF_STATIC_ASSERT(offsetof(SdfOperationBlock, op_specific) == 16, "op_specific offset");
SDF_STATIC_ASSERT(offsetof(SdfOperationBlock, bounding_box) == 208, "bounding_box offset");
SDF_STATIC_ASSERT(offsetof(SdfOperationBlock, other_box) == 416, "other_box offset");
SDF_STATIC_ASSERT(sizeof(SdfOperationUptreeBlock) == 224, "SdfOperationUptreeBlock size");
SDF_STATIC_ASSERT(offsetof(SdfOperationUptreeBlock, op_specific) == 16, "ut op_specific offset");
SDF_STATIC_ASSERT(offsetof(SdfOperationUptreeBlock, level) == 208, "ut level offset");

/* Operation / primitive codes.  The crate keeps two colliding namespaces
 * (elements.rs:92,164,224); which one applies is decided by is_primitive. */
enum {
    SDF_PRIM_SPHERE = 0,   /* elements.rs:85-118: floats[0] = radius */
    SDF_PRIM_BOX = 2,      /* build-defined (SURVEY C-7): vec4s[0] = half extents, w = 0 */
    SDF_PRIM_BOX_FRAME = 5,/* elements.rs:120-152 (private, no distance formula in the crate): vec4s[0] = dimension,
                              floats[0] = thickness; evaluated only under SDFGPU_EXT_BOX_FRAME */
    SDF_OP_UNION = 0,      /* elements.rs:157-212 */
    SDF_OP_CAA_CLONE = 1   /* elements.rs:215-262; a one-slot clone subtree is +inf in the flat query as written (faux_shader.rs:90) */
};

#define SDF_NO_PRIM 0xFFFFFFFFu

#ifdef __cplusplus
}
#endif
#endif /* SDF_BLOCKS_H */
