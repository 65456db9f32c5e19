/*
 * sdfhost.h -- flat C view of the host-side scene API (sdf_amd/csrc/host/),
 * i.e. of the crate's SdfBuilder / SdfNode / ExpandedSdfNode chain:
 *
 *   SdfBuilder::primitive(..).transform(..).operation(SdfUnion{..}).with(..)
 *       .finalize().expanded().make_buffer()          (sdf/src/node.rs:371-414,
 *                                                      :209-211, :75-138)
 *
 * Handles are opaque; every function returns 0 / a valid handle on success.
 * On failure functions return a negative code or NULL and
 * sdfhost_last_error() (thread-local) has the message the crate would have
 * panicked with.  Used by the Python mirror in sdf_amd/scene.py, by tests and
 * by bench.py to build scenes; the result of make_buffer is the byte-exact
 * input of sdfgpu_upload_tree.
 */
#ifndef SDFHOST_H
#define SDFHOST_H

#include <stddef.h>
#include <stdint.h>
#include "sdf_blocks.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdfhost_builder sdfhost_builder;   /* SdfBuilder        node.rs:371 */
typedef struct sdfhost_node sdfhost_node;         /* SdfNode           node.rs:141 */
typedef struct sdfhost_expanded sdfhost_expanded; /* ExpandedSdfNode   node.rs:20  */
typedef struct sdfhost_buffer sdfhost_buffer;     /* SdfTreeBuffer     component.rs:82 */

/* Build options = recorded deviations from the crate. */
typedef struct sdfhost_options {
    uint32_t pca_trans_inverse;  /* 1: merged boxes carry the crate's literal trans_inverse (C-3) */
    uint32_t sibling_other_box;  /* 1: other_box = sibling's box instead of own (C-2) */
    float merge_pad_rel;         /* outward padding of merged boxes; <0 selects the default 1e-5 */
    uint32_t reserved;
} sdfhost_options;

const char* sdfhost_last_error(void);

/* SdfBuilder::primitive(SdfSphere{radius}) -- node.rs:383, elements.rs:85 */
sdfhost_builder* sdfhost_primitive_sphere(float radius, const sdfhost_options* opts);
/* SdfBuilder::primitive(SdfBox{half}) -- build-defined solid box, op 2 */
sdfhost_builder* sdfhost_primitive_box(float hx, float hy, float hz, const sdfhost_options* opts);
/* SdfBoxFrame (elements.rs:120-152; private and unfinished in the crate: op 5, dimension + thickness). */
sdfhost_builder* sdfhost_primitive_box_frame(float dx, float dy, float dz, float thickness, const sdfhost_options* opts);
/* .operation(SdfUnion{smooth_radius}) -- node.rs:396, elements.rs:157; consumes b */
sdfhost_builder* sdfhost_operation_union(sdfhost_builder* b, float smooth_radius);
/* SdfBuilder::operation(SdfCaaClone { displacement, neg_limit, pos_limit }) (elements.rs:215-262).
 * The flat query evaluates a clone subtree to +inf as written (faux_shader.rs:90); see DESIGN.md. */
sdfhost_builder* sdfhost_operation_caa_clone(sdfhost_builder* b, const float displacement[3], const float neg_limit[3],
                                             const float pos_limit[3]);
/* .with(child) -- node.rs:400; consumes both, returns b */
sdfhost_builder* sdfhost_with(sdfhost_builder* b, sdfhost_builder* child);
/* .transform(Transform{translation, rotation(x,y,z,w), scale}) -- node.rs:405; consumes b */
sdfhost_builder* sdfhost_transform(sdfhost_builder* b, const float translation[3],
                                   const float rotation_xyzw[4], const float scale[3]);
/* .finalize() -- node.rs:410; consumes b */
sdfhost_node* sdfhost_finalize(sdfhost_builder* b);
void sdfhost_builder_free(sdfhost_builder* b);

/* Quat::from_euler(EulerRot::XYZ, a, b, c) as the crate's tests use it (node.rs:482) */
void sdfhost_quat_from_euler_xyz(float a, float b, float c, float out_xyzw[4]);

/* Synthetic scene of SURVEY 8d (one flat union over n random sphere/box
 * primitives); extent <= 0 selects 50 (n/1000)^(1/3). */
sdfhost_node* sdfhost_random_scene(uint32_t n, uint64_t seed, float extent,
                                   const sdfhost_options* opts);
float sdfhost_default_extent(uint32_t n);

/* SdfNode accessors */
uint32_t sdfhost_node_num_slots(const sdfhost_node* n);
int sdfhost_node_bbox(const sdfhost_node* n, SdfBoundingBoxBlock* out);   /* node.bbox.unwrap().get_bbox_block() */
/* SdfNode::nearest_neighbor (node.rs:331-368): the crate's recursive n-ary branch-and-prune
 * query on the scene tree itself (host code there and here).  out_prim (optional) receives the
 * pre-order ordinal of the returned primitive among the tree's primitives, SDF_NO_PRIM when the
 * query returned no primitive (distance +inf).  0 on success, -1 with sdfhost_last_error. */
int sdfhost_node_nearest_neighbor(const sdfhost_node* n, const float p[3], float* out_dist, uint32_t* out_prim);
void sdfhost_node_free(sdfhost_node* n);

/* SdfNode::expanded() -- node.rs:209 */
sdfhost_expanded* sdfhost_expanded_new(const sdfhost_node* n, const sdfhost_options* opts);
void sdfhost_expanded_free(sdfhost_expanded* e);

/* ExpandedSdfNode::make_buffer() -- node.rs:75 */
sdfhost_buffer* sdfhost_make_buffer(const sdfhost_expanded* e, const sdfhost_options* opts);
uint32_t sdfhost_buffer_len(const sdfhost_buffer* b);
const SdfOperationBlock* sdfhost_buffer_downtree(const sdfhost_buffer* b);
const SdfOperationUptreeBlock* sdfhost_buffer_uptree(const sdfhost_buffer* b);
void sdfhost_buffer_free(sdfhost_buffer* b);

/* SdfBoundingBox::distance_to / max_distance (obb.rs:294-307) on a block */
float sdfhost_bbox_distance_to(const SdfBoundingBoxBlock* b, const float p[3]);
float sdfhost_bbox_max_distance(const SdfBoundingBoxBlock* b, const float p[3]);

#ifdef __cplusplus
}
#endif
#endif /* SDFHOST_H */
