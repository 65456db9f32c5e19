/*
 * sdfgpu.h -- C ABI of the MI355X nearest-primitive / sphere-trace path.
 *
 * The reference crate has no FFI; the seam its author designed as the GPU
 * hand-off is
 *     faux_shader::nearest_neighbor(sdf_tree: &SdfTreeBuffer, point: Vec4) -> f32
 *     (sdf/src/faux_shader.rs:96)
 * over the #[repr(C)] records of sdf/src/component.rs:11-78 (see sdf_blocks.h).
 * Every entry point below cites the reference item it replaces.  A Rust
 * `extern "C"` block binding exactly these symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *  - one context per GPU, one process per GPU (multi-GPU = one process per
 *    device, partitioned by tile / query range; see sdfgpu_render_tiles_dev);
 *  - every function returns 0 on success or a negative sdfgpu_status; the
 *    reference's panic!/assert! sites become error codes, nothing unwinds;
 *  - host-pointer entry points copy in, run, copy out and synchronise;
 *    `_dev` entry points take device pointers and a hipStream_t (as void*) and
 *    are asynchronous with respect to the host;
 *  - the caller owns every buffer it passes; the context owns the uploaded
 *    tree and its own scratch.  A context is not thread-safe.
 */
#ifndef SDFGPU_H
#define SDFGPU_H

#include <stddef.h>
#include <stdint.h>
#include "sdf_blocks.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdfgpu_ctx sdfgpu_ctx;

typedef enum sdfgpu_status {
    SDFGPU_OK = 0,
    SDFGPU_ERR_INVALID_ARG = -1,
    SDFGPU_ERR_HIP = -2,             /* a HIP runtime call failed (message has the detail) */
    SDFGPU_ERR_NO_TREE = -3,         /* query before sdfgpu_upload_tree */
    SDFGPU_ERR_LAYOUT = -4,          /* record stride / layout not the SSE2 one */
    SDFGPU_ERR_BAD_TREE = -5,        /* len/level/shape invariants of make_buffer violated */
    SDFGPU_ERR_UNSUPPORTED_OP = -6,  /* faux_shader.rs:65,80,92 panic! sites */
    SDFGPU_ERR_TOO_DEEP = -7,        /* level >= 255 (faux_shader.rs:100 stack of 256) */
    SDFGPU_ERR_NON_FINITE = -8,      /* NaN/Inf in a matrix the path evaluates (SURVEY C-10) */
    SDFGPU_ERR_POINT_W = -9,         /* a host query point has w != 1 */
    SDFGPU_ERR_TOO_LARGE = -10,      /* k > SDFGPU_MAX_K, image too large, ... */
    SDFGPU_ERR_COMM = -11            /* RCCL missing or a collective call failed (message has the detail) */
} sdfgpu_status;

/* Record layouts a producer may have (SURVEY 8b "layout caveat"). */
typedef enum sdfgpu_layout {
    SDFGPU_LAYOUT_GLAM_SSE2 = 0      /* 624-byte down blocks, 224-byte up blocks */
} sdfgpu_layout;

/* How the kernels bound subtrees.  All modes return the same (distance, index)
 * on geometrically consistent buffers; see DESIGN.md "bounds". */
typedef enum sdfgpu_bounds {
    SDFGPU_BOUNDS_AUTO = 0,   /* BVH if the tree qualifies, else NONE */
    SDFGPU_BOUNDS_AABB = 1,   /* world AABBs recomputed bottom-up from the leaves at upload */
    SDFGPU_BOUNDS_OBB = 2,    /* the buffer's own boxes: MINDIST of faux_shader.rs:54-58 as the pruning bound, with a margin */
    SDFGPU_BOUNDS_NONE = 3,   /* no pruning: every primitive is evaluated (always valid) */
    SDFGPU_BOUNDS_BVH = 4     /* world AABBs over a hierarchy the library builds itself over the primitives
                                 (binned SAH); the buffer's topology only fixes the primitive order */
} sdfgpu_bounds;

#define SDFGPU_MAX_K 16u

typedef struct sdfgpu_tree_info {
    uint32_t n_nodes;        /* = buffer_len */
    uint32_t n_prims;
    uint32_t n_inner;
    uint32_t max_level;      /* deepest `level` (root = 1) */
    uint32_t bounds;         /* sdfgpu_bounds actually in use */
    uint32_t identity_inner; /* 1 if every internal trans_inverse is exactly I */
    uint32_t rigid_leaves;   /* 1 if every leaf trans_inverse is a rigid motion (tol 1e-4) */
    uint32_t stack_depth;     /* levels of the resident hierarchy = entries of the per-wavefront stack */
    uint64_t device_bytes;   /* compact tree bytes resident in HBM */
    float scene_lo[3];       /* root AABB (AABB mode) */
    float scene_hi[3];
    uint32_t extensions;     /* sdfgpu_extension bits that are in effect in this tree (0 = evaluated as the crate is written) */
    uint32_t grid_cells;     /* cells of the nearest-candidate grid (0 = none): march steps and k = 1 queries inside it are decided on
                                a cell's list, k-nearest queries start from its radius cap; its bytes are part of device_bytes */
    uint32_t walk;           /* sdfgpu_walk: which resident form serves this tree's queries */
    uint32_t reserved;
} sdfgpu_tree_info;

/* How a tree is walked (sdfgpu_tree_info.walk).  Results do not depend on it; cost does. */
typedef enum sdfgpu_walk {
    SDFGPU_WALK_HIERARCHY = 0,          /* pruned walk over a box hierarchy (`bounds` says whose), or every primitive (NONE) */
    SDFGPU_WALK_CHAIN = 1,              /* the same over the library's hierarchy, leaf values through the per-level point chain
                                           (buffers whose internal nodes re-base the point with rigid maps: the crate's own) */
    SDFGPU_WALK_GENERAL_PRUNED = 2,     /* the buffer's topology with the post-order combine (extensions in effect, or as below
                                           but rigid), subtrees skipped on library-derived boxes */
    SDFGPU_WALK_GENERAL_EXHAUSTIVE = 3  /* the same without skipping: non-rigid or projective maps, small trees, or smooth radii
                                           of the order of the scene -- O(nodes) per query */
} sdfgpu_walk;

/* Pinhole camera as plain data.  Ray through pixel (x, y):
 *   pw  = (origin + du * x) + dv * y        (per component, each op rounded)
 *   dir = (pw - eye) / |pw - eye|
 * mirrors the only ray the reference defines (src/main__.rs:96-97:
 * normalize(world_position - camera_position)). */
typedef struct sdfgpu_camera {
    float eye[3];
    float origin[3];  /* world position of the centre of pixel (0,0) */
    float du[3];      /* world step per pixel along +x */
    float dv[3];      /* world step per pixel along +y */
} sdfgpu_camera;

/* Sphere-trace parameters.  The reference has no march loop; the
 * loop is defined in DESIGN.md and restated by oracle/oracle.cpp::oracle_render.
 * normal_h is the `h` of calcSphereNormal (src/main__.rs:152, 0.0001 there).
 * Any parameter values are accepted (a step cap of 0, eps <= 0, t_max < t_min, a NaN far plane = none ...):
 * the loop does what it says for them.  Domain of the march: a ray is a miss outright when a coordinate
 * of the eye, or t_min, is beyond 2^60 in magnitude or not a number, or when its pixel has no direction
 * (image-plane point = eye); a ray ends as a miss when t leaves min(t_max, 2^60) (or becomes NaN).  All
 * points the loop evaluates are then inside the query domain below. */
typedef struct sdfgpu_march_params {
    uint32_t max_steps;
    float eps_hit;
    float t_min;
    float t_max;
    float normal_h;
    uint32_t flags;   /* reserved, 0 */
} sdfgpu_march_params;

/* Per-launch work counters (instrumented variant of the same kernels). */
typedef struct sdfgpu_stats {
    uint64_t queries;        /* tree walks driven, summed over lanes (render: steps not proven by kept bounds) */
    uint64_t inner_visits;   /* internal-node visits, summed over lanes */
    uint64_t sphere_evals;   /* primitive evaluations, summed over lanes */
    uint64_t box_evals;
    uint64_t wave_inner;     /* the same, counted once per wavefront */
    uint64_t wave_leaf;
    uint64_t steps;          /* march steps summed over rays */
    uint64_t rays;
    uint64_t wave_steps;     /* march steps per wavefront */
    uint64_t wave_walks;     /* ... of which needed a tree walk (the rest were proven by kept bounds) */
    uint64_t inner_needed;   /* internal-node visits a lane itself needed (<= inner_visits: the wave visits the union) */
    uint64_t sphere_needed;  /* primitive evaluations a lane itself needed: seeds, and leaves its own bound asked for */
    uint64_t box_needed;
    uint64_t list_steps;     /* march steps (summed over rays) decided on a grid cell's candidate list instead of a walk */
    uint64_t wave_lists;     /* list passes, counted once per wavefront */
    uint64_t wave_list_evals;/* primitive evaluations of those passes, counted once per wavefront */
} sdfgpu_stats;

/* ---- lifetime ---------------------------------------------------------- */
/* No reference counterpart (the crate is a CPU library). */
int sdfgpu_create(int device_id, sdfgpu_ctx** out);
void sdfgpu_destroy(sdfgpu_ctx* ctx);
const char* sdfgpu_last_error(const sdfgpu_ctx* ctx);
const char* sdfgpu_status_string(int status);
int sdfgpu_sync(sdfgpu_ctx* ctx);

/* ---- tree upload ------------------------------------------------------- */
/* Takes what ExpandedSdfNode::make_buffer returns (sdf/src/node.rs:75-138):
 * `dt` = downtree_buffer.as_ptr(), `ut` = uptree_buffer.as_ptr(),
 * `buffer_len` = SdfTreeBuffer::buffer_len (component.rs:80-86).  Validates
 * everything the kernels rely on, compacts the 624+224-byte records into the
 * resident node arrays and copies them to HBM.  Replaces the `&SdfTreeBuffer`
 * borrow of faux_shader.rs:96. */
int sdfgpu_upload_tree(sdfgpu_ctx* ctx, const SdfOperationBlock* dt,
                       const SdfOperationUptreeBlock* ut, uint32_t buffer_len,
                       uint32_t layout, uint32_t bounds);
/* Opt-in extensions (N-2 of SURVEY 8f).  The crate carries the DATA for all three and evaluates none:
 * smooth_radius is written into the union's up-block (elements.rs:207-211) and ignored by
 * uptree_dispatch (faux_shader.rs:87); SdfBoxFrame (op 5, elements.rs:120-152) has no distance formula
 * and prim_dispatch panics on it (faux_shader.rs:65); a one-slot clone evaluates to +inf because
 * uptree op 1 returns the branch it never filled (faux_shader.rs:90).  sdfgpu_upload_tree evaluates
 * buffers as written (extensions = 0).  With a bit set here:
 *   SMOOTH_UNION  a union with smooth_radius k > 0 combines as min(a,b) - h*h*k/4, h = max(k-|a-b|,0)/k
 *   BOX_FRAME     primitive op 5 is accepted: iq's box-frame distance of (dimension, thickness)
 *   CLONE_REPAIR  uptree op 1 returns the LEFT branch (the one the clone's slot filled), so the lattice
 *                 fold of downtree op 1 (faux_shader.rs:75-78) takes effect
 * A tree in which an enabled extension occurs takes the general path (reported as bounds NONE: the buffer's
 * topology with the reference's post-order combine).  Where every map of the tree is affine and rigid -- anything
 * the crate's builder emits -- its walks skip subtrees on boxes the library derives itself, bit-identical to the
 * exhaustive walk (from 256 resident nodes on, and unless the smooth radii are of the order of the scene);
 * otherwise it is exhaustive, O(nodes) per query.  Restated by oracle/oracle.cpp under oracle_set_extensions. */
typedef enum sdfgpu_extension {
    SDFGPU_EXT_SMOOTH_UNION = 1,
    SDFGPU_EXT_BOX_FRAME = 2,
    SDFGPU_EXT_CLONE_REPAIR = 4
} sdfgpu_extension;
int sdfgpu_upload_tree_ext(sdfgpu_ctx* ctx, const SdfOperationBlock* dt,
                           const SdfOperationUptreeBlock* ut, uint32_t buffer_len,
                           uint32_t layout, uint32_t bounds, uint32_t extensions);
int sdfgpu_compact_host_ext(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut,
                            uint32_t buffer_len, uint32_t layout, uint32_t bounds, uint32_t extensions,
                            sdfgpu_tree_info* info, char* errbuf, size_t errlen);
int sdfgpu_tree_info_get(const sdfgpu_ctx* ctx, sdfgpu_tree_info* out);
/* The host half of sdfgpu_upload_tree alone (validation + compaction, no GPU
 * touched): same status codes, message copied into errbuf.  For producers
 * that want to check a buffer before shipping it. */
int sdfgpu_compact_host(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut,
                        uint32_t buffer_len, uint32_t layout, uint32_t bounds,
                        sdfgpu_tree_info* info, char* errbuf, size_t errlen);

/* ---- nearest primitive ------------------------------------------------- */
/* faux_shader::nearest_neighbor (faux_shader.rs:96-208) for n points at once.
 * pts_xyzw: n x 4 floats, w must be 1 (the crate only ever passes
 * point.extend(1.0), node.rs:335,359).  out_dist[i] is the value the reference
 * returns; out_prim[i] is the pre-order ordinal of the primitive attaining it
 * (ties -> the later one, faux_shader.rs:27-29), SDF_NO_PRIM for an empty tree.
 * out_prim may be NULL.
 *
 * Domain of a query point (every query entry point, k = 1 and kNN): each coordinate finite and at most
 * 2^62 in magnitude.  A point outside it returns (NaN, SDF_NO_PRIM) in every slot, takes no part in
 * the tree walk and has no effect on the other points of the batch.  (The reference has no defined
 * result there: once a leaf value is NaN -- a NaN or infinite coordinate, inf - inf after an overflow --
 * every `a < b` is false, its union keeps whichever operand came last and its prune test never fires.)
 * Inside the domain results are the reference's, bit for bit; points further than ~1e4 scene extents
 * from the scene are legal but slow: float32 no longer separates the primitives' distances there, so
 * more and more of the tree has to be evaluated to resolve the (distance, later index) rule. */
int sdfgpu_nearest(sdfgpu_ctx* ctx, const float* pts_xyzw, size_t n,
                   float* out_dist, uint32_t* out_prim);
/* Device-pointer form; the w component is not read (taken as 1). */
int sdfgpu_nearest_dev(sdfgpu_ctx* ctx, const float* d_pts_xyzw, size_t n,
                       float* d_out_dist, uint32_t* d_out_prim, void* hip_stream);

/* Allocation-free variant of sdfgpu_nearest_dev.  NO COUNTERPART IN THE REFERENCE (the crate is a
 * CPU library): it exists for callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels and bit-identical
 * results as sdfgpu_nearest_dev; the Morton-reordering scratch (keys, permutation, radix-sort
 * storage) lives in a file-scope __device__ array of the library (8 MiB per GPU, held until the
 * process ends) whose device address is looked up once per device in sdfgpu_create.  The call
 * performs no hipMalloc / hipFree.  At most SDFGPU_STATIC_MAX_POINTS points per call
 * (SDFGPU_ERR_TOO_LARGE beyond).  Every call on a GPU -- from any context of the process -- shares
 * that GPU's one array: calls that overlap in time on one GPU race; serialise them on a stream. */
#define SDFGPU_STATIC_MAX_POINTS 262144u
int sdfgpu_nearest_dev_static(sdfgpu_ctx* ctx, const float* d_pts_xyzw, size_t n,
                              float* d_out_dist, uint32_t* d_out_prim, void* hip_stream);

/* Tuning knobs.  None of them can change a result, only the time it takes.
 * No reference counterpart. */
typedef enum sdfgpu_option {
    /* 1 (default): nearest/knn batches of >= 2048 points are processed in Morton
     * order (results are still written at the caller's index); 0: as given.  The
     * reordering scratch belongs to the context, so only one nearest/knn call
     * per context may be in flight at a time. */
    SDFGPU_OPT_REORDER = 0,
    /* render: how far (scene units) the tree walk looks beyond the nearest
     * distance so that following march steps can be proven without a walk.
     * 0 (default): no widening (steps are still proven when the bounds a walk
     * leaves behind happen to allow it); < 0: half the mean primitive size, which
     * pays off in dense scenes with long grazing rays. */
    SDFGPU_OPT_SKIP_EXTENSION = 1,
    /* render: 1 cuts the march into passes and re-packs the rays still
     * marching, 64 to a wavefront, between them (default schedule: two passes, the
     * first 40 steps long; SDFGPU_PASS_STEPS="a,b,.." in the environment of
     * sdfgpu_create sets another); 0 (default): one pass.  Trades the latency of one
     * frame for throughput with frames in flight; see DESIGN.md. */
    SDFGPU_OPT_COMPACTION = 2,
    /* render: 1 (default): the march records how many steps each 8x8 tile took, and the
     * next frame of the same size on the same stream starts its longest tiles first (the
     * few tiles that run all max_steps then overlap the bulk of the frame instead of
     * trailing it).  Frames of at least 1024 tiles, single-pass march only.  0: tiles in
     * raster order. */
    SDFGPU_OPT_TILE_ORDER = 3,
    /* instrumentation: 1: sdfgpu_nearest_stats_dev counts with the hand-scheduled walk's own visit
     * counters (queries, wave_inner, wave_leaf only) instead of the C++ state machine that fills
     * every counter; 0 (default).  The two must report the same visits: the production kernels run
     * the former, the roofline units come from the latter. */
    SDFGPU_OPT_STATS_ASM_WALK = 4,
    /* nearest / knn batches of >= 2^20 points with reordering on are cut into this many chunks (0 or 1,
     * the default: one piece): the Morton sort of chunk i + 1 runs on a helper stream of the context
     * beside the walk of chunk i.  Ordering with the caller's stream is kept with events; the call stays
     * asynchronous.  Off by default: on MI355X every extra chunk cost ~0.3 ms on the 10 M point batch
     * (2.15 ms in one piece, 2.46 / 3.02 / 4.22 ms in 2 / 4 / 8 chunks). */
    SDFGPU_OPT_QUERY_PIPELINE = 5,
    /* 1 (default): sdfgpu_upload_tree also builds the nearest-candidate grid (a uniform grid over the padded
     * scene box whose cells list every primitive that can be nearest anywhere in the cell, and bound the
     * distance of the 8th / 16th nearest; about 18 bytes per cell, ~400 cells per primitive, at most 32 M
     * cells): march steps and k = 1 queries inside it are decided on a cell's list instead of a tree walk,
     * k-nearest queries start with their prune radius capped.  0: no grid (set BEFORE the upload; a tree
     * already resident keeps what it has).  Built for pruned trees (AABB / BVH / OBB bounds, chain path
     * included) of at least 64 primitives.  Results are identical either way. */
    SDFGPU_OPT_CANDIDATE_GRID = 6
} sdfgpu_option;
int sdfgpu_set_option(sdfgpu_ctx* ctx, int option, double value);

/* k nearest primitives (extension of the k = 1 reference, README.md:7):
 * out_*[i*k + j], ascending by (distance, then larger index first), padded
 * with (+inf, SDF_NO_PRIM). */
int sdfgpu_knn(sdfgpu_ctx* ctx, const float* pts_xyzw, size_t n, uint32_t k,
               float* out_dist, uint32_t* out_prim);
int sdfgpu_knn_dev(sdfgpu_ctx* ctx, const float* d_pts_xyzw, size_t n, uint32_t k,
                   float* d_out_dist, uint32_t* d_out_prim, void* hip_stream);

/* The two bounds of the reference's union prune for one box at n host points:
 * out_min = mindist (faux_shader.rs:54-58 = SdfBoundingBox::distance_to,
 * obb.rs:294-300), out_max = maxdist (faux_shader.rs:39-52 = max_distance,
 * obb.rs:302-307), in the reference's own operation order on all four lanes.
 * Needs no uploaded tree. */
int sdfgpu_box_bounds(sdfgpu_ctx* ctx, const SdfBoundingBoxBlock* box, const float* pts_xyzw,
                      size_t n, float* out_min, float* out_max);

/* ---- sphere trace + normals -------------------------------------------- */
/* Full frame, host pointers (each may be NULL except hit):
 *   hit[w*h] u8, depth[w*h] f32 (+inf on miss), normal_xyz[3*w*h] f32
 *   (0 on miss), prim[w*h] u32 (SDF_NO_PRIM on miss).
 * Normals: tetrahedron finite difference against the hit primitive only,
 * as calcSphereNormal does (src/main__.rs:151-161). */
int sdfgpu_render(sdfgpu_ctx* ctx, const sdfgpu_camera* cam, uint32_t width,
                  uint32_t height, const sdfgpu_march_params* mp, uint8_t* hit,
                  float* depth, float* normal_xyz, uint32_t* prim);

/* Full frame, device pointers, row-major planes, asynchronous on hip_stream. */
int sdfgpu_render_dev(sdfgpu_ctx* ctx, const sdfgpu_camera* cam, uint32_t width,
                      uint32_t height, const sdfgpu_march_params* mp, uint8_t* d_hit,
                      float* d_depth, float* d_normal_xyz, uint32_t* d_prim, void* hip_stream);

/* Tile-range form for the framebuffer partitioner.  The frame is cut into
 * 8x8-pixel tiles numbered row-major; this call renders tiles
 * tile_first, tile_first + tile_stride, ... (tile_count of them) and writes
 * them compactly, tile after tile, 64 pixels each in row-major order inside
 * the tile (pixels beyond the image edge are written as misses):
 *   d_hit[tile_count*64], d_depth[...], d_normal_xyz[3*...], d_prim[...].
 * Rank r of N passes tile_first = r, tile_stride = N. */
int sdfgpu_render_tiles_dev(sdfgpu_ctx* ctx, const sdfgpu_camera* cam,
                            uint32_t width, uint32_t height,
                            const sdfgpu_march_params* mp, uint32_t tile_first,
                            uint32_t tile_stride, uint32_t tile_count,
                            uint8_t* d_hit, float* d_depth, float* d_normal_xyz,
                            uint32_t* d_prim, void* hip_stream);

/* Scatter tile-compact buffers (as written by sdfgpu_render_tiles_dev for the
 * same tile_first/stride/count) into a row-major w x h frame on the device. */
int sdfgpu_untile_dev(sdfgpu_ctx* ctx, uint32_t width, uint32_t height,
                      uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count,
                      const uint8_t* d_hit_t, const float* d_depth_t,
                      const float* d_normal_t, const uint32_t* d_prim_t,
                      uint8_t* d_hit, float* d_depth, float* d_normal_xyz,
                      uint32_t* d_prim, void* hip_stream);

/* Whole-frame assembly in one launch, for a frame whose tiles were dealt round-robin over `world`
 * ranks (rank p rendered tiles p, p + world, ... with sdfgpu_render_tiles_dev into ONE packed buffer
 * of four planes, [depth | normal xyz | prim | hit], each plane sized for ceil(tiles / world) tiles
 * of 64 pixels; sdfgpu_packed_share_bytes gives the buffer's size).  Rank p's buffer is read at
 * d_gathered + p * share_bytes, except rank `self`, whose buffer is read at d_mine (pass
 * d_gathered + self * share_bytes after an all-gather).  No counterpart in the reference (it has
 * no multi-device code); this is the assembling rank's half of sdfgpu_render_frame_comm for hosts
 * that bring their own collective. */
size_t sdfgpu_packed_share_bytes(uint32_t width, uint32_t height, uint32_t world);
int sdfgpu_untile_gathered_dev(sdfgpu_ctx* ctx, uint32_t width, uint32_t height,
                               uint32_t world, uint32_t self,
                               const void* d_gathered, const void* d_mine,
                               uint8_t* d_hit, float* d_depth, float* d_normal_xyz,
                               uint32_t* d_prim, void* hip_stream);

/* ---- multi-GPU framebuffer assembly ------------------------------------- */
/* No reference counterpart (the crate is single-threaded).  One process per GPU, each with
 * its own context and the same tree uploaded.  A communicator wraps an RCCL communicator (bound at
 * run time from librccl.so.1): one rank calls sdfgpu_comm_get_id and ships the SDFGPU_COMM_ID_BYTES
 * opaque bytes to the others by any means it has (MPI, a file, torch.distributed), then every rank
 * calls sdfgpu_comm_create -- collectively, as ncclCommInitRank requires. */
#define SDFGPU_COMM_ID_BYTES 128
typedef struct sdfgpu_comm sdfgpu_comm;
int sdfgpu_comm_get_id(sdfgpu_ctx* ctx, void* id_out);
int sdfgpu_comm_create(sdfgpu_ctx* ctx, const void* id, int rank, int world, sdfgpu_comm** out);
void sdfgpu_comm_destroy(sdfgpu_comm* comm);
/* One frame across the communicator.  Every rank calls it with the same camera, size and march
 * parameters; rank r renders tiles r, r + world, ... (sdfgpu_render_tiles_dev) into a packed buffer
 * the communicator owns; the buffers travel to `root` in ONE grouped ncclSend / ncclRecv (a
 * point-to-point transfer per peer: over xGMI each on its own link); `root` scatters them into its
 * row-major planes (as sdfgpu_render_dev writes them).  On the other ranks the output pointers are
 * not touched and may be NULL.  Asynchronous on hip_stream; frames on different streams of one
 * communicator may be in flight together (at most 8 streams keep their staging). */
int sdfgpu_render_frame_comm(sdfgpu_ctx* ctx, sdfgpu_comm* comm, const sdfgpu_camera* cam,
                             uint32_t width, uint32_t height, const sdfgpu_march_params* mp, int root,
                             uint8_t* d_hit, float* d_depth, float* d_normal_xyz, uint32_t* d_prim,
                             void* hip_stream);
/* Rules of a frame across a communicator (it is a collective):
 *  - every rank passes IDENTICAL width / height / march parameters / camera / root; nothing checks that across
 *    ranks, and ranks that disagree about the frame's size disagree about the size of the exchange;
 *  - everything that can fail locally is done before anything a peer waits on, and a rank whose own march
 *    failed (no tree uploaded, ...) still takes part in the exchange -- its share arrives as misses -- and
 *    returns its error afterwards; a rank that cannot take part at all (its staging could not be allocated)
 *    aborts the communicator (ncclCommAbort): its peers' pending operations then end with an error instead of
 *    hanging, and every later call on that communicator returns SDFGPU_ERR_COMM;
 *  - the first frame of a (communicator, stream, size) allocates staging and must run outside stream capture;
 *  - destroy communicators BEFORE the context they were created on (sdfgpu_destroy does not know them);
 *  - frames of one communicator on several streams may be in flight together, but RCCL orders the operations
 *    of one communicator, so their exchanges run one after the other;
 *  - UNEXERCISED ON MORE THAN ONE GPU: no multi-GPU machine was in reach of this build.  What has run: the
 *    world-1 path, and the grouped ncclSend / ncclRecv / untile-from-gathered path through a send to self
 *    (SDFGPU_COMM_SELF_EXCHANGE=1 in the environment of the create call makes the assembling rank exchange its
 *    own share with itself; tests/test_gpu_comm.py). */

/* One process, several GPUs (SURVEY 8b/8e "ncclCommInitAll"): a clique over n contexts, each on its own
 * device, rank i = ctxs[i]; no id exchange.  comms_out receives n communicators (destroy each with
 * sdfgpu_comm_destroy).  sdfgpu_render_frame_all is sdfgpu_render_frame_comm for all n ranks from the one
 * calling thread: every rank's tile share on its own device and stream (hip_streams[i], NULL array = the null
 * stream everywhere), then ONE ncclGroupStart / GroupEnd around all sends and receives, then the assembly on
 * `root`, whose device the output pointers belong to.  No counterpart in the reference.  Unexercised on more
 * than one GPU (see above); world 1 is covered by tests/test_gpu_comm.py. */
int sdfgpu_comm_create_all(sdfgpu_ctx* const* ctxs, int n, sdfgpu_comm** comms_out);
int sdfgpu_render_frame_all(sdfgpu_comm* const* comms, int n, const sdfgpu_camera* cam,
                            uint32_t width, uint32_t height, const sdfgpu_march_params* mp, int root,
                            uint8_t* d_hit, float* d_depth, float* d_normal_xyz, uint32_t* d_prim,
                            void* const* hip_streams);

/* ---- query-batch partitioner --------------------------------------------- */
/* No reference counterpart.  A batch of n queries is cut into `world` contiguous slices (the first n % world
 * slices one longer): sdfgpu_query_slice gives slice `rank` as [*first, *first + *count).  The sharded entry
 * points run one batch of HOST points over n_ctx contexts (each with the same tree uploaded; one per GPU, or
 * several on one GPU): slice i is copied to ctxs[i]'s device, queried there (sdfgpu_nearest_dev /
 * sdfgpu_knn_dev), and copied back into the caller's arrays at the slice's place, all contexts in flight
 * together; the call returns when every slice is back.  Results are exactly those of the unsharded call.
 * (A host whose points already live on the devices calls the _dev entry points per context itself.) */
void sdfgpu_query_slice(size_t n, uint32_t rank, uint32_t world, size_t* first, size_t* count);
int sdfgpu_nearest_sharded(sdfgpu_ctx* const* ctxs, int n_ctx, const float* pts_xyzw, size_t n,
                           float* out_dist, uint32_t* out_prim);
int sdfgpu_knn_sharded(sdfgpu_ctx* const* ctxs, int n_ctx, const float* pts_xyzw, size_t n, uint32_t k,
                       float* out_dist, uint32_t* out_prim);

/* ---- instrumentation ---------------------------------------------------- */
/* Run the counting variant of the same kernels (slower) and return the work
 * counters bench.py turns into algorithmic flops. */
int sdfgpu_render_stats(sdfgpu_ctx* ctx, const sdfgpu_camera* cam, uint32_t width,
                        uint32_t height, const sdfgpu_march_params* mp,
                        sdfgpu_stats* out);
int sdfgpu_nearest_stats_dev(sdfgpu_ctx* ctx, const float* d_pts_xyzw, size_t n,
                             sdfgpu_stats* out);
int sdfgpu_knn_stats_dev(sdfgpu_ctx* ctx, const float* d_pts_xyzw, size_t n, uint32_t k,
                         sdfgpu_stats* out);

/* Host helper: fill a sdfgpu_camera for a pinhole at `eye` looking at
 * `target` (geometry of src/main__.rs:261-264), vertical field of view
 * fov_y radians.  Pure host arithmetic in double, rounded once to float. */
void sdfgpu_camera_look_at(const float eye[3], const float target[3],
                           const float up[3], float fov_y, uint32_t width,
                           uint32_t height, sdfgpu_camera* out);

#ifdef __cplusplus
}
#endif
#endif /* SDFGPU_H */
