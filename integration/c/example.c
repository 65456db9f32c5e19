/* example.c -- the C ABI from plain C (no Python, no torch): build a scene with the host
 * builder (include/sdfhost.h), upload it, run nearest / kNN / render through include/sdfgpu.h.
 *
 *   gcc -std=c11 -O2 -I include integration/c/example.c -L sdf_amd -lsdfgpu -Wl,-rpath,$PWD/sdf_amd -lm -o example
 *
 * Exit status 0 and a line "ok ..." when every call succeeded and the answers are the ones the
 * scene dictates (two unit spheres at x = -3 and x = +3).  Run by tests/test_c_abi.py.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdfgpu.h"
#include "sdfhost.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != SDFGPU_OK) {                                                      \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, sdfgpu_status_string(rc_), \
                    ctx ? sdfgpu_last_error(ctx) : "");                              \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    sdfgpu_ctx* ctx = NULL;

    /* SdfBuilder::primitive(SdfSphere{1}).transform(x=-3).operation(SdfUnion).with(sphere at x=+3).finalize() */
    const float q_id[4] = {0.f, 0.f, 0.f, 1.f}, one[3] = {1.f, 1.f, 1.f};
    const float left[3] = {-3.f, 0.f, 0.f}, right[3] = {3.f, 0.f, 0.f};
    sdfhost_builder* a = sdfhost_transform(sdfhost_primitive_sphere(1.0f, NULL), left, q_id, one);
    sdfhost_builder* b = sdfhost_transform(sdfhost_primitive_sphere(1.0f, NULL), right, q_id, one);
    sdfhost_builder* u = sdfhost_with(sdfhost_operation_union(a, 0.0f), b);
    sdfhost_node* node = sdfhost_finalize(u);
    if (!node) { fprintf(stderr, "builder: %s\n", sdfhost_last_error()); return 1; }
    sdfhost_expanded* ex = sdfhost_expanded_new(node, NULL);
    sdfhost_buffer* buf = ex ? sdfhost_make_buffer(ex, NULL) : NULL;
    if (!buf) { fprintf(stderr, "make_buffer: %s\n", sdfhost_last_error()); return 1; }

    CHECK(sdfgpu_create(0, &ctx));
    CHECK(sdfgpu_upload_tree(ctx, sdfhost_buffer_downtree(buf), sdfhost_buffer_uptree(buf), sdfhost_buffer_len(buf),
                             SDFGPU_LAYOUT_GLAM_SSE2, SDFGPU_BOUNDS_AUTO));
    sdfgpu_tree_info info;
    CHECK(sdfgpu_tree_info_get(ctx, &info));
    if (info.n_prims != 2 || info.n_nodes != 3) { fprintf(stderr, "unexpected tree: %u prims\n", info.n_prims); return 1; }

    /* nearest: |p - c| - 1 for the nearer centre.  Primitive indices are pre-order ordinals of the
     * EXPANDED tree: the crate's split (obb.rs:189-214) puts the upper half first, so the sphere at
     * x = +3 is primitive 0 and the one at x = -3 primitive 1; the tie at x = 0 goes to the later one. */
    const float pts[4][4] = {{-3.f, 0.f, 0.f, 1.f}, {5.f, 0.f, 0.f, 1.f}, {0.f, 0.f, 0.f, 1.f}, {-3.f, 4.f, 0.f, 1.f}};
    const float want_d[4] = {-1.f, 1.f, 2.f, 3.f};
    const uint32_t want_i[4] = {1u, 0u, 1u, 1u};
    float d[4];
    uint32_t id[4];
    CHECK(sdfgpu_nearest(ctx, &pts[0][0], 4, d, id));
    for (int k = 0; k < 4; ++k)
        if (d[k] != want_d[k] || id[k] != want_i[k]) {
            fprintf(stderr, "nearest[%d] = (%g, %u), expected (%g, %u)\n", k, d[k], id[k], want_d[k], want_i[k]);
            return 1;
        }

    /* kNN with k larger than the scene: padded with (+inf, SDF_NO_PRIM) */
    float kd[4 * 3];
    uint32_t ki[4 * 3];
    CHECK(sdfgpu_knn(ctx, &pts[0][0], 4, 3, kd, ki));
    if (kd[0] != -1.f || ki[0] != 1u || kd[1] != 5.f || ki[1] != 0u || !isinf(kd[2]) || ki[2] != SDF_NO_PRIM) {
        fprintf(stderr, "knn row 0 = (%g,%u) (%g,%u) (%g,%u)\n", kd[0], ki[0], kd[1], ki[1], kd[2], ki[2]);
        return 1;
    }

    /* a 64x32 frame from z = -10: both spheres are hit, depth at a sphere's centre pixel is ~9 */
    const float eye[3] = {0.f, 0.f, -10.f}, target[3] = {0.f, 0.f, 0.f}, up[3] = {0.f, 1.f, 0.f};
    enum { W = 64, H = 32 };
    sdfgpu_camera cam;
    sdfgpu_camera_look_at(eye, target, up, 0.9f, W, H, &cam);
    sdfgpu_march_params mp = {64u, 1e-3f, 0.f, 100.f, 1e-3f, 0u};
    static uint8_t hit[W * H];
    static float depth[W * H], normal[W * H * 3];
    static uint32_t prim[W * H];
    CHECK(sdfgpu_render(ctx, &cam, W, H, &mp, hit, depth, normal, prim));
    unsigned n_hit = 0, seen[2] = {0, 0};
    float min_depth = INFINITY;
    for (int k = 0; k < W * H; ++k)
        if (hit[k]) {
            ++n_hit;
            if (prim[k] > 1u) { fprintf(stderr, "pixel %d hit primitive %u\n", k, prim[k]); return 1; }
            seen[prim[k]]++;
            if (depth[k] < min_depth) min_depth = depth[k];
            const float nl = sqrtf(normal[3 * k] * normal[3 * k] + normal[3 * k + 1] * normal[3 * k + 1] +
                                   normal[3 * k + 2] * normal[3 * k + 2]);
            if (fabsf(nl - 1.f) > 1e-4f) { fprintf(stderr, "pixel %d: |normal| = %g\n", k, nl); return 1; }
        }
    if (!seen[0] || !seen[1] || min_depth < 8.9f || min_depth > 10.f) {
        fprintf(stderr, "frame: %u hits (%u / %u), nearest depth %g\n", n_hit, seen[0], seen[1], min_depth);
        return 1;
    }

    /* errors are codes, never aborts: w != 1 is refused (node.rs:335 always extends with 1.0) */
    const float bad[4] = {0.f, 0.f, 0.f, 2.f};
    if (sdfgpu_nearest(ctx, bad, 1, d, id) != SDFGPU_ERR_POINT_W) { fprintf(stderr, "w != 1 was accepted\n"); return 1; }

    printf("ok: %u nodes, %u primitives, %u of %d pixels hit, nearest depth %.4f\n", info.n_nodes, info.n_prims, n_hit,
           W * H, min_depth);
    sdfgpu_destroy(ctx);
    sdfhost_buffer_free(buf);
    sdfhost_expanded_free(ex);
    sdfhost_node_free(node);
    return 0;
}
