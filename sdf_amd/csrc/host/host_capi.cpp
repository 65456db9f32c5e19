// host_capi.cpp -- extern "C" view of the host scene API (include/sdfhost.h).
#include <string>

#include "../../../include/sdfhost.h"
#include "sdf_host.hpp"

using namespace sdfhost;

struct sdfhost_builder { SdfBuilder b; };
struct sdfhost_node { SdfNode n; };
struct sdfhost_expanded { ExpandedSdfNode e; };
struct sdfhost_buffer { SdfTreeBuffer t; };

namespace {
thread_local std::string g_err;

BuildOptions to_opts(const sdfhost_options* o) {
    BuildOptions r;
    if (o) {
        r.pca_trans_inverse = o->pca_trans_inverse != 0;
        r.sibling_other_box = o->sibling_other_box != 0;
        if (o->merge_pad_rel >= 0.f) r.merge_pad_rel = o->merge_pad_rel;
    }
    return r;
}

template <class F>
auto guard(F&& f) -> decltype(f()) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
    } catch (...) {
        g_err = "unknown error";
    }
    return nullptr;
}

SdfBoundingBox from_block(const SdfBoundingBoxBlock* b) {
    SdfBoundingBox r;
    std::memcpy(r.matrix.m, b->matrix.m, 64);
    r.scale = Vec4(b->scale.v[0], b->scale.v[1], b->scale.v[2], b->scale.v[3]);
    std::memcpy(r.full_inverse.m, b->full_inverse.m, 64);
    std::memcpy(r.trans_inverse.m, b->trans_inverse.m, 64);
    return r;
}
}  // namespace

extern "C" {

const char* sdfhost_last_error(void) { return g_err.c_str(); }

sdfhost_builder* sdfhost_primitive_sphere(float radius, const sdfhost_options* opts) {
    return guard([&]() -> sdfhost_builder* {
        return new sdfhost_builder{SdfBuilder::primitive(std::make_unique<SdfSphere>(radius), to_opts(opts))};
    });
}

sdfhost_builder* sdfhost_primitive_box(float hx, float hy, float hz, const sdfhost_options* opts) {
    return guard([&]() -> sdfhost_builder* {
        return new sdfhost_builder{SdfBuilder::primitive(std::make_unique<SdfBox>(Vec3(hx, hy, hz)), to_opts(opts))};
    });
}

sdfhost_builder* sdfhost_primitive_box_frame(float dx, float dy, float dz, float thickness, const sdfhost_options* opts) {
    return guard([&]() -> sdfhost_builder* {
        return new sdfhost_builder{SdfBuilder::primitive(std::make_unique<SdfBoxFrame>(Vec3(dx, dy, dz), thickness), to_opts(opts))};
    });
}

sdfhost_builder* sdfhost_operation_union(sdfhost_builder* b, float smooth_radius) {
    if (!b) return nullptr;
    return guard([&]() -> sdfhost_builder* {
        std::unique_ptr<sdfhost_builder> own(b);
        return new sdfhost_builder{std::move(own->b).operation(std::make_unique<SdfUnion>(smooth_radius))};
    });
}

sdfhost_builder* sdfhost_operation_caa_clone(sdfhost_builder* b, const float displacement[3], const float neg_limit[3],
                                             const float pos_limit[3]) {
    if (!b) return nullptr;
    if (!displacement || !neg_limit || !pos_limit) { delete b; g_err = "null argument"; return nullptr; }
    return guard([&]() -> sdfhost_builder* {
        std::unique_ptr<sdfhost_builder> own(b);
        return new sdfhost_builder{std::move(own->b).operation(std::make_unique<SdfCaaClone>(
            Vec3(displacement[0], displacement[1], displacement[2]), Vec3(neg_limit[0], neg_limit[1], neg_limit[2]),
            Vec3(pos_limit[0], pos_limit[1], pos_limit[2])))};
    });
}

sdfhost_builder* sdfhost_with(sdfhost_builder* b, sdfhost_builder* child) {
    if (!b || !child) { delete b; delete child; g_err = "null builder"; return nullptr; }
    return guard([&]() -> sdfhost_builder* {
        std::unique_ptr<sdfhost_builder> own(b), ch(child);
        return new sdfhost_builder{std::move(own->b).with(std::move(ch->b))};
    });
}

sdfhost_builder* sdfhost_transform(sdfhost_builder* b, const float t[3], const float q[4], const float s[3]) {
    if (!b) return nullptr;
    return guard([&]() -> sdfhost_builder* {
        std::unique_ptr<sdfhost_builder> own(b);
        Transform tr;
        if (t) tr.translation = Vec3(t[0], t[1], t[2]);
        if (q) tr.rotation = Quat(q[0], q[1], q[2], q[3]);
        if (s) tr.scale = Vec3(s[0], s[1], s[2]);
        return new sdfhost_builder{std::move(own->b).transform(tr)};
    });
}

sdfhost_node* sdfhost_finalize(sdfhost_builder* b) {
    if (!b) return nullptr;
    return guard([&]() -> sdfhost_node* {
        std::unique_ptr<sdfhost_builder> own(b);
        return new sdfhost_node{std::move(own->b).finalize()};
    });
}

void sdfhost_builder_free(sdfhost_builder* b) { delete b; }

void sdfhost_quat_from_euler_xyz(float a, float b, float c, float out[4]) {
    Quat q = Quat::from_euler_xyz(a, b, c);
    out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
}

sdfhost_node* sdfhost_random_scene(uint32_t n, uint64_t seed, float extent, const sdfhost_options* opts) {
    return guard([&]() -> sdfhost_node* {
        return new sdfhost_node{make_random_scene(n, seed, extent, to_opts(opts))};
    });
}

float sdfhost_default_extent(uint32_t n) { return default_scene_extent(n); }

uint32_t sdfhost_node_num_slots(const sdfhost_node* n) { return n ? (uint32_t)n->n.slots.size() : 0; }

int sdfhost_node_bbox(const sdfhost_node* n, SdfBoundingBoxBlock* out) {
    if (!n || !out || !n->n.bbox) { g_err = "node has no bounding box"; return -1; }
    *out = n->n.bbox->get_bbox_block();
    return 0;
}

namespace {
// pre-order ordinal of `target` among the primitives under `n`
bool prim_ordinal(const SdfNode& n, const SdfNode* target, uint32_t* next) {
    if (n.is_primitive()) {
        if (&n == target) return true;
        *next += 1;
        return false;
    }
    for (const SdfNode& c : n.slots)
        if (prim_ordinal(c, target, next)) return true;
    return false;
}
}  // namespace

int sdfhost_node_nearest_neighbor(const sdfhost_node* n, const float p[3], float* out_dist, uint32_t* out_prim) {
    if (!n || !p || !out_dist) { g_err = "null argument"; return -1; }
    try {
        const NnResult r = n->n.nearest_neighbor(Vec3(p[0], p[1], p[2]));
        *out_dist = r.distance;
        if (out_prim) {
            uint32_t ord = 0;
            *out_prim = (r.node && r.node->is_primitive() && prim_ordinal(n->n, r.node, &ord)) ? ord : SDF_NO_PRIM;
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

void sdfhost_node_free(sdfhost_node* n) { delete n; }

sdfhost_expanded* sdfhost_expanded_new(const sdfhost_node* n, const sdfhost_options* opts) {
    if (!n) return nullptr;
    return guard([&]() -> sdfhost_expanded* {
        if (!n->n.is_finished()) throw std::logic_error("expanded(): node not finalized");
        return new sdfhost_expanded{n->n.expanded(to_opts(opts))};
    });
}

void sdfhost_expanded_free(sdfhost_expanded* e) { delete e; }

sdfhost_buffer* sdfhost_make_buffer(const sdfhost_expanded* e, const sdfhost_options* opts) {
    if (!e) return nullptr;
    return guard([&]() -> sdfhost_buffer* { return new sdfhost_buffer{e->e.make_buffer(to_opts(opts))}; });
}

uint32_t sdfhost_buffer_len(const sdfhost_buffer* b) { return b ? b->t.buffer_len : 0; }
const SdfOperationBlock* sdfhost_buffer_downtree(const sdfhost_buffer* b) {
    return b ? b->t.downtree_buffer.data() : nullptr;
}
const SdfOperationUptreeBlock* sdfhost_buffer_uptree(const sdfhost_buffer* b) {
    return b ? b->t.uptree_buffer.data() : nullptr;
}
void sdfhost_buffer_free(sdfhost_buffer* b) { delete b; }

float sdfhost_bbox_distance_to(const SdfBoundingBoxBlock* b, const float p[3]) {
    return from_block(b).distance_to(Vec3(p[0], p[1], p[2]));
}
float sdfhost_bbox_max_distance(const SdfBoundingBoxBlock* b, const float p[3]) {
    return from_block(b).max_distance(Vec3(p[0], p[1], p[2]));
}

}  // extern "C"
