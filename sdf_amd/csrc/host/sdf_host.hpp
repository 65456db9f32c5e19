// sdf_host.hpp -- host-side scene API, the C++ mirror of the crate's
// scene-construction layer (the crate is compiled code and its toolchain is
// not in this image, so the host side above the C ABI is C++).
//
//   crate item                               here
//   SdfBoundingBox      obb.rs:79-325        SdfBoundingBox
//   SdfElementInfo      elements.rs:9-61     SdfElementInfo
//   trait SdfElement    elements.rs:63-83    struct SdfElement (virtual)
//   SdfSphere           elements.rs:85-118   SdfSphere
//   (no solid box in the crate, SURVEY C-7)  SdfBox  (primitive op 2)
//   SdfUnion            elements.rs:157-212  SdfUnion
//   SdfNode             node.rs:141-235      SdfNode
//   SdfBuilder          node.rs:371-414      SdfBuilder
//   ExpandedSdfNode     node.rs:20-139       ExpandedSdfNode
//   SdfTreeBuffer       component.rs:80-96   SdfTreeBuffer
//
// Recorded deviations from the crate: see DESIGN.md.
#pragma once
#include <sys/mman.h>

#include <array>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/sdf_blocks.h"
#include "sdf_math.hpp"

namespace sdfhost {

struct BuildOptions {
    // C-3: the crate gives merged boxes trans_inverse = (T(mean) E)^-1 but
    // never re-bases the children.  false (default) writes identity, which is
    // the geometrically consistent choice; true writes the crate's literal
    // value (used by the "general path" fixture).
    bool pca_trans_inverse = false;
    // C-2: make_buffer passes a node's OWN box as other_box (node.rs:110-121).
    // false (default) keeps that; true writes the sibling's box.
    bool sibling_other_box = false;
    // Outward padding of merged boxes so that rounding in the projection
    // cannot leave a child corner outside (keeps the buffer consistent, C-6).
    float merge_pad_rel = 1e-5f;
};

// obb.rs:79-85
struct SdfBoundingBox {
    Mat4 matrix;
    Vec4 scale;
    Mat4 full_inverse;
    Mat4 trans_inverse;

    static SdfBoundingBox unit();   // obb.rs:88-100
    static SdfBoundingBox zero();   // obb.rs:102-114
    // obb.rs:117-187 (PCA-fitted box around the sub-boxes' corners)
    static SdfBoundingBox merge(const std::vector<SdfBoundingBox>& sub, const BuildOptions& o);
    // obb.rs:189-214: median split along the longest side.  Returns indices
    // into `sub`: first = upper half (split_off), second = lower half.
    std::pair<std::vector<size_t>, std::vector<size_t>> split(
        const std::vector<SdfBoundingBox>& sub) const;
    static SdfBoundingBox from_transform(const Transform& t);   // obb.rs:222-224
    SdfBoundingBox apply_transform(const Transform& t) const;   // obb.rs:226-244
    bool is_zero() const;                                        // obb.rs:289-292
    float distance_to(Vec3 p) const;                             // obb.rs:294-300 (MINDIST)
    float max_distance(Vec3 p) const;                            // obb.rs:302-307 (MAXDIST)
    Vec4 in_box_basis(Vec4 p) const { return full_inverse * p; }        // obb.rs:250-254
    Vec4 in_box_trans_basis(Vec4 p) const { return trans_inverse * p; } // obb.rs:256-260
    bool contains(Vec3 p) const;                                 // obb.rs:313-315
    SdfBoundingBoxBlock get_bbox_block() const;                  // obb.rs:317-324
};

// elements.rs:9-51
struct SdfElementInfo {
    size_t num_acc_slots = 0;
    size_t num_drawn_slots = 0;
    bool is_primitive = false;
    uint32_t op_id = 0;
    bool is_union = false;
    static SdfElementInfo primitive_info(uint32_t op) { return {0, 0, true, op, false}; }
    static SdfElementInfo strict_info(uint32_t op, size_t acc, size_t drawn) { return {acc, drawn, false, op, false}; }
    static SdfElementInfo union_info(uint32_t op) { return {0, SIZE_MAX, false, op, true}; }
    size_t num_slots() const {
        return num_drawn_slots == SIZE_MAX ? SIZE_MAX : num_acc_slots + num_drawn_slots;
    }
};

struct SdfNode;
struct ExpandedSdfNode;

// elements.rs:63-83
struct SdfElement {
    virtual ~SdfElement() = default;
    virtual SdfElementInfo get_info() const = 0;
    virtual SdfBoundingBox get_bbox(const std::vector<SdfBoundingBox>& slots, const BuildOptions& o) const = 0;
    virtual Vec3 downtree_transform(Vec3 p) const { return p; }
    virtual float distance_to(Vec3 p) const { return length(p); }
    virtual std::unique_ptr<SdfElement> clone() const = 0;
    virtual SdfOpSpecificBlock get_dt_specific_block() const;
    virtual SdfOpSpecificBlock get_ut_specific_block() const;
    virtual ExpandedSdfNode expand(const SdfNode& node, const BuildOptions& o) const = 0;
};

// elements.rs:85-118
struct SdfSphere : SdfElement {
    float radius;
    explicit SdfSphere(float r) : radius(r) {}
    SdfElementInfo get_info() const override { return SdfElementInfo::primitive_info(SDF_PRIM_SPHERE); }
    SdfBoundingBox get_bbox(const std::vector<SdfBoundingBox>&, const BuildOptions&) const override;
    float distance_to(Vec3 p) const override { return length(p) - radius; }
    std::unique_ptr<SdfElement> clone() const override { return std::make_unique<SdfSphere>(radius); }
    SdfOpSpecificBlock get_dt_specific_block() const override;
    ExpandedSdfNode expand(const SdfNode& node, const BuildOptions& o) const override;
};

// Build-defined solid box (SURVEY C-7): op 2, half extents in vec4s[0];
// distance = the crate's own exact box formula (obb.rs:297-299) on xyz.
struct SdfBox : SdfElement {
    Vec3 half;
    explicit SdfBox(Vec3 h) : half(h) {}
    SdfElementInfo get_info() const override { return SdfElementInfo::primitive_info(SDF_PRIM_BOX); }
    SdfBoundingBox get_bbox(const std::vector<SdfBoundingBox>&, const BuildOptions&) const override;
    float distance_to(Vec3 p) const override;
    std::unique_ptr<SdfElement> clone() const override { return std::make_unique<SdfBox>(half); }
    SdfOpSpecificBlock get_dt_specific_block() const override;
    ExpandedSdfNode expand(const SdfNode& node, const BuildOptions& o) const override;
};

// elements.rs:120-152: the crate's private, unfinished box frame (op 5).  As there: bounding box from
// the dimension, dimension + thickness in the op-specific block, and NO distance_to override, so the
// recursive query sees the trait default |p| (elements.rs:72-74).  The flat query of the crate panics
// on it (faux_shader.rs:65); the GPU path evaluates it only under SDFGPU_EXT_BOX_FRAME.
struct SdfBoxFrame : SdfElement {
    Vec3 dimension;
    float thickness;
    SdfBoxFrame(Vec3 d, float t) : dimension(d), thickness(t) {}
    SdfElementInfo get_info() const override { return SdfElementInfo::primitive_info(SDF_PRIM_BOX_FRAME); }
    SdfBoundingBox get_bbox(const std::vector<SdfBoundingBox>&, const BuildOptions&) const override;
    std::unique_ptr<SdfElement> clone() const override { return std::make_unique<SdfBoxFrame>(dimension, thickness); }
    SdfOpSpecificBlock get_dt_specific_block() const override;
    ExpandedSdfNode expand(const SdfNode& node, const BuildOptions& o) const override;
};

// elements.rs:157-212
struct SdfUnion : SdfElement {
    float smooth_radius;
    explicit SdfUnion(float s = 0.f) : smooth_radius(s) {}
    SdfElementInfo get_info() const override { return SdfElementInfo::union_info(SDF_OP_UNION); }
    SdfBoundingBox get_bbox(const std::vector<SdfBoundingBox>& slots, const BuildOptions& o) const override {
        return SdfBoundingBox::merge(slots, o);
    }
    std::unique_ptr<SdfElement> clone() const override { return std::make_unique<SdfUnion>(smooth_radius); }
    SdfOpSpecificBlock get_ut_specific_block() const override;
    ExpandedSdfNode expand(const SdfNode& node, const BuildOptions& o) const override;
};

// elements.rs:215-262: continuous, axis-aligned clone (lattice repetition of its one slot).
// Carried through the builder, the recursive query and make_buffer exactly as in the crate;
// NOTE what the flat query makes of it (faux_shader.rs:90 returns `right_dist`, and a one-slot
// node only ever fills the left one): a clone subtree evaluates to +inf there (SURVEY A-6).
struct SdfCaaClone : SdfElement {
    Vec3 displacement, neg_limit, pos_limit;
    SdfCaaClone(Vec3 d, Vec3 lo, Vec3 hi) : displacement(d), neg_limit(lo), pos_limit(hi) {}
    SdfElementInfo get_info() const override { return SdfElementInfo::strict_info(SDF_OP_CAA_CLONE, 0, 1); }
    SdfBoundingBox get_bbox(const std::vector<SdfBoundingBox>& slots, const BuildOptions& o) const override;
    Vec3 downtree_transform(Vec3 p) const override;
    std::unique_ptr<SdfElement> clone() const override {
        return std::make_unique<SdfCaaClone>(displacement, neg_limit, pos_limit);
    }
    SdfOpSpecificBlock get_dt_specific_block() const override;
    ExpandedSdfNode expand(const SdfNode& node, const BuildOptions& o) const override;
};

// component.rs:80-96
// Allocator whose resize() leaves trivially constructible elements uninitialised: make_buffer sizes the two
// arrays (170 MB for 100k primitives) and then writes every block exactly once, from several threads; a
// zero-fill pass over them first would cost as much as the fill.
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U>
    struct rebind { using other = DefaultInitAlloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
    // Large buffers (a 100k-primitive tree is 170 MB of blocks, written once by all threads) are asked for in
    // 2 MiB-aligned pieces with a huge-page hint: first-touch page faults, not the stores, were most of
    // make_buffer's time.
    static constexpr size_t HUGE = size_t(2) << 20;
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < 4 * HUGE) return std::allocator<T>::allocate(n);
        void* p = nullptr;
        if (posix_memalign(&p, HUGE, (bytes + HUGE - 1) / HUGE * HUGE) != 0) throw std::bad_alloc();
        (void)madvise(p, (bytes + HUGE - 1) / HUGE * HUGE, MADV_HUGEPAGE);
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) {
        if (n * sizeof(T) < 4 * HUGE) std::allocator<T>::deallocate(p, n);
        else std::free(p);
    }
};

struct SdfTreeBuffer {
    std::vector<SdfOperationBlock, DefaultInitAlloc<SdfOperationBlock>> downtree_buffer;
    std::vector<SdfOperationUptreeBlock, DefaultInitAlloc<SdfOperationUptreeBlock>> uptree_buffer;
    uint32_t buffer_len = 0;
};

// node.rs:20-139
struct ExpandedSdfNode {
    std::unique_ptr<std::array<ExpandedSdfNode, 2>> expanded_slots;  // Option<[Box<..>;2]>
    SdfBoundingBox bbox;
    std::unique_ptr<SdfElement> intern;                               // Option<Box<dyn SdfElement>>

    ExpandedSdfNode() : bbox(SdfBoundingBox::zero()) {}
    ExpandedSdfNode(ExpandedSdfNode&&) = default;
    ExpandedSdfNode& operator=(ExpandedSdfNode&&) = default;

    static ExpandedSdfNode null();                                                   // node.rs:27-33
    static ExpandedSdfNode primitive(const SdfBoundingBox& b, std::unique_ptr<SdfElement> e);   // :35-41
    static ExpandedSdfNode simple_operation(ExpandedSdfNode child, std::unique_ptr<SdfElement> e); // :43-49
    static ExpandedSdfNode operation(ExpandedSdfNode l, ExpandedSdfNode r, const SdfBoundingBox& b,
                                     std::unique_ptr<SdfElement> e);                 // :51-57
    bool is_null() const { return !intern; }                                          // :59-61
    bool is_primitive() const { return !expanded_slots && intern; }                   // :63-65
    bool is_operation() const { return expanded_slots && intern; }                    // :67-69
    bool is_union() const { return is_operation() && intern->get_info().is_union; }   // :71-73
    SdfTreeBuffer make_buffer(const BuildOptions& o = BuildOptions()) const;          // :75-138
};

struct SdfNode;
// node.rs:10-18
struct NodeDistInfo {
    float min_bound, max_bound;
};
struct NnResult {
    const SdfNode* node;
    float distance;
};

// node.rs:141-235
struct SdfNode {
    std::vector<SdfNode> slots;
    std::optional<SdfBoundingBox> bbox;
    std::unique_ptr<SdfElement> intern;

    explicit SdfNode(std::unique_ptr<SdfElement> e) : intern(std::move(e)) {}   // node.rs:148-157
    SdfNode(SdfNode&&) = default;
    SdfNode& operator=(SdfNode&&) = default;

    // node.rs:169-179; the crate returns Err(&str), here: false + message.
    bool set_slot(SdfNode child, const char** err = nullptr);
    SdfBoundingBox calc_bbox_assign(const BuildOptions& o = BuildOptions());     // node.rs:181-200
    ExpandedSdfNode expanded(const BuildOptions& o = BuildOptions()) const {     // node.rs:209-211
        return intern->expand(*this, o);
    }
    bool is_finished() const { return bbox.has_value(); }
    bool is_primitive() const { return intern->get_info().is_primitive; }
    bool is_empty() const { return slots.empty(); }
    Vec3 downtree(Vec3 p) const { return intern->downtree_transform(p); }        // node.rs:225-227
    bool could_contain(Vec3 p) const;                                            // node.rs:250-256
    NodeDistInfo bbox_dist_info(Vec3 p) const;                                   // node.rs:258-267
    // node.rs:331-368: the crate's recursive n-ary branch-and-prune query on the
    // scene tree itself (the path its own tests exercise).  Host code, as in the crate.
    NnResult nearest_neighbor(Vec3 p) const;
};

// node.rs:371-414.  The crate's asserts/expects become std::logic_error.
struct SdfBuilder {
    SdfNode root;
    BuildOptions opts;
    explicit SdfBuilder(SdfNode r, BuildOptions o = BuildOptions()) : root(std::move(r)), opts(o) {}
    static SdfBuilder primitive(std::unique_ptr<SdfElement> prim, BuildOptions o = BuildOptions());  // :376-385
    SdfBuilder operation(std::unique_ptr<SdfElement> op) &&;       // :387-398
    SdfBuilder with(SdfBuilder node) &&;                            // :400-403
    SdfBuilder transform(const Transform& t) &&;                    // :405-408
    SdfNode finalize() &&;                                          // :410-414
};

// Synthetic benchmark scene of SURVEY 8d: n primitives, sphere/box with
// probability 1/2, radius / half extents U[0.5,2), one Transform each
// (translation U[-L,L)^3, rotation from_euler(XYZ, U[0,2pi)^3), the recipe of
// node.rs:471-491 / src/main.rs:13-31), one flat SdfUnion over all of them.
// L <= 0 selects L = 50 (n/1000)^(1/3).
SdfNode make_random_scene(uint32_t n, uint64_t seed, float L, const BuildOptions& o);
float default_scene_extent(uint32_t n);

}  // namespace sdfhost
