// gpu_tree.hpp -- the resident (compact) tree the kernels traverse, and the
// host-side compaction that produces it from the crate's 624+224-byte records.
//
// The crate streams 848 B per node (component.rs:27-60).  The kernels need,
// per primitive, a 3x4 point transform + 3 parameters (64 B) and, per internal
// node, the bounds of its two children + two child references (64 B in AABB
// mode, 128 B in OBB mode).  Every record is a whole number of 64-byte lines
// and is fetched with wave-uniform scalar loads.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sdf_blocks.h"
#include "../../include/sdfgpu.h"

namespace sdfgpu {

constexpr uint32_t REF_LEAF = 0x80000000u;  // child reference: bit 31 = primitive
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;  // empty tree / absent child
constexpr uint32_t REF_INDEX = 0x7FFFFFFFu;

// One primitive.  m = rows of the leaf's trans_inverse (world -> primitive
// frame): row r = m[4r .. 4r+3], translation in m[4r+3].  index in the array
// = primitive id = pre-order ordinal.
struct alignas(64) GpuLeaf {
    float m[12];
    float param[3];   // sphere: radius,0,0   box: half extents
    uint32_t kind;    // SDF_PRIM_SPHERE / SDF_PRIM_BOX
};
static_assert(sizeof(GpuLeaf) == 64, "GpuLeaf is one 64-byte line");

// Internal node, AABB / BVH mode: world-space boxes of both children as centre
// + half extent (distance per axis = max(|p - c| - h, 0): three full-rate VALU
// operations, against two subtractions and a three-operand max for lo/hi).
struct alignas(64) GpuInner {
    float c0[3], h0[3];
    float c1[3], h1[3];
    uint32_t ref0, ref1;
    uint32_t pad[2];
};
static_assert(sizeof(GpuInner) == 64, "GpuInner is one 64-byte line");

// Internal node, OBB mode: the buffer's own boxes (full_inverse rows + scale),
// i.e. the operands of MINDIST, faux_shader.rs:54-58.
struct alignas(64) GpuInnerObb {
    float fi0[12];
    float sc0[3];
    uint32_t ref0;
    float fi1[12];
    float sc1[3];
    uint32_t ref1;
};
static_assert(sizeof(GpuInnerObb) == 128, "GpuInnerObb is two 64-byte lines");

// General path (any union tree the reference state machine accepts: internal
// nodes with a non-identity trans_inverse, projective rows): every node in
// pre-order with its full 4x4 trans_inverse, walked exhaustively with a
// per-level point stack -- the literal point chain of faux_shader.rs:160-178.
constexpr uint32_t GEN_MAX_LEVELS = 32;
constexpr uint32_t GEN_EVAL_MAX_LEVELS = 24;   // with the per-level results in LDS as well (walk_eval): 2 KB per level per wavefront
constexpr uint32_t GEN_OP_UNION = 0, GEN_OP_CLONE = 1;
struct alignas(64) GpuGenNode {
    float m[16];       // trans_inverse, column-major as in the buffer
    float param[4];    // sphere: radius; box: half extents; box frame: dimension + thickness
    uint32_t kind;     // leaves: primitive kind; internal nodes: number of primitives in the subtree
    uint32_t is_leaf;
    uint32_t level;    // root = 1
    uint32_t prim;     // leaves: primitive id; internal nodes: the first primitive id of the subtree
    uint32_t skip_to;  // index of the first node behind this node's subtree (pruned walks jump there)
    // evaluation with the real post-order combine (extensions; walk_eval in kernels.hip)
    uint32_t slot;     // which of its parent's two results this node writes (child ordinal in the buffer)
    uint32_t parent;   // index of the parent in this array (REF_NONE for the root)
    uint32_t n_up;     // internal nodes whose subtree ends with this node: combined right after it, deepest first
    uint32_t op;       // internal nodes: GEN_OP_*
    float smooth_k;    // union: smooth radius in effect (0 = plain min)
    float clone[9];    // clone: displacement, neg_limit, pos_limit (xyz each)
    // pruned evaluation (walk_eval_pruned; CompactTree::gen_prune): everything below this node lies inside the box
    // bc +- bh IN THE FRAME OF THE POINT ITS PARENT HANDS DOWN (world for plain unions; the folded frame below a
    // clone; the re-based frame below a node with its own trans_inverse), so outside the box the subtree's value is at
    // least the distance to it minus `slack` (what the smooth unions inside can take off a minimum: k/4 per level)
    float bc[3], bh[3];
    float slack;
    // pruned walks: where the point handed to this node lives in the per-lane point stack (bits 0-7), where the point it
    // hands down goes (bits 8-15), and bit 16: a plain union with an identity trans_inverse hands its own point on
    // unchanged (no product, no new slot) -- which is nearly every internal node, so the stack is a few slots deep
    // instead of one per level
    uint32_t pt_slots;
    float parent_k;    // smooth radius in effect at the parent (0: plain minimum, clone, or no parent)
};
static_assert(sizeof(GpuGenNode) == 192, "GpuGenNode is three 64-byte lines");

// Chain path: a buffer whose internal nodes re-base the point (what the crate's own SdfBoundingBox::merge
// emits, obb.rs:183-185) is pruned over world boxes of the COMPOSED maps, but a leaf value is still the
// reference's sequence of one rounded mat-vec per level (faux_shader.rs:165,175).  One record per level: rows of
// that node's trans_inverse (affine: bottom row exactly 0 0 0 1, so the 3x4 form is bit-identical).
struct alignas(64) GpuChainMat {
    float m[12];
    uint32_t pad[4];
};
static_assert(sizeof(GpuChainMat) == 64, "GpuChainMat is one 64-byte line");
constexpr uint32_t CHAIN_MAX_LEVELS = 64;

struct CompactTree {
    // chain path (empty otherwise): per primitive {first record, count} into chain_mats -- the non-identity
    // trans_inverse of its ancestors, root first (its own is in its GpuLeaf)
    std::vector<GpuChainMat> chain_mats;
    std::vector<uint32_t> leaf_chain;   // 2 per primitive
    uint32_t chain_max = 0;
    std::vector<GpuGenNode> gen;        // general path (empty otherwise)
    bool gen_eval = false;              // general path must run the post-order combine (a smooth union is in effect)
    bool gen_prune = false;             // general path: every node carries a valid box (rigid affine maps throughout): the walks prune
    uint32_t gen_pt_levels = 0;         // pruned walks: slots of the per-lane point stack (GpuGenNode::pt_slots)
    std::vector<GpuLeaf> leaves;
    std::vector<GpuInner> inner;        // AABB mode
    std::vector<GpuInnerObb> inner_obb; // OBB mode
    uint32_t root_ref = REF_NONE;
    uint32_t stack_depth = 1;           // pending far children never exceed this
    float margin_T = 0.f;               // scene term of the prune margin
    float box_scale = 1.f;              // power of two the AABB records (GpuInner c/h) are multiplied by (kernels.hip, aabb_dist2)
    float lipschitz = 1.000004f;        // how fast a leaf value can change per unit the point moves (march budget)
    float mean_extent = 0.f;            // mean primitive size (largest parameter), for tuning defaults
    sdfgpu_tree_info info{};
};

// Validate the crate-format buffers and build the compact tree.  Returns an
// sdfgpu_status; on failure `err` has the detail.  Pure host code (testable
// without a GPU through sdfgpu_compact_host).
int compact_tree(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t n,
                 uint32_t layout, uint32_t bounds, uint32_t extensions, CompactTree* out, std::string* err);

}  // namespace sdfgpu
