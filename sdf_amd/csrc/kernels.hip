// kernels.hip -- CDNA4 (gfx950) kernels of the nearest-primitive / sphere-trace
// path.  Compiled with -ffp-contract=off: every leaf value is produced by the
// same sequence of individually rounded fp32 operations as the reference
// (sdf/src/faux_shader.rs:60-67,165 over glam's Mat4*Vec4), so distances are
// bit-identical to the CPU path and the argmin (primitive index, hit mask)
// cannot flip.  Bounds arithmetic is NOT exact on purpose (explicit fmaf) and
// is only ever used with a conservative margin (DESIGN.md "pruning").
//
// Execution model: one query / ray per lane, 64 per wavefront.  The tree walk
// is WAVE-UNIFORM: the node index lives in an SGPR, node records (64-byte
// lines) arrive through the scalar cache (s_load_dwordx16), and every VALU
// instruction combines SGPR node data with the lane's own point.  A subtree is
// skipped when the ballot of "some lane still needs it" is empty.  Deferred
// ("far") children wait on a per-wavefront stack: the child reference in lane
// `sp` of one VGPR (lane-select write / v_readlane), the 64 per-lane lower bounds in
// LDS ([level][lane], conflict-free).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>

#include "gpu_tree.hpp"
#include "kernels.hpp"

namespace sdfgpu {

// ------------------------------------------------------------ primitives --

// Correctly rounded square root for the hot paths.  The compiler expands sqrtf for this target into
// v_sqrt_f32 (1 ulp) + a one-step correction from the two residuals x - s * (s -/+ 1 ulp), wrapped in a
// 2^32 pre-scale for x < 2^-96 (where a residual would lose bits) and a class test for +-0 / +inf:
// 17 instructions, more than the primitive formula around it.  The arguments here are squared
// distances, ordinary numbers unless a point sits exactly on a primitive's centre or inside a box:
// a wavefront whose arguments all lie in [2^-96, +inf) runs the same correction without the wrapper
// (the wrapper is the identity there, so the bits are the compiler's), any other wavefront runs the
// compiler's expansion.  The choice is per wavefront (one ballot), so there is no divergence.
__device__ __forceinline__ float sqrt_rn(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const bool ordinary = (u - 0x0F800000u) < 0x70000000u;   // 2^-96 <= x < +inf (and x > 0)
    if (__builtin_amdgcn_ballot_w64(!ordinary) != 0ull) return __builtin_sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) - 1u);
    const float su = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) + 1u);
    const float ed = __builtin_fmaf(-sd, s, x), eu = __builtin_fmaf(-su, s, x);
    float r = ed <= 0.f ? sd : s;
    r = eu > 0.f ? su : r;
    return r;
}

// World point -> primitive frame: rows of trans_inverse, glam Mat4*Vec4 order
// ((c0*x + c1*y) + c2*z) + c3*w with w = 1 (c3*1 is exact).  faux_shader.rs:165.
__device__ __forceinline__ void leaf_point(const float* __restrict__ m, float px, float py, float pz,
                                           float& lx, float& ly, float& lz) {
    lx = ((m[0] * px + m[1] * py) + m[2] * pz) + m[3];
    ly = ((m[4] * px + m[5] * py) + m[6] * pz) + m[7];
    lz = ((m[8] * px + m[9] * py) + m[10] * pz) + m[11];
}

// faux_shader.rs:60-67 prim_dispatch: sphere = |p.xyz| - r (scalar Vec3
// length: sqrt((x^2+y^2)+z^2)); box (build-defined op 2) = the crate's exact
// box formula, obb.rs:297-299, on xyz.
__device__ __forceinline__ float prim_value(uint32_t kind, float p0, float p1, float p2, float lx, float ly,
                                            float lz) {
    if (kind == SDF_PRIM_SPHERE) {
        return sqrt_rn((lx * lx + ly * ly) + lz * lz) - p0;
    }
    float qx = __builtin_fabsf(lx) - p0, qy = __builtin_fabsf(ly) - p1, qz = __builtin_fabsf(lz) - p2;
    float mx = qx > 0.f ? qx : 0.f, my = qy > 0.f ? qy : 0.f, mz = qz > 0.f ? qz : 0.f;
    float l = sqrt_rn((mx * mx + my * my) + mz * mz);
    return l + __builtin_fminf(__builtin_fmaxf(qx, __builtin_fmaxf(qy, qz)), 0.0f);
}

// The same value for a per-lane `kind` (the march's seed, the normal taps), where a wavefront mixes
// spheres and boxes: both squared lengths are formed, the lane's own goes through ONE correctly
// rounded sqrt (its refinement is ~15 instructions, more than either formula around it), and the
// result is finished per kind.  Every operation a lane keeps is the one prim_value performs, in the
// same order, so the bits are the same.
__device__ __forceinline__ float prim_value_lane(uint32_t kind, float p0, float p1, float p2, float lx, float ly,
                                                 float lz) {
    const float qx = __builtin_fabsf(lx) - p0, qy = __builtin_fabsf(ly) - p1, qz = __builtin_fabsf(lz) - p2;
    const float mx = qx > 0.f ? qx : 0.f, my = qy > 0.f ? qy : 0.f, mz = qz > 0.f ? qz : 0.f;
    const float ab = (mx * mx + my * my) + mz * mz;
    const float as = (lx * lx + ly * ly) + lz * lz;
    const bool sph = kind == SDF_PRIM_SPHERE;
    const float r = sqrt_rn(sph ? as : ab);
    const float in = __builtin_fminf(__builtin_fmaxf(qx, __builtin_fmaxf(qy, qz)), 0.0f);
    return sph ? r - p0 : r + in;
}

// Per-lane evaluation of one leaf record (the march's seed, the normal taps): the whole 64-byte
// line is requested at once, so that a box does not pay a second memory round trip for the
// half extents its kind selects.
__device__ __forceinline__ float leaf_eval(const GpuLeaf* __restrict__ L, float px, float py, float pz) {
    const float4* __restrict__ q = reinterpret_cast<const float4*>(L);
    const float4 r0 = q[0], r1 = q[1], r2 = q[2];
    float4 r3 = q[3];
    asm volatile("" : "+v"(r3.y), "+v"(r3.z));   // keeps the compiler from sinking half of this load behind the kind test
    const float lx = ((r0.x * px + r0.y * py) + r0.z * pz) + r0.w;
    const float ly = ((r1.x * px + r1.y * py) + r1.z * pz) + r1.w;
    const float lz = ((r2.x * px + r2.y * py) + r2.z * pz) + r2.w;
    return prim_value_lane(__builtin_bit_cast(uint32_t, r3.w), r3.x, r3.y, r3.z, lx, ly, lz);
}

// (distance, index) order of the reference's union combine composed over the
// tree (faux_shader.rs:27-29,87): smaller distance, then the
// LATER primitive.  Order-independent, so any visiting order gives the same
// answer as the reference's pre-order walk.
__device__ __forceinline__ bool is_better(float d, uint32_t id, float best, uint32_t bid) {
    // bitwise on purpose: no short-circuit branches in the walk's inner loop
    return (d < best) | ((d == best) & ((id > bid) | (bid == SDF_NO_PRIM)));
}
__device__ __forceinline__ void take_better(float d, uint32_t id, float& best, uint32_t& bid) {
    const bool b = is_better(d, id, best, bid);
    best = b ? d : best;
    bid = b ? id : bid;
}
// min without the canonicalising v_max the IEEE fminf needs (operands are never NaN here)
__device__ __forceinline__ float min_nn(float a, float b) { return a < b ? a : b; }

// --------------------------------------------------------------- bounds ---

// Squared distance to an axis-aligned box (0 inside), box as centre + half
// extent.  Not exact by design (the prune margin covers its rounding).
// Boxes and the point are in SCALED coordinates (TreeView::bscale, a power of two chosen at upload
// so that anything within 16 scene extents is below 1): the per-axis distance max(|p - c| - h, 0)
// is then one subtraction with the hardware's [0,1] result clamp instead of a subtraction and a
// max.  Scaling by a power of two is exact, so inside that range the test decides exactly what the
// unscaled one would; beyond it the clamp only makes the bound smaller, i.e. prunes less.
//
// FAR.  A point further out than that must not use the clamp: every box would then look at most sqrt(3)
// scaled units away while the search radius is larger, nothing would be pruned, and the walk would visit
// the whole tree for its wavefront (a camera 20 scene extents away, a stray query point).  A walk with
// such a driving lane (far_point, one ballot per walk) therefore runs the copy of the walk whose
// per-axis distance is the plain max(|p - c| - h, 0): one more instruction per axis, same decisions
// inside the range, real pruning outside it.
constexpr float NEAR_LIMIT = 0.9375f;   // the scene lies within 1/16 (compact.cpp), so below this no per-axis distance reaches 1
// The test is on |x| + |y| + |z| (>= the largest coordinate, so it errs towards the far walk, which is
// correct everywhere): that sum is already formed for the prune margin, so the test costs one compare.
__device__ __forceinline__ bool far_point(float px, float py, float pz, float inv_bscale) {
    return !((__builtin_fabsf(px) + __builtin_fabsf(py) + __builtin_fabsf(pz)) < NEAR_LIMIT * inv_bscale);
}
__device__ __forceinline__ float clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.f), 1.f); }
template <bool FAR>
__device__ __forceinline__ float aabb_dist2(const float* __restrict__ c, const float* __restrict__ h, float px,
                                            float py, float pz) {
    float dx = __builtin_fabsf(px - c[0]) - h[0], dy = __builtin_fabsf(py - c[1]) - h[1], dz = __builtin_fabsf(pz - c[2]) - h[2];
    if constexpr (FAR) { dx = __builtin_fmaxf(dx, 0.f); dy = __builtin_fmaxf(dy, 0.f); dz = __builtin_fmaxf(dz, 0.f); }
    else { dx = clamp01(dx); dy = clamp01(dy); dz = clamp01(dz); }
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// What the OBB walk prunes with: the SQUARE of MINDIST (faux_shader.rs:54-58: the point through
// full_inverse, |t| - 1 scaled by the box's half extents, length of the positive part) where the point is
// outside the box, 0 inside it.  A bound, not a result (the prune margin and the threshold's factor cover
// its rounding), so it is formed with fused multiply-adds and without the square root: ~21 instructions
// per box against ~55 for the value in the reference's own operation order, which box_bounds_kernel
// still computes as written (a third of those 55 were the correctly rounded square root).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float obb_outside2(const f32x16& r, float px, float py, float pz) {   // r = fi[12], sc[3], ref
    const float tx = __builtin_fmaf(r[0], px, __builtin_fmaf(r[1], py, __builtin_fmaf(r[2], pz, r[3])));
    const float ty = __builtin_fmaf(r[4], px, __builtin_fmaf(r[5], py, __builtin_fmaf(r[6], pz, r[7])));
    const float tz = __builtin_fmaf(r[8], px, __builtin_fmaf(r[9], py, __builtin_fmaf(r[10], pz, r[11])));
    const float mx = __builtin_fmaxf((__builtin_fabsf(tx) - 1.0f) * r[12], 0.f);
    const float my = __builtin_fmaxf((__builtin_fabsf(ty) - 1.0f) * r[13], 0.f);
    const float mz = __builtin_fmaxf((__builtin_fabsf(tz) - 1.0f) * r[14], 0.f);
    return __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
}

struct TreeView {
    const GpuLeaf* __restrict__ leaf;
    const GpuInner* __restrict__ inner;
    const GpuInnerObb* __restrict__ inner_obb;
    uint32_t root;
    uint32_t depth;     // LDS stack levels per wave
    uint32_t n_leaf;
    float margin_T;
    float skip_ext;     // widening of the search radius that pays for skipped walks (render)
    float inv_bscale;   // 1 / bscale
    float bscale;       // AABB modes: power-of-two scale of the boxes (point and threshold are scaled to match)
    float lip;          // max change of any leaf value per unit the point moves, rounded up (1.000004 for rigid leaves)
    uint32_t prefetch;  // 1: the walk warms the scalar cache with both children's records (trees beyond one XCD's L2)
    uint32_t xcd;       // query kernels: each XCD works through a contiguous range of the batch (xcd_block; SDFGPU_XCD_ORDER=0 turns it off)
    // nearest-candidate grid (kernels.hpp CandidateGrid) -- the march only.  One 64-byte block in HBM (GridBlock:
    // geometry + the two table addresses), fetched by ONE scalar load at the start of a list pass: kept as kernel
    // arguments its 13 values stayed live in SGPRs across the whole march loop and cost the kernel its eighth
    // wavefront per SIMD (106 SGPRs).  Null: no grid.
    const void* __restrict__ gblock;
    // chain path (kernels.hpp DeviceTree::chain; the CHAIN instantiations only): per primitive {first, count}, then the
    // records; mg_scale = 1 + longest chain (the rounding of a leaf value grows with the levels it passes through)
    const void* __restrict__ chain;
    float mg_scale;
};

// Absolute slack covering the rounding of a leaf value at point p (DESIGN.md "pruning").
template <int MODE, bool CHAIN = false>
__device__ __forceinline__ float prune_margin(const TreeView& T, float px, float py, float pz) {
    // T.lip scales it for non-rigid leaves (NONE mode only): the rounding of M p grows with |M|
    float mg_k = ((MODE == SDFGPU_BOUNDS_OBB) ? 1.2e-5f : 3.0e-6f) * T.lip;
    if constexpr (CHAIN) mg_k *= T.mg_scale;   // one rounded mat-vec per level of the chain (margin_T holds the translations' sum)
    return mg_k * (__builtin_fabsf(px) + __builtin_fabsf(py) + __builtin_fabsf(pz) + T.margin_T);
}

struct WaveCounters {
    uint32_t wave_inner = 0, wave_leaf = 0, wave_steps = 0, wave_walks = 0, wave_lists = 0, wave_list_evals = 0, lane_list_steps = 0;
    uint32_t lane_inner = 0, lane_sphere = 0, lane_box = 0, lane_queries = 0, lane_needed = 0;
    uint32_t lane_sphere_needed = 0, lane_box_needed = 0;   // ... of which the lane's own bound asked for
};

// Prune threshold for the current best.  Both kinds of box compare squared distances (0 inside the
// box): a subtree is needed iff dist2(box) <= ((best + mg)^+)^2 * (1+6e-4).  For the OBB walk this is the
// reference's test -- pruned iff m > 0 && m > best (faux_shader.rs:119-120) -- plus the margin, squared:
// inside the box (m <= 0) the key is 0 and never above the threshold, and a negative best + mg (a point
// inside a primitive) makes the threshold 0, so every box the point is outside of is pruned, as there.
template <int MODE>
__device__ __forceinline__ float prune_threshold(float best, float mg, float ext, float bscale) {
    float bp = __builtin_fmaxf(best + mg, 0.f) + ext;
    if constexpr (MODE != SDFGPU_BOUNDS_OBB) bp *= bscale;   // exact: bscale is a power of two
    return bp * bp * 1.0006f;
}
template <int MODE>
__device__ __forceinline__ bool needed(float key, float thr) {
    return key <= thr;
}

// ---- accumulators: what a walk does with a primitive value / a skipped box ----
// radius(): the distance beyond which nothing can matter any more (prune radius).

// K1: the single best under take_better's order.
struct AccNearest {
    static constexpr bool TRACKS = false;
    float best;
    uint32_t bid;
    __device__ __forceinline__ float radius() const { return best; }
    __device__ __forceinline__ void offer(float d, uint32_t id) { take_better(d, id, best, bid); }
    __device__ __forceinline__ void skipped(float) {}
};

// Render: the single best, plus lower bounds on everything that was NOT the
// winner, which let later march steps prove "the same primitive is still the
// nearest" without walking (valid for every lane, driving or not).
struct AccTracked {
    static constexpr bool TRACKS = true;
    float best;
    uint32_t bid;
    float second;   // min computed distance over evaluated primitives other than the winner
    float lb_key;   // min bound key (AABB: squared distance) over subtrees the walk skipped
    __device__ __forceinline__ float radius() const { return best; }
    __device__ __forceinline__ void offer(float d, uint32_t id) {
        const bool rival = id != bid;   // the seed primitive is met again during the walk: not a rival
        const bool b = rival & is_better(d, id, best, bid);
        const float loser = b ? best : d;
        second = (rival & (loser < second)) ? loser : second;
        best = b ? d : best;
        bid = b ? id : bid;
    }
    __device__ __forceinline__ void skipped(float key) { lb_key = min_nn(lb_key, key); }
};

// One 64-byte record through the scalar cache: a single s_load_dwordx16 at
// base + (idx << 6) (SGPR offset form: one shift instead of 64-bit pointer
// arithmetic -- the CU's one scalar unit is the busiest pipe of the walk).
__device__ __forceinline__ f32x16 load_rec64(const void* __restrict__ base, uint32_t idx) {
    f32x16 r;
    const uint32_t off = __builtin_amdgcn_readfirstlane(idx << 6);
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(base), "s"(off) : "memory");
    return r;
}

// The two 64-byte halves of one OBB node record (a child each), both requests in flight before the wait.
// Left to the compiler, `T.inner_obb + cur` became eight 16-byte VECTOR loads per node -- all 64 lanes
// fetching the same 128 bytes into 32 VGPRs each, which also held the kernel at 5 wavefronts per SIMD.
__device__ __forceinline__ void load_rec128(const void* __restrict__ base, uint32_t idx, f32x16& a, f32x16& b) {
    const uint32_t off = __builtin_amdgcn_readfirstlane(idx << 7);
    const uint32_t off1 = off + 64u;
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b)
                 : "s"(base), "s"(off), "s"(off1)
                 : "memory");
}

// The same load into the SGPR tuple the assembly walk uses for its records (s[48:63]): the list pass of the march
// and the walk never run at the same time, so sharing the tuple keeps the march kernel's SGPR count where it was.
__device__ __forceinline__ f32x16 load_rec64_s48(const void* __restrict__ base, uint32_t idx) {
    f32x16 r;
    const uint32_t off = __builtin_amdgcn_readfirstlane(idx << 6);
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&{s[48:63]}"(r) : "s"(base), "s"(off) : "memory");
    return r;
}

// One dword through the scalar cache (a wave-uniform table entry).
__device__ __forceinline__ uint32_t load_u32_scalar(const void* __restrict__ base, uint32_t idx) {
    uint32_t r;
    const uint32_t off = __builtin_amdgcn_readfirstlane(idx << 2);
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(base), "s"(off) : "memory");
    return r;
}

// 64 bytes at base + byte offset into the walk's record tuple (chain records).
__device__ __forceinline__ f32x16 load_rec64_s48_at(const void* __restrict__ base, uint32_t byte_off) {
    f32x16 r;
    const uint32_t off = __builtin_amdgcn_readfirstlane(byte_off);
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&{s[48:63]}"(r) : "s"(base), "s"(off) : "memory");
    return r;
}
__device__ __forceinline__ void load_u32x2_scalar(const void* __restrict__ base, uint32_t byte_off, uint32_t& a, uint32_t& b) {
    unsigned long long r;
    const uint32_t off = __builtin_amdgcn_readfirstlane(byte_off);
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(base), "s"(off) : "memory");
    a = (uint32_t)r;
    b = (uint32_t)(r >> 32);
}

// Bits of one float element of such a record as an integer.  A plain bit cast of
// an extracted element is miscompiled by this toolchain (it reads element 0 of
// the SGPR tuple), so the reinterpretation is one explicit scalar move.
__device__ __forceinline__ uint32_t rec_bits(float f) {
    uint32_t u;
    asm("s_mov_b32 %0, %1" : "=s"(u) : "s"(f));
    return u;
}

__device__ __forceinline__ float leaf_eval_rec(const f32x16& r, float px, float py, float pz) {
    const float lx = ((r[0] * px + r[1] * py) + r[2] * pz) + r[3];
    const float ly = ((r[4] * px + r[5] * py) + r[6] * pz) + r[7];
    const float lz = ((r[8] * px + r[9] * py) + r[10] * pz) + r[11];
    return prim_value(rec_bits(r[15]), r[12], r[13], r[14], lx, ly, lz);
}

// Value of primitive `id` at a world point.  Plain trees: the record the walk fetched.  CHAIN: the reference's point
// chain first -- one rounded 3x4 mat-vec per re-basing ancestor, root first (faux_shader.rs:165,175: never
// pre-multiplied) -- then the primitive's own record.  Chain records arrive by scalar loads into the walk's record
// tuple (which the leaf record is then fetched into again: a scalar-cache hit).
struct TreeView;
template <bool CHAIN>
__device__ __forceinline__ float leaf_value(const TreeView& T, uint32_t id, const f32x16& L, float px, float py, float pz);

template <bool CHAIN>
__device__ __forceinline__ float leaf_value(const TreeView& T, uint32_t id, const f32x16& L, float px, float py, float pz) {
    if constexpr (!CHAIN) {
        return leaf_eval_rec(L, px, py, pz);
    } else {
        uint32_t first, cnt;
        load_u32x2_scalar(T.chain, id << 3, first, cnt);
        const uint32_t mats = (T.n_leaf * 8u + 63u) & ~63u;
        float x = px, y = py, z = pz;
        for (uint32_t j = 0; j < cnt; ++j) {
            const f32x16 M = load_rec64_s48_at(T.chain, mats + ((first + j) << 6));
            const float nx = ((M[0] * x + M[1] * y) + M[2] * z) + M[3];
            const float ny = ((M[4] * x + M[5] * y) + M[6] * z) + M[7];
            const float nz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11];
            x = nx; y = ny; z = nz;
        }
        const f32x16 R = load_rec64_s48(T.leaf, id);
        return leaf_eval_rec(R, x, y, z);
    }
}

// The same value for a per-lane primitive (the march's seed on the chain path): every lane fetches its own chain's
// records with vector loads -- lanes with the same seed read the same lines -- and applies them in the reference's
// order, one rounded mat-vec per level.  The wavefront loops to the longest chain among its lanes; a lane whose chain
// has ended keeps its point.  first / cnt: the lane's {first record, count} (TreeView::chain, per primitive).
__device__ __forceinline__ float leaf_value_lane_chain(const TreeView& T, uint32_t id, uint32_t first, uint32_t cnt, bool on,
                                                       float px, float py, float pz) {
    // (the records' offset is folded into the lane's own 32-bit offset from T.chain: a second base pointer would cost
    // the march kernel an SGPR pair it does not have)
    const uint32_t off0 = ((T.n_leaf * 8u + 63u) & ~63u) + (first << 6);
    float x = px, y = py, z = pz;
    const uint32_t n = on ? cnt : 0u;
    const uint32_t n_max = __builtin_amdgcn_readfirstlane(__ockl_wfred_max_u32(n));
    for (uint32_t j = 0; j < n_max; ++j) {
        if (j < n) {
            const float4* __restrict__ q = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T.chain) + (off0 + (j << 6)));
            const float4 r0 = q[0], r1 = q[1], r2 = q[2];
            const float nx = ((r0.x * x + r0.y * y) + r0.z * z) + r0.w;
            const float ny = ((r1.x * x + r1.y * y) + r1.z * z) + r1.w;
            const float nz = ((r2.x * x + r2.y * y) + r2.z * z) + r2.w;
            x = nx; y = ny; z = nz;
        }
    }
    return on ? leaf_eval(T.leaf + id, x, y, z) : __builtin_inff();
}

// Wave-uniform branch-and-prune walk.  On return the accumulator holds the
// result over every primitive of the tree for each lane whose bit is set in
// `drive` (other lanes ride along: they see every value the wave evaluates).
// `stk` = this wave's LDS stack, [depth][64] floats.  All decisions are made
// on 64-bit lane masks in SGPRs; `ext` widens the search radius (render).
//
// Written as an explicit state machine (inner / leaf / pop) with uniform jumps:
// every branch condition is a scalar, so nothing here needs EXEC masking, and
// the prune threshold is recomputed only when a leaf changed the radius.
template <int MODE, bool COUNT, class Acc>
__device__ __forceinline__ void nn_traverse_cxx(const TreeView& T, float px, float py, float pz, unsigned long long drive,
                                                float ext, Acc& acc, float* __restrict__ stk, uint32_t lane,
                                                WaveCounters& wc) {
    uint32_t cur = T.root;
    if (cur == REF_NONE) return;
    const float mg = prune_margin<MODE>(T, px, py, pz);
    const bool mine = (drive >> lane) & 1ull;
    uint32_t sp = 0;
    int refs = 0;  // lane k holds the reference of stack entry k
    float thr = prune_threshold<MODE>(acc.radius(), mg, ext, T.bscale);
    if constexpr (COUNT) wc.lane_queries += mine ? 1u : 0u;
    cur = __builtin_amdgcn_readfirstlane(cur);
    float kcur = 0.f;   // COUNT: this lane's bound key of the node being visited (the root is always needed)
    // a driving lane beyond the clamp's range: unclamped box distances for this walk (see aabb_dist2)
    const bool far = MODE != SDFGPU_BOUNDS_OBB &&
                     (__builtin_amdgcn_ballot_w64(far_point(px, py, pz, T.inv_bscale)) & drive) != 0ull;

top:   // single loop header (keeps the control flow reducible): cur is a valid reference
    if (cur & REF_LEAF) goto visit_leaf;
    {
        float k0, k1;
        uint32_t r0, r1;
        if constexpr (MODE == SDFGPU_BOUNDS_OBB) {
            f32x16 A, B;   // child 0 / child 1: fi[12], sc[3], ref (GpuInnerObb)
            load_rec128(T.inner_obb, cur, A, B);
            k0 = obb_outside2(A, px, py, pz);
            k1 = obb_outside2(B, px, py, pz);
            r0 = rec_bits(A[15]);
            r1 = rec_bits(B[15]);
        } else {
            const f32x16 N = load_rec64(T.inner, cur);
            const float c0[3] = {N[0], N[1], N[2]}, h0[3] = {N[3], N[4], N[5]};
            const float c1[3] = {N[6], N[7], N[8]}, h1[3] = {N[9], N[10], N[11]};
            if (far) {
                k0 = aabb_dist2<true>(c0, h0, px * T.bscale, py * T.bscale, pz * T.bscale);
                k1 = aabb_dist2<true>(c1, h1, px * T.bscale, py * T.bscale, pz * T.bscale);
            } else {
                k0 = aabb_dist2<false>(c0, h0, px * T.bscale, py * T.bscale, pz * T.bscale);
                k1 = aabb_dist2<false>(c1, h1, px * T.bscale, py * T.bscale, pz * T.bscale);
            }
            r0 = rec_bits(N[12]);
            r1 = rec_bits(N[13]);
        }
        if constexpr (COUNT) {
            wc.wave_inner += 1;
            wc.lane_inner += mine ? 1u : 0u;
        }
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(needed<MODE>(k0, thr)) & drive;
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64(needed<MODE>(k1, thr)) & drive;
        if constexpr (COUNT) wc.lane_needed += (unsigned)(((m0 | m1) >> lane) & 1ull);
        if (m0 != 0ull) {
            if (m1 != 0ull) {
                // both needed by someone: the lanes that need a child vote for the nearer one
                const unsigned long long c = __builtin_amdgcn_ballot_w64(k0 <= k1);
                const uint32_t n0 = (uint32_t)__builtin_popcountll(m0 & (c | ~m1));
                const uint32_t nt = (uint32_t)__builtin_popcountll(m0 | m1);
                const bool first0 = 2u * n0 >= nt;
                stk[sp * 64u + lane] = first0 ? k1 : k0;
                refs = (lane == sp) ? (int)(first0 ? r1 : r0) : refs;   // v_writelane has no builtin here
                sp += 1;
                cur = first0 ? r0 : r1;
                kcur = first0 ? k0 : k1;
            } else {
                acc.skipped(k1);
                cur = r0;
                kcur = k0;
            }
        } else {
            acc.skipped(k0);
            if (m1 == 0ull) {
                acc.skipped(k1);
                goto pop;
            }
            cur = r1;
            kcur = k1;
        }
        goto top;
    }

visit_leaf: {
        const uint32_t id = cur & REF_INDEX;
        const f32x16 L = load_rec64(T.leaf, id);
        const float thr_before = thr;
        acc.offer(leaf_eval_rec(L, px, py, pz), id);
        thr = prune_threshold<MODE>(acc.radius(), mg, ext, T.bscale);
        if constexpr (COUNT) {
            wc.wave_leaf += 1;
            const uint32_t want = (mine && needed<MODE>(kcur, thr_before)) ? 1u : 0u;
            if (rec_bits(L[15]) == SDF_PRIM_SPHERE) { wc.lane_sphere += mine ? 1u : 0u; wc.lane_sphere_needed += want; }
            else { wc.lane_box += mine ? 1u : 0u; wc.lane_box_needed += want; }
        }
    }

pop:
    for (;;) {
        if (sp == 0) return;
        sp -= 1;
        const float k = stk[sp * 64u + lane];
        if ((__builtin_amdgcn_ballot_w64(needed<MODE>(k, thr)) & drive) != 0ull) break;
        acc.skipped(k);
    }
    cur = (uint32_t)__builtin_amdgcn_readlane(refs, sp);
    kcur = stk[sp * 64u + lane];
    goto top;
}

// ---- the same walk, inner part hand-scheduled ------------------------------
// rocprofv3 showed the CU's single scalar unit as the busiest pipe of the
// compiler's version (~44 SALU per visited node: 64-bit pointer arithmetic,
// materialised booleans, control-flow glue), so the part of the walk between
// two primitive evaluations -- descend through internal nodes, push / pop the
// per-wave stack -- is one block of gfx950 assembly: 1 shift + 1 s_load_dwordx16
// per node, SCC taken straight from the s_and_b64 that forms the lane mask,
// ~12 SALU on the two-children path, 3 on the one-child path.  Primitive
// evaluation (bit-exact arithmetic, accumulators) stays in C++.
//
// Contract of the block: on entry `cur` is an internal/leaf reference or
// REF_POP ("take the next pending subtree"); on exit `cur` is a leaf reference
// to evaluate, with that leaf's record in `rec`, or REF_NONE when the walk is
// over.  s[48:63] hold the internal node's record (and, on exit, the leaf's: the same registers, so the block costs
// 18 fixed SGPRs, placed low so that they do not raise the kernel's SGPR count), s64/s65 take the prefetches.  Hazards: no VALU-written SGPR is read by a VALU here (gfx950 needs
// two wait states for that); lane masks go VALU->SALU->VALU, which interlocks.
//
// Operand order: in a loop of nothing else, a VOP3-encoded instruction with its SGPR operand in src1 issues
// at 1.76 ns per wavefront on gfx950 and the same instruction with the SGPR in src0 at 1.01 ns, like a plain
// VOP2 (tools/ubench/issue_rates.hip, KINDs 14 / 26-31).  Hence `v_add -h, |a| clamp` for |a| - h (same
// bits: a - b = a + (-b)) rather than `v_sub |a|, h clamp` -- although in this block, three at a time between
// other instructions, the two forms measured the same (frame 1.728 vs 1.735 ms).
constexpr uint32_t REF_POP = 0xFFFFFFFEu;

// |a| - h, floored at 0, per axis: with the result clamp (near walk) or with an explicit max (far walk).
#define SDFGPU_BOX_NEAR(H0, H1, H2)                                                                    \
    "v_add_f32 %[a], -" H0 ", |%[a]| clamp\n\t"                                                        \
    "v_add_f32 %[b], -" H1 ", |%[b]| clamp\n\t"                                                        \
    "v_add_f32 %[c], -" H2 ", |%[c]| clamp\n\t"
#define SDFGPU_BOX_FAR(H0, H1, H2)                                                                     \
    "v_add_f32 %[a], -" H0 ", |%[a]|\n\t"                                                              \
    "v_add_f32 %[b], -" H1 ", |%[b]|\n\t"                                                              \
    "v_add_f32 %[c], -" H2 ", |%[c]|\n\t"                                                              \
    "v_max_f32 %[a], 0, %[a]\n\t"                                                                     \
    "v_max_f32 %[b], 0, %[b]\n\t"                                                                     \
    "v_max_f32 %[c], 0, %[c]\n\t"

#define SDFGPU_WALK_ASM(BOX, LBMIN_K0, LBMIN_K1, LBMIN_A, PREFETCH, CNT_INNER, CNT_LEAF)                                              \
    "s_cmp_eq_u32 %[cur], -2\n\t"                                                                      \
    "s_cbranch_scc1 .Lpop%=\n"                                                                         \
    ".Ltop%=:\n\t"                                                                                     \
    "s_bitcmp1_b32 %[cur], 31\n\t"                                                                     \
    "s_cbranch_scc1 .Ldone%=\n\t"                                                                      \
    "s_lshl_b32 %[t0], %[cur], 6\n\t"                                                                  \
    "s_load_dwordx16 s[48:63], %[base], %[t0]\n\t"                                                     \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                         \
    CNT_INNER                                                                                          \
    PREFETCH                                                                                           \
    "v_subrev_f32 %[a], s48, %[px]\n\t"                                                              \
    "v_subrev_f32 %[b], s49, %[py]\n\t"                                                              \
    "v_subrev_f32 %[c], s50, %[pz]\n\t"                                                              \
    BOX("s51", "s52", "s53")                                                                           \
    "v_mul_f32 %[k0], %[a], %[a]\n\t"                                                                \
    "v_fmac_f32 %[k0], %[b], %[b]\n\t"                                                               \
    "v_fmac_f32 %[k0], %[c], %[c]\n\t"                                                               \
    "v_subrev_f32 %[a], s54, %[px]\n\t"                                                              \
    "v_subrev_f32 %[b], s55, %[py]\n\t"                                                              \
    "v_subrev_f32 %[c], s56, %[pz]\n\t"                                                              \
    BOX("s57", "s58", "s59")                                                                           \
    "v_mul_f32 %[k1], %[a], %[a]\n\t"                                                                \
    "v_fmac_f32 %[k1], %[b], %[b]\n\t"                                                               \
    "v_fmac_f32 %[k1], %[c], %[c]\n\t"                                                               \
    "v_cmp_le_f32 vcc, %[k0], %[thr]\n\t"                                                              \
    "s_and_b64 %[mk0], vcc, %[drive]\n\t"                                                               \
    "v_cmp_le_f32 vcc, %[k1], %[thr]\n\t"                                                              \
    "s_cbranch_scc0 .Lno0%=\n\t"                                                                       \
    "s_and_b64 %[mk1], vcc, %[drive]\n\t"                                                               \
    "s_cbranch_scc0 .Lonly0%=\n\t"                                                                     \
    /* both children needed by someone: lanes that need one vote for the nearer */                    \
    "v_cmp_le_f32 vcc, %[k0], %[k1]\n\t"                                                               \
    "s_orn2_b64 %[t], vcc, %[mk1]\n\t"                                                                  \
    "s_and_b64 %[t], %[t], %[mk0]\n\t"                                                                  \
    "s_bcnt1_i32_b64 %[t0], %[t]\n\t"                                                                  \
    "s_or_b64 %[t], %[mk0], %[mk1]\n\t"                                                                  \
    "s_bcnt1_i32_b64 %[t1], %[t]\n\t"                                                                  \
    "s_lshl_b32 %[t0], %[t0], 1\n\t"                                                                   \
    "s_cmp_ge_u32 %[t0], %[t1]\n\t"          /* SCC = child 0 first */                                 \
    "s_cselect_b64 vcc, -1, 0\n\t"                                                                     \
    "v_cndmask_b32 %[a], %[k0], %[k1], vcc\n\t"   /* key of the deferred child */                      \
    "v_lshl_add_u32 %[b], %[sp], 8, %[sa]\n\t"                                                         \
    "ds_write_b32 %[b], %[a]\n\t"                                                                      \
    "s_cselect_b32 %[t1], s61, s60\n\t"      /* deferred reference */                                  \
    "s_cselect_b32 %[cur], s60, s61\n\t"                                                               \
    "s_mov_b32 m0, %[sp]\n\t"                                                                          \
    "s_nop 0\n\t"                                                                                      \
    "v_writelane_b32 %[refs], %[t1], m0\n\t"                                                           \
    "s_add_u32 %[sp], %[sp], 1\n\t"                                                                    \
    "s_branch .Ltop%=\n"                                                                               \
    ".Lonly0%=:\n\t"                                                                                   \
    LBMIN_K1                                                                                           \
    "s_mov_b32 %[cur], s60\n\t"                                                                        \
    "s_branch .Ltop%=\n"                                                                               \
    ".Lno0%=:\n\t"                                                                                     \
    LBMIN_K0                                                                                           \
    "s_and_b64 %[mk1], vcc, %[drive]\n\t"                                                               \
    "s_cbranch_scc0 .Lnone%=\n\t"                                                                      \
    "s_mov_b32 %[cur], s61\n\t"                                                                        \
    "s_branch .Ltop%=\n"                                                                               \
    ".Lnone%=:\n\t"                                                                                    \
    LBMIN_K1                                                                                           \
    ".Lpop%=:\n\t"                                                                                     \
    "s_cmp_eq_u32 %[sp], 0\n\t"                                                                        \
    "s_cbranch_scc1 .Lempty%=\n\t"                                                                     \
    "s_sub_u32 %[sp], %[sp], 1\n\t"                                                                    \
    "v_lshl_add_u32 %[b], %[sp], 8, %[sa]\n\t"                                                         \
    "ds_read_b32 %[a], %[b]\n\t"                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                         \
    "v_cmp_le_f32 vcc, %[a], %[thr]\n\t"                                                               \
    "s_and_b64 %[t], vcc, %[drive]\n\t"                                                                \
    "s_cbranch_scc1 .Ltake%=\n\t"                                                                      \
    LBMIN_A                                                                                            \
    "s_branch .Lpop%=\n"                                                                               \
    ".Ltake%=:\n\t"                                                                                    \
    "v_readlane_b32 %[cur], %[refs], %[sp]\n\t"                                                        \
    "s_branch .Ltop%=\n"                                                                               \
    ".Lempty%=:\n\t"                                                                                   \
    "s_mov_b32 %[cur], -1\n\t"                                                                         \
    "s_branch .Lexit%=\n"                                                                              \
    ".Ldone%=:\n\t"                        /* leaf reference: fetch its record (bit 31 shifts out) */    \
    CNT_LEAF                                                                                           \
    "s_lshl_b32 %[t0], %[cur], 6\n\t"                                                                  \
    "s_load_dwordx16 %[rec], %[lbase], %[t0]\n"                                                        \
    ".Lexit%=:\n\t"                        /* nothing (prefetches included) may still be in flight */   \
    "s_waitcnt lgkmcnt(0)\n\t"

// Warm the scalar cache with both children's records while the box tests run: the
// record of whichever child is visited next then costs a cache hit (~100 cycles)
// instead of an L2 round trip (~210).  The dwords land in two scratch SGPRs; the block
// waits for every outstanding scalar load before it ends, so none can land later in a
// register the compiler has meanwhile given to something else.  It costs 8 scalar
// instructions per node, which only pays when records actually miss: measured at 8 waves
// per SIMD, the 100k-primitive tree (12.8 MB) is 5 % faster with it (kNN 1.14 vs 1.20 ms),
// the 10k-primitive tree (1.3 MB, resident in every XCD's L2) 2.4 % slower (frame 1.83 vs
// 1.78 ms) -- so TreeView::prefetch selects, per tree (make_view), between two copies of the walk (a
// test-and-skip inside the block costs a taken branch per node and gave the 2.4 % back).
#define SDFGPU_WALK_PREFETCH                                                                           \
    "s_lshl_b32 %[t0], s60, 6\n\t"                                                                     \
    "s_bitcmp1_b32 s60, 31\n\t"                                                                        \
    "s_cselect_b64 %[t], %[lbase], %[base]\n\t"                                                        \
    "s_load_dword s64, %[t], %[t0]\n\t"                                                                \
    "s_lshl_b32 %[t1], s61, 6\n\t"                                                                     \
    "s_bitcmp1_b32 s61, 31\n\t"                                                                        \
    "s_cselect_b64 %[mk0], %[lbase], %[base]\n\t"                                                       \
    "s_load_dword s65, %[mk0], %[t1]\n\t"

// DBG: the block also counts the internal nodes and leaves it visits in two SGPRs (the instrumented
// build behind SDFGPU_OPT_STATS_ASM_WALK; tests compare them with the C++ walk's counters).
template <bool TRACK, bool DBG, bool PF, bool FAR>
__device__ __forceinline__ void walk_advance(uint32_t& cur, uint32_t& sp, int& refs, float& lb, float thr, float px,
                                             float py, float pz, unsigned long long drive, const GpuInner* base,
                                             const GpuLeaf* lbase, uint32_t sa, f32x16& rec, uint32_t& n_inner,
                                             uint32_t& n_leaf) {
    unsigned long long m0, m1, t;
    uint32_t t0, t1;
    float a, b, c, d, k0, k1;
#define SDFGPU_WALK_OUT                                                                                               \
    [cur] "+s"(cur), [sp] "+s"(sp), [refs] "+v"(refs), [lb] "+v"(lb), [mk0] "=&s"(m0), [mk1] "=&s"(m1), [t] "=&s"(t), \
      [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [k0] "=&v"(k0),          \
      [k1] "=&v"(k1), [rec] "=&{s[48:63]}"(rec)
#define SDFGPU_WALK_IN                                                                                                \
    [thr] "v"(thr), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [drive] "s"(drive), [base] "s"(base), [lbase] "s"(lbase), [sa] "v"(sa)
#define SDFGPU_WALK_CLOBBER                                                                                           \
    "vcc", "scc", "memory", "s64", "s65"
#define SDFGPU_LB_K0 "v_min_f32 %[lb], %[lb], %[k0]\n\t"
#define SDFGPU_LB_K1 "v_min_f32 %[lb], %[lb], %[k1]\n\t"
#define SDFGPU_LB_A "v_min_f32 %[lb], %[lb], %[a]\n\t"
    // The block writes M0 (lane select of v_writelane).  M0 is a reserved register to this compiler: it
    // never keeps a value live in it across other code and re-materialises it before each use of its own,
    // so it cannot be listed as a clobber (-Winline-asm: "clobber list contains reserved registers: m0").
    // PF selects the copy with / without the scalar-cache prefetch (string literals cannot be chosen by a
    // template argument, hence the two branches per flavour).
#define SDFGPU_WALK_FLAVOUR(BOX, PFTEXT)                                                                              \
    if constexpr (DBG) {                                                                                              \
        static_assert(!TRACK, "the counting walk is instantiated for the query kernels only");                        \
        asm volatile(SDFGPU_WALK_ASM(BOX, "", "", "", PFTEXT, "s_add_u32 %[ni], %[ni], 1\n\t", "s_add_u32 %[nl], %[nl], 1\n\t") \
                     : SDFGPU_WALK_OUT, [ni] "+s"(n_inner), [nl] "+s"(n_leaf) : SDFGPU_WALK_IN : SDFGPU_WALK_CLOBBER);   \
    } else if constexpr (TRACK) {                                                                                     \
        asm volatile(SDFGPU_WALK_ASM(BOX, SDFGPU_LB_K0, SDFGPU_LB_K1, SDFGPU_LB_A, PFTEXT, "", "")                    \
                     : SDFGPU_WALK_OUT : SDFGPU_WALK_IN : SDFGPU_WALK_CLOBBER);                                       \
    } else {                                                                                                          \
        asm volatile(SDFGPU_WALK_ASM(BOX, "", "", "", PFTEXT, "", "") : SDFGPU_WALK_OUT : SDFGPU_WALK_IN : SDFGPU_WALK_CLOBBER); \
    }
    if constexpr (FAR) {   // rare: one copy, without the prefetch
        static_assert(!PF, "the far walk has no prefetching copy");
        SDFGPU_WALK_FLAVOUR(SDFGPU_BOX_FAR, "")
    } else if constexpr (PF) {
        SDFGPU_WALK_FLAVOUR(SDFGPU_BOX_NEAR, SDFGPU_WALK_PREFETCH)
    } else {
        SDFGPU_WALK_FLAVOUR(SDFGPU_BOX_NEAR, "")
    }
#undef SDFGPU_WALK_FLAVOUR
#undef SDFGPU_WALK_OUT
#undef SDFGPU_WALK_IN
#undef SDFGPU_WALK_CLOBBER
}

template <int MODE, bool DBG, bool PF, bool FAR, bool CHAIN, class Acc>
__device__ __forceinline__ void nn_traverse_asm(const TreeView& T, float px, float py, float pz,
                                                unsigned long long drive, float ext, Acc& acc,
                                                float* __restrict__ stk, uint32_t lane, WaveCounters& wc) {
    uint32_t cur = __builtin_amdgcn_readfirstlane(T.root);
    if (cur == REF_NONE) return;
    const float mg = prune_margin<MODE, CHAIN>(T, px, py, pz);
    uint32_t sp = 0;
    int refs = 0;   // lane k holds the reference of stack entry k
    float lb = __builtin_inff();
    if constexpr (Acc::TRACKS) lb = acc.lb_key;
    float thr = prune_threshold<MODE>(acc.radius(), mg, ext, T.bscale);
    // LDS byte address of this lane's slot in stack level 0; level l is 256 bytes further
    const uint32_t sa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)stk + lane * 4u;
    const float sx = px * T.bscale, sy = py * T.bscale, sz = pz * T.bscale;   // the boxes' coordinates
    uint32_t n_inner = 0, n_leaf = 0;   // DBG only
    for (;;) {
        f32x16 L;   // the leaf's record, fetched inside the block (undefined when the walk is over)
        walk_advance<Acc::TRACKS, DBG, PF, FAR>(cur, sp, refs, lb, thr, sx, sy, sz, drive, T.inner, T.leaf, sa, L, n_inner, n_leaf);
        cur = __builtin_amdgcn_readfirstlane(cur);
        sp = __builtin_amdgcn_readfirstlane(sp);
        if (cur == REF_NONE) break;
        const uint32_t id = cur & REF_INDEX;
        acc.offer(leaf_value<CHAIN>(T, id, L, px, py, pz), id);
        thr = prune_threshold<MODE>(acc.radius(), mg, ext, T.bscale);
        cur = REF_POP;
    }
    if constexpr (Acc::TRACKS) acc.lb_key = lb;
    if constexpr (DBG) {
        wc.wave_inner += __builtin_amdgcn_readfirstlane(n_inner);
        wc.wave_leaf += __builtin_amdgcn_readfirstlane(n_leaf);
        wc.lane_queries += ((drive >> lane) & 1ull) ? 1u : 0u;
    }
}

// COUNT: 0 = production, 1 = the C++ state machine with every counter, 2 = the assembly walk with its
// wave-level visit counters (query kernels only; what SDFGPU_OPT_STATS_ASM_WALK selects).
// FARCHK false: the caller knows that no point it will ever pass is outside the near range (a march whose
// eye and far plane are both well inside it: launch_render_m decides on the host), so that kernel carries
// neither the per-walk test nor the far copy -- which in the march's loop cost 1.2 % of the frame, and four
// more SGPRs for a flag instead cost it its eighth wavefront per SIMD (9 %).
template <int MODE, int COUNT, bool FARCHK = true, bool CHAIN = false, class Acc>
__device__ __forceinline__ void nn_traverse(const TreeView& T, float px, float py, float pz, unsigned long long drive,
                                            float ext, Acc& acc, float* __restrict__ stk, uint32_t lane,
                                            WaveCounters& wc) {
    static_assert(!CHAIN || (MODE != SDFGPU_BOUNDS_OBB && COUNT == 0), "the chain path runs the assembly walk over the library's own hierarchy, uncounted");
    if constexpr (MODE == SDFGPU_BOUNDS_OBB || COUNT == 1) nn_traverse_cxx<MODE, COUNT != 0>(T, px, py, pz, drive, ext, acc, stk, lane, wc);
    else if (FARCHK && (__builtin_amdgcn_ballot_w64(far_point(px, py, pz, T.inv_bscale)) & drive) != 0ull)
        nn_traverse_asm<MODE, COUNT == 2, false, true, CHAIN>(T, px, py, pz, drive, ext, acc, stk, lane, wc);
    else if (__builtin_amdgcn_readfirstlane(T.prefetch)) nn_traverse_asm<MODE, COUNT == 2, true, false, CHAIN>(T, px, py, pz, drive, ext, acc, stk, lane, wc);
    else nn_traverse_asm<MODE, COUNT == 2, false, false, CHAIN>(T, px, py, pz, drive, ext, acc, stk, lane, wc);
}

__device__ __forceinline__ void flush_counters(const WaveCounters& wc, uint32_t lane, uint32_t steps, uint32_t rays,
                                               sdfgpu_stats* st) {
    // wave-reduce the per-lane counters, one atomic per counter per wave
    uint32_t v[10] = {wc.lane_queries, wc.lane_inner, wc.lane_sphere, wc.lane_box, steps, rays, wc.lane_needed,
                      wc.lane_sphere_needed, wc.lane_box_needed, wc.lane_list_steps};
#pragma unroll
    for (int i = 0; i < 10; ++i)
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
    if (lane == 0) {
        atomicAdd((unsigned long long*)&st->queries, (unsigned long long)v[0]);
        atomicAdd((unsigned long long*)&st->inner_visits, (unsigned long long)v[1]);
        atomicAdd((unsigned long long*)&st->sphere_evals, (unsigned long long)v[2]);
        atomicAdd((unsigned long long*)&st->box_evals, (unsigned long long)v[3]);
        atomicAdd((unsigned long long*)&st->steps, (unsigned long long)v[4]);
        atomicAdd((unsigned long long*)&st->rays, (unsigned long long)v[5]);
        atomicAdd((unsigned long long*)&st->inner_needed, (unsigned long long)v[6]);
        atomicAdd((unsigned long long*)&st->sphere_needed, (unsigned long long)v[7]);
        atomicAdd((unsigned long long*)&st->box_needed, (unsigned long long)v[8]);
        atomicAdd((unsigned long long*)&st->wave_inner, (unsigned long long)wc.wave_inner);
        atomicAdd((unsigned long long*)&st->wave_leaf, (unsigned long long)wc.wave_leaf);
        atomicAdd((unsigned long long*)&st->wave_steps, (unsigned long long)wc.wave_steps);
        atomicAdd((unsigned long long*)&st->wave_walks, (unsigned long long)wc.wave_walks);
        atomicAdd((unsigned long long*)&st->list_steps, (unsigned long long)v[9]);
        atomicAdd((unsigned long long*)&st->wave_lists, (unsigned long long)wc.wave_lists);
        atomicAdd((unsigned long long*)&st->wave_list_evals, (unsigned long long)wc.wave_list_evals);
    }
}

// The domain of a query point: every coordinate finite and at most 2^62 in magnitude.  Outside it the
// reference's own result is an accident: once a leaf value is NaN (a NaN or infinite coordinate, or
// inf - inf after a product overflowed) every comparison with it is false, so `if a < b {a} else {b}`
// (faux_shader.rs:27-29) keeps whatever came last and the prune test never fires; and where all values
// overflow to +inf they all tie, which no walk can resolve without visiting every primitive.  Defined
// instead (sdfgpu.h): such a point takes no part in the walk (it costs its wavefront nothing) and returns
// (NaN, SDF_NO_PRIM) in every slot.  Below 2^62 no product of a coordinate with a matrix entry of a sane
// scene overflows, so no leaf value is NaN.
constexpr float DOMAIN_LIMIT = 4611686018427387904.0f;   // 2^62
__device__ __forceinline__ bool in_domain3(float x, float y, float z) {
    return __builtin_fabsf(x) <= DOMAIN_LIMIT && __builtin_fabsf(y) <= DOMAIN_LIMIT && __builtin_fabsf(z) <= DOMAIN_LIMIT;   // false for NaN
}

// The march keeps its points inside that domain by construction (build-defined like the loop itself,
// restated by oracle_render): a ray is a miss outright when a coordinate of the eye or t_min is beyond
// 2^60 or not a number, or its direction is not finite (a pixel whose image-plane point is the eye), and
// it ends as a miss when t leaves [.., min(t_max, 2^60)] -- `!(t <= t_hi)`, which also catches a NaN.
// Then |p| <= |eye| + |t| < 2^62 at every step, so every nearest-primitive distance the loop adds is a
// number, and a garbage camera costs nothing instead of a tree-wide walk per wavefront and step.
constexpr float MARCH_LIMIT = 1152921504606846976.0f;   // 2^60
__device__ __forceinline__ bool march_domain(float ex, float ey, float ez, float t_min) {
    return __builtin_fabsf(ex) <= MARCH_LIMIT && __builtin_fabsf(ey) <= MARCH_LIMIT && __builtin_fabsf(ez) <= MARCH_LIMIT &&
           __builtin_fabsf(t_min) <= MARCH_LIMIT;
}
__device__ __forceinline__ float march_t_hi(float t_max) { return __builtin_fminf(t_max, MARCH_LIMIT); }   // fmin: the limit when t_max is NaN

// Workgroups go to the 8 XCDs round-robin (block b runs on XCD b % 8), so consecutive blocks of a spatially
// ordered batch used to land on eight different L2s, each of which saw an eighth of every neighbourhood of
// the tree.  Remapped, an XCD works through chunks of consecutive blocks, the chunks dealt round-robin (so a
// dear region of a skewed batch is shared by all XCDs): one slowly moving neighbourhood per L2.  10 M points
// against the 12.8 MB tree: k = 8 1.756 -> 1.65-1.69 ms per batch, k = 1 1.136 -> 1.08, caller-sorted k = 1
// 0.430 -> 0.37.  (At 7 wavefronts per SIMD and the round's earlier, slower walk a remap had measured as a
// loss: the record fetches were not yet what the kernel waited for.)
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t chunk) {
    // chunks of `chunk` consecutive blocks dealt round-robin: XCD x takes chunks x, x + 8, ...  (the grid is
    // padded to a multiple of 8 * chunk blocks, query_grid; a block beyond the batch finds no point and exits)
    const uint32_t x = b & 7u, l = b >> 3;
    return ((l / chunk) * 8u + x) * chunk + l % chunk;
}

template <int MODE, bool COUNT, bool CHAIN>
__device__ __forceinline__ bool list_pass(const TreeView& T, float px, float py, float pz, float mg, bool active, uint32_t lane,
                                          float& best, uint32_t& bid, float& budget, bool& adv, WaveCounters& wc);

// ----------------------------------------------------------- K1: nearest --
// faux_shader::nearest_neighbor (faux_shader.rs:96-208) for n points, plus
// the primitive index.  One point per lane.  `order` (optional) is a
// permutation that makes consecutive lanes spatially close (Morton order);
// results are written at the original index.
// A point that lies in a cell of the nearest-candidate grid with a list (kernels.hpp CandidateGrid: built for the
// march, but what a cell's list proves -- it holds every primitive that can attain or tie for the minimum anywhere in
// the cell -- is a statement about points, not about rays) is decided on that list, exactly; the wavefront walks the
// tree only for its remaining lanes (outside the padded scene box, cells without a list), if there are any.
template <int MODE, int COUNT, bool CHAIN = false>
__global__ __launch_bounds__(256) void nearest_kernel(TreeView T, const float4* __restrict__ pts,
                                                      const uint32_t* __restrict__ order, uint32_t n,
                                                      float* __restrict__ out_dist, uint32_t* __restrict__ out_prim,
                                                      sdfgpu_stats* __restrict__ st) {
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* stk = smem + wave * T.depth * 64u;
    const uint32_t blk = T.xcd ? xcd_block(blockIdx.x, T.xcd) : blockIdx.x;
    const uint32_t slot = blk * blockDim.x + threadIdx.x;
    const bool alive = slot < n;
    if (__builtin_amdgcn_ballot_w64(alive) == 0ull) return;   // a wavefront of the grid's padding (query_grid)
    const uint32_t src = alive ? (order ? order[slot] : slot) : 0u;
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
    if (alive) p = pts[src];
    float best = __builtin_inff();
    uint32_t bid = SDF_NO_PRIM;
    WaveCounters wc;
    AccNearest acc{best, bid};
    const bool in_dom = in_domain3(p.x, p.y, p.z);
    bool listed = false;
    if constexpr (COUNT == 0) {   // (the counting instantiations count the walk)
        float budget = 0.f;
        bool took = false;
        if (list_pass<MODE, false, CHAIN>(T, p.x, p.y, p.z, 0.f, alive && in_dom, lane, best, bid, budget, took, wc)) listed = took;
    }
    const unsigned long long drive = __builtin_amdgcn_ballot_w64(alive && in_dom && !listed);
    if (drive != 0ull) nn_traverse<MODE, COUNT, true, CHAIN>(T, p.x, p.y, p.z, drive, 0.f, acc, stk, lane, wc);
    best = in_dom ? (listed ? best : acc.best) : __builtin_nanf("");
    bid = in_dom ? (listed ? bid : acc.bid) : SDF_NO_PRIM;
    if (alive) {
        out_dist[src] = best;
        if (out_prim) out_prim[src] = bid;
    }
    if constexpr (COUNT) flush_counters(wc, lane, 0, 0, st);
}


// ---------------------------------------------------------------- K2: kNN --
// k nearest primitives (extension of the k = 1 reference, README.md:7: "based
// off the existing branch-and-prune R-tree kNN algorithm").  Same walk as K1;
// the prune radius is the k-th best so far.  The k best live in registers,
// sorted ascending by (distance, larger index first) -- the order that
// take_better induces -- and are padded with (+inf, SDF_NO_PRIM).
// Each candidate is ONE 64-bit key: high word = the distance as an order-preserving
// unsigned (sign-flipped float bits), low word = 0xFFFFFFFE - index, so that unsigned
// key order IS take_better's order (smaller distance first, larger index first among equal
// distances; the padding (+inf, SDF_NO_PRIM) = (ord(+inf), 0xFFFFFFFF) sorts after any real
// entry).  Inserting is then K compare-exchanges on keys instead of K evaluations of the
// two-field predicate (the offer was ~130 instructions, twice the primitive evaluation).
__device__ __forceinline__ uint32_t ord_f32(float d) {
    const uint32_t u = __builtin_bit_cast(uint32_t, d);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t k) {
    const uint32_t u = k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu);
    return __builtin_bit_cast(float, u);
}
template <int K>
struct AccK {
    static constexpr bool TRACKS = false;
    unsigned long long key[K];
    float cap = __builtin_inff();
    __device__ __forceinline__ void init() {
        const unsigned long long pad = ((unsigned long long)ord_f32(__builtin_inff()) << 32) | 0xFFFFFFFFull;
#pragma unroll
        for (int j = 0; j < K; ++j) key[j] = pad;
    }
    __device__ __forceinline__ void init_nan() {   // a non-finite query point: (NaN, SDF_NO_PRIM) in every slot
        const unsigned long long pad = ((unsigned long long)ord_f32(__builtin_nanf("")) << 32) | 0xFFFFFFFFull;
#pragma unroll
        for (int j = 0; j < K; ++j) key[j] = pad;
    }
    __device__ __forceinline__ float dist(int j) const { return unord_f32((uint32_t)(key[j] >> 32)); }
    __device__ __forceinline__ uint32_t prim(int j) const {
        const uint32_t lo = (uint32_t)key[j];
        return lo == 0xFFFFFFFFu ? SDF_NO_PRIM : 0xFFFFFFFEu - lo;
    }
    __device__ __forceinline__ float radius() const { return __builtin_fminf(dist(K - 1), cap); }
    __device__ __forceinline__ void skipped(float) {}
    __device__ __forceinline__ void offer(float nd, uint32_t nid) {
        const unsigned long long cur = ((unsigned long long)ord_f32(nd) << 32) | (unsigned long long)(0xFFFFFFFEu - nid);
        if (!(cur < key[K - 1])) return;
        // The keys are sorted, so the K comparisons against the NEW key are monotone (false ... false,
        // true ... true) and independent of one another: slot j takes its left neighbour's old key when the
        // new one sorts before that neighbour, the new key when it sorts between them, and keeps its own
        // otherwise.  K compares + 2 selects per slot, no chain from slot to slot (a compare-exchange
        // ladder is 2 compares + 2 selects per slot, each waiting for the slot before it).
        bool lt[K];
#pragma unroll
        for (int j = 0; j < K; ++j) lt[j] = cur < key[j];
#pragma unroll
        for (int j = K - 1; j > 0; --j) key[j] = lt[j - 1] ? key[j - 1] : (lt[j] ? cur : key[j]);
        key[0] = lt[0] ? cur : key[0];
    }
};

template <int MODE, int K, bool COUNT, bool CHAIN = false>
__global__ __launch_bounds__(256) void knn_kernel(TreeView T, const float4* __restrict__ pts,
                                                  const uint32_t* __restrict__ order, uint32_t n, uint32_t k,
                                                  float* __restrict__ out_dist, uint32_t* __restrict__ out_prim,
                                                  sdfgpu_stats* __restrict__ st) {
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* __restrict__ stk = smem + wave * T.depth * 64u;
    const uint32_t blk = T.xcd ? xcd_block(blockIdx.x, T.xcd) : blockIdx.x;
    const uint32_t slot = blk * blockDim.x + threadIdx.x;
    const bool alive = slot < n;
    if (__builtin_amdgcn_ballot_w64(alive) == 0ull) return;   // a wavefront of the grid's padding (query_grid)
    const uint32_t src = alive ? (order ? order[slot] : slot) : 0u;
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
    if (alive) p = pts[src];
    AccK<K> acc;
    acc.init();
    WaveCounters wc;
    const bool in_dom = in_domain3(p.x, p.y, p.z);
    // A point inside the candidate grid starts with the prune radius capped at its cell's bound on the K-th nearest
    // distance (GridCap) instead of +inf: everything that can be among its K nearest, ties included, lies within it,
    // so the result is the same and the first descents no longer wander through subtrees a converged radius would
    // have skipped (caller-sorted k = 8 batch 1.126 -> 0.98 ms with lip r for |p - c|; the exact final radius as the cap
    // would give 0.90).
    {
        if (T.gblock != nullptr) {
            const f32x16 G = load_rec64_s48(T.gblock, 0u);
            const float gx = __builtin_fminf(__builtin_fmaxf((p.x - G[0]) * G[3], -1.f), 65536.f);
            const float gy = __builtin_fminf(__builtin_fmaxf((p.y - G[1]) * G[4], -1.f), 65536.f);
            const float gz = __builtin_fminf(__builtin_fmaxf((p.z - G[2]) * G[5], -1.f), 65536.f);
            const uint32_t ix = (uint32_t)(int)__builtin_floorf(gx), iy = (uint32_t)(int)__builtin_floorf(gy), iz = (uint32_t)(int)__builtin_floorf(gz);
            const bool in = alive && in_dom && ix < rec_bits(G[6]) && iy < rec_bits(G[7]) && iz < rec_bits(G[8]);
            const uint32_t bi = ((iz >> 2) * rec_bits(G[10]) + (iy >> 2)) * rec_bits(G[9]) + (ix >> 2);
            const uint32_t caps_off = rec_bits(G[12]);
            if (in) {
                const GridCap q = *reinterpret_cast<const GridCap*>(reinterpret_cast<const char*>(T.gblock) + caps_off +
                                                                    (size_t)(((bi << 6) | ((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u)) << 3));
                // distance to the cell's centre (formed as the build formed it), rounded up
                const float ex = p.x - (G[0] + ((float)ix + 0.5f) * G[13]), ey = p.y - (G[1] + ((float)iy + 0.5f) * G[14]),
                            ez = p.z - (G[2] + ((float)iz + 0.5f) * G[15]);
                acc.cap = (K <= 8 ? q.d8 : q.d16) + __builtin_amdgcn_sqrtf((ex * ex + ey * ey) + ez * ez) * (T.lip * 1.0001f);
            }
        }
    }
    nn_traverse<MODE, COUNT, true, CHAIN>(T, p.x, p.y, p.z, __builtin_amdgcn_ballot_w64(alive && in_dom), 0.f, acc, stk, lane, wc);
    if (!in_dom) acc.init_nan();
    if (alive) {
        // whole rows as 16-byte stores when the row is exactly K wide (rows are K * 4 bytes apart, so they
        // are aligned when the arrays are)
        bool rows = false;
        if constexpr (K % 4 == 0) rows = k == (uint32_t)K && ((((uintptr_t)out_dist) | ((uintptr_t)out_prim)) & 15u) == 0;
        if (rows) {
            float4* od = reinterpret_cast<float4*>(out_dist + (size_t)src * K);
            uint4* op = reinterpret_cast<uint4*>(out_prim + (size_t)src * K);
#pragma unroll
            for (int j = 0; j + 3 < K; j += 4) {
                od[j / 4] = make_float4(acc.dist(j), acc.dist(j + 1), acc.dist(j + 2), acc.dist(j + 3));
                op[j / 4] = make_uint4(acc.prim(j), acc.prim(j + 1), acc.prim(j + 2), acc.prim(j + 3));
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if ((uint32_t)j < k) {
                    out_dist[(size_t)src * k + j] = acc.dist(j);
                    out_prim[(size_t)src * k + j] = acc.prim(j);
                }
        }
    }
    if constexpr (COUNT) flush_counters(wc, lane, 0, 0, st);
}

// ------------------------------------------------- K3 + K4: march, normal --
// March loop (build-defined; restated by oracle_render):
//   t = t_min; repeat max_steps times: p = o + d*t; (delta, id) = NN(p);
//   if delta < eps_hit -> hit at depth t; t = t + delta; if t > t_max -> miss.
// The previous step's nearest primitive is evaluated first to seed `best`, so
// the walk only has to visit what can still beat it -- and when the bounds
// kept from the last walk prove that nothing can, the walk is skipped for the
// whole wavefront (the result is the same (distance, index) either way).
// Normal (src/main__.rs:151-161): tetrahedron taps of the HIT primitive only.
//
// Ray compaction.  Step counts per ray are heavy-tailed (median 27, 1 % above
// 100 on the 10k scene), so a wavefront that keeps its 64 rays to the end runs
// mostly empty.  The march is therefore cut into PASSES over step ranges
// [b0,b1), [b1,b2), ...: rays still marching at the end of a pass store 16
// bytes of state, compacted per wavefront; a one-workgroup scan turns the
// per-wavefront survivor counts into offsets; the next pass packs survivors,
// in the same spatial order, 64 to a wavefront.  No host synchronisation
// between passes (grids are sized for the worst case, empty wavefronts exit).
struct RayState {
    float t;
    float budget;
    uint32_t last;
    uint32_t slot;   // tile_local * 64 + lane of the pixel
};
static_assert(sizeof(RayState) == 16, "RayState is one 16-byte access");

struct PassArgs {
    uint32_t step_begin, step_end;
    uint32_t final_pass;              // rays alive after step_end - 1 are misses
    const RayState* in_state;         // pass > 0: survivors of the previous pass, [src_wave*64 + rank]
    const uint32_t* in_offsets;       // exclusive scan of the previous pass's per-wave counts
    const uint32_t* in_hdr;           // {number of source waves, total survivors}
    RayState* out_state;
    uint32_t* out_counts;
    const uint32_t* tile_order;       // first pass: wave w renders tile tile_order[w] (null: tile w)
    uint32_t* tile_cost;              // first pass: min(steps the wave ran, 255) per tile (null: not recorded)
    const float4* eye;                // first pass, t_min = 0: (best, bid bits, lb_key, second) of the walk from the eye, or null
};

// The first march step of every ray of a frame stands at the same point when t_min = 0: the eye.  Its walk
// (45 internal nodes and 3 primitives per wavefront on the bench frame, 4 % of a frame's visits) is made once,
// by one wavefront whose 64 lanes all hold that point -- identical needs, so the nodes visited, the bounds kept
// and the result are what each tile's own first walk would have produced.
template <int MODE, bool CHAIN = false>
__global__ __launch_bounds__(64) void eye_walk_kernel(TreeView T, float ex, float ey, float ez, float4* __restrict__ out) {
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63u;
    AccTracked acc{__builtin_inff(), SDF_NO_PRIM, __builtin_inff(), __builtin_inff()};
    WaveCounters wc;
    if (!march_domain(ex, ey, ez, 0.f)) return;   // no ray of such a frame marches (march_domain): nobody reads *out
    nn_traverse<MODE, 0, true, CHAIN>(T, ex, ey, ez, ~0ull, T.skip_ext, acc, smem, lane, wc);
    if (lane == 0) *out = make_float4(acc.best, __builtin_bit_cast(float, acc.bid), acc.lb_key, acc.second);
}

// A march step on the nearest-candidate grid (kernels.hpp CandidateGrid; built at upload, grid.hip) instead of a
// tree walk.  Every marching lane looks up its point's cell; a cell with a list names, as a 32-bit mask into its
// brick's list of primitive ids, EVERY primitive that can attain or tie for the minimum anywhere in the cell, so
// evaluating those -- with the same leaf arithmetic, in any order, under the same (distance, later index) rule --
// gives the lane the result the exhaustive walk would.  The evaluation is wave-uniform like the walk's: the lanes'
// masks are OR-ed per brick, each set bit's record arrives by scalar load and is evaluated for all 64 lanes (a
// lane sees a superset of its own list, which cannot change its minimum).  ~6 primitive evaluations per pass
// against ~30 box tests + ~4 evaluations for a walk on the bench frame.  What the lane keeps for certifying its
// next steps: `second` over everything evaluated, and the cell's lower bound on everything not in its list.
// Returns false (and touches nothing) when no marching lane has a list: the caller walks.
template <int MODE, bool COUNT, bool CHAIN>
__device__ __forceinline__ bool list_pass(const TreeView& T, float px, float py, float pz, float mg, bool active, uint32_t lane,
                                          float& best, uint32_t& bid, float& budget, bool& adv, WaveCounters& wc) {
    if (T.gblock == nullptr) return false;
    const f32x16 G = load_rec64_s48(T.gblock, 0u);   // GridBlock: lo[3], inv_cell[3], n[3], n[0]/4, n[1]/4, bricks_off
    // cell of the point: floor((p - lo) / cell) per axis, out-of-range and non-numbers fail the unsigned compare
    const float gx = __builtin_fminf(__builtin_fmaxf((px - G[0]) * G[3], -1.f), 65536.f);
    const float gy = __builtin_fminf(__builtin_fmaxf((py - G[1]) * G[4], -1.f), 65536.f);
    const float gz = __builtin_fminf(__builtin_fmaxf((pz - G[2]) * G[5], -1.f), 65536.f);
    const uint32_t ix = (uint32_t)(int)__builtin_floorf(gx), iy = (uint32_t)(int)__builtin_floorf(gy), iz = (uint32_t)(int)__builtin_floorf(gz);
    const bool in = ((uint32_t)active & (uint32_t)(ix < rec_bits(G[6])) & (uint32_t)(iy < rec_bits(G[7])) & (uint32_t)(iz < rec_bits(G[8]))) != 0u;
    const uint32_t bi = ((iz >> 2) * rec_bits(G[10]) + (iy >> 2)) * rec_bits(G[9]) + (ix >> 2);
    // both tables live in the block's own allocation: cells right behind it, brick ids at bricks_off
    const char* __restrict__ gbase = reinterpret_cast<const char*>(T.gblock);
    const uint32_t bricks_off = rec_bits(G[11]);
    // The id row of the brick the first marching lane inside the grid stands in is requested TOGETHER with the cells:
    // its address needs the brick index only, and it is nearly always the first round's brick, so that round's row
    // costs no memory round trip of its own (cell -> row -> records were three dependent fetches; frame 1.311 ->
    // 1.282 ms.  Requesting the next distinct brick's row as well: 1.312; warming the scalar cache with all records
    // of a round before evaluating the first: 1.348 -- the extra scalar work costs more than the waits it removes).
    const unsigned long long m_in = __builtin_amdgcn_ballot_w64(in);
    if (m_in == 0ull) return false;
    const uint32_t sb0 = (uint32_t)__builtin_amdgcn_readlane((int)bi, (int)__builtin_ctzll(m_in));
    const uint32_t vids0 = *reinterpret_cast<const uint32_t*>(gbase + (bricks_off + sb0 * (GRID_BRICK_IDS * 4u) + ((lane & (GRID_BRICK_IDS - 1u)) << 2)));
    GridCell gc{0u, 0.f};
    if (in) gc = *reinterpret_cast<const GridCell*>(gbase + sizeof(GridBlock) +
                                                    (size_t)(((bi << 6) | ((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u)) << 3));
    const bool has_list = gc.mask != 0u;
    // brick of a lane that still waits for its brick's round; lanes without a list wait for none
    uint32_t bw = has_list ? bi : 0xFFFFFFFFu;
    unsigned long long pend = __builtin_amdgcn_ballot_w64(bw != 0xFFFFFFFFu);
    if (pend == 0ull) return false;
    if constexpr (COUNT) { wc.wave_lists += 1; wc.lane_list_steps += has_list ? 1u : 0u; }
    AccTracked acc{best, bid, __builtin_inff(), __builtin_inff()};
    // Neighbouring bricks share most of their candidates, and the lanes of a wavefront stand in two or three bricks:
    // lane k of `done` keeps the k-th primitive this pass has evaluated, and an id found there is not evaluated again
    // (12.3 -> ~6 evaluations per pass on the bench frame).
    uint32_t done = 0xFFFFFFFFu, n_done = 0;
    uint32_t chain_first = 0xFFFFFFFFu;   // CHAIN: record range whose end point (cx, cy, cz) is at hand (a leaf without a chain: first = 0, count = 0)
    asm volatile("" : "+v"(chain_first));
    float cx = px, cy = py, cz = pz;
    do {   // one round per brick the listed lanes stand in (2-3 on the bench frame)
        const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)bw, (int)__builtin_ctzll(pend));
        // the brick's id list: ONE vector load (lane j takes entry j), entries then come by v_readlane -- a scalar
        // load per entry put two dependent memory round trips in front of every evaluation
        uint32_t vids = vids0;
        if (sb != sb0) vids = *reinterpret_cast<const uint32_t*>(gbase + (bricks_off + sb * (GRID_BRICK_IDS * 4u) + ((lane & (GRID_BRICK_IDS - 1u)) << 2)));
        const bool mine = bw == sb;
        uint32_t m = __builtin_amdgcn_readfirstlane(__ockl_wfred_or_u32(mine ? gc.mask : 0u));
        bw = mine ? 0xFFFFFFFFu : bw;
        while (m != 0u) {
            const uint32_t b = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)vids, (int)b);
            if (__builtin_amdgcn_ballot_w64(done == id) != 0ull) continue;
            done = (lane == n_done) ? id : done;
            n_done += 1;
            if constexpr (CHAIN) {
                // two leaves of one parent share their whole chain (one record range): the second finds the point the
                // first one left behind
                uint32_t first, cnt;
                load_u32x2_scalar(T.chain, id << 3, first, cnt);
                // (kept in a VGPR between candidates: the march kernel has no SGPR to spare across the rounds)
                if (first != (uint32_t)__builtin_amdgcn_readfirstlane((int)chain_first)) {
                    chain_first = first;
                    asm volatile("" : "+v"(chain_first));
                    const uint32_t mats = (T.n_leaf * 8u + 63u) & ~63u;
                    cx = px; cy = py; cz = pz;
                    for (uint32_t j = 0; j < cnt; ++j) {
                        const f32x16 M = load_rec64_s48_at(T.chain, mats + ((first + j) << 6));
                        const float nx = ((M[0] * cx + M[1] * cy) + M[2] * cz) + M[3];
                        const float ny = ((M[4] * cx + M[5] * cy) + M[6] * cz) + M[7];
                        const float nz = ((M[8] * cx + M[9] * cy) + M[10] * cz) + M[11];
                        cx = nx; cy = ny; cz = nz;
                    }
                }
            }
            const f32x16 L = load_rec64_s48(T.leaf, id);
            acc.offer(leaf_eval_rec(L, cx, cy, cz), id);   // (not the chain path: cx, cy, cz are the point itself)
            if constexpr (COUNT) {
                wc.wave_list_evals += 1;
                const uint32_t want = (has_list && bi == sb && ((gc.mask >> b) & 1u)) ? 1u : 0u;
                if (rec_bits(L[15]) == SDF_PRIM_SPHERE) { wc.lane_sphere += has_list ? 1u : 0u; wc.lane_sphere_needed += want; }
                else { wc.lane_box += has_list ? 1u : 0u; wc.lane_box_needed += want; }
            }
        }
        pend = __builtin_amdgcn_ballot_w64(bw != 0xFFFFFFFFu);
    } while (pend != 0ull);
    // (the lane mask of "has a list" is formed again here rather than kept in an SGPR pair across the rounds: the
    // march kernel sits one SGPR below the limit of its eighth wavefront per SIMD)
    uint32_t gm = gc.mask;
    asm volatile("" : "+v"(gm));
    const bool listed = gm != 0u;
    best = listed ? acc.best : best;
    bid = listed ? acc.bid : bid;
    // everything evaluated other than the winner is >= second; everything not in the lane's own list is >= the cell's bound
    const float lb = __builtin_fminf(gc.lb, acc.second) - mg;
    budget = listed ? lb : budget;
    adv = listed;
    return true;
}

template <int MODE, bool COUNT, bool FIRST, bool FAR, bool CHAIN = false>
__global__ __launch_bounds__(256) void render_pass_kernel(TreeView T, sdfgpu_camera cam, uint32_t width, uint32_t height,
                                                          sdfgpu_march_params mp, uint32_t tile_first,
                                                          uint32_t tile_stride, uint32_t tile_count, uint32_t row_major,
                                                          PassArgs pa, uint8_t* __restrict__ o_hit,
                                                          float* __restrict__ o_depth, float* __restrict__ o_normal,
                                                          uint32_t* __restrict__ o_prim, sdfgpu_stats* __restrict__ st) {
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* stk = smem + wave * T.depth * 64u;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wave);

    float t = mp.t_min, budget = 0.f;
    uint32_t last = SDF_NO_PRIM, slot = 0;
    bool alive;
    if constexpr (FIRST) {
        if (wave_id >= tile_count) return;
        uint32_t tl = wave_id;
        if (pa.tile_order) {   // longest tiles first; anything out of range falls back to the identity
            tl = __builtin_amdgcn_readfirstlane(pa.tile_order[wave_id]);
            tl = tl < tile_count ? tl : wave_id;
        }
        slot = tl * 64u + lane;
        alive = true;
    } else {
        // header / offsets are clamped to what the scratch can hold, so that even
        // corrupted pass state (two frames sharing one scratch) cannot index outside it
        uint32_t n_src = pa.in_hdr[0], total = pa.in_hdr[1];
        n_src = n_src < tile_count ? n_src : tile_count;
        total = total < tile_count * 64u ? total : tile_count * 64u;
        if (n_src == 0u || wave_id * 64u >= total) return;
        const uint32_t j = wave_id * 64u + lane;
        alive = j < total;
        if (alive) {
            // source wave w: offsets[w] <= j < offsets[w+1]
            uint32_t lo = 0, hi = n_src;
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (pa.in_offsets[mid] <= j) lo = mid; else hi = mid;
            }
            const uint32_t rank = j - pa.in_offsets[lo];
            if (rank < 64u) {
                const RayState rs = pa.in_state[(size_t)lo * 64u + rank];
                t = rs.t; budget = rs.budget; last = rs.last;
                slot = rs.slot < tile_count * 64u ? rs.slot : 0u;
                if (last != SDF_NO_PRIM && last >= T.n_leaf) last = SDF_NO_PRIM;
            } else {
                alive = false;
            }
        }
    }
    const uint32_t tile_local = slot >> 6, pl = slot & 63u;
    const uint32_t tile = tile_first + tile_local * tile_stride;
    const uint32_t tiles_x = (width + 7u) >> 3;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t x = tx * 8u + (pl & 7u), y = ty * 8u + (pl >> 3);
    const bool inside = x < width && y < height;
    const bool mine = alive;          // this lane carries a ray (or, FIRST, an edge slot) into the pass
    alive = alive && inside;

    // ray through the pixel centre: pw = (origin + du*x) + dv*y ; d = (pw - eye)/|pw - eye|
    const float fx = (float)x, fy = (float)y;
    const float ox = cam.eye[0], oy = cam.eye[1], oz = cam.eye[2];
    float dx = ((cam.origin[0] + cam.du[0] * fx) + cam.dv[0] * fy) - ox;
    float dy = ((cam.origin[1] + cam.du[1] * fx) + cam.dv[1] * fy) - oy;
    float dz = ((cam.origin[2] + cam.du[2] * fx) + cam.dv[2] * fy) - oz;
    const float dl = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
    dx = dx / dl;
    dy = dy / dl;
    dz = dz / dl;
    alive = alive && in_domain3(dx, dy, dz) && march_domain(ox, oy, oz, mp.t_min);   // otherwise a miss outright (march_domain)
    const float t_hi = march_t_hi(mp.t_max);

    float px = 0.f, py = 0.f, pz = 0.f;
    // `budget`: proven lower bound, at the current point, on the distance of every
    // primitive other than `last`.  While last's own (exactly evaluated) distance
    // stays below it by more than the rounding margin, `last` is the unique
    // nearest primitive and the step needs no walk ("certified").
    //
    // Walk deferral.  A walk is wave-uniform: it costs the union of what its driving lanes need,
    // so a walk for three lanes costs almost what a walk for sixty-four neighbours costs.  The
    // lanes of a wavefront therefore do NOT step in lockstep: in every iteration the lanes that
    // can certify their step take it (pure VALU on the cached record of `last`), the others
    // hold their position, and only when no marching lane can certify does the wavefront walk --
    // once, for all of them.  Every lane still performs exactly its own sequence of
    // (point, nearest) steps, so results are unchanged; on the bench frame the wavefronts walk
    // 1.18 M times instead of 1.81 M and visit 35 M nodes instead of 69 M (tools/sim/wave_sim.cpp
    // reproduces the kernel's counters exactly and was used to choose the policy).
    const float ext = T.skip_ext;
    WaveCounters wc;
    uint32_t s = pa.step_begin;                  // this lane's step index (lanes no longer share one)
    uint32_t n_walks = 0, n_cheap = 0, n_lists = 0;
    // Lane state as data, not control flow: 0 = marching, 1 = hit, 2 = ended without a hit (left the
    // scene, off the image, not a ray).  The loop body is straight-line selects; the only branches are
    // wave-uniform (walk or not).
    uint32_t state = alive ? 0u : 2u;
    // record of `last`, kept in registers between walks: certified steps touch no memory
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0, c3 = c0;
    if (!CHAIN && alive && last != SDF_NO_PRIM) {
        const float4* __restrict__ q = reinterpret_cast<const float4*>(T.leaf + last);
        c0 = q[0]; c1 = q[1]; c2 = q[2]; c3 = q[3];
    }
    if (CHAIN && alive && last != SDF_NO_PRIM) {   // chain path: the seed's {first record, count} instead
        const float2 fc = reinterpret_cast<const float2*>(T.chain)[last];
        c0.x = fc.x; c0.y = fc.y;
    }
    // FIRST pass with the eye's walk at hand: taken instead of this wavefront's own first walk when every
    // marching lane really stands at the eye (a degenerate ray direction would put a lane elsewhere)
    bool eye_step = false;
    if constexpr (FIRST) eye_step = pa.eye != nullptr;
    for (;;) {
        // Wave masks are formed from the ballots of the individual compares: a ballot of a compare IS the
        // compare's own mask, while a ballot of a combined flag is materialised per lane and compared again.
        const bool a_state = state == 0u, a_steps = s < pa.step_end;
        const bool active = a_state & a_steps;                       // still marching in this pass
        const unsigned long long m_active = __builtin_amdgcn_ballot_w64(a_state) & __builtin_amdgcn_ballot_w64(a_steps);
        if (m_active == 0ull) break;
        // a lane that is not marching recomputes the point it stopped at (t is no longer advanced)
        px = ox + dx * t;
        py = oy + dy * t;
        pz = oz + dz * t;
        const float mg = prune_margin<MODE, CHAIN>(T, px, py, pz);
        const bool seeded = last != SDF_NO_PRIM;
        float seed;
        if constexpr (CHAIN) {
            // A seed's value goes through its whole chain, whose records cannot be kept in registers as a plain leaf's
            // are: every lane walks its own seed's chain with vector loads (leaf_value_lane_chain; c0.x / c0.y keep the
            // lane's {first record, count}).  (First version: the distinct seeds of the marching lanes one after the
            // other, wave-uniformly with scalar loads -- two or three whole chains per iteration instead of one: the
            // crate-buffer frame 3.59 -> 2.18 ms isolated.  The list pass stays wave-uniform: every lane evaluating its
            // own list through vector loads measured 4.6 ms -- there the lanes' candidates differ and so do their lines.)
            seed = leaf_value_lane_chain(T, last, __builtin_bit_cast(uint32_t, c0.x), __builtin_bit_cast(uint32_t, c0.y), active & seeded, px, py, pz);
        } else {
            const float lx = ((c0.x * px + c0.y * py) + c0.z * pz) + c0.w;
            const float ly = ((c1.x * px + c1.y * py) + c1.z * pz) + c1.w;
            const float lz = ((c2.x * px + c2.y * py) + c2.z * pz) + c2.w;
            seed = prim_value_lane(__builtin_bit_cast(uint32_t, c3.w), c3.x, c3.y, c3.z, lx, ly, lz);
        }
        float best = seeded ? seed : __builtin_inff();
        uint32_t bid = last;
        const bool c_bound = (best + 2.0f * mg) < budget;
        const bool certified = active & seeded & c_bound;
        if constexpr (COUNT) wc.wave_steps += 1;
        bool adv;
        const bool cheap = (m_active & __builtin_amdgcn_ballot_w64(seeded) & __builtin_amdgcn_ballot_w64(c_bound)) != 0ull;
        if (cheap) {
            adv = certified;   // the others hold their position
            n_cheap += 1;
        } else if (list_pass<MODE, COUNT, CHAIN>(T, px, py, pz, mg, active, lane, best, bid, budget, adv, wc)) {
            // no marching lane can certify, but some stand in a grid cell with a candidate list: those lanes took
            // their step on the list (list_pass); the others (outside the grid, cells without a list) hold their
            // position until nobody can step otherwise, then the wavefront walks for them
            n_lists += 1;
        } else {
            // no marching lane can certify or has a list: one walk for all of them
            const unsigned long long dmask = m_active;
            AccTracked acc{best, bid, __builtin_inff(), __builtin_inff()};
            bool shared = false;
            if constexpr (FIRST) {
                if (eye_step) {
                    const bool at_eye = (px == ox) & (py == oy) & (pz == oz) & !seeded;
                    shared = (__builtin_amdgcn_ballot_w64(at_eye) & dmask) == dmask;
                }
            }
            if (shared) {
                const float4 e = *pa.eye;
                acc.best = e.x;
                acc.bid = __builtin_bit_cast(uint32_t, e.y);
                acc.lb_key = e.z;
                acc.second = e.w;
            } else {
                if constexpr (COUNT) wc.wave_walks += 1;
                nn_traverse<MODE, COUNT, FAR, CHAIN>(T, px, py, pz, dmask, ext, acc, stk, lane, wc);
            }
            best = acc.best;
            bid = acc.bid;
            // a bound, not a result: the hardware square root (1 ulp) is inside the 0.9996
            float lb = __builtin_amdgcn_sqrtf(acc.lb_key);
            if constexpr (MODE != SDFGPU_BOUNDS_OBB) lb *= T.inv_bscale;
            lb = __builtin_fminf(lb * 0.9996f, acc.second) - mg;
            // only for the lanes the walk was for: a lane that has finished this pass's steps keeps the
            // budget it carries to the next pass (the walk saw the point it stopped at, not its next one)
            budget = active ? lb : budget;
            adv = active;
            n_walks += 1;
        }
        if constexpr (COUNT) {
            if (adv & seeded) {   // the seed evaluation of this step
                if (__builtin_bit_cast(uint32_t, c3.w) == SDF_PRIM_SPHERE) { wc.lane_sphere += 1; wc.lane_sphere_needed += 1; }
                else { wc.lane_box += 1; wc.lane_box_needed += 1; }
            }
        }
        const bool hit_now = adv & (best < mp.eps_hit);
        const bool go = adv & !hit_now;
        const float tn = t + best;
        const float bn = (budget - __builtin_fabsf(best) * T.lip) - mg;
        t = go ? tn : t;
        budget = go ? bn : budget;
        last = adv ? bid : last;
        s += adv ? 1u : 0u;
        state = hit_now ? 1u : ((go & !(tn <= t_hi)) ? 2u : state);
        eye_step = false;   // only the very first iteration of a ray's march can stand at the eye
        if (!CHAIN && !cheap) {   // the walk may have changed `last`: fetch its record once
            if ((state == 0u) & (s < pa.step_end) & (last != SDF_NO_PRIM)) {   // (an empty tree has no nearest primitive)
                const float4* __restrict__ q = reinterpret_cast<const float4*>(T.leaf + last);
                c0 = q[0]; c1 = q[1]; c2 = q[2]; c3 = q[3];
            }
        }
        if (CHAIN && !cheap) {
            if ((state == 0u) & (s < pa.step_end) & (last != SDF_NO_PRIM)) {
                const float2 fc = reinterpret_cast<const float2*>(T.chain)[last];
                c0.x = fc.x; c0.y = fc.y;
            }
        }
    }
    const uint32_t steps = s - pa.step_begin;
    const bool hit = state == 1u;
    const uint32_t hit_prim = hit ? last : SDF_NO_PRIM;
    alive = state == 0u;   // reached the end of this pass still marching

    if constexpr (FIRST) {
        if (pa.tile_cost && lane == 0) {
            // cost of the tile for the next frame's longest-first order: walks dominate, certified steps are ~1/16 of one
            const uint32_t ran = n_walks + (n_lists >> 1) + (n_cheap >> 4);
            pa.tile_cost[slot >> 6] = ran < 255u ? ran : 255u;
        }
    }
    float nx = 0.f, ny = 0.f, nz = 0.f;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
    if constexpr (CHAIN) {   // the four taps through the hit primitive's chain, per lane (as the seeds above)
        const float h = mp.normal_h;
        uint32_t hf = 0u, hc = 0u;
        if (hit) {
            const uint2 fc = reinterpret_cast<const uint2*>(T.chain)[hit_prim];
            hf = fc.x; hc = fc.y;
        }
        f0 = leaf_value_lane_chain(T, hit_prim, hf, hc, hit, px + h, py - h, pz - h);
        f1 = leaf_value_lane_chain(T, hit_prim, hf, hc, hit, px - h, py - h, pz + h);
        f2 = leaf_value_lane_chain(T, hit_prim, hf, hc, hit, px - h, py + h, pz - h);
        f3 = leaf_value_lane_chain(T, hit_prim, hf, hc, hit, px + h, py + h, pz + h);
    }
    if (hit) {
        if constexpr (!CHAIN) {
            const GpuLeaf* L = T.leaf + hit_prim;
            const float h = mp.normal_h;
            // k.xyy, k.yyx, k.yxy, k.xxx with k = (1,-1)
            f0 = leaf_eval(L, px + h, py - h, pz - h);
            f1 = leaf_eval(L, px - h, py - h, pz + h);
            f2 = leaf_eval(L, px - h, py + h, pz - h);
            f3 = leaf_eval(L, px + h, py + h, pz + h);
        }
        const float sx = ((f0 - f1) - f2) + f3;
        const float sy = ((-f0 - f1) + f2) + f3;
        const float sz = ((-f0 + f1) - f2) + f3;
        const float nl = __builtin_sqrtf((sx * sx + sy * sy) + sz * sz);
        if (nl > 0.f) {
            nx = sx / nl;
            ny = sy / nl;
            nz = sz / nl;
        }
    }

    // rays that go on to the next pass: 16 bytes of state, compacted per wavefront
    const bool carry = alive && !pa.final_pass;
    if (!pa.final_pass) {
        const unsigned long long cm = __builtin_amdgcn_ballot_w64(carry);
        if (carry) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
            RayState rs;
            rs.t = t; rs.budget = budget; rs.last = last; rs.slot = slot;
            pa.out_state[(size_t)wave_id * 64u + rank] = rs;
        }
        if (lane == 0) pa.out_counts[wave_id] = (uint32_t)__popcll(cm);
    }
    // rays that ended in this pass (hit, left the scene, or ran out of steps) write their pixel
    if (mine && !carry) {
        size_t o;
        bool wr;
        if (row_major) { o = (size_t)y * width + x; wr = inside; }
        else { o = slot; wr = true; }
        if (wr) {
            o_hit[o] = hit ? 1 : 0;
            if (o_depth) o_depth[o] = hit ? t : __builtin_inff();
            if (o_normal) { o_normal[3 * o] = nx; o_normal[3 * o + 1] = ny; o_normal[3 * o + 2] = nz; }
            if (o_prim) o_prim[o] = hit_prim;
        }
    }
    if constexpr (COUNT) flush_counters(wc, lane, steps, (FIRST && inside) ? 1u : 0u, st);
}

// Exclusive scan of the per-wavefront survivor counts of one pass (one
// workgroup; at most a few hundred thousand entries).  n comes from the
// previous header (number of waves that could have written a count).
__global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t* __restrict__ counts, uint32_t n_first,
                                                           const uint32_t* __restrict__ prev_hdr,
                                                           uint32_t* __restrict__ offsets, uint32_t* __restrict__ hdr) {
    __shared__ uint32_t part[1024];
    uint32_t n = prev_hdr ? (prev_hdr[1] + 63u) / 64u : n_first;
    n = n < n_first ? n : n_first;   // never beyond the scratch (n_first = tile_count)
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n + 1023u) / 1024u;
    const uint32_t lo = tid * chunk, hi = lo + chunk < n ? lo + chunk : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += counts[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;   // exclusive prefix of this thread's chunk
    for (uint32_t i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (tid == 1023u) {
        offsets[n] = part[1023];
        hdr[0] = n;
        hdr[1] = part[1023];
    }
}

// Tile-compact -> row-major scatter (framebuffer assembly after the gather).
__global__ __launch_bounds__(256) void untile_kernel(uint32_t width, uint32_t height, uint32_t tile_first,
                                                     uint32_t tile_stride, uint32_t tile_count,
                                                     const uint8_t* __restrict__ t_hit, const float* __restrict__ t_depth,
                                                     const float* __restrict__ t_normal, const uint32_t* __restrict__ t_prim,
                                                     uint8_t* __restrict__ o_hit, float* __restrict__ o_depth,
                                                     float* __restrict__ o_normal, uint32_t* __restrict__ o_prim) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile_local = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile_local >= tile_count) return;
    const uint32_t tile = tile_first + tile_local * tile_stride;
    const uint32_t tiles_x = (width + 7u) >> 3;
    const uint32_t x = (tile % tiles_x) * 8u + (lane & 7u), y = (tile / tiles_x) * 8u + (lane >> 3);
    if (x >= width || y >= height) return;
    const size_t s = (size_t)tile_local * 64u + lane, o = (size_t)y * width + x;
    o_hit[o] = t_hit[s];
    if (o_depth) o_depth[o] = t_depth[s];
    if (o_normal) { o_normal[3 * o] = t_normal[3 * s]; o_normal[3 * o + 1] = t_normal[3 * s + 1]; o_normal[3 * o + 2] = t_normal[3 * s + 2]; }
    if (o_prim) o_prim[o] = t_prim[s];
}

// The same scatter for a whole frame whose tiles were dealt round-robin over `world` ranks, in ONE launch
// (the assembling rank of sdfgpu_render_frame_comm: one launch per frame instead of one per rank, which
// matters when a frame's share is a quarter of a millisecond).  Rank p's packed buffer is
// [depth | normal | prim | hit] planes of `px` pixels at gathered + p * nbytes; the assembling rank's
// own share is read from `mine`.
__global__ __launch_bounds__(256) void untile_gathered_kernel(uint32_t width, uint32_t height, uint32_t world, uint32_t self,
                                                              uint32_t n_tiles, size_t px, size_t nbytes,
                                                              const char* __restrict__ gathered, const char* __restrict__ mine,
                                                              uint8_t* __restrict__ o_hit, float* __restrict__ o_depth,
                                                              float* __restrict__ o_normal, uint32_t* __restrict__ o_prim) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const uint32_t p = tile % world, tile_local = tile / world;
    const uint32_t tiles_x = (width + 7u) >> 3;
    const uint32_t x = (tile % tiles_x) * 8u + (lane & 7u), y = (tile / tiles_x) * 8u + (lane >> 3);
    if (x >= width || y >= height) return;
    const char* base = p == self ? mine : gathered + (size_t)p * nbytes;
    const float* t_depth = reinterpret_cast<const float*>(base);
    const float* t_normal = reinterpret_cast<const float*>(base + px * 4u);
    const uint32_t* t_prim = reinterpret_cast<const uint32_t*>(base + px * 16u);
    const uint8_t* t_hit = reinterpret_cast<const uint8_t*>(base + px * 20u);
    const size_t s = (size_t)tile_local * 64u + lane, o = (size_t)y * width + x;
    o_hit[o] = t_hit[s];
    if (o_depth) o_depth[o] = t_depth[s];
    if (o_normal) { o_normal[3 * o] = t_normal[3 * s]; o_normal[3 * o + 1] = t_normal[3 * s + 1]; o_normal[3 * o + 2] = t_normal[3 * s + 2]; }
    if (o_prim) o_prim[o] = t_prim[s];
}

// ------------------------------------------------------- general path -----
// Any union tree the reference state machine accepts (non-identity internal
// trans_inverse, projective rows): the literal point chain of
// faux_shader.rs:160-178 -- every node's 4x4 trans_inverse applied to the
// 4-component point handed down by its parent (glam Mat4*Vec4 order), one
// rounded mat-vec per level, primitives evaluated on xyz.  No pruning: the
// bounds of such buffers live in frames the children are not re-based into
// (SURVEY C-3), so the walk is the exhaustive one, Theta(N) per query like the
// reference's own.  Wave-uniform node index, per-level points in LDS.
struct P4 { float x, y, z, w; };

__device__ __forceinline__ P4 mat4_mul(const float* __restrict__ c, P4 p) {
    P4 r;
    r.x = ((c[0] * p.x + c[4] * p.y) + c[8] * p.z) + c[12] * p.w;
    r.y = ((c[1] * p.x + c[5] * p.y) + c[9] * p.z) + c[13] * p.w;
    r.z = ((c[2] * p.x + c[6] * p.y) + c[10] * p.z) + c[14] * p.w;
    r.w = ((c[3] * p.x + c[7] * p.y) + c[11] * p.z) + c[15] * p.w;
    return r;
}

// Box frame (build-defined extension, primitive op 5; the crate carries dimension + thickness but no
// formula, elements.rs:120-152): iq's sdBoxFrame, one rounded operation per step, restated by
// oracle.cpp::box_frame.
__device__ __forceinline__ float bar_value(float a, float b, float c) {
    const float ma = a > 0.f ? a : 0.f, mb = b > 0.f ? b : 0.f, mc = c > 0.f ? c : 0.f;
    const float l = __builtin_sqrtf((ma * ma + mb * mb) + mc * mc);
    return l + __builtin_fminf(__builtin_fmaxf(a, __builtin_fmaxf(b, c)), 0.0f);
}
__device__ __forceinline__ float prim_value_gen(uint32_t kind, const float* __restrict__ pr, float lx, float ly, float lz) {
    if (kind == SDF_PRIM_BOX_FRAME) {
        const float e = pr[3];
        const float px = __builtin_fabsf(lx) - pr[0], py = __builtin_fabsf(ly) - pr[1], pz = __builtin_fabsf(lz) - pr[2];
        const float qx = __builtin_fabsf(px + e) - e, qy = __builtin_fabsf(py + e) - e, qz = __builtin_fabsf(pz + e) - e;
        const float d0 = bar_value(px, qy, qz), d1 = bar_value(qx, py, qz), d2 = bar_value(qx, qy, pz);
        return min_nn(min_nn(d0, d1), d2);
    }
    return prim_value(kind, pr[0], pr[1], pr[2], lx, ly, lz);
}

// downtree_dispatch (faux_shader.rs:69-82): union hands the point down; clone folds it onto the lattice,
// p - d * clamp(round(p / d), lo, hi) per component (round = half to even, glam's SSE2 recipe).  The w
// lane: d.w = lo.w = hi.w = 0, so 1/0 = inf is clamped to 0 and w stays what it was.
__device__ __forceinline__ float fold1(float p, float d, float lo, float hi) {
    float q = __builtin_rintf(p / d);
    q = q > lo ? q : lo;    // _mm_max_ps(q, lo): second operand when q is NaN
    q = q < hi ? q : hi;
    return p - d * q;
}
__device__ __forceinline__ P4 down_point(const GpuGenNode* __restrict__ N, P4 r) {
    if (N->op != GEN_OP_CLONE) return r;
    P4 o;
    o.x = fold1(r.x, N->clone[0], N->clone[3], N->clone[6]);
    o.y = fold1(r.y, N->clone[1], N->clone[4], N->clone[7]);
    o.z = fold1(r.z, N->clone[2], N->clone[5], N->clone[8]);
    o.w = fold1(r.w, 0.f, 0.f, 0.f);
    return o;
}

// only: SDF_NO_PRIM = every primitive, else just that one (normal taps).
template <class Acc>
__device__ __forceinline__ void walk_general(const GpuGenNode* __restrict__ nodes, uint32_t n, P4 p, uint32_t only,
                                             Acc& acc, float4* __restrict__ stk, uint32_t lane) {
    stk[lane] = make_float4(p.x, p.y, p.z, p.w);
    for (uint32_t i = 0; i < n; ++i) {
        const GpuGenNode* __restrict__ N = nodes + i;
        const uint32_t lv = N->level;
        const float4 q4 = stk[(lv - 1u) * 64u + lane];
        const P4 r = mat4_mul(N->m, P4{q4.x, q4.y, q4.z, q4.w});
        if (N->is_leaf) {
            const float d = prim_value_gen(N->kind, N->param, r.x, r.y, r.z);
            if (only == SDF_NO_PRIM || only == N->prim) acc.offer(d, N->prim);
        } else {
            const P4 c = down_point(N, r);   // downtree_dispatch: union [p, p]; clone [lattice point, 0]
            stk[lv * 64u + lane] = make_float4(c.x, c.y, c.z, c.w);
        }
    }
}

// The same walk with the reference's post-order combine carried out for real (uptree_dispatch,
// faux_shader.rs:84-94, over the frames of :141-151): needed as soon as a combine is not a plain
// minimum, i.e. when a union's smooth radius is in effect (SDFGPU_EXT_SMOOTH_UNION).  Every node writes
// its (distance, index) into slot `slot` of its parent's level; an internal node is combined right after
// the last node of its subtree.  No pruning, so the order of evaluation is the same for every lane and
// the only per-lane state is the point and the two results of each level, in LDS.
//   smooth union: min(a,b) - h*h*k/4 with h = max(k - |a-b|, 0)/k; index of the smaller operand, ties to
//   the right like the plain union (`if a < b {a} else {b}`, faux_shader.rs:27-29).
struct EvalStacks {
    float4* pt;     // [levels + 1][64]
    float2* d;      // [levels + 2][64]
    uint2* id;      // [levels + 2][64]
    float* cut;     // [levels + 2][64] (walk_eval_pruned)
    uint32_t* ctl;  // [64]: per level, the child that waits (pruned walks)
};
__device__ __forceinline__ void walk_eval(const GpuGenNode* __restrict__ nodes, uint32_t n, P4 p, const EvalStacks& S,
                                          uint32_t lane, float& out_d, uint32_t& out_id) {
    S.pt[lane] = make_float4(p.x, p.y, p.z, p.w);
    S.d[64u + lane] = make_float2(__builtin_inff(), __builtin_inff());     // frame of level 1 (faux_shader.rs:102-106)
    S.id[64u + lane] = make_uint2(SDF_NO_PRIM, SDF_NO_PRIM);
    for (uint32_t i = 0; i < n; ++i) {
        const GpuGenNode* __restrict__ N = nodes + i;
        const uint32_t lv = N->level;
        const float4 q4 = S.pt[(lv - 1u) * 64u + lane];
        const P4 r = mat4_mul(N->m, P4{q4.x, q4.y, q4.z, q4.w});
        uint32_t a;
        if (N->is_leaf) {
            const float v = prim_value_gen(N->kind, N->param, r.x, r.y, r.z);
            float2 dd = S.d[lv * 64u + lane];
            uint2 ii = S.id[lv * 64u + lane];
            if (N->slot) { dd.y = v; ii.y = N->prim; } else { dd.x = v; ii.x = N->prim; }
            S.d[lv * 64u + lane] = dd;
            S.id[lv * 64u + lane] = ii;
            a = N->parent;
        } else {
            const P4 c = down_point(N, r);
            S.pt[lv * 64u + lane] = make_float4(c.x, c.y, c.z, c.w);
            S.d[(lv + 1u) * 64u + lane] = make_float2(__builtin_inff(), __builtin_inff());   // :177
            S.id[(lv + 1u) * 64u + lane] = make_uint2(SDF_NO_PRIM, SDF_NO_PRIM);
            a = i;   // an internal node with nothing live below it is combined at once
        }
        for (uint32_t u = 0; u < N->n_up; ++u) {
            const GpuGenNode* __restrict__ A = nodes + a;
            const uint32_t la = A->level;
            const float2 c = S.d[(la + 1u) * 64u + lane];
            const uint2 ci = S.id[(la + 1u) * 64u + lane];
            float v;
            uint32_t vi;
            if (A->op == GEN_OP_CLONE) {   // reached only when the clone is repaired: the branch its one slot filled
                v = c.x;
                vi = ci.x;
            } else {
                const bool left = c.x < c.y;
                v = left ? c.x : c.y;
                vi = left ? ci.x : ci.y;
                const float k = A->smooth_k;
                if (k > 0.f) {
                    float t = k - __builtin_fabsf(c.x - c.y);
                    t = t > 0.f ? t : 0.f;
                    const float h = t / k;
                    float q = h * h;
                    q = q * k;
                    q = q * 0.25f;
                    v = v - q;
                }
            }
            float2 dd = S.d[la * 64u + lane];
            uint2 ii = S.id[la * 64u + lane];
            if (A->slot) { dd.y = v; ii.y = vi; } else { dd.x = v; ii.x = vi; }
            S.d[la * 64u + lane] = dd;
            S.id[la * 64u + lane] = ii;
            a = A->parent;
        }
    }
    out_d = S.d[64u + lane].x;
    out_id = S.id[64u + lane].x;
}

__device__ __forceinline__ EvalStacks eval_stacks(float4* base, uint32_t pt_slots, uint32_t levels) {
    EvalStacks S;
    S.pt = base;
    S.d = reinterpret_cast<float2*>(base + (size_t)pt_slots * 64u);
    S.id = reinterpret_cast<uint2*>(S.d + (size_t)(levels + 2u) * 64u);
    S.cut = reinterpret_cast<float*>(S.id + (size_t)(levels + 2u) * 64u);
    S.ctl = nullptr;   // set by the kernel (the control stack sits at the end of the block's LDS)
    return S;
}

// ---- pruned walks of the general path (round 3) ----------------------------------------------------------------
// compact.cpp (gen_prune_data) gives every resident node a box IN THE FRAME OF THE POINT ITS PARENT HANDS DOWN -- world
// below plain unions, the folded frame below a clone, the re-based frame below a node with a trans_inverse of its own --
// and `slack`, what the smooth unions inside the subtree can take off the minimum of its leaves (k/4 per level).  For a
// handed-down point q OUTSIDE the box the subtree's value is therefore at least dist(q, box) - slack, and the walks
// below skip a subtree (jump to skip_to) when that bound shows, for every lane of the wavefront, that the subtree's
// exact value cannot matter.  A skipped subtree leaves +inf in its slot, which is what the frames are initialised
// with (faux_shader.rs:177); a lane that did not need a subtree its wavefront visits anyway simply gets the exact value.
// The bound, rounded down: the hardware square root and the subtraction lose 1e-6 relative at most, and a leaf value's
// own rounding is of the order 1e-6 (|q|_1 + T) (prune_margin); 2e-5 relative and 2e-5 (|q|_1 + T) absolute cover both.
__device__ __forceinline__ float gen_lower_bound(const GpuGenNode* __restrict__ N, float4 q, float scene_T) {
    const float dx = __builtin_fmaxf(__builtin_fabsf(q.x - N->bc[0]) - N->bh[0], 0.f);
    const float dy = __builtin_fmaxf(__builtin_fabsf(q.y - N->bc[1]) - N->bh[1], 0.f);
    const float dz = __builtin_fmaxf(__builtin_fabsf(q.z - N->bc[2]) - N->bh[2], 0.f);
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (!(d2 > 0.f)) return -__builtin_inff();   // inside the box (or not a number): no bound
    const float d = __builtin_amdgcn_sqrtf(d2);
    return (d * 0.99998f - N->slack) - 2.0e-5f * ((__builtin_fabsf(q.x) + __builtin_fabsf(q.y)) + (__builtin_fabsf(q.z) + scene_T));
}

// Both pruned walks are depth-first with the NEARER child first (by the bound of the wavefront's first live lane): the
// near subtree usually holds the minimum, and once it is known exactly the far one faces a tight radius / cut.  The
// child that waits is kept per level in `ctl` (wave-uniform, LDS; every lane writes the same value).
//
// walk_general with skipping.  only == SDF_NO_PRIM for every lane (`any_only` false): a subtree is skipped when its
// bound exceeds every live lane's prune radius (strictly: a tie could still win on the index).  Otherwise (normal
// taps): a subtree is visited only if it holds the primitive some lane asks for (primitive ids are pre-order
// ordinals, so a subtree holds a contiguous range of them).
template <class Acc>
__device__ __forceinline__ void walk_general_pruned(const GpuGenNode* __restrict__ nodes, uint32_t n, P4 p, uint32_t only, bool any_only,
                                                    bool live, float scene_T, Acc& acc, float4* __restrict__ stk,
                                                    uint32_t* __restrict__ ctl, uint32_t lane) {
    stk[lane] = make_float4(p.x, p.y, p.z, p.w);
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
    if (lm == 0ull || n == 0u) return;
    const int lead = (int)__builtin_ctzll(lm);
    uint32_t i = 0;
    // the bound of the child descended into was formed when its parent ordered its children: not formed again
    float carried_lb = 0.f;
    bool have_lb = false;
    for (;;) {
        const GpuGenNode Nv = nodes[i];   // the whole record at once: one memory round trip per visit, not one per field
        const GpuGenNode* const N = &Nv;
        const uint32_t lv = N->level;
        const uint32_t ps = N->pt_slots;
        const float4 q4 = stk[(ps & 0xFFu) * 64u + lane];
        bool want;
        if (any_only) {
            const uint32_t first = N->prim, cnt = N->is_leaf ? 1u : N->kind;
            want = live && only != SDF_NO_PRIM && (only - first) < cnt;
        } else {
            want = live && !((have_lb ? carried_lb : gen_lower_bound(N, q4, scene_T)) > acc.radius());
        }
        have_lb = false;
        bool descended = false;
        if (__builtin_amdgcn_ballot_w64(want) != 0ull) {
            const P4 r = (ps >> 16) ? P4{q4.x, q4.y, q4.z, q4.w} : mat4_mul(N->m, P4{q4.x, q4.y, q4.z, q4.w});   // (bit 16: identity map, plain union)
            if (N->is_leaf) {
                const float d = prim_value_gen(N->kind, N->param, r.x, r.y, r.z);
                if (only == SDF_NO_PRIM || only == N->prim) acc.offer(d, N->prim);
            } else {
                const P4 c = down_point(N, r);
                const float4 c4 = make_float4(c.x, c.y, c.z, c.w);
                if (!(ps >> 16)) stk[((ps >> 8) & 0xFFu) * 64u + lane] = c4;
                const uint32_t k0 = i + 1u;
                if (k0 < N->skip_to) {
                    const uint32_t k1 = nodes[k0].skip_to;
                    uint32_t first = k0, second = REF_NONE;
                    if (k1 < N->skip_to) {
                        const float b0 = gen_lower_bound(nodes + k0, c4, scene_T), b1 = gen_lower_bound(nodes + k1, c4, scene_T);
                        const uint32_t pick1 = (uint32_t)__builtin_amdgcn_readlane((int)((b1 < b0) ? 1u : 0u), lead);
                        first = pick1 ? k1 : k0;
                        second = pick1 ? k0 : k1;
                        carried_lb = pick1 ? b1 : b0;
                        have_lb = true;
                    }
                    ctl[lv] = second;
                    i = first;
                    descended = true;
                }
            }
        }
        if (descended) continue;
        // node i is done: its parent's waiting child next, or the parent is done as well
        for (;;) {
            const uint32_t P = nodes[i].parent;
            if (P == REF_NONE) return;
            const uint32_t lp = nodes[P].level;
            const uint32_t j = ctl[lp];
            if (j != REF_NONE) { ctl[lp] = REF_NONE; i = j; break; }
            i = P;
        }
    }
}

// An upper bound on the root's value for every lane: the value of ONE primitive, reached by descending from the root
// into the child whose box the first live lane's point is nearer to (the lanes of a wavefront are neighbours).  A
// union, plain or smooth, never exceeds any of its operands' values, a repaired clone hands its one operand's up.
__device__ __forceinline__ float gen_seed_value(const GpuGenNode* __restrict__ nodes, uint32_t n, P4 p, bool live, float scene_T,
                                                float4* __restrict__ pt, uint32_t lane) {
    pt[lane] = make_float4(p.x, p.y, p.z, p.w);
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
    if (lm == 0ull || n == 0u) return __builtin_inff();
    const int lead = (int)__builtin_ctzll(lm);
    uint32_t i = 0;
    for (;;) {
        const GpuGenNode Nv = nodes[i];   // the whole record at once: one memory round trip per visit, not one per field
        const GpuGenNode* const N = &Nv;
        const uint32_t ps = N->pt_slots;
        const float4 q4 = pt[(ps & 0xFFu) * 64u + lane];
        const P4 r = (ps >> 16) ? P4{q4.x, q4.y, q4.z, q4.w} : mat4_mul(N->m, P4{q4.x, q4.y, q4.z, q4.w});   // (bit 16: identity map, plain union)
        if (N->is_leaf) return prim_value_gen(N->kind, N->param, r.x, r.y, r.z);
        const P4 c = down_point(N, r);
        const float4 c4 = make_float4(c.x, c.y, c.z, c.w);
        if (!(ps >> 16)) pt[((ps >> 8) & 0xFFu) * 64u + lane] = c4;
        const uint32_t k0 = i + 1u;
        if (k0 >= N->skip_to) return __builtin_inff();   // nothing live below
        const uint32_t k1 = nodes[k0].skip_to;
        uint32_t next = k0;
        if (k1 < N->skip_to) {
            const float b0 = gen_lower_bound(nodes + k0, c4, scene_T), b1 = gen_lower_bound(nodes + k1, c4, scene_T);
            const uint32_t pick1 = (uint32_t)__builtin_amdgcn_readlane((int)((b1 < b0) ? 1u : 0u), lead);
            next = pick1 ? k1 : k0;
        }
        i = next;
    }
}

// walk_eval with skipping.  Every node is visited with a CUT: its exact value is needed only if it is below the cut;
// above it any value >= the cut will do (+inf, when the subtree is skipped).  Contract of a visit, by induction over
// the tree: the slot holds v exactly if v < cut, and something >= cut otherwise.
//   root: cut = an upper bound on the root's value (gen_seed_value), a little inflated: the root is exact.
//   children of a node with cut c and smooth radius k (0: plain minimum): v = min(a, b) - f, 0 <= f <= k/4, f = 0
//   once |a - b| >= k.  v < c needs min(a, b) < c + k/4, and an operand matters beyond being the minimum only within k
//   of the other: the child visited FIRST gets c + 1.25 k.  The child visited second is cut by the first one's result
//   a: exact (a below its cut) -> it matters only below a + k (at k = 0: only if it is <= a -- a tie goes to the
//   right-hand operand -- so the cut is a, inflated); not exact -> it matters only as the minimum, below c + k/4.
//   A clone hands its cut to its one child.
// The nearer child is visited first, so the loose first-child cuts accumulate only along the path towards the nearest
// primitives, which has to be walked anyway; everything off that path meets a cut formed from an exact value.
// Cuts are inflated by 1e-5 relative wherever they are formed, which only makes them more permissive.
__device__ __forceinline__ void walk_eval_pruned(const GpuGenNode* __restrict__ nodes, uint32_t n, P4 p, const EvalStacks& S, bool live,
                                                 float scene_T, float cut_root, uint32_t lane, float& out_d, uint32_t& out_id) {
    S.pt[lane] = make_float4(p.x, p.y, p.z, p.w);
    S.d[64u + lane] = make_float2(__builtin_inff(), __builtin_inff());
    S.id[64u + lane] = make_uint2(SDF_NO_PRIM, SDF_NO_PRIM);
    S.cut[64u + lane] = cut_root;
    out_d = __builtin_inff();
    out_id = SDF_NO_PRIM;
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
    if (lm == 0ull || n == 0u) return;
    const int lead = (int)__builtin_ctzll(lm);
    uint32_t i = 0;
    float carried_lb = 0.f;      // the bound of the child descended into, formed when its parent ordered its children
    bool have_lb = false;
    bool second_visit = false;   // node i is the child that waited: its sibling's result is in the frame
    for (;;) {
        const GpuGenNode Nv = nodes[i];   // the whole record at once: one memory round trip per visit, not one per field
        const GpuGenNode* const N = &Nv;
        const uint32_t lv = N->level;
        const uint32_t ps = N->pt_slots;
        const float4 q4 = S.pt[(ps & 0xFFu) * 64u + lane];
        const float base = S.cut[lv * 64u + lane];
        float cut = base;
        if (second_visit) {
            const float2 fr = S.d[lv * 64u + lane];
            const float a = N->slot ? fr.x : fr.y, pk = N->parent_k;
            cut = a < base ? __builtin_fminf(base, a + pk) : base - pk;
            cut = cut + 1.0e-5f * (__builtin_fabsf(cut) + pk) + 1.0e-30f;
        }
        second_visit = false;
        const bool want = live && !((have_lb ? carried_lb : gen_lower_bound(N, q4, scene_T)) >= cut);
        have_lb = false;
        bool descended = false;
        if (__builtin_amdgcn_ballot_w64(want) != 0ull) {
            const P4 r = (ps >> 16) ? P4{q4.x, q4.y, q4.z, q4.w} : mat4_mul(N->m, P4{q4.x, q4.y, q4.z, q4.w});   // (bit 16: identity map, plain union)
            if (N->is_leaf) {
                const float v = prim_value_gen(N->kind, N->param, r.x, r.y, r.z);
                float2 dd = S.d[lv * 64u + lane];
                uint2 ii = S.id[lv * 64u + lane];
                if (N->slot) { dd.y = v; ii.y = N->prim; } else { dd.x = v; ii.x = N->prim; }
                S.d[lv * 64u + lane] = dd;
                S.id[lv * 64u + lane] = ii;
            } else {
                const P4 c = down_point(N, r);
                const float4 c4 = make_float4(c.x, c.y, c.z, c.w);
                if (!(ps >> 16)) S.pt[((ps >> 8) & 0xFFu) * 64u + lane] = c4;
                S.d[(lv + 1u) * 64u + lane] = make_float2(__builtin_inff(), __builtin_inff());   // faux_shader.rs:177
                S.id[(lv + 1u) * 64u + lane] = make_uint2(SDF_NO_PRIM, SDF_NO_PRIM);
                const float k = N->smooth_k;
                const float cc = cut + 1.25f * k;
                S.cut[(lv + 1u) * 64u + lane] = cc + 1.0e-5f * (__builtin_fabsf(cc) + k) + 1.0e-30f;
                const uint32_t k0 = i + 1u;
                if (k0 < N->skip_to) {
                    const uint32_t k1 = nodes[k0].skip_to;
                    uint32_t first = k0, second = REF_NONE;
                    if (k1 < N->skip_to) {
                        const float b0 = gen_lower_bound(nodes + k0, c4, scene_T), b1 = gen_lower_bound(nodes + k1, c4, scene_T);
                        const uint32_t pick1 = (uint32_t)__builtin_amdgcn_readlane((int)((b1 < b0) ? 1u : 0u), lead);
                        first = pick1 ? k1 : k0;
                        second = pick1 ? k0 : k1;
                        carried_lb = pick1 ? b1 : b0;
                        have_lb = true;
                    }
                    S.ctl[lv] = second;
                    i = first;
                    descended = true;
                }
                // (an internal node with nothing live below it: its frame stays +inf and it is combined at once, below)
            }
        }
        if (descended) continue;
        // Node i is done (evaluated, or skipped: its slot keeps the +inf its frame was initialised with) -- unless it
        // is an internal node that was visited and has just been found childless, which is combined like any parent.
        bool combine_self = !N->is_leaf && __builtin_amdgcn_ballot_w64(want) != 0ull;
        for (;;) {
            uint32_t P;
            if (combine_self) { P = i; combine_self = false; }
            else {
                P = nodes[i].parent;
                if (P == REF_NONE) { out_d = S.d[64u + lane].x; out_id = S.id[64u + lane].x; return; }
                const uint32_t lp = nodes[P].level;
                const uint32_t j = S.ctl[lp];
                if (j != REF_NONE) { S.ctl[lp] = REF_NONE; i = j; second_visit = true; break; }
            }
            // both children of P are done: uptree_dispatch (faux_shader.rs:84-94) into P's slot of its parent's frame
            const GpuGenNode* __restrict__ A = nodes + P;
            const uint32_t la = A->level;
            const float2 c = S.d[(la + 1u) * 64u + lane];
            const uint2 ci = S.id[(la + 1u) * 64u + lane];
            float v;
            uint32_t vi;
            if (A->op == GEN_OP_CLONE) {
                v = c.x;
                vi = ci.x;
            } else {
                const bool left = c.x < c.y;
                v = left ? c.x : c.y;
                vi = left ? ci.x : ci.y;
                const float k = A->smooth_k;
                if (k > 0.f) {
                    float t = k - __builtin_fabsf(c.x - c.y);
                    t = t > 0.f ? t : 0.f;
                    const float h = t / k;
                    float q = h * h;
                    q = q * k;
                    q = q * 0.25f;
                    v = v - q;
                }
            }
            float2 dd = S.d[la * 64u + lane];
            uint2 ii = S.id[la * 64u + lane];
            if (A->slot) { dd.y = v; ii.y = vi; } else { dd.x = v; ii.x = vi; }
            S.d[la * 64u + lane] = dd;
            S.id[la * 64u + lane] = ii;
            i = P;
        }
    }
}

template <int K>
__global__ __launch_bounds__(64) void query_general_kernel(const GpuGenNode* __restrict__ nodes, uint32_t n_nodes,
                                                           uint32_t eval_levels, uint32_t prune, float scene_T, uint32_t ctl_off,
                                                           const float4* __restrict__ pts, const uint32_t* __restrict__ order, uint32_t n,
                                                           uint32_t k, float* __restrict__ out_dist,
                                                           uint32_t* __restrict__ out_prim) {
    extern __shared__ float4 gstk[];
    uint32_t* const ctl = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(gstk) + ctl_off);
    const uint32_t lane = threadIdx.x;
    const uint32_t slot = blockIdx.x * 64u + lane;
    const bool alive = slot < n;
    const uint32_t i = alive ? (order ? order[slot] : slot) : 0u;   // results go back to the caller's index
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
    if (alive) p = pts[i];
    const bool in_dom = in_domain3(p.x, p.y, p.z);   // see in_domain3: (NaN, SDF_NO_PRIM) otherwise, as on the pruned path
    if constexpr (K == 1) {
        if (eval_levels) {   // a combine that is not a plain minimum is in effect: the real post-order evaluation
            float d;
            uint32_t id;
            EvalStacks S = eval_stacks(gstk, prune ? prune : eval_levels + 1u, eval_levels);
            S.ctl = ctl;
            if (prune) {
                const bool live = alive && in_dom;
                const float U = gen_seed_value(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, live, scene_T, S.pt, lane);
                walk_eval_pruned(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, S, live, scene_T, U + 1.0e-5f * __builtin_fabsf(U) + 1.0e-30f, lane, d, id);
            } else {
                walk_eval(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, S, lane, d, id);
            }
            if (alive) {
                out_dist[i] = in_dom ? d : __builtin_nanf("");
                if (out_prim) out_prim[i] = in_dom ? id : SDF_NO_PRIM;
            }
            return;
        }
        AccNearest acc{__builtin_inff(), SDF_NO_PRIM};
        if (prune) {
            // the value of one nearby primitive as the first radius (that primitive is met again in the walk and takes
            // the slot then: an index of SDF_NO_PRIM loses every tie)
            const bool live = alive && in_dom;
            acc.best = gen_seed_value(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, live, scene_T, gstk, lane);
            walk_general_pruned(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, SDF_NO_PRIM, false, live, scene_T, acc, gstk, ctl, lane);
        } else {
            walk_general(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, SDF_NO_PRIM, acc, gstk, lane);
        }
        if (alive) {
            out_dist[i] = in_dom ? acc.best : __builtin_nanf("");
            if (out_prim) out_prim[i] = in_dom ? acc.bid : SDF_NO_PRIM;
        }
    } else {
        AccK<K> acc;
        acc.init();
        if (prune) walk_general_pruned(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, SDF_NO_PRIM, false, alive && in_dom, scene_T, acc, gstk, ctl, lane);
        else walk_general(nodes, n_nodes, P4{p.x, p.y, p.z, 1.0f}, SDF_NO_PRIM, acc, gstk, lane);
        if (!in_dom) acc.init_nan();
        if (alive) {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if ((uint32_t)j < k) {
                    out_dist[(size_t)i * k + j] = acc.dist(j);
                    out_prim[(size_t)i * k + j] = acc.prim(j);
                }
        }
    }
}

__global__ __launch_bounds__(64) void render_general_kernel(const GpuGenNode* __restrict__ nodes, uint32_t n_nodes,
                                                            uint32_t eval_levels, uint32_t prune, float scene_T, uint32_t ctl_off,
                                                            sdfgpu_camera cam, uint32_t width, uint32_t height,
                                                            sdfgpu_march_params mp, uint32_t tile_first, uint32_t tile_stride,
                                                            uint32_t tile_count, uint32_t row_major,
                                                            uint8_t* __restrict__ o_hit, float* __restrict__ o_depth,
                                                            float* __restrict__ o_normal, uint32_t* __restrict__ o_prim) {
    extern __shared__ float4 gstk[];
    uint32_t* const ctl = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(gstk) + ctl_off);
    const uint32_t lane = threadIdx.x;
    const uint32_t tile_local = blockIdx.x;
    if (tile_local >= tile_count) return;
    const uint32_t tile = tile_first + tile_local * tile_stride;
    const uint32_t tiles_x = (width + 7u) >> 3;
    const uint32_t x = (tile % tiles_x) * 8u + (lane & 7u), y = (tile / tiles_x) * 8u + (lane >> 3);
    const bool inside = x < width && y < height;
    const float fx = (float)x, fy = (float)y;
    const float ox = cam.eye[0], oy = cam.eye[1], oz = cam.eye[2];
    float dx = ((cam.origin[0] + cam.du[0] * fx) + cam.dv[0] * fy) - ox;
    float dy = ((cam.origin[1] + cam.du[1] * fx) + cam.dv[1] * fy) - oy;
    float dz = ((cam.origin[2] + cam.du[2] * fx) + cam.dv[2] * fy) - oz;
    const float dl = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
    dx = dx / dl; dy = dy / dl; dz = dz / dl;
    float t = mp.t_min;
    bool alive = inside && in_domain3(dx, dy, dz) && march_domain(ox, oy, oz, mp.t_min), hit = false;
    const float t_hi = march_t_hi(mp.t_max);
    uint32_t hit_prim = SDF_NO_PRIM;
    float px = 0.f, py = 0.f, pz = 0.f;
    for (uint32_t s = 0; s < mp.max_steps; ++s) {
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
        if (alive) { px = ox + dx * t; py = oy + dy * t; pz = oz + dz * t; }
        AccNearest acc{__builtin_inff(), SDF_NO_PRIM};
        if (prune) {
            const float U = gen_seed_value(nodes, n_nodes, P4{px, py, pz, 1.0f}, alive, scene_T, gstk, lane);
            if (eval_levels) {
                EvalStacks S = eval_stacks(gstk, prune ? prune : eval_levels + 1u, eval_levels);
                S.ctl = ctl;
                walk_eval_pruned(nodes, n_nodes, P4{px, py, pz, 1.0f}, S, alive, scene_T, U + 1.0e-5f * __builtin_fabsf(U) + 1.0e-30f, lane,
                                 acc.best, acc.bid);
            } else {
                acc.best = U;
                walk_general_pruned(nodes, n_nodes, P4{px, py, pz, 1.0f}, SDF_NO_PRIM, false, alive, scene_T, acc, gstk, ctl, lane);
            }
        } else if (eval_levels) walk_eval(nodes, n_nodes, P4{px, py, pz, 1.0f}, eval_stacks(gstk, eval_levels + 1u, eval_levels), lane, acc.best, acc.bid);
        else walk_general(nodes, n_nodes, P4{px, py, pz, 1.0f}, SDF_NO_PRIM, acc, gstk, lane);
        if (alive) {
            if (acc.best < mp.eps_hit) { hit = true; hit_prim = acc.bid; alive = false; }
            else { t = t + acc.best; if (!(t <= t_hi)) alive = false; }
        }
    }
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
        const float h = mp.normal_h;
        const float sgn[4][3] = {{1.f, -1.f, -1.f}, {-1.f, -1.f, 1.f}, {-1.f, 1.f, -1.f}, {1.f, 1.f, 1.f}};
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            AccNearest acc{__builtin_inff(), SDF_NO_PRIM};
            if (prune)
                walk_general_pruned(nodes, n_nodes, P4{px + sgn[j][0] * h, py + sgn[j][1] * h, pz + sgn[j][2] * h, 1.0f},
                                    hit ? hit_prim : SDF_NO_PRIM, true, hit, scene_T, acc, gstk, ctl, lane);
            else
                walk_general(nodes, n_nodes, P4{px + sgn[j][0] * h, py + sgn[j][1] * h, pz + sgn[j][2] * h, 1.0f},
                             hit ? hit_prim : 0xFFFFFFFEu, acc, gstk, lane);
            f[j] = acc.best;
        }
        if (hit) {
            const float sx = ((f[0] - f[1]) - f[2]) + f[3];
            const float sy = ((-f[0] - f[1]) + f[2]) + f[3];
            const float sz = ((-f[0] + f[1]) - f[2]) + f[3];
            const float nl = __builtin_sqrtf((sx * sx + sy * sy) + sz * sz);
            if (nl > 0.f) { nx = sx / nl; ny = sy / nl; nz = sz / nl; }
        }
    }
    size_t o;
    bool wr;
    if (row_major) { o = (size_t)y * width + x; wr = inside; }
    else { o = (size_t)tile_local * 64u + lane; wr = true; }
    if (wr) {
        o_hit[o] = hit ? 1 : 0;
        if (o_depth) o_depth[o] = hit ? t : __builtin_inff();
        if (o_normal) { o_normal[3 * o] = nx; o_normal[3 * o + 1] = ny; o_normal[3 * o + 2] = nz; }
        if (o_prim) o_prim[o] = hit_prim;
    }
}

// ------------------------------------------- bounds of faux_shader.rs:39-58 --
// MINDIST (A-2) and MAXDIST (A-3) exactly as written, all four lanes:
//   mindist: t = full_inverse * p; q = (|t| - 1) * scale;
//            |max(q,0)|_4 (glam dot4 order (x^2+z^2)+(y^2+w^2)) + min(max(qx,max(qy,qz)),0)
//   maxdist: max over the 8 corners v of |matrix * v - p|_4 (CmpFloat max).
__device__ __forceinline__ float len4_glam(float x, float y, float z, float w) {
    return __builtin_sqrtf((x * x + z * z) + (y * y + w * w));
}
__global__ __launch_bounds__(256) void box_bounds_kernel(const SdfBoundingBoxBlock* __restrict__ box,
                                                         const float4* __restrict__ pts, uint32_t n,
                                                         float* __restrict__ out_min, float* __restrict__ out_max) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p4 = pts[i];
    const P4 p{p4.x, p4.y, p4.z, p4.w};
    const P4 t = mat4_mul(box->full_inverse.m, p);
    const float qx = (__builtin_fabsf(t.x) - 1.0f) * box->scale.v[0];
    const float qy = (__builtin_fabsf(t.y) - 1.0f) * box->scale.v[1];
    const float qz = (__builtin_fabsf(t.z) - 1.0f) * box->scale.v[2];
    const float qw = (__builtin_fabsf(t.w) - 1.0f) * box->scale.v[3];
    const float l = len4_glam(qx > 0.f ? qx : 0.f, qy > 0.f ? qy : 0.f, qz > 0.f ? qz : 0.f, qw > 0.f ? qw : 0.f);
    out_min[i] = l + __builtin_fminf(__builtin_fmaxf(qx, __builtin_fmaxf(qy, qz)), 0.0f);
    float best = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const P4 v{(c & 1) ? -1.f : 1.f, (c & 2) ? -1.f : 1.f, (c & 4) ? -1.f : 1.f, 1.f};
        const P4 m = mat4_mul(box->matrix.m, v);
        const float d = len4_glam(m.x - p.x, m.y - p.y, m.z - p.z, m.w - p.w);
        best = (c == 0) ? d : (best > d ? best : d);
    }
    out_max[i] = best;
}

// --------------------------------------------------------------- launch ---

// Wavefronts per workgroup of the walking kernels.  Nothing in them needs a workgroup (every
// wavefront owns its rays, its LDS stack slice and its exit), so the default is ONE: the
// dispatcher then places and retires wavefronts individually instead of in groups of four
// whose LDS is held until the slowest of the four is done (measured: 1080p frame 4.03 ms at
// 4 per group, 3.68 at 2, 3.59 at 1).  SDFGPU_WG_WAVES=1..4 overrides.
static uint32_t wg_waves() {
    static const uint32_t v = [] {
        const char* e = std::getenv("SDFGPU_WG_WAVES");
        return (uint32_t)std::max(1, std::min(4, e ? std::atoi(e) : 1));
    }();
    return v;
}

// prefetch_beyond: size of the record arrays from which the walk warms the scalar cache with both children
// (SDFGPU_PREFETCH=0/1 overrides, for measurements).  Measured with 10 M points per batch on trees of
// 12 / 49 / 195 MiB (tools/query_sweep.py): k = 8 gains 3 / 8 / 10 % from it, the march on the 4K frame
// likewise; k = 1, with a third as many primitives evaluated per walk, loses 2-3 % on the 12 MiB tree and
// gains 0-8 % on the larger ones -- so one XCD's L2 (4 MiB) is the limit for every kernel but the k = 1
// query, whose limit is 32 MiB.
static TreeView make_view(const DeviceTree& d, size_t prefetch_beyond = 4u << 20) {
    TreeView v;
    v.leaf = d.leaf;
    v.inner = d.inner;
    v.inner_obb = d.inner_obb;
    v.root = d.root_ref;
    v.depth = d.stack_depth;
    v.margin_T = d.margin_T;
    v.n_leaf = d.n_leaf;
    v.skip_ext = d.skip_ext;
    v.lip = d.lipschitz;
    v.bscale = d.box_scale;
    v.inv_bscale = 1.0f / d.box_scale;
    v.gblock = d.grid.cells ? d.grid.block : nullptr;
    v.chain = d.chain;
    v.mg_scale = 1.0f + (float)d.chain_max;
    static const int forced = [] { const char* e = std::getenv("SDFGPU_PREFETCH"); return e ? std::atoi(e) : -1; }();
    v.xcd = 0;   // set by query_grid for the query kernels
    v.prefetch = forced >= 0 ? (uint32_t)(forced != 0) : (uint32_t)((size_t)d.n_leaf * (sizeof(GpuLeaf) + sizeof(GpuInner)) > prefetch_beyond);
    return v;
}

// Grid of a query kernel over nb blocks, and the chunk size of its XCD-aware block order (TreeView::xcd, 0 =
// plain order): 1/32 of the batch, at most 1024 blocks (SDFGPU_XCD_ORDER: 0 = off, n = that chunk size).
static uint32_t query_grid(uint32_t nb, TreeView* v) {
    static const int env = [] { const char* e = std::getenv("SDFGPU_XCD_ORDER"); return e ? std::atoi(e) : -1; }();
    uint32_t chunk = env >= 0 ? (uint32_t)env : std::min(1024u, nb / 32u);
    if (chunk == 0 || nb < 64u) { v->xcd = 0; return nb; }
    v->xcd = chunk;
    const uint32_t group = 8u * chunk;
    return (nb + group - 1u) / group * group;
}

template <int COUNT>
static hipError_t launch_nearest_t(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n,
                                   float* out_dist, uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s) {
    if (n == 0) return hipSuccess;
    TreeView v = make_view(d, 32u << 20);
    const uint32_t wpg = wg_waves();
    dim3 grid(query_grid((n + 64u * wpg - 1u) / (64u * wpg), &v)), block(64u * wpg);
    size_t lds = (size_t)wpg * d.stack_depth * 64 * sizeof(float);
    if (d.chain) {
        if constexpr (COUNT == 0) nearest_kernel<SDFGPU_BOUNDS_AABB, 0, true><<<grid, block, lds, s>>>(v, (const float4*)pts, order, n, out_dist, out_prim, st);
        else return hipErrorNotSupported;   // no counting instantiation of the chain path
    } else if (d.bounds == SDFGPU_BOUNDS_OBB)
        nearest_kernel<SDFGPU_BOUNDS_OBB, COUNT><<<grid, block, lds, s>>>(v, (const float4*)pts, order, n, out_dist, out_prim, st);
    else
        nearest_kernel<SDFGPU_BOUNDS_AABB, COUNT><<<grid, block, lds, s>>>(v, (const float4*)pts, order, n, out_dist, out_prim, st);
    return hipGetLastError();
}

// LDS of the general path: one float4 per level per lane; with the post-order combine, also the two
// results (distance, index) of every level.
static uint32_t general_eval_levels(const DeviceTree& d) { return d.gen_eval ? d.stack_depth : 0u; }
static size_t general_ctl_off(const DeviceTree& d) {   // where the pruned walks' control stack starts: behind everything else
    size_t b = (size_t)(d.gen_prune ? d.gen_pt_levels : d.stack_depth + 1u) * 64u * sizeof(float4);
    if (d.gen_eval) b += (size_t)(d.stack_depth + 2u) * 64u * (sizeof(float2) + sizeof(uint2) + sizeof(float));
    return b;
}
static size_t general_lds(const DeviceTree& d) { return general_ctl_off(d) + 64u * sizeof(uint32_t); }

hipError_t launch_box_bounds(const SdfBoundingBoxBlock* d_box, const float* d_pts, uint32_t n, float* d_min, float* d_max,
                             hipStream_t s) {
    if (n == 0) return hipSuccess;
    box_bounds_kernel<<<dim3((n + 255u) / 256u), dim3(256), 0, s>>>(d_box, (const float4*)d_pts, n, d_min, d_max);
    return hipGetLastError();
}

hipError_t launch_nearest(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n, float* out_dist,
                          uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s, bool stats_asm_walk) {
    if (d.gen) {   // general path: exhaustive literal chain (no counters, no reordering needed)
        if (n == 0) return hipSuccess;
        query_general_kernel<1><<<dim3((n + 63u) / 64u), dim3(64), general_lds(d), s>>>(d.gen, d.n_gen, general_eval_levels(d), d.gen_prune ? d.gen_pt_levels : 0u,
                                                                                       d.margin_T, (uint32_t)general_ctl_off(d), (const float4*)pts, order, n, 1, out_dist, out_prim);
        return hipGetLastError();
    }
    if (st) return stats_asm_walk ? launch_nearest_t<2>(d, pts, order, n, out_dist, out_prim, st, s)
                                  : launch_nearest_t<1>(d, pts, order, n, out_dist, out_prim, st, s);
    return launch_nearest_t<0>(d, pts, order, n, out_dist, out_prim, nullptr, s);
}


template <int MODE, bool COUNT, bool CHAIN = false>
static hipError_t launch_knn_m(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n, uint32_t k,
                               float* out_dist, uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s) {
    TreeView v = make_view(d);
    const uint32_t wpg = wg_waves();
    dim3 grid(query_grid((n + 64u * wpg - 1u) / (64u * wpg), &v)), block(64u * wpg);
    size_t lds = (size_t)wpg * d.stack_depth * 64 * sizeof(float);
    const float4* p4 = (const float4*)pts;
    if (k <= 2) knn_kernel<MODE, 2, COUNT, CHAIN><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim, st);
    else if (k <= 4) knn_kernel<MODE, 4, COUNT, CHAIN><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim, st);
    else if (k <= 8) knn_kernel<MODE, 8, COUNT, CHAIN><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim, st);
    else knn_kernel<MODE, 16, COUNT, CHAIN><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim, st);
    return hipGetLastError();
}

hipError_t launch_knn(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n, uint32_t k,
                      float* out_dist, uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s) {
    if (n == 0 || k == 0) return hipSuccess;
    if (d.gen) {   // general path: exhaustive literal chain, no counters
        dim3 grid((n + 63u) / 64u), block(64);
        const size_t lds = general_lds(d);
        const float4* p4 = (const float4*)pts;
        if (k <= 2) query_general_kernel<2><<<grid, block, lds, s>>>(d.gen, d.n_gen, 0u, d.gen_prune ? d.gen_pt_levels : 0u, d.margin_T, (uint32_t)general_ctl_off(d), p4, order, n, k, out_dist, out_prim);
        else if (k <= 4) query_general_kernel<4><<<grid, block, lds, s>>>(d.gen, d.n_gen, 0u, d.gen_prune ? d.gen_pt_levels : 0u, d.margin_T, (uint32_t)general_ctl_off(d), p4, order, n, k, out_dist, out_prim);
        else if (k <= 8) query_general_kernel<8><<<grid, block, lds, s>>>(d.gen, d.n_gen, 0u, d.gen_prune ? d.gen_pt_levels : 0u, d.margin_T, (uint32_t)general_ctl_off(d), p4, order, n, k, out_dist, out_prim);
        else query_general_kernel<16><<<grid, block, lds, s>>>(d.gen, d.n_gen, 0u, d.gen_prune ? d.gen_pt_levels : 0u, d.margin_T, (uint32_t)general_ctl_off(d), p4, order, n, k, out_dist, out_prim);
        return hipGetLastError();
    }
    if (d.chain) {
        if (st) return hipErrorNotSupported;   // no counting instantiation of the chain path
        return launch_knn_m<SDFGPU_BOUNDS_AABB, false, true>(d, pts, order, n, k, out_dist, out_prim, nullptr, s);
    }
    const bool obb = d.bounds == SDFGPU_BOUNDS_OBB;
    if (st) return obb ? launch_knn_m<SDFGPU_BOUNDS_OBB, true>(d, pts, order, n, k, out_dist, out_prim, st, s)
                       : launch_knn_m<SDFGPU_BOUNDS_AABB, true>(d, pts, order, n, k, out_dist, out_prim, st, s);
    return obb ? launch_knn_m<SDFGPU_BOUNDS_OBB, false>(d, pts, order, n, k, out_dist, out_prim, nullptr, s)
               : launch_knn_m<SDFGPU_BOUNDS_AABB, false>(d, pts, order, n, k, out_dist, out_prim, nullptr, s);
}

template <int MODE, bool COUNT, bool CHAIN = false>
static hipError_t launch_render_m(const DeviceTree& d, const sdfgpu_camera& cam, uint32_t w, uint32_t h,
                                  const sdfgpu_march_params& mp, uint32_t tile_first, uint32_t tile_stride,
                                  uint32_t tile_count, bool row_major, const RenderScratch& rs, const uint32_t* bounds,
                                  uint32_t n_bounds, uint8_t* hit, float* depth, float* normal, uint32_t* prim,
                                  sdfgpu_stats* st, hipStream_t s) {
    TreeView v = make_view(d);
    const uint32_t wpg = wg_waves();
    dim3 grid((tile_count + wpg - 1u) / wpg), block(64u * wpg);
    const size_t lds = (size_t)wpg * d.stack_depth * 64 * sizeof(float);
    // Can a point of this march lie outside the near range of the scaled box test (aabb_dist2)?  Each
    // coordinate of o + d t is within |o_a| + |t| of 0 (|d| = 1), t between t_min and min(t_max, 2^60), with
    // 2 % to spare for rounding.  If not -- any ordinary camera -- the kernel without the far walk runs.
    const float reach = std::fmax(std::fabs(cam.eye[0]), std::fmax(std::fabs(cam.eye[1]), std::fabs(cam.eye[2]))) +
                        std::fmax(std::fabs(mp.t_min), std::fabs(std::fmin(mp.t_max, MARCH_LIMIT)));
    const bool far = MODE != SDFGPU_BOUNDS_OBB && !(reach * 1.02f < NEAR_LIMIT * v.inv_bscale);   // true for a NaN as well
    // pass boundaries: 0 = b[0] < b[1] < ... < b[n-1] = max_steps
    for (uint32_t k = 0; k + 1 < n_bounds; ++k) {
        PassArgs pa;
        pa.step_begin = bounds[k];
        pa.step_end = bounds[k + 1];
        pa.final_pass = (k + 2 == n_bounds) ? 1u : 0u;
        pa.in_state = (const RayState*)rs.state[(k + 1) & 1];
        pa.in_offsets = rs.offsets;
        pa.in_hdr = rs.hdr[k & 1];
        pa.out_state = (RayState*)rs.state[k & 1];
        pa.out_counts = rs.counts;
        pa.tile_order = (n_bounds == 2 && !COUNT) ? rs.tile_order : nullptr;
        pa.tile_cost = (n_bounds == 2 && !COUNT) ? rs.tile_cost : nullptr;
        pa.eye = nullptr;
        if (k == 0 && rs.eye && mp.t_min == 0.0f) {
            eye_walk_kernel<MODE, CHAIN><<<dim3(1), dim3(64), (size_t)d.stack_depth * 64 * sizeof(float), s>>>(v, cam.eye[0], cam.eye[1], cam.eye[2],
                                                                                                         (float4*)rs.eye);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            pa.eye = (const float4*)rs.eye;
        }
#define SDFGPU_PASS(FIRST, FAR)                                                                                               \
    render_pass_kernel<MODE, COUNT, FIRST, FAR, CHAIN><<<grid, block, lds, s>>>(v, cam, w, h, mp, tile_first, tile_stride, tile_count, \
                                                                        row_major ? 1u : 0u, pa, hit, depth, normal, prim, st)
        if (k == 0) { if (far) SDFGPU_PASS(true, true); else SDFGPU_PASS(true, false); }
        else { if (far) SDFGPU_PASS(false, true); else SDFGPU_PASS(false, false); }
#undef SDFGPU_PASS
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (!pa.final_pass) {
            scan_counts_kernel<<<dim3(1), dim3(1024), 0, s>>>(rs.counts, tile_count, k == 0 ? nullptr : rs.hdr[k & 1], rs.offsets,
                                                              rs.hdr[(k + 1) & 1]);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

size_t render_scratch_bytes(uint32_t tile_count, size_t* state_bytes, size_t* count_bytes) {
    *state_bytes = (size_t)tile_count * 64u * sizeof(RayState);
    *count_bytes = ((size_t)tile_count + 1u) * sizeof(uint32_t);
    return 2 * *state_bytes + 2 * *count_bytes + 64;
}

hipError_t launch_render(const DeviceTree& d, const sdfgpu_camera& cam, uint32_t w, uint32_t h,
                         const sdfgpu_march_params& mp, uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count,
                         bool row_major, const RenderScratch& rs, const uint32_t* pass_bounds, uint32_t n_bounds,
                         uint8_t* hit, float* depth, float* normal, uint32_t* prim, sdfgpu_stats* st, hipStream_t s) {
    if (tile_count == 0) return hipSuccess;
    if (d.gen) {
        render_general_kernel<<<dim3(tile_count), dim3(64), general_lds(d), s>>>(d.gen, d.n_gen, general_eval_levels(d), d.gen_prune ? d.gen_pt_levels : 0u, d.margin_T, (uint32_t)general_ctl_off(d), cam, w, h, mp, tile_first,
                                                                                tile_stride, tile_count, row_major ? 1u : 0u,
                                                                                hit, depth, normal, prim);
        return hipGetLastError();
    }
    const bool obb = d.bounds == SDFGPU_BOUNDS_OBB;
#define SDFGPU_RENDER(MODE, COUNT) \
    launch_render_m<MODE, COUNT>(d, cam, w, h, mp, tile_first, tile_stride, tile_count, row_major, rs, pass_bounds, n_bounds, \
                                 hit, depth, normal, prim, st, s)
    if (d.chain) {
        if (st) return hipErrorNotSupported;   // no counting instantiation of the chain path
        return launch_render_m<SDFGPU_BOUNDS_AABB, false, true>(d, cam, w, h, mp, tile_first, tile_stride, tile_count, row_major, rs,
                                                                 pass_bounds, n_bounds, hit, depth, normal, prim, nullptr, s);
    }
    if (st) return obb ? SDFGPU_RENDER(SDFGPU_BOUNDS_OBB, true) : SDFGPU_RENDER(SDFGPU_BOUNDS_AABB, true);
    return obb ? SDFGPU_RENDER(SDFGPU_BOUNDS_OBB, false) : SDFGPU_RENDER(SDFGPU_BOUNDS_AABB, false);
#undef SDFGPU_RENDER
}

hipError_t launch_untile_gathered(uint32_t w, uint32_t h, uint32_t world, uint32_t self, size_t px, size_t nbytes,
                                  const void* gathered, const void* mine, uint8_t* hit, float* depth, float* normal,
                                  uint32_t* prim, hipStream_t s) {
    const uint32_t n_tiles = ((w + 7u) / 8u) * ((h + 7u) / 8u);
    if (n_tiles == 0) return hipSuccess;
    untile_gathered_kernel<<<dim3((n_tiles + 3u) / 4u), dim3(256), 0, s>>>(w, h, world, self, n_tiles, px, nbytes, (const char*)gathered,
                                                                          (const char*)mine, hit, depth, normal, prim);
    return hipGetLastError();
}

hipError_t launch_untile(uint32_t w, uint32_t h, uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count,
                         const uint8_t* t_hit, const float* t_depth, const float* t_normal, const uint32_t* t_prim,
                         uint8_t* hit, float* depth, float* normal, uint32_t* prim, hipStream_t s) {
    if (tile_count == 0) return hipSuccess;
    untile_kernel<<<dim3((tile_count + 3u) / 4u), dim3(256), 0, s>>>(w, h, tile_first, tile_stride, tile_count, t_hit,
                                                                     t_depth, t_normal, t_prim, hit, depth, normal, prim);
    return hipGetLastError();
}

}  // namespace sdfgpu
