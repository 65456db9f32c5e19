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

#include "gpu_tree.hpp"
#include "kernels.hpp"

namespace sdfgpu {

// ------------------------------------------------------------ primitives --

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
        return __builtin_sqrtf((lx * lx + ly * ly) + lz * lz) - p0;
    }
    float qx = __builtin_fabsf(lx) - p0, qy = __builtin_fabsf(ly) - p1, qz = __builtin_fabsf(lz) - p2;
    float mx = qx > 0.f ? qx : 0.f, my = qy > 0.f ? qy : 0.f, mz = qz > 0.f ? qz : 0.f;
    float l = __builtin_sqrtf((mx * mx + my * my) + mz * mz);
    return l + __builtin_fminf(__builtin_fmaxf(qx, __builtin_fmaxf(qy, qz)), 0.0f);
}

__device__ __forceinline__ float leaf_eval(const GpuLeaf* __restrict__ L, float px, float py, float pz) {
    float lx, ly, lz;
    leaf_point(L->m, px, py, pz, lx, ly, lz);
    return prim_value(L->kind, L->param[0], L->param[1], L->param[2], lx, ly, lz);
}

// (distance, index) order of the reference's union combine composed over the
// tree (faux_shader.rs:27-29,87): smaller distance, then the
// LATER primitive.  Order-independent, so any visiting order gives the same
// answer as the reference's pre-order walk.
__device__ __forceinline__ void take_better(float d, uint32_t id, float& best, uint32_t& bid) {
    bool b = (d < best) || (d == best && (id > bid || bid == SDF_NO_PRIM));
    best = b ? d : best;
    bid = b ? id : bid;
}

// --------------------------------------------------------------- bounds ---

// Squared distance to an axis-aligned box (0 inside).  Not exact by design.
__device__ __forceinline__ float aabb_dist2(const float* __restrict__ lo, const float* __restrict__ hi, float px,
                                            float py, float pz) {
    float dx = __builtin_fmaxf(__builtin_fmaxf(lo[0] - px, px - hi[0]), 0.f);
    float dy = __builtin_fmaxf(__builtin_fmaxf(lo[1] - py, py - hi[1]), 0.f);
    float dz = __builtin_fmaxf(__builtin_fmaxf(lo[2] - pz, pz - hi[2]), 0.f);
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// MINDIST of faux_shader.rs:54-58 in the reference's own operation order
// (rows of full_inverse, w lane dropped: it contributes exactly +0 when the
// bottom row is (0,0,0,1) and scale.w == 0, which upload checks).
__device__ __forceinline__ float obb_mindist(const float* __restrict__ fi, const float* __restrict__ sc, float px,
                                             float py, float pz) {
    float tx = ((fi[0] * px + fi[1] * py) + fi[2] * pz) + fi[3];
    float ty = ((fi[4] * px + fi[5] * py) + fi[6] * pz) + fi[7];
    float tz = ((fi[8] * px + fi[9] * py) + fi[10] * pz) + fi[11];
    float qx = (__builtin_fabsf(tx) - 1.0f) * sc[0];
    float qy = (__builtin_fabsf(ty) - 1.0f) * sc[1];
    float qz = (__builtin_fabsf(tz) - 1.0f) * sc[2];
    float mx = qx > 0.f ? qx : 0.f, my = qy > 0.f ? qy : 0.f, mz = qz > 0.f ? qz : 0.f;
    float l = __builtin_sqrtf((mx * mx + mz * mz) + (my * my));   // glam SSE2 dot4 order, w term +0
    return l + __builtin_fminf(__builtin_fmaxf(qx, __builtin_fmaxf(qy, qz)), 0.0f);
}

struct TreeView {
    const GpuLeaf* __restrict__ leaf;
    const GpuInner* __restrict__ inner;
    const GpuInnerObb* __restrict__ inner_obb;
    uint32_t root;
    uint32_t depth;     // LDS stack levels per wave
    float margin_T;
};

struct WaveCounters {
    uint32_t wave_inner = 0, wave_leaf = 0;
    uint32_t lane_inner = 0, lane_sphere = 0, lane_box = 0, lane_queries = 0;
};

// Prune threshold for the current best.  AABB mode compares squared
// distances: a subtree is needed iff dist2(box) <= ((best + mg)^+)^2 * (1+6e-4);
// OBB mode compares MINDIST directly: pruned iff m > 0 && m > best + mg
// (the reference's test, faux_shader.rs:119-120, plus the margin).
template <int MODE>
__device__ __forceinline__ float prune_threshold(float best, float mg) {
    if constexpr (MODE == SDFGPU_BOUNDS_OBB) {
        return best + mg;
    } else {
        float bp = __builtin_fmaxf(best + mg, 0.f);
        return bp * bp * 1.0006f;
    }
}
template <int MODE>
__device__ __forceinline__ bool needed(float key, float thr) {
    if constexpr (MODE == SDFGPU_BOUNDS_OBB) return !(key > 0.f && key > thr);
    else return key <= thr;
}

// Wave-uniform branch-and-prune walk.  On return (best, bid) is the minimum
// over every primitive of the tree under take_better's order, for each lane
// with alive == true.  `stk` = this wave's LDS stack, [depth][64] floats.
template <int MODE, bool COUNT>
__device__ __forceinline__ void nn_traverse(const TreeView& T, float px, float py, float pz, bool alive, float& best,
                                            uint32_t& bid, float* __restrict__ stk, uint32_t lane, WaveCounters& wc) {
    uint32_t cur = T.root;
    if (cur == REF_NONE) return;
    const float mg_k = (MODE == SDFGPU_BOUNDS_OBB) ? 1.2e-5f : 3.0e-6f;
    const float mg = mg_k * (__builtin_fabsf(px) + __builtin_fabsf(py) + __builtin_fabsf(pz) + T.margin_T);
    uint32_t sp = 0;
    int refs = 0;  // lane k holds the reference of stack entry k
    if constexpr (COUNT) wc.lane_queries += alive ? 1u : 0u;
    for (;;) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        bool pop = false;
        if (cur & REF_LEAF) {
            const uint32_t id = cur & REF_INDEX;
            const GpuLeaf* __restrict__ L = T.leaf + id;
            float d = leaf_eval(L, px, py, pz);
            take_better(d, id, best, bid);
            if constexpr (COUNT) {
                wc.wave_leaf += 1;
                if (L->kind == SDF_PRIM_SPHERE) wc.lane_sphere += alive ? 1u : 0u;
                else wc.lane_box += alive ? 1u : 0u;
            }
            pop = true;
        } else {
            float k0, k1;
            uint32_t r0, r1;
            if constexpr (MODE == SDFGPU_BOUNDS_OBB) {
                const GpuInnerObb* __restrict__ N = T.inner_obb + cur;
                k0 = obb_mindist(N->fi0, N->sc0, px, py, pz);
                k1 = obb_mindist(N->fi1, N->sc1, px, py, pz);
                r0 = N->ref0;
                r1 = N->ref1;
            } else {
                const GpuInner* __restrict__ N = T.inner + cur;
                k0 = aabb_dist2(N->lo0, N->hi0, px, py, pz);
                k1 = aabb_dist2(N->lo1, N->hi1, px, py, pz);
                r0 = N->ref0;
                r1 = N->ref1;
            }
            if constexpr (COUNT) {
                wc.wave_inner += 1;
                wc.lane_inner += alive ? 1u : 0u;
            }
            const float thr = prune_threshold<MODE>(best, mg);
            const bool w0 = alive && needed<MODE>(k0, thr);
            const bool w1 = alive && needed<MODE>(k1, thr);
            const unsigned long long b0 = __ballot(w0), b1 = __ballot(w1);
            if (b0 != 0ull && b1 != 0ull) {
                // both needed by someone: majority vote on which is nearer
                const unsigned long long n0 = __ballot(w0 && (!w1 || k0 <= k1));
                const unsigned long long n1 = __ballot(w1 && (!w0 || k1 < k0));
                const bool first0 = __popcll(n0) >= __popcll(n1);
                const uint32_t far = first0 ? r1 : r0;
                const float kfar = first0 ? k1 : k0;
                stk[sp * 64u + lane] = kfar;
                refs = (lane == sp) ? (int)far : refs;   // v_writelane has no builtin here
                sp += 1;
                cur = first0 ? r0 : r1;
            } else if (b0 != 0ull) {
                cur = r0;
            } else if (b1 != 0ull) {
                cur = r1;
            } else {
                pop = true;
            }
        }
        if (pop) {
            bool found = false;
            while (sp != 0) {
                sp -= 1;
                const float k = stk[sp * 64u + lane];
                const float thr = prune_threshold<MODE>(best, mg);
                if (__ballot(alive && needed<MODE>(k, thr)) != 0ull) {
                    cur = (uint32_t)__builtin_amdgcn_readlane(refs, sp);
                    found = true;
                    break;
                }
            }
            if (!found) return;
        }
    }
}

__device__ __forceinline__ void flush_counters(const WaveCounters& wc, uint32_t lane, uint32_t steps, uint32_t rays,
                                               sdfgpu_stats* st) {
    // wave-reduce the per-lane counters, one atomic per counter per wave
    uint32_t v[6] = {wc.lane_queries, wc.lane_inner, wc.lane_sphere, wc.lane_box, steps, rays};
#pragma unroll
    for (int i = 0; i < 6; ++i)
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
    if (lane == 0) {
        atomicAdd((unsigned long long*)&st->queries, (unsigned long long)v[0]);
        atomicAdd((unsigned long long*)&st->inner_visits, (unsigned long long)v[1]);
        atomicAdd((unsigned long long*)&st->sphere_evals, (unsigned long long)v[2]);
        atomicAdd((unsigned long long*)&st->box_evals, (unsigned long long)v[3]);
        atomicAdd((unsigned long long*)&st->steps, (unsigned long long)v[4]);
        atomicAdd((unsigned long long*)&st->rays, (unsigned long long)v[5]);
        atomicAdd((unsigned long long*)&st->wave_inner, (unsigned long long)wc.wave_inner);
        atomicAdd((unsigned long long*)&st->wave_leaf, (unsigned long long)wc.wave_leaf);
    }
}

// ----------------------------------------------------------- K1: nearest --
// faux_shader::nearest_neighbor (faux_shader.rs:96-208) for n points, plus
// the primitive index.  One point per lane.  `order` (optional) is a
// permutation that makes consecutive lanes spatially close (Morton order);
// results are written at the original index.
template <int MODE, bool COUNT>
__global__ __launch_bounds__(256) void nearest_kernel(TreeView T, const float4* __restrict__ pts,
                                                      const uint32_t* __restrict__ order, uint32_t n,
                                                      float* __restrict__ out_dist, uint32_t* __restrict__ out_prim,
                                                      sdfgpu_stats* __restrict__ st) {
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* stk = smem + wave * T.depth * 64u;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    const bool alive = slot < n;
    const uint32_t src = alive ? (order ? order[slot] : slot) : 0u;
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
    if (alive) p = pts[src];
    float best = __builtin_inff();
    uint32_t bid = SDF_NO_PRIM;
    WaveCounters wc;
    nn_traverse<MODE, COUNT>(T, p.x, p.y, p.z, alive, best, bid, stk, lane, wc);
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
template <int K>
struct KBest {
    float d[K];
    uint32_t id[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = __builtin_inff(); id[j] = SDF_NO_PRIM; }
    }
    __device__ __forceinline__ static bool before(float da, uint32_t ia, float db, uint32_t ib) {
        return (da < db) || (da == db && ia > ib && ia != SDF_NO_PRIM) || (da == db && ib == SDF_NO_PRIM && ia != SDF_NO_PRIM);
    }
    __device__ __forceinline__ void insert(float nd, uint32_t nid) {
        if (!before(nd, nid, d[K - 1], id[K - 1])) return;
        d[K - 1] = nd;
        id[K - 1] = nid;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool sw = before(d[j], id[j], d[j - 1], id[j - 1]);
            const float td = d[j]; const uint32_t ti = id[j];
            d[j] = sw ? d[j - 1] : d[j];   id[j] = sw ? id[j - 1] : id[j];
            d[j - 1] = sw ? td : d[j - 1]; id[j - 1] = sw ? ti : id[j - 1];
        }
    }
};

template <int MODE, int K>
__global__ __launch_bounds__(256) void knn_kernel(TreeView T, const float4* __restrict__ pts,
                                                  const uint32_t* __restrict__ order, uint32_t n, uint32_t k,
                                                  float* __restrict__ out_dist, uint32_t* __restrict__ out_prim) {
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* __restrict__ stk = smem + wave * T.depth * 64u;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    const bool alive = slot < n;
    const uint32_t src = alive ? (order ? order[slot] : slot) : 0u;
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
    if (alive) p = pts[src];
    const float px = p.x, py = p.y, pz = p.z;
    KBest<K> kb;
    kb.init();

    uint32_t cur = T.root;
    if (cur != REF_NONE) {
        const float mg_k = (MODE == SDFGPU_BOUNDS_OBB) ? 1.2e-5f : 3.0e-6f;
        const float mg = mg_k * (__builtin_fabsf(px) + __builtin_fabsf(py) + __builtin_fabsf(pz) + T.margin_T);
        uint32_t sp = 0;
        int refs = 0;
        for (;;) {
            cur = __builtin_amdgcn_readfirstlane(cur);
            bool pop = false;
            if (cur & REF_LEAF) {
                const uint32_t id = cur & REF_INDEX;
                kb.insert(leaf_eval(T.leaf + id, px, py, pz), id);
                pop = true;
            } else {
                float k0, k1;
                uint32_t r0, r1;
                if constexpr (MODE == SDFGPU_BOUNDS_OBB) {
                    const GpuInnerObb* __restrict__ N = T.inner_obb + cur;
                    k0 = obb_mindist(N->fi0, N->sc0, px, py, pz);
                    k1 = obb_mindist(N->fi1, N->sc1, px, py, pz);
                    r0 = N->ref0; r1 = N->ref1;
                } else {
                    const GpuInner* __restrict__ N = T.inner + cur;
                    k0 = aabb_dist2(N->lo0, N->hi0, px, py, pz);
                    k1 = aabb_dist2(N->lo1, N->hi1, px, py, pz);
                    r0 = N->ref0; r1 = N->ref1;
                }
                const float thr = prune_threshold<MODE>(kb.d[K - 1], mg);
                const bool w0 = alive && needed<MODE>(k0, thr);
                const bool w1 = alive && needed<MODE>(k1, thr);
                const unsigned long long b0 = __ballot(w0), b1 = __ballot(w1);
                if (b0 != 0ull && b1 != 0ull) {
                    const unsigned long long n0 = __ballot(w0 && (!w1 || k0 <= k1));
                    const unsigned long long n1 = __ballot(w1 && (!w0 || k1 < k0));
                    const bool first0 = __popcll(n0) >= __popcll(n1);
                    stk[sp * 64u + lane] = first0 ? k1 : k0;
                    refs = (lane == sp) ? (int)(first0 ? r1 : r0) : refs;
                    sp += 1;
                    cur = first0 ? r0 : r1;
                } else if (b0 != 0ull) {
                    cur = r0;
                } else if (b1 != 0ull) {
                    cur = r1;
                } else {
                    pop = true;
                }
            }
            if (pop) {
                bool found = false;
                while (sp != 0) {
                    sp -= 1;
                    const float kk = stk[sp * 64u + lane];
                    const float thr = prune_threshold<MODE>(kb.d[K - 1], mg);
                    if (__ballot(alive && needed<MODE>(kk, thr)) != 0ull) {
                        cur = (uint32_t)__builtin_amdgcn_readlane(refs, sp);
                        found = true;
                        break;
                    }
                }
                if (!found) break;
            }
        }
    }
    if (alive) {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((uint32_t)j < k) {
                out_dist[(size_t)src * k + j] = kb.d[j];
                out_prim[(size_t)src * k + j] = kb.id[j];
            }
    }
}

// ------------------------------------------------- K3 + K4: march, normal --
// March loop (build-defined; restated by oracle_render):
//   t = t_min; repeat max_steps times: p = o + d*t; (delta, id) = NN(p);
//   if delta < eps_hit -> hit at depth t; t = t + delta; if t > t_max -> miss.
// The previous step's nearest primitive is evaluated first to seed `best`, so
// the walk only has to visit what can still beat it.
// Normal (src/main__.rs:151-161): tetrahedron taps of the HIT primitive only.
template <int MODE, bool COUNT>
__global__ __launch_bounds__(256) void render_kernel(TreeView T, sdfgpu_camera cam, uint32_t width, uint32_t height,
                                                     sdfgpu_march_params mp, uint32_t tile_first, uint32_t tile_stride,
                                                     uint32_t tile_count, uint32_t row_major,
                                                     uint8_t* __restrict__ o_hit, float* __restrict__ o_depth,
                                                     float* __restrict__ o_normal, uint32_t* __restrict__ o_prim,
                                                     sdfgpu_stats* __restrict__ st) {
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* stk = smem + wave * T.depth * 64u;
    const uint32_t tile_local = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (tile_local >= tile_count) return;
    const uint32_t tile = tile_first + tile_local * tile_stride;
    const uint32_t tiles_x = (width + 7u) >> 3;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t x = tx * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    const bool inside = x < width && y < height;

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

    float t = mp.t_min;
    bool alive = inside;
    bool hit = false;
    uint32_t hit_prim = SDF_NO_PRIM;
    uint32_t last = SDF_NO_PRIM;
    uint32_t steps = 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    WaveCounters wc;

    for (uint32_t s = 0; s < mp.max_steps; ++s) {
        if (__ballot(alive) == 0ull) break;
        if (alive) {
            px = ox + dx * t;
            py = oy + dy * t;
            pz = oz + dz * t;
        }
        float best = __builtin_inff();
        uint32_t bid = SDF_NO_PRIM;
        if (alive && last != SDF_NO_PRIM) {   // seed: per-lane gather of one 64-byte leaf
            best = leaf_eval(T.leaf + last, px, py, pz);
            bid = last;
            if constexpr (COUNT) {
                if (T.leaf[last].kind == SDF_PRIM_SPHERE) wc.lane_sphere += 1; else wc.lane_box += 1;
            }
        }
        nn_traverse<MODE, COUNT>(T, px, py, pz, alive, best, bid, stk, lane, wc);
        if (alive) {
            steps += 1;
            last = bid;
            if (best < mp.eps_hit) {
                hit = true;
                hit_prim = bid;
                alive = false;
            } else {
                t = t + best;
                if (t > mp.t_max) alive = false;
            }
        }
    }

    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (hit) {
        const GpuLeaf* L = T.leaf + hit_prim;
        const float h = mp.normal_h;
        // k.xyy, k.yyx, k.yxy, k.xxx with k = (1,-1)
        const float f0 = leaf_eval(L, px + h, py - h, pz - h);
        const float f1 = leaf_eval(L, px - h, py - h, pz + h);
        const float f2 = leaf_eval(L, px - h, py + h, pz - h);
        const float f3 = leaf_eval(L, px + h, py + h, pz + h);
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

    if (row_major) {
        if (inside) {
            const size_t o = (size_t)y * width + x;
            o_hit[o] = hit ? 1 : 0;
            if (o_depth) o_depth[o] = hit ? t : __builtin_inff();
            if (o_normal) { o_normal[3 * o] = nx; o_normal[3 * o + 1] = ny; o_normal[3 * o + 2] = nz; }
            if (o_prim) o_prim[o] = hit_prim;
        }
    } else {
        const size_t o = (size_t)tile_local * 64u + lane;
        o_hit[o] = hit ? 1 : 0;
        if (o_depth) o_depth[o] = hit ? t : __builtin_inff();
        if (o_normal) { o_normal[3 * o] = nx; o_normal[3 * o + 1] = ny; o_normal[3 * o + 2] = nz; }
        if (o_prim) o_prim[o] = hit_prim;
    }
    if constexpr (COUNT) flush_counters(wc, lane, steps, inside ? 1u : 0u, st);
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

// --------------------------------------------------------------- launch ---

static TreeView make_view(const DeviceTree& d) {
    TreeView v;
    v.leaf = d.leaf;
    v.inner = d.inner;
    v.inner_obb = d.inner_obb;
    v.root = d.root_ref;
    v.depth = d.stack_depth;
    v.margin_T = d.margin_T;
    return v;
}

template <bool COUNT>
static hipError_t launch_nearest_t(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n,
                                   float* out_dist, uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s) {
    if (n == 0) return hipSuccess;
    TreeView v = make_view(d);
    dim3 grid((n + 255u) / 256u), block(256);
    size_t lds = (size_t)4 * d.stack_depth * 64 * sizeof(float);
    if (d.bounds == SDFGPU_BOUNDS_OBB)
        nearest_kernel<SDFGPU_BOUNDS_OBB, COUNT><<<grid, block, lds, s>>>(v, (const float4*)pts, order, n, out_dist, out_prim, st);
    else
        nearest_kernel<SDFGPU_BOUNDS_AABB, COUNT><<<grid, block, lds, s>>>(v, (const float4*)pts, order, n, out_dist, out_prim, st);
    return hipGetLastError();
}

hipError_t launch_nearest(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n, float* out_dist,
                          uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s) {
    return st ? launch_nearest_t<true>(d, pts, order, n, out_dist, out_prim, st, s)
              : launch_nearest_t<false>(d, pts, order, n, out_dist, out_prim, nullptr, s);
}


template <int MODE>
static hipError_t launch_knn_m(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n, uint32_t k,
                               float* out_dist, uint32_t* out_prim, hipStream_t s) {
    TreeView v = make_view(d);
    dim3 grid((n + 255u) / 256u), block(256);
    size_t lds = (size_t)4 * d.stack_depth * 64 * sizeof(float);
    const float4* p4 = (const float4*)pts;
    if (k <= 2) knn_kernel<MODE, 2><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim);
    else if (k <= 4) knn_kernel<MODE, 4><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim);
    else if (k <= 8) knn_kernel<MODE, 8><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim);
    else knn_kernel<MODE, 16><<<grid, block, lds, s>>>(v, p4, order, n, k, out_dist, out_prim);
    return hipGetLastError();
}

hipError_t launch_knn(const DeviceTree& d, const float* pts, const uint32_t* order, uint32_t n, uint32_t k,
                      float* out_dist, uint32_t* out_prim, hipStream_t s) {
    if (n == 0 || k == 0) return hipSuccess;
    return d.bounds == SDFGPU_BOUNDS_OBB ? launch_knn_m<SDFGPU_BOUNDS_OBB>(d, pts, order, n, k, out_dist, out_prim, s)
                                         : launch_knn_m<SDFGPU_BOUNDS_AABB>(d, pts, order, n, k, out_dist, out_prim, s);
}

template <bool COUNT>
static hipError_t launch_render_t(const DeviceTree& d, const sdfgpu_camera& cam, uint32_t w, uint32_t h,
                                  const sdfgpu_march_params& mp, uint32_t tile_first, uint32_t tile_stride,
                                  uint32_t tile_count, bool row_major, uint8_t* hit, float* depth, float* normal,
                                  uint32_t* prim, sdfgpu_stats* st, hipStream_t s) {
    if (tile_count == 0) return hipSuccess;
    TreeView v = make_view(d);
    dim3 grid((tile_count + 3u) / 4u), block(256);
    size_t lds = (size_t)4 * d.stack_depth * 64 * sizeof(float);
    if (d.bounds == SDFGPU_BOUNDS_OBB)
        render_kernel<SDFGPU_BOUNDS_OBB, COUNT><<<grid, block, lds, s>>>(v, cam, w, h, mp, tile_first, tile_stride, tile_count,
                                                                        row_major ? 1u : 0u, hit, depth, normal, prim, st);
    else
        render_kernel<SDFGPU_BOUNDS_AABB, COUNT><<<grid, block, lds, s>>>(v, cam, w, h, mp, tile_first, tile_stride, tile_count,
                                                                         row_major ? 1u : 0u, hit, depth, normal, prim, st);
    return hipGetLastError();
}

hipError_t launch_render(const DeviceTree& d, const sdfgpu_camera& cam, uint32_t w, uint32_t h,
                         const sdfgpu_march_params& mp, uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count,
                         bool row_major, uint8_t* hit, float* depth, float* normal, uint32_t* prim, sdfgpu_stats* st,
                         hipStream_t s) {
    return st ? launch_render_t<true>(d, cam, w, h, mp, tile_first, tile_stride, tile_count, row_major, hit, depth, normal, prim, st, s)
              : launch_render_t<false>(d, cam, w, h, mp, tile_first, tile_stride, tile_count, row_major, hit, depth, normal, prim, nullptr, s);
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
