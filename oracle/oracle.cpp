// oracle.cpp -- CPU restatement of the reference's nearest-primitive path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under sdf_amd/ links, loads or calls this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
// and only as the checker / the timed CPU baseline.
//
// PARITY UNPINNED: the reference (Rust, holger-kibur/sdf) cannot be compiled
// here (no cargo/rustc, un-vendored deps) and holds no golden vectors, fixtures
// or seeded tests for this path.  This file follows the
// reference source line by line; the arithmetic of its third-party crates is
// restated from their published algorithms:
//   glam 0.20 (via bevy 0.6, sdf/Cargo.toml:11), SSE2 build:
//     Mat4 * Vec4  = ((c0*x + c1*y) + c2*z) + c3*w        per component
//     Vec4::length = sqrt((x^2 + z^2) + (y^2 + w^2))      (shuffle-add dot4)
//     Vec3::length = sqrt((x^2 + y^2) + z^2)              (scalar Vec3)
//   Rust core: f32::max/min (IEEE maxNum/minNum), no FMA contraction.
// Build with -ffp-contract=off -fno-fast-math (see Makefile).
//
// Each function cites the reference lines it follows.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>
#include <algorithm>

#include "../include/sdf_blocks.h"
#include "../include/sdfgpu.h"   // sdfgpu_camera / sdfgpu_march_params PODs only

namespace {

constexpr float INF = std::numeric_limits<float>::infinity();

struct V4 { float x, y, z, w; };

inline V4 ld(const SdfVec4& v) { return V4{v.v[0], v.v[1], v.v[2], v.v[3]}; }

// glam Mat4 * Vec4 (SSE2): res = c0*x; res += c1*y; res += c2*z; res += c3*w.
inline V4 mat_mul(const SdfMat4& m, V4 p) {
    V4 r;
    const float* c = m.m;
    r.x = ((c[0] * p.x + c[4] * p.y) + c[8] * p.z) + c[12] * p.w;
    r.y = ((c[1] * p.x + c[5] * p.y) + c[9] * p.z) + c[13] * p.w;
    r.z = ((c[2] * p.x + c[6] * p.y) + c[10] * p.z) + c[14] * p.w;
    r.w = ((c[3] * p.x + c[7] * p.y) + c[11] * p.z) + c[15] * p.w;
    return r;
}

// glam Vec4::length, SSE2 dot4 order.
inline float len4(V4 v) {
    return std::sqrt((v.x * v.x + v.z * v.z) + (v.y * v.y + v.w * v.w));
}
// glam Vec3::length (scalar).
inline float len3(float x, float y, float z) {
    return std::sqrt((x * x + y * y) + z * z);
}

// faux_shader.rs:27-33 -- local min/max: NaN in `a` selects b.
inline float fmin_local(float a, float b) { return a < b ? a : b; }
inline float fmax_local(float a, float b) { return a > b ? a : b; }

// Rust f32::max / f32::min (used on scalars in faux_shader.rs:57).
inline float rust_max(float a, float b) { return std::fmax(a, b); }
inline float rust_min(float a, float b) { return std::fmin(a, b); }

// SSE _mm_max_ps(a, b) = a > b ? a : b  (glam Vec4::max, faux_shader.rs:57).
inline float sse_max(float a, float b) { return a > b ? a : b; }
inline float sse_min(float a, float b) { return a < b ? a : b; }   // _mm_min_ps: second operand on NaN

// obb.rs:37-41 CmpFloat::cmp + Iterator::max (last of equal maxima wins).
inline float cmpfloat_max(float acc, float x) {
    // max_by: if compare(acc, x) == Greater { acc } else { x }
    return (acc > x) ? acc : x;
}

// faux_shader.rs:39-52
float maxdist(const SdfBoundingBoxBlock& bb, V4 p) {
    static const float S[8][3] = {{1, 1, 1},  {-1, 1, 1},  {1, -1, 1},  {-1, -1, 1},
                                  {1, 1, -1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};
    float best = 0.f;
    for (int i = 0; i < 8; ++i) {
        V4 c = mat_mul(bb.matrix, V4{S[i][0], S[i][1], S[i][2], 1.0f});
        V4 d{c.x - p.x, c.y - p.y, c.z - p.z, c.w - p.w};
        float l = len4(d);
        best = (i == 0) ? l : cmpfloat_max(best, l);
    }
    return best;
}

// faux_shader.rs:54-58
float mindist(const SdfBoundingBoxBlock& bb, V4 p) {
    V4 t = mat_mul(bb.full_inverse, p);
    V4 s = ld(bb.scale);
    V4 q{(std::fabs(t.x) - 1.0f) * s.x, (std::fabs(t.y) - 1.0f) * s.y,
         (std::fabs(t.z) - 1.0f) * s.z, (std::fabs(t.w) - 1.0f) * s.w};
    V4 qm{sse_max(q.x, 0.f), sse_max(q.y, 0.f), sse_max(q.z, 0.f), sse_max(q.w, 0.f)};
    return len4(qm) + rust_min(rust_max(q.x, rust_max(q.y, q.z)), 0.0f);
}

// ---- build-defined extensions (opt-in; default 0 = the crate as written) ----------------------
// Mirrors sdfgpu_extension (include/sdfgpu.h).  The reference carries the data for all three but
// evaluates none of them: smooth_radius is stored (elements.rs:207-211) and ignored (faux_shader.rs:87),
// SdfBoxFrame has no distance formula and prim_dispatch panics on op 5 (elements.rs:120-152,
// faux_shader.rs:65), and a clone subtree evaluates to +inf (faux_shader.rs:90).  Set by tests through
// oracle_set_extensions; test infrastructure, not thread-safe against concurrent setters.
enum { EXT_SMOOTH_UNION = 1, EXT_BOX_FRAME = 2, EXT_CLONE_REPAIR = 4 };
uint32_t g_ext = 0;

// Box frame (iq's sdBoxFrame, restated for op 5): b = dimension, e = thickness.
//   p = |p| - b ; q = |p + e| - e ; the three edge-bar distances, then min(min(d0, d1), d2)
inline float bar(float a, float b, float c) {   // |max(v,0)| + min(max(a, max(b, c)), 0) for v = (a, b, c)
    float l = len3(sse_max(a, 0.f), sse_max(b, 0.f), sse_max(c, 0.f));
    return l + rust_min(rust_max(a, rust_max(b, c)), 0.0f);
}
inline float box_frame(const float* dim, float e, V4 p) {
    float px = std::fabs(p.x) - dim[0], py = std::fabs(p.y) - dim[1], pz = std::fabs(p.z) - dim[2];
    float qx = std::fabs(px + e) - e, qy = std::fabs(py + e) - e, qz = std::fabs(pz + e) - e;
    float d0 = bar(px, qy, qz), d1 = bar(qx, py, qz), d2 = bar(qx, qy, pz);
    return fmin_local(fmin_local(d0, d1), d2);
}

// faux_shader.rs:60-67, plus the build-defined solid box (SURVEY C-7): the
// box formula is the crate's own exact box SDF (obb.rs:297-299) on xyz with
// the half extents in vec4s[0].
bool prim_dispatch(uint32_t code, const SdfOpSpecificBlock& sp, V4 p, float* out) {
    switch (code) {
        case SDF_PRIM_SPHERE:
            *out = len3(p.x, p.y, p.z) - sp.floats[0];
            return true;
        case SDF_PRIM_BOX: {
            float qx = std::fabs(p.x) - sp.vec4s[0].v[0];
            float qy = std::fabs(p.y) - sp.vec4s[0].v[1];
            float qz = std::fabs(p.z) - sp.vec4s[0].v[2];
            float l = len3(sse_max(qx, 0.f), sse_max(qy, 0.f), sse_max(qz, 0.f));
            *out = l + rust_min(rust_max(qx, rust_max(qy, qz)), 0.0f);
            return true;
        }
        case SDF_PRIM_BOX_FRAME:
            if (!(g_ext & EXT_BOX_FRAME)) return false;   // as written: panic
            *out = box_frame(sp.vec4s[0].v, sp.floats[0], p);
            return true;
        default:
            return false;  // panic!("Unsupported primitive op code")
    }
}

// faux_shader.rs:69-82.  Clone (op 1): [p - d * clamp(round(p / d), lo, hi), ZERO], per lane
// of the Vec4 (d.w = lo.w = hi.w = 0, so the w lane stays 1: 1/0 = inf, clamp to 0, 0*0 = 0).
// glam's SSE2 Vec4::round adds and subtracts 2^23 (the DirectXMath recipe), i.e. rounds half to even:
// restated as nearbyintf in the default rounding mode; clamp = max-then-min.  As written nothing the
// state machine returns depends on these points (uptree op 1 discards what a one-slot clone computed);
// under EXT_CLONE_REPAIR they are what the clone's child is evaluated at.  From memory of glam 0.20:
// unpinned like the rest of the glam arithmetic (file header).
bool downtree_dispatch(uint32_t code, const SdfOpSpecificBlock& sp, V4 p, V4 out[2]) {
    if (code == SDF_OP_UNION) {
        out[0] = p;
        out[1] = p;
        return true;
    }
    if (code == SDF_OP_CAA_CLONE) {
        const float* d = sp.vec4s[0].v;
        const float* lo = sp.vec4s[1].v;
        const float* hi = sp.vec4s[2].v;
        const float in[4] = {p.x, p.y, p.z, p.w};
        float o[4];
        for (int k = 0; k < 4; ++k) {
            float q = std::nearbyintf(in[k] / d[k]);
            q = sse_min(sse_max(q, lo[k]), hi[k]);
            o[k] = in[k] - d[k] * q;
        }
        out[0] = V4{o[0], o[1], o[2], o[3]};
        out[1] = V4{0, 0, 0, 0};
        return true;
    }
    return false;
}

struct DI { float d; uint32_t id; };

// faux_shader.rs:84-94 with the index carried: min(l, r) is
// `if l < r {l} else {r}` so ties and NaN-left go right.
// Clone (op 1, :90) returns right_dist.  A one-slot node only ever fills branch_dists[0], so
// the right one is still the +inf the frame was initialised with (:177): as written, a clone
// subtree contributes +inf to its parent (SURVEY A-6).  Restated, not repaired.
// Extensions: a union whose up-block carries smooth_radius k > 0 (elements.rs:207-211) combines with the
// quadratic smooth minimum  min(a,b) - h*h*k/4, h = max(k - |a-b|, 0)/k  (one rounded operation per
// step, in this order; the index is the smaller operand's, ties to the right as in the plain union);
// a repaired clone hands up the branch its one slot filled.
bool uptree_dispatch(uint32_t code, const SdfOpSpecificBlock& sp, DI l, DI r, DI* out) {
    if (code == SDF_OP_UNION) {
        *out = (l.d < r.d) ? l : r;
        const float k = sp.floats[0];
        if ((g_ext & EXT_SMOOTH_UNION) && k > 0.f) {
            float diff = l.d - r.d;
            float t = k - std::fabs(diff);
            t = sse_max(t, 0.f);
            float h = t / k;
            float q = h * h;
            q = q * k;
            q = q * 0.25f;
            out->d = out->d - q;
        }
        return true;
    }
    if (code == SDF_OP_CAA_CLONE) {
        *out = (g_ext & EXT_CLONE_REPAIR) ? l : r;
        return true;
    }
    return false;
}

// faux_shader.rs:12-25
struct Frame {
    V4 pts[2];
    DI d[2];
    uint32_t fill;
};

struct Tree {
    const SdfOperationBlock* dt;
    const SdfOperationUptreeBlock* ut;
    uint32_t n;
    std::vector<uint32_t> prim_id;   // pre-order ordinal of each primitive block
    std::vector<uint32_t> prim_node; // inverse
    std::vector<int32_t> parent;
    std::vector<uint8_t> dead;       // node lies in a clone subtree: its value never reaches the root
    uint32_t max_level;
};

int build_tree(Tree& t, const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut,
               uint32_t n) {
    t.dt = dt; t.ut = ut; t.n = n;
    t.prim_id.assign(n, SDF_NO_PRIM);
    t.parent.assign(n, -1);
    t.prim_node.clear();
    t.dead.assign(n, 0);
    t.max_level = 0;
    std::vector<int32_t> at_level(258, -1);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t lv = dt[i].level;
        if (lv == 0 || lv >= 255) return -7;
        if (lv > t.max_level) t.max_level = lv;
        t.parent[i] = (lv >= 2) ? at_level[lv - 1] : -1;
        at_level[lv] = (int32_t)i;
        if ((!(g_ext & EXT_CLONE_REPAIR) && !dt[i].is_primitive && dt[i].op_code == SDF_OP_CAA_CLONE) ||
            (t.parent[i] >= 0 && t.dead[t.parent[i]]))
            t.dead[i] = 1;
        if (dt[i].is_primitive) {
            t.prim_id[i] = (uint32_t)t.prim_node.size();
            t.prim_node.push_back(i);
        }
    }
    return 0;
}

// faux_shader.rs:96-208, literally, with (distance, index) pairs.
// prune=false disables the union prune at :117-126 ("exhaustive").
// faithful_alloc=true reproduces the per-call 256-frame Vec of :100.
int faux_nearest_one(const Tree& t, V4 point, bool prune, Frame* stack, DI* out) {
    uint32_t dt_index = 0, ut_index = 0, last_dt_level = 0;
    const uint32_t n = t.n;
    stack[1].pts[0] = point;
    stack[1].pts[1] = V4{0, 0, 0, 0};
    stack[1].d[0] = DI{INF, SDF_NO_PRIM};
    stack[1].d[1] = DI{INF, SDF_NO_PRIM};
    stack[1].fill = 0;

    while (dt_index < n) {
        const SdfOperationBlock& b = t.dt[dt_index];
        const Frame& f0 = stack[b.level];
        if (f0.fill > 1) return -5;  // index out of bounds panic in the reference
        V4 dt_point = f0.pts[f0.fill];
        float cmp = f0.d[0].d;

        if (prune && b.parent_is_union) {                       // :117
            float m = mindist(b.bounding_box, dt_point);        // :118
            if (m > 0.f && (m > cmp || m > maxdist(b.other_box, dt_point))) {  // :119-121
                dt_index += 1 + b.len;                          // :122
                ut_index += 1 + b.len;                          // :123
                continue;
            }
        }

        if (b.level < last_dt_level) {                          // :129
            uint32_t last_ut_level = 0xFFFFFFFFu;
            for (;;) {
                if (ut_index >= n) return -5;
                const SdfOperationUptreeBlock& u = t.ut[ut_index];
                if (u.level >= last_ut_level) break;            // :136
                const Frame& cf = stack[u.level + 1];
                Frame& uf = stack[u.level];
                if (uf.fill > 1) return -5;
                DI r;
                if (!uptree_dispatch(u.op_code, u.op_specific, cf.d[0], cf.d[1], &r)) return -6;
                uf.d[uf.fill] = r;                              // :146
                uf.fill += 1;
                last_ut_level = u.level;
                ut_index += 1;
            }
        }

        if (b.is_primitive) {                                   // :160
            Frame& f = stack[b.level];
            if (f.fill > 1) return -5;
            float d;
            if (!prim_dispatch(b.op_code, b.op_specific,
                               mat_mul(b.bounding_box.trans_inverse, dt_point), &d))
                return -6;
            f.d[f.fill] = DI{d, t.prim_id[dt_index]};
            f.fill += 1;
            ut_index += 1;
        } else {                                                // :170
            Frame& cf = stack[b.level + 1];
            if (!downtree_dispatch(b.op_code, b.op_specific,
                                   mat_mul(b.bounding_box.trans_inverse, dt_point), cf.pts))
                return -6;
            cf.fill = 0;
            cf.d[0] = DI{INF, SDF_NO_PRIM};
            cf.d[1] = DI{INF, SDF_NO_PRIM};
        }
        last_dt_level = b.level;                                // :181
        dt_index += 1;
    }

    while (ut_index < n) {                                      // :186
        const SdfOperationUptreeBlock& u = t.ut[ut_index];
        const Frame& cf = stack[u.level + 1];
        Frame& uf = stack[u.level];
        if (uf.fill > 1) return -5;
        DI r;
        if (!uptree_dispatch(u.op_code, u.op_specific, cf.d[0], cf.d[1], &r)) return -6;
        uf.d[uf.fill] = r;
        uf.fill += 1;
        ut_index += 1;
    }
    *out = stack[1].d[0];                                       // :207
    return 0;
}

// Leaf value of primitive node `node` at world point p: every path node's
// trans_inverse applied in sequence, root first (SURVEY A.2 last bullet;
// faux_shader.rs:165,175).
int eval_leaf(const Tree& t, uint32_t node, V4 p, float* out) {
    uint32_t path[256];
    int np = 0;
    for (int32_t i = (int32_t)node; i >= 0; i = t.parent[i]) path[np++] = (uint32_t)i;
    for (int k = np - 1; k >= 1; --k) {
        const SdfOperationBlock& b = t.dt[path[k]];
        V4 pts[2];
        if (!downtree_dispatch(b.op_code, b.op_specific,
                               mat_mul(b.bounding_box.trans_inverse, p), pts))
            return -6;
        p = pts[0];
    }
    const SdfOperationBlock& b = t.dt[node];
    if (!prim_dispatch(b.op_code, b.op_specific, mat_mul(b.bounding_box.trans_inverse, p), out))
        return -6;
    return 0;
}

// Order of (distance, index) implied by faux_shader.rs:27-29 composed over a
// union tree: smaller distance first, then the LATER primitive.
inline bool better(DI a, DI b) { return a.d < b.d || (a.d == b.d && a.id > b.id); }

// "cpu_tuned" (BASELINE.md): same result as the exhaustive walk on
// consistent buffers, using the buffer's own boxes with real best-so-far
// pruning, nearer child first.  Recursion over the pre-order array.
struct Tuned {
    const Tree& t;
    float margin;
    DI best;
    void leaf(uint32_t i, V4 p) {
        const SdfOperationBlock& b = t.dt[i];
        float d;
        if (prim_dispatch(b.op_code, b.op_specific, mat_mul(b.bounding_box.trans_inverse, p), &d)) {
            DI c{d, t.prim_id[i]};
            if (better(c, best)) best = c;
        }
    }
    void visit(uint32_t i, V4 p) {
        const SdfOperationBlock& b = t.dt[i];
        if (t.dead[i]) return;   // clone subtree: +inf as written
        if (b.is_primitive) { leaf(i, p); return; }
        V4 pc = mat_mul(b.bounding_box.trans_inverse, p);
        uint32_t l = i + 1;
        uint32_t end = i + 1 + b.len;
        uint32_t r = l + 1 + t.dt[l].len;
        if (r >= end) { visit(l, pc); return; }
        float ml = mindist(t.dt[l].bounding_box, pc);
        float mr = mindist(t.dt[r].bounding_box, pc);
        uint32_t a = l, c = r;
        float ma = ml, mc = mr;
        if (mr < ml) { a = r; c = l; ma = mr; mc = ml; }
        if (!(ma > best.d + margin)) visit(a, pc);
        if (!(mc > best.d + margin)) visit(c, pc);
    }
};

int tuned_nearest_one(const Tree& t, V4 p, DI* out) {
    Tuned q{t, 0.f, DI{INF, SDF_NO_PRIM}};
    // absolute slack for the rounding of MINDIST and of the leaf chain
    q.margin = 1e-4f * (1.0f + std::fabs(p.x) + std::fabs(p.y) + std::fabs(p.z));
    if (t.n) q.visit(0, p);
    *out = q.best;
    return 0;
}

enum { MODE_FAUX = 0, MODE_EXHAUSTIVE = 1, MODE_TUNED = 2, MODE_FAUX_FAITHFUL = 3 };

int nearest_one(const Tree& t, V4 p, int mode, Frame* stack, DI* out) {
    switch (mode) {
        case MODE_FAUX: return faux_nearest_one(t, p, true, stack, out);
        case MODE_EXHAUSTIVE: return faux_nearest_one(t, p, false, stack, out);
        case MODE_TUNED: return tuned_nearest_one(t, p, out);
        case MODE_FAUX_FAITHFUL: {
            // faux_shader.rs:100: vec![LevelStackEntry::ZERO; 256] per call
            std::vector<Frame> st(256);
            std::memset(st.data(), 0, sizeof(Frame) * 256);
            return faux_nearest_one(t, p, true, st.data(), out);
        }
    }
    return -1;
}

template <class F>
int parallel_for(size_t n, int threads, F&& fn) {
    if (threads <= 1 || n < 2) return fn(0, n);
    std::vector<std::thread> th;
    std::vector<int> rc((size_t)threads, 0);
    size_t chunk = (n + (size_t)threads - 1) / (size_t)threads;
    for (int k = 0; k < threads; ++k) {
        size_t a = std::min(n, chunk * (size_t)k), b = std::min(n, a + chunk);
        th.emplace_back([&, k, a, b] { rc[(size_t)k] = fn(a, b); });
    }
    for (auto& x : th) x.join();
    for (int r : rc) if (r) return r;
    return 0;
}

}  // namespace

extern "C" {

// Opt-in extensions for the calls that follow (bit mask EXT_*; 0 = as written).  Returns the previous mask.
uint32_t oracle_set_extensions(uint32_t mask) {
    uint32_t old = g_ext;
    g_ext = mask;
    return old;
}

// n points through faux_shader::nearest_neighbor.  mode: 0 as written,
// 1 pruning disabled, 2 tuned, 3 as written incl. the per-call allocation.
int oracle_nearest(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t n,
                   const float* pts_xyzw, size_t npts, int mode, int threads, float* out_dist,
                   uint32_t* out_prim) {
    Tree t;
    int rc = build_tree(t, dt, ut, n);
    if (rc) return rc;
    return parallel_for(npts, threads, [&](size_t a, size_t b) -> int {
        std::vector<Frame> stack(256);
        std::memset(stack.data(), 0, sizeof(Frame) * 256);
        for (size_t i = a; i < b; ++i) {
            V4 p{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
            DI r;
            int e = nearest_one(t, p, mode, stack.data(), &r);
            if (e) return e;
            out_dist[i] = r.d;
            if (out_prim) out_prim[i] = r.id;
        }
        return 0;
    });
}

// k nearest: every primitive evaluated through its own path chain, then the
// k best under the (distance, later-index-first) order; padded (+inf, NO_PRIM).
int oracle_knn(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t n,
               const float* pts_xyzw, size_t npts, uint32_t k, int threads, float* out_dist,
               uint32_t* out_prim) {
    Tree t;
    int rc = build_tree(t, dt, ut, n);
    if (rc) return rc;
    return parallel_for(npts, threads, [&](size_t a, size_t b) -> int {
        std::vector<DI> all;
        all.reserve(t.prim_node.size());
        for (size_t i = a; i < b; ++i) {
            V4 p{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
            all.clear();
            for (size_t j = 0; j < t.prim_node.size(); ++j) {
                if (t.dead[t.prim_node[j]]) continue;   // below a clone: never reaches the root (uptree_dispatch)
                float d;
                int e = eval_leaf(t, t.prim_node[j], p, &d);
                if (e) return e;
                all.push_back(DI{d, (uint32_t)j});
            }
            size_t kk = std::min<size_t>(k, all.size());
            std::partial_sort(all.begin(), all.begin() + (long)kk, all.end(), better);
            for (uint32_t j = 0; j < k; ++j) {
                out_dist[i * k + j] = j < kk ? all[j].d : INF;
                out_prim[i * k + j] = j < kk ? all[j].id : SDF_NO_PRIM;
            }
        }
        return 0;
    });
}

// The same k nearest with real pruning (the CPU baseline of the kNN leg: the exhaustive form above is
// O(N) per query and would be a strawman): the walk of `Tuned`, pruning against the k-th best so far.
struct TunedK {
    const Tree& t;
    float margin;
    uint32_t k;
    std::vector<DI> best;   // ascending under better(), at most k entries
    float radius() const { return best.size() < k ? INF : best.back().d; }
    void offer(DI c) {
        if (best.size() == k && !better(c, best.back())) return;
        auto it = std::upper_bound(best.begin(), best.end(), c, better);
        best.insert(it, c);
        if (best.size() > k) best.pop_back();
    }
    void visit(uint32_t i, V4 p) {
        const SdfOperationBlock& b = t.dt[i];
        if (t.dead[i]) return;
        if (b.is_primitive) {
            float d;
            if (prim_dispatch(b.op_code, b.op_specific, mat_mul(b.bounding_box.trans_inverse, p), &d)) offer(DI{d, t.prim_id[i]});
            return;
        }
        V4 pc = mat_mul(b.bounding_box.trans_inverse, p);
        uint32_t l = i + 1, end = i + 1 + b.len, r = l + 1 + t.dt[l].len;
        if (r >= end) { visit(l, pc); return; }
        float ml = mindist(t.dt[l].bounding_box, pc), mr = mindist(t.dt[r].bounding_box, pc);
        uint32_t a = l, c = r;
        float ma = ml, mc = mr;
        if (mr < ml) { a = r; c = l; ma = mr; mc = ml; }
        if (!(ma > radius() + margin)) visit(a, pc);
        if (!(mc > radius() + margin)) visit(c, pc);
    }
};

int oracle_knn_tuned(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t n,
                     const float* pts_xyzw, size_t npts, uint32_t k, int threads, float* out_dist,
                     uint32_t* out_prim) {
    Tree t;
    int rc = build_tree(t, dt, ut, n);
    if (rc) return rc;
    return parallel_for(npts, threads, [&](size_t a, size_t b) -> int {
        TunedK q{t, 0.f, k, {}};
        q.best.reserve(k + 1);
        for (size_t i = a; i < b; ++i) {
            V4 p{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
            q.best.clear();
            q.margin = 1e-4f * (1.0f + std::fabs(p.x) + std::fabs(p.y) + std::fabs(p.z));
            if (t.n && k) q.visit(0, p);
            for (uint32_t j = 0; j < k; ++j) {
                out_dist[i * k + j] = j < q.best.size() ? q.best[j].d : INF;
                out_prim[i * k + j] = j < q.best.size() ? q.best[j].id : SDF_NO_PRIM;
            }
        }
        return 0;
    });
}

// Value of one primitive's SDF at points (used by normal tests).
int oracle_eval_prim(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t n,
                     uint32_t prim, const float* pts_xyzw, size_t npts, float* out) {
    Tree t;
    int rc = build_tree(t, dt, ut, n);
    if (rc) return rc;
    if (prim >= t.prim_node.size()) return -1;
    for (size_t i = 0; i < npts; ++i) {
        V4 p{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
        int e = eval_leaf(t, t.prim_node[prim], p, &out[i]);
        if (e) return e;
    }
    return 0;
}

// Bounds of faux_shader.rs:39-58 for node `node` at points (tests of A-2/A-3).
int oracle_bounds(const SdfOperationBlock* dt, uint32_t n, uint32_t node, const float* pts_xyzw,
                  size_t npts, float* out_mindist, float* out_maxdist) {
    if (node >= n) return -1;
    for (size_t i = 0; i < npts; ++i) {
        V4 p{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
        out_mindist[i] = mindist(dt[node].bounding_box, p);
        out_maxdist[i] = maxdist(dt[node].bounding_box, p);
    }
    return 0;
}

// Sphere trace + normals (DESIGN.md "march loop"; normals follow
// src/main__.rs:151-161).  Pixels pix_first + k*pix_stride, k < pix_count, of
// the row-major frame are rendered (a bounded sample for the CPU baseline).
// out_steps may be NULL.
int oracle_render(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t n,
                  const sdfgpu_camera* cam, uint32_t width, uint32_t height,
                  const sdfgpu_march_params* mp, int mode, int threads, size_t pix_first,
                  size_t pix_stride, size_t pix_count, uint8_t* hit, float* depth, float* normal_xyz, uint32_t* prim,
                  uint32_t* out_steps) {
    Tree t;
    int rc = build_tree(t, dt, ut, n);
    if (rc) return rc;
    (void)height;
    return parallel_for(pix_count, threads, [&](size_t a, size_t b) -> int {
        std::vector<Frame> stack(256);
        std::memset(stack.data(), 0, sizeof(Frame) * 256);
        for (size_t k = a; k < b; ++k) {
            size_t pix = pix_first + k * pix_stride;
            float fx = (float)(uint32_t)(pix % width), fy = (float)(uint32_t)(pix / width);
            float dir[3], o[3];
            float pw[3];
            for (int c = 0; c < 3; ++c) {
                pw[c] = (cam->origin[c] + cam->du[c] * fx) + cam->dv[c] * fy;
                o[c] = cam->eye[c];
                dir[c] = pw[c] - o[c];
            }
            float l = len3(dir[0], dir[1], dir[2]);
            for (int c = 0; c < 3; ++c) dir[c] = dir[c] / l;

            // Domain of the march (build-defined like the loop): the eye's coordinates and t_min
            // within 2^60 and the direction finite, else the ray is a miss outright; t stays within
            // min(t_max, 2^60), `!(t <= t_hi)` ending the ray for a NaN as well.  Every point the loop
            // evaluates then has coordinates below 2^62, where no leaf value is NaN.
            const float LIM = 1152921504606846976.0f;   // 2^60
            bool ray_ok = std::fabs(mp->t_min) <= LIM;
            for (int c = 0; c < 3; ++c) ray_ok = ray_ok && std::fabs(o[c]) <= LIM && std::fabs(dir[c]) <= LIM;   // false for NaN; |dir| <= 1 when finite
            const float t_hi = std::fmin(mp->t_max, LIM);
            float tt = mp->t_min;
            uint8_t h = 0;
            uint32_t pid = SDF_NO_PRIM;
            uint32_t steps = 0;
            V4 p{0, 0, 0, 1};
            for (uint32_t s = 0; ray_ok && s < mp->max_steps; ++s) {
                p = V4{o[0] + dir[0] * tt, o[1] + dir[1] * tt, o[2] + dir[2] * tt, 1.0f};
                DI r;
                int e = nearest_one(t, p, mode, stack.data(), &r);
                if (e) return e;
                steps += 1;
                if (r.d < mp->eps_hit) { h = 1; pid = r.id; break; }
                tt = tt + r.d;
                if (!(tt <= t_hi)) break;
            }
            float nx = 0, ny = 0, nz = 0;
            if (h) {
                // calcSphereNormal, src/main__.rs:151-161: k = (1,-1)
                //  k.xyy f(p+k.xyy h) + k.yyx f(p+k.yyx h) + k.yxy f(p+k.yxy h) + k.xxx f(p+k.xxx h)
                static const float K[4][3] = {{1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, 1, 1}};
                float f[4];
                for (int j = 0; j < 4; ++j) {
                    V4 q{p.x + K[j][0] * mp->normal_h, p.y + K[j][1] * mp->normal_h,
                         p.z + K[j][2] * mp->normal_h, 1.0f};
                    int e = eval_leaf(t, t.prim_node[pid], q, &f[j]);
                    if (e) return e;
                }
                float sx = ((K[0][0] * f[0] + K[1][0] * f[1]) + K[2][0] * f[2]) + K[3][0] * f[3];
                float sy = ((K[0][1] * f[0] + K[1][1] * f[1]) + K[2][1] * f[2]) + K[3][1] * f[3];
                float sz = ((K[0][2] * f[0] + K[1][2] * f[1]) + K[2][2] * f[2]) + K[3][2] * f[3];
                float nl = len3(sx, sy, sz);
                if (nl > 0.f) { nx = sx / nl; ny = sy / nl; nz = sz / nl; }
            }
            hit[k] = h;
            if (depth) depth[k] = h ? tt : INF;
            if (normal_xyz) { normal_xyz[3 * k] = nx; normal_xyz[3 * k + 1] = ny; normal_xyz[3 * k + 2] = nz; }
            if (prim) prim[k] = pid;
            if (out_steps) out_steps[k] = steps;
        }
        return 0;
    });
}

int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
