// compact.cpp -- validation of crate-format tree buffers and compaction into
// the resident GPU tree (gpu_tree.hpp).  Host code, runs once per upload.
//
// What is validated here is exactly what the kernels rely on, so that a kernel
// can never index out of bounds or loop forever (SURVEY 8b "Errors"): the
// invariants ExpandedSdfNode::make_buffer establishes (sdf/src/node.rs:75-138,
// SURVEY A.1) and the op codes faux_shader.rs:60-94 dispatches on.
#include "gpu_tree.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>

#include <sched.h>

namespace sdfgpu {
namespace {

struct Box3 {
    double lo[3], hi[3];
};

struct Ctx {
    const SdfOperationBlock* dt;
    uint32_t n;
    uint32_t bounds;
    std::vector<uint32_t> prim_id;
    std::vector<uint32_t> child0, child1;  // REF_NONE if absent (node indices)
    std::vector<uint8_t> dead;             // node lies in a clone subtree (evaluates to +inf as written)
    std::vector<uint32_t> live;            // primitives below the node that are not dead
    CompactTree* out;
    std::string* err;
    uint32_t max_inner_depth = 0;
};

bool finite16(const float* m) {
    for (int i = 0; i < 16; ++i)
        if (!std::isfinite(m[i])) return false;
    return true;
}

bool is_identity(const float* m) {
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            if (m[4 * c + r] != (r == c ? 1.f : 0.f)) return false;
    return true;
}

bool bottom_row_exact(const float* m) { return m[3] == 0.f && m[7] == 0.f && m[11] == 0.f && m[15] == 1.f; }

// |M^T M - I| <= RIGID_TOL elementwise.  Pruned bounds compare world-space box distances with leaf
// values, which is sound only while a leaf value is at least (1 - ~RIGID_TOL/2) x the world distance:
// the kernels' relative slack (1.0006 on squared distances, 0.9996 on kept bounds) covers that with
// room to spare; the march's budget decrement uses the measured constant (lipschitz3) itself.
constexpr double RIGID_TOL = 2e-5;
bool rigid3(const float* m, double tol) {
    // columns c of the 3x3: m[4c + r]
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int r = 0; r < 3; ++r) s += (double)m[4 * a + r] * (double)m[4 * b + r];
            if (std::fabs(s - (a == b ? 1.0 : 0.0)) > tol) return false;
        }
    return true;
}

// Upper bound on the largest singular value of the 3x3 linear part (sqrt of the infinity norm of
// M^T M): a leaf value f(p) = PRIM(M p + t) changes by at most this much per unit the point moves.
double lipschitz3(const float* m) {
    double worst = 0.0;
    for (int a = 0; a < 3; ++a) {
        double row = 0.0;
        for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int r = 0; r < 3; ++r) s += (double)m[4 * a + r] * (double)m[4 * b + r];
            row += std::fabs(s);
        }
        worst = std::max(worst, row);
    }
    return std::sqrt(worst);
}

float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}

// Centre / half-extent form of a box, rounded so that [c-h, c+h] contains it.
void set_box(float c[3], float h[3], const Box3& b) {
    for (int a = 0; a < 3; ++a) {
        c[a] = (float)(0.5 * (b.lo[a] + b.hi[a]));
        const double cd = (double)c[a];
        h[a] = round_up(std::max(b.hi[a] - cd, cd - b.lo[a]));
    }
}

// World AABB of one primitive from its world->primitive transform.
// A world -> frame map q = r p + t in double (what a chain of trans_inverse matrices composes to).
struct Aff {
    double r[3][3];
    double t[3];
};
Aff aff_of(const float* m) {   // column-major 4x4 with an affine bottom row
    Aff a;
    for (int i = 0; i < 3; ++i) {
        for (int c = 0; c < 3; ++c) a.r[i][c] = m[4 * c + i];
        a.t[i] = m[12 + i];
    }
    return a;
}
Aff aff_identity() {
    Aff a{};
    for (int i = 0; i < 3; ++i) a.r[i][i] = 1.0;
    return a;
}
Aff aff_then(const Aff& first, const Aff& second) {   // second(first(p))
    Aff o;
    for (int i = 0; i < 3; ++i) {
        for (int c = 0; c < 3; ++c) o.r[i][c] = second.r[i][0] * first.r[0][c] + second.r[i][1] * first.r[1][c] + second.r[i][2] * first.r[2][c];
        o.t[i] = second.r[i][0] * first.t[0] + second.r[i][1] * first.t[1] + second.r[i][2] * first.t[2] + second.t[i];
    }
    return o;
}
double aff_lipschitz(const Aff& a) {   // as lipschitz3, for a composed map
    double worst = 0.0;
    for (int x = 0; x < 3; ++x) {
        double row = 0.0;
        for (int y = 0; y < 3; ++y) {
            double sum = 0;
            for (int i = 0; i < 3; ++i) sum += a.r[i][x] * a.r[i][y];
            row += std::fabs(sum);
        }
        worst = std::max(worst, row);
    }
    return std::sqrt(worst);
}

// World AABB of one primitive from its world -> primitive map (its own trans_inverse, or the whole chain of
// its ancestors' and its own composed).  extra_pad: rounding accumulated over a chain, in world units.
bool aff_aabb(const Aff& M, const SdfOperationBlock& b, double extra_pad, Box3* box) {
    const double (*A)[3] = M.r;
    double I[3][3];
    double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    if (!(std::fabs(det) > 1e-12)) return false;
    double r = 1.0 / det;
    I[0][0] = c00 * r; I[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * r; I[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * r;
    I[1][0] = c01 * r; I[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * r; I[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * r;
    I[2][0] = c02 * r; I[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * r; I[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * r;
    const double* t = M.t;
    double ext[3], cen[3];
    for (int i = 0; i < 3; ++i) {
        cen[i] = -(I[i][0] * t[0] + I[i][1] * t[1] + I[i][2] * t[2]);
        if (b.op_code == SDF_PRIM_SPHERE) {
            double rad = std::fabs((double)b.op_specific.floats[0]);
            ext[i] = rad * std::sqrt(I[i][0] * I[i][0] + I[i][1] * I[i][1] + I[i][2] * I[i][2]);
        } else {
            const float* h = b.op_specific.vec4s[0].v;
            ext[i] = std::fabs(I[i][0]) * std::fabs((double)h[0]) + std::fabs(I[i][1]) * std::fabs((double)h[1]) +
                     std::fabs(I[i][2]) * std::fabs((double)h[2]);
        }
        // outward padding: the leaf value the kernels compute carries the
        // rounding of a 3x4 mat-vec at |coordinate| magnitude.
        double pad = 1e-4 * ext[i] + 64.0 * (double)FLT_EPSILON * (std::fabs(cen[i]) + ext[i] + 1.0) + extra_pad;
        box->lo[i] = cen[i] - ext[i] - pad;
        box->hi[i] = cen[i] + ext[i] + pad;
    }
    return true;
}
bool leaf_aabb(const SdfOperationBlock& b, Box3* box) { return aff_aabb(aff_of(b.bounding_box.trans_inverse.m), b, 0.0, box); }

void fill_obb(const SdfBoundingBoxBlock& bb, float fi[12], float sc[3]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) fi[4 * r + c] = bb.full_inverse.m[4 * c + r];
    sc[0] = bb.scale.v[0]; sc[1] = bb.scale.v[1]; sc[2] = bb.scale.v[2];
}

// Returns the reference of node i in the compact tree and its box / depth.
uint32_t emit(Ctx& c, uint32_t i, Box3* box, uint32_t* depth) {
    const SdfOperationBlock& b = c.dt[i];
    if (c.live[i] == 0) {   // clone subtree (or a union of nothing but those): +inf, not resident
        *depth = 0;
        return REF_NONE;
    }
    if (b.is_primitive) {
        *depth = 0;
        if (c.bounds == SDFGPU_BOUNDS_AABB) leaf_aabb(b, box);
        return REF_LEAF | c.prim_id[i];
    }
    uint32_t k0 = c.child0[i], k1 = c.child1[i];
    if (k1 != REF_NONE && c.live[k1] == 0) k1 = REF_NONE;
    if (k0 != REF_NONE && c.live[k0] == 0) { k0 = k1; k1 = REF_NONE; }
    if (k1 == REF_NONE) return emit(c, k0, box, depth);  // 1-slot union: identity pass-through
    uint32_t self = 0;
    if (c.bounds == SDFGPU_BOUNDS_OBB) {
        self = (uint32_t)c.out->inner_obb.size();
        c.out->inner_obb.emplace_back();
    } else {
        self = (uint32_t)c.out->inner.size();
        c.out->inner.emplace_back();
    }
    Box3 b0{}, b1{};
    uint32_t d0 = 0, d1 = 0;
    uint32_t r0 = emit(c, k0, &b0, &d0);
    uint32_t r1 = emit(c, k1, &b1, &d1);
    *depth = 1 + std::max(d0, d1);
    if (c.bounds == SDFGPU_BOUNDS_OBB) {
        GpuInnerObb& g = c.out->inner_obb[self];
        std::memset(&g, 0, sizeof g);
        fill_obb(c.dt[k0].bounding_box, g.fi0, g.sc0);
        fill_obb(c.dt[k1].bounding_box, g.fi1, g.sc1);
        g.ref0 = r0;
        g.ref1 = r1;
    } else {
        GpuInner& g = c.out->inner[self];
        std::memset(&g, 0, sizeof g);
        if (c.bounds == SDFGPU_BOUNDS_AABB) {
            set_box(g.c0, g.h0, b0);
            set_box(g.c1, g.h1, b1);
            for (int a = 0; a < 3; ++a) {
                box->lo[a] = std::min(b0.lo[a], b1.lo[a]);
                box->hi[a] = std::max(b0.hi[a], b1.hi[a]);
            }
        } else {  // NONE: boxes that contain everything -> nothing is ever pruned
            for (int a = 0; a < 3; ++a) {
                g.c0[a] = g.c1[a] = 0.f;
                g.h0[a] = g.h1[a] = FLT_MAX;
            }
        }
        g.ref0 = r0;
        g.ref1 = r1;
    }
    return self;
}

// ---- BVH mode: the library's own hierarchy over the primitives -------------
// Binned surface-area-heuristic build (16 bins per axis) over the primitives'
// world AABBs.  The buffer's topology is not used at all here: the result of a
// query is the minimum over ALL primitives under an order-independent tie rule,
// so any hierarchy over the same leaves returns the same (distance, index).
struct PrimBox {
    Box3 b;
    double c[3];
    uint32_t id;
};

double half_area(const Box3& b) {
    double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
void grow(Box3& a, const Box3& b) {
    for (int k = 0; k < 3; ++k) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); }
}
Box3 empty_box() { return Box3{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}; }

// `budget` = inner levels this subtree may still use (the per-wave stack of the kernels and
// their LDS footprint are sized by it): a split is only eligible when both sides still fit a
// balanced tree of budget-1 levels; when no SAH candidate does, the set is halved at the median.
//
// Records are written in pre-order into an array sized beforehand (a subtree of m primitives has
// m - 1 internal nodes, so the right child of a node at index s whose left side holds ml primitives
// sits at s + ml): subtrees own disjoint ranges of both the primitive array and the record array, so
// the two sides of the upper levels are built by separate threads and the result is the same array,
// byte for byte, whatever the thread count.
struct SahBuild {
    std::vector<PrimBox>& p;
    GpuInner* inner;
    static constexpr int NB = 16;
    static constexpr uint32_t FORK_MIN = 4096;   // primitives below which a side is not worth a thread

    uint32_t node(uint32_t lo, uint32_t hi, uint32_t budget, uint32_t self, int forks, Box3* box, uint32_t* depth) {
        if (hi - lo == 1) {
            *box = p[lo].b;
            *depth = 0;
            return REF_LEAF | p[lo].id;
        }
        const uint32_t mid = split(lo, hi, budget);
        Box3 b0{}, b1{};
        uint32_t d0 = 0, d1 = 0, r0 = 0, r1 = 0;
        const uint32_t self0 = self + 1, self1 = self + (mid - lo);
        bool forked = false;
        if (forks > 0 && mid - lo >= FORK_MIN && hi - mid >= FORK_MIN) {
            std::thread left;
            try {
                left = std::thread([&] { r0 = node(lo, mid, budget - 1, self0, forks - 1, &b0, &d0); });
                forked = true;
            } catch (const std::system_error&) {   // no thread to be had: build both sides here
            }
            if (forked) {
                r1 = node(mid, hi, budget - 1, self1, forks - 1, &b1, &d1);
                left.join();
            }
        }
        if (!forked) {
            r0 = node(lo, mid, budget - 1, self0, forks, &b0, &d0);
            r1 = node(mid, hi, budget - 1, self1, forks, &b1, &d1);
        }
        GpuInner& g = inner[self];
        std::memset(&g, 0, sizeof g);
        set_box(g.c0, g.h0, b0);
        set_box(g.c1, g.h1, b1);
        for (int a = 0; a < 3; ++a) {
            box->lo[a] = std::min(b0.lo[a], b1.lo[a]);
            box->hi[a] = std::max(b0.hi[a], b1.hi[a]);
        }
        g.ref0 = r0;
        g.ref1 = r1;
        *depth = 1 + std::max(d0, d1);
        return self;
    }

    // Binned surface-area split of p[lo, hi) (reorders the range); returns the first index of the right side.
    uint32_t split(uint32_t lo, uint32_t hi, uint32_t budget) {
        const uint64_t cap = budget - 1 >= 32 ? (1ull << 32) : (1ull << (budget - 1));   // budget >= 1 here
        double cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) { cmin[a] = std::min(cmin[a], p[i].c[a]); cmax[a] = std::max(cmax[a], p[i].c[a]); }
        // All three axes binned in one pass over the range.  Most nodes hold a handful of primitives, so
        // the work per node is kept proportional to the bins actually occupied: a bin's box is set by its
        // first primitive (no initialisation pass), and the sweeps below skip empty bins -- a candidate
        // whose own bin is empty splits the set exactly like the candidate before it (same cost, and only a
        // strictly lower cost is taken), so leaving it out changes nothing.
        Box3 bb[3][NB];
        uint32_t cnt[3][NB] = {};
        double sc[3];
        for (int a = 0; a < 3; ++a) {
            const double ext = cmax[a] - cmin[a];
            sc[a] = ext > 0.0 ? NB / ext : 0.0;
        }
        for (uint32_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                if (!(sc[a] > 0.0)) continue;
                const int k = std::min(NB - 1, (int)((p[i].c[a] - cmin[a]) * sc[a]));
                if (cnt[a][k]++ == 0) bb[a][k] = p[i].b;
                else grow(bb[a][k], p[i].b);
            }
        int best_axis = -1, best_bin = 0;
        double best_cost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            if (!(sc[a] > 0.0)) continue;
            double ra[NB + 1];     // ra[k], rn[k]: half area and count of bins k.. (0 / 0 when empty)
            uint32_t rn[NB + 1];
            ra[NB] = 0.0;
            rn[NB] = 0;
            Box3 acc = empty_box();
            for (int k = NB - 1; k > 0; --k) {
                if (cnt[a][k]) {
                    grow(acc, bb[a][k]);
                    rn[k] = rn[k + 1] + cnt[a][k];
                    ra[k] = half_area(acc);
                } else {
                    rn[k] = rn[k + 1];
                    ra[k] = ra[k + 1];
                }
            }
            acc = empty_box();
            uint32_t n = 0;
            for (int k = 0; k < NB - 1; ++k) {
                if (!cnt[a][k]) continue;
                grow(acc, bb[a][k]);
                n += cnt[a][k];
                if (rn[k + 1] == 0) break;               // nothing to the right of here any more
                if (n > cap || rn[k + 1] > cap) continue;
                const double cost = half_area(acc) * n + ra[k + 1] * rn[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            // coincident centres, or no eligible SAH split: halve at the median of the widest axis
            mid = (lo + hi) / 2;
            int wa = 0;
            for (int a = 1; a < 3; ++a)
                if (cmax[a] - cmin[a] > cmax[wa] - cmin[wa]) wa = a;
            if (cmax[wa] - cmin[wa] > 0.0)
                std::nth_element(p.begin() + lo, p.begin() + mid, p.begin() + hi, [wa](const PrimBox& x, const PrimBox& y) {
                    return x.c[wa] < y.c[wa] || (x.c[wa] == y.c[wa] && x.id < y.id);
                });
        } else {
            const double s = sc[best_axis], c0 = cmin[best_axis];
            auto it = std::partition(p.begin() + lo, p.begin() + hi, [&](const PrimBox& q) {
                return std::min(NB - 1, (int)((q.c[best_axis] - c0) * s)) <= best_bin;
            });
            mid = (uint32_t)(it - p.begin());
            if (mid == lo || mid == hi) mid = (lo + hi) / 2;
        }
        return mid;
    }
};

// Threads an upload may use: the CPUs this process may run on, at most 16 (SDFGPU_BUILD_THREADS overrides).
unsigned build_threads() {
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)CPU_COUNT(&set));
    if (const char* e = std::getenv("SDFGPU_BUILD_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));
    return std::max(1u, std::min(hw, 16u));
}
int build_forks() {   // levels of the hierarchy whose two sides get a thread each
    const unsigned hw = build_threads();
    int f = 0;
    while ((2u << f) <= hw) ++f;
    return f;
}

// f(lo, hi) over [0, n) in contiguous chunks, one per thread; chunks a thread could not be started for
// run on the calling thread.  f must only write what its own indices own.
template <class F>
void parallel_for(uint32_t n, uint32_t grain, const F& f) {
    const uint32_t nt = std::min<uint32_t>(build_threads(), std::max(1u, n / std::max(1u, grain)));
    if (nt <= 1) { f(0u, n); return; }
    const uint32_t chunk = (n + nt - 1) / nt;
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    uint32_t next = chunk;   // chunk 0 is the caller's
    try {
        for (; next < n; next += chunk) {
            const uint32_t lo = next, hi = std::min(n, next + chunk);
            th.emplace_back([&f, lo, hi] { f(lo, hi); });
        }
    } catch (const std::system_error&) {
    }
    f(0u, std::min(n, chunk));
    for (; next < n; next += chunk) f(next, std::min(n, next + chunk));   // not started: done here
    for (std::thread& t : th) t.join();
}

int fail(std::string* err, int code, const char* fmt, uint32_t a = 0, uint32_t b = 0) {
    if (err) {
        char buf[256];
        std::snprintf(buf, sizeof buf, fmt, a, b);
        *err = buf;
    }
    return code;
}

// Pruning data of the general path's walks (kernels.hip walk_eval_pruned, walk_general with `only`): per resident
// node its box in the frame of the point its parent hands down, the slack the smooth unions below it can take off a
// minimum, where the walk continues when the node is skipped, and how many ancestors are combined right after it.
// Boxes are formed bottom-up in double and padded outward.  Valid only when every map on the way is affine and rigid
// (then a primitive's value outside its box is at least the distance to the box, in every frame on the way up);
// otherwise the nodes keep "no box" (bh[0] < 0) and nothing is ever skipped.
void gen_prune_data(const Ctx& c, const SdfOperationBlock* dt, size_t n, const std::vector<uint32_t>& gen_of,
                    const std::vector<uint32_t>& parent_of, bool maps_ok, CompactTree* out) {
    std::vector<GpuGenNode>& G = out->gen;
    const uint32_t ng = (uint32_t)G.size();
    std::vector<uint32_t> node_of(ng, 0);
    for (uint32_t i = 0; i < n; ++i)
        if (gen_of[i] != REF_NONE) node_of[gen_of[i]] = i;
    // skip_to: the resident index of the first live node behind the subtree
    {
        std::vector<uint32_t> next_live(n + 1, ng);   // resident index of the first live node at or after i
        for (uint32_t i = (uint32_t)n; i-- > 0;) next_live[i] = gen_of[i] != REF_NONE ? gen_of[i] : next_live[i + 1];
        for (uint32_t g = 0; g < ng; ++g) {
            const uint32_t i = node_of[g];
            G[g].skip_to = next_live[i + dt[i].len + 1u];
        }
    }
    // parent_k, point-stack slots (parents come first), first primitive / primitive count of internal nodes
    uint32_t pt_levels = 1;
    bool pt_ok = true;
    for (uint32_t g = 0; g < ng; ++g) {
        GpuGenNode& N = G[g];
        const uint32_t pg = N.parent;
        N.parent_k = 0.f;
        uint32_t in = 0;
        if (pg != REF_NONE) {
            N.parent_k = G[pg].smooth_k;
            in = (G[pg].pt_slots >> 8) & 0xFFu;
        }
        const bool pass = !N.is_leaf && N.op == GEN_OP_UNION && is_identity(N.m);
        const uint32_t out_slot = (N.is_leaf || pass) ? in : in + 1u;
        if (out_slot > 250u) pt_ok = false;
        N.pt_slots = in | ((out_slot & 0xFFu) << 8) | (pass ? 1u << 16 : 0u);
        pt_levels = std::max(pt_levels, out_slot + 1u);
    }
    out->gen_pt_levels = pt_levels;
    for (uint32_t g = ng; g-- > 0;) {
        GpuGenNode& N = G[g];
        if (N.is_leaf) continue;
        uint32_t first = SDF_NO_PRIM, cnt = 0;
        for (uint32_t k = g + 1; k < N.skip_to; k = G[k].skip_to) {   // children
            if (G[k].is_leaf) { if (first == SDF_NO_PRIM) first = G[k].prim; cnt += 1; }
            else { if (first == SDF_NO_PRIM && G[k].kind != 0) first = G[k].prim; cnt += G[k].kind; }
        }
        N.prim = first;
        N.kind = cnt;
    }
    // (read at every upload, not once: the tests compare the pruned walks with the exhaustive ones inside one process)
    const char* env = std::getenv("SDFGPU_GEN_PRUNE");
    const bool env_on = !(env && std::atoi(env) == 0);
    out->gen_prune = false;
    // (small trees: the exhaustive walk of a few hundred nodes is cheaper than bounds and cuts -- 79 nodes: 3.7 vs 6.0 ms
    // for the test frame)
    if (!maps_ok || !env_on || ng < 256 || !pt_ok) return;
    std::vector<Box3> box(ng);
    std::vector<double> slack(ng, 0.0);
    std::vector<uint8_t> has(ng, 0);
    bool ok = true;
    for (uint32_t g = ng; g-- > 0 && ok;) {
        GpuGenNode& N = G[g];
        const uint32_t i = node_of[g];
        const Aff M = aff_of(N.m);
        if (N.is_leaf) {
            ok = aff_aabb(M, dt[i], 0.0, &box[g]);
            has[g] = 1;
            continue;
        }
        // union of the children's boxes, in the frame this node hands down
        Box3 u;
        bool any = false;
        double sl = 0.0;
        for (uint32_t k = g + 1; k < N.skip_to; k = G[k].skip_to) {
            if (!has[k]) continue;
            if (!any) u = box[k];
            else for (int a = 0; a < 3; ++a) { u.lo[a] = std::min(u.lo[a], box[k].lo[a]); u.hi[a] = std::max(u.hi[a], box[k].hi[a]); }
            any = true;
            sl = std::max(sl, slack[k]);
        }
        if (!any) continue;   // nothing live below: the node evaluates to +inf and is never worth a visit (has = 0: see below)
        if (N.op == GEN_OP_CLONE) {
            // p - d * q with q between the limits: the points that fold INTO the box are the box moved by d * q
            for (int a = 0; a < 3; ++a) {
                const double d = N.clone[a], l0 = N.clone[3 + a], l1 = N.clone[6 + a];
                if (!std::isfinite(d) || !std::isfinite(l0) || !std::isfinite(l1)) { ok = false; break; }
                const double o0 = d * std::min(l0, l1), o1 = d * std::max(l0, l1);
                u.lo[a] += std::min(std::min(o0, o1), 0.0);
                u.hi[a] += std::max(std::max(o0, o1), 0.0);
            }
        }
        // into the frame of the point handed to this node: q = R^-1 (r - t), over the eight corners
        const double (*A)[3] = M.r;
        const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
        const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
        if (!(std::fabs(det) > 1e-12)) { ok = false; break; }
        const double r = 1.0 / det;
        double I[3][3];
        I[0][0] = c00 * r; I[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * r; I[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * r;
        I[1][0] = c01 * r; I[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * r; I[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * r;
        I[2][0] = c02 * r; I[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * r; I[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * r;
        Box3 w;
        for (int a = 0; a < 3; ++a) { w.lo[a] = INFINITY; w.hi[a] = -INFINITY; }
        for (int corner = 0; corner < 8; ++corner) {
            double v[3];
            for (int a = 0; a < 3; ++a) v[a] = ((corner >> a) & 1 ? u.hi[a] : u.lo[a]) - M.t[a];
            for (int a = 0; a < 3; ++a) {
                const double x = I[a][0] * v[0] + I[a][1] * v[1] + I[a][2] * v[2];
                w.lo[a] = std::min(w.lo[a], x);
                w.hi[a] = std::max(w.hi[a], x);
            }
        }
        for (int a = 0; a < 3; ++a) {
            const double pad = 1e-5 * (w.hi[a] - w.lo[a]) + 64.0 * (double)FLT_EPSILON * (std::fabs(w.lo[a]) + std::fabs(w.hi[a]) + 1.0);
            w.lo[a] -= pad;
            w.hi[a] += pad;
            if (!std::isfinite(w.lo[a]) || !std::isfinite(w.hi[a])) ok = false;
        }
        box[g] = w;
        has[g] = 1;
        slack[g] = sl + (N.smooth_k > 0.f ? 0.25 * (double)N.smooth_k : 0.0);
    }
    if (!ok) return;
    for (uint32_t g = 0; g < ng; ++g) {
        GpuGenNode& N = G[g];
        if (has[g]) {
            set_box(N.bc, N.bh, box[g]);
            N.slack = round_up(slack[g] * (1.0 + 1e-6));
        } else {
            // an internal node without a live primitive below it: +inf wherever the point is -- a box nothing is near
            N.bc[0] = N.bc[1] = N.bc[2] = 0.f;
            N.bh[0] = N.bh[1] = N.bh[2] = 0.f;
            N.slack = -INFINITY;   // (distance - slack = +inf: always skipped)
        }
    }
    // Where the smooth radii are of the order of the scene itself everything blends with everything: no cut can skip
    // anything and the bounds only cost (measured: 700 primitives, k = 4 at a spacing of ~3: 1.6x slower than the
    // exhaustive walk; k = 0.8: 1.5x faster, 3.5x at 4000 primitives).  Decided on the root: the slack that has
    // accumulated there against the scene's half extent.
    const float root_half = std::max(G[0].bh[0], std::max(G[0].bh[1], G[0].bh[2]));
    if (has[0] && G[0].slack > 0.25f * root_half) return;
    out->gen_prune = true;
    if (has[0])   // the root's box is in world coordinates: the scene box (query batches are ordered inside it)
        for (int a = 0; a < 3; ++a) {
            out->info.scene_lo[a] = G[0].bc[a] - G[0].bh[a];
            out->info.scene_hi[a] = G[0].bc[a] + G[0].bh[a];
        }
}

}  // namespace

int compact_tree(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t n,
                 uint32_t layout, uint32_t bounds, uint32_t ext, CompactTree* out, std::string* err) {
    *out = CompactTree();
    if (layout != SDFGPU_LAYOUT_GLAM_SSE2)
        return fail(err, SDFGPU_ERR_LAYOUT, "unsupported record layout %u (only glam SSE2: 624/224-byte records)", layout);
    if (bounds > SDFGPU_BOUNDS_BVH) return fail(err, SDFGPU_ERR_INVALID_ARG, "unknown bounds mode %u", bounds);
    if (ext & ~(uint32_t)(SDFGPU_EXT_SMOOTH_UNION | SDFGPU_EXT_BOX_FRAME | SDFGPU_EXT_CLONE_REPAIR))
        return fail(err, SDFGPU_ERR_INVALID_ARG, "unknown extension bits in %u", ext);
    out->info.n_nodes = n;
    if (n == 0) {  // faux_shader.rs:104,207: empty buffer -> +inf
        out->info.bounds = bounds == SDFGPU_BOUNDS_AUTO ? (uint32_t)SDFGPU_BOUNDS_BVH : bounds;
        out->info.identity_inner = 1;
        out->info.rigid_leaves = 1;
        return SDFGPU_OK;
    }
    if (!dt || !ut) return fail(err, SDFGPU_ERR_INVALID_ARG, "null buffer pointer");
    // resident records are addressed as base + (index << 6) with a 32-bit offset (kernels.hip, load_rec64)
    if (n > (1u << 26)) return fail(err, SDFGPU_ERR_TOO_LARGE, "buffer_len %u exceeds 2^26 nodes", n);

    Ctx c;
    c.dt = dt; c.n = n; c.out = out; c.err = err;
    c.prim_id.assign(n, SDF_NO_PRIM);
    c.child0.assign(n, REF_NONE);
    c.child1.assign(n, REF_NONE);

    // ---- structure: levels, len, child counts, op codes ----
    if (dt[0].level != 1) return fail(err, SDFGPU_ERR_BAD_TREE, "root level is %u, expected 1", dt[0].level);
    if (dt[0].len != n - 1) return fail(err, SDFGPU_ERR_BAD_TREE, "root len %u does not span buffer_len %u", dt[0].len, n);
    uint32_t n_prims = 0, n_inner = 0, max_level = 0, max_level_live = 0;
    bool identity_inner = true, rigid = true, exact_rows = true, inner_rigid = true;
    // A clone node (op 1) evaluates to +inf as the flat query is written: uptree op 1 returns
    // the RIGHT branch distance (faux_shader.rs:90) and a one-slot node only fills the left one
    // (:162,177).  Its whole subtree is therefore "dead": validated, numbered, never resident.
    c.dead.assign(n, 0);
    uint32_t dead_end = 0;
    uint32_t ext_in_use = 0;     // extensions that actually occur in the live part of this tree
    float margin_T = 0.f;
    double lip = 1.0;   // largest leaf Lipschitz constant (1 for rigid leaves)
    // What the structural pass needs of a node, gathered by all threads into 24 bytes per node first: the pass
    // itself is sequential (a pre-order walk with running state), and a node's block is 624 bytes of which
    // it reads three scattered cache lines -- walking the blocks themselves was 10 ms of a 25 ms upload of
    // 100k primitives.  The pass makes the same checks in the same order as before, so the first error
    // reported is the same one.
    struct Head {
        uint32_t level, len, op_code;
        uint8_t is_primitive, parent_is_union;
        uint8_t flags;       // HEAD_* below
        float param_max;     // largest |size parameter| (radius; half extents; frame dimensions)
    };
    enum : uint8_t { HEAD_TI_FINITE = 1, HEAD_TI_IDENTITY = 2, HEAD_P0_BAD = 4, HEAD_VEC_BAD = 8, HEAD_INNER_RIGID = 16 };
    std::vector<Head> head(n);
    parallel_for(n, 8192, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            const SdfOperationBlock& b = dt[i];
            Head h;
            h.level = b.level; h.len = b.len; h.op_code = b.op_code;
            h.is_primitive = b.is_primitive; h.parent_is_union = b.parent_is_union;
            const float* ti = b.bounding_box.trans_inverse.m;
            h.flags = finite16(ti) ? HEAD_TI_FINITE : 0;
            if (!b.is_primitive && is_identity(ti)) h.flags |= HEAD_TI_IDENTITY;
            if (!b.is_primitive && bottom_row_exact(ti) && rigid3(ti, RIGID_TOL)) h.flags |= HEAD_INNER_RIGID;   // chain path
            h.param_max = 0.f;
            if (b.is_primitive) {
                const float p0 = b.op_specific.floats[0];   // sphere: radius; box frame: thickness
                if (!std::isfinite(p0)) h.flags |= HEAD_P0_BAD;
                float vmax = 0.f;
                for (int a = 0; a < 3; ++a) {               // box: half extents; box frame: dimensions
                    const float v = b.op_specific.vec4s[0].v[a];
                    if (!std::isfinite(v)) h.flags |= HEAD_VEC_BAD;
                    vmax = std::max(vmax, std::fabs(v));
                }
                h.param_max = b.op_code == SDF_PRIM_SPHERE ? std::fabs(p0) : vmax;
            }
            head[i] = h;
        }
    });
    for (uint32_t i = 0; i < n; ++i) {
        if (head[i].level == 0) return fail(err, SDFGPU_ERR_BAD_TREE, "node %u has level 0", i);
        if (head[i].level >= 255)
            return fail(err, SDFGPU_ERR_TOO_DEEP, "node %u at level %u (the reference stack holds 256 frames)", i, head[i].level);
    }
    for (uint32_t i = 0; i < n; ++i) {
        const Head& b = head[i];
        if (b.is_primitive > 1 || b.parent_is_union > 1) return fail(err, SDFGPU_ERR_BAD_TREE, "node %u: bool field is not 0/1", i);
        if ((uint64_t)i + b.len >= n) return fail(err, SDFGPU_ERR_BAD_TREE, "node %u: len %u runs past the buffer", i, b.len);
        max_level = std::max(max_level, b.level);
        if (!b.is_primitive && b.op_code == SDF_OP_CAA_CLONE) {
            if (ext & SDFGPU_EXT_CLONE_REPAIR) ext_in_use |= (i < dead_end) ? 0u : (uint32_t)SDFGPU_EXT_CLONE_REPAIR;
            else dead_end = std::max(dead_end, i + b.len + 1u);
        }
        const bool dead = i < dead_end;
        c.dead[i] = dead ? 1 : 0;
        if (!dead) max_level_live = std::max(max_level_live, b.level);
        if (!(b.flags & HEAD_TI_FINITE)) return fail(err, SDFGPU_ERR_NON_FINITE, "node %u: non-finite trans_inverse", i);
        if (b.is_primitive) {
            if (b.len != 0) return fail(err, SDFGPU_ERR_BAD_TREE, "primitive %u has len %u", i, b.len);
            if (b.op_code == SDF_PRIM_SPHERE) {
                if (b.flags & HEAD_P0_BAD) return fail(err, SDFGPU_ERR_NON_FINITE, "primitive %u: non-finite radius", i);
                margin_T = std::max(margin_T, b.param_max);
            } else if (b.op_code == SDF_PRIM_BOX) {
                if (b.flags & HEAD_VEC_BAD) return fail(err, SDFGPU_ERR_NON_FINITE, "primitive %u: non-finite half extent", i);
                margin_T = std::max(margin_T, b.param_max);
            } else if (b.op_code == SDF_PRIM_BOX_FRAME && (ext & SDFGPU_EXT_BOX_FRAME)) {
                if (b.flags & HEAD_VEC_BAD) return fail(err, SDFGPU_ERR_NON_FINITE, "primitive %u: non-finite dimension", i);
                margin_T = std::max(margin_T, b.param_max);
                if (b.flags & HEAD_P0_BAD) return fail(err, SDFGPU_ERR_NON_FINITE, "primitive %u: non-finite thickness", i);
                if (!dead) ext_in_use |= SDFGPU_EXT_BOX_FRAME;
            } else {
                return fail(err, SDFGPU_ERR_UNSUPPORTED_OP, "Unsupported primitive op code: %u (node %u)", b.op_code, i);
            }
            c.prim_id[i] = n_prims++;
        } else {
            if (b.op_code != SDF_OP_UNION && b.op_code != SDF_OP_CAA_CLONE)
                return fail(err, SDFGPU_ERR_UNSUPPORTED_OP, "Unsupported downtree op code: %u (node %u)", b.op_code, i);
            if (b.len == 0) return fail(err, SDFGPU_ERR_BAD_TREE, "operation node %u has no children", i);
            if (!dead && !(b.flags & HEAD_TI_IDENTITY)) identity_inner = false;
            if (!dead && !(b.flags & HEAD_INNER_RIGID)) inner_rigid = false;
            if (!dead) ++n_inner;
            uint32_t end = i + b.len, k = i + 1, cnt = 0;
            while (k <= end) {
                if (head[k].level != b.level + 1)
                    return fail(err, SDFGPU_ERR_BAD_TREE, "node %u: child %u is not one level below", i, k);
                if ((uint64_t)k + head[k].len > end)
                    return fail(err, SDFGPU_ERR_BAD_TREE, "node %u: child %u overruns its parent", i, k);
                if (cnt == 0) c.child0[i] = k;
                else if (cnt == 1) c.child1[i] = k;
                else return fail(err, SDFGPU_ERR_BAD_TREE, "node %u has more than two children (frames hold two, faux_shader.rs:14-15)", i);
                ++cnt;
                k += 1 + head[k].len;
            }
            if (b.op_code == SDF_OP_CAA_CLONE && cnt != 1)
                return fail(err, SDFGPU_ERR_UNSUPPORTED_OP,
                            "clone node %u has %u slots; the crate builds exactly one (elements.rs:222)", i, cnt);
        }
    }
    {
        // live leaves: affine rows, rigidity, Lipschitz constant (and / max reductions: order-independent)
        const uint32_t slots = build_threads() + 1;
        std::vector<uint8_t> not_exact(slots, 0), not_rigid(slots, 0);
        std::vector<double> lips(slots, 1.0);
        std::atomic<uint32_t> slot{0};
        parallel_for(n, 8192, [&](uint32_t lo, uint32_t hi) {
            uint8_t ne = 0, nr = 0;
            double l = 1.0;
            for (uint32_t i = lo; i < hi; ++i) {
                if (!dt[i].is_primitive || c.dead[i]) continue;
                const float* ti = dt[i].bounding_box.trans_inverse.m;
                if (!bottom_row_exact(ti)) ne = 1;
                if (!rigid3(ti, RIGID_TOL)) nr = 1;
                l = std::max(l, lipschitz3(ti));
            }
            const uint32_t k = slot.fetch_add(1) % slots;
            not_exact[k] = ne; not_rigid[k] = nr; lips[k] = l;
        });
        for (uint32_t k = 0; k < slots; ++k) {
            if (not_exact[k]) exact_rows = false;
            if (not_rigid[k]) rigid = false;
            lip = std::max(lip, lips[k]);
        }
    }
    // live primitives below every node (children follow their parent in pre-order)
    c.live.assign(n, 0);
    for (uint32_t i = n; i-- > 0;) {
        if (head[i].is_primitive) c.live[i] = c.dead[i] ? 0u : 1u;
        else c.live[i] = (c.child0[i] != REF_NONE ? c.live[c.child0[i]] : 0u) + (c.child1[i] != REF_NONE ? c.live[c.child1[i]] : 0u);
        if (c.dead[i]) c.live[i] = 0;
    }
    // ---- the uptree buffer must be the post-order of the same tree ----
    std::vector<float> node_smooth;
    {
        std::vector<uint32_t> stack, post;
        post.reserve(n);
        // iterative post-order over (child0, child1)
        std::vector<std::pair<uint32_t, uint8_t>> st;
        st.emplace_back(0u, (uint8_t)0);
        while (!st.empty()) {
            auto& top = st.back();
            uint32_t i = top.first;
            if (top.second == 0) {
                top.second = 1;
                if (c.child0[i] != REF_NONE) { st.emplace_back(c.child0[i], (uint8_t)0); continue; }
            }
            if (st.back().second == 1) {
                st.back().second = 2;
                if (c.child1[i] != REF_NONE) { st.emplace_back(c.child1[i], (uint8_t)0); continue; }
            }
            post.push_back(i);
            st.pop_back();
        }
        if (post.size() != n) return fail(err, SDFGPU_ERR_BAD_TREE, "tree does not cover the buffer");
        node_smooth.assign(n, 0.f);
        for (uint32_t j = 0; j < n; ++j) {
            const Head& b = head[post[j]];
            if (ut[j].level != b.level || ut[j].op_code != b.op_code)
                return fail(err, SDFGPU_ERR_BAD_TREE, "uptree block %u does not match downtree node %u", j, post[j]);
            // union up-blocks carry the smooth radius (elements.rs:207-211); in effect only under the extension
            if (!b.is_primitive && b.op_code == SDF_OP_UNION && (ext & SDFGPU_EXT_SMOOTH_UNION) && !c.dead[post[j]]) {
                const float k = ut[j].op_specific.floats[0];
                if (!std::isfinite(k)) return fail(err, SDFGPU_ERR_NON_FINITE, "union node %u: non-finite smooth radius", post[j]);
                if (k > 0.f) { node_smooth[post[j]] = k; ext_in_use |= SDFGPU_EXT_SMOOTH_UNION; }
            }
        }
    }

    out->info.extensions = ext_in_use;
    std::vector<uint32_t> parent_of(n, REF_NONE);
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t k : {c.child0[i], c.child1[i]})
            if (k != REF_NONE) parent_of[k] = i;
    // Chain path (round 3): internal nodes that re-base the point, but every map affine and rigid and no extension
    // in effect: the library's own hierarchy over world boxes of the composed maps prunes, leaf values keep the
    // reference's one rounded mat-vec per level.  SDFGPU_CHAIN=0 sends such buffers to the exhaustive path again.
    static const bool chain_env = [] { const char* e = std::getenv("SDFGPU_CHAIN"); return !(e && std::atoi(e) == 0); }();
    bool chain = !identity_inner && exact_rows && inner_rigid && rigid && !ext_in_use && chain_env &&
                 (bounds == SDFGPU_BOUNDS_AUTO || bounds == SDFGPU_BOUNDS_BVH);
    std::vector<uint32_t> chain_len;   // per node: non-identity internal maps from the root down to and including it
    if (chain) {
        chain_len.assign(n, 0);
        uint64_t total = 0;
        for (uint32_t i = 0; i < n && chain; ++i) {
            if (c.dead[i] || head[i].is_primitive) continue;
            const uint32_t up = parent_of[i] == REF_NONE ? 0u : chain_len[parent_of[i]];
            chain_len[i] = up + ((head[i].flags & HEAD_TI_IDENTITY) ? 0u : 1u);
            if (chain_len[i] > CHAIN_MAX_LEVELS) chain = false;
            bool leaf_child = false;
            for (uint32_t k : {c.child0[i], c.child1[i]})
                if (k != REF_NONE && head[k].is_primitive && !c.dead[k]) leaf_child = true;
            if (leaf_child) total += chain_len[i];
        }
        if (total > (1ull << 24)) chain = false;   // 1 GiB of chain records: such a buffer stays on the exhaustive path
    }
    // chain path: world -> "frame handed to the children" of every live internal node, composed in double, and the
    // sum of the translations met on the way (the magnitude the intermediate points of the chain can reach)
    std::vector<Aff> in_aff;
    std::vector<double> in_tsum;
    if (chain) {
        in_aff.resize(n);
        in_tsum.assign(n, 0.0);
        for (uint32_t i = 0; i < n; ++i) {
            if (c.dead[i] || head[i].is_primitive) continue;
            const uint32_t par = parent_of[i];
            const Aff up = par == REF_NONE ? aff_identity() : in_aff[par];
            const float* m = dt[i].bounding_box.trans_inverse.m;
            in_aff[i] = (head[i].flags & HEAD_TI_IDENTITY) ? up : aff_then(up, aff_of(m));
            in_tsum[i] = (par == REF_NONE ? 0.0 : in_tsum[par]) + std::fabs((double)m[12]) + std::fabs((double)m[13]) + std::fabs((double)m[14]);
        }
        // the pruned walk compares world distances with leaf values: the COMPOSED map of every leaf has to be an
        // isometry to the same tolerance a single leaf map is held to (tolerances of the levels add up)
        for (uint32_t i = 0; i < n && chain; ++i) {
            if (!head[i].is_primitive || c.dead[i] || parent_of[i] == REF_NONE) continue;
            const Aff M = aff_then(in_aff[parent_of[i]], aff_of(dt[i].bounding_box.trans_inverse.m));
            for (int x = 0; x < 3 && chain; ++x)
                for (int y = 0; y < 3; ++y) {
                    double sum = 0;
                    for (int r = 0; r < 3; ++r) sum += M.r[r][x] * M.r[r][y];
                    if (!(std::fabs(sum - (x == y ? 1.0 : 0.0)) <= RIGID_TOL)) { chain = false; break; }
                }
        }
    }
    if ((!identity_inner || !exact_rows || ext_in_use) && !chain) {
        // General path: no pruning, the literal per-level point chain.
        if (bounds == SDFGPU_BOUNDS_AABB || bounds == SDFGPU_BOUNDS_OBB || bounds == SDFGPU_BOUNDS_BVH)
            return fail(err, SDFGPU_ERR_BAD_TREE,
                        "pruned bounds need identity internal trans_inverse, affine leaf rows and no extension in effect; use AUTO or NONE");
        if (max_level_live > GEN_MAX_LEVELS)
            return fail(err, SDFGPU_ERR_TOO_DEEP, "general path holds %u levels, tree has %u", GEN_MAX_LEVELS, max_level_live);
        if ((ext_in_use & SDFGPU_EXT_SMOOTH_UNION) && max_level_live > GEN_EVAL_MAX_LEVELS)
            return fail(err, SDFGPU_ERR_TOO_DEEP, "smooth-union evaluation holds %u levels, tree has %u", GEN_EVAL_MAX_LEVELS,
                        max_level_live);
        out->gen.reserve(n);
        std::vector<uint32_t> gen_of(n, REF_NONE);   // node -> index in the resident array
        for (uint32_t i = 0; i < n; ++i) {
            if (c.dead[i]) continue;   // clone subtrees contribute +inf as written: left out of the walk
            gen_of[i] = (uint32_t)out->gen.size();
            out->gen.emplace_back();
            GpuGenNode& g = out->gen.back();
            std::memset(&g, 0, sizeof g);
            std::memcpy(g.m, dt[i].bounding_box.trans_inverse.m, 64);
            g.is_leaf = dt[i].is_primitive;
            g.level = dt[i].level;
            g.kind = dt[i].op_code;
            g.prim = c.prim_id[i];
            g.bh[0] = -1.f;   // no box (yet)
            const uint32_t par = parent_of[i];
            g.parent = par == REF_NONE ? REF_NONE : gen_of[par];
            g.slot = (par != REF_NONE && c.child1[par] == i) ? 1u : 0u;
            if (dt[i].is_primitive) {
                if (dt[i].op_code == SDF_PRIM_SPHERE) g.param[0] = dt[i].op_specific.floats[0];
                else for (int a = 0; a < 3; ++a) g.param[a] = dt[i].op_specific.vec4s[0].v[a];
                if (dt[i].op_code == SDF_PRIM_BOX_FRAME) g.param[3] = dt[i].op_specific.floats[0];
            } else {
                g.op = dt[i].op_code == SDF_OP_CAA_CLONE ? GEN_OP_CLONE : GEN_OP_UNION;
                g.smooth_k = node_smooth[i];
                if (g.op == GEN_OP_CLONE)
                    for (int v = 0; v < 3; ++v)
                        for (int a = 0; a < 3; ++a) g.clone[3 * v + a] = dt[i].op_specific.vec4s[v].v[a];
            }
        }
        // n_up: every live internal node is combined right after the last live node of its subtree
        for (uint32_t i = 0; i < n; ++i) {
            if (c.dead[i] || dt[i].is_primitive) continue;
            uint32_t last = i;
            for (uint32_t k = i + 1; k <= i + dt[i].len; ++k)
                if (!c.dead[k]) last = k;
            out->gen[gen_of[last]].n_up += 1;
        }
        out->gen_eval = (ext_in_use & SDFGPU_EXT_SMOOTH_UNION) != 0;
        out->margin_T = margin_T;
        gen_prune_data(c, dt, head.size(), gen_of, parent_of, exact_rows && rigid && inner_rigid, out);
        sdfgpu_tree_info& gi = out->info;
        gi.n_prims = n_prims;
        gi.n_inner = n_inner;
        gi.max_level = max_level;
        gi.bounds = SDFGPU_BOUNDS_NONE;
        gi.walk = out->gen_prune ? SDFGPU_WALK_GENERAL_PRUNED : SDFGPU_WALK_GENERAL_EXHAUSTIVE;
        gi.identity_inner = identity_inner ? 1 : 0;
        gi.rigid_leaves = rigid ? 1 : 0;
        gi.device_bytes = out->gen.size() * sizeof(GpuGenNode);
        out->stack_depth = max_level_live + 1;
        return SDFGPU_OK;
    }

    uint32_t mode = bounds;
    if (mode == SDFGPU_BOUNDS_AUTO) mode = rigid ? SDFGPU_BOUNDS_BVH : SDFGPU_BOUNDS_NONE;
    // The per-wavefront stack holds 64 pending subtrees.  Only the modes that walk the BUFFER's topology
    // depend on its depth; BVH mode rebuilds the hierarchy under its own depth budget, so any buffer the
    // reference's 256-frame stack accepts (levels < 255, checked above) is accepted there.
    if (mode != SDFGPU_BOUNDS_BVH && max_level_live > 64)
        return fail(err, SDFGPU_ERR_TOO_DEEP, "tree depth %u exceeds the 64-entry wave stack (use AUTO / BVH bounds)", max_level_live);
    if ((mode == SDFGPU_BOUNDS_AABB || mode == SDFGPU_BOUNDS_BVH || mode == SDFGPU_BOUNDS_OBB) && !rigid)
        return fail(err, SDFGPU_ERR_BAD_TREE,
                    "pruned bounds (AABB, BVH, OBB) need rigid leaf transforms (trans_inverse is not orthonormal); use AUTO or NONE");
    if (mode == SDFGPU_BOUNDS_OBB) {
        for (uint32_t i = 0; i < n; ++i) {
            const SdfBoundingBoxBlock& bb = dt[i].bounding_box;
            if (!finite16(bb.full_inverse.m) || !std::isfinite(bb.scale.v[0]) || !std::isfinite(bb.scale.v[1]) ||
                !std::isfinite(bb.scale.v[2]))
                return fail(err, SDFGPU_ERR_NON_FINITE, "node %u: non-finite bounding box", i);
            if (!bottom_row_exact(bb.full_inverse.m) || bb.scale.v[3] != 0.f)
                return fail(err, SDFGPU_ERR_BAD_TREE, "node %u: OBB bounds need an affine full_inverse and scale.w == 0", i);
        }
    }
    c.bounds = mode;

    // ---- leaves, in primitive-id order ----
    out->leaves.resize(n_prims);
    {
        // per-chunk maxima of |translation| (max is order-independent, so the result does not depend on the chunking)
        std::vector<float> chunk_T(build_threads() + 1, 0.f);
        std::atomic<uint32_t> slot{0};
        parallel_for(n, 8192, [&](uint32_t lo, uint32_t hi) {
            float mt = 0.f;
            for (uint32_t i = lo; i < hi; ++i) {
                if (!dt[i].is_primitive) continue;
                GpuLeaf& L = out->leaves[c.prim_id[i]];
                std::memset(&L, 0, sizeof L);
                const float* m = dt[i].bounding_box.trans_inverse.m;
                for (int r = 0; r < 3; ++r)
                    for (int col = 0; col < 4; ++col) L.m[4 * r + col] = m[4 * col + r];
                mt = std::max(mt, std::fabs(m[12]) + std::fabs(m[13]) + std::fabs(m[14]));
                if (chain && parent_of[i] != REF_NONE) mt = std::max(mt, (float)in_tsum[parent_of[i]] + std::fabs(m[12]) + std::fabs(m[13]) + std::fabs(m[14]));
                L.kind = dt[i].op_code;
                if (L.kind == SDF_PRIM_SPHERE) {
                    L.param[0] = dt[i].op_specific.floats[0];
                } else {
                    for (int a = 0; a < 3; ++a) L.param[a] = dt[i].op_specific.vec4s[0].v[a];
                }
            }
            chunk_T[slot.fetch_add(1) % chunk_T.size()] = mt;
        });
        for (float v : chunk_T) margin_T = std::max(margin_T, v);
    }
    double extent_sum = 0.0;   // summed in primitive order: the same value whatever the thread count
    for (const GpuLeaf& L : out->leaves)
        extent_sum += std::max({std::fabs((double)L.param[0]), std::fabs((double)L.param[1]), std::fabs((double)L.param[2])});
    out->mean_extent = n_prims ? (float)(extent_sum / n_prims) : 0.f;
    if (mode == SDFGPU_BOUNDS_OBB) out->inner_obb.reserve(n_inner);
    else out->inner.reserve(n_inner);
    Box3 root{};
    uint32_t depth = 0;
    if (mode == SDFGPU_BOUNDS_BVH) {
        std::vector<uint32_t> live_nodes;   // buffer index of every live primitive, in pre-order
        live_nodes.reserve(n_prims);
        for (uint32_t i = 0; i < n; ++i)
            if (dt[i].is_primitive && !c.dead[i]) live_nodes.push_back(i);
        std::vector<PrimBox> prims(live_nodes.size());
        std::atomic<uint32_t> singular{UINT32_MAX};   // lowest buffer index whose transform cannot be inverted
        parallel_for((uint32_t)live_nodes.size(), 4096, [&](uint32_t lo, uint32_t hi) {
            for (uint32_t j = lo; j < hi; ++j) {
                const uint32_t i = live_nodes[j];
                PrimBox& q = prims[j];
                q.id = c.prim_id[i];
                bool ok;
                if (chain && parent_of[i] != REF_NONE) {
                    const uint32_t par = parent_of[i];
                    const double pad = 8.0 * (double)FLT_EPSILON * (double)(chain_len[par] + 1u) * (in_tsum[par] + 1.0);
                    ok = aff_aabb(aff_then(in_aff[par], aff_of(dt[i].bounding_box.trans_inverse.m)), dt[i], pad, &q.b);
                } else {
                    ok = leaf_aabb(dt[i], &q.b);
                }
                if (!ok) {
                    uint32_t seen = singular.load();
                    while (i < seen && !singular.compare_exchange_weak(seen, i)) {
                    }
                    continue;
                }
                for (int a = 0; a < 3; ++a) q.c[a] = 0.5 * (q.b.lo[a] + q.b.hi[a]);
            }
        });
        if (singular.load() != UINT32_MAX) return fail(err, SDFGPU_ERR_BAD_TREE, "primitive %u: singular transform", singular.load());
        // depth budget: balanced depth + slack, never beyond the 64-entry wave stack
        const uint32_t n_live = (uint32_t)prims.size();
        uint32_t lg = 0;
        while ((1ull << lg) < n_live) ++lg;
        uint32_t slack = 8;
        if (const char* e = std::getenv("SDFGPU_BVH_SLACK")) slack = (uint32_t)std::max(0, std::atoi(e));
        const uint32_t budget = std::min(64u, std::max(1u, lg + slack));
        out->inner.resize(n_live ? n_live - 1 : 0);
        SahBuild sah{prims, out->inner.data()};
        out->root_ref = n_live ? sah.node(0, n_live, budget, 0, build_forks(), &root, &depth) : REF_NONE;
        if (depth > budget) return fail(err, SDFGPU_ERR_TOO_DEEP, "built hierarchy is %u levels deep (budget %u)", depth, budget);
    } else {
        out->root_ref = emit(c, 0, &root, &depth);
    }
    if (chain) {
        // the composed maps decide how fast a leaf value can change, and the chain records themselves
        for (uint32_t i = 0; i < n; ++i) {
            if (!dt[i].is_primitive || c.dead[i] || parent_of[i] == REF_NONE) continue;
            lip = std::max(lip, aff_lipschitz(aff_then(in_aff[parent_of[i]], aff_of(dt[i].bounding_box.trans_inverse.m))));
        }
        out->leaf_chain.assign((size_t)n_prims * 2u, 0u);
        std::vector<uint32_t> path;
        for (uint32_t a = 0; a < n; ++a) {
            if (c.dead[a] || head[a].is_primitive) continue;
            bool leaf_child = false;
            for (uint32_t k : {c.child0[a], c.child1[a]})
                if (k != REF_NONE && head[k].is_primitive && !c.dead[k]) leaf_child = true;
            if (!leaf_child) continue;
            path.clear();
            for (uint32_t k = a; k != REF_NONE; k = parent_of[k])
                if (!(head[k].flags & HEAD_TI_IDENTITY)) path.push_back(k);
            const uint32_t off = (uint32_t)out->chain_mats.size();
            for (size_t j = path.size(); j-- > 0;) {   // root first
                GpuChainMat g;
                std::memset(&g, 0, sizeof g);
                const float* m = dt[path[j]].bounding_box.trans_inverse.m;
                for (int r = 0; r < 3; ++r)
                    for (int col = 0; col < 4; ++col) g.m[4 * r + col] = m[4 * col + r];
                out->chain_mats.push_back(g);
            }
            out->chain_max = std::max(out->chain_max, (uint32_t)path.size());
            for (uint32_t k : {c.child0[a], c.child1[a]})
                if (k != REF_NONE && head[k].is_primitive && !c.dead[k]) {
                    out->leaf_chain[2u * c.prim_id[k]] = off;
                    out->leaf_chain[2u * c.prim_id[k] + 1u] = (uint32_t)path.size();
                }
        }
    }
    out->stack_depth = std::max(1u, depth);
    out->margin_T = margin_T;
    if (mode != SDFGPU_BOUNDS_OBB && !out->inner.empty()) {
        // Box records in scaled coordinates (kernels.hip, aabb_dist2): a power of two (exact) that puts
        // every per-axis distance within 16 scene extents below 1, the range of the hardware result clamp.
        double ext = 1.0;
        if (mode != SDFGPU_BOUNDS_NONE)
            for (int a = 0; a < 3; ++a) ext = std::max({ext, std::fabs(root.lo[a]), std::fabs(root.hi[a])});
        int e = 0;
        std::frexp(16.0 * ext, &e);                  // 16 ext < 2^e
        e = std::min(e, 100);
        const float sc = std::ldexp(1.0f, -e);
        out->box_scale = sc;
        for (GpuInner& g : out->inner)
            for (int a = 0; a < 3; ++a) { g.c0[a] *= sc; g.h0[a] *= sc; g.c1[a] *= sc; g.h1[a] *= sc; }
    }
    // budget decrement per unit of march distance: the Lipschitz constant, plus the rounding of the product
    out->lipschitz = round_up(lip * 1.000004);

    sdfgpu_tree_info& info = out->info;
    info.n_prims = n_prims;
    info.n_inner = (uint32_t)(mode == SDFGPU_BOUNDS_OBB ? out->inner_obb.size() : out->inner.size());
    info.max_level = max_level;
    info.stack_depth = out->stack_depth;
    info.bounds = mode;
    info.walk = chain ? SDFGPU_WALK_CHAIN : SDFGPU_WALK_HIERARCHY;
    info.identity_inner = identity_inner ? 1 : 0;
    info.rigid_leaves = rigid ? 1 : 0;
    info.device_bytes = out->leaves.size() * sizeof(GpuLeaf) + out->inner.size() * sizeof(GpuInner) +
                        out->inner_obb.size() * sizeof(GpuInnerObb) + out->chain_mats.size() * sizeof(GpuChainMat) +
                        out->leaf_chain.size() * sizeof(uint32_t);
    // scene box (query reordering uses it in every mode; exact only for rigid leaves)
    {
        Box3 sb{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
        if (mode == SDFGPU_BOUNDS_BVH && out->root_ref != REF_NONE) {
            sb = root;   // the hierarchy's root box IS the union of the live primitives' boxes
        } else {
            for (uint32_t i = 0; i < n; ++i) {
                if (!dt[i].is_primitive || c.dead[i]) continue;
                Box3 lb{};
                if (!leaf_aabb(dt[i], &lb)) continue;
                for (int a = 0; a < 3; ++a) { sb.lo[a] = std::min(sb.lo[a], lb.lo[a]); sb.hi[a] = std::max(sb.hi[a], lb.hi[a]); }
            }
        }
        for (int a = 0; a < 3; ++a) {
            info.scene_lo[a] = std::isfinite(sb.lo[a]) ? (float)sb.lo[a] : 0.f;
            info.scene_hi[a] = std::isfinite(sb.hi[a]) ? (float)sb.hi[a] : 0.f;
        }
    }
    return SDFGPU_OK;
}

}  // namespace sdfgpu
