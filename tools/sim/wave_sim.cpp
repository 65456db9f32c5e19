// wave_sim.cpp -- design-exploration tool (not product code, not test infrastructure): a CPU model
// of render_pass_kernel's wave-uniform march over the SAME resident hierarchy (compact_tree), used
// to count wave-level node visits under different lane-scheduling strategies before building them.
//   build: see tools/sim/Makefile      run: ./wave_sim [strategy] [threshold] [w h] [n_prims seed]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>
#include <map>

#include "../../include/sdfgpu.h"
#include "../../include/sdfhost.h"
#include "../../sdf_amd/csrc/gpu_tree.hpp"

using namespace sdfgpu;

static CompactTree T;
static bool g_k2 = false;
static float g_ext = 0.f;
static float g_lip;

static inline float leaf_eval(const GpuLeaf& L, float px, float py, float pz) {
    const float* m = L.m;
    float lx = ((m[0] * px + m[1] * py) + m[2] * pz) + m[3];
    float ly = ((m[4] * px + m[5] * py) + m[6] * pz) + m[7];
    float lz = ((m[8] * px + m[9] * py) + m[10] * pz) + m[11];
    if (L.kind == SDF_PRIM_SPHERE) return std::sqrt((lx * lx + ly * ly) + lz * lz) - L.param[0];
    float qx = std::fabs(lx) - L.param[0], qy = std::fabs(ly) - L.param[1], qz = std::fabs(lz) - L.param[2];
    float mx = qx > 0 ? qx : 0, my = qy > 0 ? qy : 0, mz = qz > 0 ? qz : 0;
    float l = std::sqrt((mx * mx + my * my) + mz * mz);
    return l + std::fmin(std::fmax(qx, std::fmax(qy, qz)), 0.0f);
}
static inline float aabb_d2(const float* c, const float* h, float px, float py, float pz) {
    float dx = std::fmax(std::fabs(px - c[0]) - h[0], 0.f), dy = std::fmax(std::fabs(py - c[1]) - h[1], 0.f),
          dz = std::fmax(std::fabs(pz - c[2]) - h[2], 0.f);
    return std::fma(dz, dz, std::fma(dy, dy, dx * dx));
}
static inline bool is_better(float d, uint32_t id, float best, uint32_t bid) {
    return (d < best) | ((d == best) & ((id > bid) | (bid == SDF_NO_PRIM)));
}

struct Lane {
    bool alive = false, hit = false;
    float t = 0, budget = 0, dx = 0, dy = 0, dz = 0;
    uint32_t last = SDF_NO_PRIM, last2 = SDF_NO_PRIM, sid = SDF_NO_PRIM, steps = 0;
    int pix = -1;
    // per-step scratch
    float px, py, pz, mg, best, second, third, lbk, thr;
    uint32_t bid;
    bool drive, certified, paused = false;
};

struct Stats {
    uint64_t wave_steps = 0, wave_walks = 0, wave_inner = 0, wave_leaf = 0, lane_steps = 0, lane_walks = 0, lane_needed = 0,
             lane_inner = 0, chain_inner = 0, box_tests = 0, pushes = 0;
    uint64_t by_alive_steps[65] = {0}, by_alive_inner[65] = {0}, by_drive_walks[65] = {0}, by_drive_inner[65] = {0};
    void add(const Stats& o) {
        wave_steps += o.wave_steps; wave_walks += o.wave_walks; wave_inner += o.wave_inner; wave_leaf += o.wave_leaf;
        lane_steps += o.lane_steps; lane_walks += o.lane_walks; lane_needed += o.lane_needed; lane_inner += o.lane_inner; chain_inner += o.chain_inner;
        box_tests += o.box_tests; pushes += o.pushes;
        for (int i = 0; i < 65; ++i) {
            by_alive_steps[i] += o.by_alive_steps[i]; by_alive_inner[i] += o.by_alive_inner[i];
            by_drive_walks[i] += o.by_drive_walks[i]; by_drive_inner[i] += o.by_drive_inner[i];
        }
    }
    double valu() const { return 30.0 * wave_inner + 50.0 * wave_leaf + 120.0 * wave_steps; }
};

static sdfgpu_camera cam;
static sdfgpu_march_params mp;
static float ox, oy, oz;
static uint32_t W, H;

static void look_at(const float eye[3], const float target[3], const float up[3], float fov_y, uint32_t width, uint32_t height,
                    sdfgpu_camera* out) {   // same arithmetic as sdfgpu_camera_look_at (api.cpp needs the HIP runtime)
    double f[3], r[3], u[3], fl = 0, rl = 0;
    for (int i = 0; i < 3; ++i) { f[i] = (double)target[i] - (double)eye[i]; fl += f[i] * f[i]; }
    fl = std::sqrt(fl);
    for (int i = 0; i < 3; ++i) f[i] /= fl;
    r[0] = f[1] * up[2] - f[2] * up[1]; r[1] = f[2] * up[0] - f[0] * up[2]; r[2] = f[0] * up[1] - f[1] * up[0];
    for (int i = 0; i < 3; ++i) rl += r[i] * r[i];
    rl = std::sqrt(rl);
    for (int i = 0; i < 3; ++i) r[i] /= rl;
    u[0] = r[1] * f[2] - r[2] * f[1]; u[1] = r[2] * f[0] - r[0] * f[2]; u[2] = r[0] * f[1] - r[1] * f[0];
    const double half_h = std::tan(0.5 * (double)fov_y), half_w = half_h * (double)width / (double)height;
    const double sx = 2.0 * half_w / (double)width, sy = 2.0 * half_h / (double)height;
    for (int i = 0; i < 3; ++i) {
        out->eye[i] = eye[i];
        out->du[i] = (float)(r[i] * sx);
        out->dv[i] = (float)(-u[i] * sy);
        out->origin[i] = (float)((double)eye[i] + f[i] - r[i] * (half_w - 0.5 * sx) + u[i] * (half_h - 0.5 * sy));
    }
}

static void init_ray(Lane& l, int x, int y) {
    l = Lane();
    l.pix = y * (int)W + x;
    l.alive = x < (int)W && y < (int)H;
    float fx = (float)x, fy = (float)y;
    float dx = ((cam.origin[0] + cam.du[0] * fx) + cam.dv[0] * fy) - ox;
    float dy = ((cam.origin[1] + cam.du[1] * fx) + cam.dv[1] * fy) - oy;
    float dz = ((cam.origin[2] + cam.du[2] * fx) + cam.dv[2] * fy) - oz;
    float dl = std::sqrt((dx * dx + dy * dy) + dz * dz);
    l.dx = dx / dl; l.dy = dy / dl; l.dz = dz / dl;
    l.t = mp.t_min;
}

// one wave-uniform walk for the lanes with drive set; returns wave-level inner visits
static void walk(Lane* L, Stats& st) {
    uint32_t cur = T.root_ref;
    if (cur == REF_NONE) return;
    struct E { uint32_t ref; float k[64]; };
    static thread_local std::vector<E> stk(128);
    uint32_t sp = 0;
    int ndrive = 0;
    for (int i = 0; i < 64; ++i) {
        ndrive += L[i].drive;
        float bp = (std::fmax(L[i].best + L[i].mg, 0.f) + g_ext) * T.box_scale;
        L[i].thr = bp * bp * 1.0006f;
    }
    uint64_t inner0 = st.wave_inner;
    bool chain = true;
    for (;;) {
        while (!(cur & REF_LEAF)) {
            const GpuInner& N = T.inner[cur];
            float k0[64], k1[64];
            bool any0 = false, any1 = false;
            int n0 = 0, nt = 0;
            st.wave_inner++;
            st.box_tests += 2;
            for (int i = 0; i < 64; ++i) {
                const float sc = T.box_scale;   // box records are stored scaled (compact.cpp)
                k0[i] = aabb_d2(N.c0, N.h0, L[i].px * sc, L[i].py * sc, L[i].pz * sc);
                k1[i] = aabb_d2(N.c1, N.h1, L[i].px * sc, L[i].py * sc, L[i].pz * sc);
                if (!L[i].drive) continue;
                st.lane_inner++;
                bool m0 = k0[i] <= L[i].thr, m1 = k1[i] <= L[i].thr;
                any0 |= m0; any1 |= m1;
                if (m0 | m1) { nt++; st.lane_needed++; }
                if (m0 && (k0[i] <= k1[i] || !m1)) n0++;
            }
            if (chain && !(any0 && any1)) st.chain_inner++;
            if (any0 && any1) {
                chain = false;
                bool first0 = 2 * n0 >= nt;
                st.pushes++;
                E& e = stk[sp++];
                e.ref = first0 ? N.ref1 : N.ref0;
                for (int i = 0; i < 64; ++i) e.k[i] = first0 ? k1[i] : k0[i];
                cur = first0 ? N.ref0 : N.ref1;
            } else if (any0) {
                for (int i = 0; i < 64; ++i) L[i].lbk = std::fmin(L[i].lbk, k1[i]);
                cur = N.ref0;
            } else if (any1) {
                for (int i = 0; i < 64; ++i) L[i].lbk = std::fmin(L[i].lbk, k0[i]);
                cur = N.ref1;
            } else {
                for (int i = 0; i < 64; ++i) L[i].lbk = std::fmin(L[i].lbk, std::fmin(k0[i], k1[i]));
                cur = REF_NONE;
                break;
            }
        }
        if (cur != REF_NONE) {
            uint32_t id = cur & REF_INDEX;
            st.wave_leaf++;
            for (int i = 0; i < 64; ++i) {
                float d = leaf_eval(T.leaves[id], L[i].px, L[i].py, L[i].pz);
                if (g_k2) {
                    Lane& l = L[i];
                    if (id == l.bid || id == l.sid) continue;   // a tracked seed met again
                    if (is_better(d, id, l.best, l.bid)) {
                        if (l.second < l.third) l.third = l.second;
                        l.second = l.best; l.sid = l.bid; l.best = d; l.bid = id;
                    } else if (is_better(d, id, l.second, l.sid)) {
                        if (l.second < l.third) l.third = l.second;
                        l.second = d; l.sid = id;
                    } else if (d < l.third) l.third = d;
                    float bp = std::fmax(l.best + l.mg, 0.f) * T.box_scale;
                    l.thr = bp * bp * 1.0006f;
                    continue;
                }
                bool rival = id != L[i].bid;
                bool b = rival && is_better(d, id, L[i].best, L[i].bid);
                float loser = b ? L[i].best : d;
                if (rival && loser < L[i].second) { L[i].second = loser; L[i].sid = b ? L[i].bid : id; }
                if (b) { L[i].best = d; L[i].bid = id; }
                float bp = (std::fmax(L[i].best + L[i].mg, 0.f) + g_ext) * T.box_scale;
                L[i].thr = bp * bp * 1.0006f;
            }
        }
        // pop
        bool got = false;
        while (sp > 0) {
            E& e = stk[--sp];
            bool any = false;
            for (int i = 0; i < 64; ++i) any |= L[i].drive && e.k[i] <= L[i].thr;
            if (any) { cur = e.ref; got = true; break; }
            for (int i = 0; i < 64; ++i) L[i].lbk = std::fmin(L[i].lbk, e.k[i]);
        }
        if (!got) break;
    }
    st.by_drive_walks[ndrive]++;
    st.by_drive_inner[ndrive] += st.wave_inner - inner0;
}

// SIM_WIDE: the same walk over four-wide nodes made by skipping every other level of the binary tree (a
// visit tests the boxes of up to four grandchildren, descends into the one most lanes are nearest to and
// pushes the other needed ones).  wave_inner counts wide visits (one dependent 128-byte fetch each).
static int g_wide_policy = 1;   // SIM_WIDE=1: sorted by nearest-votes; 2: most-voted first, rest by index; 3: sorted by needing lanes
static void walk_wide(Lane* L, Stats& st) {
    uint32_t cur = T.root_ref;
    if (cur == REF_NONE) return;
    struct E { uint32_t ref; float k[64]; };
    static thread_local std::vector<E> stk(256);
    uint32_t sp = 0;
    int ndrive = 0;
    for (int i = 0; i < 64; ++i) {
        ndrive += L[i].drive;
        float bp = (std::fmax(L[i].best + L[i].mg, 0.f) + g_ext) * T.box_scale;
        L[i].thr = bp * bp * 1.0006f;
    }
    uint64_t inner0 = st.wave_inner;
    for (;;) {
        while (!(cur & REF_LEAF)) {
            const GpuInner& N = T.inner[cur];
            const float* cc[4]; const float* hh[4]; uint32_t rr[4]; int nc = 0;
            auto add_child = [&](uint32_t ref, const float* c, const float* h) {
                if (ref & REF_LEAF) { cc[nc] = c; hh[nc] = h; rr[nc++] = ref; return; }
                const GpuInner& M = T.inner[ref];
                cc[nc] = M.c0; hh[nc] = M.h0; rr[nc++] = M.ref0;
                cc[nc] = M.c1; hh[nc] = M.h1; rr[nc++] = M.ref1;
            };
            add_child(N.ref0, N.c0, N.h0);
            add_child(N.ref1, N.c1, N.h1);
            st.wave_inner++;
            st.box_tests += nc;
            float k[4][64]; bool any[4] = {false, false, false, false}; int votes[4] = {0, 0, 0, 0};
            const float sc = T.box_scale;
            for (int i = 0; i < 64; ++i) {
                int bj = -1;
                for (int j = 0; j < nc; ++j) {
                    k[j][i] = aabb_d2(cc[j], hh[j], L[i].px * sc, L[i].py * sc, L[i].pz * sc);
                    if (!L[i].drive) continue;
                    if (k[j][i] <= L[i].thr) { any[j] = true; if (bj < 0 || k[j][i] < k[bj][i]) bj = j; }
                }
                if (!L[i].drive) continue;
                st.lane_inner++;
                if (bj >= 0) { votes[bj]++; st.lane_needed++; }
                if (g_wide_policy == 3) { if (bj >= 0) votes[bj]--; for (int j = 0; j < nc; ++j) votes[j] += k[j][i] <= L[i].thr; }   // needing lanes
            }
            int order[4], no = 0;
            for (int j = 0; j < nc; ++j) {
                if (any[j]) order[no++] = j;
                else for (int i = 0; i < 64; ++i) L[i].lbk = std::fmin(L[i].lbk, k[j][i]);
            }
            if (no == 0) { cur = REF_NONE; break; }
            if (g_wide_policy == 2) {   // most-voted first, the others in index order
                int bq = 0;
                for (int q = 1; q < no; ++q) if (votes[order[q]] > votes[order[bq]]) bq = q;
                std::rotate(order, order + bq, order + bq + 1);
            } else std::stable_sort(order, order + no, [&](int a, int b) { return votes[a] > votes[b]; });
            for (int q = no - 1; q >= 1; --q) {
                st.pushes++;
                E& e = stk[sp++];
                e.ref = rr[order[q]];
                for (int i = 0; i < 64; ++i) e.k[i] = k[order[q]][i];
            }
            cur = rr[order[0]];
        }
        if (cur != REF_NONE) {
            uint32_t id = cur & REF_INDEX;
            st.wave_leaf++;
            for (int i = 0; i < 64; ++i) {
                float d = leaf_eval(T.leaves[id], L[i].px, L[i].py, L[i].pz);
                bool rival = id != L[i].bid;
                bool b = rival && is_better(d, id, L[i].best, L[i].bid);
                float loser = b ? L[i].best : d;
                if (rival && loser < L[i].second) { L[i].second = loser; L[i].sid = b ? L[i].bid : id; }
                if (b) { L[i].best = d; L[i].bid = id; }
                float bp = (std::fmax(L[i].best + L[i].mg, 0.f) + g_ext) * T.box_scale;
                L[i].thr = bp * bp * 1.0006f;
            }
        }
        bool got = false;
        while (sp > 0) {
            E& e = stk[--sp];
            bool anyl = false;
            for (int i = 0; i < 64; ++i) anyl |= L[i].drive && e.k[i] <= L[i].thr;
            if (anyl) { cur = e.ref; got = true; break; }
            for (int i = 0; i < 64; ++i) L[i].lbk = std::fmin(L[i].lbk, e.k[i]);
        }
        if (!got) break;
    }
    st.by_drive_walks[ndrive]++;
    st.by_drive_inner[ndrive] += st.wave_inner - inner0;
}

// ---- SIM_GRID: nearest-candidate grid --------------------------------------------------------------------
// A uniform grid over the scene box (scaled by SIM_GRID_PAD).  For a cell with centre c and half diagonal r every
// primitive that can be the nearest one at ANY point of the cell has d_i(c) <= d_min(c) + 2 r (values are
// 1-Lipschitz), so the cell keeps that candidate list (at most SIM_GRID_K entries, else "overflow": walk) and
// lb = d_min(c) + r, a lower bound at any point of the cell on every primitive NOT in the list.
struct Grid {
    int n = 0, K = 8;
    float lo[3], inv[3], cell[3], r = 0;
    std::vector<uint32_t> off;     // n^3 + 1
    std::vector<uint32_t> idx;
    std::vector<float> lb;         // per cell; < 0: overflow
} G;

static void collect(uint32_t ref, float cx, float cy, float cz, float R, std::vector<std::pair<float, uint32_t>>& out) {
    if (ref & REF_LEAF) {
        uint32_t id = ref & REF_INDEX;
        float d = leaf_eval(T.leaves[id], cx, cy, cz);
        if (d <= R) out.push_back({d, id});
        return;
    }
    const GpuInner& N = T.inner[ref];
    const float sc = T.box_scale;
    float Rs = std::fmax(R, 0.f) * sc;
    Rs = Rs * Rs * 1.001f;
    if (aabb_d2(N.c0, N.h0, cx * sc, cy * sc, cz * sc) <= Rs) collect(N.ref0, cx, cy, cz, R, out);
    if (aabb_d2(N.c1, N.h1, cx * sc, cy * sc, cz * sc) <= Rs) collect(N.ref1, cx, cy, cz, R, out);
}
static float nearest_d(uint32_t ref, float cx, float cy, float cz, float best) {
    if (ref & REF_LEAF) return std::fmin(best, leaf_eval(T.leaves[ref & REF_INDEX], cx, cy, cz));
    const GpuInner& N = T.inner[ref];
    const float sc = T.box_scale;
    float k0 = aabb_d2(N.c0, N.h0, cx * sc, cy * sc, cz * sc), k1 = aabb_d2(N.c1, N.h1, cx * sc, cy * sc, cz * sc);
    uint32_t a = N.ref0, b = N.ref1;
    if (k1 < k0) { std::swap(a, b); std::swap(k0, k1); }
    auto thr = [&](float bst) { float bp = std::fmax(bst, 0.f) * sc; return bp * bp * 1.001f; };
    if (k0 <= thr(best)) best = nearest_d(a, cx, cy, cz, best);
    if (k1 <= thr(best)) best = nearest_d(b, cx, cy, cz, best);
    return best;
}
static void build_grid(int n, int K, float pad) {
    G.n = n; G.K = K;
    for (int a = 0; a < 3; ++a) {
        float c = 0.5f * (T.info.scene_lo[a] + T.info.scene_hi[a]), h = 0.5f * (T.info.scene_hi[a] - T.info.scene_lo[a]) * pad;
        G.lo[a] = c - h; G.cell[a] = 2 * h / n; G.inv[a] = 1.0f / G.cell[a];
    }
    G.r = 0.5f * std::sqrt(G.cell[0] * G.cell[0] + G.cell[1] * G.cell[1] + G.cell[2] * G.cell[2]);
    size_t nc = (size_t)n * n * n;
    std::vector<std::vector<uint32_t>> lists(nc);
    G.lb.assign(nc, -1.f);
#pragma omp parallel for schedule(dynamic, 256)
    for (size_t c = 0; c < nc; ++c) {
        int x = c % n, y = (c / n) % n, z = c / ((size_t)n * n);
        float cx = G.lo[0] + (x + 0.5f) * G.cell[0], cy = G.lo[1] + (y + 0.5f) * G.cell[1], cz = G.lo[2] + (z + 0.5f) * G.cell[2];
        float dmin = nearest_d(T.root_ref, cx, cy, cz, INFINITY);
        std::vector<std::pair<float, uint32_t>> cand;
        collect(T.root_ref, cx, cy, cz, dmin + 2.0f * G.r * 1.001f + 1e-3f, cand);
        if ((int)cand.size() > K) continue;   // overflow: lb stays < 0
        std::sort(cand.begin(), cand.end());
        for (auto& e : cand) lists[c].push_back(e.second);
        G.lb[c] = dmin + G.r;
    }
    G.off.assign(nc + 1, 0);
    for (size_t c = 0; c < nc; ++c) G.off[c + 1] = G.off[c] + (uint32_t)lists[c].size();
    G.idx.resize(G.off[nc]);
    size_t over = 0; uint64_t hist[33] = {0};
    for (size_t c = 0; c < nc; ++c) {
        std::copy(lists[c].begin(), lists[c].end(), G.idx.begin() + G.off[c]);
        if (G.lb[c] < 0) over++; else hist[std::min<size_t>(32, lists[c].size())]++;
    }
    printf("grid %d^3 cells, cell %.2f, r %.2f, K %d: overflow cells %.3f, mean list %.2f, bytes %.1f MB\n", n, G.cell[0], G.r, K,
           (double)over / nc, (double)G.idx.size() / (nc - over), (nc * 8.0 + G.idx.size() * 4.0) / 1e6);
    {   // bricks of 4^3 cells: union of the cells' lists
        const int B = 4, nb = n / B, cap = getenv("SIM_BRICK_CAP") ? atoi(getenv("SIM_BRICK_CAP")) : 32;
        size_t bover = 0, cells_lost = 0; double usum = 0;
        for (int bz = 0; bz < nb; ++bz) for (int by = 0; by < nb; ++by) for (int bx = 0; bx < nb; ++bx) {
            std::vector<uint32_t> u;
            int good = 0;
            for (int z = 0; z < B; ++z) for (int y = 0; y < B; ++y) for (int x = 0; x < B; ++x) {
                size_t c = ((size_t)(bz * B + z) * n + (by * B + y)) * n + (bx * B + x);
                if (G.lb[c] < 0) continue;
                good++;
                u.insert(u.end(), G.idx.begin() + G.off[c], G.idx.begin() + G.off[c + 1]);
            }
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            usum += u.size();
            if ((int)u.size() > cap) {
                bover++; cells_lost += good;
                for (int z = 0; z < B; ++z) for (int y = 0; y < B; ++y) for (int x = 0; x < B; ++x)
                    G.lb[((size_t)(bz * B + z) * n + (by * B + y)) * n + (bx * B + x)] = -1.f;
            }
        }
        printf("  bricks %d^3: mean union %.2f, bricks over the cap of %d: %.4f (cells lost %.4f)\n", nb, usum / ((double)nb * nb * nb), cap,
               (double)bover / ((double)nb * nb * nb), (double)cells_lost / nc);
    }
    printf("  list length histogram:");
    for (int i = 0; i <= 32; ++i) if (hist[i]) printf(" %d:%.3f", i, (double)hist[i] / nc);
    printf("\n");
}
static inline long grid_cell(float px, float py, float pz) {
    float fx = (px - G.lo[0]) * G.inv[0], fy = (py - G.lo[1]) * G.inv[1], fz = (pz - G.lo[2]) * G.inv[2];
    if (!(fx >= 0 && fy >= 0 && fz >= 0 && fx < G.n && fy < G.n && fz < G.n)) return -1;
    return ((long)fz * G.n + (long)fy) * G.n + (long)fx;
}

struct GridStats {
    uint64_t iters = 0, list_iters = 0, list_len_max_sum = 0, list_len_sum = 0, list_lanes = 0, cert_lanes = 0, walk_lanes = 0,
             outside = 0, overflow = 0, uniq_sum = 0, cells_sum = 0, cell_len_sum = 0, bricks_sum = 0;
    void add(const GridStats& o) {
        iters += o.iters; list_iters += o.list_iters; list_len_max_sum += o.list_len_max_sum; list_len_sum += o.list_len_sum;
        list_lanes += o.list_lanes; cert_lanes += o.cert_lanes; walk_lanes += o.walk_lanes; outside += o.outside; overflow += o.overflow;
        uniq_sum += o.uniq_sum; cells_sum += o.cells_sum; cell_len_sum += o.cell_len_sum; bricks_sum += o.bricks_sum;
    }
} g_gs;

// March step with the grid: certified lanes step on their seed; the others step on their cell's list; lanes
// whose point has no list (outside the grid, overflow cell) wait until no lane of the wave can step otherwise,
// then the wave walks for them.  policy 1: lanes with a list also wait while some lane can certify (deferral).
static void wave_step_grid(Lane* L, Stats& st, GridStats& gs, int policy) {
    int nalive = 0, ncert = 0, nlist = 0, nwalk = 0;
    long cellof[64];
    for (int i = 0; i < 64; ++i) {
        Lane& l = L[i];
        l.drive = false; l.certified = false;
        l.best = INFINITY; l.bid = SDF_NO_PRIM; l.second = INFINITY; l.lbk = INFINITY;
        cellof[i] = -1;
        if (!l.alive) continue;
        nalive++;
        l.px = ox + l.dx * l.t; l.py = oy + l.dy * l.t; l.pz = oz + l.dz * l.t;
        l.mg = 3.0e-6f * g_lip * (std::fabs(l.px) + std::fabs(l.py) + std::fabs(l.pz) + T.margin_T);
        if (l.last != SDF_NO_PRIM) {
            l.best = leaf_eval(T.leaves[l.last], l.px, l.py, l.pz);
            l.bid = l.last;
            l.certified = (l.best + 2.0f * l.mg) < l.budget;
        }
        if (l.certified) { ncert++; continue; }
        long c = grid_cell(l.px, l.py, l.pz);
        if (c >= 0 && G.lb[c] >= 0) { cellof[i] = c; nlist++; }
        else { l.drive = true; nwalk++; if (c < 0) gs.outside++; else gs.overflow++; }
    }
    if (nalive == 0) return;
    st.wave_steps++;
    st.by_alive_steps[nalive]++;
    gs.iters++;
    const bool lists_go = nlist > 0 && !((policy == 1 || policy == 2) && ncert > 0);
    if (lists_go && policy == 2) {   // only the brick most listed lanes stand in takes its step; the others wait
        std::map<long, int> cnt;
        auto brick_of = [&](long c) { long x = c % G.n, y = (c / G.n) % G.n, z = c / ((long)G.n * G.n); return ((z / 4) * (G.n / 4) + y / 4) * (G.n / 4) + x / 4; };
        for (int i = 0; i < 64; ++i) if (cellof[i] >= 0) cnt[brick_of(cellof[i])]++;
        long bestb = -1; int bestc = 0;
        for (auto& kv : cnt) if (kv.second > bestc) { bestc = kv.second; bestb = kv.first; }
        for (int i = 0; i < 64; ++i) if (cellof[i] >= 0 && brick_of(cellof[i]) != bestb) { cellof[i] = -2; }
    }
    const bool do_walk = nwalk > 0 && ncert == 0 && nlist == 0;
    if (do_walk) {
        st.wave_walks++;
        st.lane_walks += nwalk;
        for (int i = 0; i < 64; ++i) L[i].sid = SDF_NO_PRIM;
        walk(L, st);
    }
    if (lists_go) {
        gs.list_iters++;
        uint32_t mx = 0;
        std::vector<uint32_t> uniq;
        for (int i = 0; i < 64; ++i) {
            if (cellof[i] < 0) continue;
            Lane& l = L[i];
            const long c = cellof[i];
            const uint32_t n = G.off[c + 1] - G.off[c];
            mx = std::max(mx, n);
            gs.list_len_sum += n; gs.list_lanes++;
            l.best = INFINITY; l.bid = SDF_NO_PRIM; l.second = INFINITY;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t id = G.idx[G.off[c] + j];
                uniq.push_back(id);
                const float d = leaf_eval(T.leaves[id], l.px, l.py, l.pz);
                const bool b = is_better(d, id, l.best, l.bid);
                const float loser = b ? l.best : d;
                if (loser < l.second) l.second = loser;
                if (b) { l.best = d; l.bid = id; }
            }
            l.budget = std::fmin(l.second, G.lb[c]) - 2.f * l.mg;
        }
        std::sort(uniq.begin(), uniq.end());
        gs.uniq_sum += std::unique(uniq.begin(), uniq.end()) - uniq.begin();
        std::vector<long> cs;
        for (int i = 0; i < 64; ++i) if (cellof[i] >= 0) cs.push_back(cellof[i]);
        std::sort(cs.begin(), cs.end());
        cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
        gs.cells_sum += cs.size();
        for (long c : cs) gs.cell_len_sum += G.off[c + 1] - G.off[c];
        std::vector<long> bs;
        for (long c : cs) { long x = c % G.n, y = (c / G.n) % G.n, z = c / ((long)G.n * G.n); bs.push_back(((z / 4) * (G.n / 4) + y / 4) * (G.n / 4) + x / 4); }
        std::sort(bs.begin(), bs.end());
        bs.erase(std::unique(bs.begin(), bs.end()), bs.end());
        gs.bricks_sum += bs.size();
        gs.list_len_max_sum += mx;
    }
    for (int i = 0; i < 64; ++i) {
        Lane& l = L[i];
        if (!l.alive) continue;
        bool adv;
        if (l.certified) adv = true;
        else if (cellof[i] >= 0) adv = lists_go;
        else if (cellof[i] == -2) adv = false;
        else adv = do_walk;
        if (!adv) continue;
        if (l.certified) gs.cert_lanes++;
        if (do_walk && l.drive) {
            gs.walk_lanes++;
            l.budget = std::fmin(std::sqrt(l.lbk) / T.box_scale * 0.9996f, l.second) - l.mg;
        }
        l.steps++;
        st.lane_steps++;
        l.last = l.bid;
        if (l.best < mp.eps_hit) { l.hit = true; l.alive = false; }
        else {
            l.t = l.t + l.best;
            l.budget = (l.budget - std::fabs(l.best) * g_lip) - l.mg;
            if (l.t > mp.t_max) l.alive = false;
        }
        if (l.alive && l.steps >= mp.max_steps) l.alive = false;
    }
}

// One march step of a wave.  defer_thr > 0: lanes that cannot certify wait (do not advance) while fewer than
// defer_thr of them are waiting and some certified lane can still advance.
static void wave_step(Lane* L, Stats& st, int defer_thr, std::vector<int>* finished) {
    int nalive = 0, nunc = 0, ncert = 0;
    for (int i = 0; i < 64; ++i) {
        Lane& l = L[i];
        l.drive = false; l.certified = false;
        l.best = INFINITY; l.bid = SDF_NO_PRIM; l.second = INFINITY; l.lbk = INFINITY;
        if (!l.alive) continue;
        nalive++;
        l.px = ox + l.dx * l.t; l.py = oy + l.dy * l.t; l.pz = oz + l.dz * l.t;
        l.mg = 3.0e-6f * g_lip * (std::fabs(l.px) + std::fabs(l.py) + std::fabs(l.pz) + T.margin_T);
        l.third = INFINITY; l.sid = SDF_NO_PRIM;
        if (l.last != SDF_NO_PRIM) {
            l.best = leaf_eval(T.leaves[l.last], l.px, l.py, l.pz);
            l.bid = l.last;
            if (g_k2 && l.last2 != SDF_NO_PRIM) {
                float d2 = leaf_eval(T.leaves[l.last2], l.px, l.py, l.pz);
                if (is_better(d2, l.last2, l.best, l.bid)) { l.second = l.best; l.sid = l.bid; l.best = d2; l.bid = l.last2; }
                else { l.second = d2; l.sid = l.last2; }
            }
            l.certified = (l.best + 2.0f * l.mg) < l.budget;
        }
        l.drive = !l.certified;
        nunc += l.drive; ncert += l.certified;
    }
    if (nalive == 0) return;
    st.wave_steps++;
    st.by_alive_steps[nalive]++;
    uint64_t inner0 = st.wave_inner;
    bool do_walk = nunc > 0;
    bool deferring = false;
    static const int min_cert = getenv("SIM_MINCERT") ? atoi(getenv("SIM_MINCERT")) : 1;   // walk early unless at least this many lanes can certify
    if (defer_thr > 0 && nunc > 0 && nunc < defer_thr && ncert >= min_cert) { do_walk = false; deferring = true; }
    if (do_walk) {
        if (getenv("SIM_SEED2"))
            for (int i = 0; i < 64; ++i) {
                Lane& l = L[i];
                if (!l.alive || l.last2 == SDF_NO_PRIM || l.last2 == l.bid) continue;
                float d = leaf_eval(T.leaves[l.last2], l.px, l.py, l.pz);
                if (is_better(d, l.last2, l.best, l.bid)) { l.best = d; l.bid = l.last2; }
            }
        if (!g_k2) for (int i = 0; i < 64; ++i) L[i].sid = SDF_NO_PRIM;
        st.wave_walks++;
        st.lane_walks += nunc;
        static const bool wide = getenv("SIM_WIDE") != nullptr && (g_wide_policy = atoi(getenv("SIM_WIDE"))) > 0;
        if (wide) walk_wide(L, st); else walk(L, st);
    }
    st.by_alive_inner[nalive] += st.wave_inner - inner0;
    for (int i = 0; i < 64; ++i) {
        Lane& l = L[i];
        if (!l.alive) continue;
        if (deferring && l.drive) continue;   // waits: same point next time
        if (do_walk) {
            float lb = std::fmin(std::sqrt(l.lbk) / T.box_scale * 0.9996f, g_k2 ? l.third : l.second) - l.mg;
            l.budget = l.certified ? std::fmax(l.budget, lb) : lb;
        }
        l.steps++;
        st.lane_steps++;
        l.last = l.bid;
        if (do_walk || g_k2) l.last2 = l.sid;
        if (l.best < mp.eps_hit) { l.hit = true; l.alive = false; }
        else {
            l.t = l.t + l.best;
            l.budget = (l.budget - std::fabs(l.best) * g_lip) - l.mg;
            if (l.t > mp.t_max) l.alive = false;
        }
        if (l.alive && l.steps >= mp.max_steps) l.alive = false;
        if (!l.alive && finished) finished->push_back(l.pix);
    }
}

int main(int argc, char** argv) {
    std::string strat = argc > 1 ? argv[1] : "tile";
    int thr = argc > 2 ? atoi(argv[2]) : 32;
    W = argc > 3 ? atoi(argv[3]) : 1920;
    H = argc > 4 ? atoi(argv[4]) : 1080;
    uint32_t n_prims = argc > 5 ? atoi(argv[5]) : 10000;
    uint64_t seed = argc > 6 ? strtoull(argv[6], nullptr, 0) : 0x5DF00003ull;
    int defer = argc > 7 ? atoi(argv[7]) : 0;
    int tw = argc > 8 ? atoi(argv[8]) : 8, th = 64 / tw;

    if (getenv("SIM_EXT")) g_ext = atof(getenv("SIM_EXT"));
    g_k2 = getenv("SIM_K2") != nullptr;
    sdfhost_node* node = sdfhost_random_scene(n_prims, seed, 0.f, nullptr);
    sdfhost_expanded* ex = sdfhost_expanded_new(node, nullptr);
    sdfhost_buffer* buf = sdfhost_make_buffer(ex, nullptr);
    std::string err;
    int rc = compact_tree(sdfhost_buffer_downtree(buf), sdfhost_buffer_uptree(buf), sdfhost_buffer_len(buf), 0, SDFGPU_BOUNDS_AUTO, 0u, &T, &err);
    if (rc) { fprintf(stderr, "compact: %s\n", err.c_str()); return 1; }
    g_lip = T.lipschitz;
    float L = sdfhost_default_extent(n_prims);
    float eye[3] = {0, 0, -2.5f * L}, tgt[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    look_at(eye, tgt, up, (float)(M_PI / 4), W, H, &cam);
    mp = {128, 1e-3f, 0.f, 6.0f * L, 1e-4f, 0};
    ox = cam.eye[0]; oy = cam.eye[1]; oz = cam.eye[2];

    const int grid_n = getenv("SIM_GRID") ? atoi(getenv("SIM_GRID")) : 0;
    const int grid_policy = getenv("SIM_GRID_POLICY") ? atoi(getenv("SIM_GRID_POLICY")) : 0;
    if (grid_n > 0) build_grid(grid_n, getenv("SIM_GRID_K") ? atoi(getenv("SIM_GRID_K")) : 8, getenv("SIM_GRID_PAD") ? atof(getenv("SIM_GRID_PAD")) : 1.0f);
    const uint32_t tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th, n_tiles = tiles_x * tiles_y;
    Stats gst;
    std::vector<double> wave_cost;   // per-wave VALU (critical path)
    if (strat == "tile") {
        wave_cost.resize(n_tiles);
#pragma omp parallel for schedule(dynamic, 16)
        for (uint32_t t = 0; t < n_tiles; ++t) {
            Stats st;
            GridStats gs;
            Lane Ls[64];
            for (int i = 0; i < 64; ++i) init_ray(Ls[i], (t % tiles_x) * tw + i % tw, (t / tiles_x) * th + i / tw);
            double c0 = st.valu();
            for (;;) {
                bool any = false;
                for (auto& l : Ls) any |= l.alive;
                if (!any) break;
                if (grid_n > 0) wave_step_grid(Ls, st, gs, grid_policy); else wave_step(Ls, st, defer, nullptr);
            }
            wave_cost[t] = st.valu() - c0 + 30.0 * gs.list_iters + 55.0 * gs.list_len_max_sum;
#pragma omp critical
            { gst.add(st); g_gs.add(gs); }
        }
    } else if (strat == "refill") {
        // persistent waves, each owning a contiguous strip of `strip` tiles; dead lanes are refilled from the strip's
        // pixel stream when alive <= thr
        const uint32_t strip = argc > 9 ? atoi(argv[9]) : 16;
        wave_cost.resize((n_tiles + strip - 1) / strip);
#pragma omp parallel for schedule(dynamic, 4)
        for (uint32_t t0 = 0; t0 < n_tiles; t0 += strip) {
            Stats st;
            Lane Ls[64];
            uint64_t next = (uint64_t)t0 * 64, end = (uint64_t)std::min(n_tiles, t0 + strip) * 64;
            double c0 = st.valu();
            for (;;) {
                int nalive = 0;
                for (auto& l : Ls) nalive += l.alive;
                if (nalive <= thr && next < end) {
                    for (int i = 0; i < 64 && next < end; ++i)
                        if (!Ls[i].alive) {
                            uint32_t t = next >> 6, p = next & 63; next++;
                            init_ray(Ls[i], (t % tiles_x) * tw + p % tw, (t / tiles_x) * th + p / tw);
                        }
                    nalive = 0;
                    for (auto& l : Ls) nalive += l.alive;
                }
                if (nalive == 0) { if (next >= end) break; else continue; }
                wave_step(Ls, st, defer, nullptr);
            }
            wave_cost[t0 / strip] = st.valu() - c0;
#pragma omp critical
            gst.add(st);
        }
    }
    Stats& st = gst;
    printf("strategy %s thr %d defer %d tile %dx%d\n", strat.c_str(), thr, defer, tw, th);
    printf("wave_steps %lu wave_walks %lu wave_inner %lu wave_leaf %lu lane_steps %lu lane_walks %lu\n", st.wave_steps, st.wave_walks,
           st.wave_inner, st.wave_leaf, st.lane_steps, st.lane_walks);
    printf("inner/walk %.1f lane_inner/lane_walk %.1f needed/lane_walk %.1f  VALU est %.3fe9\n", (double)st.wave_inner / st.wave_walks,
           (double)st.lane_inner / st.lane_walks, (double)st.lane_needed / st.lane_walks, st.valu() / 1e9);
    if (grid_n > 0) {
        const GridStats& g = g_gs;
        printf("grid: iterations %lu, with a list pass %lu (max list length per pass %.2f, mean per lane %.2f, unique prims per pass %.2f)\n", g.iters,
               g.list_iters, (double)g.list_len_max_sum / std::max<uint64_t>(1, g.list_iters), (double)g.list_len_sum / std::max<uint64_t>(1, g.list_lanes),
               (double)g.uniq_sum / std::max<uint64_t>(1, g.list_iters));
        printf("grid: distinct cells per list pass %.2f, sum of their list lengths %.2f, distinct bricks %.2f\n", (double)g.cells_sum / std::max<uint64_t>(1, g.list_iters),
               (double)g.cell_len_sum / std::max<uint64_t>(1, g.list_iters), (double)g.bricks_sum / std::max<uint64_t>(1, g.list_iters));
        printf("grid: lane-steps certified %lu, by list %lu, by walk %lu; waits outside the grid %lu, in overflow cells %lu\n", g.cert_lanes,
               g.list_lanes, g.walk_lanes, g.outside, g.overflow);
        printf("grid: VALU est %.3fe9 (walk part %.3fe9 + lists %.3fe9)\n", (st.valu() + 30.0 * g.list_iters + 55.0 * g.list_len_max_sum) / 1e9,
               st.valu() / 1e9, (30.0 * g.list_iters + 55.0 * g.list_len_max_sum) / 1e9);
    }
    printf("box tests %lu (%.2f per node visit)  stack pushes %lu\n", st.box_tests, (double)st.box_tests / st.wave_inner, st.pushes);
    printf("leading single-child chain: %.2f inner visits per walk\n", (double)st.chain_inner / st.wave_walks);
    if (strat == "tile")
        for (int nr : {2, 4, 8}) {   // round-robin tile deal over nr ranks: heaviest rank against the mean
            std::vector<double> sum(nr, 0.0);
            double tot = 0;
            for (size_t t = 0; t < wave_cost.size(); ++t) { sum[t % nr] += wave_cost[t]; tot += wave_cost[t]; }
            printf("round-robin over %d ranks: heaviest share / mean share = %.4f\n", nr, *std::max_element(sum.begin(), sum.end()) / (tot / nr));
        }
    std::sort(wave_cost.begin(), wave_cost.end());
    printf("waves %zu  max wave VALU %.0f  p99 %.0f  median %.0f\n", wave_cost.size(), wave_cost.back(),
           wave_cost[wave_cost.size() * 99 / 100], wave_cost[wave_cost.size() / 2]);
    const int edges[] = {1, 2, 4, 8, 16, 32, 48, 63, 64};
    int lo = 1;
    printf("alive bucket: share of wave-steps, share of inner visits, inner per step\n");
    uint64_t ts = 0, ti = 0;
    for (int a = 1; a <= 64; ++a) { ts += st.by_alive_steps[a]; ti += st.by_alive_inner[a]; }
    for (int e : edges) {
        uint64_t s = 0, in = 0;
        for (int a = lo; a <= e; ++a) { s += st.by_alive_steps[a]; in += st.by_alive_inner[a]; }
        printf("  %2d..%2d: %.3f %.3f %.1f\n", lo, e, (double)s / ts, (double)in / ti, s ? (double)in / s : 0.0);
        lo = e + 1;
    }
    lo = 1;
    printf("drive bucket: share of walks, share of inner visits, inner per walk\n");
    uint64_t tw_ = 0;
    for (int a = 1; a <= 64; ++a) tw_ += st.by_drive_walks[a];
    for (int e : edges) {
        uint64_t s = 0, in = 0;
        for (int a = lo; a <= e; ++a) { s += st.by_drive_walks[a]; in += st.by_drive_inner[a]; }
        printf("  %2d..%2d: %.3f %.3f %.1f\n", lo, e, (double)s / tw_, (double)in / ti, s ? (double)in / s : 0.0);
        lo = e + 1;
    }
    return 0;
}
