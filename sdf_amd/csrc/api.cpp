// api.cpp -- the C ABI of include/sdfgpu.h over the kernels.
//
// No CPU fallback exists anywhere in this file: every query entry point ends
// in a kernel launch or returns an error code.  oracle/ is never referenced.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sdfgpu.h"
#include "gpu_tree.hpp"
#include "kernels.hpp"

using namespace sdfgpu;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct sdfgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DeviceTree tree;
    bool has_tree = false;
    sdfgpu_tree_info info{};
    std::string err;
    // grow-only scratch for the host-pointer entry points
    DevBuf in, out_a, out_b, out_c, out_d, stats;
    // One scratch set per stream the context has worked on: frames and query batches in flight on
    // different streams must not share pass state, tile schedules or sort buffers.
    struct MarchScratch {
        hipStream_t stream;
        // spatial reordering of query batches (sort.hip): keys in / out, permutation in / out, radix sort storage
        DevBuf keys_a, keys_b, idx_a, idx_b, sort_tmp;
        DevBuf eye;                 // 16 bytes: result of the frame's one walk from the eye (kernels.hpp, RenderScratch::eye)
        DevBuf buf;                 // pass state of the multi-pass march (compaction)
        // longest-first tile order from the costs the previous frame on this stream recorded
        DevBuf order;               // [cost | iota | sorted keys | order] x tiles, then the sort's temp storage
        uint32_t order_tiles = 0;   // tile count the buffers are laid out for
        size_t order_temp = 0;      // bytes of sort temp storage behind the four arrays
        uint32_t order_geom[6] = {0, 0, 0, 0, 0, 0};   // frame geometry the recorded costs belong to
        bool order_valid = false;
        uint32_t order_age = 0;     // frames rendered with the current order while the view stood still
        sdfgpu_camera order_cam{};  // the view the current order was learned from
        uint64_t stamp = 0;         // last use (least recently used entry is recycled)
        // recorded behind the last work that used this entry's buffers: what recycling waits for (the stream handle
        // may belong to a stream the caller has destroyed by then; the event is the context's own)
        hipEvent_t last_use = nullptr;
    };
    std::vector<MarchScratch> march;
    uint64_t march_clock = 0;
    bool tile_order = true;         // SDFGPU_OPT_TILE_ORDER
    bool compaction = false;
    std::vector<uint32_t> pass_steps;   // interior pass boundaries (march steps)
    bool reorder = true;
    bool stats_asm_walk = false;    // SDFGPU_OPT_STATS_ASM_WALK
    bool candidate_grid = true;     // SDFGPU_OPT_CANDIDATE_GRID: the next upload builds the march's candidate grid
    // Large query batches: the batch is cut into chunks, and the Morton sort of chunk i + 1 (bandwidth-bound,
    // on the helper stream) runs beside the walk of chunk i (VALU-bound, on the caller's stream).
    uint32_t query_chunks = 0;      // SDFGPU_OPT_QUERY_PIPELINE (0 / 1: off, the default: measured slower)
    hipStream_t aux = nullptr;
    hipEvent_t ev_start = nullptr;
    struct PipeSet {
        DevBuf keys_a, keys_b, idx_a, idx_b, tmp;
        hipEvent_t sorted = nullptr, walked = nullptr;
        bool walked_recorded = false;   // a walk of an earlier call may still read this set
    } pipe[2];
    float skip_ext_opt = 0.f;    // SDFGPU_OPT_SKIP_EXTENSION as set (< 0: half the mean primitive size)
    float mean_extent = 0.f;
    // sdfgpu_nearest_dev_static: address of sort.hip's file-scope scratch array on this device
    void* static_scratch = nullptr;
    hipError_t static_scratch_err = hipSuccess;
};

namespace {

int set_err(sdfgpu_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

int hip_fail(sdfgpu_ctx* c, hipError_t e, const char* what) {
    return set_err(c, SDFGPU_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return hip_fail(ctx, e__, #call); \
    } while (0)

int reserve(sdfgpu_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return SDFGPU_OK;
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(c, hipMalloc(&b.p, want));
    b.cap = want;
    return SDFGPU_OK;
}

void free_tree(sdfgpu_ctx* c) {
    if (c->tree.leaf) (void)hipFree(c->tree.leaf);
    if (c->tree.inner) (void)hipFree(c->tree.inner);
    if (c->tree.inner_obb) (void)hipFree(c->tree.inner_obb);
    if (c->tree.gen) (void)hipFree(c->tree.gen);
    if (c->tree.chain) (void)hipFree(c->tree.chain);
    if (c->tree.grid.block) (void)hipFree(c->tree.grid.block);   // block, cells and brick ids are one allocation
    c->tree = DeviceTree();
    c->has_tree = false;
}

int check_ready(sdfgpu_ctx* c) {
    if (!c) return SDFGPU_ERR_INVALID_ARG;
    if (!c->has_tree) return set_err(c, SDFGPU_ERR_NO_TREE, "no tree uploaded (call sdfgpu_upload_tree first)");
    HIP_TRY(c, hipSetDevice(c->device));
    return SDFGPU_OK;
}

int check_march(sdfgpu_ctx* c, const sdfgpu_camera* cam, uint32_t w, uint32_t h, const sdfgpu_march_params* mp) {
    if (!cam || !mp) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null camera / march params");
    if (w == 0 || h == 0) return set_err(c, SDFGPU_ERR_INVALID_ARG, "empty image");
    if ((uint64_t)w * h > (1ull << 30)) return set_err(c, SDFGPU_ERR_TOO_LARGE, "image larger than 2^30 pixels");
    return SDFGPU_OK;
}

sdfgpu_ctx::MarchScratch* stream_scratch(sdfgpu_ctx* c, hipStream_t stream);
void mark_used(sdfgpu_ctx* c, hipStream_t stream);

// Space-filling-curve permutation of a device point batch (nullptr when reordering is off
// or not worth it).  Scratch is grow-only, so steady-state calls allocate nothing.
constexpr size_t REORDER_MIN_POINTS = 2048;
int make_order(sdfgpu_ctx* c, const float* d_pts, size_t n, hipStream_t s, const uint32_t** order, bool use_static = false) {
    *order = nullptr;
    if (!c->reorder || n < REORDER_MIN_POINTS || (c->tree.gen && !c->tree.gen_prune)) return SDFGPU_OK;   // (an exhaustive walk visits everything in any order)
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        // twice the scene box: queries usually sit around the scene; outliers are clamped
        float cen = 0.5f * (c->info.scene_lo[a] + c->info.scene_hi[a]);
        float half = (c->info.scene_hi[a] - c->info.scene_lo[a]) + 1.0f;
        lo[a] = cen - half;
        hi[a] = cen + half;
    }
    const size_t tmp = sort_temp_bytes((uint32_t)n);
    uint32_t *keys_a, *keys_b, *idx_a, *idx_b;
    void* temp;
    if (use_static) {
        // the file-scope array of sort.hip, through the address looked up at sdfgpu_create: nothing is allocated
        constexpr size_t arr = (size_t)SDFGPU_STATIC_MAX_POINTS * sizeof(uint32_t);
        static_assert(4 * arr < STATIC_SCRATCH_BYTES, "static scratch holds the four arrays and the sort's storage");
        if (4 * arr + tmp > STATIC_SCRATCH_BYTES)
            return set_err(c, SDFGPU_ERR_TOO_LARGE, "radix sort storage does not fit the static scratch");
        char* base = (char*)c->static_scratch;
        keys_a = (uint32_t*)base;
        keys_b = (uint32_t*)(base + arr);        // the sort's 8-byte pairs: this array and the next one
        idx_a = (uint32_t*)(base + 2 * arr);
        idx_b = (uint32_t*)(base + 3 * arr);
        temp = base + 4 * arr;
    } else {
        // the buffers of the stream the batch runs on: batches in flight on different streams do not share them
        sdfgpu_ctx::MarchScratch* ms = stream_scratch(c, s);
        int rc;
        // keys_b holds the sort's 8-byte (digit, index) pairs (sort.hip); idx_a only serves the SDFGPU_SORT=hipcub comparison path
        if ((rc = reserve(c, ms->keys_a, n * 4)) || (rc = reserve(c, ms->keys_b, n * 8)) || (rc = reserve(c, ms->idx_a, n * 4)) ||
            (rc = reserve(c, ms->idx_b, n * 4)) || (rc = reserve(c, ms->sort_tmp, tmp)))
            return rc;
        keys_a = (uint32_t*)ms->keys_a.p;
        keys_b = (uint32_t*)ms->keys_b.p;
        idx_a = (uint32_t*)ms->idx_a.p;
        idx_b = (uint32_t*)ms->idx_b.p;
        temp = ms->sort_tmp.p;
    }
    HIP_TRY(c, launch_morton_order(d_pts, (uint32_t)n, lo, hi, keys_a, keys_b, idx_a, idx_b, temp, tmp, s));
    *order = idx_b;
    return SDFGPU_OK;
}


// Reordered query over a large batch, chunked and overlapped (see sdfgpu_ctx::query_chunks).  `launch`
// walks one chunk: launch(first point of the chunk, permutation of the chunk or nullptr, points in the
// chunk, index of its first point in the batch).  Everything the caller's stream has to wait for is
// expressed with events, so the call stays asynchronous and can be captured in a graph.
constexpr size_t PIPELINE_MIN_POINTS = 1u << 20;
template <class F>
int ordered_query(sdfgpu_ctx* c, const float* d_pts, size_t n, hipStream_t s, F&& launch) {
    const bool pipelined = c->reorder && !c->tree.gen && c->query_chunks > 1 && n >= PIPELINE_MIN_POINTS;
    if (!pipelined) {
        const uint32_t* order = nullptr;
        int rc = make_order(c, d_pts, n, s, &order);
        if (rc) return rc;
        rc = launch(d_pts, order, n, (size_t)0);
        if (order) mark_used(c, s);
        return rc;
    }
    if (!c->aux) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
        for (auto& ps : c->pipe) {
            HIP_TRY(c, hipEventCreateWithFlags(&ps.sorted, hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&ps.walked, hipEventDisableTiming));
        }
    }
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        float cen = 0.5f * (c->info.scene_lo[a] + c->info.scene_hi[a]);
        float half = (c->info.scene_hi[a] - c->info.scene_lo[a]) + 1.0f;
        lo[a] = cen - half;
        hi[a] = cen + half;
    }
    const size_t chunk = (((n + c->query_chunks - 1) / c->query_chunks) + 255) & ~(size_t)255;
    const size_t tmp = sort_temp_bytes((uint32_t)chunk);
    for (auto& ps : c->pipe) {
        int rc;
        if ((rc = reserve(c, ps.keys_a, chunk * 4)) || (rc = reserve(c, ps.keys_b, chunk * 8)) || (rc = reserve(c, ps.idx_a, chunk * 4)) ||
            (rc = reserve(c, ps.idx_b, chunk * 4)) || (rc = reserve(c, ps.tmp, tmp)))
            return rc;
    }
    // the helper stream starts after everything already queued on the caller's stream (the inputs) -- and after the
    // last walks of an EARLIER pipelined call, which may have run on another caller stream and may still read the
    // two sets' permutations (two large batches in flight on two streams share the sets)
    HIP_TRY(c, hipEventRecord(c->ev_start, s));
    HIP_TRY(c, hipStreamWaitEvent(c->aux, c->ev_start, 0));
    for (auto& ps : c->pipe)
        if (ps.walked_recorded) HIP_TRY(c, hipStreamWaitEvent(c->aux, ps.walked, 0));
    size_t first = 0;
    for (uint32_t k = 0; first < n; ++k, first += chunk) {
        sdfgpu_ctx::PipeSet& ps = c->pipe[k & 1];
        const size_t m = n - first < chunk ? n - first : chunk;
        if (k >= 2) HIP_TRY(c, hipStreamWaitEvent(c->aux, ps.walked, 0));   // the walk two chunks back has let go of this set
        HIP_TRY(c, launch_morton_order(d_pts + 4 * first, (uint32_t)m, lo, hi, (uint32_t*)ps.keys_a.p, (uint32_t*)ps.keys_b.p,
                                       (uint32_t*)ps.idx_a.p, (uint32_t*)ps.idx_b.p, ps.tmp.p, sort_temp_bytes((uint32_t)m), c->aux));
        HIP_TRY(c, hipEventRecord(ps.sorted, c->aux));
        HIP_TRY(c, hipStreamWaitEvent(s, ps.sorted, 0));
        int rc = launch(d_pts + 4 * first, (const uint32_t*)ps.idx_b.p, m, first);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(ps.walked, s));
        ps.walked_recorded = true;
    }
    return SDFGPU_OK;
}

// Pass boundaries of the multi-pass march for a given step budget, and the
// scratch it needs.  Default schedule: survivors thin out quickly at first and
// slowly later (DESIGN.md "ray compaction"); SDFGPU_PASS_STEPS="12,20,..."
// overrides it for tuning.
std::vector<uint32_t> default_pass_steps() {
    std::vector<uint32_t> v;
    if (const char* e = std::getenv("SDFGPU_PASS_STEPS")) {
        const char* p = e;
        while (*p) {
            char* end = nullptr;
            unsigned long x = std::strtoul(p, &end, 10);
            if (end == p) break;
            if (x > 0) v.push_back((uint32_t)x);
            p = (*end == ',') ? end + 1 : end;
        }
        return v;
    }
    return {40};   // two passes; best of the schedules measured on the bench frame
}

// Per-stream scratch of a context.  At most 16 streams keep theirs; beyond that the least recently
// used entry is recycled (its stream is synchronised first, its device memory freed), so a caller that
// creates and destroys streams over time neither leaks nor runs out.
sdfgpu_ctx::MarchScratch* stream_scratch(sdfgpu_ctx* c, hipStream_t stream) {
    for (auto& m : c->march)
        if (m.stream == stream) {
            m.stamp = ++c->march_clock;
            return &m;
        }
    if (c->march.size() < 16) {
        c->march.emplace_back();
    } else {
        size_t lru = 0;
        for (size_t i = 1; i < c->march.size(); ++i)
            if (c->march[i].stamp < c->march[lru].stamp) lru = i;
        sdfgpu_ctx::MarchScratch& old = c->march[lru];
        // wait for the entry's own event, not for a stream handle that may be dangling by now
        if (old.last_use) {
            if (hipEventSynchronize(old.last_use) != hipSuccess) (void)hipDeviceSynchronize();
            (void)hipEventDestroy(old.last_use);
        } else {
            (void)hipDeviceSynchronize();
        }
        for (DevBuf* b : {&old.buf, &old.order, &old.eye, &old.keys_a, &old.keys_b, &old.idx_a, &old.idx_b, &old.sort_tmp})
            if (b->p) (void)hipFree(b->p);
        old = sdfgpu_ctx::MarchScratch();
        std::swap(old, c->march.back());
    }
    sdfgpu_ctx::MarchScratch& m = c->march.back();
    m.stream = stream;
    m.stamp = ++c->march_clock;
    return &m;
}

// Records "this stream's scratch was last used here" (see MarchScratch::last_use); a no-op for streams without scratch.
void mark_used(sdfgpu_ctx* c, hipStream_t stream) {
    for (auto& m : c->march)
        if (m.stream == stream) {
            if (!m.last_use && hipEventCreateWithFlags(&m.last_use, hipEventDisableTiming) != hipSuccess) { m.last_use = nullptr; return; }
            (void)hipEventRecord(m.last_use, stream);
            return;
        }
}

int march_setup(sdfgpu_ctx* c, hipStream_t stream, uint32_t tile_count, uint32_t max_steps, RenderScratch* rs,
                std::vector<uint32_t>* bounds) {
    bounds->clear();
    bounds->push_back(0);
    if (c->compaction)
        for (uint32_t b : c->pass_steps)
            if (b > bounds->back() && b < max_steps) bounds->push_back(b);
    bounds->push_back(max_steps > 0 ? max_steps : 0);
    if (bounds->size() == 2 && (*bounds)[1] == 0) (*bounds)[1] = 0;   // zero steps: one empty pass
    size_t sb = 0, cb = 0;
    const size_t total = render_scratch_bytes(tile_count, &sb, &cb);
    if (bounds->size() == 2) return SDFGPU_OK;   // single pass: no pass state
    sdfgpu_ctx::MarchScratch* ms = stream_scratch(c, stream);
    DevBuf* mb = &ms->buf;
    int rc = reserve(c, *mb, total);
    if (rc) return rc;
    char* base = (char*)mb->p;
    rs->state[0] = base;
    rs->state[1] = base + sb;
    rs->counts = (uint32_t*)(base + 2 * sb);
    rs->offsets = (uint32_t*)(base + 2 * sb + cb);
    rs->hdr[0] = (uint32_t*)(base + 2 * sb + 2 * cb);
    rs->hdr[1] = rs->hdr[0] + 4;
    return SDFGPU_OK;
}

// One frame (or tile range) on `stream`: pass schedule, tile scheduling, launch.
// Tile scheduling (SDFGPU_OPT_TILE_ORDER): the single-pass march records how many steps every
// tile's wavefront ran; the next call on the SAME stream with the same frame geometry starts
// the longest tiles first, so that the few 128-step tiles run beside the bulk of the frame
// instead of after it.  A scheduling hint only: any order renders the same pixels.
int render_on(sdfgpu_ctx* c, hipStream_t stream, const sdfgpu_camera* cam, uint32_t w, uint32_t h,
              const sdfgpu_march_params* mp, uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count,
              bool row_major, uint8_t* d_hit, float* d_depth, float* d_normal, uint32_t* d_prim) {
    RenderScratch rs;
    std::vector<uint32_t> pb;
    int rc = march_setup(c, stream, tile_count, mp->max_steps, &rs, &pb);
    if (rc) return rc;
    sdfgpu_ctx::MarchScratch* ms = nullptr;
    if (!c->tree.gen && mp->t_min == 0.0f) {   // the walk from the eye, once per frame (kernels.hip, eye_walk_kernel)
        sdfgpu_ctx::MarchScratch* es = stream_scratch(c, stream);
        if ((rc = reserve(c, es->eye, 16))) return rc;
        rs.eye = (float*)es->eye.p;
    }
    uint32_t *cost = nullptr, *iota = nullptr, *keys = nullptr, *order = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    const bool single_pass = pb.size() == 2;
    if (c->tile_order && single_pass && !c->tree.gen && tile_count >= 1024 && (ms = stream_scratch(c, stream))) {
        const size_t arr = ((size_t)tile_count * sizeof(uint32_t) + 255) & ~(size_t)255;
        if (ms->order_tiles != tile_count || !ms->order.p) {
            ms->order_temp = tile_sort_temp_bytes(tile_count);
            if ((rc = reserve(c, ms->order, 4 * arr + ms->order_temp + 256))) return rc;
            ms->order_tiles = tile_count;
            ms->order_valid = false;
            HIP_TRY(c, launch_iota((uint32_t*)((char*)ms->order.p + arr), tile_count, stream));
        }
        char* base = (char*)ms->order.p;
        cost = (uint32_t*)base;
        iota = (uint32_t*)(base + arr);
        keys = (uint32_t*)(base + 2 * arr);
        order = (uint32_t*)(base + 3 * arr);
        temp = base + 4 * arr;
        temp_bytes = ms->order_temp;
        const uint32_t geom[6] = {w, h, tile_first, tile_stride, tile_count, mp->max_steps};
        if (std::memcmp(ms->order_geom, geom, sizeof geom) != 0) ms->order_valid = false;
        std::memcpy(ms->order_geom, geom, sizeof geom);
        rs.tile_cost = cost;
        rs.tile_order = ms->order_valid ? order : nullptr;
    }
    HIP_TRY(c, launch_render(c->tree, *cam, w, h, *mp, tile_first, tile_stride, tile_count, row_major, rs, pb.data(),
                             (uint32_t)pb.size(), d_hit, d_depth, d_normal, d_prim, nullptr, stream));
    if (cost) {
        // The order is a hint learned from the previous frame; sorting 32 k tiles behind EVERY frame cost 2-3 % of a
        // 1.3 ms frame (round 3: with two frames in flight "tile order off" had become the faster setting, while a
        // single frame still gains 14 % from the order: 1.31 against 1.52 ms).
        // So: a frame of the very view the order was learned from keeps it (re-sorted every 64th frame all the same);
        // any other view re-sorts from the costs it has just recorded, as before (a stale order under a moving
        // camera measured 7 % slower than a fresh one).
        constexpr uint32_t TILE_ORDER_REFRESH = 64;
        const bool same_view = std::memcmp(&ms->order_cam, cam, sizeof *cam) == 0;
        if (!ms->order_valid || !same_view || ++ms->order_age >= TILE_ORDER_REFRESH) {
            HIP_TRY(c, launch_tile_order(cost, iota, keys, order, tile_count, temp, temp_bytes, stream));
            ms->order_valid = true;
            ms->order_age = 0;
            ms->order_cam = *cam;
        }
    }
    mark_used(c, stream);
    return SDFGPU_OK;
}

// Environment overrides of the candidate grid's sizing (measurements; read once).
double grid_env(const char* name, double dflt) {
    const char* e = std::getenv(name);
    return e && *e ? std::atof(e) : dflt;
}

// The nearest-candidate grid of the tree just uploaded (kernels.hpp CandidateGrid, grid.hip).  Geometry: the scene
// box padded by an eighth of its largest extent on every side (rays enter and leave the scene through that margin,
// and walks out there are the widest ones), cut into cells of equal size, ~400 per primitive (at the bench scenes'
// density a cell is then about a fifth of the primitives' spacing: 2-3 candidates per cell), cells per axis a
// multiple of 4 (bricks).  A failure to allocate is not an error: the tree simply has no grid.
int build_grid(sdfgpu_ctx* c, const CompactTree& t) {
    static const bool env_on = grid_env("SDFGPU_GRID", 1.0) != 0.0;
    static const double per_prim = grid_env("SDFGPU_GRID_CELLS_PER_PRIM", 400.0);
    static const double max_cells = grid_env("SDFGPU_GRID_MAX_CELLS", 32.0 * 1024 * 1024);
    static const double pad_frac = grid_env("SDFGPU_GRID_PAD", 0.125);
    constexpr uint32_t GRID_MIN_PRIMS = 64;
    c->info.grid_cells = 0;
    if (!env_on || !c->candidate_grid) return SDFGPU_OK;
    const uint32_t b = t.info.bounds;
    // (every pruned mode: what a list proves is about the primitives, whichever boxes the walk that built it used)
    if (!t.gen.empty() || (b != SDFGPU_BOUNDS_AABB && b != SDFGPU_BOUNDS_BVH && b != SDFGPU_BOUNDS_OBB) || t.leaves.size() < GRID_MIN_PRIMS) return SDFGPU_OK;
    if (!(t.lipschitz < 1.001f)) return SDFGPU_OK;   // rigid leaves only (what the pruned modes accept anyway)
    float ext[3], maxext = 0.f;
    for (int a = 0; a < 3; ++a) {
        ext[a] = t.info.scene_hi[a] - t.info.scene_lo[a];
        if (!(ext[a] >= 0.f) || !(ext[a] < 1e30f)) return SDFGPU_OK;
        maxext = std::fmax(maxext, ext[a]);
    }
    if (!(maxext > 0.f)) return SDFGPU_OK;
    const float pad = (float)pad_frac * maxext;
    double vol = 1.0;
    float lo[3], len[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = t.info.scene_lo[a] - pad;
        len[a] = ext[a] + 2.0f * pad;
        vol *= (double)len[a];
    }
    const double target = std::fmin(max_cells, per_prim * (double)t.leaves.size());
    const double side = std::cbrt(vol / target);
    CandidateGrid g;
    uint64_t cells = 1;
    float diag2 = 0.f, p1max = 0.f;
    for (int a = 0; a < 3; ++a) {
        double n = std::ceil((double)len[a] / side / 4.0) * 4.0;
        n = std::fmin(std::fmax(n, 4.0), 2048.0);
        g.n[a] = (uint32_t)n;
        const float cell = len[a] / (float)g.n[a];
        g.lo[a] = lo[a];
        g.inv_cell[a] = 1.0f / cell;
        if (!(g.inv_cell[a] > 0.f) || !(g.inv_cell[a] < 1e30f)) return SDFGPU_OK;
        diag2 += cell * cell;
        cells *= g.n[a];
        p1max += std::fmax(std::fabs(lo[a]), std::fabs(lo[a] + len[a]));
    }
    if (cells > (uint64_t)(2.0 * max_cells) || cells >= (1ull << 31)) return SDFGPU_OK;
    const uint64_t bricks = cells / 64u;
    const float radius = 0.5f * std::sqrt(diag2) * 1.01f;   // + 1 %: a point's cell index is formed in fp32
    // rounding of a leaf value anywhere in the grid (the prune margin of kernels.hip, at the largest |p|_1), times 8
    const float e_max = 3.0e-6f * t.lipschitz * (1.0f + (float)t.chain_max) * (p1max + t.margin_T);
    const float margin = 8.0f * e_max;
    const size_t cells_bytes = (cells * sizeof(GridCell) + 63u) & ~(size_t)63u;
    const size_t ids_bytes = bricks * GRID_BRICK_IDS * sizeof(uint32_t);
    const size_t caps_bytes = cells * sizeof(GridCap);
    g.bytes = sizeof(GridBlock) + cells_bytes + ids_bytes + caps_bytes;
    if (g.bytes >= (1ull << 32)) return SDFGPU_OK;   // the march addresses the tables with 32-bit offsets
    if (hipMalloc(&g.block, g.bytes) != hipSuccess) { (void)hipGetLastError(); return SDFGPU_OK; }
    g.cells = (GridCell*)((char*)g.block + sizeof(GridBlock));
    g.brick_ids = (uint32_t*)((char*)g.block + sizeof(GridBlock) + cells_bytes);
    g.caps = (GridCap*)((char*)g.block + sizeof(GridBlock) + cells_bytes + ids_bytes);
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(bricks, 16384);   // 1 M cells per round: 144 MB of centres and rows
    void* temp = nullptr;
    if (hipMalloc(&temp, grid_temp_bytes(chunk)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(g.block);
        return SDFGPU_OK;
    }
    GridBlock gb{};
    for (int a = 0; a < 3; ++a) { gb.lo[a] = g.lo[a]; gb.inv_cell[a] = g.inv_cell[a]; gb.n[a] = g.n[a]; gb.cell[a] = 1.0f / g.inv_cell[a]; }
    gb.nbx = g.n[0] / 4u;
    gb.nby = g.n[1] / 4u;
    gb.bricks_off = (uint32_t)(sizeof(GridBlock) + cells_bytes);
    gb.caps_off = (uint32_t)(sizeof(GridBlock) + cells_bytes + ids_bytes);
    if (hipMemcpy(g.block, &gb, sizeof gb, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(temp);
        (void)hipFree(g.block);
        return SDFGPU_OK;
    }
    c->tree.grid = g;
    hipError_t e = build_candidate_grid(c->tree, radius, margin, temp, chunk, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(temp);
    if (e != hipSuccess) {
        (void)hipFree(g.block);
        c->tree.grid = CandidateGrid();
        return hip_fail(c, e, "candidate grid build");
    }
    c->info.grid_cells = (uint32_t)cells;
    c->info.device_bytes += g.bytes;
    return SDFGPU_OK;
}

template <class T>
int upload_vec(sdfgpu_ctx* c, const std::vector<T>& v, T** dst) {
    *dst = nullptr;
    if (v.empty()) return SDFGPU_OK;
    HIP_TRY(c, hipMalloc((void**)dst, v.size() * sizeof(T)));
    HIP_TRY(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return SDFGPU_OK;
}

}  // namespace

namespace sdfgpu {
int ctx_device(const sdfgpu_ctx* c) { return c->device; }
int ctx_set_error(sdfgpu_ctx* c, int code, const char* msg) { return set_err(c, code, msg ? msg : ""); }
}  // namespace sdfgpu

extern "C" {

const char* sdfgpu_status_string(int s) {
    switch (s) {
        case SDFGPU_OK: return "ok";
        case SDFGPU_ERR_INVALID_ARG: return "invalid argument";
        case SDFGPU_ERR_HIP: return "HIP runtime error";
        case SDFGPU_ERR_NO_TREE: return "no tree uploaded";
        case SDFGPU_ERR_LAYOUT: return "unsupported record layout";
        case SDFGPU_ERR_BAD_TREE: return "malformed tree buffer";
        case SDFGPU_ERR_UNSUPPORTED_OP: return "unsupported op code";
        case SDFGPU_ERR_TOO_DEEP: return "tree too deep";
        case SDFGPU_ERR_NON_FINITE: return "non-finite value in tree";
        case SDFGPU_ERR_POINT_W: return "query point with w != 1";
        case SDFGPU_ERR_TOO_LARGE: return "size limit exceeded";
        case SDFGPU_ERR_COMM: return "RCCL communicator error";
    }
    return "unknown status";
}

int sdfgpu_create(int device_id, sdfgpu_ctx** out) {
    if (!out) return SDFGPU_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return SDFGPU_ERR_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return SDFGPU_ERR_HIP;
    sdfgpu_ctx* c = new sdfgpu_ctx();
    c->device = device_id;
    c->pass_steps = default_pass_steps();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SDFGPU_ERR_HIP;
    }
    // device address of the allocation-free entry point's scratch (one copy per GPU), once, here,
    // with this device current; a failure is kept and reported by sdfgpu_nearest_dev_static
    c->static_scratch_err = static_scratch_address(device_id, &c->static_scratch);
    *out = c;
    return SDFGPU_OK;
}

void sdfgpu_destroy(sdfgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_tree(c);
    for (DevBuf* b : {&c->in, &c->out_a, &c->out_b, &c->out_c, &c->out_d, &c->stats})
        if (b->p) (void)hipFree(b->p);
    if (c->aux) {
        (void)hipStreamSynchronize(c->aux);
        for (auto& ps : c->pipe) {
            for (DevBuf* b : {&ps.keys_a, &ps.keys_b, &ps.idx_a, &ps.idx_b, &ps.tmp})
                if (b->p) (void)hipFree(b->p);
            if (ps.sorted) (void)hipEventDestroy(ps.sorted);
            if (ps.walked) (void)hipEventDestroy(ps.walked);
        }
        if (c->ev_start) (void)hipEventDestroy(c->ev_start);
        (void)hipStreamDestroy(c->aux);
    }
    for (auto& m : c->march) {
        for (DevBuf* b : {&m.buf, &m.order, &m.eye, &m.keys_a, &m.keys_b, &m.idx_a, &m.idx_b, &m.sort_tmp})
            if (b->p) (void)hipFree(b->p);
        if (m.last_use) (void)hipEventDestroy(m.last_use);
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* sdfgpu_last_error(const sdfgpu_ctx* c) { return c ? c->err.c_str() : "null context"; }

int sdfgpu_sync(sdfgpu_ctx* c) {
    if (!c) return SDFGPU_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

// Host-only half of upload: validation + compaction, no GPU needed.
int sdfgpu_compact_host_ext(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t buffer_len,
                            uint32_t layout, uint32_t bounds, uint32_t extensions, sdfgpu_tree_info* info, char* errbuf,
                            size_t errlen) {
    CompactTree t;
    std::string err;
    int rc = compact_tree(dt, ut, buffer_len, layout, bounds, extensions, &t, &err);
    if (info) *info = t.info;
    if (errbuf && errlen) {
        std::snprintf(errbuf, errlen, "%s", err.c_str());
    }
    return rc;
}

int sdfgpu_compact_host(const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut, uint32_t buffer_len,
                        uint32_t layout, uint32_t bounds, sdfgpu_tree_info* info, char* errbuf, size_t errlen) {
    return sdfgpu_compact_host_ext(dt, ut, buffer_len, layout, bounds, 0u, info, errbuf, errlen);
}

int sdfgpu_upload_tree_ext(sdfgpu_ctx* c, const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut,
                           uint32_t buffer_len, uint32_t layout, uint32_t bounds, uint32_t extensions) {
    if (!c) return SDFGPU_ERR_INVALID_ARG;
    CompactTree t;
    std::string err;
    int rc = compact_tree(dt, ut, buffer_len, layout, bounds, extensions, &t, &err);
    if (rc != SDFGPU_OK) return set_err(c, rc, err);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_tree(c);
    if ((rc = upload_vec(c, t.leaves, &c->tree.leaf)) != SDFGPU_OK) return rc;
    if ((rc = upload_vec(c, t.inner, &c->tree.inner)) != SDFGPU_OK) return rc;
    if ((rc = upload_vec(c, t.inner_obb, &c->tree.inner_obb)) != SDFGPU_OK) return rc;
    if ((rc = upload_vec(c, t.gen, &c->tree.gen)) != SDFGPU_OK) return rc;
    if (!t.chain_mats.empty()) {   // chain path: [per-primitive {first, count} | pad to 64 | chain records], one allocation
        const size_t head = (t.leaf_chain.size() * sizeof(uint32_t) + 63u) & ~(size_t)63u;
        const size_t bytes = head + t.chain_mats.size() * sizeof(GpuChainMat);
        HIP_TRY(c, hipMalloc(&c->tree.chain, bytes));
        HIP_TRY(c, hipMemcpy(c->tree.chain, t.leaf_chain.data(), t.leaf_chain.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy((char*)c->tree.chain + head, t.chain_mats.data(), t.chain_mats.size() * sizeof(GpuChainMat), hipMemcpyHostToDevice));
        c->tree.chain_max = t.chain_max;
    }
    c->tree.n_gen = (uint32_t)t.gen.size();
    c->tree.gen_eval = t.gen_eval;
    c->tree.gen_prune = t.gen_prune;
    c->tree.gen_pt_levels = t.gen_pt_levels;
    c->tree.root_ref = t.root_ref;
    c->tree.n_leaf = (uint32_t)t.leaves.size();
    c->tree.stack_depth = t.stack_depth;
    c->tree.bounds = t.info.bounds;
    c->tree.margin_T = t.margin_T;
    c->tree.lipschitz = t.lipschitz;
    c->tree.box_scale = t.box_scale;
    c->mean_extent = t.mean_extent;
    c->tree.skip_ext = c->skip_ext_opt < 0.f ? 0.5f * t.mean_extent : c->skip_ext_opt;
    c->info = t.info;
    c->has_tree = true;
    c->err.clear();
    return build_grid(c, t);
}

int sdfgpu_upload_tree(sdfgpu_ctx* c, const SdfOperationBlock* dt, const SdfOperationUptreeBlock* ut,
                       uint32_t buffer_len, uint32_t layout, uint32_t bounds) {
    return sdfgpu_upload_tree_ext(c, dt, ut, buffer_len, layout, bounds, 0u);
}

int sdfgpu_tree_info_get(const sdfgpu_ctx* c, sdfgpu_tree_info* out) {
    if (!c || !out) return SDFGPU_ERR_INVALID_ARG;
    if (!c->has_tree) return SDFGPU_ERR_NO_TREE;
    *out = c->info;
    return SDFGPU_OK;
}

int sdfgpu_nearest_dev(sdfgpu_ctx* c, const float* d_pts, size_t n, float* d_dist, uint32_t* d_prim, void* stream) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (n == 0) return SDFGPU_OK;
    if (!d_pts || !d_dist) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null device pointer");
    if (n > 0x7FFFFFFFull) return set_err(c, SDFGPU_ERR_TOO_LARGE, "more than 2^31-1 points in one call");
    return ordered_query(c, d_pts, n, (hipStream_t)stream, [&](const float* p, const uint32_t* order, size_t m, size_t first) -> int {
        HIP_TRY(c, launch_nearest(c->tree, p, order, (uint32_t)m, d_dist + first, d_prim ? d_prim + first : nullptr, nullptr,
                                  (hipStream_t)stream));
        return SDFGPU_OK;
    });
}

// The allocation-free variant of sdfgpu_nearest_dev: same kernels, same results; the reordering
// scratch is sort.hip's file-scope __device__ array instead of hipMalloc'ed context buffers.
// Nothing on this path calls hipMalloc / hipFree (launch_morton_order hands hipcub its storage).
int sdfgpu_nearest_dev_static(sdfgpu_ctx* c, const float* d_pts, size_t n, float* d_dist, uint32_t* d_prim, void* stream) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (n == 0) return SDFGPU_OK;
    if (!d_pts || !d_dist) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null device pointer");
    if (n > SDFGPU_STATIC_MAX_POINTS)
        return set_err(c, SDFGPU_ERR_TOO_LARGE,
                       "sdfgpu_nearest_dev_static takes at most SDFGPU_STATIC_MAX_POINTS (262144) points per call");
    if (c->static_scratch_err != hipSuccess || !c->static_scratch)
        return hip_fail(c, c->static_scratch_err != hipSuccess ? c->static_scratch_err : hipErrorInvalidSymbol,
                        "hipGetSymbolAddress(scratch) at sdfgpu_create");
    const uint32_t* order = nullptr;
    if ((rc = make_order(c, d_pts, n, (hipStream_t)stream, &order, true))) return rc;
    HIP_TRY(c, launch_nearest(c->tree, d_pts, order, (uint32_t)n, d_dist, d_prim, nullptr, (hipStream_t)stream));
    return SDFGPU_OK;
}

int sdfgpu_nearest(sdfgpu_ctx* c, const float* pts, size_t n, float* out_dist, uint32_t* out_prim) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (n == 0) return SDFGPU_OK;
    if (!pts || !out_dist) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null pointer");
    for (size_t i = 0; i < n; ++i)
        if (pts[4 * i + 3] != 1.0f) {
            char b[96];
            std::snprintf(b, sizeof b, "point %zu has w = %g (the path is defined for w = 1)", i, (double)pts[4 * i + 3]);
            return set_err(c, SDFGPU_ERR_POINT_W, b);
        }
    if ((rc = reserve(c, c->in, n * 16)) || (rc = reserve(c, c->out_a, n * 4)) || (rc = reserve(c, c->out_b, n * 4))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->in.p, pts, n * 16, hipMemcpyHostToDevice, c->stream));
    rc = sdfgpu_nearest_dev(c, (const float*)c->in.p, n, (float*)c->out_a.p, (uint32_t*)c->out_b.p, c->stream);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_dist, c->out_a.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_prim) HIP_TRY(c, hipMemcpyAsync(out_prim, c->out_b.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

int sdfgpu_nearest_stats_dev(sdfgpu_ctx* c, const float* d_pts, size_t n, sdfgpu_stats* out) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (!out) return SDFGPU_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    if (n == 0) return SDFGPU_OK;
    if (n > 0x7FFFFFFFull) return set_err(c, SDFGPU_ERR_TOO_LARGE, "more than 2^31-1 points in one call");
    if ((rc = reserve(c, c->stats, sizeof(sdfgpu_stats))) || (rc = reserve(c, c->out_a, n * 4)) || (rc = reserve(c, c->out_b, n * 4))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->stats.p, 0, sizeof(sdfgpu_stats), c->stream));
    const uint32_t* order = nullptr;
    if ((rc = make_order(c, d_pts, n, c->stream, &order))) return rc;
    HIP_TRY(c, launch_nearest(c->tree, d_pts, order, (uint32_t)n, (float*)c->out_a.p, (uint32_t*)c->out_b.p,
                              (sdfgpu_stats*)c->stats.p, c->stream, c->stats_asm_walk));
    HIP_TRY(c, hipMemcpyAsync(out, c->stats.p, sizeof(sdfgpu_stats), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

int sdfgpu_knn_stats_dev(sdfgpu_ctx* c, const float* d_pts, size_t n, uint32_t k, sdfgpu_stats* out) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (!out) return SDFGPU_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    if (k == 0 || k > SDFGPU_MAX_K) return set_err(c, SDFGPU_ERR_TOO_LARGE, "k must be in 1..SDFGPU_MAX_K (16)");
    if (n == 0) return SDFGPU_OK;
    if (n > 0x7FFFFFFFull / k) return set_err(c, SDFGPU_ERR_TOO_LARGE, "n * k exceeds 2^31-1 in one call");
    if ((rc = reserve(c, c->stats, sizeof(sdfgpu_stats))) || (rc = reserve(c, c->out_a, n * k * 4)) ||
        (rc = reserve(c, c->out_b, n * k * 4)))
        return rc;
    HIP_TRY(c, hipMemsetAsync(c->stats.p, 0, sizeof(sdfgpu_stats), c->stream));
    const uint32_t* order = nullptr;
    if ((rc = make_order(c, d_pts, n, c->stream, &order))) return rc;
    HIP_TRY(c, launch_knn(c->tree, d_pts, order, (uint32_t)n, k, (float*)c->out_a.p, (uint32_t*)c->out_b.p,
                          (sdfgpu_stats*)c->stats.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->stats.p, sizeof(sdfgpu_stats), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

int sdfgpu_set_option(sdfgpu_ctx* c, int option, double value) {
    if (!c) return SDFGPU_ERR_INVALID_ARG;
    switch (option) {
        case SDFGPU_OPT_REORDER:
            c->reorder = value != 0.0;
            return SDFGPU_OK;
        case SDFGPU_OPT_SKIP_EXTENSION:
            if (!(value == value)) return set_err(c, SDFGPU_ERR_INVALID_ARG, "skip extension is NaN");
            c->skip_ext_opt = (float)value;
            c->tree.skip_ext = value < 0.0 ? 0.5f * c->mean_extent : (float)value;
            return SDFGPU_OK;
        case SDFGPU_OPT_COMPACTION:
            c->compaction = value != 0.0;
            return SDFGPU_OK;
        case SDFGPU_OPT_QUERY_PIPELINE:
            if (!(value >= 0.0 && value <= 16.0)) return set_err(c, SDFGPU_ERR_INVALID_ARG, "query pipeline chunks must be in 0..16");
            c->query_chunks = (uint32_t)value;
            return SDFGPU_OK;
        case SDFGPU_OPT_STATS_ASM_WALK:
            c->stats_asm_walk = value != 0.0;
            return SDFGPU_OK;
        case SDFGPU_OPT_CANDIDATE_GRID:
            c->candidate_grid = value != 0.0;
            return SDFGPU_OK;
        case SDFGPU_OPT_TILE_ORDER:
            c->tile_order = value != 0.0;
            for (auto& m : c->march) m.order_valid = false;
            return SDFGPU_OK;
    }
    return set_err(c, SDFGPU_ERR_INVALID_ARG, "unknown option");
}

int sdfgpu_knn_dev(sdfgpu_ctx* c, const float* d_pts, size_t n, uint32_t k, float* d_dist, uint32_t* d_prim,
                   void* stream) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (k == 0 || k > SDFGPU_MAX_K) return set_err(c, SDFGPU_ERR_TOO_LARGE, "k must be in 1..SDFGPU_MAX_K (16)");
    if (n == 0) return SDFGPU_OK;
    if (!d_pts || !d_dist || !d_prim) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null device pointer");
    if (n > 0x7FFFFFFFull / k) return set_err(c, SDFGPU_ERR_TOO_LARGE, "n * k exceeds 2^31-1 in one call");
    return ordered_query(c, d_pts, n, (hipStream_t)stream, [&](const float* p, const uint32_t* order, size_t m, size_t first) -> int {
        HIP_TRY(c, launch_knn(c->tree, p, order, (uint32_t)m, k, d_dist + first * k, d_prim + first * k, nullptr, (hipStream_t)stream));
        return SDFGPU_OK;
    });
}

int sdfgpu_knn(sdfgpu_ctx* c, const float* pts, size_t n, uint32_t k, float* out_dist, uint32_t* out_prim) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (k == 0 || k > SDFGPU_MAX_K) return set_err(c, SDFGPU_ERR_TOO_LARGE, "k must be in 1..SDFGPU_MAX_K (16)");
    if (n == 0) return SDFGPU_OK;
    if (!pts || !out_dist || !out_prim) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null pointer");
    for (size_t i = 0; i < n; ++i)
        if (pts[4 * i + 3] != 1.0f) return set_err(c, SDFGPU_ERR_POINT_W, "a query point has w != 1");
    if ((rc = reserve(c, c->in, n * 16)) || (rc = reserve(c, c->out_a, n * k * 4)) || (rc = reserve(c, c->out_b, n * k * 4)))
        return rc;
    HIP_TRY(c, hipMemcpyAsync(c->in.p, pts, n * 16, hipMemcpyHostToDevice, c->stream));
    rc = sdfgpu_knn_dev(c, (const float*)c->in.p, n, k, (float*)c->out_a.p, (uint32_t*)c->out_b.p, c->stream);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_dist, c->out_a.p, n * k * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_prim, c->out_b.p, n * k * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

void sdfgpu_query_slice(size_t n, uint32_t rank, uint32_t world, size_t* first, size_t* count) {
    size_t lo = 0, cnt = 0;
    if (world != 0 && rank < world) {
        const size_t base = n / world, rem = n % world;
        lo = (size_t)rank * base + (rank < rem ? rank : rem);
        cnt = base + (rank < rem ? 1u : 0u);
    }
    if (first) *first = lo;
    if (count) *count = cnt;
}

// One host batch over several contexts (k == 0: the k = 1 query without rows).  Every context's slice is
// queued on that context's own stream before any is waited for, so the devices work side by side.
static int query_sharded(sdfgpu_ctx* const* ctxs, int n_ctx, const float* pts, size_t n, uint32_t k, float* out_dist,
                         uint32_t* out_prim) {
    if (!ctxs || n_ctx < 1) return SDFGPU_ERR_INVALID_ARG;
    for (int i = 0; i < n_ctx; ++i)
        if (!ctxs[i]) return SDFGPU_ERR_INVALID_ARG;
    sdfgpu_ctx* c0 = ctxs[0];
    const bool knn = k != 0;
    const uint32_t kk = knn ? k : 1u;
    if (knn && k > SDFGPU_MAX_K) return set_err(c0, SDFGPU_ERR_TOO_LARGE, "k must be in 1..SDFGPU_MAX_K (16)");
    for (int i = 0; i < n_ctx; ++i) {
        int rc = check_ready(ctxs[i]);
        if (rc) return rc;
    }
    if (n == 0) return SDFGPU_OK;
    if (!pts || !out_dist || (knn && !out_prim)) return set_err(c0, SDFGPU_ERR_INVALID_ARG, "null pointer");
    for (size_t i = 0; i < n; ++i)
        if (pts[4 * i + 3] != 1.0f) return set_err(c0, SDFGPU_ERR_POINT_W, "a query point has w != 1");
    int first_rc = SDFGPU_OK;
    for (int i = 0; i < n_ctx; ++i) {
        sdfgpu_ctx* c = ctxs[i];
        size_t lo, m;
        sdfgpu_query_slice(n, (uint32_t)i, (uint32_t)n_ctx, &lo, &m);
        if (m == 0) continue;
        int rc = [&]() -> int {
            HIP_TRY(c, hipSetDevice(c->device));
            int r;
            if ((r = reserve(c, c->in, m * 16)) || (r = reserve(c, c->out_a, m * kk * 4)) || (r = reserve(c, c->out_b, m * kk * 4))) return r;
            HIP_TRY(c, hipMemcpyAsync(c->in.p, pts + 4 * lo, m * 16, hipMemcpyHostToDevice, c->stream));
            r = knn ? sdfgpu_knn_dev(c, (const float*)c->in.p, m, k, (float*)c->out_a.p, (uint32_t*)c->out_b.p, c->stream)
                    : sdfgpu_nearest_dev(c, (const float*)c->in.p, m, (float*)c->out_a.p, (uint32_t*)c->out_b.p, c->stream);
            if (r) return r;
            HIP_TRY(c, hipMemcpyAsync(out_dist + lo * kk, c->out_a.p, m * kk * 4, hipMemcpyDeviceToHost, c->stream));
            if (out_prim) HIP_TRY(c, hipMemcpyAsync(out_prim + lo * kk, c->out_b.p, m * kk * 4, hipMemcpyDeviceToHost, c->stream));
            return SDFGPU_OK;
        }();
        if (rc && !first_rc) {
            first_rc = rc;
            if (c != c0) c0->err = c->err;   // the caller reads the message from the first context
            break;
        }
    }
    for (int i = 0; i < n_ctx; ++i) {   // wait for whatever was queued, also after a failure
        sdfgpu_ctx* c = ctxs[i];
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            if (!first_rc) first_rc = set_err(c0, SDFGPU_ERR_HIP, "hipStreamSynchronize (sharded query)");
        }
    }
    return first_rc;
}

int sdfgpu_nearest_sharded(sdfgpu_ctx* const* ctxs, int n_ctx, const float* pts, size_t n, float* out_dist, uint32_t* out_prim) {
    return query_sharded(ctxs, n_ctx, pts, n, 0u, out_dist, out_prim);
}

int sdfgpu_knn_sharded(sdfgpu_ctx* const* ctxs, int n_ctx, const float* pts, size_t n, uint32_t k, float* out_dist,
                       uint32_t* out_prim) {
    if (k == 0) return ctxs && n_ctx > 0 && ctxs[0] ? set_err(ctxs[0], SDFGPU_ERR_TOO_LARGE, "k must be in 1..SDFGPU_MAX_K (16)") : SDFGPU_ERR_INVALID_ARG;
    return query_sharded(ctxs, n_ctx, pts, n, k, out_dist, out_prim);
}

int sdfgpu_box_bounds(sdfgpu_ctx* c, const SdfBoundingBoxBlock* box, const float* pts, size_t n, float* out_min,
                      float* out_max) {
    if (!c) return SDFGPU_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n == 0) return SDFGPU_OK;
    if (!box || !pts || !out_min || !out_max) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null pointer");
    if (n > 0x7FFFFFFFull) return set_err(c, SDFGPU_ERR_TOO_LARGE, "more than 2^31-1 points in one call");
    int rc;
    if ((rc = reserve(c, c->in, n * 16)) || (rc = reserve(c, c->out_a, n * 4)) || (rc = reserve(c, c->out_b, n * 4)) ||
        (rc = reserve(c, c->stats, sizeof(SdfBoundingBoxBlock))))
        return rc;
    HIP_TRY(c, hipMemcpyAsync(c->in.p, pts, n * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->stats.p, box, sizeof(SdfBoundingBoxBlock), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, launch_box_bounds((const SdfBoundingBoxBlock*)c->stats.p, (const float*)c->in.p, (uint32_t)n,
                                 (float*)c->out_a.p, (float*)c->out_b.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_min, c->out_a.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_max, c->out_b.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

int sdfgpu_render_tiles_dev(sdfgpu_ctx* c, const sdfgpu_camera* cam, uint32_t w, uint32_t h,
                            const sdfgpu_march_params* mp, uint32_t tile_first, uint32_t tile_stride,
                            uint32_t tile_count, uint8_t* d_hit, float* d_depth, float* d_normal, uint32_t* d_prim,
                            void* stream) {
    int rc = check_ready(c);
    if (rc) return rc;
    if ((rc = check_march(c, cam, w, h, mp))) return rc;
    if (tile_count == 0) return SDFGPU_OK;
    if (!d_hit) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null hit buffer");
    const uint64_t tiles = (uint64_t)((w + 7) / 8) * ((h + 7) / 8);
    if (tile_stride == 0 || (uint64_t)tile_first + (uint64_t)(tile_count - 1) * tile_stride >= tiles)
        return set_err(c, SDFGPU_ERR_INVALID_ARG, "tile range outside the frame");
    return render_on(c, (hipStream_t)stream, cam, w, h, mp, tile_first, tile_stride, tile_count, false, d_hit, d_depth,
                     d_normal, d_prim);
}

int sdfgpu_render_dev(sdfgpu_ctx* c, const sdfgpu_camera* cam, uint32_t w, uint32_t h, const sdfgpu_march_params* mp,
                      uint8_t* d_hit, float* d_depth, float* d_normal, uint32_t* d_prim, void* stream) {
    int rc = check_ready(c);
    if (rc) return rc;
    if ((rc = check_march(c, cam, w, h, mp))) return rc;
    if (!d_hit) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null hit buffer");
    const uint32_t tiles = ((w + 7) / 8) * ((h + 7) / 8);
    return render_on(c, (hipStream_t)stream, cam, w, h, mp, 0, 1, tiles, true, d_hit, d_depth, d_normal, d_prim);
}

int sdfgpu_untile_dev(sdfgpu_ctx* c, uint32_t w, uint32_t h, uint32_t tile_first, uint32_t tile_stride,
                      uint32_t tile_count, const uint8_t* t_hit, const float* t_depth, const float* t_normal,
                      const uint32_t* t_prim, uint8_t* d_hit, float* d_depth, float* d_normal, uint32_t* d_prim,
                      void* stream) {
    if (!c) return SDFGPU_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (w == 0 || h == 0 || tile_count == 0) return SDFGPU_OK;
    if (!t_hit || !d_hit) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null hit buffer");
    if ((d_depth && !t_depth) || (d_normal && !t_normal) || (d_prim && !t_prim))
        return set_err(c, SDFGPU_ERR_INVALID_ARG, "destination plane without a source plane");
    const uint64_t tiles = (uint64_t)((w + 7) / 8) * ((h + 7) / 8);
    if (tile_stride == 0 || (uint64_t)tile_first + (uint64_t)(tile_count - 1) * tile_stride >= tiles)
        return set_err(c, SDFGPU_ERR_INVALID_ARG, "tile range outside the frame");
    HIP_TRY(c, launch_untile(w, h, tile_first, tile_stride, tile_count, t_hit, t_depth, t_normal, t_prim, d_hit,
                             d_depth, d_normal, d_prim, (hipStream_t)stream));
    return SDFGPU_OK;
}

size_t sdfgpu_packed_share_bytes(uint32_t w, uint32_t h, uint32_t world) {
    if (world == 0) return 0;
    const uint64_t tiles = (uint64_t)((w + 7) / 8) * ((h + 7) / 8);
    return (size_t)((tiles + world - 1) / world) * 64u * 21u;   // depth 4 + normal 12 + prim 4 + hit 1 bytes per pixel
}

int sdfgpu_untile_gathered_dev(sdfgpu_ctx* c, uint32_t w, uint32_t h, uint32_t world, uint32_t self, const void* d_gathered,
                               const void* d_mine, uint8_t* d_hit, float* d_depth, float* d_normal, uint32_t* d_prim,
                               void* stream) {
    if (!c) return SDFGPU_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (w == 0 || h == 0) return SDFGPU_OK;
    if (world == 0 || self >= world) return set_err(c, SDFGPU_ERR_INVALID_ARG, "world / self out of range");
    if ((uint64_t)w * h > (1ull << 30)) return set_err(c, SDFGPU_ERR_TOO_LARGE, "image larger than 2^30 pixels");
    if (!d_hit || !d_mine || (world > 1 && !d_gathered)) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null buffer");
    const size_t nbytes = sdfgpu_packed_share_bytes(w, h, world);
    HIP_TRY(c, launch_untile_gathered(w, h, world, self, nbytes / 21u, nbytes, d_gathered, d_mine, d_hit, d_depth, d_normal, d_prim,
                                      (hipStream_t)stream));
    return SDFGPU_OK;
}

int sdfgpu_render(sdfgpu_ctx* c, const sdfgpu_camera* cam, uint32_t w, uint32_t h, const sdfgpu_march_params* mp,
                  uint8_t* hit, float* depth, float* normal, uint32_t* prim) {
    int rc = check_ready(c);
    if (rc) return rc;
    if ((rc = check_march(c, cam, w, h, mp))) return rc;
    if (!hit) return set_err(c, SDFGPU_ERR_INVALID_ARG, "null hit buffer");
    const size_t np = (size_t)w * h;
    if ((rc = reserve(c, c->out_a, np)) || (rc = reserve(c, c->out_b, np * 4)) || (rc = reserve(c, c->out_c, np * 12)) ||
        (rc = reserve(c, c->out_d, np * 4)))
        return rc;
    const uint32_t tiles = ((w + 7) / 8) * ((h + 7) / 8);
    if ((rc = render_on(c, c->stream, cam, w, h, mp, 0, 1, tiles, true, (uint8_t*)c->out_a.p, (float*)c->out_b.p,
                        (float*)c->out_c.p, (uint32_t*)c->out_d.p)))
        return rc;
    HIP_TRY(c, hipMemcpyAsync(hit, c->out_a.p, np, hipMemcpyDeviceToHost, c->stream));
    if (depth) HIP_TRY(c, hipMemcpyAsync(depth, c->out_b.p, np * 4, hipMemcpyDeviceToHost, c->stream));
    if (normal) HIP_TRY(c, hipMemcpyAsync(normal, c->out_c.p, np * 12, hipMemcpyDeviceToHost, c->stream));
    if (prim) HIP_TRY(c, hipMemcpyAsync(prim, c->out_d.p, np * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

int sdfgpu_render_stats(sdfgpu_ctx* c, const sdfgpu_camera* cam, uint32_t w, uint32_t h,
                        const sdfgpu_march_params* mp, sdfgpu_stats* out) {
    int rc = check_ready(c);
    if (rc) return rc;
    if ((rc = check_march(c, cam, w, h, mp))) return rc;
    if (!out) return SDFGPU_ERR_INVALID_ARG;
    const size_t np = (size_t)w * h;
    if ((rc = reserve(c, c->out_a, np)) || (rc = reserve(c, c->stats, sizeof(sdfgpu_stats)))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->stats.p, 0, sizeof(sdfgpu_stats), c->stream));
    const uint32_t tiles = ((w + 7) / 8) * ((h + 7) / 8);
    RenderScratch rs;
    std::vector<uint32_t> pb;
    if ((rc = march_setup(c, c->stream, tiles, mp->max_steps, &rs, &pb))) return rc;
    if (!c->tree.gen && mp->t_min == 0.0f) {   // count what the timed kernel does: one walk from the eye per frame
        sdfgpu_ctx::MarchScratch* es = stream_scratch(c, c->stream);
        if ((rc = reserve(c, es->eye, 16))) return rc;
        rs.eye = (float*)es->eye.p;
    }
    HIP_TRY(c, launch_render(c->tree, *cam, w, h, *mp, 0, 1, tiles, true, rs, pb.data(), (uint32_t)pb.size(),
                             (uint8_t*)c->out_a.p, nullptr, nullptr, nullptr, (sdfgpu_stats*)c->stats.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->stats.p, sizeof(sdfgpu_stats), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SDFGPU_OK;
}

void sdfgpu_camera_look_at(const float eye[3], const float target[3], const float up[3], float fov_y, uint32_t width,
                           uint32_t height, sdfgpu_camera* out) {
    double f[3], r[3], u[3];
    double fl = 0, rl = 0;
    for (int i = 0; i < 3; ++i) { f[i] = (double)target[i] - (double)eye[i]; fl += f[i] * f[i]; }
    fl = std::sqrt(fl);
    for (int i = 0; i < 3; ++i) f[i] /= fl;
    // right = f x up, true up = right x f  (right-handed, image y grows downward)
    r[0] = f[1] * up[2] - f[2] * up[1];
    r[1] = f[2] * up[0] - f[0] * up[2];
    r[2] = f[0] * up[1] - f[1] * up[0];
    for (int i = 0; i < 3; ++i) rl += r[i] * r[i];
    rl = std::sqrt(rl);
    for (int i = 0; i < 3; ++i) r[i] /= rl;
    u[0] = r[1] * f[2] - r[2] * f[1];
    u[1] = r[2] * f[0] - r[0] * f[2];
    u[2] = r[0] * f[1] - r[1] * f[0];
    const double half_h = std::tan(0.5 * (double)fov_y);           // image plane at distance 1
    const double half_w = half_h * (double)width / (double)height;
    const double sx = 2.0 * half_w / (double)width, sy = 2.0 * half_h / (double)height;
    for (int i = 0; i < 3; ++i) {
        out->eye[i] = eye[i];
        out->du[i] = (float)(r[i] * sx);
        out->dv[i] = (float)(-u[i] * sy);
        double o = (double)eye[i] + f[i] - r[i] * (half_w - 0.5 * sx) + u[i] * (half_h - 0.5 * sy);
        out->origin[i] = (float)o;
    }
}

}  // extern "C"
