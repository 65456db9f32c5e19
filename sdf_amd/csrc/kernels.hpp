// kernels.hpp -- launch interface between the C ABI (api.cpp) and kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/sdfgpu.h"
#include "gpu_tree.hpp"

namespace sdfgpu {

// Nearest-candidate grid (grid.hip): a uniform grid of cells over the padded scene box, cells grouped into
// bricks of 4 x 4 x 4.  For a cell with centre c and (inflated) half diagonal r, every primitive that can be
// the nearest one -- or tie for it -- at ANY point of the cell has value(c) <= min value(c) + 2 lip r + M (leaf
// values are lip-Lipschitz, M covers their rounding), so that short candidate list decides a query anywhere in
// the cell without a tree walk.  A brick keeps the union of its cells' lists (at most 32 primitive ids); a
// cell keeps a 32-bit mask into its brick's list and `lb` = a lower bound, at any point of the cell, on every
// primitive NOT in its list.  mask == 0: the cell has no list (too many candidates): walk.
struct GridCell {
    uint32_t mask;
    float lb;
};
// The 8th- and 16th-nearest value at the cell's centre c, plus the rounding margin M.  Each of the j nearest primitives
// at c is within d_j(c) + lip |p - c| + rounding of a point p, so d_j(c) + M + lip |p - c| bounds the distance of p's
// j-th nearest primitive from above: the k-nearest query starts its walk with the prune radius capped there instead
// of at +inf (+inf here: the tree has fewer primitives than that).
struct GridCap {
    float d8, d16;
};
constexpr uint32_t GRID_BRICK_IDS = 32;   // entries of a brick's candidate list (one mask bit each)
constexpr uint32_t GRID_CELL_K = 16;      // candidates looked at per cell (the K of the k-nearest query that builds it)
struct CandidateGrid {
    // ONE allocation: [GridBlock | cells | brick ids | caps].  The march addresses both tables relative to `block` (a
    // pointer pair it holds anyway), which is worth two SGPRs it does not have to spare.
    GridCell* cells = nullptr;      // [bricks][64], cell (x, y, z) of a brick at (z & 3) << 4 | (y & 3) << 2 | (x & 3)
    uint32_t* brick_ids = nullptr;  // [bricks][GRID_BRICK_IDS]
    GridCap* caps = nullptr;        // [bricks][64], as cells
    float lo[3] = {0.f, 0.f, 0.f};
    float inv_cell[3] = {0.f, 0.f, 0.f};
    uint32_t n[3] = {0, 0, 0};      // cells per axis (multiples of 4)
    uint64_t bytes = 0;
    void* block = nullptr;          // GridBlock in HBM: what the march reads (one scalar load per list pass)
};
// The 64 bytes the march kernel fetches at the start of a list pass.
struct alignas(64) GridBlock {
    float lo[3];
    float inv_cell[3];
    uint32_t n[3];
    uint32_t nbx, nby;              // bricks per axis (x, y)
    uint32_t bricks_off;            // byte offset of the brick id table from this block (the cells follow the block directly)
    uint32_t caps_off;              // byte offset of the k-nearest radius caps
    float cell[3];                  // cell size per axis (1 / inv_cell as the build computed it: cell centres are lo + (i + 0.5) * cell)
};
static_assert(sizeof(GridBlock) == 64, "GridBlock is one 64-byte scalar load");

// Device-resident compact tree (pointers into HBM).
struct DeviceTree {
    CandidateGrid grid;                 // cells == nullptr: none (general path, OBB / NONE bounds, tiny trees, switched off)
    GpuLeaf* leaf = nullptr;
    GpuInner* inner = nullptr;
    GpuInnerObb* inner_obb = nullptr;
    // chain path (gpu_tree.hpp GpuChainMat; null otherwise): ONE allocation, [n_leaf x {first, count}] padded to 64
    // bytes, then the chain records
    void* chain = nullptr;
    uint32_t chain_max = 0;             // longest chain (scales the rounding margin)
    GpuGenNode* gen = nullptr;          // general path: all nodes, pre-order
    uint32_t n_gen = 0;
    bool gen_eval = false;              // general path runs the post-order combine (smooth union in effect)
    bool gen_prune = false;             // general path: nodes carry boxes, the walks skip subtrees (gpu_tree.hpp GpuGenNode)
    uint32_t gen_pt_levels = 0;         // ... and their per-lane point stack has this many slots
    uint32_t root_ref = REF_NONE;
    uint32_t n_leaf = 0;
    uint32_t stack_depth = 1;
    uint32_t bounds = SDFGPU_BOUNDS_AABB;
    float margin_T = 0.f;
    float skip_ext = 0.f;
    float lipschitz = 1.000004f;
    float box_scale = 1.f;
};

// Internal accessors of the opaque context for comm.cpp (defined in api.cpp).
int ctx_device(const sdfgpu_ctx* c);
int ctx_set_error(sdfgpu_ctx* c, int code, const char* msg);

// st != nullptr selects the counting instantiation of the same kernel.
// stats_asm_walk (with st): count with the assembly walk's own wave-level counters instead of the C++ walk.
hipError_t launch_nearest(const DeviceTree& d, const float* pts_xyzw, const uint32_t* order, uint32_t n,
                          float* out_dist, uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s, bool stats_asm_walk = false);

// k nearest (k <= SDFGPU_MAX_K): out[i*k + j], ascending by (distance, larger index first).
hipError_t launch_knn(const DeviceTree& d, const float* pts_xyzw, const uint32_t* order, uint32_t n, uint32_t k,
                      float* out_dist, uint32_t* out_prim, sdfgpu_stats* st, hipStream_t s);

// Morton order of a query batch inside [lo, hi): idx_out = permutation (sort.hip).
size_t sort_temp_bytes(uint32_t n);
hipError_t launch_morton_order(const float* pts_xyzw, uint32_t n, const float lo[3], const float hi[3],
                               uint32_t* keys_in, uint32_t* keys_out, uint32_t* idx_in, uint32_t* idx_out, void* temp,
                               size_t temp_bytes, hipStream_t s);

// Scratch of the allocation-free nearest query (sort.hip): a file-scope __device__ array of
// STATIC_SCRATCH_BYTES, laid out as four arrays of SDFGPU_STATIC_MAX_POINTS 32-bit words (Morton
// keys in / out, permutation in / out) followed by the radix sort's temporary storage.
// static_scratch_address returns the array's address on `device` (which must be current).
constexpr size_t STATIC_SCRATCH_BYTES = 8u << 20;
hipError_t static_scratch_address(int device, void** ptr);

// Scratch of the multi-pass march (ray compaction): two state buffers of
// tile_count*64 16-byte records, per-wave counts and offsets (tile_count+1), two headers.
struct RenderScratch {
    void* state[2] = {nullptr, nullptr};
    uint32_t* counts = nullptr;
    uint32_t* offsets = nullptr;
    uint32_t* hdr[2] = {nullptr, nullptr};
    // tile scheduling: wave w of the (single-pass) march renders tile tile_order[w] of the call's
    // tile range; every wave records min(march steps it ran, 255) in tile_cost[its tile].  Either may be null.
    const uint32_t* tile_order = nullptr;
    uint32_t* tile_cost = nullptr;
    // 16 bytes: the result of the walk from the eye (every ray's first march step stands there when t_min = 0),
    // made once per frame by a one-wavefront kernel instead of once per tile.  Null: every tile walks it itself.
    float* eye = nullptr;
};
size_t tile_sort_temp_bytes(uint32_t n);
hipError_t launch_iota(uint32_t* v, uint32_t n, hipStream_t s);
hipError_t launch_tile_order(const uint32_t* cost, const uint32_t* iota, uint32_t* keys_out, uint32_t* order_out,
                             uint32_t n, void* temp, size_t temp_bytes, hipStream_t s);
size_t render_scratch_bytes(uint32_t tile_count, size_t* state_bytes, size_t* count_bytes);

// pass_bounds: 0 = b[0] < ... < b[n-1] = max_steps (n >= 2; n == 2 is a single pass).
// st != nullptr selects the counting instantiation of the same kernels.
hipError_t launch_render(const DeviceTree& d, const sdfgpu_camera& cam, uint32_t w, uint32_t h,
                         const sdfgpu_march_params& mp, uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count,
                         bool row_major, const RenderScratch& rs, const uint32_t* pass_bounds, uint32_t n_bounds,
                         uint8_t* hit, float* depth, float* normal, uint32_t* prim, sdfgpu_stats* st, hipStream_t s);

// Builds the candidate grid of a resident tree (AABB / BVH bounds) on `s`: cell centres -> k-nearest query
// (the library's own kNN kernel, K = GRID_CELL_K) -> per-brick union.  `d.grid` must describe the geometry (lo,
// inv_cell, n) and point at cells / brick_ids sized for it; temp holds the per-chunk centres and rows.
size_t grid_temp_bytes(uint32_t chunk_bricks);
hipError_t build_candidate_grid(const DeviceTree& d, float cell_radius_inflated, float list_margin, void* temp,
                                uint32_t chunk_bricks, hipStream_t s);

// MINDIST / MAXDIST of one box at n points (faux_shader.rs:39-58, all four lanes).
hipError_t launch_box_bounds(const SdfBoundingBoxBlock* d_box, const float* d_pts_xyzw, uint32_t n, float* d_min,
                             float* d_max, hipStream_t s);

hipError_t launch_untile(uint32_t w, uint32_t h, uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count,
                         const uint8_t* t_hit, const float* t_depth, const float* t_normal, const uint32_t* t_prim,
                         uint8_t* hit, float* depth, float* normal, uint32_t* prim, hipStream_t s);

// Whole-frame assembly on the gathering rank in one launch: rank p's packed planes ([depth | normal | prim |
// hit], `px` pixels each) at gathered + p * nbytes, the caller's own share at `mine`.
hipError_t launch_untile_gathered(uint32_t w, uint32_t h, uint32_t world, uint32_t self, size_t px, size_t nbytes,
                                  const void* gathered, const void* mine, uint8_t* hit, float* depth, float* normal,
                                  uint32_t* prim, hipStream_t s);

}  // namespace sdfgpu
