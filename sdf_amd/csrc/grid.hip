// grid.hip -- build of the nearest-candidate grid (kernels.hpp, CandidateGrid) at upload.
//
// No reference counterpart: the crate evaluates one point at a time on the CPU and has no acceleration
// structure besides its tree.  The grid changes no result: a march step whose point lies in a cell with a
// list evaluates exactly that list with the reference's own leaf arithmetic, and the list provably contains
// every primitive that can attain -- or tie for -- the minimum anywhere in the cell (proof below), so the
// (distance, later index) it returns is the one the exhaustive walk returns.
//
// Proof sketch.  v_i(p) = computed value of primitive i at p, e(p) >= |v_i(p) - true distance| (the prune
// margin of kernels.hip), lip = Lipschitz bound of the leaf transforms, r = half diagonal of the cell,
// inflated by 1 % (a point's cell index is formed in fp32: the point may lie a few ulps outside the cell it
// is assigned to).  j* = the computed-nearest primitive at the centre c.  For a primitive i with
// v_i(c) > v_j*(c) + 2 lip r + M and a point p of the cell:
//     v_i(p)  >= v_i(c) - lip r - 2 e   >  v_j*(c) + lip r + M - 2 e
//     v_j*(p) <= v_j*(c) + lip r + 2 e
// so v_i(p) - v_j*(p) > M - 4 e >= 0 with M = 8 e_max: i is strictly farther than j* at p, it can neither be
// the minimum nor tie for it.  The same line gives the cell's `lb`: every primitive outside the list has
// true distance > v_j*(c) + lip r at p (the march subtracts its own rounding margin, as it does for a walk).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

extern "C" __device__ uint32_t __ockl_wfred_min_u32(uint32_t);

namespace sdfgpu {

static_assert(GRID_CELL_K == 16, "GridCap keeps the 8th and the 16th of a cell's GRID_CELL_K rows");

// centre of cell `lane` of brick b: bricks are numbered x fastest, cells inside a brick likewise
__global__ __launch_bounds__(256) void grid_centres_kernel(uint32_t brick0, uint32_t n_bricks, uint32_t nbx, uint32_t nby,
                                                           float lox, float loy, float loz, float cx, float cy, float cz,
                                                           float4* __restrict__ pts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bl = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (bl >= n_bricks) return;
    const uint32_t b = brick0 + bl;
    const uint32_t bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
    const uint32_t x = bx * 4u + (lane & 3u), y = by * 4u + ((lane >> 2) & 3u), z = bz * 4u + (lane >> 4);
    pts[(size_t)bl * 64u + lane] = make_float4(lox + ((float)x + 0.5f) * cx, loy + ((float)y + 0.5f) * cy,
                                               loz + ((float)z + 0.5f) * cz, 1.0f);
}

// One wavefront per brick, one cell per lane.  rows = the GRID_CELL_K nearest (value, id) at the cell's centre,
// ascending.  Candidates of the cell: value <= nearest + list_span; if all K qualify there may be more: no list
// for that cell (unless the tree has no more than K primitives).  Brick list = union over its cells, taken in
// ascending id order by repeated wave-wide minimum; more than GRID_BRICK_IDS entries: no lists in the brick.
__global__ __launch_bounds__(256) void grid_brick_kernel(uint32_t brick0, uint32_t n_bricks, uint32_t n_leaf,
                                                         const float* __restrict__ rows_d, const uint32_t* __restrict__ rows_i,
                                                         float list_span, float lb_add, float cap_add, GridCell* __restrict__ cells,
                                                         uint32_t* __restrict__ brick_ids, GridCap* __restrict__ caps) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bl = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (bl >= n_bricks) return;
    const size_t row = ((size_t)bl * 64u + lane) * GRID_CELL_K;
    float d[GRID_CELL_K];
    uint32_t id[GRID_CELL_K];
#pragma unroll
    for (uint32_t j = 0; j < GRID_CELL_K; ++j) {
        d[j] = rows_d[row + j];
        id[j] = rows_i[row + j];
    }
    const float d0 = d[0];
    const float limit = d0 + list_span;
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t j = 0; j < GRID_CELL_K; ++j) cnt += (id[j] != SDF_NO_PRIM && d[j] <= limit) ? 1u : 0u;
    // a usable centre has a nearest primitive with an ordinary value
    bool listed = id[0] != SDF_NO_PRIM && d0 == d0 && __builtin_fabsf(d0) < __builtin_inff();
    if (cnt == GRID_CELL_K && n_leaf > GRID_CELL_K) listed = false;
    if (!listed) cnt = 0;
    uint32_t mask = 0, n_union = 0;
    uint32_t prev = 0;
    bool first = true, overflow = false;
    for (;;) {
        uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t j = 0; j < GRID_CELL_K; ++j) {
            const bool ok = j < cnt && (first || id[j] > prev) && id[j] < mine;
            mine = ok ? id[j] : mine;
        }
        const uint32_t s = __builtin_amdgcn_readfirstlane(__ockl_wfred_min_u32(mine));
        if (s == 0xFFFFFFFFu) break;
        if (n_union == GRID_BRICK_IDS) { overflow = true; break; }
        bool has = false;
#pragma unroll
        for (uint32_t j = 0; j < GRID_CELL_K; ++j) has |= j < cnt && id[j] == s;
        mask |= has ? (1u << n_union) : 0u;
        if (lane == 0) brick_ids[(size_t)(brick0 + bl) * GRID_BRICK_IDS + n_union] = s;
        n_union += 1;
        prev = s;
        first = false;
    }
    if (lane >= n_union && lane < GRID_BRICK_IDS) brick_ids[(size_t)(brick0 + bl) * GRID_BRICK_IDS + lane] = 0u;   // never referenced
    GridCell c;
    c.mask = (overflow || !listed) ? 0u : mask;
    c.lb = d0 + lb_add;
    cells[(size_t)(brick0 + bl) * 64u + lane] = c;
    // k-nearest radius caps (kernels.hpp GridCap)
    GridCap q;
    const float inf = __builtin_inff();
    q.d8 = (id[7] != SDF_NO_PRIM && d[7] == d[7]) ? d[7] + cap_add : inf;
    q.d16 = (id[GRID_CELL_K - 1] != SDF_NO_PRIM && d[GRID_CELL_K - 1] == d[GRID_CELL_K - 1]) ? d[GRID_CELL_K - 1] + cap_add : inf;
    caps[(size_t)(brick0 + bl) * 64u + lane] = q;
}

size_t grid_temp_bytes(uint32_t chunk_bricks) {
    const size_t cells = (size_t)chunk_bricks * 64u;
    return cells * (sizeof(float4) + GRID_CELL_K * (sizeof(float) + sizeof(uint32_t)));
}

hipError_t build_candidate_grid(const DeviceTree& d, float cell_radius_inflated, float list_margin, void* temp,
                                uint32_t chunk_bricks, hipStream_t s) {
    const CandidateGrid& g = d.grid;
    const uint32_t nbx = g.n[0] / 4u, nby = g.n[1] / 4u, nbz = g.n[2] / 4u;
    const uint64_t n_bricks = (uint64_t)nbx * nby * nbz;
    if (n_bricks == 0 || chunk_bricks == 0) return hipSuccess;
    const float list_span = 2.0f * d.lipschitz * cell_radius_inflated + list_margin;
    const float lb_add = d.lipschitz * cell_radius_inflated;
    const float cap_add = list_margin;   // the query adds lip |p - c| itself
    char* base = (char*)temp;
    float4* pts = (float4*)base;
    float* rows_d = (float*)(base + (size_t)chunk_bricks * 64u * sizeof(float4));
    uint32_t* rows_i = (uint32_t*)((char*)rows_d + (size_t)chunk_bricks * 64u * GRID_CELL_K * sizeof(float));
    DeviceTree walk = d;        // the query that builds the grid must not consult it
    walk.grid = CandidateGrid();
    for (uint64_t b0 = 0; b0 < n_bricks; b0 += chunk_bricks) {
        const uint32_t nb = (uint32_t)(n_bricks - b0 < chunk_bricks ? n_bricks - b0 : chunk_bricks);
        const dim3 grid((nb + 3u) / 4u), block(256);
        grid_centres_kernel<<<grid, block, 0, s>>>((uint32_t)b0, nb, nbx, nby, g.lo[0], g.lo[1], g.lo[2], 1.0f / g.inv_cell[0],
                                                   1.0f / g.inv_cell[1], 1.0f / g.inv_cell[2], pts);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        // cells of a brick are consecutive points: one wavefront of the query walks for one brick
        e = launch_knn(walk, (const float*)pts, nullptr, nb * 64u, GRID_CELL_K, rows_d, rows_i, nullptr, s);
        if (e != hipSuccess) return e;
        grid_brick_kernel<<<grid, block, 0, s>>>((uint32_t)b0, nb, d.n_leaf, rows_d, rows_i, list_span, lb_add, cap_add, g.cells, g.brick_ids, g.caps);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sdfgpu
