// sort.hip -- spatial reordering of query batches.  The walk in kernels.hip is
// wave-uniform: a wavefront visits the union of what its 64 lanes need, so 64
// unrelated points would each pay for all the others.  Queries are therefore
// processed along a space-filling curve (30-bit Hilbert index inside the scene
// box); results are written back at the caller's original index, so the order is
// invisible.  (Names still say "morton": the first version used Z-order.)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <cstdlib>
#include <mutex>
#include <string>

#include "kernels.hpp"

namespace sdfgpu {

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ uint32_t curve_key30(const float4 p, float lox, float loy, float loz, float sx, float sy, float sz) {
    const float fx = __builtin_fminf(__builtin_fmaxf((p.x - lox) * sx, 0.f), 1023.f);
    const float fy = __builtin_fminf(__builtin_fmaxf((p.y - loy) * sy, 0.f), 1023.f);
    const float fz = __builtin_fminf(__builtin_fmaxf((p.z - loz) * sz, 0.f), 1023.f);
    // NaN coordinates fall to cell 0 (fmax/fmin return the non-NaN operand)
    // Hilbert index of the cell (Skilling, "Programming the Hilbert curve", 2004: axes -> transposed index,
    // 10 bits per axis), bits interleaved most significant axis first.  Z-order was the first choice; along
    // the Hilbert curve consecutive cells are always face neighbours, so 64 consecutive points form a more
    // compact group and the wavefront visits fewer nodes: k = 8 over 10 M points 2.14 -> 2.02 ms.
    uint32_t X[3] = {(uint32_t)fx, (uint32_t)fy, (uint32_t)fz};
    for (uint32_t Q = 1u << 9; Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (X[a] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = 1u << 9; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
}

__global__ __launch_bounds__(256) void morton_kernel(const float4* __restrict__ pts, uint32_t n, float lox, float loy,
                                                     float loz, float sx, float sy, float sz,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    keys[i] = curve_key30(pts[i], lox, loy, loz, sx, sy, sz);
    idx[i] = i;
}

// ---- the batch sort, written for this key shape --------------------------------------------------------------
// What the walk needs from the sort is a permutation that puts spatial neighbours into the same wavefront: 24 bits
// of curve key (16 M cells for at most a few tens of millions of points) and a 32-bit index, order among equal keys
// free.  A general radix sort (hipcub onesweep: key computation + three 8-bit passes over key AND value, 720 MB moved
// for 10 M points, 0.35 ms) is more than that needs.  Here: two 12-bit digits, most significant first.
//   1. keys_hist: every block takes a contiguous share of the batch, computes the keys (stored: 4 B per point) and
//      counts its share per high digit in LDS -> table[digit][block].
//   2. row_scan + base_scan: exclusive prefix over (digit, block): where each block's points of each digit go.
//   3. scatter: the same blocks re-read their keys and write (low digit, index) to their places: the batch is now
//      grouped by high digit ("buckets" of n / 4096 points).
//   4. bucket_sort: a workgroup per bucket counts, scans and places its points by low digit in LDS counters and
//      writes the final permutation.  A bucket of more than RS_CHUNK points (clustered batches, outliers clamped
//      to the border cells) is cut into chunks that are sorted one by one: the permutation is then sorted within
//      every chunk only, which costs coherence in that one cell and nothing else -- any permutation is a valid
//      processing order, results are written back by index.
// No pass carries the points themselves (counted in round 2: that moves more than the gather it saves).  The order
// among equal keys, and of a bucket's points before step 4, depends on the order LDS atomics are served in: the
// permutation can differ from run to run between points of one cell, the results cannot.
constexpr uint32_t RS_DIGITS = 4096;      // 12 bits
constexpr uint32_t RS_MAX_BLOCKS = 512;   // shares of the batch the key / histogram pass is cut into (one workgroup of 1024 each)
constexpr uint32_t RS_MIN_SHARE = 16384;  // points per share at least (small batches use fewer blocks)
// The scatter pass can run one workgroup per `group` consecutive shares (SDFGPU_RS_GROUP, a measurement knob).  Measured
// on 10 M random points, 512 shares: group 1 / 2 / 4 / 8 / 16 = 126 / 126 / 169 / 246 / 457 us -- the pass is bound by how
// many scattered 8-byte stores are in flight, not by the L2 footprint of its open lines, so every share gets a workgroup.
static uint32_t rs_group() {
    static const uint32_t v = [] { const char* e = std::getenv("SDFGPU_RS_GROUP"); const int g = e ? std::atoi(e) : 1; return (uint32_t)(g < 1 ? 1 : g); }();
    return v;
}
constexpr uint32_t RS_CHUNK = 65536;      // most points one bucket_sort workgroup places (a load-balance limit: the counters are all the LDS it needs)

__global__ __launch_bounds__(1024) void rs_keys_hist_kernel(const float4* __restrict__ pts, uint32_t n, uint32_t share, uint32_t n_blocks,
                                                            float lox, float loy, float loz, float sx, float sy, float sz,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ table) {
    __shared__ uint32_t h[RS_DIGITS];
    for (uint32_t j = threadIdx.x; j < RS_DIGITS; j += 1024u) h[j] = 0u;
    __syncthreads();
    const uint32_t b = blockIdx.x;
    const uint32_t lo = b * share, hi = lo + share < n ? lo + share : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
        const uint32_t k = curve_key30(pts[i], lox, loy, loz, sx, sy, sz) >> 6;   // the leading 24 bits
        keys[i] = k;
        atomicAdd(&h[k >> 12], 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < RS_DIGITS; j += 1024u) table[(size_t)j * n_blocks + b] = h[j];
}

// exclusive prefix inside every digit's row of the table (one wavefront per row), and the row's total
__global__ __launch_bounds__(256) void rs_row_scan_kernel(uint32_t* __restrict__ table, uint32_t n_blocks, uint32_t* __restrict__ totals) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= RS_DIGITS) return;
    uint32_t* __restrict__ r = table + (size_t)row * n_blocks;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += 64u) {
        const uint32_t i = c0 + lane;
        const uint32_t v = i < n_blocks ? r[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            incl += lane >= (uint32_t)o ? t : 0u;
        }
        if (i < n_blocks) r[i] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
    if (lane == 0) totals[row] = carry;
}

// base[d] = points of all digits below d; chunk_base[d] = bucket_sort workgroups of all digits below d
__global__ __launch_bounds__(1024) void rs_base_scan_kernel(const uint32_t* __restrict__ totals, uint32_t* __restrict__ base,
                                                            uint32_t* __restrict__ chunk_base) {
    __shared__ uint32_t pa[1024], pb[1024];
    const uint32_t tid = threadIdx.x;
    uint32_t v[4], c[4], sa = 0, sb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = totals[tid * 4u + j];
        c[j] = (v[j] + RS_CHUNK - 1u) / RS_CHUNK;
        sa += v[j];
        sb += c[j];
    }
    pa[tid] = sa;
    pb[tid] = sb;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t a = tid >= off ? pa[tid - off] : 0u, b = tid >= off ? pb[tid - off] : 0u;
        __syncthreads();
        pa[tid] += a;
        pb[tid] += b;
        __syncthreads();
    }
    uint32_t ra = pa[tid] - sa, rb = pb[tid] - sb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        base[tid * 4u + j] = ra;
        chunk_base[tid * 4u + j] = rb;
        ra += v[j];
        rb += c[j];
    }
    if (tid == 1023u) {
        base[RS_DIGITS] = ra;
        chunk_base[RS_DIGITS] = rb;
    }
}

__global__ __launch_bounds__(1024) void rs_scatter_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t share, uint32_t n_blocks, uint32_t group,
                                                          const uint32_t* __restrict__ table, const uint32_t* __restrict__ base,
                                                          uint2* __restrict__ pairs) {
    __shared__ uint32_t off[RS_DIGITS];
    // `group` consecutive shares: their places follow one another in every digit's row, so the first share's offsets serve all
    const uint32_t b = blockIdx.x * group;
    for (uint32_t j = threadIdx.x; j < RS_DIGITS; j += 1024u) off[j] = base[j] + table[(size_t)j * n_blocks + b];
    __syncthreads();
    const uint64_t lo64 = (uint64_t)b * share, hi64 = lo64 + (uint64_t)group * share;
    const uint32_t lo = (uint32_t)(lo64 < n ? lo64 : n), hi = (uint32_t)(hi64 < n ? hi64 : n);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
        const uint32_t k = keys[i];
        const uint32_t at = atomicAdd(&off[k >> 12], 1u);
        pairs[at] = make_uint2(k & 0xFFFu, i);   // one 8-byte store
    }
}

__global__ __launch_bounds__(1024) void rs_bucket_sort_kernel(const uint2* __restrict__ pairs,
                                                              const uint32_t* __restrict__ base, const uint32_t* __restrict__ chunk_base,
                                                              uint32_t* __restrict__ order) {
    __shared__ uint32_t h[RS_DIGITS];
    __shared__ uint32_t part[1024];
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    if (w >= chunk_base[RS_DIGITS]) return;
    uint32_t d = 0, hi_d = RS_DIGITS;          // the digit whose chunk this is: chunk_base[d] <= w < chunk_base[d + 1]
    while (hi_d - d > 1u) {
        const uint32_t mid = (d + hi_d) >> 1;
        if (chunk_base[mid] <= w) d = mid; else hi_d = mid;
    }
    const uint32_t begin = base[d] + (w - chunk_base[d]) * RS_CHUNK;
    const uint32_t end = begin + RS_CHUNK < base[d + 1] ? begin + RS_CHUNK : base[d + 1];
    for (uint32_t j = tid; j < RS_DIGITS; j += 1024u) h[j] = 0u;
    __syncthreads();
    for (uint32_t i = begin + tid; i < end; i += 1024u) atomicAdd(&h[pairs[i].x], 1u);
    __syncthreads();
    // exclusive scan of the 4096 counters: 4 consecutive ones per thread, then the 1024 partial sums
    uint32_t loc[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { loc[j] = h[tid * 4u + j]; sum += loc[j]; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t a = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += a;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[tid * 4u + j] = run; run += loc[j]; }
    __syncthreads();
    for (uint32_t i = begin + tid; i < end; i += 1024u) {
        const uint2 e = pairs[i];
        const uint32_t at = atomicAdd(&h[e.x], 1u);
        order[begin + at] = e.y;
    }
}

static uint32_t rs_max_blocks() {
    static const uint32_t v = [] { const char* e = std::getenv("SDFGPU_RS_BLOCKS"); const int g = e ? std::atoi(e) : (int)RS_MAX_BLOCKS; return (uint32_t)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }();
    return v;
}
static uint32_t rs_blocks(uint32_t n) {
    const uint32_t b = (n + RS_MIN_SHARE - 1u) / RS_MIN_SHARE;
    return b < 1u ? 1u : (b > rs_max_blocks() ? rs_max_blocks() : b);
}
static size_t rs_temp_bytes(uint32_t n) {   // table + totals + base + chunk_base
    return ((size_t)RS_DIGITS * rs_blocks(n) + RS_DIGITS + 2u * (RS_DIGITS + 1u) + 64u) * sizeof(uint32_t);
}
static bool use_library_sort() {   // SDFGPU_SORT=hipcub: the round-2 path (key kernel + hipcub radix sort), for comparison
    static const bool v = [] { const char* e = std::getenv("SDFGPU_SORT"); return e && std::string(e) == "hipcub"; }();
    return v;
}

size_t sort_temp_bytes(uint32_t n) {
    if (!use_library_sort()) return rs_temp_bytes(n);
    size_t bytes = 0;
    uint32_t* k = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, k, k, (int)n, 0, 30, (hipStream_t)0);
    return bytes;
}

hipError_t launch_morton_order(const float* pts, uint32_t n, const float lo[3], const float hi[3], uint32_t* keys_in,
                               uint32_t* keys_out, uint32_t* idx_in, uint32_t* idx_out, void* temp, size_t temp_bytes,
                               hipStream_t s) {
    if (n == 0) return hipSuccess;
    float sc[3];
    for (int a = 0; a < 3; ++a) {
        float ext = hi[a] - lo[a];
        sc[a] = ext > 0.f ? 1024.0f / ext : 0.f;
    }
    if (!use_library_sort()) {
        // keys_in: keys; keys_out: n (low digit, index) PAIRS of 8 bytes grouped by high digit (idx_in is not used);
        // idx_out: the permutation
        if (temp_bytes < rs_temp_bytes(n)) return hipErrorInvalidValue;
        const uint32_t nb = rs_blocks(n), share = (n + nb - 1u) / nb;
        uint32_t* table = (uint32_t*)temp;
        uint32_t* totals = table + (size_t)RS_DIGITS * nb;
        uint32_t* base = totals + RS_DIGITS;
        uint32_t* chunk_base = base + (RS_DIGITS + 1u);
        rs_keys_hist_kernel<<<dim3(nb), dim3(1024), 0, s>>>((const float4*)pts, n, share, nb, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], keys_in, table);
        rs_row_scan_kernel<<<dim3(RS_DIGITS / 4u), dim3(256), 0, s>>>(table, nb, totals);
        rs_base_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(totals, base, chunk_base);
        const uint32_t group = rs_group();
        rs_scatter_kernel<<<dim3((nb + group - 1u) / group), dim3(1024), 0, s>>>(keys_in, n, share, nb, group, table, base, (uint2*)keys_out);
        // one workgroup per chunk: at most one per digit plus one per RS_CHUNK points (the others find nothing to do)
        rs_bucket_sort_kernel<<<dim3(RS_DIGITS + (n + RS_CHUNK - 1u) / RS_CHUNK), dim3(1024), 0, s>>>((const uint2*)keys_out, base, chunk_base, idx_out);
        return hipGetLastError();
    }
    morton_kernel<<<dim3((n + 255u) / 256u), dim3(256), 0, s>>>((const float4*)pts, n, lo[0], lo[1], lo[2], sc[0], sc[1],
                                                                sc[2], keys_in, idx_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // As many leading bits as there are points to tell apart (about one point per cell); every 8
    // bits less is one radix pass.  Measured on 10 M points: 24 bits 2.46 ms per kNN batch, 30 bits
    // 2.57, 16 bits 3.32 (neighbours inside a wavefront stop being neighbours).
    int bits = 0;
    while ((1ull << bits) < n) ++bits;
    if (const char* e = std::getenv("SDFGPU_MORTON_BITS")) bits = std::atoi(e);
    bits = bits < 8 ? 8 : (bits > 30 ? 30 : bits);
    bits = (bits + 7) / 8 * 8;           // whole radix digits cost the same
    if (bits > 30) bits = 30;
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (int)n, 30 - bits, 30,
                                              s);
}

// ---- scratch of the allocation-free entry point (sdfgpu_nearest_dev_static) --
// The reordering above needs device memory beyond the caller's inputs and outputs: two key
// arrays, two index arrays and the radix sort's temporary storage.  sdfgpu_nearest_dev keeps
// them in grow-only hipMalloc'ed buffers of the context; the allocation-free variant keeps them
// here, in the code object's own memory, so that it never has to allocate.  Each GPU the library
// runs on holds its own copy of this array at its own address until the process ends; the
// array's name is only a host-side stand-in for it, so host code asks the runtime for the
// device address once per device and passes THAT to kernels and to hipcub.
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

hipError_t static_scratch_address(int device, void** ptr) {
    // looked up with `device` current (the caller, sdfgpu_create, has just set it)
    static std::mutex mu;
    static void* known[64] = {nullptr};
    *ptr = nullptr;
    if (device < 0 || device >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(mu);
    if (!known[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        known[device] = p;
    }
    *ptr = known[device];
    return hipSuccess;
}

// ---- tile scheduling (render) ------------------------------------------------
// Longest-first order of a frame's tiles from the step counts the previous frame on the same
// stream recorded: stable descending sort on the cost's bits [2, 8) (64 buckets of 4 march
// steps; tiles of one bucket keep their spatial order, which keeps neighbours -- and their
// scalar-cache lines -- together).
#ifndef TILE_SORT_LO
#define TILE_SORT_LO 2
#endif
__global__ __launch_bounds__(256) void iota_kernel(uint32_t* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) v[i] = i;
}

size_t tile_sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    uint32_t* k = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, k, k, k, k, (int)n, TILE_SORT_LO, 8, (hipStream_t)0);
    return bytes;
}

hipError_t launch_iota(uint32_t* v, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    iota_kernel<<<dim3((n + 255u) / 256u), dim3(256), 0, s>>>(v, n);
    return hipGetLastError();
}

hipError_t launch_tile_order(const uint32_t* cost, const uint32_t* iota, uint32_t* keys_out, uint32_t* order_out,
                             uint32_t n, void* temp, size_t temp_bytes, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, cost, keys_out, iota, order_out, (int)n, TILE_SORT_LO, 8, s);
}

}  // namespace sdfgpu
