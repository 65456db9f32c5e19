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

#include "kernels.hpp"

namespace sdfgpu {

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void morton_kernel(const float4* __restrict__ pts, uint32_t n, float lox, float loy,
                                                     float loz, float sx, float sy, float sz,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
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
    keys[i] = (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
    idx[i] = i;
}

size_t sort_temp_bytes(uint32_t n) {
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
