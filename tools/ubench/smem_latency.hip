// smem_latency.hip -- dependent scalar-load chain: cycles per s_load_dwordx16 (one lone wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
__global__ void chase(const unsigned* base, unsigned start, int n, long long* out) {
    unsigned cur = __builtin_amdgcn_readfirstlane(start);
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        unsigned off = cur << 6, nxt;
        asm volatile("s_load_dwordx16 s[80:95], %1, %2\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s80"
                     : "=s"(nxt) : "s"(base), "s"(off)
                     : "memory", "s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95");
        cur = nxt;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = cur; }
}
int main() {
    for (size_t recs : {64ul, 128ul, 1024ul, 16384ul, 262144ul, 4194304ul}) {   // 4 KB .. 256 MB
        std::vector<unsigned> h(recs * 16, 0);
        std::vector<unsigned> perm(recs);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(perm.begin() + 1, perm.end(), rng);
        for (size_t i = 0; i < recs; ++i) h[perm[i] * 16] = perm[(i + 1) % recs];
        unsigned* d; long long* o;
        hipMalloc(&d, h.size() * 4); hipMalloc(&o, 16);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const int n = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0u, n, o);
            hipDeviceSynchronize();
        }
        long long r[2];
        hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
        printf("%8zu records (%8.1f KB): %.1f cycles per dependent s_load_dwordx16\n", recs, recs * 64 / 1024.0, (double)r[0] / n);
        hipFree(d); hipFree(o);
    }
    return 0;
}
