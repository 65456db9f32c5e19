// comm.cpp -- multi-GPU framebuffer assembly behind the C ABI: one process per GPU, the
// tree replicated, a frame's 8x8 tiles dealt round-robin over the ranks, ONE exchange at the end of
// the frame: every rank's packed tile buffer travels to the assembling rank with a grouped
// ncclSend / ncclRecv (one point-to-point transfer per peer, each on its own xGMI link; nothing
// inside the march loop), and that rank scatters the tiles into its row-major planes.
//
// The reference has no multi-device code at all; this file has no counterpart there.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the library PyTorch-ROCm and /opt/rocm both
// ship under that soname): a process that already has RCCL loaded gets that copy, a single-GPU
// user never loads it, and libsdfgpu.so carries no link-time dependency on a 570 MB library.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sdfgpu.h"
#include "kernels.hpp"

namespace {

struct Rccl {
    void* handle = nullptr;
    std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

template <class F>
bool bind(void* h, const char* name, F* out, std::string* err) {
    *out = reinterpret_cast<F>(dlsym(h, name));
    if (!*out) *err = std::string("librccl: missing symbol ") + name;
    return *out != nullptr;
}

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.err = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
            return;
        }
        const bool ok = bind(r.handle, "ncclGetUniqueId", &r.GetUniqueId, &r.err) &&
                        bind(r.handle, "ncclCommInitRank", &r.CommInitRank, &r.err) &&
                        bind(r.handle, "ncclCommDestroy", &r.CommDestroy, &r.err) &&
                        bind(r.handle, "ncclGroupStart", &r.GroupStart, &r.err) &&
                        bind(r.handle, "ncclGroupEnd", &r.GroupEnd, &r.err) && bind(r.handle, "ncclSend", &r.Send, &r.err) &&
                        bind(r.handle, "ncclRecv", &r.Recv, &r.err) &&
                        bind(r.handle, "ncclGetErrorString", &r.GetErrorString, &r.err);
        if (!ok) r.handle = nullptr;
    });
    return r;
}

constexpr size_t BYTES_PER_PIXEL = 21;   // depth 4 + normal 12 + prim 4 + hit 1
constexpr size_t TILE_PIXELS = 64;

struct Staging {
    hipStream_t stream = nullptr;
    void* mine = nullptr;       // this rank's packed tiles: [depth | normal | prim | hit], planes of `px` pixels
    void* gathered = nullptr;   // assembling rank: world x nbytes
    size_t nbytes = 0;
    uint64_t stamp = 0;
};

}  // namespace

struct sdfgpu_comm {
    sdfgpu_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    std::vector<Staging> staging;   // one per stream that has rendered through this communicator
    uint64_t clock = 0;
};

namespace {

int fail(sdfgpu_ctx* c, int code, const std::string& msg) { return sdfgpu::ctx_set_error(c, code, msg.c_str()); }

int nccl_fail(sdfgpu_ctx* c, ncclResult_t e, const char* what) {
    return fail(c, SDFGPU_ERR_COMM, std::string(what) + ": " + rccl().GetErrorString(e));
}

void release(Staging& s) {
    if (s.mine) (void)hipFree(s.mine);
    if (s.gathered) (void)hipFree(s.gathered);
    s = Staging();
}

// Staging of (communicator, stream); the least recently used entry is recycled beyond 8 streams
// (after synchronising the stream it belonged to).
int staging_for(sdfgpu_comm* m, hipStream_t stream, size_t nbytes, bool root, Staging** out) {
    Staging* s = nullptr;
    for (auto& e : m->staging)
        if (e.stream == stream && e.mine) s = &e;
    if (!s) {
        if (m->staging.size() < 8) {
            m->staging.emplace_back();
            s = &m->staging.back();
        } else {
            s = &m->staging[0];
            for (auto& e : m->staging)
                if (e.stamp < s->stamp) s = &e;
            if (hipStreamSynchronize(s->stream) != hipSuccess) (void)hipDeviceSynchronize();
            release(*s);
        }
        s->stream = stream;
    }
    if (s->nbytes != nbytes || (root && m->world > 1 && !s->gathered)) {
        hipStream_t keep = s->stream;
        if (s->mine && hipStreamSynchronize(keep) != hipSuccess) (void)hipDeviceSynchronize();
        release(*s);
        s->stream = keep;
        if (hipMalloc(&s->mine, nbytes) != hipSuccess) return fail(m->ctx, SDFGPU_ERR_HIP, "hipMalloc (tile staging)");
        if (root && m->world > 1 && hipMalloc(&s->gathered, nbytes * (size_t)m->world) != hipSuccess)
            return fail(m->ctx, SDFGPU_ERR_HIP, "hipMalloc (gather staging)");
        s->nbytes = nbytes;
    }
    s->stamp = ++m->clock;
    *out = s;
    return SDFGPU_OK;
}

struct Planes {
    uint8_t* hit;
    float* depth;
    float* normal;
    uint32_t* prim;
};
Planes planes_of(void* packed, size_t px) {
    char* b = (char*)packed;
    return Planes{(uint8_t*)(b + px * 20), (float*)b, (float*)(b + px * 4), (uint32_t*)(b + px * 16)};
}

}  // namespace

extern "C" {

int sdfgpu_comm_get_id(sdfgpu_ctx* c, void* id_out) {
    if (!c || !id_out) return SDFGPU_ERR_INVALID_ARG;
    Rccl& r = rccl();
    if (!r.handle) return fail(c, SDFGPU_ERR_COMM, r.err);
    static_assert(sizeof(ncclUniqueId) == SDFGPU_COMM_ID_BYTES, "SDFGPU_COMM_ID_BYTES is sizeof(ncclUniqueId)");
    ncclUniqueId id;
    ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return nccl_fail(c, e, "ncclGetUniqueId");
    std::memcpy(id_out, &id, sizeof id);
    return SDFGPU_OK;
}

int sdfgpu_comm_create(sdfgpu_ctx* c, const void* id_bytes, int rank, int world, sdfgpu_comm** out) {
    if (!c || !id_bytes || !out) return SDFGPU_ERR_INVALID_ARG;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, SDFGPU_ERR_INVALID_ARG, "rank / world out of range");
    Rccl& r = rccl();
    if (!r.handle) return fail(c, SDFGPU_ERR_COMM, r.err);
    if (hipSetDevice(sdfgpu::ctx_device(c)) != hipSuccess) return fail(c, SDFGPU_ERR_HIP, "hipSetDevice");
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    ncclResult_t e = r.CommInitRank(&comm, world, id, rank);
    if (e != ncclSuccess) return nccl_fail(c, e, "ncclCommInitRank");
    sdfgpu_comm* m = new sdfgpu_comm();
    m->ctx = c;
    m->comm = comm;
    m->rank = rank;
    m->world = world;
    *out = m;
    return SDFGPU_OK;
}

void sdfgpu_comm_destroy(sdfgpu_comm* m) {
    if (!m) return;
    (void)hipSetDevice(sdfgpu::ctx_device(m->ctx));
    for (auto& s : m->staging) {
        if (s.mine && hipStreamSynchronize(s.stream) != hipSuccess) (void)hipDeviceSynchronize();
        release(s);
    }
    if (m->comm) (void)rccl().CommDestroy(m->comm);
    delete m;
}

int sdfgpu_render_frame_comm(sdfgpu_ctx* c, sdfgpu_comm* m, const sdfgpu_camera* cam, uint32_t w, uint32_t h,
                             const sdfgpu_march_params* mp, int root, uint8_t* d_hit, float* d_depth, float* d_normal,
                             uint32_t* d_prim, void* stream_) {
    if (!c || !m || m->ctx != c) return SDFGPU_ERR_INVALID_ARG;
    if (root < 0 || root >= m->world) return fail(c, SDFGPU_ERR_INVALID_ARG, "root out of range");
    if (!cam || !mp || w == 0 || h == 0) return fail(c, SDFGPU_ERR_INVALID_ARG, "null camera / march params or empty image");
    if ((uint64_t)w * h > (1ull << 30)) return fail(c, SDFGPU_ERR_TOO_LARGE, "image larger than 2^30 pixels");
    const bool is_root = m->rank == root;
    if (is_root && !d_hit) return fail(c, SDFGPU_ERR_INVALID_ARG, "null hit buffer on the assembling rank");
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t tiles = ((w + 7) / 8) * ((h + 7) / 8);
    const uint32_t world = (uint32_t)m->world, rank = (uint32_t)m->rank;
    const size_t per = (tiles + world - 1) / world;          // tiles of the largest share
    const size_t px = per * TILE_PIXELS;
    const size_t nbytes = px * BYTES_PER_PIXEL;
    auto count_of = [&](uint32_t r) { return r < tiles ? (tiles - r + world - 1) / world : 0u; };
    Staging* s = nullptr;
    int rc = staging_for(m, stream, nbytes, is_root, &s);
    if (rc) return rc;
    // 1. this rank's tiles r, r + world, ... into its packed buffer
    const Planes mine = planes_of(s->mine, px);
    if ((rc = sdfgpu_render_tiles_dev(c, cam, w, h, mp, rank, world, count_of(rank), mine.hit, mine.depth, mine.normal, mine.prim,
                                      stream)))
        return rc;
    // 2. one grouped exchange: every peer sends its buffer to the assembling rank
    if (world > 1) {
        Rccl& r = rccl();
        ncclResult_t e = r.GroupStart();
        if (e != ncclSuccess) return nccl_fail(c, e, "ncclGroupStart");
        if (is_root) {
            for (uint32_t p = 0; p < world && e == ncclSuccess; ++p)
                if (p != rank) e = r.Recv((char*)s->gathered + (size_t)p * nbytes, nbytes, ncclUint8, (int)p, m->comm, stream);
        } else {
            e = r.Send(s->mine, nbytes, ncclUint8, root, m->comm, stream);
        }
        const ncclResult_t e2 = r.GroupEnd();
        if (e != ncclSuccess) return nccl_fail(c, e, "ncclSend / ncclRecv");
        if (e2 != ncclSuccess) return nccl_fail(c, e2, "ncclGroupEnd");
    }
    // 3. assembly on the root: tile-compact -> row-major, every rank's share in one launch
    if (is_root) {
        if (hipSetDevice(sdfgpu::ctx_device(c)) != hipSuccess) return fail(c, SDFGPU_ERR_HIP, "hipSetDevice");
        if (sdfgpu::launch_untile_gathered(w, h, world, rank, px, nbytes, s->gathered, s->mine, d_hit, d_depth, d_normal, d_prim,
                                           stream) != hipSuccess)
            return fail(c, SDFGPU_ERR_HIP, "untile launch");
    }
    return SDFGPU_OK;
}

}  // extern "C"
