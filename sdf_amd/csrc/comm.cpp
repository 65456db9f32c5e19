// comm.cpp -- multi-GPU framebuffer assembly behind the C ABI: the tree replicated, a
// frame's 8x8 tiles dealt round-robin over the ranks, ONE exchange at the end of the frame: every
// rank's packed tile buffer travels to the assembling rank with a grouped ncclSend / ncclRecv (one
// point-to-point transfer per peer, each on its own xGMI link; nothing inside the march loop), and
// that rank scatters the tiles into its row-major planes.  Two shapes of host are served: one
// process per GPU (sdfgpu_comm_create + sdfgpu_render_frame_comm, the unique id shipped by the
// host's own means) and one process driving several GPUs (sdfgpu_comm_create_all over
// ncclCommInitAll + sdfgpu_render_frame_all, no id exchange).
//
// The reference has no multi-device code at all; this file has no counterpart there.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the library PyTorch-ROCm and /opt/rocm both
// ship under that soname): a process that already has RCCL loaded gets that copy, a single-GPU
// user never loads it, and libsdfgpu.so carries no link-time dependency on a 570 MB library.
//
// Collective safety.  A frame is a collective: a rank that returned early would leave the assembling
// rank's stream waiting in ncclRecv for ever.  So every step that can fail locally (argument checks,
// staging, the march of the rank's own tiles) comes BEFORE anything a peer waits on, and a rank whose
// local step failed still takes part in the exchange (its share travels as misses) and reports its
// error afterwards.  Only a rank that has no buffer to send at all (staging could not be allocated)
// cannot take part: it aborts the communicator (ncclCommAbort), so that its peers' pending operations
// end with an error instead of hanging, and every later call on that communicator fails at once.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
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
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
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
                        bind(r.handle, "ncclCommInitAll", &r.CommInitAll, &r.err) &&
                        bind(r.handle, "ncclCommDestroy", &r.CommDestroy, &r.err) &&
                        bind(r.handle, "ncclCommAbort", &r.CommAbort, &r.err) &&
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
constexpr size_t MAX_STAGING = 8;

struct Staging {
    hipStream_t stream = nullptr;
    void* mine = nullptr;       // this rank's packed tiles: [depth | normal | prim | hit], planes of `px` pixels
    void* gathered = nullptr;   // assembling rank: world x nbytes
    size_t nbytes = 0;
    uint64_t stamp = 0;
    // recorded behind the last operation that used this entry: what recycling waits for.  (The stream handle
    // itself may belong to a stream the caller has destroyed by then; the event is the communicator's own.)
    hipEvent_t done = nullptr;
};

}  // namespace

struct sdfgpu_comm {
    sdfgpu_ctx* ctx = nullptr;
    int device = 0;                 // cached: destroy must not reach through ctx (which may be gone)
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool dead = false;              // aborted after a local failure that made taking part impossible
    // Test / bring-up switch (SDFGPU_COMM_SELF_EXCHANGE=1 in the environment of the create call): the assembling
    // rank also sends its own share to itself and reads it back from the gather buffer, so the grouped
    // ncclSend / ncclRecv / untile-from-gathered path runs even in a communicator of one rank.
    bool self_exchange = false;
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
    if (s.done) (void)hipEventDestroy(s.done);
    s = Staging();
}

// Waits for the last use of an entry (its own event), not for the stream it was used on.
void quiesce(Staging& s) {
    if (s.done && hipEventSynchronize(s.done) != hipSuccess) (void)hipDeviceSynchronize();
}

bool capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

bool needs_gather(const sdfgpu_comm* m, bool root) { return root && (m->world > 1 || m->self_exchange); }

// Staging of (communicator, stream); the least recently used entry is recycled beyond MAX_STAGING streams
// (after waiting for the event recorded behind its last use).
int staging_for(sdfgpu_comm* m, hipStream_t stream, size_t nbytes, bool root, Staging** out) {
    Staging* s = nullptr;
    for (auto& e : m->staging)
        if (e.stream == stream && e.mine) s = &e;
    const bool fresh = !s;
    const bool resize = s && (s->nbytes != nbytes || (needs_gather(m, root) && !s->gathered));
    if ((fresh || resize) && capturing(stream))
        return fail(m->ctx, SDFGPU_ERR_INVALID_ARG,
                    "the first frame of a (communicator, stream, size) allocates staging: run it once outside stream capture");
    if (fresh) {
        if (m->staging.size() < MAX_STAGING) {
            m->staging.emplace_back();
            s = &m->staging.back();
        } else {
            s = &m->staging[0];
            for (auto& e : m->staging)
                if (e.stamp < s->stamp) s = &e;
            quiesce(*s);
            release(*s);
        }
        s->stream = stream;
    }
    if (fresh || resize) {
        quiesce(*s);
        release(*s);
        s->stream = stream;
        if (hipEventCreateWithFlags(&s->done, hipEventDisableTiming) != hipSuccess)
            return fail(m->ctx, SDFGPU_ERR_HIP, "hipEventCreate (tile staging)");
        if (hipMalloc(&s->mine, nbytes) != hipSuccess) {
            release(*s);
            return fail(m->ctx, SDFGPU_ERR_HIP, "hipMalloc (tile staging)");
        }
        if (needs_gather(m, root) && hipMalloc(&s->gathered, nbytes * (size_t)m->world) != hipSuccess) {
            release(*s);
            return fail(m->ctx, SDFGPU_ERR_HIP, "hipMalloc (gather staging)");
        }
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

struct FrameGeom {
    uint32_t tiles, world;
    size_t px, nbytes;
    uint32_t count_of(uint32_t r) const { return r < tiles ? (tiles - r + world - 1) / world : 0u; }
};
FrameGeom geom_of(uint32_t w, uint32_t h, uint32_t world) {
    FrameGeom g;
    g.tiles = ((w + 7) / 8) * ((h + 7) / 8);
    g.world = world;
    g.px = (size_t)((g.tiles + world - 1) / world) * TILE_PIXELS;   // pixels of the largest share
    g.nbytes = g.px * BYTES_PER_PIXEL;
    return g;
}

// What cannot differ between the ranks of a frame and needs nothing from the device.
int check_frame_args(sdfgpu_ctx* c, const sdfgpu_comm* m, const sdfgpu_camera* cam, uint32_t w, uint32_t h,
                     const sdfgpu_march_params* mp, int root, const uint8_t* d_hit) {
    if (m->dead) return fail(c, SDFGPU_ERR_COMM, "communicator was aborted after an earlier local failure; create a new one");
    if (root < 0 || root >= m->world) return fail(c, SDFGPU_ERR_INVALID_ARG, "root out of range");
    if (!cam || !mp || w == 0 || h == 0) return fail(c, SDFGPU_ERR_INVALID_ARG, "null camera / march params or empty image");
    if ((uint64_t)w * h > (1ull << 30)) return fail(c, SDFGPU_ERR_TOO_LARGE, "image larger than 2^30 pixels");
    if (m->rank == root && !d_hit) return fail(c, SDFGPU_ERR_INVALID_ARG, "null hit buffer on the assembling rank");
    return SDFGPU_OK;
}

// The rank can no longer take part in a frame its peers may already be waiting in: end the communicator so
// that their pending operations fail instead of hanging.
void abort_comm(sdfgpu_comm* m) {
    if (m->world == 1 && !m->self_exchange) return;   // nobody is waiting for a communicator of one: the error code is enough
    if (m->comm) (void)rccl().CommAbort(m->comm);
    m->comm = nullptr;
    m->dead = true;
}

// Step 1, local: staging, then this rank's tiles r, r + world, ... into its packed buffer.  *render_rc keeps a
// failure of the march (the exchange still happens, with the share cleared to misses).  A non-zero return means
// the rank has nothing to send; the communicator is aborted then.
int frame_local(sdfgpu_comm* m, const sdfgpu_camera* cam, uint32_t w, uint32_t h, const sdfgpu_march_params* mp, int root,
                hipStream_t stream, const FrameGeom& g, Staging** s_out, int* render_rc) {
    *render_rc = SDFGPU_OK;
    if (hipSetDevice(m->device) != hipSuccess) {
        abort_comm(m);
        return fail(m->ctx, SDFGPU_ERR_HIP, "hipSetDevice");
    }
    Staging* s = nullptr;
    int rc = staging_for(m, stream, g.nbytes, m->rank == root, &s);
    if (rc) {
        abort_comm(m);
        return rc;
    }
    const Planes mine = planes_of(s->mine, g.px);
    rc = sdfgpu_render_tiles_dev(m->ctx, cam, w, h, mp, (uint32_t)m->rank, g.world, g.count_of((uint32_t)m->rank), mine.hit,
                                 mine.depth, mine.normal, mine.prim, stream);
    if (rc) {
        *render_rc = rc;
        (void)hipMemsetAsync(s->mine, 0, g.nbytes, stream);   // hit = 0 everywhere: the share travels as misses
    }
    *s_out = s;
    return SDFGPU_OK;
}

// Step 2, inside an open group: the assembling rank receives every peer's buffer, every peer sends its own.
ncclResult_t frame_exchange(sdfgpu_comm* m, Staging* s, int root, hipStream_t stream, const FrameGeom& g) {
    Rccl& r = rccl();
    ncclResult_t e = ncclSuccess;
    const bool is_root = m->rank == root;
    if (is_root) {
        for (uint32_t p = 0; p < g.world && e == ncclSuccess; ++p)
            if ((int)p != m->rank || m->self_exchange)
                e = r.Recv((char*)s->gathered + (size_t)p * g.nbytes, g.nbytes, ncclUint8, (int)p, m->comm, stream);
    }
    if (e == ncclSuccess && (!is_root || m->self_exchange)) e = r.Send(s->mine, g.nbytes, ncclUint8, root, m->comm, stream);
    return e;
}

// Step 3, assembling rank: tile-compact -> row-major, every rank's share in one launch.
int frame_assemble(sdfgpu_comm* m, Staging* s, uint32_t w, uint32_t h, const FrameGeom& g, uint8_t* d_hit, float* d_depth,
                   float* d_normal, uint32_t* d_prim, hipStream_t stream) {
    if (hipSetDevice(m->device) != hipSuccess) return fail(m->ctx, SDFGPU_ERR_HIP, "hipSetDevice");
    const void* own = m->self_exchange ? (const void*)((const char*)s->gathered + (size_t)m->rank * g.nbytes) : s->mine;
    if (sdfgpu::launch_untile_gathered(w, h, g.world, (uint32_t)m->rank, g.px, g.nbytes, s->gathered, own, d_hit, d_depth, d_normal,
                                       d_prim, stream) != hipSuccess)
        return fail(m->ctx, SDFGPU_ERR_HIP, "untile launch");
    return SDFGPU_OK;
}

sdfgpu_comm* new_comm(sdfgpu_ctx* c, ncclComm_t comm, int rank, int world) {
    sdfgpu_comm* m = new sdfgpu_comm();
    m->ctx = c;
    m->device = sdfgpu::ctx_device(c);
    m->comm = comm;
    m->rank = rank;
    m->world = world;
    const char* e = std::getenv("SDFGPU_COMM_SELF_EXCHANGE");
    m->self_exchange = e && std::atoi(e) != 0;
    return m;
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
    *out = new_comm(c, comm, rank, world);
    return SDFGPU_OK;
}

int sdfgpu_comm_create_all(sdfgpu_ctx* const* ctxs, int n, sdfgpu_comm** out) {
    if (!ctxs || !out || n < 1) return SDFGPU_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return SDFGPU_ERR_INVALID_ARG;
    sdfgpu_ctx* c0 = ctxs[0];
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        devs[(size_t)i] = sdfgpu::ctx_device(ctxs[i]);
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i])
                return fail(c0, SDFGPU_ERR_INVALID_ARG, "two contexts of one communicator clique on the same device (RCCL needs one device per rank)");
    }
    Rccl& r = rccl();
    if (!r.handle) return fail(c0, SDFGPU_ERR_COMM, r.err);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    ncclResult_t e = r.CommInitAll(comms.data(), n, devs.data());
    if (e != ncclSuccess) return nccl_fail(c0, e, "ncclCommInitAll");
    for (int i = 0; i < n; ++i) out[i] = new_comm(ctxs[i], comms[(size_t)i], i, n);
    return SDFGPU_OK;
}

void sdfgpu_comm_destroy(sdfgpu_comm* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (auto& s : m->staging) {
        quiesce(s);
        release(s);
    }
    if (m->comm) (void)rccl().CommDestroy(m->comm);
    delete m;
}

int sdfgpu_render_frame_comm(sdfgpu_ctx* c, sdfgpu_comm* m, const sdfgpu_camera* cam, uint32_t w, uint32_t h,
                             const sdfgpu_march_params* mp, int root, uint8_t* d_hit, float* d_depth, float* d_normal,
                             uint32_t* d_prim, void* stream_) {
    if (!c || !m || m->ctx != c) return SDFGPU_ERR_INVALID_ARG;
    int rc = check_frame_args(c, m, cam, w, h, mp, root, d_hit);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const FrameGeom g = geom_of(w, h, (uint32_t)m->world);
    Staging* s = nullptr;
    int render_rc = SDFGPU_OK;
    if ((rc = frame_local(m, cam, w, h, mp, root, stream, g, &s, &render_rc))) return rc;
    if (m->world > 1 || m->self_exchange) {
        Rccl& r = rccl();
        ncclResult_t e = r.GroupStart();
        if (e != ncclSuccess) {
            abort_comm(m);
            return nccl_fail(c, e, "ncclGroupStart");
        }
        e = frame_exchange(m, s, root, stream, g);
        const ncclResult_t e2 = r.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess) {
            abort_comm(m);
            return e != ncclSuccess ? nccl_fail(c, e, "ncclSend / ncclRecv") : nccl_fail(c, e2, "ncclGroupEnd");
        }
    }
    if (m->rank == root && (rc = frame_assemble(m, s, w, h, g, d_hit, d_depth, d_normal, d_prim, stream))) return rc;
    (void)hipEventRecord(s->done, stream);
    return render_rc;   // the frame was exchanged either way; a failed march of this rank's share is reported here
}

int sdfgpu_render_frame_all(sdfgpu_comm* const* comms, int n, const sdfgpu_camera* cam, uint32_t w, uint32_t h,
                            const sdfgpu_march_params* mp, int root, uint8_t* d_hit, float* d_depth, float* d_normal,
                            uint32_t* d_prim, void* const* streams) {
    if (!comms || n < 1) return SDFGPU_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!comms[i] || comms[i]->world != n || comms[i]->rank != i) return SDFGPU_ERR_INVALID_ARG;
    if (root < 0 || root >= n) return fail(comms[0]->ctx, SDFGPU_ERR_INVALID_ARG, "root out of range");
    int rc;
    for (int i = 0; i < n; ++i)
        if ((rc = check_frame_args(comms[i]->ctx, comms[i], cam, w, h, mp, root, d_hit))) return rc;
    const FrameGeom g = geom_of(w, h, (uint32_t)n);
    std::vector<Staging*> st((size_t)n, nullptr);
    int first_rc = SDFGPU_OK;
    auto stream_of = [&](int i) { return streams ? (hipStream_t)streams[i] : (hipStream_t) nullptr; };
    // every rank's local step first; one that cannot take part ends the frame for all (nothing is waiting yet)
    for (int i = 0; i < n; ++i) {
        int render_rc = SDFGPU_OK;
        if ((rc = frame_local(comms[i], cam, w, h, mp, root, stream_of(i), g, &st[(size_t)i], &render_rc))) {
            for (int j = 0; j < n; ++j)
                if (j != i && !comms[j]->dead) abort_comm(comms[j]);
            return rc;
        }
        if (render_rc && !first_rc) first_rc = render_rc;
    }
    if (n > 1 || comms[0]->self_exchange) {
        Rccl& r = rccl();
        ncclResult_t e = r.GroupStart();   // ONE group over all devices: a thread that drives several ranks must not let one rank's launch wait for another's
        for (int i = 0; i < n && e == ncclSuccess; ++i) e = frame_exchange(comms[i], st[(size_t)i], root, stream_of(i), g);
        const ncclResult_t e2 = r.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess) {
            for (int j = 0; j < n; ++j) abort_comm(comms[j]);
            return e != ncclSuccess ? nccl_fail(comms[root]->ctx, e, "ncclSend / ncclRecv") : nccl_fail(comms[root]->ctx, e2, "ncclGroupEnd");
        }
    }
    if ((rc = frame_assemble(comms[root], st[(size_t)root], w, h, g, d_hit, d_depth, d_normal, d_prim, stream_of(root)))) return rc;
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(comms[i]->device);
        (void)hipEventRecord(st[(size_t)i]->done, stream_of(i));
    }
    return first_rc;
}

}  // extern "C"
