// build_threads.cpp -- test program (tests/test_build_threads.py compiles and runs it on the CPU).
// Neither the flattened buffers nor the resident tree may depend on how many threads built them: compact_tree is run on the same
// buffers with SDFGPU_BUILD_THREADS = 1, 2, 5 and 16 and every array and scalar is compared byte for
// byte; and the first error reported must be the one with the lowest node index, whichever pass
// (the threaded matrix check or the sequential structural pass) finds it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sdfgpu.h"
#include "sdfhost.h"
#include "gpu_tree.hpp"
#include "host/sdf_host.hpp"

using namespace sdfgpu;

static int failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                      \
            std::printf("\n");                             \
            ++failures;                                    \
        }                                                  \
    } while (0)

template <class T>
static bool same_bytes(const std::vector<T>& a, const std::vector<T>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}

static bool same_tree(const CompactTree& a, const CompactTree& b) {
    return same_bytes(a.inner, b.inner) && same_bytes(a.leaves, b.leaves) && same_bytes(a.inner_obb, b.inner_obb) &&
           a.root_ref == b.root_ref && a.stack_depth == b.stack_depth &&
           std::memcmp(&a.box_scale, &b.box_scale, 4) == 0 && std::memcmp(&a.margin_T, &b.margin_T, 4) == 0 &&
           std::memcmp(&a.mean_extent, &b.mean_extent, 4) == 0 && std::memcmp(&a.lipschitz, &b.lipschitz, 4) == 0 &&
           std::memcmp(&a.info, &b.info, sizeof a.info) == 0;
}

int main() {
    const uint32_t sizes[] = {1u, 2u, 37u, 5000u, 40000u};   // 40000: both sides of the top levels get threads
    for (uint32_t n : sizes) {
        sdfhost_node* node = sdfhost_random_scene(n, 0xB17D0000ull + n, 0.f, nullptr);
        sdfhost_expanded* ex = sdfhost_expanded_new(node, nullptr);
        sdfhost_buffer* buf = sdfhost_make_buffer(ex, nullptr);
        const uint32_t len = sdfhost_buffer_len(buf);
        if (n >= 5000u) {
            // the host side: expansion (SdfUnion::expand) and flattening (make_buffer) fork their upper levels too
            for (const char* th : {"1", "3", "16"}) {
                setenv("SDFGPU_BUILD_THREADS", th, 1);
                sdfhost_expanded* ex2 = sdfhost_expanded_new(node, nullptr);
                sdfhost_buffer* buf2 = sdfhost_make_buffer(ex2, nullptr);
                CHECK(sdfhost_buffer_len(buf2) == len &&
                          std::memcmp(sdfhost_buffer_downtree(buf2), sdfhost_buffer_downtree(buf), (size_t)len * sizeof(SdfOperationBlock)) == 0 &&
                          std::memcmp(sdfhost_buffer_uptree(buf2), sdfhost_buffer_uptree(buf), (size_t)len * sizeof(SdfOperationUptreeBlock)) == 0,
                      "n=%u: buffers flattened by %s threads differ", n, th);
                sdfhost_buffer_free(buf2);
                sdfhost_expanded_free(ex2);
            }
        }
        for (uint32_t bounds : {(uint32_t)SDFGPU_BOUNDS_AUTO, (uint32_t)SDFGPU_BOUNDS_AABB}) {
            CompactTree ref;
            std::string err;
            setenv("SDFGPU_BUILD_THREADS", "1", 1);
            int rc = compact_tree(sdfhost_buffer_downtree(buf), sdfhost_buffer_uptree(buf), len, 0, bounds, 0u, &ref, &err);
            CHECK(rc == SDFGPU_OK, "n=%u bounds=%u: %s", n, bounds, err.c_str());
            for (const char* th : {"2", "5", "16"}) {
                setenv("SDFGPU_BUILD_THREADS", th, 1);
                CompactTree t;
                rc = compact_tree(sdfhost_buffer_downtree(buf), sdfhost_buffer_uptree(buf), len, 0, bounds, 0u, &t, &err);
                CHECK(rc == SDFGPU_OK, "n=%u bounds=%u threads=%s: %s", n, bounds, th, err.c_str());
                CHECK(same_tree(ref, t), "n=%u bounds=%u: tree built by %s threads differs from the one built by 1", n, bounds, th);
            }
        }
        if (n == 40000u) {
            // lowest index wins, whichever pass finds it
            std::vector<SdfOperationBlock> dt(sdfhost_buffer_downtree(buf), sdfhost_buffer_downtree(buf) + len);
            setenv("SDFGPU_BUILD_THREADS", "8", 1);
            CompactTree t;
            std::string err;
            dt[30001].bounding_box.trans_inverse.m[5] = NAN;
            dt[20001].bounding_box.trans_inverse.m[0] = INFINITY;
            dt[25000].is_primitive = 7;
            int rc = compact_tree(dt.data(), sdfhost_buffer_uptree(buf), len, 0, SDFGPU_BOUNDS_AUTO, 0u, &t, &err);
            CHECK(rc == SDFGPU_ERR_NON_FINITE && err.find("node 20001") != std::string::npos, "got %d '%s'", rc, err.c_str());
            dt[10000].parent_is_union = 2;
            rc = compact_tree(dt.data(), sdfhost_buffer_uptree(buf), len, 0, SDFGPU_BOUNDS_AUTO, 0u, &t, &err);
            CHECK(rc == SDFGPU_ERR_BAD_TREE && err.find("node 10000") != std::string::npos, "got %d '%s'", rc, err.c_str());
        }
        sdfhost_buffer_free(buf);
        sdfhost_expanded_free(ex);
        sdfhost_node_free(node);
    }
    {
        // A node that cannot be expanded (a clone without its child, elements.rs:260-262 unwraps slot 0), deep
        // inside a union big enough for the expansion to fork: the error has to come back to the caller as an
        // exception from whichever thread met it, not end the process there.
        using namespace sdfhost;
        setenv("SDFGPU_BUILD_THREADS", "8", 1);
        SdfNode root(std::make_unique<SdfUnion>(0.f));
        for (int i = 0; i < 6000; ++i) {
            Transform t;
            t.translation = Vec3((float)(i % 100) * 3.f, (float)((i / 100) % 60) * 3.f, (float)(i % 7));
            if (i == 4321) {
                SdfNode clone(std::make_unique<SdfCaaClone>(Vec3(4.f, 4.f, 4.f), Vec3(-1.f, -1.f, -1.f), Vec3(1.f, 1.f, 1.f)));
                clone.set_slot(SdfBuilder::primitive(std::make_unique<SdfSphere>(1.f)).transform(t).finalize());
                root.set_slot(std::move(clone));
            } else {
                root.set_slot(SdfBuilder::primitive(std::make_unique<SdfSphere>(1.f)).transform(t).finalize());
            }
        }
        root.calc_bbox_assign();
        root.slots[4321].slots.clear();   // boxes are in place; now take the clone's child away
        bool threw = false;
        try {
            ExpandedSdfNode ex = root.expanded();
        } catch (const std::exception& e) {
            threw = true;
        }
        CHECK(threw, "expanding a union that holds an unexpandable node did not report an error");
    }
    if (failures == 0) std::printf("ok\n");
    return failures ? 1 : 0;
}
