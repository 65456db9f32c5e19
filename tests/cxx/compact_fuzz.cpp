// compact_fuzz.cpp -- test program (tests/test_compact_fuzz.py builds it with AddressSanitizer and UBSan).
// Upload validation is what lets the kernels run without bounds checks, so it must survive ANY bytes:
// valid buffers are corrupted at random (structure fields, flags, op codes, matrix entries, whole blocks)
// and handed to compact_tree in every bounds mode and extension mask.  Whatever it answers, it must not
// read or write out of bounds; and when it accepts a buffer, every reference of the resident tree must
// point inside the resident arrays and the stack depth must cover the tree.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "sdfgpu.h"
#include "sdfhost.h"
#include "gpu_tree.hpp"

using namespace sdfgpu;

static int failures = 0;
#define CHECK(cond, ...)                                     \
    do {                                                     \
        if (!(cond)) {                                       \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                        \
            std::printf("\n");                               \
            ++failures;                                      \
        }                                                    \
    } while (0)

// depth of the resident hierarchy below `ref`, checking every reference on the way; -1 on a bad reference
static int walk(const CompactTree& t, uint32_t ref, int budget) {
    if (budget < 0) return -1;   // deeper than anything the builder may emit, or a cycle
    if (ref == REF_NONE) return -1;
    if (ref & REF_LEAF) return ((ref & REF_INDEX) < t.leaves.size()) ? 0 : -1;
    uint32_t r0, r1;
    if (!t.inner_obb.empty()) {
        if (ref >= t.inner_obb.size()) return -1;
        r0 = t.inner_obb[ref].ref0;
        r1 = t.inner_obb[ref].ref1;
    } else {
        if (ref >= t.inner.size()) return -1;
        r0 = t.inner[ref].ref0;
        r1 = t.inner[ref].ref1;
    }
    const int d0 = walk(t, r0, budget - 1), d1 = r1 == REF_NONE ? 0 : walk(t, r1, budget - 1);
    if (d0 < 0 || d1 < 0) return -1;
    return 1 + (d0 > d1 ? d0 : d1);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    std::mt19937_64 rng(0xC0FFEE);
    auto rnd = [&](uint64_t n) { return (uint64_t)(rng() % n); };
    const float specials[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, NAN, 1e30f, -1e30f, 1e-30f, 3.4e38f};
    int accepted = 0, rejected = 0;
    for (uint32_t n_prims : {1u, 2u, 5u, 40u, 300u}) {
        sdfhost_node* node = sdfhost_random_scene(n_prims, 0xF00Dull + n_prims, 0.f, nullptr);
        sdfhost_expanded* ex = sdfhost_expanded_new(node, nullptr);
        sdfhost_buffer* buf = sdfhost_make_buffer(ex, nullptr);
        const uint32_t len = sdfhost_buffer_len(buf);
        const std::vector<SdfOperationBlock> dt0(sdfhost_buffer_downtree(buf), sdfhost_buffer_downtree(buf) + len);
        const std::vector<SdfOperationUptreeBlock> ut0(sdfhost_buffer_uptree(buf), sdfhost_buffer_uptree(buf) + len);
        for (int it = 0; it < iters; ++it) {
            std::vector<SdfOperationBlock> dt = dt0;
            std::vector<SdfOperationUptreeBlock> ut = ut0;
            const int n_hits = 1 + (int)rnd(3);
            for (int h = 0; h < n_hits; ++h) {
                const uint32_t i = (uint32_t)rnd(len);
                switch (rnd(10)) {
                    case 0: dt[i].level = (uint32_t)rnd(300); break;
                    case 1: dt[i].len = (uint32_t)rnd(2 * len + 2); break;
                    case 2: dt[i].op_code = (uint32_t)rnd(8); break;
                    case 3: dt[i].is_primitive = (uint8_t)rnd(3); break;
                    case 4: dt[i].parent_is_union = (uint8_t)rnd(3); break;
                    case 5: dt[i].bounding_box.trans_inverse.m[rnd(16)] = specials[rnd(sizeof specials / sizeof *specials)]; break;
                    case 6: dt[i].op_specific.floats[0] = specials[rnd(sizeof specials / sizeof *specials)]; break;
                    case 7: ut[i].level = (uint32_t)rnd(300); break;
                    case 8: ut[i].op_code = (uint32_t)rnd(8); break;
                    case 9: {   // a whole block of random bytes
                        unsigned char* b = reinterpret_cast<unsigned char*>(&dt[i]);
                        for (size_t k = 0; k < sizeof(SdfOperationBlock); ++k) b[k] = (unsigned char)rng();
                        break;
                    }
                }
            }
            const uint32_t bounds = (uint32_t)rnd(6);        // 0..4 valid, 5 unknown
            const uint32_t ext = (uint32_t)rnd(9);           // 0..7 valid masks, 8 unknown
            const uint32_t use_len = rnd(8) == 0 ? (uint32_t)rnd(len + 1) : len;   // sometimes a truncated view
            CompactTree t;
            std::string err;
            const int rc = compact_tree(dt.data(), ut.data(), use_len, SDFGPU_LAYOUT_GLAM_SSE2, bounds, ext, &t, &err);
            if (rc != SDFGPU_OK) {
                ++rejected;
                CHECK(!err.empty(), "error code %d without a message", rc);
                continue;
            }
            ++accepted;
            if (use_len == 0) continue;
            if (!t.gen.empty()) {   // general path: pre-order array, parents inside it
                for (size_t g = 0; g < t.gen.size(); ++g)
                    CHECK(t.gen[g].parent == REF_NONE || t.gen[g].parent < g, "general node %zu: parent %u", g, t.gen[g].parent);
                CHECK(t.gen.size() <= use_len, "general path holds more nodes than the buffer");
            } else if (t.root_ref != REF_NONE) {
                const int d = walk(t, t.root_ref, 70);
                CHECK(d >= 0, "accepted tree holds a reference outside the resident arrays (or deeper than 70)");
                CHECK(d < 0 || (uint32_t)d <= t.stack_depth, "tree depth %d beyond the declared stack depth %u", d, t.stack_depth);
                CHECK(t.stack_depth <= 64, "stack depth %u beyond the 64-entry wave stack", t.stack_depth);
            }
        }
        sdfhost_buffer_free(buf);
        sdfhost_expanded_free(ex);
        sdfhost_node_free(node);
    }
    std::printf("accepted %d rejected %d\n", accepted, rejected);
    if (failures == 0) std::printf("ok\n");
    return failures ? 1 : 0;
}
