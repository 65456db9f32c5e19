// sdf_host.cpp -- host-side scene builder (see sdf_host.hpp for the map to
// the crate).  Runs once per scene; not on the GPU hot path.
#include "sdf_host.hpp"

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <exception>
#include <numeric>
#include <system_error>
#include <thread>

#include <sched.h>

namespace sdfhost {

// Flattening is host work on the changing part of a scene (a rebuilt tree is re-expanded and re-flattened),
// and both halves of it are recursions over independent subtrees: the two sides of the upper levels run
// on separate threads.  Every node's arithmetic is its own, so the buffers are the same bytes whatever
// the thread count (tests/test_host_builder.py).  Threads: the CPUs this process may run on, at most 16;
// SDFGPU_BUILD_THREADS overrides (the same variable the upload in compact.cpp reads).
static int host_forks() {
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)CPU_COUNT(&set));
    if (const char* e = std::getenv("SDFGPU_BUILD_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));
    hw = std::max(1u, std::min(hw, 16u));
    int f = 0;
    while ((2u << f) <= hw) ++f;
    return f;
}

// run a() on a new thread beside b() when a thread can be had, else one after the other.  An exception
// of either side (an unset slot, an unfinished node: the builder reports those by throwing) is carried
// back to the caller after both sides have finished; a()'s wins when both throw, as in the serial order.
template <class A, class B>
static void fork_join(bool parallel, A&& a, B&& b) {
    if (parallel) {
        std::exception_ptr ea, eb;
        std::thread t;
        bool started = false;
        try {
            t = std::thread([&a, &ea] {
                try {
                    a();
                } catch (...) {
                    ea = std::current_exception();
                }
            });
            started = true;
        } catch (const std::system_error&) {
        }
        if (started) {
            try {
                b();
            } catch (...) {
                eb = std::current_exception();
            }
            t.join();
            if (ea) std::rethrow_exception(ea);
            if (eb) std::rethrow_exception(eb);
            return;
        }
    }
    a();
    b();
}

// ---------------------------------------------------------------- math ----

static bool invert3(const double a[3][3], double inv[3][3]) {
    double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
    double c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2];
    double c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    if (det == 0.0 || !std::isfinite(det)) return false;
    double r = 1.0 / det;
    inv[0][0] = c00 * r;
    inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * r;
    inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * r;
    inv[1][0] = c01 * r;
    inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * r;
    inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * r;
    inv[2][0] = c02 * r;
    inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * r;
    inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * r;
    return true;
}

bool invert(const Mat4& a, Mat4* out) {
    if (a.is_affine()) {
        double A[3][3], I[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) A[r][c] = a.at(r, c);
        if (!invert3(A, I)) return false;
        Mat4 o = Mat4::identity();
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) o.at(r, c) = (float)I[r][c];
            double t = -(I[r][0] * a.at(0, 3) + I[r][1] * a.at(1, 3) + I[r][2] * a.at(2, 3));
            o.at(r, 3) = (float)t;
        }
        *out = o;
        return true;
    }
    // general cofactor expansion (double)
    double m[16], inv[16];
    for (int i = 0; i < 16; ++i) m[i] = a.m[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0 || !std::isfinite(det)) return false;
    double r = 1.0 / det;
    for (int i = 0; i < 16; ++i) out->m[i] = (float)(inv[i] * r);
    return true;
}

Mat4 Transform::compute_matrix() const {
    const Quat& q = rotation;
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
    float yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    Mat4 r = Mat4::zeros();
    r.m[0] = (1.f - (yy + zz)) * scale.x; r.m[1] = (xy + wz) * scale.x; r.m[2] = (xz - wy) * scale.x;
    r.m[4] = (xy - wz) * scale.y; r.m[5] = (1.f - (xx + zz)) * scale.y; r.m[6] = (yz + wx) * scale.y;
    r.m[8] = (xz + wy) * scale.z; r.m[9] = (yz - wx) * scale.z; r.m[10] = (1.f - (xx + yy)) * scale.z;
    r.m[12] = translation.x; r.m[13] = translation.y; r.m[14] = translation.z; r.m[15] = 1.f;
    return r;
}

void jacobi_eigen3(const double a_in[3][3], double V[3][3], double w[3]) {
    double a[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { a[i][j] = a_in[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = a[i][i];
}

// ------------------------------------------------------ SdfBoundingBox ----

static const float VERT_LIST[8][3] = {  // obb.rs:13-22
    {1, 1, 1}, {-1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {1, 1, -1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};

SdfBoundingBox SdfBoundingBox::unit() {
    SdfBoundingBox b;
    b.matrix = Mat4::identity();
    b.scale = Vec4(1, 1, 1, 0);
    b.full_inverse = Mat4::identity();
    b.trans_inverse = Mat4::identity();
    return b;
}

SdfBoundingBox SdfBoundingBox::zero() {
    SdfBoundingBox b;
    b.matrix = Mat4::zeros();
    b.scale = Vec4(0, 0, 0, 0);
    b.full_inverse = Mat4::zeros();
    b.full_inverse.m[0] = b.full_inverse.m[5] = b.full_inverse.m[10] = INFINITY;
    b.full_inverse.m[15] = 1.f;
    b.trans_inverse = Mat4::identity();
    return b;
}

SdfBoundingBox SdfBoundingBox::merge(const std::vector<SdfBoundingBox>& sub, const BuildOptions& o) {
    if (sub.empty()) return zero();                              // obb.rs:118-120
    const size_t nv = sub.size() * 8;
    std::vector<double> v(nv * 3);
    double mean[3] = {0, 0, 0};
    double max_abs = 0.0;
    size_t k = 0;
    for (const auto& b : sub)                                    // obb.rs:122-130
        for (int i = 0; i < 8; ++i, ++k) {
            Vec4 c = b.matrix * Vec4(VERT_LIST[i][0], VERT_LIST[i][1], VERT_LIST[i][2], 1.f);
            v[3 * k] = c.x; v[3 * k + 1] = c.y; v[3 * k + 2] = c.z;
            mean[0] += c.x; mean[1] += c.y; mean[2] += c.z;
            max_abs = std::max({max_abs, std::fabs((double)c.x), std::fabs((double)c.y), std::fabs((double)c.z)});
        }
    for (double& m : mean) m /= (double)nv;                      // obb.rs:132
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (size_t i = 0; i < nv; ++i) {                            // obb.rs:134-140
        double d[3] = {v[3 * i] - mean[0], v[3 * i + 1] - mean[1], v[3 * i + 2] - mean[2]};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov[r][c] += d[r] * d[c];
    }
    const double denom = (double)(nv - 1);
    for (auto& row : cov) for (double& x : row) x /= denom;
    double E[3][3], ew[3];
    jacobi_eigen3(cov, E, ew);                                   // obb.rs:142 (symmetric_eigen)

    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};                 // obb.rs:158-165 (start at zero)
    for (size_t i = 0; i < nv; ++i) {
        double d[3] = {v[3 * i] - mean[0], v[3 * i + 1] - mean[1], v[3 * i + 2] - mean[2]};
        for (int a = 0; a < 3; ++a) {
            double p = d[0] * E[0][a] + d[1] * E[1][a] + d[2] * E[2][a];
            lo[a] = std::min(lo[a], p);
            hi[a] = std::max(hi[a], p);
        }
    }
    double cen[3], sc[3];
    double pad_abs = 16.0 * (double)FLT_EPSILON * (max_abs + 1.0);
    double smax = 0.0;
    for (int a = 0; a < 3; ++a) {                                // obb.rs:167-169
        cen[a] = (hi[a] + lo[a]) / 2.0;
        sc[a] = (hi[a] - cen[a]) * (1.0 + (double)o.merge_pad_rel) + pad_abs;
        smax = std::max(smax, sc[a]);
    }
    for (double& s : sc) s = std::max(s, 1e-6 * smax + 1e-12);   // flat boxes stay invertible

    // obb.rs:172-174: T(mean) * E * T(centroid) * S(scale)
    SdfBoundingBox b;
    b.matrix = Mat4::zeros();
    for (int a = 0; a < 3; ++a)
        for (int r = 0; r < 3; ++r) b.matrix.at(r, a) = (float)(E[r][a] * sc[a]);
    for (int r = 0; r < 3; ++r)
        b.matrix.at(r, 3) = (float)(mean[r] + E[r][0] * cen[0] + E[r][1] * cen[1] + E[r][2] * cen[2]);
    b.matrix.at(3, 3) = 1.f;
    b.scale = Vec4((float)sc[0], (float)sc[1], (float)sc[2], 0.f);
    if (!invert(b.matrix, &b.full_inverse))                       // obb.rs:180-182
        throw std::runtime_error("SdfBoundingBox::merge: singular box matrix");
    if (o.pca_trans_inverse) {                                   // obb.rs:183-185
        Mat4 te = Mat4::identity();
        for (int a = 0; a < 3; ++a)
            for (int r = 0; r < 3; ++r) te.at(r, a) = (float)E[r][a];
        for (int r = 0; r < 3; ++r) te.at(r, 3) = (float)mean[r];
        if (!invert(te, &b.trans_inverse))
            throw std::runtime_error("SdfBoundingBox::merge: singular eigenbasis");
    } else {
        b.trans_inverse = Mat4::identity();                      // SURVEY C-3
    }
    return b;
}

std::pair<std::vector<size_t>, std::vector<size_t>> SdfBoundingBox::split(
    const std::vector<SdfBoundingBox>& sub) const {
    std::vector<size_t> inds(sub.size());
    std::iota(inds.begin(), inds.end(), (size_t)0);
    float s2[3];
    for (int a = 0; a < 3; ++a) {                                // obb.rs:192-198
        Vec4 c = matrix.col(a);
        s2[a] = ((c.x * c.x + c.y * c.y) + c.z * c.z) + c.w * c.w;
    }
    int ax = (s2[0] >= s2[1] && s2[0] >= s2[2]) ? 0 : (s2[1] >= s2[2] ? 1 : 2);  // :200-208
    Vec4 c = matrix.col(ax);
    float n = std::sqrt(s2[ax]);
    Vec4 axis(c.x / n, c.y / n, c.z / n, c.w / n);
    std::vector<float> key(sub.size());
    for (size_t i = 0; i < sub.size(); ++i) {                    // obb.rs:210
        Vec4 t = sub[i].matrix.col(3);
        key[i] = ((t.x * axis.x + t.y * axis.y) + t.z * axis.z) + t.w * axis.w;
    }
    // sort_unstable_by_key leaves ties implementation-defined; here: by index.
    std::stable_sort(inds.begin(), inds.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    size_t half = inds.size() / 2;                               // obb.rs:212: split_off(len/2)
    std::vector<size_t> upper(inds.begin() + (long)half, inds.end());
    inds.resize(half);
    return {std::move(upper), std::move(inds)};
}

SdfBoundingBox SdfBoundingBox::from_transform(const Transform& t) { return unit().apply_transform(t); }

SdfBoundingBox SdfBoundingBox::apply_transform(const Transform& t) const {
    Transform no_scale = t;
    no_scale.scale = Vec3(1, 1, 1);                              // obb.rs:227-228
    Mat4 m = t.compute_matrix();
    Mat4 minv, nsinv;
    if (!invert(m, &minv) || !invert(no_scale.compute_matrix(), &nsinv))
        throw std::runtime_error("apply_transform: singular transform");
    SdfBoundingBox b;
    b.matrix = m * matrix;                                       // obb.rs:229-231
    b.full_inverse = full_inverse * minv;                        // obb.rs:232-234
    b.trans_inverse = trans_inverse * nsinv;                     // obb.rs:235-237
    b.scale = Vec4(scale.x * t.scale.x, scale.y * t.scale.y, scale.z * t.scale.z, scale.w * 0.f);  // :240
    return b;
}

bool SdfBoundingBox::is_zero() const {
    for (int a = 0; a < 3; ++a) {
        Vec4 c = matrix.col(a);
        float l = std::sqrt((c.x * c.x + c.y * c.y) + c.z * c.z);
        if (!(l <= FLT_EPSILON)) return false;
    }
    return true;
}

float SdfBoundingBox::distance_to(Vec3 p) const {
    Vec4 t = in_box_basis(Vec4(p, 1.f));
    Vec4 q((std::fabs(t.x) - 1.f) * scale.x, (std::fabs(t.y) - 1.f) * scale.y,
           (std::fabs(t.z) - 1.f) * scale.z, (std::fabs(t.w) - 1.f) * scale.w);
    float mx = std::max(q.x, 0.f), my = std::max(q.y, 0.f), mz = std::max(q.z, 0.f), mw = std::max(q.w, 0.f);
    float l = std::sqrt((mx * mx + mz * mz) + (my * my + mw * mw));
    return l + std::fmin(std::fmax(std::fmax(q.y, q.z), q.x), 0.f);
}

float SdfBoundingBox::max_distance(Vec3 p) const {
    float best = 0.f;
    for (int i = 0; i < 8; ++i) {
        Vec4 c = matrix * Vec4(VERT_LIST[i][0], VERT_LIST[i][1], VERT_LIST[i][2], 1.f);
        float dx = c.x - p.x, dy = c.y - p.y, dz = c.z - p.z, dw = c.w - 1.f;
        float l = std::sqrt(((dx * dx + dy * dy) + dz * dz) + dw * dw);
        best = (i == 0 || l >= best) ? l : best;
    }
    return best;
}

bool SdfBoundingBox::contains(Vec3 p) const {
    Vec4 t = full_inverse * Vec4(p, 1.f);
    return std::max({std::fabs(t.x), std::fabs(t.y), std::fabs(t.z)}) <= 1.0f;
}

SdfBoundingBoxBlock SdfBoundingBox::get_bbox_block() const {
    SdfBoundingBoxBlock b;
    std::memset(&b, 0, sizeof b);
    std::memcpy(b.matrix.m, matrix.m, 64);
    b.scale.v[0] = scale.x; b.scale.v[1] = scale.y; b.scale.v[2] = scale.z; b.scale.v[3] = scale.w;
    std::memcpy(b.full_inverse.m, full_inverse.m, 64);
    std::memcpy(b.trans_inverse.m, trans_inverse.m, 64);
    return b;
}

// ------------------------------------------------------------ elements ----

static SdfOpSpecificBlock zero_block() {
    SdfOpSpecificBlock b;
    std::memset(&b, 0, sizeof b);
    return b;
}
SdfOpSpecificBlock SdfElement::get_dt_specific_block() const { return zero_block(); }
SdfOpSpecificBlock SdfElement::get_ut_specific_block() const { return zero_block(); }

SdfBoundingBox SdfSphere::get_bbox(const std::vector<SdfBoundingBox>&, const BuildOptions&) const {
    return SdfBoundingBox::from_transform(Transform::from_scale(Vec3::splat(radius)));   // elements.rs:95-97
}
SdfOpSpecificBlock SdfSphere::get_dt_specific_block() const {
    SdfOpSpecificBlock b = zero_block();
    b.floats[0] = radius;                                        // elements.rs:109-113
    return b;
}
ExpandedSdfNode SdfSphere::expand(const SdfNode& node, const BuildOptions&) const {
    return ExpandedSdfNode::primitive(node.bbox.value(), clone());   // elements.rs:115-117
}

SdfBoundingBox SdfBox::get_bbox(const std::vector<SdfBoundingBox>&, const BuildOptions&) const {
    return SdfBoundingBox::from_transform(Transform::from_scale(half));
}
// ---- recursive query, node.rs:250-267, 331-368 -------------------------------
bool SdfNode::could_contain(Vec3 p) const {
    if (!is_finished()) throw std::logic_error("Tried testing point-in-bbox on an unfinished SDF node!");
    return bbox->contains(p);
}

NodeDistInfo SdfNode::bbox_dist_info(Vec3 p) const {
    if (!is_finished()) throw std::logic_error("Tried getting bounding-box distance info on an unfinished SDF node!");
    return NodeDistInfo{bbox->distance_to(p), bbox->max_distance(p)};
}

namespace {
// CmpFloat (obb.rs:26-41): partial_cmp().unwrap_or(Equal) -- a NaN compares equal to everything.
inline bool cmpfloat_less(float a, float b) { return a < b; }
}  // namespace

NnResult SdfNode::nearest_neighbor(Vec3 p) const {
    if (is_primitive()) {   // :332-337
        const Vec4 q = bbox.value().in_box_trans_basis(Vec4(p, 1.f));
        return NnResult{this, intern->distance_to(Vec3(q.x, q.y, q.z))};
    }
    // :338-341 bounds of every slot at the point as given to this node
    std::vector<std::pair<size_t, NodeDistInfo>> bounds;
    bounds.reserve(slots.size());
    for (size_t i = 0; i < slots.size(); ++i) bounds.emplace_back(i, slots[i].bbox_dist_info(p));
    if (bounds.empty()) throw std::logic_error("nearest_neighbor on an operation without slots");   // .min().unwrap()
    // :342-344 smallest MAXDIST (Iterator::min keeps the FIRST of equal minima)
    float min_maxdist = bounds[0].second.max_bound;
    for (size_t i = 1; i < bounds.size(); ++i)
        if (cmpfloat_less(bounds[i].second.max_bound, min_maxdist)) min_maxdist = bounds[i].second.max_bound;
    // :345 sort_unstable_by_key(min_bound): order among equal keys is unspecified in the crate; here by slot index
    std::stable_sort(bounds.begin(), bounds.end(), [](const auto& a, const auto& b) {
        return cmpfloat_less(a.second.min_bound, b.second.min_bound);
    });
    // :346-367 take_while(min_bound < min_maxdist), fold keeping the strictly smaller distance
    NnResult accum{this, INFINITY};
    const Vec4 q = bbox.value().in_box_trans_basis(Vec4(p, 1.f));
    const Vec3 local(q.x, q.y, q.z);
    for (const auto& [i, bound] : bounds) {
        if (!(bound.min_bound < min_maxdist)) break;
        if (bound.min_bound > accum.distance) continue;
        const SdfNode& child = slots[i];
        const NnResult child_nn = child.nearest_neighbor(child.downtree(local));
        if (child_nn.distance < accum.distance) accum = child_nn;
    }
    return accum;
}

float SdfBox::distance_to(Vec3 p) const {
    float qx = std::fabs(p.x) - half.x, qy = std::fabs(p.y) - half.y, qz = std::fabs(p.z) - half.z;
    float l = length(Vec3(std::max(qx, 0.f), std::max(qy, 0.f), std::max(qz, 0.f)));
    return l + std::fmin(std::fmax(qx, std::fmax(qy, qz)), 0.f);
}
SdfOpSpecificBlock SdfBox::get_dt_specific_block() const {
    SdfOpSpecificBlock b = zero_block();
    b.vec4s[0].v[0] = half.x; b.vec4s[0].v[1] = half.y; b.vec4s[0].v[2] = half.z; b.vec4s[0].v[3] = 0.f;
    return b;
}
ExpandedSdfNode SdfBox::expand(const SdfNode& node, const BuildOptions&) const {
    return ExpandedSdfNode::primitive(node.bbox.value(), clone());
}

SdfBoundingBox SdfBoxFrame::get_bbox(const std::vector<SdfBoundingBox>&, const BuildOptions&) const {
    return SdfBoundingBox::from_transform(Transform::from_scale(dimension));   // elements.rs:131-133
}
SdfOpSpecificBlock SdfBoxFrame::get_dt_specific_block() const {                // elements.rs:142-147
    SdfOpSpecificBlock b = zero_block();
    b.vec4s[0].v[0] = dimension.x; b.vec4s[0].v[1] = dimension.y; b.vec4s[0].v[2] = dimension.z; b.vec4s[0].v[3] = 0.f;
    b.floats[0] = thickness;
    return b;
}
ExpandedSdfNode SdfBoxFrame::expand(const SdfNode& node, const BuildOptions&) const {
    return ExpandedSdfNode::primitive(node.bbox.value(), clone());
}

SdfOpSpecificBlock SdfUnion::get_ut_specific_block() const {
    SdfOpSpecificBlock b = zero_block();
    b.floats[0] = smooth_radius;                                 // elements.rs:207-211
    return b;
}

// elements.rs:171-199.  Deviation C-1: split() returns positions inside the
// subset; they are mapped back through index_vec before recursing (the crate
// passes them on as if they were slot indices).
static ExpandedSdfNode union_recurse(const SdfUnion& self, const SdfNode& node,
                                     const std::vector<size_t>& index_vec, const BuildOptions& o, int forks) {
    if (index_vec.size() == 1) return node.slots[index_vec[0]].expanded(o);
    std::vector<SdfBoundingBox> boxes;
    boxes.reserve(index_vec.size());
    for (size_t i : index_vec) boxes.push_back(node.slots[i].bbox.value());
    SdfBoundingBox merged = SdfBoundingBox::merge(boxes, o);
    auto halves = merged.split(boxes);
    std::vector<size_t> l, r;
    l.reserve(halves.first.size());
    r.reserve(halves.second.size());
    for (size_t k : halves.first) l.push_back(index_vec[k]);
    for (size_t k : halves.second) r.push_back(index_vec[k]);
    boxes.clear();
    boxes.shrink_to_fit();
    ExpandedSdfNode ln, rn;
    const bool par = forks > 0 && l.size() >= 512 && r.size() >= 512;
    fork_join(par, [&] { ln = union_recurse(self, node, l, o, par ? forks - 1 : forks); },
              [&] { rn = union_recurse(self, node, r, o, par ? forks - 1 : forks); });
    return ExpandedSdfNode::operation(std::move(ln), std::move(rn), merged, self.clone());
}

ExpandedSdfNode SdfUnion::expand(const SdfNode& node, const BuildOptions& o) const {
    if (node.slots.size() == 1) return node.slots[0].expanded(o);   // elements.rs:173
    if (node.slots.empty()) return ExpandedSdfNode::null();          // elements.rs:174
    std::vector<size_t> all(node.slots.size());
    std::iota(all.begin(), all.end(), (size_t)0);
    return union_recurse(*this, node, all, o, host_forks());
}

// ---------------------------------------------------------- SdfCaaClone ----
// elements.rs:224-229
SdfBoundingBox SdfCaaClone::get_bbox(const std::vector<SdfBoundingBox>&, const BuildOptions&) const {
    const Vec3 c((neg_limit.x + pos_limit.x) / 2.f * displacement.x, (neg_limit.y + pos_limit.y) / 2.f * displacement.y,
                 (neg_limit.z + pos_limit.z) / 2.f * displacement.z);
    const Vec3 sc((pos_limit.x - neg_limit.x) * displacement.x, (pos_limit.y - neg_limit.y) * displacement.y,
                  (pos_limit.z - neg_limit.z) * displacement.z);
    return SdfBoundingBox::from_transform(Transform::from_translation(c).with_scale(sc));
}

// elements.rs:234-241 (Rust `%` on f32 is fmod: the result takes the sign of the dividend)
Vec3 SdfCaaClone::downtree_transform(Vec3 p) const {
    const Vec3 l(p.x - displacement.x / 2.0f, p.y - displacement.y / 2.0f, p.z - displacement.z / 2.0f);
    return Vec3(std::fmod(l.x, displacement.x), std::fmod(l.y, displacement.y), std::fmod(l.z, displacement.z));
}

// elements.rs:252-258
SdfOpSpecificBlock SdfCaaClone::get_dt_specific_block() const {
    SdfOpSpecificBlock r;
    std::memset(&r, 0, sizeof r);
    const Vec3* src[3] = {&displacement, &neg_limit, &pos_limit};
    for (int k = 0; k < 3; ++k) {
        r.vec4s[k].v[0] = src[k]->x; r.vec4s[k].v[1] = src[k]->y; r.vec4s[k].v[2] = src[k]->z; r.vec4s[k].v[3] = 0.f;
    }
    return r;
}

// elements.rs:260-262
ExpandedSdfNode SdfCaaClone::expand(const SdfNode& node, const BuildOptions& o) const {
    if (node.slots.empty()) throw std::logic_error("SdfCaaClone::expand: slot 0 is empty");   // .get(0).unwrap()
    return ExpandedSdfNode::simple_operation(node.slots[0].expanded(o), clone());
}

// ------------------------------------------------------ ExpandedSdfNode ----

ExpandedSdfNode ExpandedSdfNode::null() { return ExpandedSdfNode(); }

ExpandedSdfNode ExpandedSdfNode::primitive(const SdfBoundingBox& b, std::unique_ptr<SdfElement> e) {
    ExpandedSdfNode n;
    n.bbox = b;
    n.intern = std::move(e);
    return n;
}

ExpandedSdfNode ExpandedSdfNode::simple_operation(ExpandedSdfNode child, std::unique_ptr<SdfElement> e) {
    ExpandedSdfNode n;
    n.bbox = child.bbox;
    n.expanded_slots = std::make_unique<std::array<ExpandedSdfNode, 2>>();
    (*n.expanded_slots)[0] = std::move(child);
    (*n.expanded_slots)[1] = null();
    n.intern = std::move(e);
    return n;
}

ExpandedSdfNode ExpandedSdfNode::operation(ExpandedSdfNode l, ExpandedSdfNode r, const SdfBoundingBox& b,
                                           std::unique_ptr<SdfElement> e) {
    ExpandedSdfNode n;
    n.bbox = b;
    n.expanded_slots = std::make_unique<std::array<ExpandedSdfNode, 2>>();
    (*n.expanded_slots)[0] = std::move(l);
    (*n.expanded_slots)[1] = std::move(r);
    n.intern = std::move(e);
    return n;
}

// node.rs:75-138.  The crate pushes blocks as it recurses; the positions are known beforehand (a node is
// followed by its left subtree, then its right one, in the down buffer; preceded by both in the up buffer),
// so the blocks are written in place into arrays sized from the subtree counts, and the two sides of the
// upper levels are filled by separate threads.
static uint32_t count_nodes(const ExpandedSdfNode& n) {
    if (n.is_null()) return 0;
    if (n.is_primitive()) return 1;
    const auto& s = *n.expanded_slots;
    return 1u + count_nodes(s[0]) + count_nodes(s[1]);
}

// returns the number of nodes written; d0 / u0 = first free position of the down / up buffer
static uint32_t make_buffer_recurse(SdfTreeBuffer& buf, const ExpandedSdfNode& root, const SdfBoundingBox& other_box,
                                    uint32_t level, bool parent_is_union, const BuildOptions& o, uint32_t d0, uint32_t u0,
                                    int forks) {
    if (root.is_null()) return 0;                                // node.rs:85-87
    SdfElementInfo info = root.intern->get_info();
    SdfOperationBlock& blk = buf.downtree_buffer[d0];
    std::memset(&blk, 0, sizeof blk);
    blk.op_code = info.op_id;                                    // node.rs:97-106
    blk.is_primitive = info.is_primitive ? 1 : 0;
    blk.parent_is_union = parent_is_union ? 1 : 0;
    blk.level = level;
    blk.op_specific = root.intern->get_dt_specific_block();
    blk.bounding_box = root.bbox.get_bbox_block();
    blk.other_box = other_box.get_bbox_block();

    uint32_t below = 0;
    if (!root.is_primitive()) {                                  // node.rs:108-122
        const auto& s = *root.expanded_slots;
        const bool u = root.is_union();
        // node.rs:113,119 pass each child's OWN box (C-2); optional sibling box.
        const SdfBoundingBox& ob0 = (o.sibling_other_box && !s[1].is_null()) ? s[1].bbox : s[0].bbox;
        const SdfBoundingBox& ob1 = (o.sibling_other_box && !s[0].is_null()) ? s[0].bbox : s[1].bbox;
        const uint32_t n0 = count_nodes(s[0]);
        uint32_t w0 = 0, w1 = 0;
        const bool par = forks > 0 && n0 >= 1024 && !s[1].is_null() && !s[1].is_primitive();
        fork_join(par, [&] { w0 = make_buffer_recurse(buf, s[0], ob0, level + 1, u, o, d0 + 1, u0, par ? forks - 1 : forks); },
                  [&] { w1 = make_buffer_recurse(buf, s[1], ob1, level + 1, u, o, d0 + 1 + n0, u0 + n0, par ? forks - 1 : forks); });
        below = w0 + w1;
    }
    SdfOperationUptreeBlock& ub = buf.uptree_buffer[u0 + below];
    std::memset(&ub, 0, sizeof ub);
    ub.op_code = info.op_id;                                     // node.rs:124-129
    ub.parent_is_union = parent_is_union ? 1 : 0;
    ub.op_specific = root.intern->get_ut_specific_block();
    ub.level = level;
    blk.len = below;                                             // :131 (blocks pushed after this one)
    return 1u + below;
}

SdfTreeBuffer ExpandedSdfNode::make_buffer(const BuildOptions& o) const {
    SdfTreeBuffer buf;
    const uint32_t n = count_nodes(*this);
    buf.downtree_buffer.resize(n);
    buf.uptree_buffer.resize(n);
    SdfBoundingBox throwaway = SdfBoundingBox::zero();           // node.rs:134
    make_buffer_recurse(buf, *this, throwaway, 1, false, o, 0, 0, host_forks());   // node.rs:135
    buf.buffer_len = n;                                          // node.rs:136
    return buf;
}

// -------------------------------------------------------------- SdfNode ----

bool SdfNode::set_slot(SdfNode child, const char** err) {
    SdfElementInfo info = intern->get_info();
    if (info.is_primitive) { if (err) *err = "Can't set slots on primitive!"; return false; }
    if (slots.size() >= info.num_slots()) { if (err) *err = "Slots already full!"; return false; }
    slots.push_back(std::move(child));
    return true;
}

SdfBoundingBox SdfNode::calc_bbox_assign(const BuildOptions& o) {
    SdfElementInfo info = intern->get_info();
    if (!(info.is_union || slots.size() >= info.num_slots()))    // node.rs:182-186
        throw std::logic_error("Tried calculating bounding box of SDF node without all required slots filled!");
    if (bbox) return *bbox;
    std::vector<SdfBoundingBox> sub;
    sub.reserve(slots.size());
    for (auto& s : slots) {                                      // node.rs:190-196
        SdfBoundingBox b = s.calc_bbox_assign(o);
        if (!b.is_zero()) sub.push_back(b);
    }
    bbox = intern->get_bbox(sub, o);
    return *bbox;
}

// ------------------------------------------------------------ SdfBuilder ----

SdfBuilder SdfBuilder::primitive(std::unique_ptr<SdfElement> prim, BuildOptions o) {
    if (!prim->get_info().is_primitive) throw std::logic_error("SdfBuilder::primitive: not a primitive");
    return SdfBuilder(SdfNode(std::move(prim)), o);
}

SdfBuilder SdfBuilder::operation(std::unique_ptr<SdfElement> op) && {
    if (op->get_info().is_primitive) throw std::logic_error("SdfBuilder::operation: is a primitive");
    SdfNode n(std::move(op));
    const char* err = nullptr;
    if (!n.set_slot(std::move(root), &err)) throw std::logic_error(std::string("Couldn't set operation child: ") + err);
    return SdfBuilder(std::move(n), opts);
}

SdfBuilder SdfBuilder::with(SdfBuilder node) && {
    const char* err = nullptr;
    if (!root.set_slot(std::move(node.root), &err))
        throw std::logic_error(std::string("Couldn't set operation slot: ") + err);
    return std::move(*this);
}

SdfBuilder SdfBuilder::transform(const Transform& t) && {
    root.bbox = root.calc_bbox_assign(opts).apply_transform(t);  // node.rs:405-408
    return std::move(*this);
}

SdfNode SdfBuilder::finalize() && {
    root.calc_bbox_assign(opts);
    return std::move(root);
}

// ---------------------------------------------------------------- scene ----

float default_scene_extent(uint32_t n) { return 50.0f * std::cbrt((float)n / 1000.0f); }

SdfNode make_random_scene(uint32_t n, uint64_t seed, float L, const BuildOptions& o) {
    if (n == 0) {
        SdfNode empty(std::make_unique<SdfUnion>(0.f));
        empty.bbox = SdfBoundingBox::zero();                     // node.rs:159-167 (SdfNode::empty)
        return empty;
    }
    if (!(L > 0.f)) L = default_scene_extent(n);
    SplitMix64 rng(seed);
    const float TWO_PI = 6.28318530717958647692f;
    SdfNode root(std::make_unique<SdfUnion>(0.f));
    root.slots.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        std::unique_ptr<SdfElement> prim;
        if (rng.coin()) {
            prim = std::make_unique<SdfSphere>(rng.range(0.5f, 2.0f));
        } else {
            float hx = rng.range(0.5f, 2.0f), hy = rng.range(0.5f, 2.0f), hz = rng.range(0.5f, 2.0f);
            prim = std::make_unique<SdfBox>(Vec3(hx, hy, hz));
        }
        Transform t;
        float tx = rng.range(-L, L), ty = rng.range(-L, L), tz = rng.range(-L, L);
        t.translation = Vec3(tx, ty, tz);
        float ea = rng.range(0.f, TWO_PI), eb = rng.range(0.f, TWO_PI), ec = rng.range(0.f, TWO_PI);
        t.rotation = Quat::from_euler_xyz(ea, eb, ec);
        SdfNode leaf = SdfBuilder::primitive(std::move(prim), o).transform(t).finalize();
        root.set_slot(std::move(leaf));
    }
    root.calc_bbox_assign(o);
    return root;
}

}  // namespace sdfhost
