// sdf_math.hpp -- small fixed-size linear algebra for the host-side scene
// builder.  Column-major 4x4 like glam::Mat4 / the record layout in
// include/sdf_blocks.h.  Host only (scene construction happens once per scene;
// none of this is on the GPU hot path).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace sdfhost {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    static Vec3 splat(float v) { return Vec3(v, v, v); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float length(Vec3 a) { return std::sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); }

struct Vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    Vec4() = default;
    Vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    Vec4(Vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    Vec3 truncate() const { return Vec3(x, y, z); }
};

// Unit quaternion (x, y, z, w), glam convention.
struct Quat {
    float x = 0, y = 0, z = 0, w = 1;
    Quat() = default;
    Quat(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    static Quat identity() { return Quat(0, 0, 0, 1); }
    static Quat from_rotation_x(float a) { return Quat(std::sin(a * 0.5f), 0, 0, std::cos(a * 0.5f)); }
    static Quat from_rotation_y(float a) { return Quat(0, std::sin(a * 0.5f), 0, std::cos(a * 0.5f)); }
    static Quat from_rotation_z(float a) { return Quat(0, 0, std::sin(a * 0.5f), std::cos(a * 0.5f)); }
    // glam Quat::from_euler(EulerRot::XYZ, a, b, c) (node.rs:482-487 recipe)
    static Quat from_euler_xyz(float a, float b, float c);
};
inline Quat operator*(Quat p, Quat q) {
    return Quat(p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y,
                p.w * q.y - p.x * q.z + p.y * q.w + p.z * q.x,
                p.w * q.z + p.x * q.y - p.y * q.x + p.z * q.w,
                p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z);
}
inline Quat Quat::from_euler_xyz(float a, float b, float c) {
    return from_rotation_x(a) * from_rotation_y(b) * from_rotation_z(c);
}

struct Mat4 {
    float m[16];  // m[4*col + row]
    static Mat4 zeros() { Mat4 r; std::memset(r.m, 0, sizeof r.m); return r; }
    static Mat4 identity() { Mat4 r = zeros(); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
    static Mat4 translation(Vec3 t) { Mat4 r = identity(); r.m[12] = t.x; r.m[13] = t.y; r.m[14] = t.z; return r; }
    static Mat4 scaling(Vec3 s) { Mat4 r = identity(); r.m[0] = s.x; r.m[5] = s.y; r.m[10] = s.z; return r; }
    float& at(int row, int col) { return m[4 * col + row]; }
    float at(int row, int col) const { return m[4 * col + row]; }
    Vec4 col(int c) const { return Vec4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]); }
    bool is_affine() const { return m[3] == 0.f && m[7] == 0.f && m[11] == 0.f && m[15] == 1.f; }
    bool is_identity() const {
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r)
                if (at(r, c) != (r == c ? 1.f : 0.f)) return false;
        return true;
    }
};

inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a.at(row, k) * b.at(k, c);
            r.at(row, c) = s;
        }
    return r;
}
inline Vec4 operator*(const Mat4& a, Vec4 v) {
    Vec4 r;
    r.x = ((a.m[0] * v.x + a.m[4] * v.y) + a.m[8] * v.z) + a.m[12] * v.w;
    r.y = ((a.m[1] * v.x + a.m[5] * v.y) + a.m[9] * v.z) + a.m[13] * v.w;
    r.z = ((a.m[2] * v.x + a.m[6] * v.y) + a.m[10] * v.z) + a.m[14] * v.w;
    r.w = ((a.m[3] * v.x + a.m[7] * v.y) + a.m[11] * v.z) + a.m[15] * v.w;
    return r;
}

// 4x4 inverse.  Affine inputs (bottom row exactly 0,0,0,1 -- every matrix the
// builder produces) are inverted as [A|t]^-1 = [A^-1 | -A^-1 t] in double and
// keep an exact bottom row, which is what lets the GPU fast path treat leaf
// transforms as 3x4.  Anything else goes through the cofactor expansion
// (glam Mat4::inverse / nalgebra try_inverse are both cofactor inverses).
// Returns false if singular.
bool invert(const Mat4& a, Mat4* out);

// bevy Transform: scale, then rotate, then translate.
struct Transform {
    Vec3 translation{0, 0, 0};
    Quat rotation = Quat::identity();
    Vec3 scale{1, 1, 1};
    static Transform identity() { return Transform(); }
    static Transform from_translation(Vec3 t) { Transform r; r.translation = t; return r; }
    static Transform from_rotation(Quat q) { Transform r; r.rotation = q; return r; }
    static Transform from_scale(Vec3 s) { Transform r; r.scale = s; return r; }
    static Transform from_xyz(float x, float y, float z) { return from_translation(Vec3(x, y, z)); }
    Transform with_scale(Vec3 s) const { Transform r = *this; r.scale = s; return r; }
    // glam Mat4::from_scale_rotation_translation
    Mat4 compute_matrix() const;
};

// Jacobi eigen-decomposition of a symmetric 3x3 (double).  Columns of V are
// orthonormal eigenvectors.  (The crate calls nalgebra symmetric_eigen on a
// 4x4 whose w row/column is zero, obb.rs:142; parity with it is unpinned.)
void jacobi_eigen3(const double a[3][3], double V[3][3], double w[3]);

// Deterministic PRNG (the crate uses thread_rng(), which is unreproducible).
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float uniform() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }  // [0,1)
    float range(float lo, float hi) { return lo + (hi - lo) * uniform(); }
    bool coin() { return (next() >> 63) != 0; }
};

}  // namespace sdfhost
