// hmath.h — host-side vector / matrix helpers of the MI355X back end's C++ front (scene loading + CPU BVH build).
// Semantics follow the reference's template/tmplmath.{h,cpp} for the subset the path needs; every sum is written
// with its association explicit because the acceleration structures must come out bit-identical.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace crt {

struct float2 { float x = 0, y = 0; };
struct float3 {
    float x = 0, y = 0, z = 0;
    float3() = default;
    float3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit float3(float s) : x(s), y(s), z(s) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(const float3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, const float3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(const float3& a, const float3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float3 normalize(const float3& v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }   // tmplmath.h:480
inline float lesser(float a, float b) { return a < b ? a : b; }     // tmplmath.h:122 / _mm_min_ps operand order
inline float greater(float a, float b) { return a > b ? a : b; }    // tmplmath.h:123 / _mm_max_ps
inline float3 fminf3(const float3& a, const float3& b) { return {lesser(a.x, b.x), lesser(a.y, b.y), lesser(a.z, b.z)}; }
inline float3 fmaxf3(const float3& a, const float3& b) { return {greater(a.x, b.x), greater(a.y, b.y), greater(a.z, b.z)}; }

constexpr float PI = 3.14159265358979323846264f;               // template/common.h:8
const float Deg2Red = (PI * 2) / 360.0f;                        // infra/helper.h:152

// row-major 4x4, identity by default (tmplmath.h:638-642)
struct mat4 {
    float cell[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    static mat4 Identity() { return mat4(); }
    static mat4 Translate(const float3& p) { mat4 r; r.cell[3] = p.x; r.cell[7] = p.y; r.cell[11] = p.z; return r; }
    static mat4 RotateX(float a) { mat4 r; r.cell[5] = cosf(a); r.cell[6] = -sinf(a); r.cell[9] = sinf(a); r.cell[10] = cosf(a); return r; }
    static mat4 RotateY(float a) { mat4 r; r.cell[0] = cosf(a); r.cell[2] = sinf(a); r.cell[8] = -sinf(a); r.cell[10] = cosf(a); return r; }
    static mat4 RotateZ(float a) { mat4 r; r.cell[0] = cosf(a); r.cell[1] = -sinf(a); r.cell[4] = sinf(a); r.cell[5] = cosf(a); return r; }
    static mat4 Scale(const float3& s) { mat4 r; r.cell[0] = s.x; r.cell[5] = s.y; r.cell[10] = s.z; return r; }
    // tmplmath.h:745-768 — transpose of the 3x3 block, translation re-expressed; rows 3 stay identity
    mat4 FastInvertedTransformNoScale() const
    {
        mat4 r;
        r.cell[0] = cell[0]; r.cell[1] = cell[4]; r.cell[2] = cell[8];
        r.cell[4] = cell[1]; r.cell[5] = cell[5]; r.cell[6] = cell[9];
        r.cell[8] = cell[2]; r.cell[9] = cell[6]; r.cell[10] = cell[10];
        r.cell[3] = -(cell[3] * r.cell[0] + cell[7] * r.cell[1] + cell[11] * r.cell[2]);
        r.cell[7] = -(cell[3] * r.cell[4] + cell[7] * r.cell[5] + cell[11] * r.cell[6]);
        r.cell[11] = -(cell[3] * r.cell[8] + cell[7] * r.cell[9] + cell[11] * r.cell[10]);
        return r;
    }
};
inline mat4 operator*(const mat4& a, const mat4& b)   // tmplmath.cpp:109-122
{
    mat4 r;
    for (int row = 0; row < 4; row++) for (int col = 0; col < 4; col++) {
        const float* ar = &a.cell[row * 4];
        r.cell[row * 4 + col] = (ar[0] * b.cell[col]) + (ar[1] * b.cell[col + 4]) + (ar[2] * b.cell[col + 8]) + (ar[3] * b.cell[col + 12]);
    }
    return r;
}
// float4(a, w) * M with the scalar left-to-right summation of tmplmath.cpp:155-169
inline float3 TransformPosition(const float3& a, const mat4& M)
{
    const float* m = M.cell;
    return {m[0] * a.x + m[1] * a.y + m[2] * a.z + m[3] * 1.0f, m[4] * a.x + m[5] * a.y + m[6] * a.z + m[7] * 1.0f, m[8] * a.x + m[9] * a.y + m[10] * a.z + m[11] * 1.0f};
}
inline float3 TransformVector(const float3& a, const mat4& M)
{
    const float* m = M.cell;
    return {m[0] * a.x + m[1] * a.y + m[2] * a.z + m[3] * 0.0f, m[4] * a.x + m[5] * a.y + m[6] * a.z + m[7] * 0.0f, m[8] * a.x + m[9] * a.y + m[10] * a.z + m[11] * 0.0f};
}

// axis-aligned box with the reference aabb's defaults and operand order (tmplmath.h:568-622)
struct aabb {
    float3 bmin3{1e34f, 1e34f, 1e34f}, bmax3{-1e34f, -1e34f, -1e34f};
    void Grow(const float3& p) { bmin3 = fminf3(bmin3, p); bmax3 = fmaxf3(bmax3, p); }
    void Grow(const aabb& b) { bmin3 = fminf3(bmin3, b.bmin3); bmax3 = fmaxf3(bmax3, b.bmax3); }
    float Area() const
    {
        const float e0 = bmax3.x - bmin3.x, e1 = bmax3.y - bmin3.y, e2 = bmax3.z - bmin3.z;
        const float a = e0 * e1 + e0 * e2 + e1 * e2;
        return (0.0f < a) ? a : 0.0f;    // std::max(0.0f, a)
    }
};

} // namespace crt
