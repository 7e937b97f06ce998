/*
 * crt_oracle.cpp — CPU ORACLE for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch restatement (plain C++17, scalar fp32, no SIMD, no contraction) of the algorithm in
 * willake/cpu-ray-tracer (reference tree /root/reference), written to be the checker for the HIP product
 * path and the CPU baseline of bench.py.  Nothing under cpu-ray-tracer_amd/ may include, link or call it.
 *
 * PINNING STATUS (see DESIGN.md §"Oracle"):
 *   - pinned by the real reference, compiled unpatched where it lies (oracle/_ref; tests/test_oracle_pinning.py live + tests/golden ref_*):
 *       BVH build (node array, triangleIndices order), BVH::Refit, IntersectAABB / IntersectTri / IntersectBVH results
 *       [infra/bvh.cpp — blas_bvh.cpp's build/traverse code is textually identical modulo names],
 *       OBJ triangulation + float parsing [lib/tiny_obj_loader.h], texture decode [lib/stb_image.h],
 *       Camera default frustum / SetCameraState / GetPrimaryRay [template/camera.h], Texture packing + Sample, Material::GetAlbedo
 *       [template/texture.h, material.h].
 *   - PARITY UNPINNED (restated line by line from the cited reference lines, no executable reference
 *     and no reference-held golden vector exists: the reference has no tests and is MSVC/Windows-only):
 *       Renderer::Sample/ProcessTile/Tick, RNG, Quad/Plane, TLAS build + traversal,
 *       BLAS instance transforms, GetHitInfo, skydome lookup, scene assembly.
 *
 * Conventions pinned here (reference leaves them to the compiler; DESIGN.md "pinned choices"):
 *   - argument evaluation order = MSVC's right-to-left: in GetPrimaryRay(x+rnd, y+rnd) the FIRST draw is the
 *     y jitter; in diffusereflection's make_float3(rnd,rnd,rnd) the first draw is z, then y, then x.
 *   - no FMA contraction (build with -ffp-contract=off), IEEE divide/sqrt.
 *   - expf / atan2f / acosf are the deterministic crt_* functions below (identical formulas in the HIP
 *     kernels), so sky-texel selection and absorption are bit-reproducible across CPU and GPU.
 */
#include "crt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

typedef unsigned int uint;

// ------------------------------------------------------------------------------------------------
// small vector math — one IEEE op per component, association written out explicitly
// (template/tmplmath.h:221-367, 458, 480, 506, 512)
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct V2 { float x, y; };
static inline V3 v3(float a, float b, float c) { V3 r; r.x = a; r.y = b; r.z = c; return r; }
static inline V3 v3(float s) { return v3(s, s, s); }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float b) { return v3(a.x * b, a.y * b, a.z * b); }
static inline V3 operator*(float b, V3 a) { return v3(b * a.x, b * a.y, b * a.z); }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           // (xx+yy)+zz
static inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float rsqrtf_(float x) { return 1.0f / sqrtf(x); }                             // tmplmath.h:124
static inline V3 normalize(V3 v) { float inv = rsqrtf_(dot(v, v)); return v * inv; }        // tmplmath.h:480
static inline V3 reflect(V3 i, V3 n) { return i - 2.0f * n * dot(n, i); }                    // tmplmath.h:506
static inline float tmin_(float a, float b) { return a < b ? a : b; }                        // tmplmath.h:122 fminf
static inline float tmax_(float a, float b) { return a > b ? a : b; }                        // tmplmath.h:123 fmaxf
static inline float smin_(float a, float b) { return (b < a) ? b : a; }                      // std::min
static inline float smax_(float a, float b) { return (a < b) ? b : a; }                      // std::max
static inline V3 vmin(V3 a, V3 b) { return v3(tmin_(a.x, b.x), tmin_(a.y, b.y), tmin_(a.z, b.z)); }
static inline V3 vmax(V3 a, V3 b) { return v3(tmax_(a.x, b.x), tmax_(a.y, b.y), tmax_(a.z, b.z)); }
static inline float clampf(float f, float a, float b) { return tmax_(a, tmin_(f, b)); }      // tmplmath.h:435
static inline int clampi(int f, int a, int b) { int m = (b < f) ? b : f; return (a < m) ? m : a; } // :436 max(a,min(f,b))
static inline float comp(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

const float kPI = 3.14159265358979323846264f;      // template/common.h:8
const float kINVPI = 0.31830988618379067153777f;   // :9
const float kINV2PI = 0.15915494309189533576888f;  // :10
const float kEPS = 0.001f;                          // renderer.h:12

// ------------------------------------------------------------------------------------------------
// deterministic transcendental functions (DESIGN.md "numerics"): only + - * / sqrt floor and bit ops,
// so the same source gives the same bits on x86 and on gfx950.
// ------------------------------------------------------------------------------------------------
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float det_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return bits2f(0x7f800000u);
    if (x < -103.972084045410f) return 0.0f;
    float fk = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - fk * 0.693359375f;
    r = r - fk * -2.12194440e-4f;
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * z + r;
    p = p + 1.0f;
    int k = (int)fk;
    // scale by 2^k in two steps so results stay correct through the subnormal range
    int k1 = k / 2, k2 = k - k1;
    float s1 = bits2f((uint32_t)(k1 + 127) << 23), s2 = bits2f((uint32_t)(k2 + 127) << 23);
    return p * s1 * s2;
}

static float det_atanf_pos(float x) // x >= 0
{
    float y0, t;
    if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; t = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; t = (x - 1.0f) / (x + 1.0f); }
    else { y0 = 0.0f; t = x; }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    p = p * z * t + t;
    return y0 + p;
}

static float det_atan2f(float y, float x)
{
    if (x != x || y != y) return x + y;
    uint32_t sy = f2bits(y) & 0x80000000u, sx = f2bits(x) & 0x80000000u;
    float ax = fabsf(x), ay = fabsf(y);
    float r;
    if (ay == 0.0f) r = sx ? kPI : 0.0f;
    else if (ax == 0.0f) r = 1.5707963267948966f;
    else if (ax == INFINITY && ay == INFINITY) r = sx ? 2.356194490192345f : 0.7853981633974483f;
    else {
        float a = det_atanf_pos(ay / ax);
        r = sx ? (kPI - a) : a;
    }
    return bits2f(f2bits(r) | sy); // r >= 0 here; give it the sign of y
}

static float det_asinf_small(float x) // |x| <= 0.5
{
    float z = x * x;
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    p = p * z * x + x;
    return p;
}

static float det_acosf(float x)
{
    if (x != x) return x;
    if (x > 1.0f || x < -1.0f) return bits2f(0x7fc00000u);
    if (x > 0.5f) {
        float s = sqrtf(0.5f * (1.0f - x));
        return 2.0f * det_asinf_small(s);
    }
    if (x < -0.5f) {
        float s = sqrtf(0.5f * (1.0f + x));
        return kPI - 2.0f * det_asinf_small(s);
    }
    return 1.5707963267948966f - det_asinf_small(x);
}

// ------------------------------------------------------------------------------------------------
// RNG (template/tmplmath.cpp:5-16, 27-34)
// ------------------------------------------------------------------------------------------------
static inline uint wang_hash(uint s)
{
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u; s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return s;
}
static inline uint init_seed(uint base) { return wang_hash((base + 1u) * 17u); }
static inline uint random_uint(uint& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
static inline float random_float(uint& s) { return random_uint(s) * 2.3283064365387e-10f; }

// ------------------------------------------------------------------------------------------------
// 4x4 matrix, row-major, identity by default (template/tmplmath.h:638-768, tmplmath.cpp:109-191)
// ------------------------------------------------------------------------------------------------
struct M4 { float c[16]; };
static M4 m4_identity() { M4 m; for (int i = 0; i < 16; i++) m.c[i] = (i % 5 == 0) ? 1.0f : 0.0f; return m; }
static M4 m4_mul(const M4& a, const M4& b) // tmplmath.cpp:109-122
{
    M4 r;
    for (int i = 0; i < 16; i += 4) for (int j = 0; j < 4; ++j)
        r.c[i + j] = (a.c[i + 0] * b.c[j + 0]) + (a.c[i + 1] * b.c[j + 4]) + (a.c[i + 2] * b.c[j + 8]) + (a.c[i + 3] * b.c[j + 12]);
    return r;
}
static M4 m4_translate(V3 p) { M4 r = m4_identity(); r.c[3] = p.x; r.c[7] = p.y; r.c[11] = p.z; return r; }   // :735
static M4 m4_rotx(float a) { M4 r = m4_identity(); r.c[5] = cosf(a); r.c[6] = -sinf(a); r.c[9] = sinf(a); r.c[10] = cosf(a); return r; }  // :673
static M4 m4_roty(float a) { M4 r = m4_identity(); r.c[0] = cosf(a); r.c[2] = sinf(a); r.c[8] = -sinf(a); r.c[10] = cosf(a); return r; }  // :674
static M4 m4_rotz(float a) { M4 r = m4_identity(); r.c[0] = cosf(a); r.c[1] = -sinf(a); r.c[4] = sinf(a); r.c[5] = cosf(a); return r; }   // :675
static M4 m4_scale(V3 s) { M4 r = m4_identity(); r.c[0] = s.x; r.c[5] = s.y; r.c[10] = s.z; return r; }      // :677
static M4 m4_fast_inverted_noscale(const M4& m) // tmplmath.h:745-768
{
    M4 r = m4_identity();
    r.c[0] = m.c[0]; r.c[1] = m.c[4]; r.c[2] = m.c[8];
    r.c[4] = m.c[1]; r.c[5] = m.c[5]; r.c[6] = m.c[9];
    r.c[8] = m.c[2]; r.c[9] = m.c[6]; r.c[10] = m.c[10];
    r.c[3] = -(m.c[3] * r.c[0] + m.c[7] * r.c[1] + m.c[11] * r.c[2]);
    r.c[7] = -(m.c[3] * r.c[4] + m.c[7] * r.c[5] + m.c[11] * r.c[6]);
    r.c[11] = -(m.c[3] * r.c[8] + m.c[7] * r.c[9] + m.c[11] * r.c[10]);
    return r;
}
// float4(a,w) * M, scalar path, left-to-right sums (tmplmath.cpp:155-169)
static inline V3 transform_position(V3 a, const M4& M)
{
    return v3(M.c[0] * a.x + M.c[1] * a.y + M.c[2] * a.z + M.c[3] * 1.0f,
              M.c[4] * a.x + M.c[5] * a.y + M.c[6] * a.z + M.c[7] * 1.0f,
              M.c[8] * a.x + M.c[9] * a.y + M.c[10] * a.z + M.c[11] * 1.0f);
}
static inline V3 transform_vector(V3 a, const M4& M)
{
    return v3(M.c[0] * a.x + M.c[1] * a.y + M.c[2] * a.z + M.c[3] * 0.0f,
              M.c[4] * a.x + M.c[5] * a.y + M.c[6] * a.z + M.c[7] * 0.0f,
              M.c[8] * a.x + M.c[9] * a.y + M.c[10] * a.z + M.c[11] * 0.0f);
}
// SSE paths sum (x+y)+(z+w) resp. (x+y)+z (tmplmath.cpp:170-191)
static inline V3 transform_position_sse(V3 a, const M4& M)
{
    return v3((a.x * M.c[0] + a.y * M.c[1]) + (a.z * M.c[2] + 1.0f * M.c[3]),
              (a.x * M.c[4] + a.y * M.c[5]) + (a.z * M.c[6] + 1.0f * M.c[7]),
              (a.x * M.c[8] + a.y * M.c[9]) + (a.z * M.c[10] + 1.0f * M.c[11]));
}
static inline V3 transform_vector_sse(V3 a, const M4& M)
{
    return v3((a.x * M.c[0] + a.y * M.c[1]) + a.z * M.c[2],
              (a.x * M.c[4] + a.y * M.c[5]) + a.z * M.c[6],
              (a.x * M.c[8] + a.y * M.c[9]) + a.z * M.c[10]);
}

// ------------------------------------------------------------------------------------------------
// aabb with the SSE min/max semantics of template/tmplmath.h:568-622
// ------------------------------------------------------------------------------------------------
static inline float mm_min(float a, float b) { return a < b ? a : b; } // _mm_min_ps(a,b)
static inline float mm_max(float a, float b) { return a > b ? a : b; } // _mm_max_ps(a,b)
struct Box {
    V3 lo, hi;
    Box() { lo = v3(1e34f); hi = v3(-1e34f); }                              // :613
    void grow(V3 p) { lo = v3(mm_min(lo.x, p.x), mm_min(lo.y, p.y), mm_min(lo.z, p.z)); hi = v3(mm_max(hi.x, p.x), mm_max(hi.y, p.y), mm_max(hi.z, p.z)); }
    void grow(const Box& b) { lo = v3(mm_min(lo.x, b.lo.x), mm_min(lo.y, b.lo.y), mm_min(lo.z, b.lo.z)); hi = v3(mm_max(hi.x, b.hi.x), mm_max(hi.y, b.hi.y), mm_max(hi.z, b.hi.z)); }
    float area() const { float e0 = hi.x - lo.x, e1 = hi.y - lo.y, e2 = hi.z - lo.z; return smax_(0.0f, e0 * e1 + e0 * e2 + e1 * e2); } // :593-598
};

// ------------------------------------------------------------------------------------------------
// ray / triangle / node records (template/ray.h, infra/helper.h:6-26, infra/blas_bvh.h:13-20)
// ------------------------------------------------------------------------------------------------
struct Ray {
    V3 O, D, rD;
    float t = 1e34f;
    float bu = 0, bv = 0;
    int objIdx = -1, triIdx = -1;
    int traversed = 0, tested = 0;
    bool inside = false;
};
static Ray make_ray(V3 o, V3 d) // ray.h:15-24
{
    Ray r; r.O = o; r.D = d; r.t = 1e34f;
    r.rD = v3(1 / d.x, 1 / d.y, 1 / d.z);
    r.objIdx = -1;
    return r;
}

struct Counters {
    uint64_t rays = 0, primary = 0, interior = 0, leaf = 0, tri = 0, tlas = 0, visits = 0, meshhits = 0;
    void add(const Counters& o) { rays += o.rays; primary += o.primary; interior += o.interior; leaf += o.leaf; tri += o.tri; tlas += o.tlas; visits += o.visits; meshhits += o.meshhits; }
};

typedef orc_tri Tri;        // 112-byte AoS, same field order as infra/helper.h:6-26
typedef orc_bvh_node Node;  // 32 bytes
static inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
static inline void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// ------------------------------------------------------------------------------------------------
// binned-SAH BVH (infra/bvh.cpp:4-24, 45-178; identical code in infra/blas_bvh.cpp:82-256)
// ------------------------------------------------------------------------------------------------
struct Bvh {
    std::vector<Node> nodes;
    std::vector<Tri> tris;
    std::vector<uint> triIdx;
    uint nodesUsed = 1, maxDepth = 0;
    int blasObjIdx = -1;      // BLASBVH::objIdx (blas_bvh.cpp:59); -1 = single-level BVH, use tri.objIdx
    int matIdx = -1;
    M4 T = m4_identity(), invT = m4_identity();
    Box worldBounds;

    void update_bounds(uint ni) // bvh.cpp:45-61
    {
        Node& n = nodes[ni];
        V3 lo = v3(1e30f), hi = v3(-1e30f);
        for (uint first = n.leftFirst, i = 0; i < n.triCount; i++) {
            const Tri& t = tris[triIdx[first + i]];
            lo = vmin(lo, ld3(t.vertex0)); lo = vmin(lo, ld3(t.vertex1)); lo = vmin(lo, ld3(t.vertex2));
            hi = vmax(hi, ld3(t.vertex0)); hi = vmax(hi, ld3(t.vertex1)); hi = vmax(hi, ld3(t.vertex2));
        }
        st3(n.aabbMin, lo); st3(n.aabbMax, hi);
    }
    void refit() // bvh.cpp:26-43 (blas_bvh.cpp:104-121): bottom-up, node 1 skipped as the reference's loop does
    {
        for (int i = (int)nodesUsed - 1; i >= 0; i--) if (i != 1) {
            Node& n = nodes[(size_t)i];
            if (n.triCount > 0) { update_bounds((uint)i); continue; }
            const Node& l = nodes[n.leftFirst]; const Node& r = nodes[n.leftFirst + 1];
            st3(n.aabbMin, vmin(ld3(l.aabbMin), ld3(r.aabbMin)));
            st3(n.aabbMax, vmax(ld3(l.aabbMax), ld3(r.aabbMax)));
        }
    }
    float node_cost(const Node& n) // bvh.cpp:117-122
    {
        float ex = n.aabbMax[0] - n.aabbMin[0], ey = n.aabbMax[1] - n.aabbMin[1], ez = n.aabbMax[2] - n.aabbMin[2];
        float area = ex * ey + ey * ez + ez * ex;
        return n.triCount * area;
    }
    float best_split(const Node& n, int& axis, float& splitPos) // bvh.cpp:124-178
    {
        const int BINS = 8;
        float bestCost = 1e30f;
        for (int a = 0; a < 3; a++) {
            float bmin = 1e30f, bmax = -1e30f;
            for (uint i = 0; i < n.triCount; i++) {
                float c = tris[triIdx[n.leftFirst + i]].centroid[a];
                bmin = smin_(bmin, c); bmax = smax_(bmax, c);
            }
            if (bmin == bmax) continue;
            Box binBox[BINS]; int binCount[BINS] = {0, 0, 0, 0, 0, 0, 0, 0};
            float scale = BINS / (bmax - bmin);
            for (uint i = 0; i < n.triCount; i++) {
                const Tri& t = tris[triIdx[n.leftFirst + i]];
                int b = (int)((t.centroid[a] - bmin) * scale);
                if (BINS - 1 < b) b = BINS - 1;               // min(BINS-1, b)
                binCount[b]++;
                binBox[b].grow(ld3(t.vertex0)); binBox[b].grow(ld3(t.vertex1)); binBox[b].grow(ld3(t.vertex2));
            }
            float leftArea[BINS - 1], rightArea[BINS - 1];
            int leftCount[BINS - 1], rightCount[BINS - 1];
            Box leftBox, rightBox; int leftSum = 0, rightSum = 0;
            for (int i = 0; i < BINS - 1; i++) {
                leftSum += binCount[i]; leftCount[i] = leftSum;
                leftBox.grow(binBox[i]); leftArea[i] = leftBox.area();
                rightSum += binCount[BINS - 1 - i]; rightCount[BINS - 2 - i] = rightSum;
                rightBox.grow(binBox[BINS - 1 - i]); rightArea[BINS - 2 - i] = rightBox.area();
            }
            scale = (bmax - bmin) / BINS;
            for (int i = 0; i < BINS - 1; i++) {
                float cost = leftCount[i] * leftArea[i] + rightCount[i] * rightArea[i];
                if (cost < bestCost) { axis = a; splitPos = bmin + scale * (i + 1); bestCost = cost; }
            }
        }
        return bestCost;
    }
    void subdivide(uint ni, uint depth) // bvh.cpp:63-115
    {
        Node& n = nodes[ni];
        if (n.triCount <= 2) return;
        int axis = 0; float splitPos = 0;
        float splitCost = best_split(n, axis, splitPos);
        float nosplit = node_cost(n);
        if (splitCost >= nosplit) return;
        int i = n.leftFirst, j = i + n.triCount - 1;
        while (i <= j) {
            if (tris[triIdx[i]].centroid[axis] < splitPos) i++;
            else { uint tmp = triIdx[i]; triIdx[i] = triIdx[j]; triIdx[j] = tmp; j--; }
        }
        int leftCount = i - n.leftFirst;
        if (leftCount == 0 || leftCount == (int)n.triCount) return;
        int L = nodesUsed++, R = nodesUsed++;
        nodes[L].leftFirst = n.leftFirst; nodes[L].triCount = leftCount;
        nodes[R].leftFirst = i; nodes[R].triCount = n.triCount - leftCount;
        n.leftFirst = L; n.triCount = 0;
        update_bounds(L); update_bounds(R);
        if (depth > maxDepth) maxDepth = depth;
        subdivide(L, depth + 1);
        subdivide(R, depth + 1);
    }
    void build() // bvh.cpp:4-24
    {
        triIdx.resize(tris.size());
        for (size_t i = 0; i < tris.size(); i++) triIdx[i] = (uint)i;
        nodes.assign(tris.size() * 2 - 1, Node());
        memset(nodes.data(), 0, nodes.size() * sizeof(Node));
        nodesUsed = 1; maxDepth = 0;
        nodes[0].leftFirst = 0; nodes[0].triCount = (uint)tris.size();
        update_bounds(0);
        subdivide(0, 0);
    }

    // bvh.cpp:181-190
    static inline float hit_aabb(const Ray& r, const float* lo, const float* hi)
    {
        float tx1 = (lo[0] - r.O.x) * r.rD.x, tx2 = (hi[0] - r.O.x) * r.rD.x;
        float tmin = smin_(tx1, tx2), tmax = smax_(tx1, tx2);
        float ty1 = (lo[1] - r.O.y) * r.rD.y, ty2 = (hi[1] - r.O.y) * r.rD.y;
        tmin = smax_(tmin, smin_(ty1, ty2)); tmax = smin_(tmax, smax_(ty1, ty2));
        float tz1 = (lo[2] - r.O.z) * r.rD.z, tz2 = (hi[2] - r.O.z) * r.rD.z;
        tmin = smax_(tmin, smin_(tz1, tz2)); tmax = smin_(tmax, smax_(tz1, tz2));
        if (tmax >= tmin && tmin < r.t && tmax > 0) return tmin; else return 1e30f;
    }
    // bvh.cpp:203-222 / blas_bvh.cpp:281-300
    inline void hit_tri(Ray& r, const Tri& tri, uint ti) const
    {
        V3 v0 = ld3(tri.vertex0);
        V3 e1 = ld3(tri.vertex1) - v0, e2 = ld3(tri.vertex2) - v0;
        V3 h = cross(r.D, e2);
        float a = dot(e1, h);
        if (a > -0.0001f && a < 0.0001f) return;
        float f = 1 / a;
        V3 s = r.O - v0;
        float u = f * dot(s, h);
        if (u < 0 || u > 1) return;
        V3 q = cross(s, e1);
        float v = f * dot(r.D, q);
        if (v < 0 || u + v > 1) return;
        float t = f * dot(e2, q);
        if (t > 0.0001f) {
            if (t < r.t) {
                r.t = smin_(r.t, t);
                r.objIdx = (blasObjIdx >= 0) ? blasObjIdx : tri.objIdx;
                r.triIdx = (int)ti; r.bu = u; r.bv = v;
            }
        }
    }
    // bvh.cpp:224-258 (BVH_FASTER_RAY)
    void traverse(Ray& r, Counters& cn) const
    {
        uint node = 0, stack[64]; uint sp = 0;
        while (1) {
            r.traversed++;
            const Node& n = nodes[node];
            if (n.triCount > 0) {
                cn.leaf++;
                for (uint i = 0; i < n.triCount; i++) {
                    uint ti = triIdx[n.leftFirst + i];
                    r.tested++; cn.tri++;
                    hit_tri(r, tris[ti], ti);
                }
                if (sp == 0) break; else node = stack[--sp];
                continue;
            }
            cn.interior++;
            uint c1 = n.leftFirst, c2 = n.leftFirst + 1;
            float d1 = hit_aabb(r, nodes[c1].aabbMin, nodes[c1].aabbMax);
            float d2 = hit_aabb(r, nodes[c2].aabbMin, nodes[c2].aabbMax);
            if (d1 > d2) { float td = d1; d1 = d2; d2 = td; uint tc = c1; c1 = c2; c2 = tc; }
            if (d1 == 1e30f) { if (sp == 0) break; else node = stack[--sp]; }
            else { node = c1; if (d2 != 1e30f) stack[sp++] = c2; }
        }
    }
    // blas_bvh.cpp:363-374
    void set_transform(const M4& t)
    {
        T = t; invT = m4_fast_inverted_noscale(t);
        V3 lo = ld3(nodes[0].aabbMin), hi = ld3(nodes[0].aabbMax);
        worldBounds = Box();
        for (int i = 0; i < 8; i++)
            worldBounds.grow(transform_position(v3(i & 1 ? hi.x : lo.x, i & 2 ? hi.y : lo.y, i & 4 ? hi.z : lo.z), t));
    }
    // blas_bvh.cpp:376-389
    void intersect_instance(Ray& r, Counters& cn) const
    {
        Ray tr = r;                       // Ray(const Ray&) copies everything the traversal reads; `tested` restarts at 0 (ray.h:10-14)
        tr.tested = 0;
        tr.O = transform_position_sse(r.O, invT);
        tr.D = transform_vector_sse(r.D, invT);
        tr.rD = v3(1 / tr.D.x, 1 / tr.D.y, 1 / tr.D.z);
        traverse(tr, cn);
        tr.O = r.O; tr.D = r.D; tr.rD = r.rD;
        int testedTotal = r.tested + tr.tested;
        r = tr;
        r.tested = testedTotal;           // oracle keeps the per-query total (documented deviation from the reference's reset)
    }
};

// ------------------------------------------------------------------------------------------------
// TLAS (infra/tlas_bvh.cpp:4-111)
// ------------------------------------------------------------------------------------------------
struct Tlas {
    std::vector<orc_tlas_node> nodes; uint nodesUsed = 0;
    std::vector<Bvh*> blas;

    int best_match(const int* list, int N, int A) const // :57-70
    {
        float smallest = 1e30f; int best = -1;
        for (int B = 0; B < N; B++) if (B != A) {
            V3 bmax = vmax(ld3(nodes[list[A]].aabbMax), ld3(nodes[list[B]].aabbMax));
            V3 bmin = vmin(ld3(nodes[list[A]].aabbMin), ld3(nodes[list[B]].aabbMin));
            V3 e = bmax - bmin;
            float area = e.x * e.y + e.y * e.z + e.z * e.x;
            if (area < smallest) { smallest = area; best = B; }
        }
        return best;
    }
    bool build(std::string& err) // :17-55
    {
        int count = (int)blas.size();
        if (count > 256) { err = "TLAS supports at most 256 BLAS (tlas_bvh.cpp:21)"; return false; }
        orc_tlas_node zero; memset(&zero, 0, sizeof(zero));
        nodes.assign(2 * (size_t)count, zero);
        int idx[256], n = count;
        nodesUsed = 1;
        for (int i = 0; i < count; i++) {
            idx[i] = nodesUsed;
            st3(nodes[nodesUsed].aabbMin, blas[i]->worldBounds.lo);
            st3(nodes[nodesUsed].aabbMax, blas[i]->worldBounds.hi);
            nodes[nodesUsed].BLAS = i;
            nodes[nodesUsed++].leftRight = 0;
        }
        int A = 0, B = best_match(idx, n, A);
        while (n > 1) {
            int C = best_match(idx, n, B);
            if (A == C) {
                int ia = idx[A], ib = idx[B];
                orc_tlas_node& nn = nodes[nodesUsed];
                nn.leftRight = ia + (ib << 16);
                st3(nn.aabbMin, vmin(ld3(nodes[ia].aabbMin), ld3(nodes[ib].aabbMin)));
                st3(nn.aabbMax, vmax(ld3(nodes[ia].aabbMax), ld3(nodes[ib].aabbMax)));
                idx[A] = nodesUsed++;
                idx[B] = idx[n - 1];
                B = best_match(idx, --n, A);
            } else { A = B; B = C; }
        }
        nodes[0] = nodes[idx[A]];
        return true;
    }
    void traverse(Ray& r, Counters& cn) const // :83-111
    {
        uint node = 0, stack[64]; uint sp = 0;
        while (1) {
            r.traversed++; cn.tlas++;
            const orc_tlas_node& n = nodes[node];
            if (n.leftRight == 0) {
                cn.visits++;
                blas[n.BLAS]->intersect_instance(r, cn);
                if (sp == 0) break; else node = stack[--sp];
                continue;
            }
            uint c1 = n.leftRight & 0xffff, c2 = n.leftRight >> 16;
            float d1 = Bvh::hit_aabb(r, nodes[c1].aabbMin, nodes[c1].aabbMax);
            float d2 = Bvh::hit_aabb(r, nodes[c2].aabbMin, nodes[c2].aabbMax);
            if (d1 > d2) { float td = d1; d1 = d2; d2 = td; uint tc = c1; c1 = c2; c2 = tc; }
            if (d1 == 1e30f) { if (sp == 0) break; else node = stack[--sp]; }
            else { node = c1; if (d2 != 1e30f) stack[sp++] = c2; }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// textures / materials / analytic primitives
// ------------------------------------------------------------------------------------------------
struct Tex {
    std::vector<uint32_t> px; int w = 0, h = 0;
    V3 sample(float u, float v) const // template/texture.h:61-96
    {
        if (px.empty()) return v3(0);
        u = clampf(u, 0.0f, 1.0f);
        v = 1 - clampf(v, 0.0f, 1.0f);
        int x = (int)(u * w), y = (int)(v * h);
        x = clampi(x, 0, w - 1); y = clampi(y, 0, h - 1);
        uint32_t p = px[(size_t)x + (size_t)y * w];
        float s = 1 / 255.0f;
        return v3(((p >> 16) & 0xFF) * s, ((p >> 8) & 0xFF) * s, (p & 0xFF) * s);
    }
};
struct Mat { // template/material.h
    bool isLight = false; V3 albedo = v3(1.0f);
    float reflectivity = 0, refractivity = 0; V3 absorption = v3(0.0f);
    bool hasTex = false; Tex tex;
    V3 get_albedo(V2 uv) const { return hasTex ? tex.sample(uv.x, uv.y) : albedo; }
};
struct PlanePrim { // template/primitives.h:100-179
    V3 N = v3(0, 1, 0); float d = 1; int objIdx = 1; float invto = 1.f;
    void intersect(Ray& r) const
    {
        float t = -(dot(r.O, N) + d) / (dot(r.D, N));
        if (t < r.t && t > 0) { r.t = t; r.objIdx = objIdx; }
    }
    V2 uv(V3 I) const
    {
        V2 o; o.x = 0; o.y = 0;
        if (N.y == 1) {
            float u = I.x, v = I.z;
            u *= invto; v *= invto;
            u = u - floorf(u); v = v - floorf(v);
            o.x = u; o.y = v;
        }
        return o;
    }
};
struct QuadPrim { // template/primitives.h:321-375
    float size = 0.5f; M4 T = m4_identity(), invT = m4_identity(); int objIdx = 0;
    void intersect(Ray& r) const
    {
        const float* c = invT.c;
        const float Oy = c[4] * r.O.x + c[5] * r.O.y + c[6] * r.O.z + c[7];
        const float Dy = c[4] * r.D.x + c[5] * r.D.y + c[6] * r.D.z;
        const float t = Oy / -Dy;
        if (t < r.t && t > 0) {
            const float Ox = c[0] * r.O.x + c[1] * r.O.y + c[2] * r.O.z + c[3];
            const float Oz = c[8] * r.O.x + c[9] * r.O.y + c[10] * r.O.z + c[11];
            const float Dx = c[0] * r.D.x + c[1] * r.D.y + c[2] * r.D.z;
            const float Dz = c[8] * r.D.x + c[9] * r.D.y + c[10] * r.D.z;
            const float Ix = Ox + t * Dx, Iz = Oz + t * Dz;
            if (Ix > -size && Ix < size && Iz > -size && Iz < size) { r.t = t; r.objIdx = objIdx; }
        }
    }
    bool occluded(const Ray& r) const
    {
        const float* c = invT.c;
        const float Oy = c[4] * r.O.x + c[5] * r.O.y + c[6] * r.O.z + c[7];
        const float Dy = c[4] * r.D.x + c[5] * r.D.y + c[6] * r.D.z;
        const float t = Oy / -Dy;
        if (t < r.t && t > 0) {
            const float Ox = c[0] * r.O.x + c[1] * r.O.y + c[2] * r.O.z + c[3];
            const float Oz = c[8] * r.O.x + c[9] * r.O.y + c[10] * r.O.z + c[11];
            const float Dx = c[0] * r.D.x + c[1] * r.D.y + c[2] * r.D.z;
            const float Dz = c[8] * r.D.x + c[9] * r.D.y + c[10] * r.D.z;
            const float Ix = Ox + t * Dx, Iz = Oz + t * Dz;
            return Ix > -size && Ix < size && Iz > -size && Iz < size;
        }
        return false;
    }
    V3 normal() const { return v3(-T.c[1], -T.c[5], -T.c[9]); }
};


// ------------------------------------------------------------------------------------------------
// PrimitiveScene (second half): infra/scene/primitive_scene.cpp + template/primitives.h (Sphere :31, Cube :187, Quad :321, Torus :380), the
// SPEEDTRIX / single-light configuration the headers select.  PARITY UNPINNED: neither file compiles with the compilers of this image (MSVC-only __m128 member
// access) and the reference holds no fixture of this scene.  The torus solves its quartic in double precision with cos(acos(x) / 3): the reference takes both from
// the C runtime (unspecified to the bit); here they are det_acos / det_cos below — plain IEEE double + - * / sqrt, the published fdlibm algorithms (e_acos.c,
// k_cos.c, k_sin.c, one-step e_rem_pio2.c) — so that this file and the HIP kernels (render_prim.hip) agree bit for bit.
// ------------------------------------------------------------------------------------------------
static inline uint32_t hi_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline double with_words(uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double d; memcpy(&d, &u, 8); return d; }
static double det_acos(double x)
{
    const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
        pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
        pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
        qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const uint32_t hx = hi_word(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {                                  // |x| >= 1
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);                             // NaN
    }
    if (ix < 0x3fe00000u) {                                   // |x| < 0.5
        if (ix <= 0x3c600000u) return pio2_hi + pio2_lo;
        const double z = x * x;
        const double pp = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = pp / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx & 0x80000000u) {                            // x < -0.5
        const double z = (one + x) * 0.5;
        const double pp = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double sq = sqrt(z), r = pp / q, w = r * sq - pio2_lo;
        return pi - 2.0 * (sq + w);
    } else {                                                  // x > 0.5
        const double z = (one - x) * 0.5, sq = sqrt(z);
        const double df = with_words(hi_word(sq), 0u);
        const double c = (z - df * df) / (sq + df);
        const double pp = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = pp / q, w = r * sq + c;
        return 2.0 * (df + w);
    }
}
static double det_cos(double x)                               // 0 <= x < 3 pi / 4 is all the torus needs (acos(..) / 3 <= pi / 3); other inputs: NaN
{
    const double one = 1.0, C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11, S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
        pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const uint32_t ix = hi_word(x) & 0x7fffffffu;
    if (!(x >= 0.0) || ix >= 0x4002d97cu) return (x - x) / (x - x);
    if (ix <= 0x3fe921fbu) {                                  // |x| <= pi / 4: __kernel_cos(x, 0)
        if (ix < 0x3e400000u) return one;
        const double z = x * x;
        const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        if (ix < 0x3fd33333u) return one - (0.5 * z - (z * r - x * 0.0));
        const double qx = (ix > 0x3fe90000u) ? 0.28125 : with_words(ix - 0x00200000u, 0u);
        const double hz = 0.5 * z - qx, a = one - qx;
        return a - (hz - (z * r - x * 0.0));
    }
    // pi / 4 < x < 3 pi / 4: x = pi / 2 + y, cos(x) = -sin(y) (__ieee754_rem_pio2 with n = 1, then __kernel_sin(y0, y1, 1))
    double z = x - pio2_1, y0, y1;
    if (ix != 0x3ff921fbu) { y0 = z - pio2_1t; y1 = (z - y0) - pio2_1t; }
    else { z -= pio2_2; y0 = z - pio2_2t; y1 = (z - y0) - pio2_2t; }
    const double zz = y0 * y0, v = zz * y0;
    const double r = S2 + zz * (S3 + zz * (S4 + zz * (S5 + zz * S6)));
    const double sn = y0 - ((zz * (0.5 * y1 - v * r) - y1) - v * S1);
    return -sn;
}
static double cbrt_fast(double n)                             // Torus::cbrtFast, primitives.h:548-556 (float literals as written: 10.0f, 2.0f / 3.0f, 3.0f)
{
    double x1 = n / 10.0f, x2 = 1.0f; int turn = 0;
    while (fabs(x1 - x2) > 0.00000001 && turn++ < 100) { x1 = x2; x2 = (2.0f / 3.0f * x1) + (n / (3.0f * x1 * x1)); }
    return x2;
}
static M4 m4_inverted(const M4& m)                            // mat4::Inverted, tmplmath.h:769-813: the MESA cofactor formula, terms in the reference's order
{
    static const signed char T[16][6][4] = {
        {{1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {1, 9, 7, 14}, {1, 13, 6, 11}, {-1, 13, 7, 10}}, {{-1, 1, 10, 15}, {1, 1, 11, 14}, {1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {1, 13, 3, 10}},
        {{1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {1, 5, 3, 14}, {1, 13, 2, 7}, {-1, 13, 3, 6}}, {{-1, 1, 6, 11}, {1, 1, 7, 10}, {1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {1, 9, 3, 6}},
        {{-1, 4, 10, 15}, {1, 4, 11, 14}, {1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {1, 12, 7, 10}}, {{1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {1, 8, 3, 14}, {1, 12, 2, 11}, {-1, 12, 3, 10}},
        {{-1, 0, 6, 15}, {1, 0, 7, 14}, {1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {1, 12, 3, 6}}, {{1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {1, 4, 3, 10}, {1, 8, 2, 7}, {-1, 8, 3, 6}},
        {{1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {1, 8, 7, 13}, {1, 12, 5, 11}, {-1, 12, 7, 9}}, {{-1, 0, 9, 15}, {1, 0, 11, 13}, {1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {1, 12, 3, 9}},
        {{1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {1, 4, 3, 13}, {1, 12, 1, 7}, {-1, 12, 3, 5}}, {{-1, 0, 5, 11}, {1, 0, 7, 9}, {1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {1, 8, 3, 5}},
        {{-1, 4, 9, 14}, {1, 4, 10, 13}, {1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {1, 12, 6, 9}}, {{1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {1, 8, 2, 13}, {1, 12, 1, 10}, {-1, 12, 2, 9}},
        {{-1, 0, 5, 14}, {1, 0, 6, 13}, {1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {1, 12, 2, 5}}, {{1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {1, 4, 2, 9}, {1, 8, 1, 6}, {-1, 8, 2, 5}}};
    float inv[16];
    for (int i = 0; i < 16; i++) {
        float acc = 0;
        for (int k = 0; k < 6; k++) {
            const float t = m.c[T[i][k][1]] * m.c[T[i][k][2]] * m.c[T[i][k][3]];
            if (k == 0) acc = T[i][k][0] < 0 ? -t : t; else acc = T[i][k][0] < 0 ? acc - t : acc + t;
        }
        inv[i] = acc;
    }
    const float det = m.c[0] * inv[0] + m.c[1] * inv[4] + m.c[2] * inv[8] + m.c[3] * inv[12];
    M4 r = m4_identity();
    if (det != 0) { const float invdet = 1.0f / det; for (int i = 0; i < 16; i++) r.c[i] = inv[i] * invdet; }
    return r;
}

struct PrimScene {
    QuadPrim quad; V3 spherePos = v3(0); V3 cubeMin = v3(0), cubeMax = v3(0); M4 cubeM = m4_identity(), cubeInvM = m4_identity();
    M4 torusT = m4_identity(), torusInvT = m4_identity(); float rt2 = 0, rc2 = 0, r2 = 0;
    Tex red, blue;
    Mat materials[11];
    void construct()                                           // PrimitiveScene::PrimitiveScene, primitive_scene.cpp:4-42
    {
        quad.size = 1 * 0.5f; quad.objIdx = 0;
        cubeMin = v3(0) - 0.5f * v3(1.15f); cubeMax = v3(0) + 0.5f * v3(1.15f);
        const float a = 0.8f, b = 0.25f; rc2 = a * a; rt2 = b * b; r2 = (a + b) * (a + b);        // Torus(10, 0.8f, 0.25f)
        torusT = m4_mul(m4_translate(v3(-0.25f, 0, 2)), m4_rotx(kPI / 4)); torusInvT = m4_inverted(torusT);
        for (Mat& m : materials) m = Mat();
        materials[0].isLight = true; materials[1].reflectivity = 1.0f; materials[3].refractivity = 1.0f; materials[3].absorption = v3(0.5f, 0, 0.5f);
        materials[6].reflectivity = 0.3f; materials[10].refractivity = 1.0f;
        set_time(0);
    }
    void set_time(float t)                                     // PrimitiveScene::SetTime, primitive_scene.cpp:44-68
    {
        M4 M1 = m4_mul(m4_mul(m4_translate(v3(0, 2.6f, 2)), m4_rotz(sinf(t * 0.6f) * 0.1f)), m4_translate(v3(0, -0.9f, 0)));
        quad.T = M1; quad.invT = m4_fast_inverted_noscale(M1);
        M4 M2base = m4_mul(m4_rotx(kPI / 4), m4_rotz(kPI / 4));
        M4 M2 = m4_mul(m4_mul(m4_translate(v3(1.8f, 0, 2.5f)), m4_roty(t * 0.5f)), M2base);
        cubeM = M2; cubeInvM = m4_fast_inverted_noscale(M2);
        const float f = fmodf(t, 2.0f) - 1, tm = 1 - f * f;
        spherePos = v3(-1.8f, -0.4f + tm, 1);
    }
    static bool is_overridden(int objIdx) { return objIdx >= 4 && objIdx <= 6; }      // Material(true): left wall, right wall, floor
    void find_nearest(Ray& r) const                            // PrimitiveScene::FindNearest, primitive_scene.cpp:92-175 (SPEEDTRIX branches)
    {
        {   // room walls: per axis the plane the ray looks at (D >= 0: the far one), d = 0 - (O + x) * rD, first unconditional then strict <
            const float xmin[3] = {3, 1, 3}, xmax[3] = {-2.99f, -2, -3.99f};
            float t3[3]; int id3[3];
            for (int a = 0; a < 3; a++) {
                const bool sel = comp(r.D, a) >= 0;
                const float x = sel ? xmax[a] : xmin[a];
                id3[a] = sel ? 5 + 2 * a : 4 + 2 * a;
                const float d = 0.0f - ((comp(r.O, a) + x) * comp(r.rD, a));
                t3[a] = (d <= 0.0f) ? 1e34f : d;
            }
            r.t = t3[0]; r.objIdx = id3[0];
            if (t3[1] < r.t) { r.t = t3[1]; r.objIdx = id3[1]; }
            if (t3[2] < r.t) { r.t = t3[2]; r.objIdx = id3[2]; }
        }
        quad.intersect(r);
        {   // bouncing ball (r = 0.6): front intersection only
            const V3 oc = r.O - spherePos;
            const float b = dot(oc, r.D);
            const float d = b * b - (dot(oc, oc) - 0.36f);
            if (d > 0) { const float t = -b - sqrtf(d); if (t < r.t && t > 0) { r.t = t; r.objIdx = 1; } }
        }
        {   // rounded corners (r = 8, seen from inside): back intersection only
            const V3 oc = r.O - v3(0, 2.5f, -3.07f);
            const float b = dot(oc, r.D);
            const float d = b * b - (dot(oc, oc) - 64.0f);
            if (d > 0) { const float t = sqrtf(d) - b; if (t < r.t && t > 0) { r.t = t; r.objIdx = 2; } }
        }
        {   // Cube::Intersect, primitives.h:200-225 (SSE summation order: (x + y) + (z + w) for O, (x + y) + z for D; _mm_min_ps / _mm_max_ps operand order)
            const V3 O = transform_position_sse(r.O, cubeInvM), D = transform_vector_sse(r.D, cubeInvM);
            const V3 rd = v3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
            const V3 t1 = (cubeMin - O) * rd, t2 = (cubeMax - O) * rd;
            const V3 vmaxv = v3(mm_max(t1.x, t2.x), mm_max(t1.y, t2.y), mm_max(t1.z, t2.z)), vminv = v3(mm_min(t1.x, t2.x), mm_min(t1.y, t2.y), mm_min(t1.z, t2.z));
            const float tmax = smin_(vmaxv.x, smin_(vmaxv.y, vmaxv.z)), tmin = smax_(vminv.x, smax_(vminv.y, vminv.z));
            if (tmin < tmax) { if (tmin > 0) { if (tmin < r.t) { r.t = tmin; r.objIdx = 3; } } else if (tmax > 0) { if (tmax < r.t) { r.t = tmax; r.objIdx = 3; } } }
        }
        torus_intersect(r);
    }
    void torus_intersect(Ray& ray) const                       // Torus::Intersect, primitives.h:386-453 (double precision)
    {
        const V3 O = transform_position_sse(ray.O, torusInvT), D = transform_vector_sse(ray.D, torusInvT);
        double po = 1, m = dot(O, O), k3 = dot(O, D), k32 = k3 * k3;
        const double v = k32 - m + r2;
        if (v < 0) return;
        double k = (m - rt2 - rc2) * 0.5, k2 = k32 + rc2 * D.z * D.z + k;
        double k1 = k * k3 + rc2 * O.z * D.z, k0 = k * k + rc2 * O.z * O.z - rc2 * rt2;
        if (fabs(k3 * (k32 - k2) + k1) < 0.0001) {
            const double tmp = k1; k1 = k3; k3 = tmp;
            po = -1; k0 = 1 / k0; k1 = k1 * k0; k2 = k2 * k0; k3 = k3 * k0; k32 = k3 * k3;
        }
        double c2 = 2 * k2 - 3 * k32, c1 = k3 * (k32 - k2) + k1;
        double c0 = k3 * (k3 * (-3 * k32 + 4 * k2) - 8 * k1) + 4 * k0;
        c2 *= 0.33333333333; c1 *= 2; c0 *= 0.33333333333;
        const double Q = c2 * c2 + c0, R = 3 * c0 * c2 - c2 * c2 * c2 - c1 * c1;
        double h = R * R - Q * Q * Q, z;
        if (h < 0) { const double sQ = sqrt(Q); z = 2 * sQ * det_cos(det_acos(R / (sQ * Q)) * 0.33333333333); }
        else { const double sQ = cbrt_fast(sqrt(h) + fabs(R)); z = copysign(fabs(sQ + Q / sQ), R); }
        z = c2 - z;
        double d1 = z - 3 * c2, d2 = z * z - 3 * c0;
        if (fabs(d1) < 1.0e-8) { if (d2 < 0) return; d2 = sqrt(d2); }
        else { if (d1 < 0) return; d1 = sqrt(d1 * 0.5); d2 = c1 / d1; }
        double t = 1e20;
        h = d1 * d1 - z + d2;
        if (h > 0) {
            h = sqrt(h);
            double t1 = -d1 - h - k3, t2 = -d1 + h - k3;
            t1 = (po < 0) ? 2 / t1 : t1; t2 = (po < 0) ? 2 / t2 : t2;
            if (t1 > 0) t = t1;
            if (t2 > 0) t = (t2 < t) ? t2 : t;
        }
        h = d1 * d1 - z - d2;
        if (h > 0) {
            h = sqrt(h);
            double t1 = d1 - h - k3, t2 = d1 + h - k3;
            t1 = (po < 0) ? 2 / t1 : t1; t2 = (po < 0) ? 2 / t2 : t2;
            if (t1 > 0) t = (t1 < t) ? t1 : t;
            if (t2 > 0) t = (t2 < t) ? t2 : t;
        }
        const float ft = (float)t;
        if (ft > 0 && ft < ray.t) { ray.t = ft; ray.objIdx = 10; }
    }
    V3 normal(int objIdx, V3 I) const                          // PrimitiveScene::GetHitInfo, primitive_scene.cpp:202-236 (before the flip towards the ray)
    {
        switch (objIdx) {
        case 0: return quad.normal();
        case 1: return (I - spherePos) * (1 / 0.6f);
        case 2: return (I - v3(0, 2.5f, -3.07f)) * (1 / 8.0f);
        case 3: {                                              // Cube::GetNormal, primitives.h:276-291
            const V3 o = transform_position(I, cubeInvM);
            V3 N = v3(-1, 0, 0);
            const float d0 = fabsf(o.x - cubeMin.x), d1 = fabsf(o.x - cubeMax.x), d2 = fabsf(o.y - cubeMin.y), d3 = fabsf(o.y - cubeMax.y), d4 = fabsf(o.z - cubeMin.z), d5 = fabsf(o.z - cubeMax.z);
            float minDist = d0;
            if (d1 < minDist) { minDist = d1; N.x = 1; }
            if (d2 < minDist) { minDist = d2; N = v3(0, -1, 0); }
            if (d3 < minDist) { minDist = d3; N = v3(0, 1, 0); }
            if (d4 < minDist) { minDist = d4; N = v3(0, 0, -1); }
            if (d5 < minDist) { minDist = d5; N = v3(0, 0, 1); }
            return transform_vector(N, cubeM);
        }
        case 10: {                                             // Torus::GetNormal, primitives.h:525-530
            const V3 L = transform_position(I, torusInvT);
            const V3 N = normalize(L * (v3(dot(L, L) - rt2) - rc2 * v3(1, 1, -1)));
            return transform_vector(N, torusT);
        }
        default: { V3 N = v3(0); const float s = 1 - 2 * (float)(objIdx & 1); const int a = (objIdx - 4) / 2; if (a == 0) N.x = s; else if (a == 1) N.y = s; else N.z = s; return N; }
        }
    }
    V3 albedo_override(int objIdx, V3 I) const                 // PrimitiveScene::GetAlbedo -> Plane::GetAlbedo, primitives.h:134-172, for the planes whose material overrides
    {
        if (objIdx == 6) {                                     // floor, N.y == 1: checkerboard
            int ix = (int)(I.x * 2 + 96.01f), iz = (int)(I.z * 2 + 96.01f);
            if (ix == 98 && iz == 98) { ix = (int)(I.x * 32.01f); iz = (int)(I.z * 32.01f); }
            if (ix == 94 && iz == 98) { ix = (int)(I.x * 64.01f); iz = (int)(I.z * 64.01f); }
            return v3(((ix + iz) & 1) ? 1 : 0.3f);
        }
        const Tex& tx = (objIdx == 4) ? red : blue;            // N.x == 1: red.png, N.x == -1: blue.png (512 x 512)
        const int ix = (int)((I.z - 4) * (512.0f / 7)), iy = (int)((2 - I.y) * (512.0f / 3));
        const uint32_t px = tx.px.empty() ? 0u : tx.px[(size_t)(ix & 511) + (size_t)(iy & 511) * 512];
        return v3((float)((px >> 16) & 255), (float)((px >> 8) & 255), (float)(px & 255)) * (1.0f / 255.0f);
    }
};

struct ObjDesc {
    std::vector<float> pos, nrm, uv; int nCorners = 0;
    V3 position, rotation, scale; int matIdx = 0;
};

struct HitInfo { V3 N; V2 uv; const Mat* mat; };

} // namespace

// ------------------------------------------------------------------------------------------------
// context = scene (FileScene | TLASFileScene) + Renderer state
// ------------------------------------------------------------------------------------------------
// FileScene built with USE_KDTree / USE_Grid (infra/scene/file_scene.h:10-12): `acc` of FindNearest / IsOccluded is that structure (defined at the end of this file)

struct orc_ctx {
    int kind = 0;
    std::string err;
    // scene description
    V3 lightPos = v3(0, 3, 1);
    Tex floorTex, sky;
    std::vector<Mat> materials;
    std::vector<ObjDesc> objects;
    // built scene
    bool built = false;
    Mat primMat[2];
    PlanePrim floor; QuadPrim light;
    PrimScene prim;                     // kind 2: PrimitiveScene (the scene IS its eleven hard-coded primitives; no BVH, no XML)
    std::vector<Bvh*> bvhs;            // kind 0: one; kind 1: one per object
    std::vector<int> objMat;           // FileScene: models[objIdx-2]->matIdx
    Tlas tlas;
    // renderer
    int W = 0, H = 0; float aspect = 1;
    V3 camPos, camTarget, topLeft, topRight, bottomLeft;
    std::vector<float> acc; std::vector<uint32_t> screen;
    int spp = 1, passes = 1, depthLimit = 5;
    float energy = 0;
    int tileFirst = 0, tileCount = -1;
    Counters counters;
    std::vector<uint32_t> tileSeedOut; int tileSeedSpp = -1;

    void (*accelFn)(void*, Ray&) = nullptr;     // (set by orc_set_render_accel, where KdTree / UGrid are defined)
    int accel = 0; void* accelH = nullptr;      // orc_set_render_accel: 0 = BVH / TLAS, 1 = KD-tree, 2 = grid (a FileScene's `acc`; handle of orc_kd_build / orc_grid_build)

    ~orc_ctx() { for (Bvh* b : bvhs) delete b; }

    // ---- scene queries ----
    void accel_intersect(Ray& r, Counters& cn) const   // acc.Intersect(ray): file_scene.cpp:174, 183
    {
        if (kind == 0 && accel != 0) accelFn(accelH, r);
        else if (kind == 0) bvhs[0]->traverse(r, cn); else tlas.traverse(r, cn);
    }
    void find_nearest(Ray& r, Counters& cn) const // file_scene.cpp:170-175, tlas_file_scene.cpp:201-206; primitive_scene.cpp:92-175
    {
        cn.rays++;
        if (kind == 2) { prim.find_nearest(r); return; }
        light.intersect(r);
        floor.intersect(r);
        accel_intersect(r, cn);
        if (r.objIdx >= 2) cn.meshhits++;
    }
    V3 sky_color(const Ray& r) const // file_scene.cpp:142-154
    {
        if (kind == 2) return v3(0);                                         // PrimitiveScene::GetSkyColor
        float phi = det_atan2f(-r.D.z, r.D.x) + kPI;
        float theta = det_acosf(-r.D.y);
        float u = phi * kINV2PI, v = theta * kINVPI;
        return sky.sample(u, v);
    }
    HitInfo hit_info(const Ray& r, V3 I) const // file_scene.cpp:189-214, tlas_file_scene.cpp:220-260
    {
        HitInfo h; h.N = v3(0); h.uv.x = 0; h.uv.y = 0; h.mat = nullptr;
        if (kind == 2) { h.N = prim.normal(r.objIdx, I); h.mat = &prim.materials[r.objIdx]; }
        else if (r.objIdx == 0) { h.N = light.normal(); h.mat = &primMat[0]; }
        else if (r.objIdx == 1) { h.N = floor.N; h.uv = floor.uv(I); h.mat = &primMat[1]; }
        else {
            const Bvh* b = (kind == 0) ? bvhs[0] : tlas.blas[r.objIdx - 2];
            const Tri& t = b->tris[r.triIdx];
            float w = 1 - r.bu - r.bv;
            V3 N = w * ld3(t.normal0) + r.bu * ld3(t.normal1) + r.bv * ld3(t.normal2);
            if (kind == 0) h.N = normalize(N);                               // bvh.cpp:290-297
            else h.N = normalize(transform_vector(N, b->T));                 // blas_bvh.cpp:391-398
            h.uv.x = w * t.uv0[0] + r.bu * t.uv1[0] + r.bv * t.uv2[0];
            h.uv.y = w * t.uv0[1] + r.bu * t.uv1[1] + r.bv * t.uv2[1];
            int m = (kind == 0) ? objMat[t.objIdx - 2] : b->matIdx;
            h.mat = &materials[m];
        }
        if (dot(h.N, r.D) > 0) h.N = -h.N;
        return h;
    }
    static V3 diffuse_reflection(V3 N, uint& seed) // tmplmath.h:535-544, draw order pinned z,y,x
    {
        V3 R;
        do {
            float rz = random_float(seed) * 2 - 1;
            float ry = random_float(seed) * 2 - 1;
            float rx = random_float(seed) * 2 - 1;
            R = v3(rx, ry, rz);
        } while (dot(R, R) > 1);
        if (dot(R, N) < 0) R = R * -1.0f;
        return normalize(R);
    }
    // 3. PathTracer/renderer.cpp:50-100 (+ HandleMirror :20-25, HandleDielectric :27-45)
    V3 sample(Ray& ray, uint& seed, int depth, Counters& cn) const
    {
        find_nearest(ray, cn);
        if (ray.objIdx == -1) return sky_color(ray);
        if (depth >= depthLimit) return v3(0);
        V3 I = ray.O + ray.t * ray.D;
        HitInfo hi = hit_info(ray, I);
        V3 N = hi.N;
        V3 albedo = (kind == 2 && PrimScene::is_overridden(ray.objIdx)) ? prim.albedo_override(ray.objIdx, I) : hi.mat->get_albedo(hi.uv);   // renderer.cpp:61 (isAlbedoOverridden)
        if (hi.mat->isLight) return v3(24, 24, 22);
        float reflectivity = hi.mat->reflectivity, refractivity = hi.mat->refractivity;
        V3 medium = v3(1);
        if (ray.inside) {
            V3 a = hi.mat->absorption * -ray.t;
            medium = v3(det_expf(a.x), det_expf(a.y), det_expf(a.z));
        }
        float r = random_float(seed);
        if (r < reflectivity) {
            V3 R = reflect(ray.D, N);
            Ray nr = make_ray(I + R * kEPS, R);
            return albedo * medium * sample(nr, seed, depth + 1, cn);
        } else if (r < reflectivity + refractivity) {
            V3 R = reflect(ray.D, N);
            Ray rr = make_ray(I + R * kEPS, R);
            float n1 = ray.inside ? 1.2f : 1, n2 = ray.inside ? 1 : 1.2f;
            float eta = n1 / n2, cosi = dot(-ray.D, N);
            float cost2 = 1.0f - eta * eta * (1 - cosi * cosi);
            float Fr = 1;
            if (cost2 > 0) {
                float a = n1 - n2, b = n1 + n2, R0 = (a * a) / (b * b), c = 1 - cosi;
                Fr = R0 + (1 - R0) * (c * c * c * c * c);
                V3 T = eta * ray.D + ((eta * cosi - sqrtf(fabsf(cost2))) * N);
                Ray tr = make_ray(I + T * kEPS, T);
                tr.inside = !ray.inside;
                if (random_float(seed) > Fr) return albedo * medium * sample(tr, seed, depth + 1, cn);
            }
            return albedo * medium * sample(rr, seed, depth + 1, cn);
        } else {
            V3 R = diffuse_reflection(N, seed);
            V3 brdf = albedo * kINVPI;
            Ray nr = make_ray(I + R * kEPS, R);
            return medium * brdf * 2 * kPI * dot(R, N) * sample(nr, seed, depth + 1, cn);
        }
    }
    Ray primary_ray(float x, float y) const // template/camera.h:23-30
    {
        const float u = x * (1.0f / W), v = y * (1.0f / H);
        V3 P = topLeft + u * (topRight - topLeft) + v * (bottomLeft - topLeft);
        return make_ray(camPos, normalize(P - camPos));
    }
    static uint32_t rgb8(float x, float y, float z) // template/precomp.h:325-341 (scalar branch)
    {
        uint r = (uint)(255.0f * smin_(1.0f, x)), g = (uint)(255.0f * smin_(1.0f, y)), b = (uint)(255.0f * smin_(1.0f, z));
        return (r << 16) + (g << 8) + b;
    }
    // renderer.cpp:117-131
    void process_tile(int tx, int ty, float& sum, Counters& cn, uint32_t* seedOut)
    {
        float scale = 1.0f / (spp + passes);
        uint seed = init_seed((uint)(tx + ty * W + spp * 1799));
        for (int y = ty * 16, v = 0; v < 16; v++, y++) for (int x = tx * 16, u = 0; u < 16; u++, x++) {
            float* a = &acc[4 * ((size_t)x + (size_t)y * W)];
            for (int p = 0; p < passes; p++) {
                float jy = random_float(seed);     // pinned: right-to-left argument evaluation
                float jx = random_float(seed);
                Ray pr = primary_ray((float)x + jx, (float)y + jy);
                cn.primary++;
                V3 c = sample(pr, seed, 0, cn);
                a[0] += c.x; a[1] += c.y; a[2] += c.z; a[3] += 0.0f;
            }
            float px = a[0] * scale, py = a[1] * scale, pz = a[2] * scale;
            sum += px + py + pz;
            screen[(size_t)x + (size_t)y * W] = rgb8(px, py, pz);
        }
        if (seedOut) *seedOut = seed;
    }
};

// ------------------------------------------------------------------------------------------------
// scene construction
// ------------------------------------------------------------------------------------------------
namespace {

struct VKey { uint32_t b[8]; bool operator==(const VKey& o) const { return memcmp(b, o.b, sizeof(b)) == 0; } };
struct VKeyHash { size_t operator()(const VKey& k) const { size_t h = 1469598103934665603ull; for (int i = 0; i < 8; i++) { h ^= k.b[i]; h *= 1099511628211ull; } return h; } };

// vertex de-duplication (infra/model.cpp:16-54, infra/blas_bvh.cpp:16-56): equal vertices (float ==, so
// -0 == +0) share the FIRST occurrence's value
static void dedup(const ObjDesc& o, std::vector<float>& P, std::vector<float>& Nn, std::vector<float>& U, std::vector<uint32_t>& idx)
{
    std::unordered_map<VKey, uint32_t, VKeyHash> seen;
    for (int c = 0; c < o.nCorners; c++) {
        float v[8] = {o.pos[3 * c], o.pos[3 * c + 1], o.pos[3 * c + 2],
                      o.nrm.empty() ? 0.0f : o.nrm[3 * c], o.nrm.empty() ? 0.0f : o.nrm[3 * c + 1], o.nrm.empty() ? 0.0f : o.nrm[3 * c + 2],
                      o.uv.empty() ? 0.0f : o.uv[2 * c], o.uv.empty() ? 0.0f : o.uv[2 * c + 1]};
        VKey k; bool nan = false;
        for (int i = 0; i < 8; i++) { uint32_t b = f2bits(v[i]); if (b == 0x80000000u) b = 0; k.b[i] = b; nan = nan || v[i] != v[i]; }
        if (nan) {
            // a vertex with a NaN component equals nothing, not even itself: `uniqueVertices[vertex] = size` appends it, and the second
            // `uniqueVertices[vertex]` of model.cpp:50 misses again and value-initialises a fresh entry: the corner gets index 0
            P.insert(P.end(), v, v + 3); Nn.insert(Nn.end(), v + 3, v + 6); U.insert(U.end(), v + 6, v + 8);
            idx.push_back(0u);
            continue;
        }
        auto it = seen.find(k);
        uint32_t id;
        if (it == seen.end()) {
            id = (uint32_t)(P.size() / 3);
            seen.emplace(k, id);
            P.insert(P.end(), v, v + 3); Nn.insert(Nn.end(), v + 3, v + 6); U.insert(U.end(), v + 6, v + 8);
        } else id = it->second;
        idx.push_back(id);
    }
}

static const float kDeg2Rad = (kPI * 2) / 360.0f; // infra/helper.h:152

} // namespace

extern "C" {

orc_ctx* orc_create(int kind) { orc_ctx* c = new orc_ctx(); c->kind = kind == 2 ? 2 : (kind ? 1 : 0); if (c->kind == 2) { c->prim.construct(); c->built = true; } return c; }
// PrimitiveScene (kind 2): the two wall images (red.png / blue.png as Surface loads them: 0x00RRGGBB, 512 x 512; NULL = black) and the animation time
int orc_prim_setup(orc_ctx* c, const uint32_t* red, const uint32_t* blue)
{
    if (!c || c->kind != 2) return -1;
    if (red) { c->prim.red.px.assign(red, red + 512 * 512); c->prim.red.w = c->prim.red.h = 512; }
    if (blue) { c->prim.blue.px.assign(blue, blue + 512 * 512); c->prim.blue.w = c->prim.blue.h = 512; }
    return 0;
}
int orc_prim_set_time(orc_ctx* c, float t) { if (!c || c->kind != 2) return -1; c->prim.set_time(t); return 0; }
// 16 x 5 matrices (quad T, invT, cube M, invM, torus T... invT) + sphere position + torus radii: what a host front must reproduce
int orc_prim_state(orc_ctx* c, float* out)
{
    if (!c || c->kind != 2) return -1;
    const M4* ms[6] = {&c->prim.quad.T, &c->prim.quad.invT, &c->prim.cubeM, &c->prim.cubeInvM, &c->prim.torusT, &c->prim.torusInvT};
    for (int k = 0; k < 6; k++) memcpy(out + 16 * k, ms[k]->c, 64);
    st3(out + 96, c->prim.spherePos); out[99] = c->prim.rt2; out[100] = c->prim.rc2; out[101] = c->prim.r2; st3(out + 102, c->prim.cubeMin); st3(out + 105, c->prim.cubeMax);
    return 0;
}
double orc_det_acos(double x) { return det_acos(x); }
double orc_det_cos(double x) { return det_cos(x); }
void orc_destroy(orc_ctx* c) { delete c; }
const char* orc_last_error(orc_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int orc_set_light_position(orc_ctx* c, const float p[3]) { c->lightPos = v3(p[0], p[1], p[2]); return 0; }
static int set_tex(Tex& t, const uint32_t* rgb, int w, int h)
{
    t.w = w; t.h = h; t.px.assign(rgb, rgb + (size_t)w * h); return 0;
}
int orc_set_floor_texture(orc_ctx* c, const uint32_t* rgb, int w, int h) { return set_tex(c->floorTex, rgb, w, h); }
int orc_set_skydome(orc_ctx* c, const uint32_t* rgb, int w, int h) { return set_tex(c->sky, rgb, w, h); }
int orc_add_material(orc_ctx* c, float refl, float refr, const float ab[3], const uint32_t* tex, int w, int h)
{
    Mat m; m.reflectivity = refl; m.refractivity = refr; m.absorption = v3(ab[0], ab[1], ab[2]);
    if (tex) { m.hasTex = true; set_tex(m.tex, tex, w, h); }
    c->materials.push_back(m);
    return (int)c->materials.size() - 1;
}
int orc_add_object(orc_ctx* c, const float* pos, const float* nrm, const float* uv, int n, const float p[3], const float r[3], const float s[3], int mat)
{
    if (n <= 0 || n % 3) { c->err = "n_corners must be a positive multiple of 3"; return -1; }
    ObjDesc o; o.nCorners = n;
    o.pos.assign(pos, pos + 3 * (size_t)n);
    if (nrm) o.nrm.assign(nrm, nrm + 3 * (size_t)n);
    if (uv) o.uv.assign(uv, uv + 2 * (size_t)n);
    o.position = v3(p[0], p[1], p[2]); o.rotation = v3(r[0], r[1], r[2]); o.scale = v3(s[0], s[1], s[2]); o.matIdx = mat;
    c->objects.push_back(std::move(o));
    return (int)c->objects.size() - 1;
}

int orc_build(orc_ctx* c)
{
    if (c->floorTex.px.empty() || c->sky.px.empty()) { c->err = "floor texture and skydome are required (file_scene.cpp:12,22)"; return -1; }
    for (const ObjDesc& o : c->objects) if (o.matIdx < 0 || o.matIdx >= (int)c->materials.size()) { c->err = "material_idx out of range"; return -1; }
    for (Bvh* b : c->bvhs) delete b;
    c->bvhs.clear(); c->objMat.clear();
    // file_scene.cpp:10-19 / tlas_file_scene.cpp:10-19
    c->primMat[0] = Mat(); c->primMat[0].isLight = true;
    c->primMat[1] = Mat(); c->primMat[1].hasTex = true; c->primMat[1].tex = c->floorTex;
    c->light = QuadPrim(); c->light.objIdx = 0; c->light.size = 1 * 0.5f;
    c->floor = PlanePrim(); c->floor.objIdx = 1; c->floor.N = v3(0, 1, 0); c->floor.d = 1;
    c->floor.invto = 1.f / (float)(c->floorTex.w / 100);              // integer division (file_scene.cpp:16)
    M4 M1 = m4_translate(c->lightPos);
    c->light.T = M1; c->light.invT = m4_fast_inverted_noscale(M1);

    int objId = 2;
    if (c->kind == 0) {
        Bvh* acc = new Bvh();
        for (const ObjDesc& o : c->objects) {
            // file_scene.cpp:45-48
            M4 T = m4_mul(m4_mul(m4_mul(m4_mul(m4_translate(o.position), m4_rotx(o.rotation.x * kDeg2Rad)), m4_roty(o.rotation.y * kDeg2Rad)),
                                 m4_rotz(o.rotation.z * kDeg2Rad)), m4_scale(o.scale));
            M4 invT = m4_fast_inverted_noscale(T);
            std::vector<float> P, Nn, U; std::vector<uint32_t> idx;
            dedup(o, P, Nn, U, idx);
            // model.cpp:62-80
            for (size_t i = 0; i + 2 < idx.size(); i += 3) {
                Tri t; memset(&t, 0, sizeof(t));
                uint32_t a = idx[i], b = idx[i + 1], d = idx[i + 2];
                st3(t.vertex0, transform_position(ld3(&P[3 * a]), T));
                st3(t.vertex1, transform_position(ld3(&P[3 * b]), T));
                st3(t.vertex2, transform_position(ld3(&P[3 * d]), T));
                st3(t.normal0, normalize(transform_vector(ld3(&Nn[3 * a]), invT)));
                st3(t.normal1, normalize(transform_vector(ld3(&Nn[3 * b]), invT)));
                st3(t.normal2, normalize(transform_vector(ld3(&Nn[3 * d]), invT)));
                t.uv0[0] = U[2 * a]; t.uv0[1] = U[2 * a + 1]; t.uv1[0] = U[2 * b]; t.uv1[1] = U[2 * b + 1]; t.uv2[0] = U[2 * d]; t.uv2[1] = U[2 * d + 1];
                st3(t.centroid, (ld3(t.vertex0) + ld3(t.vertex1) + ld3(t.vertex2)) * 0.3333f);
                t.objIdx = objId;
                acc->tris.push_back(t);
            }
            c->objMat.push_back(o.matIdx);
            objId++;
        }
        if (acc->tris.empty()) { delete acc; c->err = "scene has no triangles"; return -1; }
        acc->build();
        c->bvhs.push_back(acc);
    } else {
        for (const ObjDesc& o : c->objects) {
            // tlas_file_scene.cpp:46-52
            M4 T = m4_mul(m4_mul(m4_mul(m4_translate(o.position), m4_rotx(o.rotation.x * kDeg2Rad)), m4_roty(o.rotation.y * kDeg2Rad)), m4_rotz(o.rotation.z * kDeg2Rad));
            M4 S = m4_scale(o.scale);
            Bvh* b = new Bvh(); b->blasObjIdx = objId; b->matIdx = o.matIdx;
            std::vector<float> P, Nn, U; std::vector<uint32_t> idx;
            dedup(o, P, Nn, U, idx);
            // blas_bvh.cpp:61-77
            for (size_t i = 0; i + 2 < idx.size(); i += 3) {
                Tri t; memset(&t, 0, sizeof(t));
                uint32_t a = idx[i], bb = idx[i + 1], d = idx[i + 2];
                st3(t.vertex0, transform_position(ld3(&P[3 * a]), S));
                st3(t.vertex1, transform_position(ld3(&P[3 * bb]), S));
                st3(t.vertex2, transform_position(ld3(&P[3 * d]), S));
                st3(t.normal0, ld3(&Nn[3 * a])); st3(t.normal1, ld3(&Nn[3 * bb])); st3(t.normal2, ld3(&Nn[3 * d]));
                t.uv0[0] = U[2 * a]; t.uv0[1] = U[2 * a + 1]; t.uv1[0] = U[2 * bb]; t.uv1[1] = U[2 * bb + 1]; t.uv2[0] = U[2 * d]; t.uv2[1] = U[2 * d + 1];
                st3(t.centroid, (ld3(t.vertex0) + ld3(t.vertex1) + ld3(t.vertex2)) * 0.3333f);
                t.objIdx = objId;
                b->tris.push_back(t);
            }
            if (b->tris.empty()) { delete b; c->err = "object has no triangles"; return -1; }
            b->build();
            b->set_transform(T);
            c->bvhs.push_back(b);
            objId++;
        }
        if (c->bvhs.empty()) { c->err = "scene has no objects"; return -1; }
        c->tlas = Tlas();
        c->tlas.blas = c->bvhs;
        if (!c->tlas.build(c->err)) return -1;
    }
    c->built = true;
    return 0;
}

int orc_bvh_count(orc_ctx* c) { return (int)c->bvhs.size(); }
int orc_bvh_info(orc_ctx* c, int i, uint32_t* nodesUsed, uint32_t* triCount, uint32_t* maxDepth)
{
    if (i < 0 || i >= (int)c->bvhs.size()) return -1;
    if (nodesUsed) *nodesUsed = c->bvhs[i]->nodesUsed;
    if (triCount) *triCount = (uint32_t)c->bvhs[i]->tris.size();
    if (maxDepth) *maxDepth = c->bvhs[i]->maxDepth;
    return 0;
}
int orc_bvh_copy(orc_ctx* c, int i, orc_bvh_node* nodes, uint32_t* triIndices, orc_tri* tris)
{
    if (i < 0 || i >= (int)c->bvhs.size()) return -1;
    Bvh* b = c->bvhs[i];
    if (nodes) memcpy(nodes, b->nodes.data(), sizeof(Node) * b->nodesUsed);
    if (triIndices) memcpy(triIndices, b->triIdx.data(), 4 * b->triIdx.size());
    if (tris) memcpy(tris, b->tris.data(), sizeof(Tri) * b->tris.size());
    return 0;
}
// moved vertices + Refit (the reference's animation hook; "next" row of SURVEY 8(f)): replaces the vertex positions of BVH i
// (3 x 3 floats per triangle, reference triangle order), refits it, and for a TLAS scene re-derives the instance's world bounds
// (SetTransform, blas_bvh.cpp:363-374) and rebuilds the TLAS (tlas_bvh.cpp:17-70) as a per-frame animation loop would
int orc_bvh_move_and_refit(orc_ctx* c, int i, const float* positions, uint32_t triCount)
{
    if (!c->built || i < 0 || i >= (int)c->bvhs.size()) return -1;
    Bvh* b = c->bvhs[i];
    if (triCount != b->tris.size()) return -1;
    for (uint32_t t = 0; t < triCount; t++) {
        memcpy(b->tris[t].vertex0, positions + 9 * (size_t)t, 12); memcpy(b->tris[t].vertex1, positions + 9 * (size_t)t + 3, 12);
        memcpy(b->tris[t].vertex2, positions + 9 * (size_t)t + 6, 12);
    }
    b->refit();
    if (c->kind == 1) { b->set_transform(b->T); if (!c->tlas.build(c->err)) return -1; }
    return 0;
}
// instance motion: BLASBVH::SetTransform(T) (blas_bvh.cpp:363-374) of BLAS i, then TLASBVH::Build (tlas_bvh.cpp:17-55) — what an animation loop does per frame
int orc_set_blas_transform(orc_ctx* c, int i, const float T[16])
{
    if (!c->built || c->kind != 1 || i < 0 || i >= (int)c->bvhs.size()) return -1;
    M4 m; memcpy(m.c, T, 64);
    c->bvhs[i]->set_transform(m);
    return c->tlas.build(c->err) ? 0 : -1;
}
int orc_blas_transform(orc_ctx* c, int i, float T[16], float invT[16], float lo[3], float hi[3])
{
    if (i < 0 || i >= (int)c->bvhs.size()) return -1;
    Bvh* b = c->bvhs[i];
    memcpy(T, b->T.c, 64); memcpy(invT, b->invT.c, 64); st3(lo, b->worldBounds.lo); st3(hi, b->worldBounds.hi);
    return 0;
}
int orc_tlas_copy(orc_ctx* c, orc_tlas_node* nodes, uint32_t* nodesUsed)
{
    if (c->kind != 1) return -1;
    if (nodes) memcpy(nodes, c->tlas.nodes.data(), sizeof(orc_tlas_node) * c->tlas.nodes.size());
    if (nodesUsed) *nodesUsed = c->tlas.nodesUsed;
    return 0;
}

int orc_renderer_init(orc_ctx* c, int W, int H)
{
    if (W <= 0 || H <= 0) { c->err = "bad resolution"; return -1; }
    c->W = W; c->H = H; c->aspect = (float)W / (float)H;
    // Camera() template/camera.h:14-22
    c->camPos = v3(0, 0, -2); c->camTarget = v3(0, 0, -1);
    c->topLeft = v3(-c->aspect, 1, 0); c->topRight = v3(c->aspect, 1, 0); c->bottomLeft = v3(-c->aspect, -1, 0);
    c->acc.assign(4 * (size_t)W * H, 0.0f); c->screen.assign((size_t)W * H, 0u);
    c->spp = 1; c->passes = 1; c->depthLimit = 5; c->energy = 0;
    return 0;
}
int orc_set_camera_state(orc_ctx* c, const float p[3], const float t[3]) // camera.h:61-73
{
    c->camPos = v3(p[0], p[1], p[2]); c->camTarget = v3(t[0], t[1], t[2]);
    V3 ahead = normalize(c->camTarget - c->camPos);
    V3 tmpUp = v3(0, 1, 0);
    V3 right = normalize(cross(tmpUp, ahead));
    V3 up = normalize(cross(ahead, right));
    right = normalize(cross(up, ahead));
    c->topLeft = c->camPos + 2 * ahead - c->aspect * right + up;
    c->topRight = c->camPos + 2 * ahead + c->aspect * right + up;
    c->bottomLeft = c->camPos + 2 * ahead - c->aspect * right - up;
    return 0;
}
int orc_get_camera(orc_ctx* c, float p[3], float tl[3], float tr[3], float bl[3])
{
    st3(p, c->camPos); st3(tl, c->topLeft); st3(tr, c->topRight); st3(bl, c->bottomLeft); return 0;
}
// Camera::GetPrimaryRay (template/camera.h:23-30) for given pixel coordinates (jitter included by the caller)
int orc_primary_rays(orc_ctx* c, const float* xy, size_t n, float* O, float* D)
{
    if (c->W <= 0) return -1;
    for (size_t i = 0; i < n; i++) {
        const Ray r = c->primary_ray(xy[2 * i], xy[2 * i + 1]);
        st3(O + 3 * i, r.O); st3(D + 3 * i, r.D);
    }
    return 0;
}
// Texture::Sample (template/texture.h:61-96) on caller-provided 0x00RRGGBB texels
int orc_texture_sample(const uint32_t* px, int w, int h, const float* uv, size_t n, float* rgb)
{
    Tex t; t.px.assign(px, px + (size_t)w * h); t.w = w; t.h = h;
    for (size_t i = 0; i < n; i++) st3(rgb + 3 * i, t.sample(uv[2 * i], uv[2 * i + 1]));
    return 0;
}
int orc_set_params(orc_ctx* c, int depthLimit, int passes) { c->depthLimit = depthLimit; c->passes = passes; return 0; }
int orc_clear(orc_ctx* c) { std::fill(c->acc.begin(), c->acc.end(), 0.0f); c->spp = 1; return 0; }
int orc_set_spp(orc_ctx* c, int spp) { c->spp = spp; return 0; }
int orc_get_spp(orc_ctx* c) { return c->spp; }
int orc_set_tile_range(orc_ctx* c, int first, int count) { c->tileFirst = first; c->tileCount = count; return 0; }

// One Tick (renderer.cpp:144-168) over the tiles [first, last), tiles handed out dynamically to `nThreads` workers.  Every worker keeps its
// counters and the tile's energy sum in locals (its own cache lines) and publishes them once per tile / once per frame: neighbouring
// entries of a shared array written on every traversal step would bounce between the cores' caches and serialise the workers.
namespace {
struct FramePool {                                   // persistent workers for a run of frames: one spawn / join per orc_render, not per Tick
    orc_ctx* c; int nThreads, frames, tw, first, last;
    std::atomic<int> next{0}, arrived{0}, generation{0};
    std::vector<float> sums;
    std::mutex mu;
    void frame_begin() { std::fill(sums.begin(), sums.end(), 0.0f); c->tileSeedOut.assign(sums.size(), 0u); c->tileSeedSpp = c->spp; next.store(first); }
    void frame_end() { c->energy = 0; for (float s : sums) c->energy += s; c->spp += c->passes; }
    void barrier(bool leaderWork)                    // sense-reversing spin barrier; the last arrival closes the frame and opens the next
    {
        const int gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nThreads) {
            if (leaderWork) { frame_end(); if (--frames > 0) frame_begin(); }
            arrived.store(0, std::memory_order_relaxed);
            generation.store(gen + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (generation.load(std::memory_order_acquire) == gen) { if (++spins > 2000) std::this_thread::yield(); }
        }
    }
    void worker()
    {
        Counters cn;
        for (;;) {
            for (;;) {
                const int i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= last) break;
                float sum = 0.0f; uint32_t seedOut = 0;
                c->process_tile(i % tw, i / tw, sum, cn, &seedOut);
                sums[i] = sum; c->tileSeedOut[i] = seedOut;
            }
            barrier(true);
            if (frames <= 0) break;
        }
        std::lock_guard<std::mutex> g(mu);
        c->counters.add(cn);
    }
};
} // namespace

int orc_render(orc_ctx* c, int frames, int nThreads)
{
    if (!c->built || c->W == 0) { c->err = "scene not built or renderer not initialised"; return -1; }
    if (frames <= 0) return 0;
    const int tw = c->W / 16, th = c->H / 16, tiles = tw * th;     // truncating division: trailing rows/cols stay untouched
    FramePool p; p.c = c; p.nThreads = nThreads < 1 ? 1 : nThreads; p.frames = frames; p.tw = tw;
    p.first = 0; p.last = tiles;
    if (c->tileCount >= 0) { p.first = c->tileFirst; p.last = std::min(tiles, c->tileFirst + c->tileCount); }
    p.sums.assign(tiles, 0.0f);
    p.frame_begin();
    if (p.nThreads == 1) p.worker();
    else { std::vector<std::thread> th_; for (int t = 0; t < p.nThreads; t++) th_.emplace_back([&p] { p.worker(); }); for (auto& t : th_) t.join(); }
    return 0;
}
int orc_tick(orc_ctx* c, int nThreads) { return orc_render(c, 1, nThreads); }
const float* orc_accumulator(orc_ctx* c) { return c->acc.data(); }
const uint32_t* orc_screen(orc_ctx* c) { return c->screen.data(); }
float orc_energy(orc_ctx* c) { return c->energy; }
int orc_get_counters(orc_ctx* c, orc_counters* o)
{
    o->rays = c->counters.rays; o->primary = c->counters.primary; o->interior_iters = c->counters.interior; o->leaf_iters = c->counters.leaf;
    o->tri_tests = c->counters.tri; o->tlas_iters = c->counters.tlas; o->blas_visits = c->counters.visits; o->mesh_hits = c->counters.meshhits;
    return 0;
}
int orc_reset_counters(orc_ctx* c) { c->counters = Counters(); return 0; }
int orc_tile_seed_after_frame(orc_ctx* c, int spp, int tile, uint32_t* out)
{
    if (spp != c->tileSeedSpp || tile < 0 || tile >= (int)c->tileSeedOut.size()) return -1;
    *out = c->tileSeedOut[tile]; return 0;
}

int orc_find_nearest(orc_ctx* c, const orc_ray_in* rays, orc_hit* hits, size_t n)
{
    if (!c->built) { c->err = "scene not built"; return -1; }
    Counters cn;
    for (size_t i = 0; i < n; i++) {
        Ray r = make_ray(ld3(rays[i].O), ld3(rays[i].D));
        r.inside = rays[i].inside != 0;
        c->find_nearest(r, cn);
        hits[i].t = r.t; hits[i].u = r.bu; hits[i].v = r.bv; hits[i].objIdx = r.objIdx; hits[i].triIdx = r.triIdx;
        hits[i].traversed = r.traversed; hits[i].tested = r.tested;
    }
    c->counters.add(cn);
    return 0;
}
int orc_sample(orc_ctx* c, const orc_ray_in* ray, uint32_t* seed, float rgb[3])
{
    if (!c->built) { c->err = "scene not built"; return -1; }
    Counters cn;
    Ray r = make_ray(ld3(ray->O), ld3(ray->D)); r.inside = ray->inside != 0;
    uint s = *seed;
    V3 col = c->sample(r, s, 0, cn);
    *seed = s; st3(rgb, col);
    c->counters.add(cn);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Whitted integrator (2. WhittedStyle/renderer.cpp:21-126, 131-157; file_scene.cpp:156-162, 177-187)
// ------------------------------------------------------------------------------------------------
namespace {
struct Whitted {
    orc_ctx* c; Counters cn;
    V3 light_pos() const
    {
        V3 c1 = transform_position(v3(-0.5f, 0, -0.5f), c->light.T), c2 = transform_position(v3(0.5f, 0, 0.5f), c->light.T);
        return (c1 + c2) * 0.5f - v3(0, 0.01f, 0);
    }
    bool occluded(const Ray& ray)
    {
        if (c->light.occluded(ray)) return true;
        Ray sh = ray; sh.t = 1e34f;
        cn.rays++;
        c->accel_intersect(sh, cn);
        return sh.objIdx > -1;
    }
    V3 direct(V3 I, V3 N)
    {
        V3 L = light_pos() - I;
        float dist = sqrtf(dot(L, L));
        L = L * (1 / dist);
        float ndotl = dot(N, L);
        if (ndotl < kEPS) return v3(0);
        Ray s = make_ray(I + L * kEPS, L); s.t = dist - 2 * kEPS;
        V3 irr = v3(0);
        if (!occluded(s)) {
            float att = 1 / (dist * dist);
            V3 inr = v3(24, 24, 22) * att;
            irr = inr * dot(N, L);
        }
        return irr;
    }
    V3 trace(Ray& ray, int depth)
    {
        if (depth > c->depthLimit) return v3(0);
        c->find_nearest(ray, cn);
        if (ray.objIdx == -1) return c->sky_color(ray);
        V3 I = ray.O + ray.t * ray.D;
        HitInfo hi = c->hit_info(ray, I);
        V3 N = hi.N, albedo = hi.mat->get_albedo(hi.uv);
        if (hi.mat->isLight) return v3(24, 24, 22);
        V3 out = v3(0);
        float refl = hi.mat->reflectivity, refr = hi.mat->refractivity;
        float diffuseness = 1 - (refl + refr);
        if (refl > 0.0f) {
            V3 R = reflect(ray.D, N);
            Ray r = make_ray(I + R * kEPS, R);
            out = out + refl * albedo * trace(r, depth + 1);
        } else if (refr > 0.0f) {
            V3 R = reflect(ray.D, N);
            Ray r = make_ray(I + R * kEPS, R);
            float n1 = ray.inside ? 1.2f : 1, n2 = ray.inside ? 1 : 1.2f;
            float eta = n1 / n2, cosi = dot(-ray.D, N);
            float cost2 = 1.0f - eta * eta * (1 - cosi * cosi);
            float Fr = 1;
            if (cost2 > 0) {
                float a = n1 - n2, b = n1 + n2, R0 = (a * a) / (b * b), cc = 1 - cosi;
                Fr = R0 + (1 - R0) * (cc * cc * cc * cc * cc);
                V3 T = eta * ray.D + ((eta * cosi - sqrtf(fabsf(cost2))) * N);
                Ray t = make_ray(I + T * kEPS, T);
                t.inside = !ray.inside;
                out = out + albedo * (1 - Fr) * trace(t, depth + 1);
            }
            out = out + albedo * Fr * trace(r, depth + 1);
        }
        if (diffuseness > 0) {
            V3 irr = direct(I, N);
            V3 ambient = v3(0.3f, 0.3f, 0.3f);
            V3 brdf = albedo * kINVPI;
            out = out + diffuseness * brdf * (irr + ambient);
        }
        V3 medium = v3(1);
        if (ray.inside) {
            V3 ab = hi.mat->absorption;
            medium = v3(det_expf(ab.x * -ray.t), det_expf(ab.y * -ray.t), det_expf(ab.z * -ray.t));
        }
        return medium * out;
    }
};
} // namespace

int orc_whitted_render(orc_ctx* c, int nThreads)
{
    if (!c->built || c->W == 0) { c->err = "scene not built or renderer not initialised"; return -1; }
    if (nThreads < 1) nThreads = 1;
    std::atomic<int> next(0);
    std::vector<Counters> cns(nThreads);
    auto worker = [&](int tid) {
        Whitted w; w.c = c;
        for (;;) {
            int y = next.fetch_add(1);
            if (y >= c->H) break;
            for (int x = 0; x < c->W; x++) {
                Ray pr = c->primary_ray((float)x, (float)y);
                w.cn.primary++;
                V3 col = w.trace(pr, 0);
                size_t i = (size_t)x + (size_t)y * c->W;
                c->screen[i] = orc_ctx::rgb8(col.x, col.y, col.z);
                c->acc[4 * i] = col.x; c->acc[4 * i + 1] = col.y; c->acc[4 * i + 2] = col.z; c->acc[4 * i + 3] = 0;
            }
        }
        cns[tid] = w.cn;
    };
    if (nThreads == 1) worker(0);
    else { std::vector<std::thread> th; for (int t = 0; t < nThreads; t++) th.emplace_back(worker, t); for (auto& t : th) t.join(); }
    for (auto& cn : cns) c->counters.add(cn);
    return 0;
}

// PNG scanline un-filtering for the oracle-side image reader (oracle/orc.py inflates with Python's zlib);
// raw = h rows of (1 filter byte + stride bytes), fb = bytes per complete pixel.  Independent of the product's decoder.
int orc_png_unfilter(const uint8_t* raw, uint8_t* out, int stride, int h, int fb)
{
    for (int y = 0; y < h; y++) {
        const uint8_t ft = raw[(size_t)(stride + 1) * y]; const uint8_t* in = raw + (size_t)(stride + 1) * y + 1;
        uint8_t* cur = out + (size_t)stride * y; const uint8_t* up = y ? out + (size_t)stride * (y - 1) : nullptr;
        for (int x = 0; x < stride; x++) {
            int a = x >= fb ? cur[x - fb] : 0, b = up ? up[x] : 0, c = (up && x >= fb) ? up[x - fb] : 0, pr;
            if (ft == 0) pr = 0; else if (ft == 1) pr = a; else if (ft == 2) pr = b; else if (ft == 3) pr = (a + b) >> 1;
            else if (ft == 4) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else return -1;
            cur[x] = (uint8_t)(in[x] + pr);
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Alternative accelerators of FileScene: KDTree (infra/kdtree.cpp — the shipped default, file_scene.h:10-12) and Grid (infra/grid.cpp).
// Restated on flat arrays; the tree is numbered in pre-order (node, left subtree, right subtree).  Pinned to the real reference through
// oracle/_ref (ref_kd_* / ref_grid_*) and tests/golden/ref_alt_accel.npz.
// ------------------------------------------------------------------------------------------------
namespace {
// Möller–Trumbore of kdtree.cpp:122-141 / grid.cpp:63-82 (the same arithmetic as bvh.cpp:203-222)
static inline void alt_hit_tri(Ray& r, const Tri& tri, uint ti)
{
    V3 v0 = ld3(tri.vertex0);
    V3 e1 = ld3(tri.vertex1) - v0, e2 = ld3(tri.vertex2) - v0;
    V3 h = cross(r.D, e2);
    float a = dot(e1, h);
    if (a > -0.0001f && a < 0.0001f) return;
    float f = 1 / a;
    V3 s = r.O - v0;
    float u = f * dot(s, h);
    if (u < 0 || u > 1) return;
    V3 q = cross(s, e1);
    float v = f * dot(r.D, q);
    if (v < 0 || u + v > 1) return;
    float t = f * dot(e2, q);
    if (t > 0.0001f) { if (t < r.t) { r.t = smin_(r.t, t); r.objIdx = tri.objIdx; r.triIdx = (int)ti; r.bu = u; r.bv = v; } }
}
static inline bool alt_hit_aabb(const Ray& r, V3 lo, V3 hi, float& tminOut, float& tmaxOut)   // kdtree.cpp:109-120, grid.cpp:52-61
{
    float tx1 = (lo.x - r.O.x) * r.rD.x, tx2 = (hi.x - r.O.x) * r.rD.x;
    float tmin = smin_(tx1, tx2), tmax = smax_(tx1, tx2);
    float ty1 = (lo.y - r.O.y) * r.rD.y, ty2 = (hi.y - r.O.y) * r.rD.y;
    tmin = smax_(tmin, smin_(ty1, ty2)); tmax = smin_(tmax, smax_(ty1, ty2));
    float tz1 = (lo.z - r.O.z) * r.rD.z, tz2 = (hi.z - r.O.z) * r.rD.z;
    tmin = smax_(tmin, smin_(tz1, tz2)); tmax = smin_(tmax, smax_(tz1, tz2));
    tminOut = tmin; tmaxOut = tmax;
    return tmax >= tmin && tmin < r.t && tmax > 0;
}
static inline void setcomp(V3& v, int a, float x) { if (a == 0) v.x = x; else if (a == 1) v.y = x; else v.z = x; }

struct KdTree {
    std::vector<Tri> tris; std::vector<Box> triBounds;
    std::vector<orc_kd_node> nodes; std::vector<uint32_t> refs;
    uint32_t maxDepth = 0, nodesUsed = 1;
    // kdtree.cpp:45-107: spatial median of the longest axis, stop at depth 20 or <= 2 triangles; straddling triangles go to both sides
    uint32_t subdivide(V3 lo, V3 hi, std::vector<uint>& idx, int depth)
    {
        const uint32_t me = (uint32_t)nodes.size();
        nodes.push_back(orc_kd_node());
        { orc_kd_node& n = nodes[me]; st3(n.aabbMin, lo); st3(n.aabbMax, hi); n.left = n.right = -1; n.splitAxis = 0; n.splitDistance = 0; n.firstTri = (uint32_t)refs.size(); n.triCount = 0; }
        const uint triCount = (uint)idx.size();
        bool leaf = depth >= 20 || triCount <= 2;
        if (!leaf) {
            if ((uint32_t)depth > maxDepth) maxDepth = (uint32_t)depth;
            V3 extent = hi - lo;
            int axis = 0;
            if (extent.y > extent.x) axis = 1;
            if (extent.z > comp(extent, axis)) axis = 2;
            float distance = comp(extent, axis) * 0.5f;
            float splitPos = comp(lo, axis) + distance;
            std::vector<uint> L, R;
            for (uint i = 0; i < triCount; i++) {
                uint t = idx[i];
                if (comp(triBounds[t].hi, axis) < splitPos) L.push_back(t);
                else if ((double)comp(triBounds[t].lo, axis) > (double)splitPos - 0.001) R.push_back(t);     // `splitPos - 0.001` is a double expression
                else { L.push_back(t); R.push_back(t); }
            }
            nodesUsed += 2;
            V3 lhi = hi, rlo = lo; setcomp(lhi, axis, splitPos); setcomp(rlo, axis, splitPos);
            idx.clear(); idx.shrink_to_fit();
            const uint32_t l = subdivide(lo, lhi, L, depth + 1), r = subdivide(rlo, hi, R, depth + 1);
            orc_kd_node& n = nodes[me]; n.left = (int32_t)l; n.right = (int32_t)r; n.splitAxis = axis; n.splitDistance = distance;
        } else {
            orc_kd_node& n = nodes[me]; n.firstTri = (uint32_t)refs.size(); n.triCount = triCount;
            refs.insert(refs.end(), idx.begin(), idx.end());
        }
        return me;
    }
    void build()   // kdtree.cpp:4-43
    {
        triBounds.resize(tris.size());
        Box b;
        for (size_t i = 0; i < tris.size(); i++) { Box tb; tb.grow(ld3(tris[i].vertex0)); tb.grow(ld3(tris[i].vertex1)); tb.grow(ld3(tris[i].vertex2)); b.grow(tb); triBounds[i] = tb; }
        std::vector<uint> all(tris.size()); for (size_t i = 0; i < tris.size(); i++) all[i] = (uint)i;
        subdivide(b.lo, b.hi, all, 0);
    }
    void intersect(Ray& r, int ni) const   // kdtree.cpp:143-202
    {
        const orc_kd_node& n = nodes[ni];
        float tmin, tmax;
        r.traversed++;
        if (!alt_hit_aabb(r, ld3(n.aabbMin), ld3(n.aabbMax), tmin, tmax)) return;
        if (n.left < 0) { for (uint i = 0; i < n.triCount; i++) { uint ti = refs[n.firstTri + i]; alt_hit_tri(r, tris[ti], ti); r.tested++; } return; }
        const int axis = n.splitAxis;
        const float splitPos = n.aabbMin[axis] + n.splitDistance;
        const float t = (splitPos - comp(r.O, axis)) / comp(r.D, axis);
        const int first = comp(r.D, axis) > 0 ? n.left : n.right, second = comp(r.D, axis) > 0 ? n.right : n.left;
        if ((double)t < (double)tmin + 0.001) intersect(r, second);              // the plane lies before the box: only the far side
        else if ((double)t > (double)tmax - 0.001) intersect(r, first);          // ... behind the box: only the near side
        else { intersect(r, first); if (r.t < t) return; intersect(r, second); }
    }
};

struct UGrid {
    std::vector<Tri> tris; int res[3] = {0, 0, 0}; V3 cell = v3(0.0f); Box bounds;
    std::vector<uint32_t> cellStart; std::vector<int32_t> refs;
    static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }     // tmplmath clamp(int)
    void build()   // grid.cpp:4-50
    {
        for (const Tri& t : tris) { Box tb; tb.grow(ld3(t.vertex0)); tb.grow(ld3(t.vertex1)); tb.grow(ld3(t.vertex2)); bounds.grow(tb); }
        V3 size = bounds.hi - bounds.lo;
        float cubeRoot = powf(5 * (int)tris.size() / (size.x * size.y * size.z), 1 / 3.f);
        for (int i = 0; i < 3; i++) { int r = (int)floorf(comp(size, i) * cubeRoot); res[i] = std::max(1, std::min(r, 128)); }
        cell = v3(size.x / res[0], size.y / res[1], size.z / res[2]);
        std::vector<std::vector<int32_t>> cells((size_t)res[0] * res[1] * res[2]);
        for (size_t ti = 0; ti < tris.size(); ti++) {
            Box b; b.grow(ld3(tris[ti].vertex0)); b.grow(ld3(tris[ti].vertex1)); b.grow(ld3(tris[ti].vertex2));
            int mn[3], mx[3];
            for (int k = 0; k < 3; k++) {
                mn[k] = clampi((int)((comp(b.lo, k) - comp(bounds.lo, k)) / comp(cell, k)), 0, res[k] - 1);
                mx[k] = clampi((int)((comp(b.hi, k) - comp(bounds.lo, k)) / comp(cell, k)), 0, res[k] - 1);
            }
            for (int iz = mn[2]; iz <= mx[2]; ++iz) for (int iy = mn[1]; iy <= mx[1]; ++iy) for (int ix = mn[0]; ix <= mx[0]; ++ix)
                cells[(size_t)ix + (size_t)iy * res[0] + (size_t)iz * res[0] * res[1]].push_back((int32_t)ti);
        }
        cellStart.clear(); refs.clear();
        for (auto& c : cells) { cellStart.push_back((uint32_t)refs.size()); refs.insert(refs.end(), c.begin(), c.end()); }
        cellStart.push_back((uint32_t)refs.size());
    }
    void intersect(Ray& r) const   // grid.cpp:89-153 (3D-DDA; mailboxing is compiled out in the reference)
    {
        float tmn, tmx;
        if (!alt_hit_aabb(r, bounds.lo, bounds.hi, tmn, tmx)) return;
        int exitc[3], step[3], c[3]; float deltaT[3], next[3];
        for (int i = 0; i < 3; ++i) {
            float rayOrigCell = comp(r.O, i) - comp(bounds.lo, i);
            c[i] = clampi((int)floorf(rayOrigCell / comp(cell, i)), 0, res[i] - 1);
            if (comp(r.D, i) < 0) { deltaT[i] = -comp(cell, i) * comp(r.rD, i); next[i] = (c[i] * comp(cell, i) - rayOrigCell) * comp(r.rD, i); exitc[i] = -1; step[i] = -1; }
            else { deltaT[i] = comp(cell, i) * comp(r.rD, i); next[i] = ((c[i] + 1) * comp(cell, i) - rayOrigCell) * comp(r.rD, i); exitc[i] = res[i]; step[i] = 1; }
        }
        for (;;) {
            r.traversed++;
            const uint32_t index = (uint32_t)c[0] + (uint32_t)c[1] * res[0] + (uint32_t)c[2] * res[0] * res[1];
            for (uint32_t k = cellStart[index]; k < cellStart[index + 1]; k++) { r.tested++; alt_hit_tri(r, tris[refs[k]], (uint)refs[k]); }
            const uint k = ((next[0] < next[1]) << 2) + ((next[0] < next[2]) << 1) + ((next[1] < next[2]));
            static const uint8_t map[8] = {2, 1, 2, 1, 2, 2, 0, 0};
            const uint8_t axis = map[k];
            if (r.t < next[axis]) break;
            c[axis] += step[axis];
            if (c[axis] == exitc[axis]) break;
            next[axis] += deltaT[axis];
        }
    }
};
} // namespace

int orc_set_render_accel(orc_ctx* c, int kind, void* h)      // kind 1: h = orc_kd_build(..), 2: orc_grid_build(..) over the scene's triangle array (orc_bvh_copy order); 0: the BVH again
{
    if (!c || kind < 0 || kind > 2 || (kind != 0 && (!h || c->kind != 0))) return -1;
    c->accel = kind; c->accelH = kind ? h : nullptr;
    c->accelFn = kind == 1 ? +[](void* p, Ray& r) { ((KdTree*)p)->intersect(r, 0); } : +[](void* p, Ray& r) { ((UGrid*)p)->intersect(r); };
    return 0;
}
void* orc_kd_build(const orc_tri* tris, uint32_t n) { KdTree* k = new KdTree(); k->tris.assign(tris, tris + n); k->build(); return k; }
void orc_kd_info(void* h, uint32_t* nodes, uint32_t* refs, uint32_t* maxDepth, uint32_t* nodesUsed) { KdTree* k = (KdTree*)h; *nodes = (uint32_t)k->nodes.size(); *refs = (uint32_t)k->refs.size(); *maxDepth = k->maxDepth; *nodesUsed = k->nodesUsed; }
void orc_kd_dump(void* h, orc_kd_node* nodes, uint32_t* refs) { KdTree* k = (KdTree*)h; memcpy(nodes, k->nodes.data(), k->nodes.size() * sizeof(orc_kd_node)); memcpy(refs, k->refs.data(), k->refs.size() * 4); }
void orc_kd_intersect(void* h, const float* O, const float* D, uint32_t n, orc_hit* out)
{
    KdTree* k = (KdTree*)h;
    for (uint32_t i = 0; i < n; i++) {
        Ray r = make_ray(ld3(O + 3 * i), ld3(D + 3 * i));
        k->intersect(r, 0);
        out[i].t = r.t; out[i].u = r.bu; out[i].v = r.bv; out[i].objIdx = r.objIdx; out[i].triIdx = r.triIdx; out[i].traversed = r.traversed; out[i].tested = r.tested;
    }
}
void orc_kd_free(void* h) { delete (KdTree*)h; }
void* orc_grid_build(const orc_tri* tris, uint32_t n) { UGrid* g = new UGrid(); g->tris.assign(tris, tris + n); g->build(); return g; }
void orc_grid_info(void* h, int32_t res[3], float cell[3], float lo[3], float hi[3], uint32_t* refs)
{
    UGrid* g = (UGrid*)h; for (int k = 0; k < 3; k++) res[k] = g->res[k]; st3(cell, g->cell); st3(lo, g->bounds.lo); st3(hi, g->bounds.hi); *refs = (uint32_t)g->refs.size();
}
void orc_grid_dump(void* h, uint32_t* cellStart, int32_t* refs) { UGrid* g = (UGrid*)h; memcpy(cellStart, g->cellStart.data(), g->cellStart.size() * 4); memcpy(refs, g->refs.data(), g->refs.size() * 4); }
void orc_grid_intersect(void* h, const float* O, const float* D, uint32_t n, orc_hit* out)
{
    UGrid* g = (UGrid*)h;
    for (uint32_t i = 0; i < n; i++) {
        Ray r = make_ray(ld3(O + 3 * i), ld3(D + 3 * i));
        g->intersect(r);
        out[i].t = r.t; out[i].u = r.bu; out[i].v = r.bv; out[i].objIdx = r.objIdx; out[i].triIdx = r.triIdx; out[i].traversed = r.traversed; out[i].tested = r.tested;
    }
}
void orc_grid_free(void* h) { delete (UGrid*)h; }

// probes of the restated tmplmath.h / helper.h pieces, same layout as oracle/ref_build/ref_harness.cpp's ref_math_probe / ref_vertex_dedup
void orc_math_probe(const float* in, uint32_t n, float* out)
{
    for (uint32_t i = 0; i < n; i++, in += 12, out += 120) {
        const V3 a = v3(in[0], in[1], in[2]), b = v3(in[3], in[4], in[5]), ang = v3(in[6], in[7], in[8]), sc = v3(in[9], in[10], in[11]);
        float* o = out;
        st3(o, normalize(a)); st3(o + 3, reflect(a, b)); st3(o + 6, cross(a, b)); o[9] = dot(a, b); o += 10;
        const M4 mt = m4_translate(a), rx = m4_rotx(ang.x), ry = m4_roty(ang.y), rz = m4_rotz(ang.z), ms = m4_scale(sc);
        memcpy(o, mt.c, 64); memcpy(o + 16, rx.c, 64); memcpy(o + 32, ry.c, 64); memcpy(o + 48, rz.c, 64); memcpy(o + 64, ms.c, 64); o += 80;
        M4 m = ry; m.c[3] = a.x; m.c[7] = a.y; m.c[11] = a.z;
        const M4 inv = m4_fast_inverted_noscale(m);
        memcpy(o, inv.c, 64); o += 16;
        Box bb; bb.grow(a); bb.grow(b); bb.grow(sc);
        st3(o, bb.lo); st3(o + 3, bb.hi); o[6] = bb.area(); o += 7;
        Box b1, b2; b1.grow(a); b1.grow(b); b2.grow(ang); b2.grow(sc); b1.grow(b2);
        st3(o, b1.lo); st3(o + 3, b1.hi); o[6] = b1.area();
    }
}
uint32_t orc_vertex_dedup(const float* v8, uint32_t n, uint32_t* idx, float* unique8)
{
    ObjDesc o; o.nCorners = (int)n;
    for (uint32_t i = 0; i < n; i++) { o.pos.insert(o.pos.end(), v8 + 8 * i, v8 + 8 * i + 3); o.nrm.insert(o.nrm.end(), v8 + 8 * i + 3, v8 + 8 * i + 6); o.uv.insert(o.uv.end(), v8 + 8 * i + 6, v8 + 8 * i + 8); }
    std::vector<float> P, Nn, U; std::vector<uint32_t> ix;
    dedup(o, P, Nn, U, ix);
    memcpy(idx, ix.data(), ix.size() * 4);
    for (size_t k = 0; k < P.size() / 3; k++) { memcpy(unique8 + 8 * k, &P[3 * k], 12); memcpy(unique8 + 8 * k + 3, &Nn[3 * k], 12); memcpy(unique8 + 8 * k + 6, &U[2 * k], 8); }
    return (uint32_t)(P.size() / 3);
}

float orc_expf(float x) { return det_expf(x); }
float orc_atan2f(float y, float x) { return det_atan2f(y, x); }
float orc_acosf(float x) { return det_acosf(x); }
uint32_t orc_init_seed(uint32_t b) { return init_seed(b); }
uint32_t orc_random_uint(uint32_t* s) { uint x = *s; uint r = random_uint(x); *s = x; return r; }

} // extern "C"
