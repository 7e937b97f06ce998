// dev_common.h — device-side helpers shared by the kernels of kernels.hip and render_pool.hip: fp32 vector helpers, the deterministic
// exp / atan2 / acos, the RNG, the slab / Möller–Trumbore / quad / plane tests, the reference-order sequential traversal, texture lookup.
// Everything is `__device__ __forceinline__` (or a plain __device__ function defined once per translation unit in an anonymous sense:
// the two .hip files are separate translation units without relocatable device code, so each gets its own copy).
//
// Numerics: compiled with -ffp-contract=off; only IEEE + - * / sqrt, so results are bit-identical with a scalar
// CPU evaluation of the same expressions.  min/max follow the reference's std::min/std::max operand order.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "layout.h"

namespace crt {

// ------------------------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
__device__ __forceinline__ f3 operator*(float b, f3 a) { return mk3(b * a.x, b * a.y, b * a.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// 1.0f / x in three instructions instead of the eleven of the IEEE sequence (v_div_scale .. v_div_fixup): v_rcp_f32 (1 ulp) followed by one Newton
// step r + r * (1 - x * r), two fma, IS the correctly rounded quotient for every 2^-126 <= |x| <= 2^126 — compared with 1.0f / x over all
// 4 227 858 434 such inputs on gfx950 (tools/microbench/exact_rcp.hip: 0 differences) — so results stay bit-identical.  If any active lane of the
// wave holds another input (0, denormal, > 2^126, inf, NaN) the wave takes the IEEE division instead (a scalar branch: never if-converted).
// (-DCRT_IEEE_DIV builds the IEEE sequence everywhere: the A/B switch of tools/ab_bench.py)
#ifdef CRT_IEEE_DIV
__device__ __forceinline__ float rcp_nr(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float rcp_nr(float x) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
#endif
__device__ __forceinline__ float rcp_exact(float x)
{
    float r = rcp_nr(x);
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ax >= 0x1p-126f && ax <= 0x1p126f)) != 0, 0)) r = 1.0f / x;
    return r;
}
// the same where a caller only uses the result if |x| is far above 2^-126 (hit_tri: |det| >= 1e-4): one comparison
__device__ __forceinline__ float rcp_exact_large(float x)
{
    float r = rcp_nr(x);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(x) <= 0x1p126f)) != 0, 0)) r = 1.0f / x;
    return r;
}
__device__ __forceinline__ f3 rcp_exact3(f3 v)
{
    f3 r = mk3(rcp_nr(v.x), rcp_nr(v.y), rcp_nr(v.z));
    const float ax = __builtin_fabsf(v.x), ay = __builtin_fabsf(v.y), az = __builtin_fabsf(v.z);
    const bool okay = ax >= 0x1p-126f && ay >= 0x1p-126f && az >= 0x1p-126f && ax <= 0x1p126f && ay <= 0x1p126f && az <= 0x1p126f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!okay) != 0, 0)) r = mk3(1.0f / v.x, 1.0f / v.y, 1.0f / v.z);
    return r;
}
__device__ __forceinline__ f3 normalize3(f3 v) { float inv = rcp_exact(__builtin_sqrtf(dot3(v, v))); return v * inv; }
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }   // std::min(a,b)
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }   // std::max(a,b)
__device__ __forceinline__ float min_tm(float a, float b) { return a < b ? a : b; }       // tmplmath fminf
__device__ __forceinline__ float max_tm(float a, float b) { return a > b ? a : b; }       // tmplmath fmaxf
__device__ __forceinline__ float clamp_tm(float f, float a, float b) { return max_tm(a, min_tm(f, b)); }
__device__ __forceinline__ int clampi(int f, int a, int b) { int m = (b < f) ? b : f; return (a < m) ? m : a; }
__device__ __forceinline__ float asf(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t asu(float f) { return __float_as_uint(f); }

#define CRT_PI 3.14159265358979323846264f
#define CRT_INVPI 0.31830988618379067153777f
#define CRT_INV2PI 0.15915494309189533576888f
#define CRT_EPS 0.001f

// deterministic exp / atan2 / acos: same formulas, same operation order as the checker's restatement (DESIGN.md "numerics")
static __device__ float crt_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return asf(0x7f800000u);
    if (x < -103.972084045410f) return 0.0f;
    float fk = __builtin_floorf(x * 1.44269504088896341f + 0.5f);
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
    int k1 = k / 2, k2 = k - k1;
    float s1 = asf((uint32_t)(k1 + 127) << 23), s2 = asf((uint32_t)(k2 + 127) << 23);
    return p * s1 * s2;
}
// The three range reductions of atan share ONE division site and the two of acos ONE square root: -(1/x) == (-1)/x,
// x/1 == x and 1+x == 1-|x| (x < 0) are exact identities of IEEE arithmetic, so the values are those of the branchy form.
__device__ __forceinline__ float crt_atan_pos(float x)
{
    const bool big = x > 2.414213562373095f, mid = x > 0.4142135623730950f;
    const float y0 = big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.0f);
    const float num = big ? -1.0f : (mid ? x - 1.0f : x);
    const float den = big ? x : (mid ? x + 1.0f : 1.0f);
    const float t = num / den;
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    p = p * z * t + t;
    return y0 + p;
}
__device__ __forceinline__ float crt_atan2f(float y, float x)
{
    const uint32_t sy = asu(y) & 0x80000000u, sx = asu(x) & 0x80000000u;
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float inf = asf(0x7f800000u);
    const float a = crt_atan_pos(ay / ax);
    float r = sx ? (CRT_PI - a) : a;
    if (ax == inf && ay == inf) r = sx ? 2.356194490192345f : 0.7853981633974483f;
    if (ax == 0.0f) r = 1.5707963267948966f;
    if (ay == 0.0f) r = sx ? CRT_PI : 0.0f;
    r = asf(asu(r) | sy);
    return (x != x || y != y) ? x + y : r;
}
__device__ __forceinline__ float crt_asin_small(float x)
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
__device__ __forceinline__ float crt_acosf(float x)
{
    const bool hi = x > 0.5f, lo = x < -0.5f;
    const float s = __builtin_sqrtf(0.5f * (1.0f - __builtin_fabsf(x)));
    const float p = crt_asin_small((hi || lo) ? s : x);
    float r = hi ? 2.0f * p : (lo ? CRT_PI - 2.0f * p : 1.5707963267948966f - p);
    if (x > 1.0f || x < -1.0f) r = asf(0x7fc00000u);
    return (x != x) ? x : r;
}

// RNG: WangHash seed + xorshift32 (template/tmplmath.cpp:5-16, 27-34)
__device__ __forceinline__ uint32_t wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16); s *= 9u; s = s ^ (s >> 4); s *= 0x27d4eb2du; s = s ^ (s >> 15); return s;
}
__device__ __forceinline__ uint32_t init_seed(uint32_t base) { return wang_hash((base + 1u) * 17u); }
__device__ __forceinline__ float rnd(uint32_t& s)
{
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return (float)s * 2.3283064365387e-10f;
}
// RandomFloat(seed) * 2 - 1 (diffusereflection, tmplmath.h:540) in one rounding: 2.3283064365387e-10f is exactly 2^-32 as a float, so the
// scaling of the converted integer and the doubling are both exact and the only rounding is the final subtraction — fma((float)s, 2^-31, -1)
// is that same single rounding of the same real number
__device__ __forceinline__ float rnd_pm1(uint32_t& s)
{
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return __builtin_fmaf((float)s, 4.656612873077392578125e-10f, -1.0f);
}


struct Hit { float t, u, v; int objIdx, triIdx; };
struct Cnt { uint32_t rays, primary, interior, leaf, tri, tlas, visits, meshhits; };

typedef float rec4 __attribute__((ext_vector_type(4)));     // a fetched 16-byte piece of a record (native vector: usable as an asm operand)
__device__ __forceinline__ rec4 ld4(const void* p) { return *reinterpret_cast<const rec4*>(p); }
// record fetch: scalar base + 32-bit per-lane byte offset (global_load_dwordx4 v, v_off, s[base:base+1])
__device__ __forceinline__ rec4 ldg(const char* __restrict__ base, uint32_t byteOff) { return *reinterpret_cast<const rec4*>(base + byteOff); }
__device__ __forceinline__ bool finite3(f3 v)
{
    const uint32_t m = 0x7f800000u;
    return ((asu(v.x) & m) != m) && ((asu(v.y) & m) != m) && ((asu(v.z) & m) != m);
}

// slab test, infra/bvh.cpp:181-190, with the reference's std::min / std::max operand order (NaN-exact)
__device__ __forceinline__ float box_exact(rec4 lo, rec4 hi, f3 O, f3 rD, float tray)
{
    float tx1 = (lo.x - O.x) * rD.x, tx2 = (hi.x - O.x) * rD.x;
    float tmin = min_std(tx1, tx2), tmax = max_std(tx1, tx2);
    float ty1 = (lo.y - O.y) * rD.y, ty2 = (hi.y - O.y) * rD.y;
    tmin = max_std(tmin, min_std(ty1, ty2)); tmax = min_std(tmax, max_std(ty1, ty2));
    float tz1 = (lo.z - O.z) * rD.z, tz2 = (hi.z - O.z) * rD.z;
    tmin = max_std(tmin, min_std(tz1, tz2)); tmax = min_std(tmax, max_std(tz1, tz2));
    return (tmax >= tmin && tmin < tray && tmax > 0) ? tmin : 1e30f;
}
// same test with v_min/v_max(3): identical decisions whenever no product is NaN, i.e. whenever all three rD are
// finite (0 * inf is the only NaN source); the sign of a zero result never reaches a comparison that can tell.
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float box_fast(rec4 lo, rec4 hi, f3 O, f3 rD, float tray)
{
    float tx1 = (lo.x - O.x) * rD.x, tx2 = (hi.x - O.x) * rD.x;
    float ty1 = (lo.y - O.y) * rD.y, ty2 = (hi.y - O.y) * rD.y;
    float tz1 = (lo.z - O.z) * rD.z, tz2 = (hi.z - O.z) * rD.z;
    float tmin = vmax3(vmin(tx1, tx2), vmin(ty1, ty2), vmin(tz1, tz2));
    float tmax = vmin3(vmax(tx1, tx2), vmax(ty1, ty2), vmax(tz1, tz2));
    return (tmax >= tmin && tmin < tray && tmax > 0) ? tmin : 1e30f;
}

// Möller–Trumbore on a fetched LeafTri {a = v0|shadeIdx, b = e1|objIdx, c = e2|remain}, infra/bvh.cpp:203-222
// (strict '<' keeps the first of equal hits); every early-out of the reference is folded into one predicate
__device__ __forceinline__ void hit_tri(rec4 a, rec4 b, rec4 c, f3 O, f3 D, Hit& h)
{
    const f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
    const f3 hh = cross3(D, e2);
    const float det = dot3(e1, hh);
    const float f = rcp_exact_large(det);                               // 1 / det; only used when |det| >= 1e-4 (the `ok` predicate)
    const f3 s = O - v0;
    const float u = f * dot3(s, hh);
    const f3 q = cross3(s, e1);
    const float v = f * dot3(D, q);
    const float t = f * dot3(e2, q);
    const bool ok = !(det > -0.0001f && det < 0.0001f) && !(u < 0 || u > 1) && !(v < 0 || u + v > 1) && (t > 0.0001f) && (t < h.t);
    if (ok) { h.t = t; h.u = u; h.v = v; h.triIdx = (int)asu(a.w); h.objIdx = (int)asu(b.w); }
}

// Quad::Intersect + Plane::Intersect (template/primitives.h:331-346, 107-111): the two analytic primitives FindNearest
// tests before the acceleration structure (file_scene.cpp:170-175)
// (the operands as values: render_pool_kernel reads them from the kernel-argument segment inside its passes instead of holding them in scalar registers)
struct LightFloor { float lightInvT[12]; float lightSize; uint32_t lightAxis, floorAxisY; float floorN[3]; float floorD; };
__device__ __forceinline__ void hit_light_floor(const LightFloor& sc, f3 O, f3 D, Hit& h)
{
    // Scene::lightAxis / floorAxisY (set at upload): the quad's invT has an identity rotation block / the plane's normal is exactly
    // (0,1,0) — what FileScene and TLASFileScene always build (file_scene.cpp:15-19).  Then 1*x == x and the 0*x terms only add
    // zeros, so the general expressions reduce to the short ones below; the two can differ in the SIGN OF A ZERO only, which no
    // comparison here can see and which never reaches an accepted t (accepted hits have a non-zero numerator and denominator).
    {
        const float* c = sc.lightInvT;
        float Oy, Dy;
        if (sc.lightAxis) { Oy = O.y + c[7]; Dy = D.y; }
        else { Oy = c[4] * O.x + c[5] * O.y + c[6] * O.z + c[7]; Dy = c[4] * D.x + c[5] * D.y + c[6] * D.z; }
        const float t = Oy / -Dy;
        if (t < h.t && t > 0) {
            float Ox, Oz, Dx, Dz;
            if (sc.lightAxis) { Ox = O.x + c[3]; Oz = O.z + c[11]; Dx = D.x; Dz = D.z; }
            else {
                Ox = c[0] * O.x + c[1] * O.y + c[2] * O.z + c[3];
                Oz = c[8] * O.x + c[9] * O.y + c[10] * O.z + c[11];
                Dx = c[0] * D.x + c[1] * D.y + c[2] * D.z;
                Dz = c[8] * D.x + c[9] * D.y + c[10] * D.z;
            }
            const float Ix = Ox + t * Dx, Iz = Oz + t * Dz;
            const float size = sc.lightSize;
            if (Ix > -size && Ix < size && Iz > -size && Iz < size) { h.t = t; h.objIdx = 0; }
        }
    }
    {
        float num, den;
        if (sc.floorAxisY) { num = O.y + sc.floorD; den = D.y; }
        else { const f3 N = mk3(sc.floorN[0], sc.floorN[1], sc.floorN[2]); num = dot3(O, N) + sc.floorD; den = dot3(D, N); }
        const float t = -num / den;
        if (t < h.t && t > 0) { h.t = t; h.objIdx = 1; }
    }
}

__device__ __forceinline__ void hit_light_floor(const Scene& sc, f3 O, f3 D, Hit& h)
{
    LightFloor lf;
#pragma unroll
    for (int i = 0; i < 12; i++) lf.lightInvT[i] = sc.lightInvT[i];
    lf.lightSize = sc.lightSize; lf.lightAxis = sc.lightAxis; lf.floorAxisY = sc.floorAxisY;
    lf.floorN[0] = sc.floorN[0]; lf.floorN[1] = sc.floorN[1]; lf.floorN[2] = sc.floorN[2]; lf.floorD = sc.floorD;
    hit_light_floor(lf, O, D, h);
}

// Scene fields that only the shading passes need (the root's child pair, the camera, the light quad / floor plane: 50 dwords) are NOT held in scalar registers
// across the render kernels' loops — with them render_pool_kernel spills ~50 SGPRs into VGPR lanes and pays v_readlane / v_writelane around every pass.  The passes re-read them from the
// kernel-argument segment instead (the Scene block is the kernel's first argument: offset 0), through a pointer the optimiser cannot hoist: a few s_load_dwordx8/x16
// per pass from the scalar cache.
static_assert(offsetof(Scene, topLeft) == offsetof(Scene, camPos) + 12 && offsetof(Scene, bottomLeft) == offsetof(Scene, camPos) + 36 && offsetof(Scene, invW) == offsetof(Scene, camPos) + 48 && offsetof(Scene, invH) == offsetof(Scene, camPos) + 52, "camera block");
static_assert(offsetof(Scene, lightSize) == offsetof(Scene, lightInvT) + 60 && offsetof(Scene, floorN) == offsetof(Scene, lightInvT) + 76 && offsetof(Scene, floorD) == offsetof(Scene, lightInvT) + 88 && offsetof(Scene, floorInvto) == offsetof(Scene, floorN) + 16, "light / floor block");
static_assert(offsetof(Scene, floorAxisY) == offsetof(Scene, lightAxis) + 4 && offsetof(Scene, skyW) == offsetof(Scene, skyOffset) + 4 && offsetof(Scene, skyH) == offsetof(Scene, skyOffset) + 8, "flag / sky words");
static_assert(offsetof(Material, absorption) == 8 && offsetof(Material, texOffset) == 20 && offsetof(Material, texW) == 24 && offsetof(Material, texH) == 28, "Material words");
typedef const float __attribute__((address_space(4)))* kernarg_f;
__device__ __forceinline__ kernarg_f scene_floats(size_t byteOffset)
{
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + byteOffset;
    asm volatile("" : "+s"(p));
    return (kernarg_f)p;
}

// BLASBVH::Intersect's ray transform (infra/blas_bvh.cpp:376-381): invT rows, SSE summation order (x+y)+(z+w) / (x+y)+z
__device__ __forceinline__ void to_object_space(rec4 r0, rec4 r1, rec4 r2, f3 O, f3 D, f3& Oo, f3& Do, f3& rDo)
{
    Oo = mk3((O.x * r0.x + O.y * r0.y) + (O.z * r0.z + 1.0f * r0.w),
             (O.x * r1.x + O.y * r1.y) + (O.z * r1.z + 1.0f * r1.w),
             (O.x * r2.x + O.y * r2.y) + (O.z * r2.z + 1.0f * r2.w));
    Do = mk3((D.x * r0.x + D.y * r0.y) + D.z * r0.z,
             (D.x * r1.x + D.y * r1.y) + D.z * r1.z,
             (D.x * r2.x + D.y * r2.y) + D.z * r2.z);
    rDo = rcp_exact3(Do);
}

// ------------------------------------------------------------------------------------------------------------
// sequential reference-order traversal (used by find_nearest_kernel: it reports Ray::traversed / tested, which
// count loop trips in the reference's order — infra/bvh.cpp:224-258, infra/tlas_bvh.cpp:83-111)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void traverse_bvh_seq(const Scene& sc, uint32_t rootRef, f3 O, f3 D, f3 rD, Hit& h, uint32_t* stk, Cnt& cn,
                                                 int& traversed, int& tested)
{
    const char* __restrict__ g = sc.geom;
    uint32_t cur = rootRef, sp = 0;
    for (;;) {
        traversed++;
        const uint32_t off = (cur & kRefOffsetMask) << 4;
        if (cur & kRefInterior) {
            cn.interior++;
            const rec4 alo = ldg(g, off), ahi = ldg(g, off + 16), blo = ldg(g, off + 32), bhi = ldg(g, off + 48);
            float d1 = box_exact(alo, ahi, O, rD, h.t), d2 = box_exact(blo, bhi, O, rD, h.t);
            uint32_t r1 = asu(alo.w), r2 = asu(blo.w);
            if (d1 > d2) { float td = d1; d1 = d2; d2 = td; uint32_t tr = r1; r1 = r2; r2 = tr; }
            if (d1 == 1e30f) { if (sp == 0) break; cur = stk[(--sp) * 64]; }
            else { cur = r1; if (d2 != 1e30f) { stk[sp * 64] = r2; sp++; } }
        } else {
            cn.leaf++;
            uint32_t o = off;
            for (;;) {
                const rec4 a = ldg(g, o), b = ldg(g, o + 16), c = ldg(g, o + 32);
                tested++; cn.tri++;
                hit_tri(a, b, c, O, D, h);
                if (asu(c.w) <= 1u) break;
                o += 48;
            }
            if (sp == 0) break;
            cur = stk[(--sp) * 64];
        }
    }
}

__device__ __forceinline__ void find_nearest_seq(const Scene& sc, f3 O, f3 D, f3 rD, Hit& h, uint32_t* stk, Cnt& cn, int& traversed, int& tested)
{
    const char* __restrict__ g = sc.geom;
    cn.rays++;
    hit_light_floor(sc, O, D, h);
    if (sc.kind == 0) {
        traverse_bvh_seq(sc, sc.rootRef, O, D, rD, h, stk, cn, traversed, tested);
    } else {
        uint32_t* tstk = stk + sc.bvhStack * 64;     // TLAS entries live above the BVH part of this lane's column
        uint32_t cur = sc.rootRef, sp = 0;
        for (;;) {
            traversed++; cn.tlas++;
            if ((cur & kRefTlasLeaf) == kRefTlasLeaf) {
                cn.visits++;
                const uint32_t io = sc.instOff + (cur & 0xffffu) * 128u;
                const rec4 r0 = ldg(g, io), r1 = ldg(g, io + 16), r2 = ldg(g, io + 32), ids = ldg(g, io + 48);
                f3 Oo, Do, rDo; to_object_space(r0, r1, r2, O, D, Oo, Do, rDo);
                traverse_bvh_seq(sc, asu(ids.z), Oo, Do, rDo, h, stk, cn, traversed, tested);
                if (sp == 0) break;
                cur = tstk[(--sp) * 64];
            } else {
                const uint32_t o1 = sc.tlasOff + (cur & 0x7fffu) * 32u, o2 = sc.tlasOff + ((cur >> 15) & 0x7fffu) * 32u;
                const rec4 alo = ldg(g, o1), ahi = ldg(g, o1 + 16), blo = ldg(g, o2), bhi = ldg(g, o2 + 16);
                float d1 = box_exact(alo, ahi, O, rD, h.t), d2 = box_exact(blo, bhi, O, rD, h.t);
                uint32_t r1 = asu(alo.w), r2 = asu(blo.w);
                if (d1 > d2) { float td = d1; d1 = d2; d2 = td; uint32_t tr = r1; r1 = r2; r2 = tr; }
                if (d1 == 1e30f) { if (sp == 0) break; cur = tstk[(--sp) * 64]; }
                else { cur = r1; if (d2 != 1e30f) { tstk[sp * 64] = r2; sp++; } }
            }
        }
    }
    if (h.objIdx >= 2) cn.meshhits++;
}

// Texture::Sample, template/texture.h:61-96, in its two halves: nearest-texel index in the pooled texel array, and 0x00RRGGBB -> float3
__device__ __forceinline__ uint32_t tex_index(uint32_t offset, int w, int hgt, float u, float v)
{
    u = clamp_tm(u, 0.0f, 1.0f);
    v = 1 - clamp_tm(v, 0.0f, 1.0f);
    int x = (int)(u * w), y = (int)(v * hgt);
    x = clampi(x, 0, w - 1); y = clampi(y, 0, hgt - 1);
    return offset + (uint32_t)x + (uint32_t)y * (uint32_t)w;
}
__device__ __forceinline__ f3 tex_unpack(uint32_t p)
{
    const float s = 1 / 255.0f;
    return mk3(((p >> 16) & 0xFF) * s, ((p >> 8) & 0xFF) * s, (p & 0xFF) * s);
}
__device__ __forceinline__ f3 tex_sample(const Scene& sc, uint32_t offset, int w, int hgt, float u, float v)
{
    return tex_unpack(sc.texels[tex_index(offset, w, hgt, u, v)]);
}

// GetSkyColor, infra/scene/file_scene.cpp:142-154
__device__ __forceinline__ f3 sky_color(const Scene& sc, f3 D)
{
    float phi = crt_atan2f(-D.z, D.x) + CRT_PI;
    float theta = crt_acosf(-D.y);
    return tex_sample(sc, sc.skyOffset, sc.skyW, sc.skyH, phi * CRT_INV2PI, theta * CRT_INVPI);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

} // namespace crt
