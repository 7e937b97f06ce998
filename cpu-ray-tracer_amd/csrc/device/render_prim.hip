// render_prim.hip — PrimitiveScene (second half: infra/scene/primitive_scene.cpp + template/primitives.h Sphere :31, Cube :187, Quad :321, Torus :380)
// behind the same two seams as the triangle scenes: scene.FindNearest for a ray buffer (find_nearest_prim_kernel) and Renderer::Sample per tile
// (render_prim_kernel: the sequential form — one wavefront per (tile, 64-frame window), lane = frame, a plain per-lane path loop).
// The scene is the reference's hard-coded demo room: six walls, the swinging light quad, the bouncing mirror ball, the "rounded corners" sphere, the spinning
// glass cube and the glass torus, in the SPEEDTRIX / single-light configuration its headers select.  No acceleration structure: every ray tests all eleven.
//
// PARITY UNPINNED.  Neither reference file compiles with this image's compilers (MSVC-only __m128 member access) and the reference holds no fixture of this scene; the
// kernels are checked bit for bit against the repo's CPU restatement of the same classes (tests/test_gpu_primitive_scene.py).  The torus solves its quartic in double precision with
// cos(acos(x) / 3), which the reference takes from the C runtime (unspecified to the bit): both sides use the same fdlibm-style det_acos / det_cos below (plain
// IEEE double + - * / sqrt), so oracle and kernel agree exactly; against a Windows build of the reference the torus' hit distances can differ in the last place.
//
// Numerics: -ffp-contract=off; fp32 as everywhere, fp64 only inside the torus test.  No MFMA.
#include "dev_common.h"

namespace crt {

struct PrimDev {                          // the scene at one animation time (crt_primitive_scene flattened), passed to the kernels by value
    float quadInvT[12], quadNrm[3], quadSize;
    float spherePos[3], pad0;
    float cubeInvM[12], cubeM[12], cubeMin[3], cubeMax[3];
    float torusInvT[12], torusT[12], rt2, rc2, r2, pad1;
    float refl[11], refr[11], absorb[33];
    float pad2;
    const uint32_t* red; const uint32_t* blue;     // 512 x 512 texels 0x00RRGGBB (the left / right wall's albedo override), may be null (black)
};

__device__ __forceinline__ uint32_t hi_word(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ double with_words(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }
// acos / cos in IEEE double arithmetic only (the published fdlibm algorithms e_acos.c, k_cos.c, k_sin.c, one step of e_rem_pio2.c); same code as the oracle's
static __device__ double det_acos(double x)
{
    const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
        pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
        pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
        qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const uint32_t hx = hi_word(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);
    }
    if (ix < 0x3fe00000u) {
        if (ix <= 0x3c600000u) return pio2_hi + pio2_lo;
        const double z = x * x;
        const double pp = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = pp / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx & 0x80000000u) {
        const double z = (one + x) * 0.5;
        const double pp = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double sq = __builtin_sqrt(z), r = pp / q, w = r * sq - pio2_lo;
        return pi - 2.0 * (sq + w);
    } else {
        const double z = (one - x) * 0.5, sq = __builtin_sqrt(z);
        const double df = with_words(hi_word(sq), 0u);
        const double c = (z - df * df) / (sq + df);
        const double pp = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = pp / q, w = r * sq + c;
        return 2.0 * (df + w);
    }
}
static __device__ double det_cos(double x)
{
    const double one = 1.0, C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11, S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
        pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const uint32_t ix = hi_word(x) & 0x7fffffffu;
    if (!(x >= 0.0) || ix >= 0x4002d97cu) return (x - x) / (x - x);
    if (ix <= 0x3fe921fbu) {
        if (ix < 0x3e400000u) return one;
        const double z = x * x;
        const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        if (ix < 0x3fd33333u) return one - (0.5 * z - (z * r - x * 0.0));
        const double qx = (ix > 0x3fe90000u) ? 0.28125 : with_words(ix - 0x00200000u, 0u);
        const double hz = 0.5 * z - qx, a = one - qx;
        return a - (hz - (z * r - x * 0.0));
    }
    double z = x - pio2_1, y0, y1;
    if (ix != 0x3ff921fbu) { y0 = z - pio2_1t; y1 = (z - y0) - pio2_1t; }
    else { z -= pio2_2; y0 = z - pio2_2t; y1 = (z - y0) - pio2_2t; }
    const double zz = y0 * y0, v = zz * y0;
    const double r = S2 + zz * (S3 + zz * (S4 + zz * (S5 + zz * S6)));
    const double sn = y0 - ((zz * (0.5 * y1 - v * r) - y1) - v * S1);
    return -sn;
}
static __device__ double cbrt_fast(double n)                  // Torus::cbrtFast, primitives.h:548-556 (float literals as written)
{
    double x1 = n / 10.0f, x2 = 1.0f; int turn = 0;
    while (__builtin_fabs(x1 - x2) > 0.00000001 && turn++ < 100) { x1 = x2; x2 = (2.0f / 3.0f * x1) + (n / (3.0f * x1 * x1)); }
    return x2;
}
__device__ __forceinline__ float compf(f3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
// TransformPosition_SSE / TransformVector_SSE (tmplmath.cpp:170-191): (x + y) + (z + w) / (x + y) + z;  TransformPosition / TransformVector (:162-169): left to right
__device__ __forceinline__ f3 xf_pos_sse(const float* m, f3 a) { return mk3((a.x * m[0] + a.y * m[1]) + (a.z * m[2] + 1.0f * m[3]), (a.x * m[4] + a.y * m[5]) + (a.z * m[6] + 1.0f * m[7]), (a.x * m[8] + a.y * m[9]) + (a.z * m[10] + 1.0f * m[11])); }
__device__ __forceinline__ f3 xf_vec_sse(const float* m, f3 a) { return mk3((a.x * m[0] + a.y * m[1]) + a.z * m[2], (a.x * m[4] + a.y * m[5]) + a.z * m[6], (a.x * m[8] + a.y * m[9]) + a.z * m[10]); }
__device__ __forceinline__ f3 xf_pos(const float* m, f3 a) { return mk3(m[0] * a.x + m[1] * a.y + m[2] * a.z + m[3] * 1.0f, m[4] * a.x + m[5] * a.y + m[6] * a.z + m[7] * 1.0f, m[8] * a.x + m[9] * a.y + m[10] * a.z + m[11] * 1.0f); }
__device__ __forceinline__ f3 xf_vec(const float* m, f3 a) { return mk3(m[0] * a.x + m[1] * a.y + m[2] * a.z + m[3] * 0.0f, m[4] * a.x + m[5] * a.y + m[6] * a.z + m[7] * 0.0f, m[8] * a.x + m[9] * a.y + m[10] * a.z + m[11] * 0.0f); }

// Torus::Intersect, primitives.h:386-453 (double precision; the float products inside the double expressions are float products, as written there)
static __device__ void torus_intersect(const PrimDev& p, f3 Ow, f3 Dw, float& rt, int& obj)
{
    const f3 O = xf_pos_sse(p.torusInvT, Ow), D = xf_vec_sse(p.torusInvT, Dw);
    const float rt2 = p.rt2, rc2 = p.rc2, r2 = p.r2;
    double po = 1, m = dot3(O, O), k3 = dot3(O, D), k32 = k3 * k3;
    const double v = k32 - m + r2;
    if (v < 0) return;
    double k = (m - rt2 - rc2) * 0.5, k2 = k32 + rc2 * D.z * D.z + k;
    double k1 = k * k3 + rc2 * O.z * D.z, k0 = k * k + rc2 * O.z * O.z - rc2 * rt2;
    if (__builtin_fabs(k3 * (k32 - k2) + k1) < 0.0001) {
        const double tmp = k1; k1 = k3; k3 = tmp;
        po = -1; k0 = 1 / k0; k1 = k1 * k0; k2 = k2 * k0; k3 = k3 * k0; k32 = k3 * k3;
    }
    double c2 = 2 * k2 - 3 * k32, c1 = k3 * (k32 - k2) + k1;
    double c0 = k3 * (k3 * (-3 * k32 + 4 * k2) - 8 * k1) + 4 * k0;
    c2 *= 0.33333333333; c1 *= 2; c0 *= 0.33333333333;
    const double Q = c2 * c2 + c0, R = 3 * c0 * c2 - c2 * c2 * c2 - c1 * c1;
    double h = R * R - Q * Q * Q, z;
    if (h < 0) { const double sQ = __builtin_sqrt(Q); z = 2 * sQ * det_cos(det_acos(R / (sQ * Q)) * 0.33333333333); }
    else { const double sQ = cbrt_fast(__builtin_sqrt(h) + __builtin_fabs(R)); z = __builtin_copysign(__builtin_fabs(sQ + Q / sQ), R); }
    z = c2 - z;
    double d1 = z - 3 * c2, d2 = z * z - 3 * c0;
    if (__builtin_fabs(d1) < 1.0e-8) { if (d2 < 0) return; d2 = __builtin_sqrt(d2); }
    else { if (d1 < 0) return; d1 = __builtin_sqrt(d1 * 0.5); d2 = c1 / d1; }
    double t = 1e20;
    h = d1 * d1 - z + d2;
    if (h > 0) {
        h = __builtin_sqrt(h);
        double t1 = -d1 - h - k3, t2 = -d1 + h - k3;
        t1 = (po < 0) ? 2 / t1 : t1; t2 = (po < 0) ? 2 / t2 : t2;
        if (t1 > 0) t = t1;
        if (t2 > 0) t = (t2 < t) ? t2 : t;
    }
    h = d1 * d1 - z - d2;
    if (h > 0) {
        h = __builtin_sqrt(h);
        double t1 = d1 - h - k3, t2 = d1 + h - k3;
        t1 = (po < 0) ? 2 / t1 : t1; t2 = (po < 0) ? 2 / t2 : t2;
        if (t1 > 0) t = (t1 < t) ? t1 : t;
        if (t2 > 0) t = (t2 < t) ? t2 : t;
    }
    const float ft = (float)t;
    if (ft > 0 && ft < rt) { rt = ft; obj = 10; }
}

// PrimitiveScene::FindNearest, primitive_scene.cpp:92-175 (SPEEDTRIX branches): walls, light quad, the two spheres, cube, torus — in that order, strict <
static __device__ void prim_find_nearest(const PrimDev& p, f3 O, f3 D, f3 rD, float& rt, int& obj)
{
    {
        const float xmin[3] = {3, 1, 3}, xmax[3] = {-2.99f, -2, -3.99f};
        float t3[3]; int id3[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const bool sel = compf(D, a) >= 0;
            const float x = sel ? xmax[a] : xmin[a];
            id3[a] = sel ? 5 + 2 * a : 4 + 2 * a;
            const float d = 0.0f - ((compf(O, a) + x) * compf(rD, a));
            t3[a] = (d <= 0.0f) ? 1e34f : d;
        }
        rt = t3[0]; obj = id3[0];
        if (t3[1] < rt) { rt = t3[1]; obj = id3[1]; }
        if (t3[2] < rt) { rt = t3[2]; obj = id3[2]; }
    }
    {   // Quad::Intersect, primitives.h:331-346
        const float* c = p.quadInvT;
        const float Oy = c[4] * O.x + c[5] * O.y + c[6] * O.z + c[7];
        const float Dy = c[4] * D.x + c[5] * D.y + c[6] * D.z;
        const float t = Oy / -Dy;
        if (t < rt && t > 0) {
            const float Ox = c[0] * O.x + c[1] * O.y + c[2] * O.z + c[3];
            const float Oz = c[8] * O.x + c[9] * O.y + c[10] * O.z + c[11];
            const float Dx = c[0] * D.x + c[1] * D.y + c[2] * D.z;
            const float Dz = c[8] * D.x + c[9] * D.y + c[10] * D.z;
            const float Ix = Ox + t * Dx, Iz = Oz + t * Dz;
            const float size = p.quadSize;
            if (Ix > -size && Ix < size && Iz > -size && Iz < size) { rt = t; obj = 0; }
        }
    }
    {
        const f3 oc = O - mk3(p.spherePos[0], p.spherePos[1], p.spherePos[2]);
        const float b = dot3(oc, D);
        const float d = b * b - (dot3(oc, oc) - 0.36f);
        if (d > 0) { const float t = -b - __builtin_sqrtf(d); if (t < rt && t > 0) { rt = t; obj = 1; } }
    }
    {
        const f3 oc = O - mk3(0, 2.5f, -3.07f);
        const float b = dot3(oc, D);
        const float d = b * b - (dot3(oc, oc) - 64.0f);
        if (d > 0) { const float t = __builtin_sqrtf(d) - b; if (t < rt && t > 0) { rt = t; obj = 2; } }
    }
    {   // Cube::Intersect, primitives.h:200-225
        const f3 o = xf_pos_sse(p.cubeInvM, O), d = xf_vec_sse(p.cubeInvM, D);
        const f3 rd = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const f3 bmin = mk3(p.cubeMin[0], p.cubeMin[1], p.cubeMin[2]), bmax = mk3(p.cubeMax[0], p.cubeMax[1], p.cubeMax[2]);
        const f3 t1 = (bmin - o) * rd, t2 = (bmax - o) * rd;
        const f3 vmaxv = mk3(max_tm(t1.x, t2.x), max_tm(t1.y, t2.y), max_tm(t1.z, t2.z)), vminv = mk3(min_tm(t1.x, t2.x), min_tm(t1.y, t2.y), min_tm(t1.z, t2.z));   // _mm_max_ps / _mm_min_ps
        const float tmax = min_std(vmaxv.x, min_std(vmaxv.y, vmaxv.z)), tmin = max_std(vminv.x, max_std(vminv.y, vminv.z));
        if (tmin < tmax) { if (tmin > 0) { if (tmin < rt) { rt = tmin; obj = 3; } } else if (tmax > 0) { if (tmax < rt) { rt = tmax; obj = 3; } } }
    }
    torus_intersect(p, O, D, rt, obj);
}

// PrimitiveScene::GetHitInfo's normal (before the flip towards the ray), primitive_scene.cpp:202-236
static __device__ f3 prim_normal(const PrimDev& p, int obj, f3 I)
{
    if (obj == 0) return mk3(p.quadNrm[0], p.quadNrm[1], p.quadNrm[2]);
    if (obj == 1) return (I - mk3(p.spherePos[0], p.spherePos[1], p.spherePos[2])) * (1 / 0.6f);
    if (obj == 2) return (I - mk3(0, 2.5f, -3.07f)) * (1 / 8.0f);
    if (obj == 3) {
        const f3 o = xf_pos(p.cubeInvM, I);
        f3 N = mk3(-1, 0, 0);
        const float d0 = __builtin_fabsf(o.x - p.cubeMin[0]), d1 = __builtin_fabsf(o.x - p.cubeMax[0]), d2 = __builtin_fabsf(o.y - p.cubeMin[1]), d3 = __builtin_fabsf(o.y - p.cubeMax[1]),
                    d4 = __builtin_fabsf(o.z - p.cubeMin[2]), d5 = __builtin_fabsf(o.z - p.cubeMax[2]);
        float minDist = d0;
        if (d1 < minDist) { minDist = d1; N.x = 1; }
        if (d2 < minDist) { minDist = d2; N = mk3(0, -1, 0); }
        if (d3 < minDist) { minDist = d3; N = mk3(0, 1, 0); }
        if (d4 < minDist) { minDist = d4; N = mk3(0, 0, -1); }
        if (d5 < minDist) { minDist = d5; N = mk3(0, 0, 1); }
        return xf_vec(p.cubeM, N);
    }
    if (obj == 10) {
        const f3 L = xf_pos(p.torusInvT, I);
        const float dd = dot3(L, L) - p.rt2;
        const f3 N = normalize3(L * (mk3(dd, dd, dd) - p.rc2 * mk3(1, 1, -1)));
        return xf_vec(p.torusT, N);
    }
    f3 N = mk3(0, 0, 0); const float s = 1 - 2 * (float)(obj & 1); const int a = (obj - 4) / 2;
    if (a == 0) N.x = s; else if (a == 1) N.y = s; else N.z = s;
    return N;
}
// Plane::GetAlbedo (primitives.h:134-172) for the three planes whose material overrides the albedo: 4 left wall (red.png), 5 right wall (blue.png), 6 floor (checkerboard)
static __device__ f3 prim_albedo_override(const PrimDev& p, int obj, f3 I)
{
    if (obj == 6) {
        int ix = (int)(I.x * 2 + 96.01f), iz = (int)(I.z * 2 + 96.01f);
        if (ix == 98 && iz == 98) { ix = (int)(I.x * 32.01f); iz = (int)(I.z * 32.01f); }
        if (ix == 94 && iz == 98) { ix = (int)(I.x * 64.01f); iz = (int)(I.z * 64.01f); }
        const float c = ((ix + iz) & 1) ? 1.0f : 0.3f;
        return mk3(c, c, c);
    }
    const uint32_t* tx = (obj == 4) ? p.red : p.blue;
    const int ix = (int)((I.z - 4) * (512.0f / 7)), iy = (int)((2 - I.y) * (512.0f / 3));
    const uint32_t px = tx ? tx[(uint32_t)(ix & 511) + (uint32_t)(iy & 511) * 512u] : 0u;
    return mk3((float)((px >> 16) & 255u), (float)((px >> 8) & 255u), (float)(px & 255u)) * (1.0f / 255.0f);
}

struct RayIn { float O[3]; float D[3]; int32_t inside; };
struct HitOut { float t, u, v; int32_t objIdx, triIdx, traversed, tested; };
__global__ __launch_bounds__(64) void find_nearest_prim_kernel(const PrimDev p, const RayIn* __restrict__ rays, HitOut* __restrict__ hits, uint32_t n)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const RayIn r = rays[i];
    const f3 O = mk3(r.O[0], r.O[1], r.O[2]), D = mk3(r.D[0], r.D[1], r.D[2]);
    const f3 rD = mk3(1 / D.x, 1 / D.y, 1 / D.z);                        // Ray ctor, template/ray.h:15-24
    float t = 1e34f; int obj = -1;
    prim_find_nearest(p, O, D, rD, t, obj);
    HitOut o; o.t = t; o.u = 0; o.v = 0; o.objIdx = obj; o.triIdx = -1; o.traversed = 0; o.tested = 0;
    hits[i] = o;
}

// Renderer::ProcessTile + Sample (renderer.cpp:50-131) over the primitive scene: block = one wavefront = one (tile, 64-frame window), lane = frame
__global__ __launch_bounds__(64) void render_prim_kernel(const Scene sc, const PrimDev p, float4* __restrict__ slab, Counters* __restrict__ counters,
                                                          uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                                          uint32_t sppFirst, uint32_t frames, uint32_t passes)
{
    __shared__ float fstAll[15 * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t windows = (frames + 63u) / 64u;
    const uint32_t tl = blockIdx.x / windows, win = blockIdx.x - tl * windows;
    if (tl >= tileCount) return;
    sppFirst += win * 64u * passes;
    frames = (frames - win * 64u < 64u) ? frames - win * 64u : 64u;
    if (lane >= frames) return;
    slab += (size_t)win * ((size_t)tileCount * 256u * 64u * passes);
    const uint32_t tile = tileFirst + tl * tileStride;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    float* fst = fstAll + lane;
    uint32_t nRays = 0, nPrimary = 0;
    const uint32_t items = 256u * passes;
    uint32_t seed = init_seed(tx + ty * (uint32_t)sc.W + (sppFirst + lane * passes) * 1799u);   // renderer.cpp:120
    const f3 camPos = mk3(sc.camPos[0], sc.camPos[1], sc.camPos[2]);
    const f3 TL = mk3(sc.topLeft[0], sc.topLeft[1], sc.topLeft[2]), TR = mk3(sc.topRight[0], sc.topRight[1], sc.topRight[2]), BL = mk3(sc.bottomLeft[0], sc.bottomLeft[1], sc.bottomLeft[2]);
    for (uint32_t item = 0; item < items; item++) {
        const uint32_t pix = (passes == 1u) ? item : item / passes;
        const int x = (int)(tx * 16u + (pix & 15u)), y = (int)(ty * 16u + (pix >> 4));
        const float jy = rnd(seed);                                               // pinned: first draw is the y jitter
        const float jx = rnd(seed);
        const float u = ((float)x + jx) * sc.invW, vv = ((float)y + jy) * sc.invH;
        const f3 P = TL + u * (TR - TL) + vv * (BL - TL);
        const f3 v = P - camPos;
        f3 O = camPos, D = v * rcp_exact(__builtin_sqrtf(dot3(v, v)));
        bool inside = false; int depth = 0;
        nPrimary++;
        f3 L = mk3(0, 0, 0);
        for (;;) {
            const f3 rD = rcp_exact3(D);
            float t = 1e34f; int obj = -1;
            nRays++;
            prim_find_nearest(p, O, D, rD, t, obj);
            if (obj == -1) { L = mk3(0, 0, 0); break; }                            // PrimitiveScene::GetSkyColor
            if (depth >= sc.depthLimit) { L = mk3(0, 0, 0); break; }
            const f3 I = O + t * D;
            f3 N = prim_normal(p, obj, I);
            if (dot3(N, D) > 0) N = -N;
            const f3 c = (obj >= 4 && obj <= 6) ? prim_albedo_override(p, obj, I) : mk3(1.0f, 1.0f, 1.0f);     // material->isAlbedoOverridden ? scene.GetAlbedo : material->albedo
            if (obj == 0) { L = mk3(24, 24, 22); break; }                          // materials[0].isLight
            f3 medium = mk3(1, 1, 1);
            if (inside) {
                const f3 ab = mk3(p.absorb[3 * obj], p.absorb[3 * obj + 1], p.absorb[3 * obj + 2]) * -t;
                medium = mk3(crt_expf(ab.x), crt_expf(ab.y), crt_expf(ab.z));
            }
            const float refl = p.refl[obj], refr = p.refr[obj];
            f3 nv, factor; bool newInside = false;
            const float r = rnd(seed);
            if (r < refl) {
                nv = D - 2.0f * N * dot3(N, D);
                factor = c * medium;
            } else if (r < refl + refr) {
                nv = D - 2.0f * N * dot3(N, D);
                const float n1 = inside ? 1.2f : 1, n2 = inside ? 1 : 1.2f;
                const float eta = n1 / n2, cosi = dot3(-D, N);
                const float cost2 = 1.0f - eta * eta * (1 - cosi * cosi);
                if (cost2 > 0) {
                    const float a = n1 - n2, b2 = n1 + n2, R0 = (a * a) / (b2 * b2), cc = 1 - cosi;
                    const float Fr = R0 + (1 - R0) * (cc * cc * cc * cc * cc);
                    const f3 T = eta * D + ((eta * cosi - __builtin_sqrtf(__builtin_fabsf(cost2))) * N);
                    if (rnd(seed) > Fr) { nv = T; newInside = !inside; }
                }
                factor = c * medium;
            } else {
                f3 Rr;
                do {
                    const float rz = rnd_pm1(seed);                                // draw order pinned z, y, x
                    const float ry = rnd_pm1(seed);
                    const float rx = rnd_pm1(seed);
                    Rr = mk3(rx, ry, rz);
                } while (dot3(Rr, Rr) > 1);
                if (dot3(Rr, N) < 0) Rr = Rr * -1.0f;
                nv = Rr * rcp_exact(__builtin_sqrtf(dot3(Rr, Rr)));
                const f3 brdf = c * CRT_INVPI;
                const f3 pre = medium * brdf * 2.0f * CRT_PI;
                factor = pre * dot3(nv, N);
            }
            float* fd = fst + (uint32_t)(3 * depth) * 64u;
            fd[0] = factor.x; fd[64] = factor.y; fd[128] = factor.z;
            depth++;
            O = I + nv * CRT_EPS; D = nv; inside = newInside;
        }
#pragma unroll
        for (int k = 4; k >= 0; k--)
            if (depth > k) { const float* fd = fst + (uint32_t)(3 * k) * 64u; L = mk3(fd[0], fd[64], fd[128]) * L; }
        uint32_t pass = 0;
        if (passes != 1u) pass = item - pix * passes;
        slab[((size_t)tl * 256u + pix) * (64u * passes) + (lane * passes + pass)] = make_float4(L.x, L.y, L.z, 0.0f);
    }
    atomicAdd(&counters->v[0], (unsigned long long)nRays);
    atomicAdd(&counters->v[1], (unsigned long long)nPrimary);
}

} // namespace crt

extern "C" hipError_t crt_launch_find_nearest_prim(const crt::PrimDev* p, const void* rays, void* hits, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(crt::find_nearest_prim_kernel, dim3((n + 63u) / 64u), dim3(64), 0, stream, *p, (const crt::RayIn*)rays, (crt::HitOut*)hits, n);
    return hipGetLastError();
}
extern "C" hipError_t crt_launch_render_prim(const crt::Scene* sc, const crt::PrimDev* p, void* slab, crt::Counters* counters, uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                             uint32_t sppFirst, uint32_t frames, uint32_t passes, hipStream_t stream)
{
    if (tileCount == 0 || frames == 0) return hipSuccess;
    const uint32_t windows = (frames + 63u) / 64u;
    if ((unsigned long long)tileCount * windows > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(crt::render_prim_kernel, dim3(tileCount * windows), dim3(64), 0, stream, *sc, *p, (float4*)slab, counters, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes);
    return hipGetLastError();
}
