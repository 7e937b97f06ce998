// alt_common.h — device-side traversal of FileScene's alternative acceleration structures, shared by the query kernels (alt_accel.hip), the render kernel that traces
// Renderer::Sample through them (render_narrow.hip, MODE 2 / 3) and the Whitted kernel (kernels.hip):
//   kd_intersect     KDTree::Intersect   (infra/kdtree.cpp:143-207; the accelerator FileScene ships enabled, infra/scene/file_scene.h:10-12)
//   grid_intersect   Grid::Intersect     (infra/grid.cpp:89-161, 3D-DDA)
// Both report Ray::traversed / Ray::tested as the reference counts them.  Numerics as everywhere: -ffp-contract=off, IEEE + - * / only, std::min / std::max operand
// order; the two `double` comparisons of the KD traversal (`t < tmin + 0.001` mixes a float with a double literal) are evaluated in double like the reference.
#pragma once
#include "dev_common.h"

namespace crt {

struct KdNode { float lo[3]; int32_t left; float hi[3]; int32_t right; float splitDistance; int32_t splitAxis; uint32_t firstTri, triCount; };   // = crt_kd_node, 48 B; left < 0: leaf
struct AltTri { float v0[3]; uint32_t triIdx; float e1[3]; int32_t objIdx; float e2[3]; uint32_t pad; };                                          // Möller–Trumbore operands, reference triangle order
struct AltAccelDev {
    const KdNode* kdNodes; const uint32_t* kdRefs; uint32_t kdStack;         // kdStack: entries per lane (tree height + 1)
    const AltTri* tris;
    int32_t res[3]; float cell[3]; float lo[3], hi[3]; const uint32_t* cellStart; const int32_t* cellRefs;
};

// IntersectAABB of kdtree.cpp:109-120 / grid.cpp:52-61: the slab test that also hands tmin / tmax out
__device__ __forceinline__ bool alt_box(const float* lo, const float* hi, f3 O, f3 rD, float tray, float& tminOut, float& tmaxOut)
{
    float tx1 = (lo[0] - O.x) * rD.x, tx2 = (hi[0] - O.x) * rD.x;
    float tmin = min_std(tx1, tx2), tmax = max_std(tx1, tx2);
    float ty1 = (lo[1] - O.y) * rD.y, ty2 = (hi[1] - O.y) * rD.y;
    tmin = max_std(tmin, min_std(ty1, ty2)); tmax = min_std(tmax, max_std(ty1, ty2));
    float tz1 = (lo[2] - O.z) * rD.z, tz2 = (hi[2] - O.z) * rD.z;
    tmin = max_std(tmin, min_std(tz1, tz2)); tmax = min_std(tmax, max_std(tz1, tz2));
    tminOut = tmin; tmaxOut = tmax;
    return tmax >= tmin && tmin < tray && tmax > 0;
}
__device__ __forceinline__ void alt_tri(const AltTri* __restrict__ tris, uint32_t ti, f3 O, f3 D, Hit& h)
{
    const rec4* p = reinterpret_cast<const rec4*>(tris + ti);
    hit_tri(p[0], p[1], p[2], O, D, h);          // the same Möller–Trumbore as bvh.cpp:203-222 (kdtree.cpp:122-141 and grid.cpp:63-82 repeat it)
}
__device__ __forceinline__ float comp(f3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// KDTree::Intersect: IntersectKDTree(ray, root) with the recursion turned into a stack of (far child, plane distance) in this lane's LDS column:
// node index at stk[k * 128], its plane distance at stk[k * 128 + 64]; the `if (ray.t < t) return` of the caller frame is applied at pop time
__device__ __forceinline__ void kd_intersect(const AltAccelDev& acc, f3 O, f3 D, f3 rD, Hit& h, uint32_t* stkNode, int& traversed, int& tested)
{
    uint32_t sp = 0; int32_t node = 0;
    for (;;) {
        // IntersectKDTree(ray, node), kdtree.cpp:143-202
        bool descend = false;
        traversed++;
        const KdNode nd = acc.kdNodes[node];
        float tmin, tmax;
        if (alt_box(nd.lo, nd.hi, O, rD, h.t, tmin, tmax)) {
            if (nd.left < 0) {
                for (uint32_t k = 0; k < nd.triCount; k++) { alt_tri(acc.tris, acc.kdRefs[nd.firstTri + k], O, D, h); tested++; }
            } else {
                const int axis = nd.splitAxis;
                const float splitPos = nd.lo[axis] + nd.splitDistance;
                const float t = (splitPos - comp(O, axis)) / comp(D, axis);
                const bool pos = comp(D, axis) > 0;
                const int32_t first = pos ? nd.left : nd.right, second = pos ? nd.right : nd.left;
                if ((double)t < (double)tmin + 0.001) node = second;                    // the plane lies before the box: only the far side
                else if ((double)t > (double)tmax - 0.001) node = first;               // ... behind it: only the near side
                else { stkNode[sp * 128u] = (uint32_t)second; stkNode[sp * 128u + 64u] = asu(t); sp++; node = first; }
                descend = true;
            }
        }
        if (descend) continue;
        // return to the caller frames: `IntersectKDTree(first); if (ray.t < t) return; IntersectKDTree(second);`
        bool found = false;
        while (sp > 0) {
            sp--;
            const float t = asf(stkNode[sp * 128u + 64u]);
            if (h.t < t) continue;
            node = (int32_t)stkNode[sp * 128u]; found = true; break;
        }
        if (!found) break;
    }
}

// Grid::Intersect: IntersectGrid, grid.cpp:89-153 (3D-DDA over the cells the ray crosses)
__device__ __forceinline__ void grid_intersect(const AltAccelDev& acc, f3 O, f3 D, f3 rD, Hit& h, int& traversed, int& tested)
{
    float tmn, tmx;
    if (!alt_box(acc.lo, acc.hi, O, rD, h.t, tmn, tmx)) return;
    int exitc[3], step[3], c[3]; float deltaT[3], next[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float rayOrigCell = comp(O, a) - acc.lo[a];
        c[a] = clampi((int)__builtin_floorf(rayOrigCell / acc.cell[a]), 0, acc.res[a] - 1);
        if (comp(D, a) < 0) { deltaT[a] = -acc.cell[a] * comp(rD, a); next[a] = ((float)c[a] * acc.cell[a] - rayOrigCell) * comp(rD, a); exitc[a] = -1; step[a] = -1; }
        else { deltaT[a] = acc.cell[a] * comp(rD, a); next[a] = ((float)(c[a] + 1) * acc.cell[a] - rayOrigCell) * comp(rD, a); exitc[a] = acc.res[a]; step[a] = 1; }
    }
    for (;;) {
        traversed++;
        const uint32_t index = (uint32_t)c[0] + (uint32_t)c[1] * (uint32_t)acc.res[0] + (uint32_t)c[2] * (uint32_t)acc.res[0] * (uint32_t)acc.res[1];
        const uint32_t e = acc.cellStart[index + 1];
        for (uint32_t k = acc.cellStart[index]; k < e; k++) { tested++; alt_tri(acc.tris, (uint32_t)acc.cellRefs[k], O, D, h); }
        const uint32_t k = ((uint32_t)(next[0] < next[1]) << 2) + ((uint32_t)(next[0] < next[2]) << 1) + (uint32_t)(next[1] < next[2]);
        const int axis = (0x00221212u >> (4u * k)) & 0xfu;                 // map[8] = {2, 1, 2, 1, 2, 2, 0, 0}, grid.cpp:141
        const float nx = axis == 0 ? next[0] : (axis == 1 ? next[1] : next[2]);
        if (h.t < nx) break;
        bool out = false;
        if (axis == 0) { c[0] += step[0]; out = c[0] == exitc[0]; next[0] += deltaT[0]; }
        else if (axis == 1) { c[1] += step[1]; out = c[1] == exitc[1]; next[1] += deltaT[1]; }
        else { c[2] += step[2]; out = c[2] == exitc[2]; next[2] += deltaT[2]; }
        if (out) break;
    }
}

} // namespace crt
