// alt_accel.hip — scene.FindNearest over FileScene's alternative acceleration structures for a buffer of rays:
//   find_nearest_alt_kernel<1>   KDTree::Intersect   (infra/kdtree.cpp:143-207; the accelerator FileScene ships enabled, infra/scene/file_scene.h:10-12)
//   find_nearest_alt_kernel<2>   Grid::Intersect     (infra/grid.cpp:89-161, 3D-DDA)
// Both run FileScene::FindNearest's order (file_scene.cpp:170-175): light quad, floor plane, then the accelerator, and report Ray::traversed / Ray::tested as the
// reference counts them.  Same interface as find_nearest_kernel.  The render path through these structures (crt_set_render_accel) is render_narrow.hip's full-wave
// mode over alt_common.h's sequential kd_intersect / grid_intersect; the kernels here are the same traversals, step for step, in PERSISTENT-WAVE form (round 3):
//   * a ray is not tied to a lane for the launch: a wavefront draws rays from a launch-wide cursor, and a lane whose ray is finished takes the next one as soon as
//     a quarter of the wavefront is idle, so lanes pay for their own ray's length, not for the longest ray among 64 (before: 10.5 % / 7.8 % of the lanes
//     busy in an average VALU instruction, profiles/r02b_other_kernels.json);
//   * a trip of the wave's loop runs each KIND of step once for the lanes that are at it — box / cell step, ONE triangle test, the return to the caller frame / the
//     DDA advance — instead of nesting the triangle loop of one lane's leaf inside the step of all.
// Per ray nothing changes: the same nodes, cells and triangles in the same order with the same arithmetic (the result records are compared bit for bit with the real
// kdtree.cpp / grid.cpp, tests/golden/ref_alt_rays.npz).
#include "alt_common.h"

namespace crt {

struct RayIn { float O[3]; float D[3]; int32_t inside; };
struct HitOut { float t, u, v; int32_t objIdx, triIdx, traversed, tested; };

constexpr uint32_t kQueryRefill = 16u;                                   // idle lanes that trigger the next draw from the cursor

template <int ACCEL>
__global__ __launch_bounds__(64) void find_nearest_alt_kernel(const Scene sc, const AltAccelDev acc, const RayIn* __restrict__ rays, HitOut* __restrict__ hits, uint32_t n, uint32_t* __restrict__ cursor)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t* stkNode = lds + lane;                                       // KD-tree: (far child, plane distance) frames in this lane's LDS column, alt_common.h layout
    // the ray in this lane
    uint32_t mode = 0u;                                                   // 0 idle, 1 at a node / cell, 2 in a triangle list
    uint32_t idx = 0; f3 O = mk3(0, 0, 0), D = O, rD = O;
    Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
    int traversed = 0, tested = 0;
    uint32_t triK = 0, triEnd = 0;
    // KD-tree
    int32_t node = 0; uint32_t sp = 0;
    // grid (3D-DDA state, grid.cpp:95-128)
    int exitc[3] = {0, 0, 0}, step[3] = {0, 0, 0}, c[3] = {0, 0, 0}; float deltaT[3] = {0, 0, 0}, next[3] = {0, 0, 0};
    bool more = true;                                                     // wave-uniform: the cursor has rays left
    for (;;) {
        // ---------------- refill: idle lanes draw the next rays ----------------
        const uint64_t mIdle = __builtin_amdgcn_ballot_w64(mode == 0u);
        const uint32_t nIdle = (uint32_t)__popcll(mIdle);
        if (more && (nIdle >= kQueryRefill)) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(cursor, nIdle);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            more = base + nIdle < n;
            const uint32_t my = base + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mIdle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mIdle, 0u));
            if (mode == 0u && my < n) {
                idx = my;
                const RayIn r = rays[idx];
                O = mk3(r.O[0], r.O[1], r.O[2]); D = mk3(r.D[0], r.D[1], r.D[2]);
                rD = mk3(1 / D.x, 1 / D.y, 1 / D.z);                      // Ray ctor, template/ray.h:15-24
                h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1; traversed = 0; tested = 0;
                hit_light_floor(sc, O, D, h);
                mode = 1u;
                if (ACCEL == 1) { node = 0; sp = 0; }
                else {
                    // Grid::Intersect up to the loop, grid.cpp:89-128
                    float tmn, tmx;
                    if (!alt_box(acc.lo, acc.hi, O, rD, h.t, tmn, tmx)) mode = 3u;          // misses the grid: finished
                    else {
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const float rayOrigCell = comp(O, a) - acc.lo[a];
                            c[a] = clampi((int)__builtin_floorf(rayOrigCell / acc.cell[a]), 0, acc.res[a] - 1);
                            if (comp(D, a) < 0) { deltaT[a] = -acc.cell[a] * comp(rD, a); next[a] = ((float)c[a] * acc.cell[a] - rayOrigCell) * comp(rD, a); exitc[a] = -1; step[a] = -1; }
                            else { deltaT[a] = acc.cell[a] * comp(rD, a); next[a] = ((float)(c[a] + 1) * acc.cell[a] - rayOrigCell) * comp(rD, a); exitc[a] = acc.res[a]; step[a] = 1; }
                        }
                    }
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(mode != 0u) == 0ull) break;       // nothing in flight and nothing left to draw (`more` is false: else the refill above ran)
        bool leave = false;                                               // this lane's step is over: KD-tree -> return to the caller frames, grid -> advance the DDA
        if (ACCEL == 1) {
            // ---------------- IntersectKDTree(ray, node), kdtree.cpp:143-202: one node per trip ----------------
            if (mode == 1u) {
                traversed++;
                const KdNode nd = acc.kdNodes[node];
                float tmin, tmax;
                leave = true;
                if (alt_box(nd.lo, nd.hi, O, rD, h.t, tmin, tmax)) {
                    if (nd.left < 0) {
                        if (nd.triCount) { triK = nd.firstTri; triEnd = nd.firstTri + nd.triCount; mode = 2u; leave = false; }
                    } else {
                        const int axis = nd.splitAxis;
                        const float splitPos = nd.lo[axis] + nd.splitDistance;
                        const float t = (splitPos - comp(O, axis)) / comp(D, axis);
                        const bool pos = comp(D, axis) > 0;
                        const int32_t first = pos ? nd.left : nd.right, second = pos ? nd.right : nd.left;
                        if ((double)t < (double)tmin + 0.001) node = second;                    // the plane lies before the box: only the far side
                        else if ((double)t > (double)tmax - 0.001) node = first;               // ... behind it: only the near side
                        else { stkNode[sp * 128u] = (uint32_t)second; stkNode[sp * 128u + 64u] = asu(t); sp++; node = first; }
                        leave = false;
                    }
                }
            } else if (mode == 2u) {
                // ---------------- one triangle of the leaf, kdtree.cpp:152-158 ----------------
                alt_tri(acc.tris, acc.kdRefs[triK], O, D, h); tested++;
                triK++;
                if (triK == triEnd) { leave = true; mode = 1u; }
            }
            if (leave) {
                // return to the caller frames: `IntersectKDTree(first); if (ray.t < t) return; IntersectKDTree(second);`
                bool found = false;
                while (sp > 0) {
                    sp--;
                    const float t = asf(stkNode[sp * 128u + 64u]);
                    if (h.t < t) continue;
                    node = (int32_t)stkNode[sp * 128u]; found = true; break;
                }
                if (!found) mode = 3u;
            }
        } else {
            // ---------------- one cell of the 3D-DDA, grid.cpp:129-152 ----------------
            if (mode == 1u) {
                traversed++;
                const uint32_t index = (uint32_t)c[0] + (uint32_t)c[1] * (uint32_t)acc.res[0] + (uint32_t)c[2] * (uint32_t)acc.res[0] * (uint32_t)acc.res[1];
                triK = acc.cellStart[index]; triEnd = acc.cellStart[index + 1];
                if (triK < triEnd) mode = 2u; else leave = true;
            } else if (mode == 2u) {
                tested++; alt_tri(acc.tris, (uint32_t)acc.cellRefs[triK], O, D, h);
                triK++;
                if (triK == triEnd) { leave = true; mode = 1u; }
            }
            if (leave) {
                const uint32_t k = ((uint32_t)(next[0] < next[1]) << 2) + ((uint32_t)(next[0] < next[2]) << 1) + (uint32_t)(next[1] < next[2]);
                const int axis = (0x00221212u >> (4u * k)) & 0xfu;             // map[8] = {2, 1, 2, 1, 2, 2, 0, 0}, grid.cpp:141
                const float nx = axis == 0 ? next[0] : (axis == 1 ? next[1] : next[2]);
                if (h.t < nx) mode = 3u;
                else {
                    bool out = false;
                    if (axis == 0) { c[0] += step[0]; out = c[0] == exitc[0]; next[0] += deltaT[0]; }
                    else if (axis == 1) { c[1] += step[1]; out = c[1] == exitc[1]; next[1] += deltaT[1]; }
                    else { c[2] += step[2]; out = c[2] == exitc[2]; next[2] += deltaT[2]; }
                    if (out) mode = 3u;
                }
            }
        }
        if (mode == 3u) {                                                  // finished: the result record, and the lane is free
            HitOut o; o.t = h.t; o.u = h.u; o.v = h.v; o.objIdx = h.objIdx; o.triIdx = h.triIdx; o.traversed = traversed; o.tested = tested;
            hits[idx] = o;
            mode = 0u;
        }
    }
}

} // namespace crt

// wavefronts of a persistent query launch: enough to fill the device several times over (they hide each other's fetch latency), never more than the rays need
static uint32_t query_waves(uint32_t n, uint32_t ldsBytes)
{
    uint32_t perCu = ldsBytes ? (160u * 1024u) / ldsBytes : 16u; if (perCu > 16u) perCu = 16u; if (perCu < 4u) perCu = 4u;      // 4 wavefronts per SIMD, LDS stacks permitting (measured: 8 per SIMD is no faster for the grid and 17 % slower for the BVH)
    const uint32_t need = (n + 63u) / 64u, fill = 256u * perCu; return need < fill ? need : fill;
}

extern "C" hipError_t crt_launch_find_nearest_alt(int kind, const crt::Scene* sc, const crt::AltAccelDev* acc, const void* rays, void* hits, uint32_t n, uint32_t* cursor, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (!cursor) return hipErrorInvalidValue;
    if (hipMemsetAsync(cursor, 0, 4, stream) != hipSuccess) return hipGetLastError();
    const uint32_t ldsBytes = kind == 1 ? acc->kdStack * 128u * 4u : 0u;
    dim3 grid(query_waves(n, ldsBytes)), block(64);
    if (kind == 1) hipLaunchKernelGGL(crt::find_nearest_alt_kernel<1>, grid, block, ldsBytes, stream, *sc, *acc, (const crt::RayIn*)rays, (crt::HitOut*)hits, n, cursor);
    else hipLaunchKernelGGL(crt::find_nearest_alt_kernel<2>, grid, block, 0, stream, *sc, *acc, (const crt::RayIn*)rays, (crt::HitOut*)hits, n, cursor);
    return hipGetLastError();
}
