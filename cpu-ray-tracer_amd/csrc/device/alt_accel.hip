// alt_accel.hip — scene.FindNearest over FileScene's alternative acceleration structures, one ray per lane:
//   find_nearest_kd_kernel     KDTree::Intersect   (infra/kdtree.cpp:143-207; the accelerator FileScene ships enabled, infra/scene/file_scene.h:10-12)
//   find_nearest_grid_kernel   Grid::Intersect     (infra/grid.cpp:89-161, 3D-DDA)
// Both run FileScene::FindNearest's order (file_scene.cpp:170-175): light quad, floor plane, then the accelerator (alt_common.h), and report Ray::traversed /
// Ray::tested as the reference counts them.  Query kernels (latency-bound pointer chasing), same interface as find_nearest_kernel.  The render path through these
// structures (crt_set_render_accel) is render_narrow.hip's full-wave mode.
#include "alt_common.h"

namespace crt {

struct RayIn { float O[3]; float D[3]; int32_t inside; };
struct HitOut { float t, u, v; int32_t objIdx, triIdx, traversed, tested; };

template <int ACCEL>
__global__ __launch_bounds__(64) void find_nearest_alt_kernel(const Scene sc, const AltAccelDev acc, const RayIn* __restrict__ rays, HitOut* __restrict__ hits, uint32_t n)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x, i = blockIdx.x * 64u + lane;
    if (i >= n) return;
    const RayIn r = rays[i];
    const f3 O = mk3(r.O[0], r.O[1], r.O[2]), D = mk3(r.D[0], r.D[1], r.D[2]);
    const f3 rD = mk3(1 / D.x, 1 / D.y, 1 / D.z);                        // Ray ctor, template/ray.h:15-24
    Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
    int traversed = 0, tested = 0;
    hit_light_floor(sc, O, D, h);
    if (ACCEL == 1) kd_intersect(acc, O, D, rD, h, lds + lane, traversed, tested);
    else grid_intersect(acc, O, D, rD, h, traversed, tested);
    HitOut o; o.t = h.t; o.u = h.u; o.v = h.v; o.objIdx = h.objIdx; o.triIdx = h.triIdx; o.traversed = traversed; o.tested = tested;
    hits[i] = o;
}

} // namespace crt

extern "C" hipError_t crt_launch_find_nearest_alt(int kind, const crt::Scene* sc, const crt::AltAccelDev* acc, const void* rays, void* hits, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    dim3 grid((n + 63u) / 64u), block(64);
    if (kind == 1) hipLaunchKernelGGL(crt::find_nearest_alt_kernel<1>, grid, block, acc->kdStack * 128u * 4u, stream, *sc, *acc, (const crt::RayIn*)rays, (crt::HitOut*)hits, n);
    else hipLaunchKernelGGL(crt::find_nearest_alt_kernel<2>, grid, block, 0, stream, *sc, *acc, (const crt::RayIn*)rays, (crt::HitOut*)hits, n);
    return hipGetLastError();
}
