// pool_common.h — what render_pool_kernel (render_pool.hip) and render_duo_kernel (render_duo.hip) share: the parked stream record and the lane-mask helpers
#pragma once
#include "dev_common.h"

namespace crt {

// parked stream state, SoA in LDS: field f of stream s at st[f * S + s]
enum : uint32_t {
    F_OX = 0, F_OY, F_OZ, F_DX, F_DY, F_DZ,                      // ray origin, direction (world space)
    F_RX, F_RY, F_RZ,                                            // reciprocal direction: needed until the walk is over (swap-in, return from a BLAS)
    F_T,                                                         // nearest hit distance: after the quad / plane tests, then FindNearest's result
    F_SEED, F_META,                                              // RNG state; item | depth << 11 | inside << 14 | fresh << 15 | (hit objIdx + 1) << 16
    F_U, F_V,                                                    // barycentrics of a mesh hit (after the walk)
    F_COUNT,                                                     // 14 dwords = 56 bytes per parked stream
    // slots with two lives:
    F_CUR = F_U, F_PEND = F_V,                                   // what a READY stream needs before the walk: node reference to start at, far root child to push (0 = none)
    F_TRI = F_RX                                                 // after the walk: the hit triangle's global shade index
};
// The path's throughput factors (albedo*medium*... of each bounce, multiplied on unwind: 15 floats, written once per bounce, read once
// per path) live in a global scratch area behind the launch's sample slab instead of LDS — 8 KB per wave that buy a third wave per SIMD:
// component j of depth k of stream s of block b at fac[(b * 15 + 3k + j) * S + s].
constexpr uint32_t kMetaItemMask = 0x7ffu, kMetaDepthShift = 11u, kMetaInside = 1u << 14, kMetaFresh = 1u << 15, kMetaObjShift = 16u, kMetaLowMask = 0xffffu;

// Lane sets of the loop are explicit 64-bit scalar masks (mRes: lanes holding a resident stream; mNode / mTri / mTlas: what each resident lane is at) and a
// per-lane predicate is `lane_in(mask)` = the mask used directly as the execution / select mask (amdgcn inverse ballot): no v_cndmask + v_cmp round trip per
// ballot, and `resident` never has to be re-derived from lane state.
__device__ __forceinline__ bool lane_in(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ uint64_t finite3_mask(f3 v)          // lanes whose three components are all finite: three compares straight into scalar masks
{
    const uint32_t m = 0x7f800000u;
    return __builtin_amdgcn_ballot_w64((asu(v.x) & m) != m) & __builtin_amdgcn_ballot_w64((asu(v.y) & m) != m) & __builtin_amdgcn_ballot_w64((asu(v.z) & m) != m);
}
__device__ __forceinline__ uint32_t rank_in(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }   // set bits below this lane

} // namespace crt
