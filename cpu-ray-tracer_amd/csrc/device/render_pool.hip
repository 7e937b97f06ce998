// render_pool.hip — render_pool_kernel: the path tracer's per-tile sample loop ("3. PathTracer/renderer.cpp":117-131, Sample :50-100,
// FindNearest infra/scene/file_scene.cpp:170-175 / tlas_file_scene.cpp:201-206, IntersectBVH infra/bvh.cpp:224-258) as a STREAM POOL.
//
// The unit of work is the reference's RNG stream: one xorshift32 stream per (tile, frame), consumed serially over the tile's 256
// pixels (renderer.cpp:120-126).  A wavefront owns S > 64 such streams of one tile (S consecutive frames) but has only 64 lanes, and
// a stream is not tied to a lane:
//   * a stream that is WALKING the acceleration structure is resident in a lane: its ray, nearest hit, node reference and
//     pre-loaded node / triangle record live in that lane's registers, its traversal stack in the lane's LDS column;
//   * a stream that WAITS for shading is parked in LDS (ray, hit, RNG state, pixel counter, the path's throughput factors —
//     14 dwords; the path's throughput factors in a global scratch area) and queued by what it needs next: the END queue (the path ended: sky lookup or light / depth limit, unwind the
//     throughput factors, store the sample, generate the next pixel's primary ray) or the BOUNCE queue (surface hit: normal,
//     uv, albedo, mirror / dielectric / rejection-sampled diffuse direction);
//   * a shading pass takes up to 64 streams from ONE queue — all lanes run the same branch of Renderer::Sample — starts each
//     stream's next FindNearest (light quad, floor plane, the root step from the kernel arguments) and queues the stream as
//     READY to walk, or, when the ray misses both root children, straight back into a shading queue;
//   * lanes whose stream finished walking take the next READY stream.
// So the traversal phases run on lanes that are (nearly) all walking and the shading passes on (nearly) full wavefronts, instead
// of every phase running on the sub-set of 64 fixed streams that happens to be in that state (render_tiles_kernel).  Per stream
// nothing changes: every ray visits the reference's nodes in the reference's order and every float expression is evaluated as
// written there, so the image is bit-identical to render_tiles_kernel's and to the CPU oracle's.
//
// Numerics: -ffp-contract=off, IEEE + - * / sqrt only (dev_common.h).  No MFMA: pointer chasing + slab / Möller–Trumbore tests.
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
constexpr uint32_t kQueueMask = 127u;                            // queues are rings of 128 one-byte stream ids (S <= 128)

#ifndef CRT_POOL_MIN_WAVES
#define CRT_POOL_MIN_WAVES 4         // waves per SIMD the register budget must allow (<= 128 VGPRs)
#endif
#ifndef CRT_POOL_SHADE_MIN
#define CRT_POOL_SHADE_MIN 64     // streams a shading pass waits for ...
#endif
#ifndef CRT_POOL_NODE_STEPS
#define CRT_POOL_NODE_STEPS 2     // NODE steps per trip of the lanes that stay at interior nodes
#endif
#ifndef CRT_POOL_NODE2_MIN
#define CRT_POOL_NODE2_MIN 16     // ... as long as at least this many lanes take the further step
#endif
#ifndef CRT_POOL_STARVE
#define CRT_POOL_STARVE 40        // ... unless fewer than this many streams are walking or ready to walk
#endif

#ifdef CRT_POOL_STAMPS
// diagnostic build only (-DCRT_POOL_STAMPS, tools/pool_stamps.py): shader-clock time of the sections of the loop, summed over all waves:
// [0] walk phases + swap out (incl. the wait for the records), [1] swap in + record loads + pass decision, [2] END pass up to its new_ray, [3] END's new_ray, [4] END tail (unwind, store),
// [5] BOUNCE up to the material draw, [6] material draw + rejection loop, [7] normalise + factor store, [8] BOUNCE's new_ray, [9] trips, [10] END passes, [11] BOUNCE passes
__device__ unsigned long long g_poolStamps[16];
#define CRT_PSTAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define CRT_PACC(i, a, b) (pst[i] += (b) - (a))
#else
#define CRT_PSTAMP(var)
#define CRT_PACC(i, a, b)
#endif
#ifdef CRT_POOL_DENS
// diagnostic build only (-DCRT_POOL_DENS, tools/pool_density.py): how often every section of the loop runs and with how many lanes, summed over all waves
__device__ unsigned long long g_poolDens[32];
#define CRT_DENS(i, v) (dens[i] += (uint32_t)(v))
#define CRT_DENS_MASK(i, pred) (dens[i] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(pred)))
#else
#define CRT_DENS(i, v)
#define CRT_DENS_MASK(i, pred)
#endif
#ifdef CRT_POOL_TIMELINE
// diagnostic build only (-DCRT_POOL_TIMELINE, tools/pool_timeline.py): start / end wall clock (100 MHz) and compute unit of every wavefront of the last launch
__device__ unsigned long long* g_poolTimeline = nullptr;
#endif
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

template <int KIND, bool COUNT, int S>
__global__ __launch_bounds__(64, CRT_POOL_MIN_WAVES) void render_pool_kernel(const Scene sc, float4* __restrict__ slab, float* __restrict__ facScratch, Counters* __restrict__ counters,
                                                             unsigned long long* __restrict__ tileClocks, const uint32_t* __restrict__ tileOrder,
                                                             uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                                             uint32_t sppFirst, uint32_t frames, uint32_t passes, uint32_t groups, uint32_t rankFirst, uint32_t* __restrict__ tileCost, unsigned long long* __restrict__ launchClk)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
#ifdef CRT_POOL_TIMELINE
    const unsigned long long tl0 = wall_clock64();
#endif
    const unsigned long long clk0 = (COUNT || tileCost) ? wall_clock64() : 0ull;
    if (launchClk && lane == 0) { atomicMax(&launchClk[0], ~clk0); atomicMax(&launchClk[1], clk0); }     // first / last wavefront start of a measuring job (abi.cpp adopt_job_costs)
    // block -> (tile rank, group of S frames), rank-major: all groups of the expensive tiles (listed first by the host) are dispatched first;
    // rankFirst > 0: the tiles before it in the order are rendered by a concurrent render_tiles_kernel launch (a split job, abi.cpp)
    const uint32_t rank0 = blockIdx.x / groups, grp = blockIdx.x - rank0 * groups, rank = rank0 + rankFirst;
    if (rank >= tileCount) return;
    const uint32_t tl = tileOrder ? tileOrder[rank] : rank;
    const uint32_t frame0 = grp * (uint32_t)S;                                   // first frame (of the launch) of this wave's streams
    const uint32_t nStreams = (frames - frame0 < (uint32_t)S) ? frames - frame0 : (uint32_t)S;
    const uint32_t tile = tileFirst + tl * tileStride;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const char* __restrict__ geom = sc.geom;
    float* __restrict__ fac = facScratch + (size_t)blockIdx.x * (15u * (uint32_t)S);

    // `cur` and the stack entries of this kernel are 16-bit references (layout.h: ref16 — the children's are stored next to the 32-bit ones in
    // every NodePair); 2 bytes per stack entry instead of 4 is what lets 128 parked streams + the stacks fit 4 waves per SIMD.
    // Traversal stack of the stream resident in a lane: entry i (1 .. stackDepth) at LDS byte  2 * lane + 128 * i;  entry 0 is a dummy below the bottom, so the
    // speculative read of the top and the dead store above it need no bounds logic, and the stack pointer IS the byte address of the top entry (`spB`).
    char* const ldsB = reinterpret_cast<char*>(lds);
    const uint32_t laneB = lane * 2u;
    uint32_t* st = lds + (sc.stackDepth + 1u) * 32u;                             // parked stream state (behind (stackDepth + 1) * 64 two-byte entries)
    float* stf = reinterpret_cast<float*>(st);
    uint8_t* qEnd = reinterpret_cast<uint8_t*>(st + F_COUNT * S);
    uint8_t* qBnc = qEnd + 128, * qRdy = qEnd + 256;
    auto stk_top = [&](uint32_t at) -> uint32_t { return *reinterpret_cast<const uint16_t*>(ldsB + at); };
    auto stk_put = [&](uint32_t at, uint32_t v) { *reinterpret_cast<uint16_t*>(ldsB + at) = (uint16_t)v; };

    Cnt cn; cn.rays = cn.primary = cn.interior = cn.leaf = cn.tri = cn.tlas = cn.visits = cn.meshhits = 0;
    uint32_t trips = 0;
#ifdef CRT_POOL_DENS
    uint32_t dens[32]; for (int i = 0; i < 32; i++) dens[i] = 0;
#endif
    const uint32_t items = 256u * passes;                                        // (pixel, pass) pairs in stream order
    const uint32_t rowLen = 64u * passes;                                        // float4 per pixel row of the slab
    const f3 nil3 = mk3(0.0f, 0.0f, 0.0f);                                       // placeholder of values no lane reads

    // every stream starts in the END queue as "fresh": the pass only generates its first primary ray
    for (uint32_t s = lane; s < nStreams; s += 64u) {
        st[F_SEED * S + s] = init_seed(tx + ty * (uint32_t)sc.W + (sppFirst + (frame0 + s) * passes) * 1799u);   // renderer.cpp:120
        st[F_META * S + s] = kMetaFresh;
        qEnd[s] = (uint8_t)s;
    }
    uint32_t endH = 0, endT = nStreams, bncH = 0, bncT = 0, rdyH = 0, rdyT = 0;  // queue heads / tails (wave-uniform)

    // The stream resident in this lane (if any: mRes).  Invariant at the top of a trip: a resident stream is walking (cur != done) — streams whose walk ended
    // leave their lane in the trip that ended it, and only streams with a walk ahead are ever READY.
    uint64_t mRes = 0ull;                                                        // lanes holding a stream
    uint64_t mFinite = ~0ull;                                                    // ... whose reciprocal direction is finite in all components (v_min / v_max slab test is exact)
    uint32_t sid = 0, cur = kRefDone, spB = laneB, metaLo = 0u;
    f3 tO = nil3, tD = nil3, trD = nil3;
    Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
    rec4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;

    // ---- the start of a stream's next scene.FindNearest, shared by both shading passes: normalise, reciprocal direction, light quad,
    // floor plane, root step; parks the new ray and queues the stream (READY, or END / BOUNCE when the ray never enters the tree)
    auto new_ray = [&](bool act, uint32_t s, f3 v, bool norm, f3 O, uint32_t seed, uint32_t meta) {
        f3 D = v, rD = v; Hit nh; nh.t = 1e34f; nh.u = 0; nh.v = 0; nh.objIdx = -1; nh.triIdx = -1;
        uint32_t ncur = kRefDone, pend = 0u;
        if (act) {
            const float inv = rcp_exact(__builtin_sqrtf(dot3(v, v)));           // normalize(): v * (1 / sqrtf(dot(v, v)))
            D = norm ? v * inv : v;
            rD = rcp_exact3(D);
            cn.rays++;
            {
                const kernarg_f lp = scene_floats(offsetof(Scene, lightInvT));      // lightInvT[12] lightNrm[3] lightSize lightPos[3] floorN[3] floorD: one block of the Scene
                const kernarg_f ax = scene_floats(offsetof(Scene, lightAxis));      // lightAxis, floorAxisY
                LightFloor lf;
#pragma unroll
                for (int i = 0; i < 12; i++) lf.lightInvT[i] = lp[i];
                lf.lightSize = lp[15]; lf.floorN[0] = lp[19]; lf.floorN[1] = lp[20]; lf.floorN[2] = lp[21]; lf.floorD = lp[22];
                lf.lightAxis = asu(ax[0]); lf.floorAxisY = asu(ax[1]);
                hit_light_floor(lf, O, D, nh);
            }
            if (sc.rootIsPair) {
                // bvh.cpp:244-257 / tlas_bvh.cpp:96-110 at the root with an empty stack, from the child pair in the kernel arguments
                const kernarg_f rp = scene_floats(offsetof(Scene, rootPair));
                const rec4 a0 = {rp[0], rp[1], rp[2], rp[3]}, a1 = {rp[4], rp[5], rp[6], rp[7]};
                const rec4 b0 = {rp[8], rp[9], rp[10], rp[11]}, b1 = {rp[12], rp[13], rp[14], rp[15]};
                float d1, d2;
                if (__builtin_amdgcn_ballot_w64(!finite3(rD)) == 0ull) { d1 = box_fast(a0, a1, O, rD, nh.t); d2 = box_fast(b0, b1, O, rD, nh.t); }
                else { d1 = box_exact(a0, a1, O, rD, nh.t); d2 = box_exact(b0, b1, O, rD, nh.t); }
                const bool sw = d1 > d2;
                const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
                const uint32_t rn = sw ? asu(b1.w) : asu(a1.w), rf = sw ? asu(a1.w) : asu(b1.w);   // the children's ref16
                const bool hitN = dn != 1e30f;
                pend = (hitN && df != 1e30f) ? rf : 0u;
                ncur = hitN ? rn : kRefDone;
                if (COUNT) { if (KIND == 0) cn.interior++; else cn.tlas++; }
            } else ncur = sc.rootRef16;
            if (COUNT && KIND == 0 && (ncur & kRef16TagMask) == 0u && ncur != kRefDone) cn.leaf++;
            stf[F_OX * S + s] = O.x; stf[F_OY * S + s] = O.y; stf[F_OZ * S + s] = O.z;
            stf[F_DX * S + s] = D.x; stf[F_DY * S + s] = D.y; stf[F_DZ * S + s] = D.z;
            stf[F_RX * S + s] = rD.x; stf[F_RY * S + s] = rD.y; stf[F_RZ * S + s] = rD.z;
            stf[F_T * S + s] = nh.t;
            st[F_SEED * S + s] = seed; st[F_META * S + s] = meta | ((uint32_t)(nh.objIdx + 1) << kMetaObjShift);
            st[F_CUR * S + s] = ncur; st[F_PEND * S + s] = pend;
        }
        // queue: the walk is needed only when the ray enters the tree; otherwise FindNearest is already over (renderer.cpp:52-55, 69).  Every set is the ballot of ONE
        // comparison combined with scalar logic.
        const uint32_t depth = (meta >> kMetaDepthShift) & 7u;
        const uint64_t mAct = __builtin_amdgcn_ballot_w64(act);
        const uint64_t mW = mAct & __builtin_amdgcn_ballot_w64(ncur != kRefDone);
        const uint64_t mStop = __builtin_amdgcn_ballot_w64((uint32_t)(nh.objIdx + 1) <= 1u) | __builtin_amdgcn_ballot_w64((int)depth >= sc.depthLimit);   // miss, light, or depth limit
        const uint64_t mE = mAct & ~mW & mStop, mB = mAct & ~mW & ~mStop;
        CRT_DENS(20, __popcll(mAct)); CRT_DENS(19, __popcll(mW));
        if (lane_in(mW)) qRdy[(rdyT + rank_in(mW)) & kQueueMask] = (uint8_t)s;
        if (lane_in(mE)) qEnd[(endT + rank_in(mE)) & kQueueMask] = (uint8_t)s;
        if (lane_in(mB)) qBnc[(bncT + rank_in(mB)) & kQueueMask] = (uint8_t)s;
        rdyT += (uint32_t)__popcll(mW); endT += (uint32_t)__popcll(mE); bncT += (uint32_t)__popcll(mB);
    };
    // back to the world-space ray when a BLAS is finished (two-level scenes): the parked copy is the world-space ray
    auto world_ray = [&]() {
        tO = mk3(stf[F_OX * S + sid], stf[F_OY * S + sid], stf[F_OZ * S + sid]);
        tD = mk3(stf[F_DX * S + sid], stf[F_DY * S + sid], stf[F_DZ * S + sid]);
        trD = mk3(stf[F_RX * S + sid], stf[F_RY * S + sid], stf[F_RZ * S + sid]);
    };

    // ---------------- NODE step (infra/bvh.cpp:244-257 / tlas_bvh.cpp:96-110) of the lanes in m, on their pre-loaded NodePair ----------------
    uint64_t mNode2 = 0ull;
    auto node_step = [&](uint64_t m) {
        const bool allFinite = (m & ~mFinite) == 0ull;
        if (lane_in(m)) {
            if (COUNT) { if (KIND == 1 && (cur & kRef16TlasBit) != 0u) cn.tlas++; else cn.interior++; }
            const uint32_t top = stk_top(spB);                               // speculative: lands during the slab arithmetic (the dummy entry when the stack is empty)
            float d1, d2;
            if (__builtin_expect(allFinite, 1)) { d1 = box_fast(q0, q1, tO, trD, h.t); d2 = box_fast(q2, q3, tO, trD, h.t); }
            else { d1 = box_exact(q0, q1, tO, trD, h.t); d2 = box_exact(q2, q3, tO, trD, h.t); }
            const bool sw = d1 > d2;                                         // near child first (strict >: ties keep child 1)
            const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
            const uint32_t rn = sw ? asu(q3.w) : asu(q1.w), rf = sw ? asu(q1.w) : asu(q3.w);   // the children's ref16
            stk_put(spB + 128u, rf);                                         // dead store unless `push`
            const bool hitN = dn != 1e30f, push = hitN && df != 1e30f;
            const bool pop = !hitN && spB != laneB;
            cur = hitN ? rn : (pop ? top : kRefDone);
            spB = spB + (push ? 128u : 0u) - (pop ? 128u : 0u);
            if (COUNT && (cur & kRef16TagMask) == 0u && cur != kRefDone) cn.leaf++;
        }
    };
    // the record of `cur` into q0..q3 for the lanes in m (the others keep theirs): record offset of a 16-bit reference = index * record size + section base (layout.h)
    auto load_records = [&](uint64_t m) {
        const uint32_t idx = cur & kRef16IndexMask;
        const bool inter = (cur & kRef16Interior) != 0u;
        uint32_t oa = idx * (inter ? 64u : 48u) + (inter ? 0u : sc.leafOff - 48u);             // NodePair | LeafTri (one multiply-add on selected operands: no divergent arms)
        if (KIND == 1 && (cur & kRef16TlasBit) != 0u)
            oa = (cur & kRef16Interior) ? sc.instOff + idx * 128u : sc.tlasPairOff + idx * 64u;   // TLAS leaf: Instance {invT rows, ids} | TLAS interior: its child pair
        if (lane_in(m)) { q0 = ldg(geom, oa); q1 = ldg(geom, oa + 16u); q2 = ldg(geom, oa + 32u); q3 = ldg(geom, oa + 48u); }
    };

    // ---------------- TRI step of the lanes in m: one triangle of the current leaf (infra/bvh.cpp:203-222, 232-243), on the pre-loaded LeafTri ----------------
    auto tri_step = [&](uint64_t m) {
        CRT_DENS(3, 1); CRT_DENS(4, __popcll(m));
        if (lane_in(m)) {
            if (COUNT) cn.tri++;
            const uint32_t top = stk_top(spB);
            hit_tri(q0, q1, q2, tO, tD, h);
            const bool more = asu(q2.w) > 1u;                                // the leaf's next LeafTri is the next index
            const bool pop = !more && spB != laneB;
            const uint32_t next = more ? cur + 1u : (pop ? top : kRefDone);
            spB -= pop ? 128u : 0u;
            if (COUNT && !more && (next & kRef16TagMask) == 0u && next != kRefDone) cn.leaf++;
            cur = next;
        }
    };

#ifdef CRT_POOL_STAMPS
    unsigned long long pst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        CRT_PSTAMP(p0);
        if (mRes != 0ull) {
            // ---------------- D. walk: the phases of the resident streams (TLAS leaf / NODE / TRI), on the records loaded by the previous trip ----------------
            // what each resident lane is at, from its 16-bit reference: 10 = BVH interior, 00 = leaf triangle (never 0 here), 01 = TLAS interior, 11 = TLAS leaf
            uint64_t mNode, mTlas = 0ull;
            if (KIND == 0) mNode = mRes & __builtin_amdgcn_ballot_w64(cur > 0x7fffu);
            else { mTlas = mRes & __builtin_amdgcn_ballot_w64(cur >= kRef16TlasLeaf); mNode = mRes & __builtin_amdgcn_ballot_w64(cur - kRef16TlasBit < 0x8000u); }
            const uint64_t mTri = mRes & ~(mNode | mTlas);
            // The record loads issued at the end of the previous trip are first needed here.  Naming all four tuples in one
            // empty asm keeps the register allocator from splitting a loaded tuple across the back-edge.
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            CRT_DENS(22, __popcll(mRes));
            if (mNode) { CRT_DENS(1, 1); CRT_DENS(2, __popcll(mNode)); }

            if (KIND == 1 && mTlas) { CRT_DENS(23, 1); CRT_DENS(24, __popcll(mTlas)); }
            uint64_t mBack = 0ull;                                                 // lanes that popped the return marker: BLAS finished, back to the TLAS level
            if (KIND == 1 && mTlas != 0ull) {
                if (lane_in(mTlas)) {
                    // ---------------- TLAS leaf (infra/tlas_bvh.cpp:91-95) -> enter the BLAS (BLASBVH::Intersect, blas_bvh.cpp:376-381) ----------
                    if (COUNT) { cn.tlas++; cn.visits++; }
                    const f3 O = mk3(stf[F_OX * S + sid], stf[F_OY * S + sid], stf[F_OZ * S + sid]);
                    const f3 D = mk3(stf[F_DX * S + sid], stf[F_DY * S + sid], stf[F_DZ * S + sid]);
                    to_object_space(q0, q1, q2, O, D, tO, tD, trD);
                    spB += 128u; stk_put(spB, kRef16Return);
                    const uint32_t next = asu(q3.y);                                 // Instance::rootRef16
                    if (COUNT && (next & kRef16TagMask) == 0u && next != kRefDone) cn.leaf++;
                    cur = next;
                }
                mFinite = (mFinite & ~mTlas) | (mTlas & finite3_mask(trD));
            }
            if (mNode != 0ull) node_step(mNode);
#if CRT_POOL_NODE_STEPS > 1
            // ---------------- further NODE steps in the same trip (while-while): the fixed part of a trip — swap out / in, queue bookkeeping, the pass decision — is
            // paid once for several node steps of the lanes that stay at interior nodes.  The lanes that just stepped fetch their next record now (an interior pair, or
            // the first triangle of a leaf, which then joins THIS trip's TRI phase); everyone else keeps the record the previous trip loaded.
            uint64_t mLeafNow = 0ull;                                              // lanes that reached a leaf in an earlier step of this trip and have its first triangle loaded
#pragma unroll
            for (int rep = 1; rep < CRT_POOL_NODE_STEPS; rep++) {
                const uint64_t mStepped = rep == 1 ? mNode : mNode2;
                mNode2 = mStepped & (KIND == 0 ? __builtin_amdgcn_ballot_w64(cur > 0x7fffu) : __builtin_amdgcn_ballot_w64(cur - kRef16TlasBit < 0x8000u));
                if ((uint32_t)__popcll(mNode2) < (uint32_t)CRT_POOL_NODE2_MIN) { mNode2 = 0ull; break; }
                const uint64_t mLeaf = mStepped & __builtin_amdgcn_ballot_w64((cur & kRef16TagMask) == 0u) & ~__builtin_amdgcn_ballot_w64(cur == kRefDone);
                load_records(mNode2 | mLeaf);
                mLeafNow |= mLeaf;
                CRT_DENS(28, 1); CRT_DENS(29, __popcll(mNode2));
                node_step(mNode2);
            }
            const uint64_t mTriAll = mTri | mLeafNow;
            const uint64_t mTriRan = mTriAll;
#else
            const uint64_t mTriAll = mTri;
            const uint64_t mTriRan = mTriAll;
#endif
            if (mTriAll != 0ull) tri_step(mTriAll);
            if (KIND == 1) {
                // a popped return marker: the BLAS is finished — back to the world-space ray, pop the TLAS entry below (rare: once per BLAS visit)
                mBack = (mNode | mTriRan) & __builtin_amdgcn_ballot_w64(cur == kRef16Return);
                if (mBack != 0ull) {
                    if (lane_in(mBack)) {
                        world_ray();
                        const bool pop = spB != laneB;
                        const uint32_t top = stk_top(spB);
                        cur = pop ? top : kRefDone; spB -= pop ? 128u : 0u;
                        if (COUNT && (cur & kRef16TagMask) == 0u && cur != kRefDone) cn.leaf++;
                    }
                    mFinite = (mFinite & ~mBack) | (mBack & finite3_mask(trD));
                }
            }
            asm volatile("" ::: "memory");          // (the sections exchange stream state through LDS across lanes: nothing is carried over in registers)
            // ---------------- A. swap out: streams whose walk is over park their hit and queue for shading ------------------------------
            const uint64_t mFin = mRes & __builtin_amdgcn_ballot_w64(cur == kRefDone);
            if (mFin != 0ull) {
                CRT_DENS(5, 1); CRT_DENS(6, __popcll(mFin));
                const uint32_t depth = (metaLo >> kMetaDepthShift) & 7u;
                const uint64_t mStop = __builtin_amdgcn_ballot_w64((uint32_t)(h.objIdx + 1) <= 1u) | __builtin_amdgcn_ballot_w64((int)depth >= sc.depthLimit);   // miss, light, or depth limit
                const uint64_t mE = mFin & mStop, mB = mFin & ~mStop;
                if (lane_in(mFin)) {
                    stf[F_T * S + sid] = h.t; stf[F_U * S + sid] = h.u; stf[F_V * S + sid] = h.v;
                    st[F_TRI * S + sid] = (uint32_t)h.triIdx;
                    st[F_META * S + sid] = metaLo | ((uint32_t)(h.objIdx + 1) << kMetaObjShift);
                }
                if (lane_in(mE)) qEnd[(endT + rank_in(mE)) & kQueueMask] = (uint8_t)sid;
                if (lane_in(mB)) qBnc[(bncT + rank_in(mB)) & kQueueMask] = (uint8_t)sid;
                endT += (uint32_t)__popcll(mE); bncT += (uint32_t)__popcll(mB);
                mRes &= ~mFin;
            }
            asm volatile("" ::: "memory");
        }
        CRT_PSTAMP(p1); CRT_PACC(0, p0, p1);
        if (__builtin_expect(mRes == 0ull, 0)) {
            if ((endT - endH) + (bncT - bncH) + (rdyT - rdyH) == 0u) break;       // every stream has rendered its 256 pixels
        }
        if (COUNT) trips++;
        CRT_DENS(0, 1);
        // ---------------- C. swap in: free lanes take the next READY streams; E. the record loads (consumed by the next trip's walk) ----
        {
            const uint32_t nRdy = rdyT - rdyH;
            const uint64_t mFree = ~mRes;
            if (nRdy != 0u) if (mFree != 0ull) {
                const uint32_t myRank = rank_in(mFree);
                const uint64_t mTake = mFree & __builtin_amdgcn_ballot_w64(myRank < nRdy);      // the first min(nRdy, free) free lanes
                CRT_DENS(7, 1); CRT_DENS(8, __popcll(mTake));
                if (lane_in(mTake)) {
                    sid = qRdy[(rdyH + myRank) & kQueueMask];
                    tO = mk3(stf[F_OX * S + sid], stf[F_OY * S + sid], stf[F_OZ * S + sid]);
                    tD = mk3(stf[F_DX * S + sid], stf[F_DY * S + sid], stf[F_DZ * S + sid]);
                    trD = mk3(stf[F_RX * S + sid], stf[F_RY * S + sid], stf[F_RZ * S + sid]);
                    const uint32_t meta = st[F_META * S + sid];
                    h.t = stf[F_T * S + sid]; h.objIdx = (int)(meta >> kMetaObjShift) - 1; h.u = 0; h.v = 0; h.triIdx = -1;
                    metaLo = meta & kMetaLowMask;
                    cur = st[F_CUR * S + sid];
                    const uint32_t pend = st[F_PEND * S + sid];
                    stk_put(laneB + 128u, pend);                                     // a dead store unless the far root child was hit
                    spB = laneB + (pend ? 128u : 0u);
                }
                mFinite = (mFinite & ~mTake) | (mTake & finite3_mask(trD));
                mRes |= mTake;
                rdyH += (uint32_t)__popcll(mTake);
            }
            if (mRes != 0ull) {
                CRT_DENS(21, 1);
                // record offset of a 16-bit reference: index * record size + section base (layout.h); lanes without a stream re-read record 0
                const uint32_t idx = cur & kRef16IndexMask;
                const bool inter = (cur & kRef16Interior) != 0u;
                uint32_t oa = idx * (inter ? 64u : 48u) + (inter ? 0u : sc.leafOff - 48u);             // NodePair | LeafTri (one multiply-add on selected operands: no divergent arms)
                if (KIND == 1 && (cur & kRef16TlasBit) != 0u)
                    oa = (cur & kRef16Interior) ? sc.instOff + idx * 128u : sc.tlasPairOff + idx * 64u;   // TLAS leaf: Instance {invT rows, ids} | TLAS interior: its child pair
                if (!lane_in(mRes)) oa = 0u;
                q0 = ldg(geom, oa); q1 = ldg(geom, oa + 16u); q2 = ldg(geom, oa + 32u); q3 = ldg(geom, oa + 48u);
            }
        }
        asm volatile("" ::: "memory");
        {
            // ---------------- B. shading passes (their latency-free arithmetic also covers the record loads just issued) --------------------
            const uint32_t nEnd = endT - endH, nBnc = bncT - bncH, nRdy = rdyT - rdyH, nRes = (uint32_t)__popcll(mRes);
            // a shading pass waits for a full wavefront of streams unless the walking side runs dry
            const bool starving = nRes + nRdy < (uint32_t)CRT_POOL_STARVE;
            bool runEnd = false, runBnc = false;
            if ((nEnd | nBnc) >= (uint32_t)CRT_POOL_SHADE_MIN || starving) {      // (64 is a power of two: either count >= 64 <=> the OR is; rings hold <= 128)
                runEnd = nEnd >= (uint32_t)CRT_POOL_SHADE_MIN || (starving && nEnd > 0u && nEnd >= nBnc);
                runBnc = nBnc >= (uint32_t)CRT_POOL_SHADE_MIN || (starving && nBnc > nEnd);
            }

            CRT_PSTAMP(p2b); CRT_PACC(1, p1, p2b);
            if (runEnd) {
                // ---------------- B1. END pass: the path of each stream ended (renderer.cpp:54-55, 69) or has not begun ----------------
                const uint32_t n = nEnd < 64u ? nEnd : 64u;
                const bool act = lane < n;
                const uint32_t s = act ? qEnd[(endH + lane) & kQueueMask] : 0u;
                endH += n;
                uint32_t meta = 0, seed = 0; int obj = -1; f3 D = nil3;
                if (act) { meta = st[F_META * S + s]; seed = st[F_SEED * S + s]; obj = (int)(meta >> kMetaObjShift) - 1; D = mk3(stf[F_DX * S + s], stf[F_DY * S + s], stf[F_DZ * S + s]); }
                const bool first = (meta & kMetaFresh) != 0u;
                const int depth = (int)((meta >> kMetaDepthShift) & 7u);
                uint32_t item = meta & kMetaItemMask;
                const bool ended = act && !first, miss = ended && obj == -1;
                CRT_DENS(9, 1); CRT_DENS(10, n); CRT_DENS_MASK(11, miss); CRT_DENS_MASK(13, act && first); CRT_DENS_MASK(25, ended && depth > 0); CRT_DENS_MASK(26, ended && depth > 1); CRT_DENS_MASK(27, ended && depth > 2);
                if (ended && obj >= 2) cn.meshhits++;
                // the path's throughput factors: fetched now (device-scope loads: they were written by this wave's BOUNCE passes, possibly from
                // another lane), needed after the sky lookup; most paths end at depth 0..2 and a level is fetched only by the lanes that deep
                float fk[15];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    fk[3 * k] = fk[3 * k + 1] = fk[3 * k + 2] = 0.0f;
                    if (ended && depth > k) {
#pragma unroll
                        for (int j = 0; j < 3; j++) fk[3 * k + j] = __hip_atomic_load(fac + (3 * k + j) * S + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // GetSkyColor (file_scene.cpp:142-154): the texel fetch is issued here and consumed after the next ray has been set up — the skydome is far
                // larger than the caches, and the (long) latency of this one load is then covered by the arithmetic of ray generation and new_ray
                uint32_t skyTexel = 0u;
                if (miss) {
                    const float phi = crt_atan2f(-D.z, D.x) + CRT_PI, theta = crt_acosf(-D.y);
                    const kernarg_f sk = scene_floats(offsetof(Scene, skyOffset));   // skyOffset, skyW, skyH
                    skyTexel = sc.texels[tex_index(asu(sk[0]), (int)asu(sk[1]), (int)asu(sk[2]), phi * CRT_INV2PI, theta * CRT_INVPI)];
                }
                bool gen = act && first;
                size_t sampleAt = 0;
                if (ended) {
                    uint32_t pix = item, pass = 0;
                    if (passes != 1u) { pix = item / passes; pass = item - pix * passes; }
                    const uint32_t fr = frame0 + s;                                // frame of the launch -> (64-frame window, sample row position)
                    sampleAt = (((size_t)(fr >> 6) * tileCount + tl) * 256u + pix) * rowLen + ((fr & 63u) * passes + pass);
                    item++;
                    gen = item < items;                                            // else: the stream has rendered its 256 pixels
                }
                // the camera (camPos, topLeft, topRight, bottomLeft: 12 floats of the Scene block) is read here, not held in registers across the loop
                const kernarg_f cam = scene_floats(offsetof(Scene, camPos));
                const f3 camPos = mk3(cam[0], cam[1], cam[2]);
                f3 v = camPos;
                if (gen) {                                                         // ProcessTile + Camera::GetPrimaryRay (renderer.cpp:125-126, camera.h:23-30)
                    const uint32_t pix = (passes == 1u) ? item : item / passes;
                    const int x = (int)(tx * 16u + (pix & 15u)), y = (int)(ty * 16u + (pix >> 4));
                    const float jy = rnd(seed);                                    // pinned: first draw is the y jitter
                    const float jx = rnd(seed);
                    const float u = ((float)x + jx) * cam[12], vv = ((float)y + jy) * cam[13];     // invW, invH follow the camera in the Scene block
                    const f3 TL = mk3(cam[3], cam[4], cam[5]), TR = mk3(cam[6], cam[7], cam[8]), BL = mk3(cam[9], cam[10], cam[11]);
                    const f3 P = TL + u * (TR - TL) + vv * (BL - TL);
                    v = P - camPos;
                    cn.primary++;
                }
                CRT_DENS_MASK(12, gen);
                CRT_PSTAMP(e1); CRT_PACC(2, p2b, e1);
                new_ray(gen, s, v, true, camPos, seed, item);                      // depth 0, outside, not fresh
                CRT_PSTAMP(e2); CRT_PACC(3, e1, e2);
                if (ended) {
                    // the finished path's radiance: sky colour / light (24,24,22) / 0 at the depth limit (renderer.cpp:54-55, 69; GetLightColor file_scene.cpp:164-167), times the
                    // throughput factors in recursion order (innermost first: albedo*medium*Sample(...) multiplies on return)
                    f3 L = miss ? tex_unpack(skyTexel) : ((depth >= sc.depthLimit) ? mk3(0, 0, 0) : mk3(24, 24, 22));
#pragma unroll
                    for (int k = 4; k >= 0; k--)
                        if (depth > k) L = mk3(fk[3 * k], fk[3 * k + 1], fk[3 * k + 2]) * L;
                    slab[sampleAt] = make_float4(L.x, L.y, L.z, 0.0f);
                }
                // every load of this pass is named as consumed here, at its end, on every path: the registers they wrote are re-used by the next sections, and a load
                // the compiler cannot prove finished would put a wait for ALL outstanding loads — the record loads in flight included — in front of that first re-use
#ifdef CRT_POOL_STAMPS
                { CRT_PSTAMP(e3); CRT_PACC(4, e2, e3); pst[10]++; }
#endif
                asm volatile("" :: "v"(fk[0]), "v"(fk[1]), "v"(fk[2]), "v"(fk[3]), "v"(fk[4]), "v"(fk[5]), "v"(fk[6]), "v"(fk[7]), "v"(fk[8]), "v"(fk[9]), "v"(fk[10]), "v"(fk[11]), "v"(fk[12]), "v"(fk[13]), "v"(fk[14]), "v"(skyTexel));
            }
            CRT_PSTAMP(p3);
            asm volatile("" ::: "memory");
            if (runBnc) {
                // ---------------- B2. BOUNCE pass: surface hit (floor or mesh) below the depth limit (renderer.cpp:56-99) ----------------
                const uint32_t n = nBnc < 64u ? nBnc : 64u;
                const bool act = lane < n;
                const uint32_t s = act ? qBnc[(bncH + lane) & kQueueMask] : 0u;
                bncH += n;
                f3 O = nil3, D = nil3; float ht = 0, hu = 0, hv = 0; int obj = 1; uint32_t tri = 0, seed = 0, meta = 0;
                if (act) {
                    tri = st[F_TRI * S + s];
                    O = mk3(stf[F_OX * S + s], stf[F_OY * S + s], stf[F_OZ * S + s]);
                    D = mk3(stf[F_DX * S + s], stf[F_DY * S + s], stf[F_DZ * S + s]);
                    ht = stf[F_T * S + s]; hu = stf[F_U * S + s]; hv = stf[F_V * S + s];
                    seed = st[F_SEED * S + s]; meta = st[F_META * S + s]; obj = (int)(meta >> kMetaObjShift) - 1;
                }
                const bool mesh = act && obj >= 2;
                CRT_DENS(14, 1); CRT_DENS(15, n); CRT_DENS_MASK(16, mesh);
                rec4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;                 // the hit triangle's ShadeTri
                if (mesh) { const uint32_t so = sc.shadeOff + tri * 64u; s0 = ldg(geom, so); s1 = ldg(geom, so + 16u); s2 = ldg(geom, so + 32u); s3 = ldg(geom, so + 48u); cn.meshhits++; }
                const bool inside = (meta & kMetaInside) != 0u;
                const int depth = (int)((meta >> kMetaDepthShift) & 7u);
                const uint32_t item = meta & kMetaItemMask;
                float tu = 0, tv = 0; uint32_t tOff = 0; int tW = 0, tH = 0;
                f3 I = O, N = O, absorb = O; float refl = 0, refr = 0;
                f3 v = O, pre = O; bool norm = false, diffuse = false, newInside = false;
                if (act) {
                    I = O + ht * D;
                    if (obj == 1) {                                                // floor: Plane::GetNormal / GetUV (primitives.h:112-133)
                        const kernarg_f fl = scene_floats(offsetof(Scene, floorN));      // floorN[3], floorD, floorInvto
                        const kernarg_f fm = scene_floats(offsetof(Scene, floorMat));    // Material: reflectivity, refractivity, absorption[3], texOffset, texW, texH
                        N = mk3(fl[0], fl[1], fl[2]);
                        if (N.y == 1) {
                            float u = I.x, vv = I.z;
                            u *= fl[4]; vv *= fl[4];
                            tu = u - __builtin_floorf(u); tv = vv - __builtin_floorf(vv);
                        }
                        refl = fm[0]; refr = fm[1];
                        absorb = mk3(fm[2], fm[3], fm[4]);
                        tOff = asu(fm[5]); tW = (int)asu(fm[6]); tH = (int)asu(fm[7]);
                    } else {                                                       // mesh: GetNormal / GetUV (bvh.cpp:290-305, blas_bvh.cpp:391-406)
                        const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
                        const float w = 1 - hu - hv;
                        const f3 Nn = w * n0 + hu * n1 + hv * n2;
                        tu = w * s2.y + hu * s2.w + hv * s3.y;
                        tv = w * s2.z + hu * s3.x + hv * s3.z;
                        const rec4* mp = reinterpret_cast<const rec4*>(sc.mats + (int)asu(s3.w));
                        const rec4 m0 = mp[0], m1 = mp[1];
                        refl = m0.x; refr = m0.y; absorb = mk3(m0.z, m0.w, m1.x);
                        tOff = asu(m1.y); tW = (int)asu(m1.z); tH = (int)asu(m1.w);
                        if (KIND == 0) {
                            N = normalize3(Nn);
                        } else {
                            const uint32_t io = sc.instOff + (uint32_t)(obj - 2) * 128u + 64u;   // Instance::T rows
                            const rec4 r0 = ldg(geom, io), r1 = ldg(geom, io + 16), r2 = ldg(geom, io + 32);
                            N = normalize3(mk3(r0.x * Nn.x + r0.y * Nn.y + r0.z * Nn.z + r0.w * 0.0f,
                                               r1.x * Nn.x + r1.y * Nn.y + r1.z * Nn.z + r1.w * 0.0f,
                                               r2.x * Nn.x + r2.y * Nn.y + r2.z * Nn.z + r2.w * 0.0f));
                        }
                    }
                    if (dot3(N, D) > 0) N = -N;
                    // Material::GetAlbedo: the texel fetch is issued here and consumed after the direction has been drawn (its latency hides behind the rejection loop)
                    uint32_t texel = 0x00ffffffu;
                    if (tW > 0) texel = sc.texels[tex_index(tOff, tW, tH, tu, tv)];
                    f3 medium = mk3(1, 1, 1);
                    if (inside) {
                        const f3 ab = absorb * -ht;
                        medium = mk3(crt_expf(ab.x), crt_expf(ab.y), crt_expf(ab.z));
                    }
                    CRT_PSTAMP(b1); CRT_PACC(5, p3, b1);
                    const float r = rnd(seed);
                    const bool mirror = r < refl, dielectric = !mirror && r < refl + refr;
                    if (mirror) {                                                  // HandleMirror, renderer.cpp:20-25
                        v = D - 2.0f * N * dot3(N, D);
                    } else if (dielectric) {                                       // HandleDielectric, renderer.cpp:27-45
                        v = D - 2.0f * N * dot3(N, D);
                        const float n1 = inside ? 1.2f : 1, n2 = inside ? 1 : 1.2f;
                        const float eta = n1 / n2, cosi = dot3(-D, N);
                        const float cost2 = 1.0f - eta * eta * (1 - cosi * cosi);
                        if (cost2 > 0) {
                            const float a = n1 - n2, b2 = n1 + n2, R0 = (a * a) / (b2 * b2), cc = 1 - cosi;
                            const float Fr = R0 + (1 - R0) * (cc * cc * cc * cc * cc);
                            const f3 T = eta * D + ((eta * cosi - __builtin_sqrtf(__builtin_fabsf(cost2))) * N);
                            if (rnd(seed) > Fr) { v = T; newInside = !inside; }
                        }
                    } else {                                                       // diffuse, renderer.cpp:93-99; diffusereflection tmplmath.h:535-544
                        f3 Rr;
                        do {
                            const float rz = rnd_pm1(seed);                        // draw order pinned z, y, x
                            const float ry = rnd_pm1(seed);
                            const float rx = rnd_pm1(seed);
                            Rr = mk3(rx, ry, rz);
                        } while (dot3(Rr, Rr) > 1);
                        if (dot3(Rr, N) < 0) Rr = Rr * -1.0f;
                        v = Rr; norm = true; diffuse = true;
                    }
                    // albedo (1,1,1 untextured: 0xffffff * (1/255) == 1 exactly) and the factor without the cosine term of the diffuse branch
                    const f3 c = (tW > 0) ? tex_unpack(texel) : mk3(1.0f, 1.0f, 1.0f);
                    if (diffuse) {
                        const f3 brdf = c * CRT_INVPI;
                        pre = medium * brdf * 2.0f * CRT_PI;                       // ... * dot(R, N) once R is normalised
                    } else pre = c * medium;
                    CRT_PSTAMP(b2); CRT_PACC(6, b1, b2);
                    // normalize(R) of the diffuse branch; the bounce's throughput factor and the new origin use the normalised direction
                    const float inv = rcp_exact(__builtin_sqrtf(dot3(v, v)));
                    const f3 nv = norm ? v * inv : v;
                    v = nv;
                    const f3 factor = diffuse ? pre * dot3(nv, N) : pre;
                    float* fd = fac + (3 * depth) * S + s;                         // depth <= 4 here
                    __hip_atomic_store(fd, factor.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(fd + S, factor.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(fd + 2 * S, factor.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    O = I + nv * CRT_EPS;
                    CRT_PSTAMP(b3); CRT_PACC(7, b2, b3);
                }
                CRT_PSTAMP(b4);
                new_ray(act, s, v, false, O, seed, item | ((uint32_t)(depth + 1) << kMetaDepthShift) | (newInside ? kMetaInside : 0u));
#ifdef CRT_POOL_STAMPS
                { CRT_PSTAMP(b5); CRT_PACC(8, b4, b5); pst[11]++; }
#endif
                asm volatile("" :: "v"(s0), "v"(s1), "v"(s2), "v"(s3));            // (as at the end of the END pass)
            }
#ifdef CRT_POOL_STAMPS
            pst[9]++;
#endif
        }
    }
#ifdef CRT_POOL_STAMPS
    if (lane == 0) for (int i = 0; i < 12; i++) atomicAdd(&g_poolStamps[i], pst[i]);
#endif
#ifdef CRT_POOL_DENS
    if (lane == 0) for (int i = 0; i < 32; i++) if (dens[i]) atomicAdd(&g_poolDens[i], (unsigned long long)dens[i]);
#endif

#ifdef CRT_POOL_TIMELINE
    if (lane == 0 && g_poolTimeline) { uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_poolTimeline[3 * (size_t)blockIdx.x] = tl0; g_poolTimeline[3 * (size_t)blockIdx.x + 1] = wall_clock64(); g_poolTimeline[3 * (size_t)blockIdx.x + 2] = hw; }
#endif
    // what this tile costs (100 MHz ticks): this wavefront's duration per 64 streams (close to what the tile's one-stream-per-lane wavefront takes on an idle chip);
    // full groups only — a wavefront with fewer streams than S runs them less densely.  The host orders and plans later jobs with it (abi.cpp plan_job).
    if (tileCost && lane == 0 && nStreams == (uint32_t)S) atomicMax(&tileCost[tl], (uint32_t)((wall_clock64() - clk0) * 64ull / (uint32_t)S));
    // ... and how much of this wavefront ran after the launch's last wavefront had started (the launch's drain: abi.cpp adopt_job_costs)
    if (launchClk && lane == 0) { const unsigned long long now = wall_clock64(), last = __hip_atomic_load(&launchClk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), from = last > clk0 ? last : clk0; if (now > from) atomicAdd(&launchClk[2], now - from); }
    if (COUNT && tileClocks && lane == 0 && groups == 1u) {                     // instrumentation: per-tile wall time + loop trips (one group per tile only)
        tileClocks[2 * tl] = wall_clock64() - clk0;        // 100 MHz constant clock
        tileClocks[2 * tl + 1] = trips;
    }
    uint32_t vals[8] = {cn.rays, cn.primary, cn.interior, cn.leaf, cn.tri, cn.tlas, cn.visits, cn.meshhits};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (!COUNT && i >= 2 && i != 7) continue;
        uint32_t sum = wave_sum(vals[i]);
        if (lane == 0 && sum) atomicAdd(&counters->v[i], (unsigned long long)sum);
    }
}

} // namespace crt

// streams per wavefront: 128 for jobs of >= 128 frames, 64 (streams = lanes) for shorter ones
#ifndef CRT_POOL_STREAMS
#define CRT_POOL_STREAMS 128
#endif
extern "C" uint32_t crt_pool_streams(uint32_t frames) { return frames > 64u ? (uint32_t)CRT_POOL_STREAMS : 64u; }
#ifndef CRT_POOL_EXTRA_LDS
#define CRT_POOL_EXTRA_LDS 0       // occupancy experiments only: unused LDS bytes per wavefront
#endif
// LDS of one wavefront: traversal stacks ((stackDepth + 1 dummy) two-byte entries per lane) + parked stream state + the three one-byte ring queues
extern "C" uint32_t crt_pool_lds_bytes(uint32_t stackDepth, uint32_t streams) { return (stackDepth + 1u) * 64u * 2u + crt::F_COUNT * streams * 4u + 3u * 128u + (uint32_t)CRT_POOL_EXTRA_LDS; }
// bytes of throughput-factor scratch a launch of `windows` 64-frame windows needs behind its sample slab (15 floats per stream; a wave's
// group of streams may reach past the last window, hence 128 stream slots per window)
extern "C" size_t crt_pool_scratch_bytes_per_window(uint32_t tileCount) { return (size_t)tileCount * 128u * 15u * 4u; }

#ifdef CRT_POOL_STAMPS
extern "C" int crt_debug_pool_stamps(unsigned long long* out, int reset)       // after a sync: the section clocks summed over every pool wave since the last reset
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(crt::g_poolStamps), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(crt::g_poolStamps), z, 128) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifdef CRT_POOL_DENS
extern "C" int crt_debug_pool_density(unsigned long long* out, int reset)      // after a sync: the 32 section counters summed over every pool wave since the last reset
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(crt::g_poolDens), 256) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(crt::g_poolDens), z, 256) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifdef CRT_POOL_TIMELINE
static unsigned long long* g_timelineHost = nullptr; static size_t g_timelineCount = 0;
extern "C" size_t crt_debug_pool_timeline(unsigned long long* out, size_t cap)        // after a sync: 3 words per wavefront of the last pool launch
{
    const size_t n = g_timelineCount < cap ? g_timelineCount : cap;
    if (out && n) (void)hipMemcpy(out, g_timelineHost, n * 24, hipMemcpyDeviceToHost);
    return g_timelineCount;
}
#endif
extern "C" hipError_t crt_launch_render_pool(const crt::Scene* sc, void* slab, void* facScratch, crt::Counters* counters, unsigned long long* tileClocks, const uint32_t* tileOrder,
                                             uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX, uint32_t sppFirst,
                                             uint32_t frames, uint32_t passes, int collectStats, uint32_t rankFirst, uint32_t* tileCost, unsigned long long* launchClk, hipStream_t stream)
{
    if (tileCount == 0 || frames == 0) return hipSuccess;
    if (!sc->ref16ok) return hipErrorInvalidValue;                              // the host launches render_tiles_kernel for such scenes
    const uint32_t S = crt_pool_streams(frames);
    const uint32_t groups = (frames + S - 1u) / S;
    if ((unsigned long long)tileCount * groups > 0x7fffffffull) return hipErrorInvalidValue;
    if (rankFirst >= tileCount) return hipSuccess;
    dim3 grid((tileCount - rankFirst) * groups), block(64);
#ifdef CRT_POOL_TIMELINE
    { static unsigned long long* buf = nullptr; static size_t cap = 0;
      if (cap < (size_t)grid.x) { if (buf) (void)hipFree(buf); (void)hipMalloc((void**)&buf, (size_t)grid.x * 24); cap = grid.x; (void)hipMemcpyToSymbol(HIP_SYMBOL(crt::g_poolTimeline), &buf, sizeof(buf)); }
      (void)hipMemsetAsync(buf, 0, (size_t)grid.x * 24, stream); g_timelineHost = buf; g_timelineCount = grid.x; }
#endif
    const uint32_t ldsBytes = crt_pool_lds_bytes(sc->stackDepth, S);
#define CRT_LAUNCH(K, C, SS) hipLaunchKernelGGL((crt::render_pool_kernel<K, C, SS>), grid, block, ldsBytes, stream, *sc, (float4*)slab, (float*)facScratch, counters, tileClocks, tileOrder, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes, groups, rankFirst, tileCost, launchClk)
#define CRT_LAUNCH_S(K, C) do { if (S == 64u) CRT_LAUNCH(K, C, 64); else CRT_LAUNCH(K, C, CRT_POOL_STREAMS); } while (0)
    if (sc->kind == 0) { if (collectStats) CRT_LAUNCH_S(0, true); else CRT_LAUNCH_S(0, false); }
    else { if (collectStats) CRT_LAUNCH_S(1, true); else CRT_LAUNCH_S(1, false); }
#undef CRT_LAUNCH_S
#undef CRT_LAUNCH
    return hipGetLastError();
}
