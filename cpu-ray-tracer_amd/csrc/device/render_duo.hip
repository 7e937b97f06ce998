// render_duo.hip — render_duo_kernel: the stream pool of render_pool.hip split over TWO wavefronts of one workgroup that share one pool of S = 256 parked streams in LDS:
//   * the WALKER wavefront (wave 0) only walks: its 64 lanes hold streams that are inside the acceleration structure (NODE / TRI / TLAS phases, two NODE steps per trip),
//     finished streams are parked and queued for shading, free lanes take the next READY stream;
//   * the SHADER wavefront (wave 1) only shades: END and BOUNCE passes over up to 64 queued streams each, including the start of every stream's next FindNearest
//     (light quad, floor plane, root step), after which the stream is READY to walk — or straight back in a shading queue (84 % of the bunny scene's rays never
//     enter the tree).
// Why: in render_pool_kernel ONE wavefront alternates between the two kinds of work, and of its 128 streams about 64 sit in the END / BOUNCE queues waiting for a
// queue to reach a wavefront's worth — so only ~41 of 64 lanes walk in an average trip and passes run with ~51 lanes.  Here the queues still fill to 64 before a pass
// runs, but 256 streams feed them: the walker's lanes stay (nearly) full and every pass is (nearly) a full wavefront.
//
// The two wavefronts run asynchronously — no barrier anywhere — and exchange stream ids through single-producer / single-consumer byte rings in LDS:
//   walker -> shader   qEndA, qBncA   (streams whose walk is over)            tails published by the walker (ctl[C_END_A], ctl[C_BNC_A])
//   shader -> walker   qRdy           (streams whose new ray enters the tree)  tail published by the shader (ctl[C_RDY])
//   shader -> shader   qEndB, qBncB   (rays that never enter the tree)         private to the shader wavefront
// A producer writes ring entries and the parked stream state first, then publishes the tail with a workgroup-scope release; the consumer reads the tail with an acquire
// and only then the entries (heads are private to the consumer).  The walker also publishes how many streams it holds and has taken (ctl[C_WALK]) so that the shader
// can tell when the walking side runs dry and run a partial pass; the shader counts the streams that have rendered their 256 pixels and raises ctl[C_DONE] when all
// have, which is the walker's exit.  Every wait is a poll with s_sleep: a waiting wavefront issues next to nothing.  Progress: a stream is always in exactly one place
// (a walker lane, or one ring); the walker never waits while it holds a stream, and the shader, when no queue is full, runs a partial pass as soon as the walker
// holds and can take fewer than CRT_POOL_STARVE streams — so the workgroup cannot stall with unfinished streams.
//
// Per stream nothing changes against render_pool_kernel / render_tiles_kernel / the CPU oracle: the same nodes in the same order, every float expression as the
// reference writes it (renderer.cpp:50-131, bvh.cpp:224-258, tlas_bvh.cpp:83-111), so the image is bit-identical.  -ffp-contract=off, IEEE + - * / sqrt only.
#include "pool_common.h"

namespace crt {

constexpr uint32_t kDuoRing = 256u, kDuoRingMask = 255u;          // rings of 256 one-byte stream ids (S <= 256)
enum : uint32_t { C_END_A = 0, C_BNC_A, C_RDY, C_WALK, C_DONE, C_COUNT = 16 };   // control words (dwords) behind the rings

#ifndef CRT_DUO_STARVE
#define CRT_DUO_STARVE (64 * CRT_DUO_SETS + 8)         // the shader runs a partial pass when the walker holds + can take fewer streams than this
#endif
#ifndef CRT_DUO_SETS
#define CRT_DUO_SETS 2           // sets of up to 64 streams the walker alternates between
#endif

#ifdef CRT_DUO_STATS
// diagnostic build only (-DCRT_DUO_STATS, tools/duo_stats.py): [0] walker trips with streams, [1] lanes holding a stream summed, [2] walker idle polls, [3] END passes, [4] their lanes,
// [5] BOUNCE passes, [6] their lanes, [7] shader idle polls, [8] walker clocks busy, [9] walker clocks idle, [10] shader clocks busy, [11] shader clocks idle, [12] second NODE steps, [13] their lanes
__device__ unsigned long long g_duoStats[16];
#define DUO_STAT(i, v) (dst[i] += (unsigned long long)(v))
#else
#define DUO_STAT(i, v)
#endif
// Control words: the fences name the LDS address space only — the rings, the parked state and the words themselves all live in LDS, and a wavefront's LDS operations
// execute in order — so publishing a tail never waits for the global loads a pass has in flight (the sky texel, the throughput factors).
__device__ __forceinline__ uint32_t ctl_load(const uint32_t* p)
{
    const uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return v;
}
__device__ __forceinline__ void ctl_store(uint32_t* p, uint32_t v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int KIND, bool COUNT, int S>
__global__ __launch_bounds__(128, 4) void render_duo_kernel(const Scene sc, float4* __restrict__ slab, float* __restrict__ facScratch, Counters* __restrict__ counters,
                                                           unsigned long long* __restrict__ tileClocks, const uint32_t* __restrict__ tileOrder,
                                                           uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                                           uint32_t sppFirst, uint32_t frames, uint32_t passes, uint32_t groups, uint32_t rankFirst, uint32_t* __restrict__ tileCost, unsigned long long* __restrict__ launchClk)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const bool walker = threadIdx.x < 64u;
    const unsigned long long clk0 = (COUNT || tileCost) ? wall_clock64() : 0ull;
    if (launchClk && threadIdx.x == 0) { atomicMax(&launchClk[0], ~clk0); atomicMax(&launchClk[1], clk0); }
    const uint32_t rank0 = blockIdx.x / groups, grp = blockIdx.x - rank0 * groups, rank = rank0 + rankFirst;
    if (rank >= tileCount) return;                                                   // (whole workgroup)
    const uint32_t tl = tileOrder ? tileOrder[rank] : rank;
    const uint32_t frame0 = grp * (uint32_t)S;
    const uint32_t nStreams = (frames - frame0 < (uint32_t)S) ? frames - frame0 : (uint32_t)S;
    const uint32_t tile = tileFirst + tl * tileStride;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const char* __restrict__ geom = sc.geom;
    float* __restrict__ fac = facScratch + (size_t)blockIdx.x * (15u * (uint32_t)S);

    // LDS: [walker's traversal stacks: (stackDepth + 1) two-byte entries per lane, render_pool.hip's layout][parked stream state][five rings][control words]
    char* const ldsB = reinterpret_cast<char*>(lds);
    const uint32_t laneB = lane * 2u;
    uint32_t* st = lds + (uint32_t)CRT_DUO_SETS * (sc.stackDepth + 1u) * 32u;
    float* stf = reinterpret_cast<float*>(st);
    uint8_t* qEndA = reinterpret_cast<uint8_t*>(st + F_COUNT * S);
    uint8_t* qBncA = qEndA + kDuoRing, * qRdy = qEndA + 2u * kDuoRing, * qEndB = qEndA + 3u * kDuoRing, * qBncB = qEndA + 4u * kDuoRing;
    uint32_t* ctl = reinterpret_cast<uint32_t*>(qEndA + 5u * kDuoRing);

#ifdef CRT_DUO_STATS
    unsigned long long dst[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tPrev = __builtin_readcyclecounter();
#define DUO_CLK(i) { const unsigned long long tNow = __builtin_readcyclecounter(); dst[i] += tNow - tPrev; tPrev = tNow; }
#else
#define DUO_CLK(i)
#endif
    Cnt cn; cn.rays = cn.primary = cn.interior = cn.leaf = cn.tri = cn.tlas = cn.visits = cn.meshhits = 0;
    uint32_t trips = 0;
    const f3 nil3 = mk3(0.0f, 0.0f, 0.0f);

    // ---- set-up by the shader wavefront: every stream starts in its private END ring as "fresh" (the pass only generates its first primary ray), the control words
    // read 0.  ONE barrier, reached unconditionally by both wavefronts before any waiting begins; from there on they run asynchronously.
    if (!walker) {
        for (uint32_t s = lane; s < nStreams; s += 64u) {
            st[F_SEED * S + s] = init_seed(tx + ty * (uint32_t)sc.W + (sppFirst + (frame0 + s) * passes) * 1799u);   // renderer.cpp:120
            st[F_META * S + s] = kMetaFresh;
            qEndB[s] = (uint8_t)s;
        }
        if (lane < C_COUNT) ctl[lane] = 0u;
    }
    __syncthreads();                                                                 // the only barrier: both wavefronts reach it unconditionally, before any waiting begins

    if (walker) {
        // =====================================================  WALKER  =====================================================
        // The walker holds TWO sets of up to 64 streams (one per lane and set) and alternates between them: while the records of one set's next step are in flight
        // the other set's step executes — a wavefront that only walks has nothing else to cover that round trip with (render_pool_kernel covers it with its passes).
        auto stk_top = [&](uint32_t at) -> uint32_t { return *reinterpret_cast<const uint16_t*>(ldsB + at); };
        auto stk_put = [&](uint32_t at, uint32_t v) { *reinterpret_cast<uint16_t*>(ldsB + at) = (uint16_t)v; };
        struct WSet {
            uint64_t mRes, mFinite; uint32_t sid, cur, spB, base, metaLo; f3 tO, tD, trD; Hit h; rec4 q0, q1, q2, q3;
        };
        WSet W[CRT_DUO_SETS];
#pragma unroll
        for (int k = 0; k < CRT_DUO_SETS; k++) {
            W[k].mRes = 0ull; W[k].mFinite = ~0ull; W[k].sid = 0; W[k].cur = kRefDone; W[k].base = laneB + (uint32_t)k * (sc.stackDepth + 1u) * 128u; W[k].spB = W[k].base; W[k].metaLo = 0u;
            W[k].tO = nil3; W[k].tD = nil3; W[k].trD = nil3; W[k].h.t = 1e34f; W[k].h.u = 0; W[k].h.v = 0; W[k].h.objIdx = -1; W[k].h.triIdx = -1;
            const rec4 z = {0, 0, 0, 0}; W[k].q0 = z; W[k].q1 = z; W[k].q2 = z; W[k].q3 = z;
        }
        uint32_t endAT = 0, bncAT = 0, rdyH = 0;                                     // tails this wavefront owns, head of the ring it consumes
        auto trip = [&](WSet& w) {
            if (w.mRes != 0ull) {
                DUO_STAT(0, 1); DUO_STAT(1, __popcll(w.mRes));
                uint64_t mNode, mTlas = 0ull;
                if (KIND == 0) mNode = w.mRes & __builtin_amdgcn_ballot_w64(w.cur > 0x7fffu);
                else { mTlas = w.mRes & __builtin_amdgcn_ballot_w64(w.cur >= kRef16TlasLeaf); mNode = w.mRes & __builtin_amdgcn_ballot_w64(w.cur - kRef16TlasBit < 0x8000u); }
                const uint64_t mTri = w.mRes & ~(mNode | mTlas);
                asm volatile("" : "+v"(w.q0), "+v"(w.q1), "+v"(w.q2), "+v"(w.q3));
                if (KIND == 1 && mTlas != 0ull) {
                    if (lane_in(mTlas)) {                                            // TLAS leaf (tlas_bvh.cpp:91-95) -> enter the BLAS (blas_bvh.cpp:376-381)
                        if (COUNT) { cn.tlas++; cn.visits++; }
                        const f3 O = mk3(stf[F_OX * S + w.sid], stf[F_OY * S + w.sid], stf[F_OZ * S + w.sid]);
                        const f3 D = mk3(stf[F_DX * S + w.sid], stf[F_DY * S + w.sid], stf[F_DZ * S + w.sid]);
                        to_object_space(w.q0, w.q1, w.q2, O, D, w.tO, w.tD, w.trD);
                        w.spB += 128u; stk_put(w.spB, kRef16Return);
                        const uint32_t next = asu(w.q3.y);
                        if (COUNT && (next & kRef16TagMask) == 0u && next != kRefDone) cn.leaf++;
                        w.cur = next;
                    }
                    w.mFinite = (w.mFinite & ~mTlas) | (mTlas & finite3_mask(w.trD));
                }
                if (mNode != 0ull) {                                                 // NODE step (infra/bvh.cpp:244-257 / tlas_bvh.cpp:96-110) on the pre-loaded NodePair
                    const bool allFinite = (mNode & ~w.mFinite) == 0ull;
                    if (lane_in(mNode)) {
                        if (COUNT) { if (KIND == 1 && (w.cur & kRef16TlasBit) != 0u) cn.tlas++; else cn.interior++; }
                        const uint32_t top = stk_top(w.spB);
                        float d1, d2;
                        if (__builtin_expect(allFinite, 1)) { d1 = box_fast(w.q0, w.q1, w.tO, w.trD, w.h.t); d2 = box_fast(w.q2, w.q3, w.tO, w.trD, w.h.t); }
                        else { d1 = box_exact(w.q0, w.q1, w.tO, w.trD, w.h.t); d2 = box_exact(w.q2, w.q3, w.tO, w.trD, w.h.t); }
                        const bool sw = d1 > d2;
                        const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
                        const uint32_t rn = sw ? asu(w.q3.w) : asu(w.q1.w), rf = sw ? asu(w.q1.w) : asu(w.q3.w);
                        stk_put(w.spB + 128u, rf);
                        const bool hitN = dn != 1e30f, push = hitN && df != 1e30f;
                        const bool pop = !hitN && w.spB != w.base;
                        w.cur = hitN ? rn : (pop ? top : kRefDone);
                        w.spB = w.spB + (push ? 128u : 0u) - (pop ? 128u : 0u);
                        if (COUNT && (w.cur & kRef16TagMask) == 0u && w.cur != kRefDone) cn.leaf++;
                    }
                }
                if (mTri != 0ull) {
                    if (lane_in(mTri)) {                                             // one triangle of the current leaf (infra/bvh.cpp:203-222, 232-243)
                        if (COUNT) cn.tri++;
                        const uint32_t top = stk_top(w.spB);
                        hit_tri(w.q0, w.q1, w.q2, w.tO, w.tD, w.h);
                        const bool more = asu(w.q2.w) > 1u;
                        const bool pop = !more && w.spB != w.base;
                        const uint32_t next = more ? w.cur + 1u : (pop ? top : kRefDone);
                        w.spB -= pop ? 128u : 0u;
                        if (COUNT && !more && (next & kRef16TagMask) == 0u && next != kRefDone) cn.leaf++;
                        w.cur = next;
                    }
                }
                if (KIND == 1) {
                    const uint64_t mBack = (mNode | mTri) & __builtin_amdgcn_ballot_w64(w.cur == kRef16Return);
                    if (mBack != 0ull) {                                             // a popped return marker: the BLAS is finished — back to the world-space ray
                        if (lane_in(mBack)) {
                            w.tO = mk3(stf[F_OX * S + w.sid], stf[F_OY * S + w.sid], stf[F_OZ * S + w.sid]);
                            w.tD = mk3(stf[F_DX * S + w.sid], stf[F_DY * S + w.sid], stf[F_DZ * S + w.sid]);
                            w.trD = mk3(stf[F_RX * S + w.sid], stf[F_RY * S + w.sid], stf[F_RZ * S + w.sid]);
                            const bool pop = w.spB != w.base;
                            const uint32_t top = stk_top(w.spB);
                            w.cur = pop ? top : kRefDone; w.spB -= pop ? 128u : 0u;
                            if (COUNT && (w.cur & kRef16TagMask) == 0u && w.cur != kRefDone) cn.leaf++;
                        }
                        w.mFinite = (w.mFinite & ~mBack) | (mBack & finite3_mask(w.trD));
                    }
                }
                asm volatile("" ::: "memory");
                // swap out: streams whose walk is over park their hit and go to the shader's rings
                const uint64_t mFin = w.mRes & __builtin_amdgcn_ballot_w64(w.cur == kRefDone);
                if (mFin != 0ull) {
                    const uint32_t depth = (w.metaLo >> kMetaDepthShift) & 7u;
                    const uint64_t mStop = __builtin_amdgcn_ballot_w64((uint32_t)(w.h.objIdx + 1) <= 1u) | __builtin_amdgcn_ballot_w64((int)depth >= sc.depthLimit);
                    const uint64_t mE = mFin & mStop, mB = mFin & ~mStop;
                    if (lane_in(mFin)) {
                        stf[F_T * S + w.sid] = w.h.t; stf[F_U * S + w.sid] = w.h.u; stf[F_V * S + w.sid] = w.h.v;
                        st[F_TRI * S + w.sid] = (uint32_t)w.h.triIdx;
                        st[F_META * S + w.sid] = w.metaLo | ((uint32_t)(w.h.objIdx + 1) << kMetaObjShift);
                    }
                    if (lane_in(mE)) qEndA[(endAT + rank_in(mE)) & kDuoRingMask] = (uint8_t)w.sid;
                    if (lane_in(mB)) qBncA[(bncAT + rank_in(mB)) & kDuoRingMask] = (uint8_t)w.sid;
                    endAT += (uint32_t)__popcll(mE); bncAT += (uint32_t)__popcll(mB);
                    w.mRes &= ~mFin;
                    if (lane == 0) { if (mE) ctl_store(ctl + C_END_A, endAT); if (mB) ctl_store(ctl + C_BNC_A, bncAT); }      // release: entries and parked state first
                }
                asm volatile("" ::: "memory");
            }
            // swap in: free lanes of this set take the next READY streams
            const uint64_t mFree = ~w.mRes;
            if (mFree != 0ull) {
                const uint32_t rdyT = ctl_load(ctl + C_RDY);                          // acquire: the entries and the parked rays behind it are visible
                const uint32_t nRdy = rdyT - rdyH;
                if (nRdy != 0u) {
                    const uint32_t myRank = rank_in(mFree);
                    const uint64_t mTake = mFree & __builtin_amdgcn_ballot_w64(myRank < nRdy);
                    if (lane_in(mTake)) {
                        w.sid = qRdy[(rdyH + myRank) & kDuoRingMask];
                        w.tO = mk3(stf[F_OX * S + w.sid], stf[F_OY * S + w.sid], stf[F_OZ * S + w.sid]);
                        w.tD = mk3(stf[F_DX * S + w.sid], stf[F_DY * S + w.sid], stf[F_DZ * S + w.sid]);
                        w.trD = mk3(stf[F_RX * S + w.sid], stf[F_RY * S + w.sid], stf[F_RZ * S + w.sid]);
                        const uint32_t meta = st[F_META * S + w.sid];
                        w.h.t = stf[F_T * S + w.sid]; w.h.objIdx = (int)(meta >> kMetaObjShift) - 1; w.h.u = 0; w.h.v = 0; w.h.triIdx = -1;
                        w.metaLo = meta & kMetaLowMask;
                        w.cur = st[F_CUR * S + w.sid];
                        const uint32_t pend = st[F_PEND * S + w.sid];
                        stk_put(w.base + 128u, pend);
                        w.spB = w.base + (pend ? 128u : 0u);
                    }
                    w.mFinite = (w.mFinite & ~mTake) | (mTake & finite3_mask(w.trD));
                    w.mRes |= mTake;
                    rdyH += (uint32_t)__popcll(mTake);
                }
            }
            if (w.mRes != 0ull) {                                                    // the record loads, consumed by this set's next trip — the other set's trip runs meanwhile
                const uint32_t idx = w.cur & kRef16IndexMask;
                const bool inter = (w.cur & kRef16Interior) != 0u;
                uint32_t oa = idx * (inter ? 64u : 48u) + (inter ? 0u : sc.leafOff - 48u);
                if (KIND == 1 && (w.cur & kRef16TlasBit) != 0u)
                    oa = (w.cur & kRef16Interior) ? sc.instOff + idx * 128u : sc.tlasPairOff + idx * 64u;
                if (!lane_in(w.mRes)) oa = 0u;
                w.q0 = ldg(geom, oa); w.q1 = ldg(geom, oa + 16u); w.q2 = ldg(geom, oa + 32u); w.q3 = ldg(geom, oa + 48u);
            }
        };
        for (;;) {
            uint32_t held = 0;
#pragma unroll
            for (int k = 0; k < CRT_DUO_SETS; k++) { trip(W[k]); held += (uint32_t)__popcll(W[k].mRes); }
            if (COUNT) trips++;
            if (lane == 0) ctl_store(ctl + C_WALK, (rdyH << 8) | held);              // what this side holds and has taken: the shader's starvation test
            if (held == 0u) {
                if (ctl_load(ctl + C_DONE) != 0u) break;                              // every stream has rendered its 256 pixels
                DUO_CLK(8);
                __builtin_amdgcn_s_sleep(4);
                DUO_STAT(2, 1); DUO_CLK(9);
            }
        }
    } else {
        // =====================================================  SHADER  =====================================================
        const uint32_t items = 256u * passes;
        const uint32_t rowLen = 64u * passes;
        uint32_t endAH = 0, bncAH = 0, endBH = 0, endBT = nStreams, bncBH = 0, bncBT = 0, rdyT = 0;
        uint32_t finished = 0;
        // the start of a stream's next scene.FindNearest (render_pool.hip new_ray): normalise, reciprocal direction, light quad, floor plane, root step; parks the ray and
        // queues the stream: READY for the walker, or this wavefront's own END / BOUNCE ring when the ray never enters the tree
        auto new_ray = [&](bool act, uint32_t s, f3 v, bool norm, f3 O, uint32_t seed, uint32_t meta) {
            f3 D = v, rD = v; Hit nh; nh.t = 1e34f; nh.u = 0; nh.v = 0; nh.objIdx = -1; nh.triIdx = -1;
            uint32_t ncur = kRefDone, pend = 0u;
            if (act) {
                const float inv = rcp_exact(__builtin_sqrtf(dot3(v, v)));
                D = norm ? v * inv : v;
                rD = rcp_exact3(D);
                cn.rays++;
                {
                    const kernarg_f lp = scene_floats(offsetof(Scene, lightInvT));
                    const kernarg_f ax = scene_floats(offsetof(Scene, lightAxis));
                    LightFloor lf;
#pragma unroll
                    for (int i = 0; i < 12; i++) lf.lightInvT[i] = lp[i];
                    lf.lightSize = lp[15]; lf.floorN[0] = lp[19]; lf.floorN[1] = lp[20]; lf.floorN[2] = lp[21]; lf.floorD = lp[22];
                    lf.lightAxis = asu(ax[0]); lf.floorAxisY = asu(ax[1]);
                    hit_light_floor(lf, O, D, nh);
                }
                if (sc.rootIsPair) {
                    const kernarg_f rp = scene_floats(offsetof(Scene, rootPair));
                    const rec4 a0 = {rp[0], rp[1], rp[2], rp[3]}, a1 = {rp[4], rp[5], rp[6], rp[7]};
                    const rec4 b0 = {rp[8], rp[9], rp[10], rp[11]}, b1 = {rp[12], rp[13], rp[14], rp[15]};
                    float d1, d2;
                    if (__builtin_amdgcn_ballot_w64(!finite3(rD)) == 0ull) { d1 = box_fast(a0, a1, O, rD, nh.t); d2 = box_fast(b0, b1, O, rD, nh.t); }
                    else { d1 = box_exact(a0, a1, O, rD, nh.t); d2 = box_exact(b0, b1, O, rD, nh.t); }
                    const bool sw = d1 > d2;
                    const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
                    const uint32_t rn = sw ? asu(b1.w) : asu(a1.w), rf = sw ? asu(a1.w) : asu(b1.w);
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
            const uint32_t depth = (meta >> kMetaDepthShift) & 7u;
            const uint64_t mAct = __builtin_amdgcn_ballot_w64(act);
            const uint64_t mW = mAct & __builtin_amdgcn_ballot_w64(ncur != kRefDone);
            const uint64_t mStop = __builtin_amdgcn_ballot_w64((uint32_t)(nh.objIdx + 1) <= 1u) | __builtin_amdgcn_ballot_w64((int)depth >= sc.depthLimit);
            const uint64_t mE = mAct & ~mW & mStop, mB = mAct & ~mW & ~mStop;
            if (lane_in(mW)) qRdy[(rdyT + rank_in(mW)) & kDuoRingMask] = (uint8_t)s;
            if (lane_in(mE)) qEndB[(endBT + rank_in(mE)) & kDuoRingMask] = (uint8_t)s;
            if (lane_in(mB)) qBncB[(bncBT + rank_in(mB)) & kDuoRingMask] = (uint8_t)s;
            rdyT += (uint32_t)__popcll(mW); endBT += (uint32_t)__popcll(mE); bncBT += (uint32_t)__popcll(mB);
            if (mW != 0ull && lane == 0) ctl_store(ctl + C_RDY, rdyT);                // release: ring entries and parked rays first
        };
        for (;;) {
            const uint32_t endAT = ctl_load(ctl + C_END_A), bncAT = ctl_load(ctl + C_BNC_A);      // acquire: the walker's entries and parked hits
            const uint32_t nEndB = endBT - endBH, nBncB = bncBT - bncBH;
            const uint32_t nEnd = nEndB + (endAT - endAH), nBnc = nBncB + (bncAT - bncAH);
            const uint32_t w = ctl_load(ctl + C_WALK);
            const uint32_t supply = (w & 0xffu) + ((rdyT - (w >> 8)) & 0xffffffu);        // streams the walker holds + READY ones it has not taken yet
            const bool starving = supply < (uint32_t)CRT_DUO_STARVE;
            bool runEnd = false, runBnc = false;
            if ((nEnd | nBnc) >= 64u || starving) {
                runEnd = nEnd >= 64u || (starving && nEnd > 0u && nEnd >= nBnc);
                runBnc = nBnc >= 64u || (starving && nBnc > nEnd);
            }
            if (!runEnd && !runBnc) {
                if (finished == nStreams) { if (lane == 0) ctl_store(ctl + C_DONE, 1u); break; }
                DUO_CLK(10);
                __builtin_amdgcn_s_sleep(4);
                DUO_STAT(7, 1); DUO_CLK(11);
                continue;
            }
            if (runEnd) {
                // ---------------- END pass: the path of each stream ended (renderer.cpp:54-55, 69) or has not begun ----------------
                const uint32_t n = nEnd < 64u ? nEnd : 64u;
                DUO_STAT(3, 1); DUO_STAT(4, n);
                const uint32_t nb = nEndB < n ? nEndB : n;                            // this wavefront's own entries first, then the walker's
                const bool act = lane < n;
                uint32_t s = 0u;
                if (act) s = lane < nb ? qEndB[(endBH + lane) & kDuoRingMask] : qEndA[(endAH + lane - nb) & kDuoRingMask];
                endBH += nb; endAH += n - nb;
                uint32_t meta = 0, seed = 0; int obj = -1; f3 D = nil3;
                if (act) { meta = st[F_META * S + s]; seed = st[F_SEED * S + s]; obj = (int)(meta >> kMetaObjShift) - 1; D = mk3(stf[F_DX * S + s], stf[F_DY * S + s], stf[F_DZ * S + s]); }
                const bool first = (meta & kMetaFresh) != 0u;
                const int depth = (int)((meta >> kMetaDepthShift) & 7u);
                uint32_t item = meta & kMetaItemMask;
                const bool ended = act && !first, miss = ended && obj == -1;
                if (ended && obj >= 2) cn.meshhits++;
                float fk[15];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    fk[3 * k] = fk[3 * k + 1] = fk[3 * k + 2] = 0.0f;
                    if (ended && depth > k) {
#pragma unroll
                        for (int j = 0; j < 3; j++) fk[3 * k + j] = __hip_atomic_load(fac + (3 * k + j) * S + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                uint32_t skyTexel = 0u;
                if (miss) {                                                          // GetSkyColor (file_scene.cpp:142-154)
                    const float phi = crt_atan2f(-D.z, D.x) + CRT_PI, theta = crt_acosf(-D.y);
                    const kernarg_f sk = scene_floats(offsetof(Scene, skyOffset));
                    skyTexel = sc.texels[tex_index(asu(sk[0]), (int)asu(sk[1]), (int)asu(sk[2]), phi * CRT_INV2PI, theta * CRT_INVPI)];
                }
                bool gen = act && first;
                size_t sampleAt = 0;
                if (ended) {
                    uint32_t pix = item, pass = 0;
                    if (passes != 1u) { pix = item / passes; pass = item - pix * passes; }
                    const uint32_t fr = frame0 + s;
                    sampleAt = (((size_t)(fr >> 6) * tileCount + tl) * 256u + pix) * rowLen + ((fr & 63u) * passes + pass);
                    item++;
                    gen = item < items;
                }
                finished += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ended && !gen));      // streams that have rendered their 256 pixels
                const kernarg_f cam = scene_floats(offsetof(Scene, camPos));
                const f3 camPos = mk3(cam[0], cam[1], cam[2]);
                f3 v = camPos;
                if (gen) {                                                           // ProcessTile + Camera::GetPrimaryRay (renderer.cpp:125-126, camera.h:23-30)
                    const uint32_t pix = (passes == 1u) ? item : item / passes;
                    const int x = (int)(tx * 16u + (pix & 15u)), y = (int)(ty * 16u + (pix >> 4));
                    const float jy = rnd(seed);
                    const float jx = rnd(seed);
                    const float u = ((float)x + jx) * cam[12], vv = ((float)y + jy) * cam[13];
                    const f3 TL = mk3(cam[3], cam[4], cam[5]), TR = mk3(cam[6], cam[7], cam[8]), BL = mk3(cam[9], cam[10], cam[11]);
                    const f3 P = TL + u * (TR - TL) + vv * (BL - TL);
                    v = P - camPos;
                    cn.primary++;
                }
                new_ray(gen, s, v, true, camPos, seed, item);
                if (ended) {
                    f3 L = miss ? tex_unpack(skyTexel) : ((depth >= sc.depthLimit) ? mk3(0, 0, 0) : mk3(24, 24, 22));
#pragma unroll
                    for (int k = 4; k >= 0; k--)
                        if (depth > k) L = mk3(fk[3 * k], fk[3 * k + 1], fk[3 * k + 2]) * L;
                    slab[sampleAt] = make_float4(L.x, L.y, L.z, 0.0f);
                }
                asm volatile("" :: "v"(fk[0]), "v"(fk[1]), "v"(fk[2]), "v"(fk[3]), "v"(fk[4]), "v"(fk[5]), "v"(fk[6]), "v"(fk[7]), "v"(fk[8]), "v"(fk[9]), "v"(fk[10]), "v"(fk[11]), "v"(fk[12]), "v"(fk[13]), "v"(fk[14]), "v"(skyTexel));
            }
            asm volatile("" ::: "memory");
            if (runBnc) {
                // ---------------- BOUNCE pass: surface hit (floor or mesh) below the depth limit (renderer.cpp:56-99) ----------------
                const uint32_t n = nBnc < 64u ? nBnc : 64u;
                DUO_STAT(5, 1); DUO_STAT(6, n);
                const uint32_t nb = nBncB < n ? nBncB : n;
                const bool act = lane < n;
                uint32_t s = 0u;
                if (act) s = lane < nb ? qBncB[(bncBH + lane) & kDuoRingMask] : qBncA[(bncAH + lane - nb) & kDuoRingMask];
                bncBH += nb; bncAH += n - nb;
                f3 O = nil3, D = nil3; float ht = 0, hu = 0, hv = 0; int obj = 1; uint32_t tri = 0, seed = 0, meta = 0;
                if (act) {
                    tri = st[F_TRI * S + s];
                    O = mk3(stf[F_OX * S + s], stf[F_OY * S + s], stf[F_OZ * S + s]);
                    D = mk3(stf[F_DX * S + s], stf[F_DY * S + s], stf[F_DZ * S + s]);
                    ht = stf[F_T * S + s]; hu = stf[F_U * S + s]; hv = stf[F_V * S + s];
                    seed = st[F_SEED * S + s]; meta = st[F_META * S + s]; obj = (int)(meta >> kMetaObjShift) - 1;
                }
                const bool mesh = act && obj >= 2;
                rec4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
                if (mesh) { const uint32_t so = sc.shadeOff + tri * 64u; s0 = ldg(geom, so); s1 = ldg(geom, so + 16u); s2 = ldg(geom, so + 32u); s3 = ldg(geom, so + 48u); cn.meshhits++; }
                const bool inside = (meta & kMetaInside) != 0u;
                const int depth = (int)((meta >> kMetaDepthShift) & 7u);
                const uint32_t item = meta & kMetaItemMask;
                float tu = 0, tv = 0; uint32_t tOff = 0; int tW = 0, tH = 0;
                f3 I = O, N = O, absorb = O; float refl = 0, refr = 0;
                f3 v = O, pre = O; bool norm = false, diffuse = false, newInside = false;
                if (act) {
                    I = O + ht * D;
                    if (obj == 1) {                                                  // floor: Plane::GetNormal / GetUV (primitives.h:112-133)
                        const kernarg_f fl = scene_floats(offsetof(Scene, floorN));
                        const kernarg_f fm = scene_floats(offsetof(Scene, floorMat));
                        N = mk3(fl[0], fl[1], fl[2]);
                        if (N.y == 1) {
                            float u = I.x, vv = I.z;
                            u *= fl[4]; vv *= fl[4];
                            tu = u - __builtin_floorf(u); tv = vv - __builtin_floorf(vv);
                        }
                        refl = fm[0]; refr = fm[1];
                        absorb = mk3(fm[2], fm[3], fm[4]);
                        tOff = asu(fm[5]); tW = (int)asu(fm[6]); tH = (int)asu(fm[7]);
                    } else {                                                         // mesh: GetNormal / GetUV (bvh.cpp:290-305, blas_bvh.cpp:391-406)
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
                            const uint32_t io = sc.instOff + (uint32_t)(obj - 2) * 128u + 64u;
                            const rec4 r0 = ldg(geom, io), r1 = ldg(geom, io + 16), r2 = ldg(geom, io + 32);
                            N = normalize3(mk3(r0.x * Nn.x + r0.y * Nn.y + r0.z * Nn.z + r0.w * 0.0f,
                                               r1.x * Nn.x + r1.y * Nn.y + r1.z * Nn.z + r1.w * 0.0f,
                                               r2.x * Nn.x + r2.y * Nn.y + r2.z * Nn.z + r2.w * 0.0f));
                        }
                    }
                    if (dot3(N, D) > 0) N = -N;
                    uint32_t texel = 0x00ffffffu;
                    if (tW > 0) texel = sc.texels[tex_index(tOff, tW, tH, tu, tv)];
                    f3 medium = mk3(1, 1, 1);
                    if (inside) {
                        const f3 ab = absorb * -ht;
                        medium = mk3(crt_expf(ab.x), crt_expf(ab.y), crt_expf(ab.z));
                    }
                    const float r = rnd(seed);
                    const bool mirror = r < refl, dielectric = !mirror && r < refl + refr;
                    if (mirror) {                                                    // HandleMirror, renderer.cpp:20-25
                        v = D - 2.0f * N * dot3(N, D);
                    } else if (dielectric) {                                         // HandleDielectric, renderer.cpp:27-45
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
                    } else {                                                         // diffuse, renderer.cpp:93-99; diffusereflection tmplmath.h:535-544
                        f3 Rr;
                        do {
                            const float rz = rnd_pm1(seed);                          // draw order pinned z, y, x
                            const float ry = rnd_pm1(seed);
                            const float rx = rnd_pm1(seed);
                            Rr = mk3(rx, ry, rz);
                        } while (dot3(Rr, Rr) > 1);
                        if (dot3(Rr, N) < 0) Rr = Rr * -1.0f;
                        v = Rr; norm = true; diffuse = true;
                    }
                    const f3 c = (tW > 0) ? tex_unpack(texel) : mk3(1.0f, 1.0f, 1.0f);
                    if (diffuse) {
                        const f3 brdf = c * CRT_INVPI;
                        pre = medium * brdf * 2.0f * CRT_PI;
                    } else pre = c * medium;
                    const float inv = rcp_exact(__builtin_sqrtf(dot3(v, v)));
                    const f3 nv = norm ? v * inv : v;
                    v = nv;
                    const f3 factor = diffuse ? pre * dot3(nv, N) : pre;
                    float* fd = fac + (3 * depth) * S + s;
                    __hip_atomic_store(fd, factor.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(fd + S, factor.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(fd + 2 * S, factor.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    O = I + nv * CRT_EPS;
                }
                new_ray(act, s, v, false, O, seed, item | ((uint32_t)(depth + 1) << kMetaDepthShift) | (newInside ? kMetaInside : 0u));
                asm volatile("" :: "v"(s0), "v"(s1), "v"(s2), "v"(s3));
            }
            asm volatile("" ::: "memory");
        }
    }

#ifdef CRT_DUO_STATS
    if (walker) { DUO_CLK(8); } else { DUO_CLK(10); }
    if (lane == 0) for (int i = 0; i < 16; i++) if (dst[i]) atomicAdd(&g_duoStats[i], dst[i]);
#endif
    // what this tile costs (100 MHz ticks per 64 streams; full groups only), the launch's drain, instrumentation — by the walker, which leaves after the shader has finished
    if (walker) {
        if (tileCost && lane == 0 && nStreams == (uint32_t)S) atomicMax(&tileCost[tl], (uint32_t)((wall_clock64() - clk0) * 64ull / (uint32_t)S));
        if (launchClk && lane == 0) { const unsigned long long now = wall_clock64(), last = __hip_atomic_load(&launchClk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), from = last > clk0 ? last : clk0; if (now > from) atomicAdd(&launchClk[2], 2ull * (now - from)); }   // (two wavefronts' worth of slots)
        if (COUNT && tileClocks && lane == 0 && groups == 1u) { tileClocks[2 * tl] = wall_clock64() - clk0; tileClocks[2 * tl + 1] = trips; }
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

#ifdef CRT_DUO_STATS
extern "C" int crt_debug_duo_stats(unsigned long long* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(crt::g_duoStats), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(crt::g_duoStats), z, 128) != hipSuccess) return -1; }
    return 0;
}
#endif
extern "C" uint32_t crt_duo_streams() { return 256u; }
// LDS of one workgroup: the walker's traversal stacks + 256 parked streams + five byte rings + control words
extern "C" uint32_t crt_duo_lds_bytes(uint32_t stackDepth) { return (uint32_t)CRT_DUO_SETS * (stackDepth + 1u) * 64u * 2u + crt::F_COUNT * 256u * 4u + 5u * crt::kDuoRing + crt::C_COUNT * 4u; }

extern "C" hipError_t crt_launch_render_duo(const crt::Scene* sc, void* slab, void* facScratch, crt::Counters* counters, unsigned long long* tileClocks, const uint32_t* tileOrder,
                                            uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX, uint32_t sppFirst,
                                            uint32_t frames, uint32_t passes, int collectStats, uint32_t rankFirst, uint32_t* tileCost, unsigned long long* launchClk, hipStream_t stream)
{
    if (tileCount == 0 || frames == 0) return hipSuccess;
    if (!sc->ref16ok) return hipErrorInvalidValue;
    const uint32_t S = 256u;
    const uint32_t groups = (frames + S - 1u) / S;
    if ((unsigned long long)tileCount * groups > 0x7fffffffull) return hipErrorInvalidValue;
    if (rankFirst >= tileCount) return hipSuccess;
    dim3 grid((tileCount - rankFirst) * groups), block(128);
    const uint32_t ldsBytes = crt_duo_lds_bytes(sc->stackDepth);
#define CRT_LAUNCH(K, C) hipLaunchKernelGGL((crt::render_duo_kernel<K, C, 256>), grid, block, ldsBytes, stream, *sc, (float4*)slab, (float*)facScratch, counters, tileClocks, tileOrder, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes, groups, rankFirst, tileCost, launchClk)
    if (sc->kind == 0) { if (collectStats) CRT_LAUNCH(0, true); else CRT_LAUNCH(0, false); }
    else { if (collectStats) CRT_LAUNCH(1, true); else CRT_LAUNCH(1, false); }
#undef CRT_LAUNCH
    return hipGetLastError();
}
