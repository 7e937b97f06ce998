// kernels.hip — hand-written gfx950 (CDNA4) kernels of the path-tracing hot path.
//
//   render_tiles_kernel   one 64-lane wavefront per (16x16 image tile, 64-frame window); lane f owns the RNG stream of frame f
//                         of that tile (the reference seeds one xorshift32 stream per (tile, frame) and consumes it serially
//                         over the tile's 256 pixels — "3. PathTracer/renderer.cpp":117-131).  One grid covers every window
//                         of a crt_render job, expensive tiles first.  The wave is a small wavefront pipeline of its own:
//                         every lane is a state machine {needs shading / ray-gen, at a TLAS node, at a BVH interior node, at
//                         a leaf triangle}, and each trip of the wave's loop runs the phases that have lanes waiting
//                         (ballot population counts).  Lanes never wait for the longest ray of the wave: a lane whose path
//                         ends regenerates its next pixel's primary ray and re-enters traversal while its neighbours keep
//                         walking.  Traversal stacks and the paths' throughput factors are per-lane columns in LDS.
//                         Finished paths write their radiance sample to the sample slab in HBM.
//   accumulate_kernel     adds the slab's samples to the float4 accumulator in (window, frame, pass) order (bit-exact with the
//                         reference's `accumulator[..] +=` order, renderer.cpp:124), streaming the slab at HBM rate — no float
//                         atomics anywhere.
//   find_nearest_kernel   scene.FindNearest for a ray buffer (parity / query entry).
//   whitted_kernel        the Whitted-style integrator ("2. WhittedStyle/renderer.cpp":21-157), one thread per pixel.
//   resolve_kernel        screen pixels + per-tile energy sums (renderer.cpp:119,127-129).
//
// Numerics: compiled with -ffp-contract=off; only IEEE + - * / sqrt, so results are bit-identical with a scalar
// CPU evaluation of the same expressions.  min/max follow the reference's std::min/std::max operand order.
// No MFMA: the path is pointer chasing + slab / Möller–Trumbore tests.
#include "alt_common.h"

namespace crt {

// ------------------------------------------------------------------------------------------------------------
// render_tiles_kernel<KIND, COUNT>: grid = (tiles owned by this ctx) x (64-frame windows of the launch), block = one wavefront (lane = frame).
// slab layout: float4 [window][tileLocal][pixel 0..255][sample 0..64*passes), sample = frame*passes + pass (a partial last window leaves the
// tail of its rows unused).  Pixel-major: the lanes of a wave are the consecutive samples of a pixel, so lanes that finish the same pixel
// write neighbouring 16-byte pieces of one row and the L2 merges them into whole lines before they leave for HBM; accumulate_kernel
// transposes through LDS on the read side.
//
// Per-lane traversal state is ONE packed reference `cur` (layout.h), the 64-byte record it names — already
// PRE-LOADED into registers q0..q3 by the trip that produced it — and a stack in a per-lane LDS column:
//     cur == 0 (done)      -> SHADE phase: shade the hit with the pre-loaded ShadeTri (or end the path: unwind, write the
//                             sample, generate the next pixel's primary ray), start FindNearest for the new ray (quad,
//                             plane, first traversal step from the root's child pair in the kernel arguments)
//     BVH interior         -> NODE phase: two slab tests on the pre-loaded NodePair, ordered descend / push / pop
//     BVH leaf             -> TRI phase: ONE Möller–Trumbore test on the pre-loaded LeafTri, next triangle or pop
//     TLAS interior        -> NODE phase as well (two-level scenes): its two child TlasNodes are fetched into the NodePair layout
//     TLAS leaf            -> TLAS phase: enter the BLAS through the pre-loaded invT rows (return marker on the stack)
// One trip of the wave's loop = ballot the states, run each phase that has enough lanes (thresholds below: SHADE waits
// for 24 lanes unless nothing else can run, the others run whenever populated), then issue the record loads: four 16-byte loads at
// `geom + 32-bit offset`.  The loads fly while the next trip's ballots and the other phases' arithmetic execute, and a
// lane only ever pays for its own ray's length, never for the longest ray in the wave.
// ------------------------------------------------------------------------------------------------------------
#ifndef CRT_SHADE_BATCH
#define CRT_SHADE_BATCH 24          // lanes the SHADE phase waits for in a launch of few windows (latency matters: its heaviest tile ends it)
#endif
#ifndef CRT_SHADE_BATCH_JOB
#define CRT_SHADE_BATCH_JOB 40      // ... and in a many-window job, where fuller SHADE runs buy throughput (measured: -2.7 % per window, +7 % latency)
#endif
#ifndef CRT_TRI_BATCH
#define CRT_TRI_BATCH 1
#endif
#ifndef CRT_NODE_BATCH
#define CRT_NODE_BATCH 1
#endif

#ifndef CRT_TILES_NODE_STEPS
#define CRT_TILES_NODE_STEPS 2     // NODE steps per trip of the lanes that stay at interior nodes
#endif
#ifndef CRT_MIN_WAVES
#define CRT_MIN_WAVES 5      // waves per SIMD the register budget must allow (<= 96 VGPRs)
#endif
template <int KIND, bool COUNT>
__global__ __launch_bounds__(64, CRT_MIN_WAVES) void render_tiles_kernel(const Scene sc, float4* __restrict__ slab,
                                                           Counters* __restrict__ counters, unsigned long long* __restrict__ tileClocks,
                                                           const uint32_t* __restrict__ tileOrder,
                                                           uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                                           uint32_t sppFirst, uint32_t frames, uint32_t passes, uint32_t windows,
                                                           const uint32_t* __restrict__ blockDesc, uint32_t* __restrict__ tileCost, uint32_t rankCount, unsigned long long* __restrict__ launchClk)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    const unsigned long long clk0 = (COUNT || tileCost) ? wall_clock64() : 0ull;
    // a job that measures: when did the first and the last wavefront of the launch start (the host derives the launch's machine time from it, abi.cpp adopt_job_costs)
    if (launchClk && lane == 0) { atomicMax(&launchClk[0], ~clk0); atomicMax(&launchClk[1], clk0); }
    // block -> tile.  The kernel's duration is set by its most expensive tiles (one serial RNG stream per lane), so the host
    // lists the tiles whose pixels can see the meshes FIRST (tileOrder): the dispatcher starts them first and they are dealt
    // round-robin over the 8 XCDs / 256 CUs instead of piling up on the XCDs that own the image rows of the model.  The
    // geometry fits every XCD's L2, so nothing is lost by not giving an XCD a contiguous run of tiles (measured: a
    // contiguous-per-XCD mapping is 7-14 % slower).  Speed only; any bijection gives the same image.
    // One launch covers `windows` consecutive 64-frame windows of the progressive render (block = (tile rank, window), rank-major):
    // all windows of the expensive tiles are dispatched first and the cheap tiles fill the machine behind them, so a long job is
    // one grid whose duration is total work / machine throughput instead of a sequence of launches that each end on their
    // heaviest tile.
    // Latency mode (a launch of ONE window with a block table): the launch ends on its most expensive tile — one serial stream per lane, and every
    // trip of that wave pays for all the phases its 64 lanes populate (NODE and TRI nearly always; a lone wave issues one instruction per 6 - 9 cycles
    // whatever its lane count, tools/microbench/lone_wave.hip).  The host therefore measures the tiles (tileCost, below) and gives the expensive ones
    // to SEVERAL wavefronts of fewer lanes each: fewer phases are populated per trip, so each stream's serial chain advances faster, and the idle part of the
    // chip takes the extra wavefronts.  blockDesc[block] = local tile index | first frame << 16 | log2(lanes) << 22 | window << 25; blocks are listed most expensive tile first.  (A job of
    // several windows uses the same table form for its most expensive tiles, abi.cpp build_job_table.)
    uint32_t rank, win = 0u, laneBase = 0u, myLanes = 64u;
    if (blockDesc) { const uint32_t d = blockDesc[blockIdx.x]; rank = d & 0xffffu; laneBase = (d >> 16) & 63u; myLanes = 1u << ((d >> 22) & 7u); win = d >> 25; }
    else { rank = blockIdx.x / windows; win = blockIdx.x - rank * windows; }
    if (rank >= (blockDesc ? tileCount : rankCount)) return;                        // rankCount <= tileCount: only the first tiles of the order (a split job, abi.cpp)
    const uint32_t tl = blockDesc ? rank : (tileOrder ? tileOrder[rank] : rank);       // (a block table names local tile indices itself)
    sppFirst += win * 64u * passes;
    frames = (frames - win * 64u < 64u) ? frames - win * 64u : 64u;               // frames of THIS window (the last one may be partial)
    if (blockDesc) frames = frames > laneBase ? ((frames - laneBase < myLanes) ? frames - laneBase : myLanes) : 0u;
    slab += (size_t)win * ((size_t)tileCount * 256u * 64u * passes);              // this window's region of the sample slab
    const uint32_t tile = tileFirst + tl * tileStride;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    uint32_t* stk = lds + lane;
    const char* __restrict__ geom = sc.geom;

    Cnt cn; cn.rays = cn.primary = cn.interior = cn.leaf = cn.tri = cn.tlas = cn.visits = cn.meshhits = 0;
    uint32_t trips = 0;
    const bool narrowWave = myLanes < 64u;
    const int shadeBatch = narrowWave ? (int)((myLanes * 3u + 7u) / 8u) : (windows >= 8u ? CRT_SHADE_BATCH_JOB : CRT_SHADE_BATCH);   // (24 of 64 lanes, scaled to a narrow wavefront)
#ifdef CRT_STAMPS
    // diagnostic build (-DCRT_STAMPS): shader-clock time per phase of this wave; never compiled into the product
    unsigned long long stT[6] = {0, 0, 0, 0, 0, 0}; uint32_t stN[4] = {0, 0, 0, 0}; uint32_t stL[4] = {0, 0, 0, 0};
#define CRT_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define CRT_STAMP(var)
#endif
    const uint32_t items = 256u * passes;                                     // (pixel, pass) pairs in stream order
    const f3 nil3 = mk3(0.0f, 0.0f, 0.0f);                                    // placeholder of values no lane reads (the camera is read where a primary ray is generated)

    bool live = lane < frames;
    uint64_t liveMask = __builtin_amdgcn_ballot_w64(live);                    // lanes whose stream still has pixels to render
    uint32_t seed = init_seed(tx + ty * (uint32_t)sc.W + (sppFirst + (laneBase + lane) * passes) * 1799u);   // renderer.cpp:120
    uint32_t item = 0;
    // world-space ray of the current path segment, its nearest hit so far, path state
    f3 O = nil3, D = nil3, rD = nil3; bool inside = false; int depth = 0;
    Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
    // throughput factors of depths 0..4 (albedo*medium*... of each bounce, multiplied on unwind): written once per bounce and read once
    // per path, so they live in a per-lane LDS column behind the traversal stack (component j of depth k at fst[(3k + j) * 64]) instead
    // of 15 registers — the kernel then fits 5 waves per SIMD — and a bounce stores through a computed address instead of a select chain
    float* fst = reinterpret_cast<float*>(lds + sc.stackDepth * 64u + lane);
    // traversal state; (tO, tD, trD) = ray in the space of the structure being walked (object space inside a BLAS)
    uint32_t cur = kRefDone, sp = 0;
    f3 tO = nil3, tD = nil3, trD = nil3;
    bool rayFinite = true;                                                    // all of trD finite -> v_min/v_max slab test is exact
    bool fresh = true;                                                        // true: SHADE phase must generate a primary ray
    rec4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;                        // pre-loaded record of `cur`

    // Traversal stack: one LDS column per lane, entry i at stk[i * 64]; `sp` = number of entries.  The hot phases are
    // written branch-free: the current top is read speculatively at the start of a phase (it arrives while the slab /
    // triangle arithmetic runs), a push always stores to slot sp (a dead store when the far child was missed) and
    // only the pointer moves under a select.
#define CRT_TOP() (stk[(sp ? sp - 1u : 0u) * 64u])

    for (;;) {
        // state ballots: the compares write their lane masks straight into scalar registers; `liveMask` (maintained below) removes
        // finished lanes with one scalar AND instead of a per-lane predicate round trip
        const uint64_t mDone = __builtin_amdgcn_ballot_w64(cur == kRefDone) & liveMask;
        // (two-level scenes: a TLAS interior node is walked by the NODE phase too — its two child TlasNodes arrive in q0..q3 in the NodePair
        // layout {lo, ref, hi, -} x 2, and the ordered descend / push / pop is the same code; only TLAS leaves need the TLAS phase)
        const uint64_t mNode = __builtin_amdgcn_ballot_w64(KIND == 1 ? (((cur >> 31) ^ (cur >> 30)) & 1u) != 0u : (cur & 0xC0000000u) == kRefInterior) & liveMask;
        const uint64_t mTri = __builtin_amdgcn_ballot_w64(cur != kRefDone && (cur & 0xC0000000u) == 0u) & liveMask;
        const uint64_t mTlas = (KIND == 1) ? (__builtin_amdgcn_ballot_w64((cur & 0xC0000000u) == kRefTlasLeaf) & liveMask) : 0ull;
        const bool isDone = live && cur == kRefDone;
        const bool isNode = live && (KIND == 1 ? (((cur >> 31) ^ (cur >> 30)) & 1u) != 0u : (cur & 0xC0000000u) == kRefInterior);
        const bool isTri = live && cur != kRefDone && (cur & 0xC0000000u) == 0u;
        const bool isTlas = (KIND == 1) && live && (cur & 0xC0000000u) == kRefTlasLeaf;
        const int nDone = __popcll(mDone), nNode = __popcll(mNode), nTri = __popcll(mTri);
        const int nTlas = (KIND == 1) ? __popcll(mTlas) : 0;
        if (nDone + nNode + nTri + nTlas == 0) break;
        if (COUNT) trips++;
        // a phase runs when enough lanes wait for it; when no phase reaches its batch size the most populated one runs
        const bool bigNode = nNode >= CRT_NODE_BATCH, bigTri = nTri >= CRT_TRI_BATCH, bigShade = nDone >= shadeBatch;
        const bool none = !bigNode && !bigTri && !bigShade && nTlas == 0;
        const bool runTlas = nTlas > 0;
        const bool runNode = bigNode || (none && nNode > 0 && nNode >= nTri && nNode >= nDone);
        const bool runTri = bigTri || (none && nTri > 0 && nTri > nNode && nTri >= nDone);
        const bool runShade = bigShade || (none && nDone > 0 && nDone > nNode && nDone > nTri);
        // The record loads issued at the end of the previous trip are first needed here.  Naming all four tuples in one
        // empty asm keeps the register allocator from splitting a loaded tuple across the back-edge (it otherwise copies one
        // component right behind the loads, which puts a vmcnt wait — the whole fetch latency — at the end of every trip).
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
#ifdef CRT_STAMPS
        CRT_STAMP(s0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CRT_STAMP(s1);
        stT[0] += s1 - s0;
        if (runShade) { stN[0]++; stL[0] += nDone; } if (runNode) { stN[1]++; stL[1] += nNode; } if (runTri) { stN[2]++; stL[2] += nTri; }
        stN[3]++;
#endif

        if (runShade && isDone) {
            // ---------------- SHADE / RAY-GEN phase: one step of Renderer::Sample ("3. PathTracer/renderer.cpp":50-100) ----------
            // Written as a sequence of stages that as many lanes as possible share, instead of one if/else tree per outcome: the
            // sky / floor / mesh texel fetch is ONE site, the direction normalisation of a diffuse bounce and of a new primary ray
            // is ONE site, and the new ray's reciprocal direction, quad / plane tests and root step are ONE site.  Every lane
            // still evaluates exactly the reference's expressions in the reference's order.
            // stage 1: what did FindNearest return (renderer.cpp:52-55, 69)
            // (Scene fields only this phase needs are read from the kernel-argument segment here instead of living in scalar registers across the loop: dev_common.h scene_floats)
            const kernarg_f cam = scene_floats(offsetof(Scene, camPos));          // camPos, topLeft, topRight, bottomLeft, invW, invH
            const f3 camPos = mk3(cam[0], cam[1], cam[2]);
            const bool first = fresh;                                             // no path yet: only generate the first primary ray
            const bool miss = !first && h.objIdx == -1;                           // -> GetSkyColor
            const bool stop = !first && h.objIdx != -1 && (depth >= sc.depthLimit || h.objIdx == 0);   // depth limit -> 0, light -> (24,24,22)
            const bool surf = !first && !miss && !stop;                           // floor or mesh: the path bounces
            if (!first && h.objIdx >= 2) cn.meshhits++;
            // stage 2: texture coordinates + surface data
            const kernarg_f sk = scene_floats(offsetof(Scene, skyOffset));        // skyOffset, skyW, skyH
            float tu = 0, tv = 0; uint32_t tOff = asu(sk[0]); int tW = (int)asu(sk[1]), tH = (int)asu(sk[2]);
            f3 I = O, N = O, absorb = O; float refl = 0, refr = 0;
            if (miss) {                                                           // GetSkyColor, file_scene.cpp:142-154
                const float phi = crt_atan2f(-D.z, D.x) + CRT_PI, theta = crt_acosf(-D.y);
                tu = phi * CRT_INV2PI; tv = theta * CRT_INVPI;
            }
            if (surf) {
                I = O + h.t * D;
                if (h.objIdx == 1) {                                              // floor: Plane::GetNormal / GetUV (primitives.h:112-133)
                    const kernarg_f fl = scene_floats(offsetof(Scene, floorN));     // floorN[3], floorD, floorInvto
                    const kernarg_f fm = scene_floats(offsetof(Scene, floorMat));   // Material: reflectivity, refractivity, absorption[3], texOffset, texW, texH
                    N = mk3(fl[0], fl[1], fl[2]);
                    if (N.y == 1) {
                        float u = I.x, v = I.z;
                        u *= fl[4]; v *= fl[4];
                        tu = u - __builtin_floorf(u); tv = v - __builtin_floorf(v);
                    }
                    refl = fm[0]; refr = fm[1];
                    absorb = mk3(fm[2], fm[3], fm[4]);
                    tOff = asu(fm[5]); tW = (int)asu(fm[6]); tH = (int)asu(fm[7]);
                } else {                                                          // mesh: GetNormal / GetUV (bvh.cpp:290-305, blas_bvh.cpp:391-406)
                    const f3 n0 = mk3(q0.x, q0.y, q0.z), n1 = mk3(q0.w, q1.x, q1.y), n2 = mk3(q1.z, q1.w, q2.x);   // (q0..q3) = the hit's ShadeTri
                    const float w = 1 - h.u - h.v;
                    const f3 Nn = w * n0 + h.u * n1 + h.v * n2;
                    tu = w * q2.y + h.u * q2.w + h.v * q3.y;
                    tv = w * q2.z + h.u * q3.x + h.v * q3.z;
                    const rec4* mp = reinterpret_cast<const rec4*>(sc.mats + (int)asu(q3.w));
                    const rec4 m0 = mp[0], m1 = mp[1];
                    refl = m0.x; refr = m0.y; absorb = mk3(m0.z, m0.w, m1.x);
                    tOff = asu(m1.y); tW = (int)asu(m1.z); tH = (int)asu(m1.w);
                    if (KIND == 0) {
                        N = normalize3(Nn);
                    } else {
                        const uint32_t io = sc.instOff + (uint32_t)(h.objIdx - 2) * 128u + 64u;   // Instance::T rows
                        const rec4 r0 = ldg(geom, io), r1 = ldg(geom, io + 16), r2 = ldg(geom, io + 32);
                        N = normalize3(mk3(r0.x * Nn.x + r0.y * Nn.y + r0.z * Nn.z + r0.w * 0.0f,
                                           r1.x * Nn.x + r1.y * Nn.y + r1.z * Nn.z + r1.w * 0.0f,
                                           r2.x * Nn.x + r2.y * Nn.y + r2.z * Nn.z + r2.w * 0.0f));
                    }
                }
                if (dot3(N, D) > 0) N = -N;
            }
            // stage 3: the one texel fetch (Texture::Sample) — sky colour of a miss, albedo of a textured surface
            f3 c = mk3(1.0f, 1.0f, 1.0f);
            if (miss || (surf && tW > 0)) c = tex_sample(sc, tOff, tW, tH, tu, tv);
            // stage 4a: the bounce (renderer.cpp:76-99).  `v` = outgoing direction (a diffuse one still to be normalised), `pre` = the
            // throughput factor without the cosine term of the diffuse branch
            f3 v = O, pre = O; bool norm = false, diffuse = false, newInside = false;
            if (surf) {
                f3 medium = mk3(1, 1, 1);
                if (inside) {
                    const f3 ab = absorb * -h.t;
                    medium = mk3(crt_expf(ab.x), crt_expf(ab.y), crt_expf(ab.z));
                }
                const float r = rnd(seed);
                if (r < refl) {                                                   // HandleMirror, renderer.cpp:20-25
                    v = D - 2.0f * N * dot3(N, D);
                    pre = c * medium;
                } else if (r < refl + refr) {                                     // HandleDielectric, renderer.cpp:27-45
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
                    pre = c * medium;
                } else {                                                          // diffuse, renderer.cpp:93-99; diffusereflection tmplmath.h:535-544
                    f3 Rr;
                    do {
                        const float rz = rnd_pm1(seed);                           // draw order pinned z, y, x
                        const float ry = rnd_pm1(seed);
                        const float rx = rnd_pm1(seed);
                        Rr = mk3(rx, ry, rz);
                    } while (dot3(Rr, Rr) > 1);
                    if (dot3(Rr, N) < 0) Rr = Rr * -1.0f;
                    v = Rr; norm = true; diffuse = true;
                    const f3 brdf = c * CRT_INVPI;
                    pre = medium * brdf * 2.0f * CRT_PI;                          // ... * dot(R, N) once R is normalised (stage 5)
                }
            }
            // stage 4b: the path ended: unwind the recursion (innermost factor first: albedo*medium*Sample(...) multiplies on return),
            // store the sample, move on to the stream's next pixel
            bool gen = first;
            if (miss || stop) {
                f3 L = miss ? c : ((depth >= sc.depthLimit) ? mk3(0, 0, 0) : mk3(24, 24, 22));   // GetLightColor, file_scene.cpp:164-167
                // most paths end at depth 0..2: a level is read (and multiplied) only when some lane of the wave is that deep
#pragma unroll
                for (int k = 4; k >= 0; k--)
                    if (depth > k) L = mk3(fst[(3 * k) * 64], fst[(3 * k + 1) * 64], fst[(3 * k + 2) * 64]) * L;
                uint32_t pix = item, pass = 0;
                if (passes != 1u) { pix = item / passes; pass = item - pix * passes; }
                slab[((size_t)tl * 256u + pix) * (64u * passes) + ((laneBase + lane) * passes + pass)] = make_float4(L.x, L.y, L.z, 0.0f);
                item++;
                gen = true;
                if (item >= items) { live = false; gen = false; }
            }
            if (gen) {                                                            // ProcessTile + Camera::GetPrimaryRay (renderer.cpp:125-126, camera.h:23-30)
                const uint32_t pix = (passes == 1u) ? item : item / passes;
                const int x = (int)(tx * 16u + (pix & 15u)), y = (int)(ty * 16u + (pix >> 4));
                const float jy = rnd(seed);                                       // pinned: first draw is the y jitter
                const float jx = rnd(seed);
                const f3 TL = mk3(cam[3], cam[4], cam[5]), TR = mk3(cam[6], cam[7], cam[8]), BL = mk3(cam[9], cam[10], cam[11]);
                const float u = ((float)x + jx) * cam[12], vv = ((float)y + jy) * cam[13];
                const f3 P = TL + u * (TR - TL) + vv * (BL - TL);
                v = P - camPos; norm = true;
                inside = false; depth = 0; fresh = false;
                cn.primary++;
            }
            // stage 5: the new ray (bounce or primary) and the start of its scene.FindNearest: light quad, floor plane, root step
            if (live) {
                const float inv = rcp_exact(__builtin_sqrtf(dot3(v, v)));         // normalize(): v * (1 / sqrtf(dot(v, v)))
                const f3 nv = norm ? v * inv : v;
                if (surf) {
                    const f3 factor = diffuse ? pre * dot3(nv, N) : pre;
                    float* fd = fst + depth * 192;                                // depth <= 4 here: a surface hit at depth >= depthLimit (<= 5) ended the path
                    fd[0] = factor.x; fd[64] = factor.y; fd[128] = factor.z;
                    depth++;
                    O = I + nv * CRT_EPS; inside = newInside;
                } else O = camPos;
                D = nv; rD = rcp_exact3(nv);
                cn.rays++;
                h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
                {
                    const kernarg_f lp = scene_floats(offsetof(Scene, lightInvT));   // lightInvT[12] lightNrm[3] lightSize lightPos[3] floorN[3] floorD
                    const kernarg_f ax = scene_floats(offsetof(Scene, lightAxis));   // lightAxis, floorAxisY
                    LightFloor lf;
#pragma unroll
                    for (int i = 0; i < 12; i++) lf.lightInvT[i] = lp[i];
                    lf.lightSize = lp[15]; lf.floorN[0] = lp[19]; lf.floorN[1] = lp[20]; lf.floorN[2] = lp[21]; lf.floorD = lp[22];
                    lf.lightAxis = asu(ax[0]); lf.floorAxisY = asu(ax[1]);
                    hit_light_floor(lf, O, D, h);
                }
                tO = O; tD = D; trD = rD; rayFinite = finite3(rD);
                sp = 0;
                if (sc.rootIsPair) {
                    // the root's two children travel in the kernel arguments (scalar registers): the first traversal step
                    // (bvh.cpp:244-257 / tlas_bvh.cpp:96-110 at the root, empty stack) happens here, at this phase's lane
                    // density, and a ray that misses both boxes never leaves the SHADE state
                    const kernarg_f rp = scene_floats(offsetof(Scene, rootPair));
                    const rec4 a0 = {rp[0], rp[1], rp[2], rp[3]}, a1 = {rp[4], rp[5], rp[6], rp[7]};
                    const rec4 b0 = {rp[8], rp[9], rp[10], rp[11]}, b1 = {rp[12], rp[13], rp[14], rp[15]};
                    float d1, d2;
                    if (__builtin_amdgcn_ballot_w64(!rayFinite) == 0ull) { d1 = box_fast(a0, a1, O, rD, h.t); d2 = box_fast(b0, b1, O, rD, h.t); }
                    else { d1 = box_exact(a0, a1, O, rD, h.t); d2 = box_exact(b0, b1, O, rD, h.t); }
                    const bool sw = d1 > d2;
                    const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
                    const uint32_t rn = sw ? asu(b0.w) : asu(a0.w), rf = sw ? asu(a0.w) : asu(b0.w);
                    stk[0] = rf;                                                  // dead store unless pushed
                    const bool hitN = dn != 1e30f;
                    sp = (hitN && df != 1e30f) ? 1u : 0u;
                    cur = hitN ? rn : kRefDone;
                    if (COUNT) { if (KIND == 0) cn.interior++; else cn.tlas++; }
                } else cur = sc.rootRef;
                if (COUNT && KIND == 0 && (cur & 0xC0000000u) == 0u && cur != kRefDone) cn.leaf++;
            }
        }
        if (runShade) liveMask = __builtin_amdgcn_ballot_w64(live);             // streams end only in the SHADE phase
#ifdef CRT_STAMPS
        CRT_STAMP(s2); stT[1] += s2 - s1;
#endif
        if (KIND == 1 && runTlas && isTlas) {
            // ---------------- TLAS phase: a TLAS leaf (infra/tlas_bvh.cpp:91-95) -> enter the BLAS (BLASBVH::Intersect, blas_bvh.cpp:376-381):
            // object-space ray through the pre-loaded invT rows, return marker on the stack, BLAS root
            if (COUNT) { cn.tlas++; cn.visits++; }
            to_object_space(q0, q1, q2, O, D, tO, tD, trD);
            rayFinite = finite3(trD);
            stk[sp * 64u] = kRefReturn; sp++;
            const uint32_t next = asu(q3.z);                                      // Instance::rootRef
            if (COUNT && (next & 0xC0000000u) == 0u && next != kRefDone) cn.leaf++;
            cur = next;
        }
#ifdef CRT_STAMPS
        CRT_STAMP(s3);
#endif
        if (runNode) {
            // ---------------- NODE phase (infra/bvh.cpp:244-257) -------------------------------------------------
            const bool allFinite = __builtin_amdgcn_ballot_w64(isNode && !rayFinite) == 0ull;
            if (isNode) {
                if (COUNT) { if (KIND == 1 && (cur & kRefTlasBit) != 0u) cn.tlas++; else cn.interior++; }
                uint32_t top = CRT_TOP();                                            // speculative: lands during the slab arithmetic
                float d1, d2;
                if (allFinite) { d1 = box_fast(q0, q1, tO, trD, h.t); d2 = box_fast(q2, q3, tO, trD, h.t); }
                else { d1 = box_exact(q0, q1, tO, trD, h.t); d2 = box_exact(q2, q3, tO, trD, h.t); }
                const bool sw = d1 > d2;                                             // near child first (strict >: ties keep child 1)
                const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
                const uint32_t rn = sw ? asu(q2.w) : asu(q0.w), rf = sw ? asu(q0.w) : asu(q2.w);
                stk[sp * 64u] = rf;                                                  // dead store unless `push`
                const bool hitN = dn != 1e30f, push = hitN && df != 1e30f;
                bool pop = !hitN && sp != 0u;
                uint32_t next = hitN ? rn : (pop ? top : kRefDone);
                sp = sp + (push ? 1u : 0u) - (pop ? 1u : 0u);
                if (KIND == 1 && next == kRefReturn) {                               // BLAS finished: back to the world-space ray, pop the TLAS entry below
                    tO = O; tD = D; trD = rD; rayFinite = finite3(rD);
                    pop = sp != 0u; top = CRT_TOP();
                    next = pop ? top : kRefDone; sp -= pop ? 1u : 0u;
                }
                if (COUNT && (next & 0xC0000000u) == 0u && next != kRefDone) cn.leaf++;
                cur = next;
            }
        }
        // ---------------- a second NODE step in the same trip (while-while, round 3): the lanes that just stepped and are still at an interior node fetch their next
        // pair now and step again, so the trip's fixed part (state ballots, phase selection, the other phases' skeleton) is paid once per two node steps; a lane that
        // just reached a leaf fetches its first triangle with them and joins THIS trip's TRI phase.  Same steps in the same order per lane.
        bool triNow = isTri;
#if CRT_TILES_NODE_STEPS > 1
        if (runNode) {
            const bool node2 = isNode && (KIND == 1 ? (((cur >> 31) ^ (cur >> 30)) & 1u) != 0u : (cur & 0xC0000000u) == kRefInterior);
            const bool leaf2 = isNode && cur != kRefDone && (cur & 0xC0000000u) == 0u;
            const uint64_t mNode2 = __builtin_amdgcn_ballot_w64(node2);
            if (mNode2 != 0ull) {
                if (node2 || leaf2) {
                    uint32_t oa = (cur & kRefOffsetMask) << 4, ob = oa + 32u;                  // NodePair / LeafTri
                    if (KIND == 1 && (cur & kRefTlasBit) != 0u) { oa = sc.tlasOff + (cur & 0x7fffu) * 32u; ob = sc.tlasOff + ((cur >> 15) & 0x7fffu) * 32u; }   // TLAS interior: the two child nodes
                    q0 = ldg(geom, oa); q1 = ldg(geom, oa + 16u); q2 = ldg(geom, ob); q3 = ldg(geom, ob + 16u);
                }
                triNow = isTri || leaf2;
                const bool allFinite2 = __builtin_amdgcn_ballot_w64(node2 && !rayFinite) == 0ull;
                if (node2) {
                    if (COUNT) { if (KIND == 1 && (cur & kRefTlasBit) != 0u) cn.tlas++; else cn.interior++; }
                    uint32_t top = CRT_TOP();
                    float d1, d2;
                    if (allFinite2) { d1 = box_fast(q0, q1, tO, trD, h.t); d2 = box_fast(q2, q3, tO, trD, h.t); }
                    else { d1 = box_exact(q0, q1, tO, trD, h.t); d2 = box_exact(q2, q3, tO, trD, h.t); }
                    const bool sw = d1 > d2;
                    const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
                    const uint32_t rn = sw ? asu(q2.w) : asu(q0.w), rf = sw ? asu(q0.w) : asu(q2.w);
                    stk[sp * 64u] = rf;
                    const bool hitN = dn != 1e30f, push = hitN && df != 1e30f;
                    bool pop = !hitN && sp != 0u;
                    uint32_t next = hitN ? rn : (pop ? top : kRefDone);
                    sp = sp + (push ? 1u : 0u) - (pop ? 1u : 0u);
                    if (KIND == 1 && next == kRefReturn) {
                        tO = O; tD = D; trD = rD; rayFinite = finite3(rD);
                        pop = sp != 0u; top = CRT_TOP();
                        next = pop ? top : kRefDone; sp -= pop ? 1u : 0u;
                    }
                    if (COUNT && (next & 0xC0000000u) == 0u && next != kRefDone) cn.leaf++;
                    cur = next;
                }
            }
        }
#endif
#ifdef CRT_STAMPS
        CRT_STAMP(s4); stT[2] += s4 - s3;
#endif
        if ((runTri || triNow != isTri) && triNow) {
            // ---------------- TRI phase: one triangle of the current leaf (infra/bvh.cpp:203-222, 232-243) -----------
            if (COUNT) cn.tri++;
            uint32_t top = CRT_TOP();                                                // speculative, as in the NODE phase
            hit_tri(q0, q1, q2, tO, tD, h);
            const bool more = asu(q2.w) > 1u;                                        // next LeafTri of this leaf (48 B = 3 units)
            bool pop = !more && sp != 0u;
            uint32_t next = more ? cur + 3u : (pop ? top : kRefDone);
            sp -= pop ? 1u : 0u;
            if (KIND == 1 && next == kRefReturn) {
                tO = O; tD = D; trD = rD; rayFinite = finite3(rD);
                pop = sp != 0u; top = CRT_TOP();
                next = pop ? top : kRefDone; sp -= pop ? 1u : 0u;
            }
            if (COUNT && !more && (next & 0xC0000000u) == 0u && next != kRefDone) cn.leaf++;
            cur = next;
        }
#ifdef CRT_STAMPS
        CRT_STAMP(s5); stT[3] += s5 - s4;
#endif
        // ---------------- issue the record loads of every lane that moved (consumed by a later trip) ---------------
        {
            const bool doneNow = cur == kRefDone;
            uint32_t oa = doneNow ? sc.shadeOff + (uint32_t)h.triIdx * 64u : (cur & kRefOffsetMask) << 4;   // ShadeTri of the hit | NodePair / LeafTri
            uint32_t ob = oa + 32u;
            if (KIND == 1 && !doneNow && (cur & kRefTlasBit) != 0u) {
                if (cur & kRefInterior) { oa = sc.instOff + (cur & 0xffffu) * 128u; ob = oa + 32u; }                // TLAS leaf: Instance {invT rows, ids}
                else { oa = sc.tlasOff + (cur & 0x7fffu) * 32u; ob = sc.tlasOff + ((cur >> 15) & 0x7fffu) * 32u; } // TLAS interior: the two child nodes
            }
            // unconditional per lane on purpose: a lane that did not move re-fetches its record (an L1 hit) and a lane with nothing to
            // fetch reads record 0, so q0..q3 are plain loop-carried load results and nothing has to wait for them here.  Only when NO
            // lane of the wave has a record to fetch (tiles that see only sky and floor: every ray ends at the root step) the loads —
            // and the memory round trip the next trip would wait for — are skipped altogether.
            const bool nothing = !live || (doneNow && h.objIdx < 2);
            if (nothing) { oa = 0u; ob = 32u; }
            if (__builtin_amdgcn_ballot_w64(!nothing) != 0ull) { q0 = ldg(geom, oa); q1 = ldg(geom, oa + 16u); q2 = ldg(geom, ob); q3 = ldg(geom, ob + 16u); }
        }
#ifdef CRT_STAMPS
        CRT_STAMP(s6); stT[4] += s6 - s5; stT[5] += s6 - s0;
#endif
    }
#undef CRT_TOP

    // what this tile cost (100 MHz wall clock ticks; the longest of its wavefronts): the host's latency mode sizes the next launch's wavefronts with it
    if (tileCost && lane == 0) atomicMax(&tileCost[tl], (uint32_t)(wall_clock64() - clk0));
    // ... and how much of this wavefront ran after the launch's last wavefront had started (the launch's drain: abi.cpp adopt_job_costs)
    if (launchClk && lane == 0) { const unsigned long long now = wall_clock64(), last = __hip_atomic_load(&launchClk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), from = last > clk0 ? last : clk0; if (now > from) atomicAdd(&launchClk[2], now - from); }
    if (COUNT && tileClocks && lane == 0 && windows == 1u) {                 // instrumentation build only: per-tile wall time + loop trips
        tileClocks[2 * tl] = wall_clock64() - clk0;        // 100 MHz constant clock
        tileClocks[2 * tl + 1] = trips;
#ifdef CRT_STAMPS
        unsigned long long* dbg = tileClocks + 2 * (size_t)tileCount + 16 * (size_t)tl;
        for (int i = 0; i < 6; i++) dbg[i] = stT[i];
        for (int i = 0; i < 4; i++) dbg[6 + i] = stN[i];
        for (int i = 0; i < 4; i++) dbg[10 + i] = stL[i];
#endif
    }
    // wave-level reduction of the counters, one atomic per counter per wave
    uint32_t vals[8] = {cn.rays, cn.primary, cn.interior, cn.leaf, cn.tri, cn.tlas, cn.visits, cn.meshhits};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (!COUNT && i >= 2 && i != 7) continue;
        uint32_t s = wave_sum(vals[i]);
        if (lane == 0 && s) atomicAdd(&counters->v[i], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------------------------
// accumulate_kernel: accumulator[pixel] += the slab's samples in (window, frame, pass) order — the reference's
// `accumulator[..] +=` order, renderer.cpp:124.  The slab is pixel-major ([tile][pixel][sample]); this kernel streams it with
// fully coalesced 1 KB loads and transposes through LDS:
//   block = one owned tile, 4 wavefronts; wavefront w owns the tile's image rows 4w .. 4w+3 (16 pixels each) and never talks to the others;
//   per (row, window, block of 64 samples): lane l fetches sample l of each of the row's 16 pixels (16 loads of 1 KB per wave in flight),
//   parks them in the wave's LDS region, then lane (pixel j = l / 4, channel = l % 4) adds its 64 values IN ORDER to its running sum.
// The running sums of a row are 64 consecutive floats of the accumulator (one 256-byte line).  Every addition of a channel happens
// in one lane, in frame order; no float atomics anywhere.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kAccRowF4 = 65u;                  // float4 per pixel row in LDS: 64 samples + 1 pad (bank spread for the channel-wise reads)
constexpr uint32_t kAccWaveLds = 16u * kAccRowF4 * 16u;   // bytes per wavefront
__global__ __launch_bounds__(256) void accumulate_kernel(const float4* __restrict__ slab, float* __restrict__ acc,
                                                          uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                                          uint32_t W, uint32_t frames, uint32_t passes)
{
    extern __shared__ float4 accLds[];
    const uint32_t tl = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (tl >= tileCount) return;
    const uint32_t tile = tileFirst + tl * tileStride;
    const uint32_t x0 = (tile % tilesX) * 16u, y0 = (tile / tilesX) * 16u;
    const uint32_t rowLen = 64u * passes;                                      // float4 per pixel row of the slab
    const size_t winStride = (size_t)tileCount * 256u * rowLen;
    float4* my4 = accLds + wave * (16u * kAccRowF4);
    const float* myF = reinterpret_cast<const float*>(my4) + (lane >> 2) * (kAccRowF4 * 4u) + (lane & 3u);
    for (uint32_t r = 0; r < 4u; r++) {
        const uint32_t v = wave * 4u + r;                                      // row of the tile
        float* __restrict__ ap = acc + ((size_t)(x0 + (size_t)(y0 + v) * W)) * 4u + lane;   // 16 pixels x 4 channels = 64 consecutive floats
        float a = *ap;
        const float4* __restrict__ rowBase = slab + ((size_t)tl * 256u + v * 16u) * rowLen;
        for (uint32_t f0 = 0; f0 < frames; f0 += 64u, rowBase += winStride) {
            const uint32_t S = ((frames - f0 < 64u) ? frames - f0 : 64u) * passes;   // valid samples of this window's rows
            for (uint32_t b = 0; b < S; b += 64u) {
                const uint32_t n = (S - b < 64u) ? S - b : 64u;
                float4 t[16];
                if (lane < n) {
#pragma unroll
                    for (int j = 0; j < 16; j++) t[j] = rowBase[(size_t)j * rowLen + b + lane];
#pragma unroll
                    for (int j = 0; j < 16; j++) my4[j * kAccRowF4 + lane] = t[j];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // wave-private region: LDS executes a wave's operations in order
                uint32_t s = 0;
                for (; s + 8u <= n; s += 8u) {
                    float x[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) x[k] = myF[(s + k) * 4u];
#pragma unroll
                    for (int k = 0; k < 8; k++) a += x[k];
                }
                for (; s < n; s++) a += myF[s * 4u];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        *ap = a;
    }
}

// ------------------------------------------------------------------------------------------------------------
// find_nearest_kernel: scene.FindNearest for a buffer of rays in the reference's traversal order (reports Ray::traversed / tested as bvh.cpp:224-258 and
// tlas_bvh.cpp:83-111 count their loop trips).  Persistent-wave form (round 3; before: one ray per lane for the whole launch, 19 % of the lanes busy): a wavefront
// draws rays from a launch-wide cursor, a lane whose ray is finished takes the next one as soon as a quarter of the wavefront is idle, and a trip of the loop runs
// each kind of step once for the lanes that are at it — TLAS node / leaf, BVH node, ONE triangle of a leaf — exactly the steps of find_nearest_seq (dev_common.h),
// which the render kernels' sequential forms still call.
// ------------------------------------------------------------------------------------------------------------
struct RayIn { float O[3]; float D[3]; int32_t inside; };
struct HitOut { float t, u, v; int32_t objIdx, triIdx, traversed, tested; };
constexpr uint32_t kQueryRefill = 16u;                                   // idle lanes that trigger the next draw from the cursor

__global__ __launch_bounds__(64) void find_nearest_kernel(const Scene sc, const RayIn* __restrict__ rays,
                                                           HitOut* __restrict__ hits, uint32_t n, Counters* __restrict__ counters, uint32_t* __restrict__ cursor)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    const char* __restrict__ g = sc.geom;
    uint32_t* stk = lds + lane;                                           // BVH entries of this lane's column: entry i at [i * 64]
    uint32_t* tstk = stk + sc.bvhStack * 64;                              // TLAS entries above them
    Cnt cn; cn.rays = cn.primary = cn.interior = cn.leaf = cn.tri = cn.tlas = cn.visits = cn.meshhits = 0;
    // the ray in this lane
    uint32_t mode = 0u;                                                   // 0 idle, 1 at a BVH reference (`cur`), 2 inside a leaf (`leafAt`), 3 at a TLAS reference (`tcur`), 4 finished
    uint32_t idx = 0; f3 O = mk3(0, 0, 0), D = O, rD = O, Oo = O, Do = O, rDo = O;   // world-space ray; the ray the BVH is walked with (object space inside a BLAS)
    Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
    int traversed = 0, tested = 0;
    uint32_t cur = 0, sp = 0, tcur = 0, tsp = 0, leafAt = 0;
    bool more = true;                                                     // wave-uniform: the cursor has rays left
    for (;;) {
        // ---------------- refill: idle lanes draw the next rays ----------------
        const uint64_t mIdle = __builtin_amdgcn_ballot_w64(mode == 0u);
        const uint32_t nIdle = (uint32_t)__popcll(mIdle);
        if (more && nIdle >= kQueryRefill) {
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
                cn.rays++;
                hit_light_floor(sc, O, D, h);
                if (sc.kind == 0) { Oo = O; Do = D; rDo = rD; cur = sc.rootRef; sp = 0; mode = 1u; }
                else { tcur = sc.rootRef; tsp = 0; mode = 3u; }
            }
        }
        if (__builtin_amdgcn_ballot_w64(mode != 0u) == 0ull) break;       // nothing in flight, nothing left to draw
        // ---------------- TLAS step (tlas_bvh.cpp:83-111): a leaf enters its BLAS, an interior node orders its children ----------------
        if (mode == 3u) {
            traversed++; cn.tlas++;
            bool pop = false;
            if ((tcur & kRefTlasLeaf) == kRefTlasLeaf) {
                cn.visits++;
                const uint32_t io = sc.instOff + (tcur & 0xffffu) * 128u;
                const rec4 r0 = ldg(g, io), r1 = ldg(g, io + 16), r2 = ldg(g, io + 32), ids = ldg(g, io + 48);
                to_object_space(r0, r1, r2, O, D, Oo, Do, rDo);
                cur = asu(ids.z); sp = 0; mode = 1u;
            } else {
                const uint32_t o1 = sc.tlasOff + (tcur & 0x7fffu) * 32u, o2 = sc.tlasOff + ((tcur >> 15) & 0x7fffu) * 32u;
                const rec4 alo = ldg(g, o1), ahi = ldg(g, o1 + 16), blo = ldg(g, o2), bhi = ldg(g, o2 + 16);
                float d1 = box_exact(alo, ahi, O, rD, h.t), d2 = box_exact(blo, bhi, O, rD, h.t);
                uint32_t r1 = asu(alo.w), r2 = asu(blo.w);
                if (d1 > d2) { float td = d1; d1 = d2; d2 = td; uint32_t tr = r1; r1 = r2; r2 = tr; }
                if (d1 == 1e30f) pop = true;
                else { tcur = r1; if (d2 != 1e30f) { tstk[tsp * 64] = r2; tsp++; } }
            }
            if (pop) { if (tsp == 0) mode = 4u; else tcur = tstk[(--tsp) * 64]; }
        }
        // ---------------- BVH step (bvh.cpp:224-258): an interior node orders its children, a leaf starts its triangle list ----------------
        bool back = false;                                                // this lane pops the BVH stack
        if (mode == 1u) {
            traversed++;
            const uint32_t off = (cur & kRefOffsetMask) << 4;
            if (cur & kRefInterior) {
                cn.interior++;
                const rec4 alo = ldg(g, off), ahi = ldg(g, off + 16), blo = ldg(g, off + 32), bhi = ldg(g, off + 48);
                float d1 = box_exact(alo, ahi, Oo, rDo, h.t), d2 = box_exact(blo, bhi, Oo, rDo, h.t);
                uint32_t r1 = asu(alo.w), r2 = asu(blo.w);
                if (d1 > d2) { float td = d1; d1 = d2; d2 = td; uint32_t tr = r1; r1 = r2; r2 = tr; }
                if (d1 == 1e30f) back = true;
                else { cur = r1; if (d2 != 1e30f) { stk[sp * 64] = r2; sp++; } }
            } else { cn.leaf++; leafAt = off; mode = 2u; }
        }
        // ---------------- one triangle of the leaf (bvh.cpp:232-243) ----------------
        if (mode == 2u) {
            const rec4 a = ldg(g, leafAt), b = ldg(g, leafAt + 16), c = ldg(g, leafAt + 32);
            tested++; cn.tri++;
            hit_tri(a, b, c, Oo, Do, h);
            if (asu(c.w) <= 1u) { back = true; mode = 1u; } else leafAt += 48;
        }
        if (back) {
            if (sp != 0) cur = stk[(--sp) * 64];
            else if (sc.kind == 0) mode = 4u;                              // the BVH is done
            else if (tsp == 0) mode = 4u;                                  // the BLAS is done: back to the TLAS loop, tlas_bvh.cpp:95
            else { tcur = tstk[(--tsp) * 64]; mode = 3u; }
        }
        if (mode == 4u) {                                                  // finished: the result record, and the lane is free
            if (h.objIdx >= 2) cn.meshhits++;
            int triIdx = h.triIdx;
            if (sc.kind != 0 && h.objIdx >= 2) triIdx -= (int)asu(ldg(g, sc.instOff + (uint32_t)(h.objIdx - 2) * 128u + 48u).x);   // - Instance::shadeBase
            HitOut o; o.t = h.t; o.u = h.u; o.v = h.v; o.objIdx = h.objIdx; o.triIdx = triIdx; o.traversed = traversed; o.tested = tested;
            hits[idx] = o;
            mode = 0u;
        }
    }
    uint32_t vals[8] = {cn.rays, cn.primary, cn.interior, cn.leaf, cn.tri, cn.tlas, cn.visits, cn.meshhits};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t s = wave_sum(vals[k]);
        if (lane == 0 && s) atomicAdd(&counters->v[k], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------------------------
// whitted_kernel: the reference's second front-end ("2. WhittedStyle/renderer.cpp":21-157) behind the same boundary.
// Deterministic (no RNG), so one thread per pixel.  Trace()'s recursion (a dielectric spawns a refracted AND a
// reflected ray) runs on an explicit frame stack in post-order, so every float sum and product happens in the
// reference's order:  return medium * (((0 + cA*Trace(a)) + cB*Trace(b)) + C).
// ------------------------------------------------------------------------------------------------------------
struct WFrame { f3 cA, cB, C, medium, out, bO, bD; int flags; };     // flags: 1 hasA, 2 hasB, 4 hasC, 8 b.inside, 16 waiting for B
constexpr int kWhittedMaxDepth = 7;

template <int ACCEL>
__device__ __forceinline__ bool whitted_occluded(const Scene& sc, const AltAccelDev& alt, f3 O, f3 D, float tmax, uint32_t* stk, Cnt& cn)   // FileScene::IsOccluded, file_scene.cpp:177-187
{
    {   // Quad::IsOccluded, primitives.h:347-362
        const float* c = sc.lightInvT;
        const float Oy = c[4] * O.x + c[5] * O.y + c[6] * O.z + c[7];
        const float Dy = c[4] * D.x + c[5] * D.y + c[6] * D.z;
        const float t = Oy / -Dy;
        if (t < tmax && t > 0) {
            const float Ox = c[0] * O.x + c[1] * O.y + c[2] * O.z + c[3];
            const float Oz = c[8] * O.x + c[9] * O.y + c[10] * O.z + c[11];
            const float Dx = c[0] * D.x + c[1] * D.y + c[2] * D.z;
            const float Dz = c[8] * D.x + c[9] * D.y + c[10] * D.z;
            const float Ix = Ox + t * Dx, Iz = Oz + t * Dz;
            const float size = sc.lightSize;
            if (Ix > -size && Ix < size && Iz > -size && Iz < size) return true;
        }
    }
    // shadow.t = 1e34f; acc.Intersect(shadow): a full nearest-hit query over the whole ray (bug-compatible: not clipped at the light)
    Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
    const f3 rD = mk3(1 / D.x, 1 / D.y, 1 / D.z);
    int traversed = 0, tested = 0;
    cn.rays++;
    if (ACCEL == 1) kd_intersect(alt, O, D, rD, h, stk, traversed, tested);          // (stk: this lane's column, two words per entry)
    else if (ACCEL == 2) grid_intersect(alt, O, D, rD, h, traversed, tested);
    else if (sc.kind == 0) traverse_bvh_seq(sc, sc.rootRef, O, D, rD, h, stk, cn, traversed, tested);
    else {
        Hit hh = h; Cnt dummy = cn;
        // TLAS walk without the quad / plane tests: reuse find_nearest_seq's TLAS part through a light-less copy is not possible, so inline it
        const char* __restrict__ g = sc.geom;
        uint32_t* tstk = stk + sc.bvhStack * 64;
        uint32_t cur = sc.rootRef, sp = 0;
        for (;;) {
            cn.tlas++;
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
        (void)hh; (void)dummy;
    }
    return h.objIdx > -1;
}

// ACCEL: 0 = the scene's BVH / TLAS, 1 / 2 = FileScene's KD-tree / uniform grid (crt_set_render_accel) for both the nearest-hit and the shadow queries
template <int ACCEL>
__global__ __launch_bounds__(64) void whitted_kernel(const Scene sc, const AltAccelDev alt, float4* __restrict__ acc, uint32_t* __restrict__ pixels, Counters* __restrict__ counters)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * 64u + lane;
    const uint32_t W = (uint32_t)sc.W, H = (uint32_t)sc.H;
    uint32_t* stk = lds + lane;
    Cnt cn; cn.rays = cn.primary = cn.interior = cn.leaf = cn.tri = cn.tlas = cn.visits = cn.meshhits = 0;
    if (idx < W * H) {
        const uint32_t x = idx % W, y = idx / W;
        const f3 camPos = mk3(sc.camPos[0], sc.camPos[1], sc.camPos[2]);
        const f3 TL = mk3(sc.topLeft[0], sc.topLeft[1], sc.topLeft[2]);
        const f3 TR = mk3(sc.topRight[0], sc.topRight[1], sc.topRight[2]);
        const f3 BL = mk3(sc.bottomLeft[0], sc.bottomLeft[1], sc.bottomLeft[2]);
        const float u = (float)x * sc.invW, v = (float)y * sc.invH;                   // GetPrimaryRay((float)x, (float)y), no jitter
        const f3 P = TL + u * (TR - TL) + v * (BL - TL);
        f3 O = camPos, D = normalize3(P - camPos); bool inside = false;
        cn.primary++;
        WFrame fr[kWhittedMaxDepth];
        int sp = 0;                       // frames in use; the ray (O, D, inside) is traced at depth `sp`
        f3 res = mk3(0, 0, 0);
        bool descend = true;
        for (;;) {
            if (descend) {
                // ---- Trace(ray, depth = sp) up to the point where it needs its children --------------------------------
                bool terminal = true;
                if (sp > sc.depthLimit) res = mk3(0, 0, 0);                             // renderer.cpp:23 (checked BEFORE tracing)
                else {
                    Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
                    const f3 rD = mk3(1 / D.x, 1 / D.y, 1 / D.z);
                    int traversed = 0, tested = 0;
                    if (ACCEL == 0) find_nearest_seq(sc, O, D, rD, h, stk, cn, traversed, tested);
                    else {
                        cn.rays++;
                        hit_light_floor(sc, O, D, h);
                        if (ACCEL == 1) kd_intersect(alt, O, D, rD, h, stk, traversed, tested); else grid_intersect(alt, O, D, rD, h, traversed, tested);
                        if (h.objIdx >= 2) cn.meshhits++;
                    }
                    if (h.objIdx == -1) res = sky_color(sc, D);
                    else if (h.objIdx == 0) res = mk3(24, 24, 22);
                    else {
                        const f3 I = O + h.t * D;
                        f3 N; float tu = 0, tv = 0; Material m;
                        if (h.objIdx == 1) {
                            N = mk3(sc.floorN[0], sc.floorN[1], sc.floorN[2]);
                            if (N.y == 1) { float fu = I.x, fv = I.z; fu *= sc.floorInvto; fv *= sc.floorInvto; tu = fu - __builtin_floorf(fu); tv = fv - __builtin_floorf(fv); }
                            m = sc.floorMat;
                        } else {
                            const uint32_t so = sc.shadeOff + (uint32_t)h.triIdx * 64u;
                            const rec4 s0 = ldg(sc.geom, so), s1 = ldg(sc.geom, so + 16), s2 = ldg(sc.geom, so + 32), s3 = ldg(sc.geom, so + 48);
                            const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
                            const float w = 1 - h.u - h.v;
                            const f3 Nn = w * n0 + h.u * n1 + h.v * n2;
                            tu = w * s2.y + h.u * s2.w + h.v * s3.y;
                            tv = w * s2.z + h.u * s3.x + h.v * s3.z;
                            const rec4* mp = reinterpret_cast<const rec4*>(sc.mats + (int)asu(s3.w));
                            const rec4 m0 = mp[0], m1 = mp[1];
                            m.reflectivity = m0.x; m.refractivity = m0.y; m.absorption[0] = m0.z; m.absorption[1] = m0.w; m.absorption[2] = m1.x;
                            m.texOffset = asu(m1.y); m.texW = (int)asu(m1.z); m.texH = (int)asu(m1.w);
                            if (sc.kind == 0) N = normalize3(Nn);
                            else {
                                const uint32_t io = sc.instOff + (uint32_t)(h.objIdx - 2) * 128u + 64u;
                                const rec4 r0 = ldg(sc.geom, io), r1 = ldg(sc.geom, io + 16), r2 = ldg(sc.geom, io + 32);
                                N = normalize3(mk3(r0.x * Nn.x + r0.y * Nn.y + r0.z * Nn.z + r0.w * 0.0f,
                                                   r1.x * Nn.x + r1.y * Nn.y + r1.z * Nn.z + r1.w * 0.0f,
                                                   r2.x * Nn.x + r2.y * Nn.y + r2.z * Nn.z + r2.w * 0.0f));
                            }
                        }
                        if (dot3(N, D) > 0) N = -N;
                        const f3 albedo = (m.texW > 0) ? tex_sample(sc, m.texOffset, m.texW, m.texH, tu, tv) : mk3(1.0f, 1.0f, 1.0f);
                        WFrame& f = fr[sp];
                        f.flags = 0; f.out = mk3(0, 0, 0);
                        const float refl = m.reflectivity, refr = m.refractivity;
                        const float diffuseness = 1 - (refl + refr);
                        f3 aO = O, aD = D; bool aInside = false;
                        if (refl > 0.0f) {                                               // renderer.cpp:49-54
                            const f3 R = D - 2.0f * N * dot3(N, D);
                            aO = I + R * CRT_EPS; aD = R; aInside = false;
                            f.cA = refl * albedo; f.flags |= 1;
                        } else if (refr > 0.0f) {                                        // renderer.cpp:55-72
                            const f3 R = D - 2.0f * N * dot3(N, D);
                            const float n1 = inside ? 1.2f : 1, n2 = inside ? 1 : 1.2f;
                            const float eta = n1 / n2, cosi = dot3(-D, N);
                            const float cost2 = 1.0f - eta * eta * (1 - cosi * cosi);
                            float Fr = 1;
                            if (cost2 > 0) {
                                const float a = n1 - n2, b = n1 + n2, R0 = (a * a) / (b * b), c = 1 - cosi;
                                Fr = R0 + (1 - R0) * (c * c * c * c * c);
                                const f3 T = eta * D + ((eta * cosi - __builtin_sqrtf(__builtin_fabsf(cost2))) * N);
                                aO = I + T * CRT_EPS; aD = T; aInside = !inside;
                                f.cA = albedo * (1 - Fr); f.flags |= 1;
                            }
                            f.bO = I + R * CRT_EPS; f.bD = R; f.cB = albedo * Fr; f.flags |= 2;
                        }
                        if (diffuseness > 0) {                                           // renderer.cpp:74-80, DirectIllumination :105-126
                            f3 irr = mk3(0, 0, 0);
                            f3 L = mk3(sc.lightPos[0], sc.lightPos[1], sc.lightPos[2]) - I;
                            const float dist = __builtin_sqrtf(dot3(L, L));
                            L = L * (1 / dist);
                            const float ndotl = dot3(N, L);
                            if (!(ndotl < CRT_EPS)) {
                                if (!whitted_occluded<ACCEL>(sc, alt, I + L * CRT_EPS, L, dist - 2 * CRT_EPS, stk, cn)) {
                                    const float att = 1 / (dist * dist);
                                    const f3 inr = mk3(24, 24, 22) * att;
                                    irr = inr * dot3(N, L);
                                }
                            }
                            const f3 brdf = albedo * CRT_INVPI;
                            f.C = diffuseness * brdf * (irr + mk3(0.3f, 0.3f, 0.3f)); f.flags |= 4;
                        }
                        f.medium = mk3(1, 1, 1);
                        if (inside) f.medium = mk3(crt_expf(m.absorption[0] * -h.t), crt_expf(m.absorption[1] * -h.t), crt_expf(m.absorption[2] * -h.t));
                        terminal = false;
                        if (f.flags & 1) { O = aO; D = aD; inside = aInside; sp++; }                      // trace child A next
                        else if (f.flags & 2) { O = f.bO; D = f.bD; inside = false; f.flags |= 16; sp++; } // (not reachable in the reference: B only exists with refr > 0)
                        else { res = (f.flags & 4) ? f.out + f.C : f.out; res = f.medium * res; terminal = true; }
                    }
                }
                if (terminal) descend = false;
            } else {
                // ---- a child returned `res` to the frame below it ------------------------------------------------------
                if (sp == 0) break;
                WFrame& f = fr[sp - 1];
                bool finalize = true;
                if (!(f.flags & 16)) {                                                   // it was child A (or the only child)
                    if (f.flags & 1) f.out = f.out + f.cA * res;
                    if (f.flags & 2) { O = f.bO; D = f.bD; inside = false; f.flags |= 16; descend = true; finalize = false; }
                } else f.out = f.out + f.cB * res;
                if (finalize) {
                    f3 o = f.out;
                    if (f.flags & 4) o = o + f.C;
                    res = f.medium * o;
                    sp--;
                }
            }
        }
        acc[idx] = make_float4(res.x, res.y, res.z, 0.0f);
        const uint32_t r = (uint32_t)(255.0f * min_std(1.0f, res.x)), g = (uint32_t)(255.0f * min_std(1.0f, res.y)), bb = (uint32_t)(255.0f * min_std(1.0f, res.z));
        pixels[idx] = (r << 16) + (g << 8) + bb;
    }
    uint32_t vals[8] = {cn.rays, cn.primary, cn.interior, cn.leaf, cn.tri, cn.tlas, cn.visits, cn.meshhits};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t s = wave_sum(vals[k]);
        if (lane == 0 && s) atomicAdd(&counters->v[k], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------------------------
// resolve_kernel: one thread per owned tile; pixels in ProcessTile order so the per-tile energy sum is bit-exact
// (renderer.cpp:119,127-129; RGBF32_to_RGB8 scalar branch, template/precomp.h:336-340)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void resolve_kernel(const float4* __restrict__ acc, uint32_t* __restrict__ pixels, float* __restrict__ tileSums,
                                                      uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                                      uint32_t W, float scale)
{
    const uint32_t tl = blockIdx.x * blockDim.x + threadIdx.x;
    if (tl >= tileCount) return;
    const uint32_t tile = tileFirst + tl * tileStride;
    const uint32_t x0 = (tile % tilesX) * 16u, y0 = (tile / tilesX) * 16u;
    float sum = 0;
    for (uint32_t v = 0; v < 16; v++) for (uint32_t u = 0; u < 16; u++) {
        const size_t i = (x0 + u) + (size_t)(y0 + v) * W;
        const float4 a = acc[i];
        const float px = a.x * scale, py = a.y * scale, pz = a.z * scale;
        sum += px + py + pz;
        const uint32_t r = (uint32_t)(255.0f * min_std(1.0f, px)), g = (uint32_t)(255.0f * min_std(1.0f, py)), bb = (uint32_t)(255.0f * min_std(1.0f, pz));
        pixels[i] = (r << 16) + (g << 8) + bb;
    }
    tileSums[tile] = sum;
}

} // namespace crt

// ------------------------------------------------------------------------------------------------------------
// launch wrappers (called from abi.cpp)
// ------------------------------------------------------------------------------------------------------------
// self-check of dev_common.h's short reciprocals on the device at hand (tests/test_gpu_reciprocal.py through crt_debug_check_reciprocals): every one of the
// 2^32 float bit patterns through rcp_exact / rcp_exact_large / rcp_exact3 against the IEEE division, bit for bit (two NaNs count as equal).
// out = {inputs, rcp_exact differences, rcp_exact_large differences among |x| >= 1e-4 and NaN, rcp_exact3 differences}
namespace crt {
__device__ __forceinline__ bool same_bits(float a, float b) { return asu(a) == asu(b) || (a != a && b != b); }
__global__ __launch_bounds__(256) void check_reciprocals_kernel(unsigned long long* out)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nthreads = gridDim.x * 256u;
    unsigned long long n = 0, bad1 = 0, bad2 = 0, bad3 = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += nthreads) {
        const float x = asf((uint32_t)b), want = 1.0f / x;
        n++;
        if (!same_bits(rcp_exact(x), want)) bad1++;
        if (!(__builtin_fabsf(x) < 0.0001f) && !same_bits(rcp_exact_large(x), want)) bad2++;
        // the other two components: a value from another part of the number line, and one that leaves the guard range now and then
        const float y = asf((uint32_t)b * 2654435761u), z = asf(((uint32_t)b >> 7) * 40503u + 0x3f000000u);
        const f3 r = rcp_exact3(mk3(x, y, z));
        if (!same_bits(r.x, want) || !same_bits(r.y, 1.0f / y) || !same_bits(r.z, 1.0f / z)) bad3++;
    }
    atomicAdd(&out[0], n); atomicAdd(&out[1], bad1); atomicAdd(&out[2], bad2); atomicAdd(&out[3], bad3);
}
} // namespace crt
extern "C" hipError_t crt_launch_check_reciprocals(unsigned long long* out, hipStream_t stream)
{
    hipLaunchKernelGGL(crt::check_reciprocals_kernel, dim3(4096), dim3(256), 0, stream, out);
    return hipGetLastError();
}

// blockDesc / nBlocks: block table (see the kernel; the launch renders exactly the table's blocks), else nullptr / 0; tileCost: nullptr or one uint32 per tile, atomicMax'ed
extern "C" hipError_t crt_launch_render(const crt::Scene* sc, void* slab, crt::Counters* counters, unsigned long long* tileClocks, const uint32_t* tileOrder,
                                        uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX, uint32_t sppFirst,
                                        uint32_t frames, uint32_t passes, uint32_t ldsBytes, int collectStats, const uint32_t* blockDesc, uint32_t nBlocks, uint32_t* tileCost, uint32_t rankCount, unsigned long long* launchClk, hipStream_t stream)
{
    if (tileCount == 0 || frames == 0) return hipSuccess;
    const uint32_t windows = (frames + 63u) / 64u;                      // one 64-lane wavefront per (tile, 64-frame window)
    if ((unsigned long long)tileCount * windows > 0x7fffffffull) return hipErrorInvalidValue;
    if (collectStats || nBlocks == 0u) blockDesc = nullptr;
    if (blockDesc && (tileCount > 0x10000u || windows > 64u)) return hipErrorInvalidValue;
    if (rankCount == 0u || rankCount > tileCount) rankCount = tileCount;          // the first rankCount tiles of the order only (all windows)
    dim3 grid(blockDesc ? nBlocks : rankCount * windows), block(64);
#define CRT_LAUNCH(K, C) hipLaunchKernelGGL((crt::render_tiles_kernel<K, C>), grid, block, ldsBytes + 15u * 64u * 4u /* throughput-factor columns */, stream, *sc, (float4*)slab, counters, tileClocks, tileOrder, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes, windows, blockDesc, tileCost, rankCount, launchClk)
    if (sc->kind == 0) { if (collectStats) CRT_LAUNCH(0, true); else CRT_LAUNCH(0, false); }
    else { if (collectStats) CRT_LAUNCH(1, true); else CRT_LAUNCH(1, false); }
#undef CRT_LAUNCH
    return hipGetLastError();
}

// how many wavefronts of render_tiles_kernel the device holds at once (registers: 5 per SIMD; LDS: the stack columns) — the size the latency tuner fits its block tables to
extern "C" uint32_t crt_render_resident_waves(int device, int kind, uint32_t ldsBytes)
{
    int perCu = 0, cus = 0;
    const size_t lds = (size_t)ldsBytes + 15u * 64u * 4u;
    hipError_t e = kind == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, crt::render_tiles_kernel<0, false>, 64, lds)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, crt::render_tiles_kernel<1, false>, 64, lds);
    if (e != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || perCu <= 0 || cus <= 0) { (void)hipGetLastError(); return 5120u; }
    return (uint32_t)perCu * (uint32_t)cus;
}

extern "C" hipError_t crt_launch_accumulate(const void* slab, void* acc, uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount,
                                            uint32_t tilesX, uint32_t W, uint32_t frames, uint32_t passes, hipStream_t stream)
{
    if (tileCount == 0 || frames == 0) return hipSuccess;
    dim3 grid(tileCount), block(256);
    hipLaunchKernelGGL(crt::accumulate_kernel, grid, block, 4u * crt::kAccWaveLds, stream, (const float4*)slab, (float*)acc, tileFirst, tileStride, tileCount, tilesX, W, frames, passes);
    return hipGetLastError();
}

extern "C" hipError_t crt_launch_find_nearest(const crt::Scene* sc, const void* rays, void* hits, uint32_t n, crt::Counters* counters,
                                              uint32_t ldsBytes, uint32_t* cursor, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (!cursor) return hipErrorInvalidValue;
    if (hipMemsetAsync(cursor, 0, 4, stream) != hipSuccess) return hipGetLastError();
    uint32_t perCu = ldsBytes ? (160u * 1024u) / ldsBytes : 16u; if (perCu > 16u) perCu = 16u; if (perCu < 4u) perCu = 4u;      // 4 wavefronts per SIMD, LDS stacks permitting (measured: 8 per SIMD is no faster for the grid and 17 % slower for the BVH)
    const uint32_t need = (n + 63u) / 64u, fill = 256u * perCu;          // persistent wavefronts: the device full once, never more than the rays need
    dim3 grid(need < fill ? need : fill), block(64);
    hipLaunchKernelGGL(crt::find_nearest_kernel, grid, block, ldsBytes, stream, *sc, (const crt::RayIn*)rays, (crt::HitOut*)hits, n, counters, cursor);
    return hipGetLastError();
}

extern "C" hipError_t crt_launch_whitted(const crt::Scene* sc, int accel, const crt::AltAccelDev* alt, void* acc, uint32_t* pixels, crt::Counters* counters, uint32_t ldsBytes, hipStream_t stream)
{
    const uint32_t n = (uint32_t)sc->W * (uint32_t)sc->H;
    dim3 grid((n + 63u) / 64u), block(64);
    if (accel == 1) hipLaunchKernelGGL(crt::whitted_kernel<1>, grid, block, alt->kdStack * 128u * 4u, stream, *sc, *alt, (float4*)acc, pixels, counters);
    else if (accel == 2) hipLaunchKernelGGL(crt::whitted_kernel<2>, grid, block, 256u, stream, *sc, *alt, (float4*)acc, pixels, counters);
    else { const crt::AltAccelDev none{}; hipLaunchKernelGGL(crt::whitted_kernel<0>, grid, block, ldsBytes, stream, *sc, none, (float4*)acc, pixels, counters); }
    return hipGetLastError();
}

extern "C" hipError_t crt_launch_resolve(const void* acc, uint32_t* pixels, float* tileSums, uint32_t tileFirst, uint32_t tileStride,
                                         uint32_t tileCount, uint32_t tilesX, uint32_t W, float scale, hipStream_t stream)
{
    if (tileCount == 0) return hipSuccess;
    dim3 grid((tileCount + 63u) / 64u), block(64);
    hipLaunchKernelGGL(crt::resolve_kernel, grid, block, 0, stream, (const float4*)acc, pixels, tileSums, tileFirst, tileStride, tileCount, tilesX, W, scale);
    return hipGetLastError();
}
