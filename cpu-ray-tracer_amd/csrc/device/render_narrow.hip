// render_narrow.hip — render_narrow_kernel: the per-tile sample loop ("3. PathTracer/renderer.cpp":117-131, Sample :50-100, FindNearest
// infra/scene/file_scene.cpp:170-175, IntersectBVH infra/bvh.cpp:224-258) for block-table wavefronts of 1 .. 8 lanes — the LATENCY form.
//
// A launch ends on its slowest wavefront, and a wavefront is as slow as the serial chain of its streams (one xorshift32 stream per (tile, frame), consumed
// serially over the tile's 256 pixels: renderer.cpp:120-126).  render_tiles_kernel lets the host hand an expensive tile's 64 streams to 64 / L wavefronts of
// L lanes, but each of them still pays that kernel's phase machinery per step (state ballots, phase selection, every populated phase, one trip per step:
// ~150-200 instructions and one exposed record fetch per step for a lone wave, which issues one instruction per ~5 cycles whatever its lane count).
// This kernel is what such a wavefront should run instead: every lane is a plain sequential path tracer — no phases, no queues — and the traversal hides
// what a lone wave cannot overlap otherwise:
//   * at an interior node the records of BOTH children are requested as soon as the node's own record (which names them) has arrived, before the two slab
//     tests run: the slab arithmetic overlaps the fetch, and the step that follows a descent finds its record already in registers;
//   * the far child's record goes onto a record stack in LDS together with its reference, so a pop costs an LDS read, not a memory round trip;
//   * the top of the tree — its first 512 child pairs in breadth-first order, a copy with re-written child references (Scene::topOff, abi.cpp crt_build_treetop) —
//     lives in LDS, shared by the four wavefronts of a workgroup: a dependent 64-byte fetch costs ~110 ns from LDS against 220 ns (idle chip) .. 430+ ns (4 096
//     wavefronts in flight) from L2 (tools/microbench/chase.hip), and about half of a ray's node steps are in those levels.
// Per stream nothing changes: every ray visits the reference's nodes in the reference's order and every float expression is evaluated as written there
// (the same dev_common.h helpers as the other two render kernels), so the samples are bit-identical to theirs and to the CPU oracle's.
// Two-level scenes walk TLAS and BLAS with the sequential reference-order traversal of dev_common.h (find_nearest_seq), without the speculation.
//
// Numerics: -ffp-contract=off, IEEE + - * / sqrt only (dev_common.h).  No MFMA: pointer chasing + slab / Möller–Trumbore tests.
#include "alt_common.h"

namespace crt {

#ifndef CRT_NARROW_MAX
#define CRT_NARROW_MAX 2
#endif
constexpr uint32_t kNarrowMaxLanes = CRT_NARROW_MAX;     // widest wavefront the kernel accepts; what the host routes to it is its decision (abi.cpp split_by_width: nothing by default)
struct Rec { rec4 a, b, c, d; };                           // a fetched 64-byte record (NodePair: child 0 = {a, b}, child 1 = {c, d}; LeafTri: a, b, c)

// per-lane LDS of a FileScene wavefront: [stackDepth references][stackDepth records of 16 dwords][15 throughput factors], dwords
__device__ __host__ __forceinline__ uint32_t narrow_lane_dwords(uint32_t stackDepth) { return (stackDepth * 17u + 15u + 3u) & ~3u; }   // (a multiple of 4: the records are 16-byte accesses)

constexpr uint32_t kProbeWaves = 8u;                       // cost probe: wavefronts (of 64 one-path lanes) per tile
constexpr uint32_t kNarrowWaves = 4u;                      // wavefronts per workgroup: they share the treetop in LDS, each renders one block of the table

// PROBE = true is the cost probe of the latency mode (abi.cpp probe_tile_costs): kProbeWaves wavefronts per tile, lane l of wavefront q traces ONE path through pixel (4 l + q) % 256 of the
// tile with a seed of its own (256 paths per tile per four wavefronts), nothing is stored, and every wavefront adds the number of traversal / shading steps its 64 paths took to tileCost[tile] (zeroed by the host) — an estimate of
// what the tile's streams will cost (a stream = 256 such paths), available well under a millisecond after a camera or scene change instead of after a first full render.
// MODE 0 = block-table wavefronts of <= kNarrowMaxLanes lanes (above), MODE 1 = the cost probe (PROBE),
// MODE 2 / 3 = whole (tile, window) wavefronts, lane = frame, that trace Renderer::Sample through FileScene's KD-tree / uniform grid instead of the BVH
// (crt_set_render_accel: the reference's shipped FileScene traces through the KD-tree, file_scene.h:10-12, file_scene.cpp:170-175) — the sequential form:
// entry = tile rank * windows + window.
template <int KIND, int MODE>
__global__ __launch_bounds__(256, 4) void render_narrow_kernel(const Scene sc, const AltAccelDev acc, float4* __restrict__ slab, Counters* __restrict__ counters,
                                                              uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                                              uint32_t sppFirst, uint32_t frames, uint32_t passes,
                                                              const uint32_t* __restrict__ blockDesc, uint32_t nBlocks, uint32_t* __restrict__ tileCost)
{
    constexpr bool PROBE = MODE == 1, ALT = MODE >= 2;
    extern __shared__ uint32_t ldsAll[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long clk0 = tileCost ? wall_clock64() : 0ull;
    const char* __restrict__ geom = sc.geom;
    // the treetop into LDS, by all 256 threads (FileScene only)
    const rec4* ldsTop = reinterpret_cast<const rec4*>(ldsAll);
    const uint32_t topDwords = (KIND == 0 && MODE == 0) ? sc.topCount * 16u : 0u;
    if (KIND == 0 && MODE == 0 && sc.topCount) {
        rec4* dst = reinterpret_cast<rec4*>(ldsAll);
        for (uint32_t i = threadIdx.x; i < sc.topCount * 4u; i += 256u) dst[i] = ldg(geom, sc.topOff + i * 16u);
        __syncthreads();
    }
    // blockDesc[block] = local tile index | first frame << 16 | log2(lanes) << 22 | window << 25 (the table format of render_tiles_kernel; abi.cpp)
    const uint32_t entry = blockIdx.x * kNarrowWaves + wave;
    if (entry >= nBlocks) return;
    const uint32_t windowsAll = (frames + 63u) / 64u;
    const uint32_t d = PROBE ? ((entry / kProbeWaves) | (6u << 22)) : ALT ? ((entry / windowsAll) | (6u << 22) | ((entry % windowsAll) << 25)) : blockDesc[entry];   // (probe / accelerator modes: all 64 lanes)
    const uint32_t tl = d & 0xffffu, laneBase = (d >> 16) & 63u, myLanes = 1u << ((d >> 22) & 7u), win = d >> 25;
    if (tl >= tileCount || (MODE == 0 && myLanes > kNarrowMaxLanes)) return;
    sppFirst += win * 64u * passes;
    frames = (frames - win * 64u < 64u) ? frames - win * 64u : 64u;               // frames of THIS window (the last one may be partial)
    frames = frames > laneBase ? ((frames - laneBase < myLanes) ? frames - laneBase : myLanes) : 0u;
    if (lane >= frames) return;                                                   // the wavefront runs with `frames` (<= kNarrowMaxLanes) lanes from here on
    slab += (size_t)win * ((size_t)tileCount * 256u * 64u * passes);              // this window's region of the sample slab
    const uint32_t tile = tileFirst + tl * tileStride;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    // this wavefront's LDS behind the treetop
    uint32_t* lds = ldsAll + topDwords + wave * ((KIND == 0 && MODE == 0) ? kNarrowMaxLanes * narrow_lane_dwords(sc.stackDepth) : ALT ? (acc.kdStack * 2u + 15u) * 64u : (sc.stackDepth + 15u) * 64u);

    uint32_t nRays = 0, nPrimary = 0, nMesh = 0;
    const uint32_t items = PROBE ? 1u : 256u * passes;                            // (pixel, pass) pairs in stream order
    uint32_t seed = init_seed(tx + ty * (uint32_t)sc.W + (sppFirst + (laneBase + lane) * passes) * 1799u);   // renderer.cpp:120
    if (PROBE) seed = init_seed(0x9e3779b9u ^ ((tile * kProbeWaves + entry % kProbeWaves) * 64u + lane));
    uint32_t steps = 0;                                                           // probe: traversal + weighted shading steps of this lane's path

    // LDS of this lane
    uint32_t* refStk; rec4* recStk; float* fst; uint32_t* seqStk = nullptr; uint32_t fstStride;
    if (KIND == 0 && MODE == 0) {
        uint32_t* mine = lds + lane * narrow_lane_dwords(sc.stackDepth);
        recStk = reinterpret_cast<rec4*>(mine); refStk = mine + sc.stackDepth * 16u; fst = reinterpret_cast<float*>(refStk + sc.stackDepth); fstStride = 1u;
    } else {
        seqStk = lds + lane; refStk = nullptr; recStk = nullptr;                  // find_nearest_seq's column layout: entry i at [i * 64] (KD-tree: two words per entry, [i * 128] and [i * 128 + 64])
        fst = reinterpret_cast<float*>(lds + (ALT ? acc.kdStack * 2u : sc.stackDepth) * 64u + lane); fstStride = 64u;
    }

    const kernarg_f cam = scene_floats(offsetof(Scene, camPos));                  // camPos, topLeft, topRight, bottomLeft, invW, invH
    const f3 camPos = mk3(cam[0], cam[1], cam[2]);
    const f3 TL = mk3(cam[3], cam[4], cam[5]), TR = mk3(cam[6], cam[7], cam[8]), BL = mk3(cam[9], cam[10], cam[11]);
    const float invW = cam[12], invH = cam[13];

    for (uint32_t item = 0; item < items; item++) {
        // ---------------- ProcessTile + Camera::GetPrimaryRay (renderer.cpp:125-126, camera.h:23-30) ----------------
        const uint32_t pix = PROBE ? ((lane * 4u + entry % kProbeWaves) & 255u) : ((passes == 1u) ? item : item / passes);
        const int x = (int)(tx * 16u + (pix & 15u)), y = (int)(ty * 16u + (pix >> 4));
        const float jy = rnd(seed);                                               // pinned: first draw is the y jitter
        const float jx = rnd(seed);
        const float u = ((float)x + jx) * invW, vv = ((float)y + jy) * invH;
        const f3 P = TL + u * (TR - TL) + vv * (BL - TL);
        f3 v = P - camPos;
        f3 O = camPos, D = v * rcp_exact(__builtin_sqrtf(dot3(v, v)));            // normalize()
        bool inside = false; int depth = 0;
        nPrimary++;
        f3 L = mk3(0, 0, 0);
        for (;;) {
            // ---------------- scene.FindNearest (file_scene.cpp:170-175 / tlas_file_scene.cpp:201-206) ----------------
            const f3 rD = rcp_exact3(D);
            Hit h; h.t = 1e34f; h.u = 0; h.v = 0; h.objIdx = -1; h.triIdx = -1;
            nRays++;
            if (ALT) {
                int traversed = 0, tested = 0;
                hit_light_floor(sc, O, D, h);                                      // FileScene::FindNearest: light quad, floor plane, then the accelerator
                if (MODE == 2) kd_intersect(acc, O, D, rD, h, seqStk, traversed, tested);
                else grid_intersect(acc, O, D, rD, h, traversed, tested);
            } else if (KIND == 1 || PROBE) {
                Cnt cn; cn.rays = cn.primary = cn.interior = cn.leaf = cn.tri = cn.tlas = cn.visits = cn.meshhits = 0;
                int traversed = 0, tested = 0;
                find_nearest_seq(sc, O, D, rD, h, seqStk, cn, traversed, tested);
                if (PROBE) steps += cn.interior + cn.tri + cn.tlas + 3u;          // a shading step weighs about three traversal steps
            } else {
                hit_light_floor(sc, O, D, h);
                // ---- BVH::IntersectBVH (bvh.cpp:224-258): ordered, stack-based; `q` = the record of `cur`.  References: kRefInterior | offset (pair in HBM / L2),
                // kRefTop | index (pair in the LDS treetop), leaf offset, 0 = done
                uint32_t cur = sc.rootRef, sp = 0;
                Rec q;
                if (sc.topCount) { cur = kRefTop; q.a = ldsTop[0]; q.b = ldsTop[1]; q.c = ldsTop[2]; q.d = ldsTop[3]; }
                else if (sc.rootIsPair) {
                    const kernarg_f rp = scene_floats(offsetof(Scene, rootPair));   // the root's child pair travels in the kernel arguments
                    q.a = rec4{rp[0], rp[1], rp[2], rp[3]}; q.b = rec4{rp[4], rp[5], rp[6], rp[7]}; q.c = rec4{rp[8], rp[9], rp[10], rp[11]}; q.d = rec4{rp[12], rp[13], rp[14], rp[15]};
                } else {
                    const uint32_t o = (cur & kRefOffsetMask) << 4;
                    q.a = ldg(geom, o); q.b = ldg(geom, o + 16u); q.c = ldg(geom, o + 32u); q.d = ldg(geom, o + 48u);
                }
                const bool rayFinite = finite3(rD);
                auto pop = [&]() {
                    if (sp != 0u) {
                        sp--; cur = refStk[sp];
                        const rec4* e = ((cur & 0xC0000000u) == kRefTop) ? ldsTop + (cur & kRefOffsetMask) * 4u : recStk + sp * 4u;      // treetop entry | the record pushed with the reference
                        q.a = e[0]; q.b = e[1]; q.c = e[2]; q.d = e[3];
                    } else cur = kRefDone;
                };
                while (cur != kRefDone) {
                    if ((cur & 0xC0000000u) != 0u) {
                        // interior node.  The records of children that are not in the treetop are requested before the slab tests (a leaf child's record is its first LeafTri)
                        const uint32_t ra = asu(q.a.w), rb = asu(q.c.w);
                        const bool ta = (ra & 0xC0000000u) == kRefTop, tb = (rb & 0xC0000000u) == kRefTop;
                        Rec qa, qb;
                        qa.a = qa.b = qa.c = qa.d = q.a; qb = qa;
                        if (!ta) { const uint32_t oa = (ra & kRefOffsetMask) << 4; qa.a = ldg(geom, oa); qa.b = ldg(geom, oa + 16u); qa.c = ldg(geom, oa + 32u); qa.d = ldg(geom, oa + 48u); }
                        if (!tb) { const uint32_t ob = (rb & kRefOffsetMask) << 4; qb.a = ldg(geom, ob); qb.b = ldg(geom, ob + 16u); qb.c = ldg(geom, ob + 32u); qb.d = ldg(geom, ob + 48u); }
                        float d1, d2;
                        if (__builtin_amdgcn_ballot_w64(!rayFinite) == 0ull) { d1 = box_fast(q.a, q.b, O, rD, h.t); d2 = box_fast(q.c, q.d, O, rD, h.t); }
                        else { d1 = box_exact(q.a, q.b, O, rD, h.t); d2 = box_exact(q.c, q.d, O, rD, h.t); }
                        const bool sw = d1 > d2;                                   // near child first (strict >: ties keep child 1)
                        const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
                        const uint32_t rn = sw ? rb : ra, rf = sw ? ra : rb;
                        if (dn != 1e30f) {
                            if (df != 1e30f) {                                     // push the far child: its reference and, unless it is a treetop entry, its record
                                refStk[sp] = rf;
                                if ((rf & 0xC0000000u) != kRefTop) {
                                    rec4* e = recStk + sp * 4u;
                                    e[0] = sw ? qa.a : qb.a; e[1] = sw ? qa.b : qb.b; e[2] = sw ? qa.c : qb.c; e[3] = sw ? qa.d : qb.d;
                                }
                                sp++;
                            }
                            cur = rn;
                            if ((rn & 0xC0000000u) == kRefTop) { const rec4* e = ldsTop + (rn & kRefOffsetMask) * 4u; q.a = e[0]; q.b = e[1]; q.c = e[2]; q.d = e[3]; }
                            else { q.a = sw ? qb.a : qa.a; q.b = sw ? qb.b : qa.b; q.c = sw ? qb.c : qa.c; q.d = sw ? qb.d : qa.d; }
                        } else pop();
                    } else {
                        // leaf: its triangles one by one (bvh.cpp:232-243); `q` = the current LeafTri (48 B of the 64 fetched)
                        hit_tri(q.a, q.b, q.c, O, D, h);
                        if (asu(q.c.w) > 1u) {                                     // the leaf's next LeafTri follows (48 B = 3 units)
                            cur += 3u;
                            const uint32_t o = (cur & kRefOffsetMask) << 4;
                            q.a = ldg(geom, o); q.b = ldg(geom, o + 16u); q.c = ldg(geom, o + 32u);
                        } else pop();
                    }
                }
            }
            if (h.objIdx >= 2) nMesh++;
            // ---------------- Renderer::Sample (renderer.cpp:50-100) ----------------
            if (h.objIdx == -1) { L = sky_color(sc, D); break; }                   // GetSkyColor, file_scene.cpp:142-154
            if (depth >= sc.depthLimit) { L = mk3(0, 0, 0); break; }
            if (h.objIdx == 0) { L = mk3(24, 24, 22); break; }                     // GetLightColor, file_scene.cpp:164-167
            const f3 I = O + h.t * D;
            f3 N; float tu = 0, tv = 0, refl, refr; f3 absorb; uint32_t tOff; int tW, tH;
            if (h.objIdx == 1) {                                                   // floor: Plane::GetNormal / GetUV (primitives.h:112-133)
                N = mk3(sc.floorN[0], sc.floorN[1], sc.floorN[2]);
                if (N.y == 1) {
                    float uu = I.x, vw = I.z;
                    uu *= sc.floorInvto; vw *= sc.floorInvto;
                    tu = uu - __builtin_floorf(uu); tv = vw - __builtin_floorf(vw);
                }
                refl = sc.floorMat.reflectivity; refr = sc.floorMat.refractivity;
                absorb = mk3(sc.floorMat.absorption[0], sc.floorMat.absorption[1], sc.floorMat.absorption[2]);
                tOff = sc.floorMat.texOffset; tW = sc.floorMat.texW; tH = sc.floorMat.texH;
            } else {                                                               // mesh: GetNormal / GetUV (bvh.cpp:290-305, blas_bvh.cpp:391-406)
                const uint32_t so = sc.shadeOff + (uint32_t)h.triIdx * 64u;
                const rec4 s0 = ldg(geom, so), s1 = ldg(geom, so + 16u), s2 = ldg(geom, so + 32u), s3 = ldg(geom, so + 48u);
                const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
                const float w = 1 - h.u - h.v;
                const f3 Nn = w * n0 + h.u * n1 + h.v * n2;
                tu = w * s2.y + h.u * s2.w + h.v * s3.y;
                tv = w * s2.z + h.u * s3.x + h.v * s3.z;
                const rec4* mp = reinterpret_cast<const rec4*>(sc.mats + (int)asu(s3.w));
                const rec4 m0 = mp[0], m1 = mp[1];
                refl = m0.x; refr = m0.y; absorb = mk3(m0.z, m0.w, m1.x);
                tOff = asu(m1.y); tW = (int)asu(m1.z); tH = (int)asu(m1.w);
                if (KIND == 0) N = normalize3(Nn);
                else {
                    const uint32_t io = sc.instOff + (uint32_t)(h.objIdx - 2) * 128u + 64u;   // Instance::T rows
                    const rec4 r0 = ldg(geom, io), r1 = ldg(geom, io + 16), r2 = ldg(geom, io + 32);
                    N = normalize3(mk3(r0.x * Nn.x + r0.y * Nn.y + r0.z * Nn.z + r0.w * 0.0f,
                                       r1.x * Nn.x + r1.y * Nn.y + r1.z * Nn.z + r1.w * 0.0f,
                                       r2.x * Nn.x + r2.y * Nn.y + r2.z * Nn.z + r2.w * 0.0f));
                }
            }
            if (dot3(N, D) > 0) N = -N;
            f3 c = mk3(1.0f, 1.0f, 1.0f);
            if (tW > 0) c = tex_sample(sc, tOff, tW, tH, tu, tv);                  // Material::GetAlbedo
            f3 medium = mk3(1, 1, 1);
            if (inside) {
                const f3 ab = absorb * -h.t;
                medium = mk3(crt_expf(ab.x), crt_expf(ab.y), crt_expf(ab.z));
            }
            f3 nv, factor; bool newInside = false;
            const float r = rnd(seed);
            if (r < refl) {                                                        // HandleMirror, renderer.cpp:20-25
                nv = D - 2.0f * N * dot3(N, D);
                factor = c * medium;
            } else if (r < refl + refr) {                                          // HandleDielectric, renderer.cpp:27-45
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
            } else {                                                               // diffuse, renderer.cpp:93-99; diffusereflection tmplmath.h:535-544
                f3 Rr;
                do {
                    const float rz = rnd_pm1(seed);                                // draw order pinned z, y, x
                    const float ry = rnd_pm1(seed);
                    const float rx = rnd_pm1(seed);
                    Rr = mk3(rx, ry, rz);
                } while (dot3(Rr, Rr) > 1);
                if (dot3(Rr, N) < 0) Rr = Rr * -1.0f;
                nv = Rr * rcp_exact(__builtin_sqrtf(dot3(Rr, Rr)));                // normalize(R)
                const f3 brdf = c * CRT_INVPI;
                const f3 pre = medium * brdf * 2.0f * CRT_PI;
                factor = pre * dot3(nv, N);
            }
            // the bounce's throughput factor (albedo*medium*... multiplies on return: depth <= 4 here)
            float* fd = fst + (uint32_t)(3 * depth) * fstStride;
            fd[0] = factor.x; fd[fstStride] = factor.y; fd[2u * fstStride] = factor.z;
            depth++;
            O = I + nv * CRT_EPS; D = nv; inside = newInside;
        }
        // unwind the recursion (innermost factor first), store the sample
#pragma unroll
        for (int k = 4; k >= 0; k--)
            if (depth > k) { const float* fd = fst + (uint32_t)(3 * k) * fstStride; L = mk3(fd[0], fd[fstStride], fd[2u * fstStride]) * L; }
        if (!PROBE) {
            uint32_t pass = 0;
            if (passes != 1u) pass = item - pix * passes;
            slab[((size_t)tl * 256u + pix) * (64u * passes) + ((laneBase + lane) * passes + pass)] = make_float4(L.x, L.y, L.z, 0.0f);
        } else if (L.x != L.x) steps++;                                            // (keeps the probe's shading arithmetic alive)
    }
    if (PROBE) { const uint32_t sum = wave_sum(steps); if (lane == 0) atomicAdd(&tileCost[tl], sum); return; }
    // what this tile cost (100 MHz wall clock ticks; the longest of its wavefronts): the host's latency mode sizes the next launch's wavefronts with it
    if (tileCost && lane == 0) atomicMax(&tileCost[tl], (uint32_t)(wall_clock64() - clk0));
    atomicAdd(&counters->v[0], (unsigned long long)nRays);
    atomicAdd(&counters->v[1], (unsigned long long)nPrimary);
    if (nMesh) atomicAdd(&counters->v[7], (unsigned long long)nMesh);
}

} // namespace crt

extern "C" uint32_t crt_narrow_max_lanes(void) { return crt::kNarrowMaxLanes; }

// blockDesc / nBlocks: block table of wavefronts of <= 8 lanes (format of render_tiles_kernel); ldsTiles: that kernel's traversal-stack bytes (two-level scenes use its column layout)
extern "C" hipError_t crt_launch_render_narrow(const crt::Scene* sc, void* slab, crt::Counters* counters, uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX,
                                               uint32_t sppFirst, uint32_t frames, uint32_t passes, uint32_t ldsTiles, const uint32_t* blockDesc, uint32_t nBlocks, uint32_t* tileCost, hipStream_t stream)
{
    if (tileCount == 0 || frames == 0 || nBlocks == 0 || !blockDesc) return hipSuccess;
    (void)ldsTiles;
    const crt::AltAccelDev none{};
    dim3 grid((nBlocks + crt::kNarrowWaves - 1u) / crt::kNarrowWaves), block(64u * crt::kNarrowWaves);
    if (sc->kind == 0) {
        const uint32_t ldsBytes = sc->topCount * 64u + crt::kNarrowWaves * crt::kNarrowMaxLanes * crt::narrow_lane_dwords(sc->stackDepth) * 4u;
        hipLaunchKernelGGL((crt::render_narrow_kernel<0, 0>), grid, block, ldsBytes, stream, *sc, none, (float4*)slab, counters, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes, blockDesc, nBlocks, tileCost);
    } else
        hipLaunchKernelGGL((crt::render_narrow_kernel<1, 0>), grid, block, crt::kNarrowWaves * (sc->stackDepth + 15u) * 64u * 4u, stream, *sc, none, (float4*)slab, counters, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes, blockDesc, nBlocks, tileCost);
    return hipGetLastError();
}

// the latency mode's cost probe: kProbeWaves wavefronts per owned tile, 64 paths each, step counts summed into tileCost[tile]
extern "C" uint32_t crt_probe_paths() { return 64u * crt::kProbeWaves; }
extern "C" hipError_t crt_launch_probe(const crt::Scene* sc, uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount, uint32_t tilesX, uint32_t* tileCost, hipStream_t stream)
{
    if (tileCount == 0 || !tileCost || tileCount > 0x10000u) return hipSuccess;
    const crt::AltAccelDev none{};
    const uint32_t nProbe = tileCount * crt::kProbeWaves;
    dim3 grid((nProbe + crt::kNarrowWaves - 1u) / crt::kNarrowWaves), block(64u * crt::kNarrowWaves);
    const uint32_t ldsBytes = crt::kNarrowWaves * (sc->stackDepth + 15u) * 64u * 4u;
    if (hipMemsetAsync(tileCost, 0, (size_t)tileCount * 4, stream) != hipSuccess) return hipGetLastError();
    if (sc->kind == 0) hipLaunchKernelGGL((crt::render_narrow_kernel<0, 1>), grid, block, ldsBytes, stream, *sc, none, (float4*)nullptr, (crt::Counters*)nullptr, tileFirst, tileStride, tileCount, tilesX, 1u, 64u, 1u, (const uint32_t*)nullptr, nProbe, tileCost);
    else hipLaunchKernelGGL((crt::render_narrow_kernel<1, 1>), grid, block, ldsBytes, stream, *sc, none, (float4*)nullptr, (crt::Counters*)nullptr, tileFirst, tileStride, tileCount, tilesX, 1u, 64u, 1u, (const uint32_t*)nullptr, nProbe, tileCost);
    return hipGetLastError();
}

// Renderer::Sample through FileScene's KD-tree (accel 1) or uniform grid (accel 2): one wavefront per (owned tile, 64-frame window) of the launch, lane = frame
extern "C" hipError_t crt_launch_render_alt(int accel, const crt::Scene* sc, const crt::AltAccelDev* acc, void* slab, crt::Counters* counters, uint32_t tileFirst, uint32_t tileStride, uint32_t tileCount,
                                            uint32_t tilesX, uint32_t sppFirst, uint32_t frames, uint32_t passes, hipStream_t stream)
{
    if (tileCount == 0 || frames == 0) return hipSuccess;
    if (sc->kind != 0 || (accel != 1 && accel != 2)) return hipErrorInvalidValue;
    const uint32_t windows = (frames + 63u) / 64u;
    if ((unsigned long long)tileCount * windows > 0x7fffffffull || tileCount > 0x10000u || windows > 64u) return hipErrorInvalidValue;
    const uint32_t entries = tileCount * windows;
    dim3 grid((entries + crt::kNarrowWaves - 1u) / crt::kNarrowWaves), block(64u * crt::kNarrowWaves);
    const uint32_t ldsBytes = crt::kNarrowWaves * (acc->kdStack * 2u + 15u) * 64u * 4u;
    if (ldsBytes > 64u * 1024u) return hipErrorInvalidValue;
    if (accel == 1) hipLaunchKernelGGL((crt::render_narrow_kernel<0, 2>), grid, block, ldsBytes, stream, *sc, *acc, (float4*)slab, counters, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes, (const uint32_t*)nullptr, entries, (uint32_t*)nullptr);
    else hipLaunchKernelGGL((crt::render_narrow_kernel<0, 3>), grid, block, ldsBytes, stream, *sc, *acc, (float4*)slab, counters, tileFirst, tileStride, tileCount, tilesX, sppFirst, frames, passes, (const uint32_t*)nullptr, entries, (uint32_t*)nullptr);
    return hipGetLastError();
}
