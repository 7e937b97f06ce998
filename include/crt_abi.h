/*
 * crt_abi.h — C ABI of the MI355X path-tracing back end (libcrt_amd.so).
 *
 * The reference (willake/cpu-ray-tracer) has no plugin/FFI seam: the path tracer is compiled into the
 * application.  The two seams it does have are
 *     upper:  TheApp::Init / Tick(float) and the public Renderer members (accumulator, spp, passes, energy, …)
 *             — template/precomp.h:344-361, "3. PathTracer/renderer.h":27-53
 *     lower:  BaseScene::FindNearest / GetHitInfo / GetSkyColor and the public members of the accel classes
 *             (bvhNodes, triangles, triangleIndices, nodesUsed, T, invT, blas) — infra/scene/base_scene.h:16-32,
 *             infra/bvh.h:37-43, infra/blas_bvh.h:48-57, infra/tlas_bvh.h:27-31
 * This header is what a binding behind those seams calls.  The host application keeps loading scenes and
 * building the SAH-BVH / TLAS on the CPU exactly as today; it hands the BUILT arrays (reference layouts,
 * borrowed pointers, copied during the call) to crt_upload_scene once, and replaces the body of
 * Renderer::Tick's tile loop by crt_render.  INTEGRATION.md shows the binding.
 *
 * Conventions: every function returns 0 on success or a negative crt_status; no function throws or exits
 * (the reference's FatalError/exit and std::runtime_error — template/opencl.cpp:14-27, infra/blas_bvh.cpp:11-14 —
 * become error codes + crt_last_error).  Plain pointers and sizes only; no C++ or torch types.
 * One host thread drives one ctx; work is asynchronous on the ctx's HIP stream until crt_sync / a read.
 */
#ifndef CRT_ABI_H
#define CRT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_ABI_VERSION 3

typedef enum crt_status {
    CRT_OK = 0,
    CRT_ERR_INVALID = -1,      /* bad argument / inconsistent scene description            */
    CRT_ERR_DEVICE = -2,       /* HIP runtime error (message in crt_last_error)            */
    CRT_ERR_NO_DEVICE = -3,    /* no gfx950 device visible: the product never falls back to a CPU path */
    CRT_ERR_UNSUPPORTED = -4,  /* valid input outside this build's limits (stated in the message) */
    CRT_ERR_STATE = -5,        /* call order (e.g. render before upload)                   */
    CRT_ERR_IO = -6            /* host loader: file missing / malformed                    */
} crt_status;

typedef struct crt_ctx crt_ctx;

/* ---- reference record layouts (bit-compatible with the reference structs) ------------------------- */

/* BVHNode — infra/blas_bvh.h:13-20 (32 bytes).  Leaf iff triCount > 0; children at leftFirst, leftFirst+1. */
typedef struct crt_bvh_node { float aabbMin[3], aabbMax[3]; uint32_t leftFirst, triCount; } crt_bvh_node;

/* Tri — infra/helper.h:6-26 (112 bytes AoS). */
typedef struct crt_tri {
    float vertex0[3], vertex1[3], vertex2[3];
    float normal0[3], normal1[3], normal2[3];
    float uv0[2], uv1[2], uv2[2];
    float centroid[3];
    int32_t objIdx;
} crt_tri;

/* TLASBVHNode — infra/tlas_bvh.h:7-14 (32 bytes).  Leaf iff leftRight == 0; children = lo/hi 16 bits. */
typedef struct crt_tlas_node { float aabbMin[3]; uint32_t leftRight; float aabbMax[3]; uint32_t BLAS; } crt_tlas_node;

/* Texture — template/texture.h:15-48: 0x00RRGGBB texels, row 0 = top of the image. */
typedef struct crt_texture { const uint32_t* pixels; int32_t width, height; } crt_texture;

/* Material — template/material.h:6-46.  texture = index into crt_scene_desc.textures or -1. */
typedef struct crt_material { float reflectivity, refractivity; float absorption[3]; int32_t texture; } crt_material;

/* One acceleration structure: BVH (infra/bvh.h) for CRT_SCENE_FILE, BLASBVH (infra/blas_bvh.h) for CRT_SCENE_TLAS. */
typedef struct crt_bvh {
    const crt_bvh_node* nodes;         /* bvhNodes.data()                                              */
    uint32_t nodesUsed;                /* nodesUsed (root = 0, node 1.. allocated pairwise)            */
    const crt_tri* triangles;          /* triangles.data()                                             */
    uint32_t triCount;                 /* triangles.size()                                             */
    const uint32_t* triangleIndices;   /* triangleIndices.data()                                       */
    int32_t objIdx;                    /* BLASBVH::objIdx (hit id written by IntersectTri); ignored for CRT_SCENE_FILE */
    int32_t matIdx;                    /* BLASBVH::matIdx; ignored for CRT_SCENE_FILE                  */
    float T[16], invT[16];             /* BLASBVH::T / invT, row-major mat4; ignored for CRT_SCENE_FILE */
} crt_bvh;

typedef enum crt_scene_kind { CRT_SCENE_FILE = 0 /* FileScene, USE_BVH */, CRT_SCENE_TLAS = 1 /* TLASFileScene, TLAS_USE_BVH */ } crt_scene_kind;

typedef struct crt_scene_desc {
    int32_t kind;
    const crt_bvh* bvhs; uint32_t bvhCount;              /* FILE: exactly 1 (FileScene::acc); TLAS: tlas.blas[]            */
    const crt_tlas_node* tlasNodes; uint32_t tlasNodeCount; /* TLAS only: tlasNode[0 .. 2*blasCount)                         */
    const int32_t* objMatIdx; uint32_t objCount;         /* FILE only: models[i]->matIdx for object id i+2                 */
    const crt_material* materials; uint32_t materialCount;
    const crt_texture* textures; uint32_t textureCount;
    int32_t floorTexture;                                /* primitiveMaterials[1].textureDiffuse (index into textures)      */
    int32_t skyTexture;                                  /* skydome                                                          */
    float lightT[16], lightInvT[16], lightSize;          /* Quad light (template/primitives.h:321-375): T, invT, size       */
    float floorN[3], floorD, floorInvto;                 /* Plane floor (primitives.h:100-179): N, d, invto                 */
} crt_scene_desc;

typedef struct crt_config {
    int32_t width, height;       /* SCRWIDTH / SCRHEIGHT (template/camera.h:4-5)                                      */
    int32_t depthLimit;          /* Renderer::depthLimit (renderer.h:53), default 5                                   */
    int32_t device;              /* HIP device ordinal                                                                  */
    /* image tiles owned by this ctx: tile = tileFirst + i*tileStride, i in [0, tileCount); tileCount < 0 = all.
     * Tiles are numbered x-major as in Renderer::Tick (renderer.cpp:151-152).  Other pixels are never touched. */
    int32_t tileFirst, tileStride, tileCount;
    int32_t maxFramesPerLaunch;  /* frames rendered per kernel launch; 0 = default (4096 = 64 windows of 64 frames, one wavefront
                                    per (tile, window)); values > 64 are rounded down to whole windows; the sample-slab pool
                                    (at most half of the free HBM) may lower it                                            */
    int32_t collectStats;        /* !=0: kernels also count node iterations / triangle tests / BLAS visits / mesh hits   */
    int32_t renderStreams;       /* HIP streams the render launches rotate over, so that independent launches (consecutive crt_render
                                    calls, or the pieces of one long call) overlap on the GPU; 0 = default (7), 1 = one at a time */
} crt_config;

typedef struct crt_ray { float O[3]; float D[3]; int32_t inside; } crt_ray;
typedef struct crt_hit {
    float t; float u, v; int32_t objIdx; int32_t triIdx;
    int32_t traversed;   /* node iterations, as Ray::traversed (bvh.cpp:231, tlas_bvh.cpp:89)                      */
    int32_t tested;      /* triangle tests over the whole query                                                     */
} crt_hit;

typedef struct crt_counters {
    uint64_t rays;            /* FindNearest calls = primary + secondary rays                                       */
    uint64_t primary;
    uint64_t interior_iters;  /* I   (valid when collectStats)                                                      */
    uint64_t leaf_iters;
    uint64_t tri_tests;       /* T                                                                                   */
    uint64_t tlas_iters;
    uint64_t blas_visits;     /* V                                                                                   */
    uint64_t mesh_hits;       /* H                                                                                   */
} crt_counters;

typedef struct crt_timing {                /* covers every launch since the previous crt_get_timing call                       */
    float render_kernel_ms;   /* Σ duration of the path-tracing kernel launches (HIP events on the stream each was launched on; launches on
                                 different render streams overlap, so this sum can exceed wall time)                 */
    float resolve_kernel_ms;  /* Σ duration of the ordered accumulate kernels                                        */
    uint32_t render_launches; /* number of path-tracing kernel launches                                              */
    uint32_t pool_launches;   /* ... of which render_pool_kernel (stream pool; the others are render_tiles_kernel)   */
    uint32_t split_launches;  /* ... of which split: the most expensive tiles by a concurrent render_tiles_kernel    */
    uint32_t reserved;
} crt_timing;

/* ---- life cycle ----------------------------------------------------------------------------------- */
int  crt_abi_version(void);
int  crt_device_count(void);                                   /* number of visible HIP devices (0 = none)          */
int  crt_create(crt_ctx** out, const crt_config* cfg);
void crt_destroy(crt_ctx* ctx);
const char* crt_last_error(crt_ctx* ctx);                      /* ctx may be NULL: error of the last failed crt_create on this thread */

/* ---- scene / camera (lower seam) -------------------------------------------------------------------- */
int  crt_upload_scene(crt_ctx* ctx, const crt_scene_desc* scene);   /* flattens to the device layout and copies; host pointers are not kept */
/* In-place update of an uploaded scene for animation (renderer.cpp:147 `animating`), same description as at upload, same topology:
 *   CRT_UPDATE_TRANSFORMS  two-level scenes: every BLAS's T / invT as BLASBVH::SetTransform left them (blas_bvh.cpp:363-374) and the node array of the
 *                          TLASBVH::Build that followed (tlas_bvh.cpp:17-55) — instance motion;
 *   CRT_UPDATE_BOUNDS      every BVH's node boxes and triangle vertices as BVH::Refit / BLASBVH::Refit left them (bvh.cpp:26-43; node / triangle counts,
 *                          child indices and triangleIndices unchanged) and, for two-level scenes, the rebuilt TLAS.
 * Only the affected sections of the device geometry buffer are rewritten (a few KB for transforms): no device allocation, no host wait for the GPU;
 * frames submitted earlier still see the old scene, frames submitted later the new one. */
#define CRT_UPDATE_TRANSFORMS 1u
#define CRT_UPDATE_BOUNDS     2u
int  crt_update_scene(crt_ctx* ctx, const crt_scene_desc* scene, uint32_t what);
int  crt_set_camera(crt_ctx* ctx, const float camPos[3], const float topLeft[3], const float topRight[3], const float bottomLeft[3]);
                                                                /* Camera members used by GetPrimaryRay (camera.h:23-30) */

/* ---- rendering (upper seam: the tile loop of Renderer::Tick) --------------------------------------- */
/* Renders `frames` consecutive Ticks: frame k uses spp = spp_first + k*passes for its tile seeds
 * (renderer.cpp:120,167) and adds passes samples per pixel into the accumulator in frame order.
 * Asynchronous: the call is cut into launches of up to cfg.maxFramesPerLaunch frames (one grid covers all their 64-frame windows);
 * launches rotate over cfg.renderStreams HIP streams and overlap with those of earlier crt_render calls; the accumulation order
 * is kept by events.  crt_sync / any read waits for everything. */
int  crt_render(crt_ctx* ctx, uint32_t spp_first, uint32_t frames, uint32_t passes);
int  crt_sync(crt_ctx* ctx);
/* Optional: sizes the device-side sample-slab pool for an upcoming crt_render(.., frames, passes) now (allocating tens of GB takes
 * seconds and synchronises the device; without this call the first crt_render that needs a larger pool pays for it). */
int  crt_reserve(crt_ctx* ctx, uint32_t frames, uint32_t passes);
int  crt_clear(crt_ctx* ctx);                                   /* Renderer::ClearAccumulator (renderer.cpp:15-18)    */
int  crt_read_accumulator(crt_ctx* ctx, float* host_rgba /* float4[width*height] */);
/* screen->pixels and Renderer::energy as ProcessTile/Tick leave them (renderer.cpp:119,127-129,155-157):
 * pixel = accumulator * scale, scale = 1/(spp+passes) of the LAST rendered frame.  Either output may be NULL. */
int  crt_resolve_screen(crt_ctx* ctx, float scale, uint32_t* host_pixels /* width*height */, float* energy);

/* ---- Whitted-style integrator ("2. WhittedStyle/renderer.cpp":21-157): one deterministic Tick -------------------
 * Every pixel of the image (rows are not tile-truncated in this renderer) gets accumulator = float4(Trace(primary), 0)
 * and screen pixel = RGBF32_to_RGB8 of it; host_pixels may be NULL.  Synchronous. */
int  crt_whitted_tick(crt_ctx* ctx, uint32_t* host_pixels /* width*height or NULL */);

/* ---- query entry = scene.FindNearest(ray) ------------------------------------------------------------ */
int  crt_find_nearest(crt_ctx* ctx, const crt_ray* rays, crt_hit* hits, size_t n);

/* ---- FileScene's alternative accelerators: KDTree (infra/kdtree.cpp — the one the reference ships enabled, infra/scene/file_scene.h:10-12)
 * and Grid (infra/grid.cpp), built on the host exactly as there and attached to an uploaded CRT_SCENE_FILE scene.  The reference's KDTreeNode is
 * pointer-linked with a std::vector per node (infra/blas_kdtree.h:15-24), so there is no layout to be bit-compatible with: nodes are passed flattened in
 * PRE-ORDER (node, left subtree, right subtree), leaves naming a range of kdTriIndices; Grid's cells (x-major: ix + iy*rx + iz*rx*ry) as a prefix array.
 * crt_find_nearest_alt = scene.FindNearest with that accelerator in place of the BVH (light quad, floor plane, accelerator: file_scene.cpp:170-175);
 * crt_hit.traversed / tested count as Ray::traversed / Ray::tested do there.  The render kernels walk the SAH-BVH only. */
#define CRT_ACCEL_KDTREE 1
#define CRT_ACCEL_GRID   2
typedef struct crt_kd_node {
    float aabbMin[3]; int32_t left;          /* KDTreeNode::aabbMin; index of node->left, < 0 = leaf (isLeaf)                    */
    float aabbMax[3]; int32_t right;
    float splitDistance; int32_t splitAxis;  /* interior: splitPos = aabbMin[splitAxis] + splitDistance (kdtree.cpp:161-162)       */
    uint32_t firstTri, triCount;             /* leaf: triIndices = kdTriIndices[firstTri .. firstTri + triCount)                    */
} crt_kd_node;
typedef struct crt_alt_accel {
    int32_t kind;                                                    /* CRT_ACCEL_KDTREE or CRT_ACCEL_GRID                          */
    const crt_tri* triangles; uint32_t triCount;                     /* KDTree::triangles / Grid::triangles (= FileScene's triangle array) */
    const crt_kd_node* kdNodes; uint32_t kdNodeCount;                /* KD-tree: pre-order nodes, root = 0                          */
    const uint32_t* kdTriIndices; uint32_t kdTriIndexCount;
    int32_t gridResolution[3]; float gridCellSize[3];                /* Grid::resolution, cellSize                                  */
    float gridMin[3], gridMax[3];                                    /* Grid::localBounds                                           */
    const uint32_t* gridCellStart;                                   /* rx*ry*rz + 1 entries: cell c holds gridCellTris[start[c] .. start[c+1]) */
    const int32_t* gridCellTris; uint32_t gridCellTriCount;
} crt_alt_accel;
int  crt_upload_alt_accel(crt_ctx* ctx, const crt_alt_accel* accel);   /* after crt_upload_scene of a CRT_SCENE_FILE scene; one structure per kind is kept */
int  crt_find_nearest_alt(crt_ctx* ctx, int kind, const crt_ray* rays, crt_hit* hits, size_t n);
/* ABI 3: which structure crt_render (Renderer::Sample) and crt_whitted_tick (Renderer::Trace, incl. its shadow rays) trace through: 0 = the scene's BVH / TLAS
 * (default), CRT_ACCEL_KDTREE / CRT_ACCEL_GRID = the uploaded alternative accelerator — what the reference's FileScene does when built with USE_KDTree (its shipped
 * setting, infra/scene/file_scene.h:10-12, file_scene.cpp:170-187) / USE_Grid.  Bug-compatible: the KD traversal loses hits for rays with a direction component of
 * exactly 0 (kdtree.cpp:161-201).  The sequential form (one wavefront per tile and 64-frame window); reset by crt_upload_scene / crt_upload_alt_accel of the kind. */
int  crt_set_render_accel(crt_ctx* ctx, int kind);

/* ---- PrimitiveScene (second half): infra/scene/primitive_scene.cpp — the reference's hard-coded demo room (six walls, swinging light quad,
 * bouncing mirror ball, "rounded corners" sphere, spinning glass cube, glass torus; template/primitives.h Sphere :31, Cube :187, Quad :321, Torus :380; the
 * SPEEDTRIX / single-light configuration its headers select).  The binding passes the scene as its constructor + SetTime(t) leave it: the members below.
 * After crt_upload_primitive_scene, crt_render (Renderer::Sample), crt_find_nearest (objIdx 0 .. 10, u = v = 0, triIdx = -1) and the accumulator entry points
 * work on this scene (it replaces an uploaded triangle scene; crt_whitted_tick, crt_update_scene and the alternative accelerators do not apply).
 * PARITY UNPINNED: checked bit for bit against the repo's oracle only (the reference files need MSVC); the torus' double-precision cos(acos(x) / 3) is a
 * deterministic fdlibm-style evaluation on both sides, so its hit distances can differ from a Windows build of the reference in the last place. */
typedef struct crt_primitive_scene {
    float quadT[16], quadInvT[16]; float quadSize;             /* Quad quad: T, invT (FastInvertedTransformNoScale), size (= 0.5)            */
    float spherePos[3];                                        /* Sphere sphere (r = 0.6): pos; sphere2 is constant (0, 2.5, -3.07), r = 8      */
    float cubeMin[3], cubeMax[3], cubeM[16], cubeInvM[16];     /* Cube cube: b[0], b[1], M, invM                                               */
    float torusT[16], torusInvT[16];                           /* Torus torus: T, invT (mat4::Inverted)                                        */
    float torusRt2, torusRc2, torusR2;                         /*              rt2, rc2, r2                                                     */
    float reflectivity[11], refractivity[11], absorption[11][3];   /* Material materials[11] (isLight: index 0; isAlbedoOverridden: 4, 5, 6)   */
    crt_texture red, blue;                                     /* Plane::GetAlbedo's "../assets/red.png" / "blue.png" as Surface loads them (0x00RRGGBB, 512 x 512); pixels may be NULL */
} crt_primitive_scene;
int  crt_upload_primitive_scene(crt_ctx* ctx, const crt_primitive_scene* scene);

/* ---- instrumentation ---------------------------------------------------------------------------------- */
int  crt_get_counters(crt_ctx* ctx, crt_counters* out);        /* cumulative since create / crt_reset_counters       */
int  crt_reset_counters(crt_ctx* ctx);
int  crt_get_timing(crt_ctx* ctx, crt_timing* out);            /* syncs the stream                                   */
/* collectStats contexts only: for each owned tile i of the LAST render launch, out[2i] = wall time of the tile's
 * wavefront and out[2i+1] = its start stamp, both in ticks of the 100 MHz constant clock (load-balance map). */
int  crt_get_tile_clocks(crt_ctx* ctx, uint64_t* out /* 2 * tileCount */);

/* ---- multi-GPU plumbing ---------------------------------------------------------------------------------
 * The accumulator can live in caller-owned device memory (e.g. a torch tensor that torch.distributed/RCCL
 * reduces over xGMI).  Must be width*height*16 bytes, 16-byte aligned, on cfg.device.  NULL = back to internal. */
int  crt_bind_accumulator(crt_ctx* ctx, void* device_ptr);
int  crt_accumulator_device_ptr(crt_ctx* ctx, void** device_ptr);

#ifdef __cplusplus
}
#endif
#endif
