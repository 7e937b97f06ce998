/*
 * crt_oracle.h — C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a single-file CPU restatement of the reference's path-tracing hot path
 * (willake/cpu-ray-tracer: Renderer::Tick -> ProcessTile -> Sample -> FindNearest -> BVH/TLAS).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (cpu-ray-tracer_amd/) never links, imports or calls anything declared here.
 *
 * Pinning status is documented at the top of crt_oracle.cpp and in DESIGN.md.
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* reference layouts (sizeof Tri = 112, BVHNode = 32, TLASBVHNode = 32) */
typedef struct orc_bvh_node { float aabbMin[3], aabbMax[3]; uint32_t leftFirst, triCount; } orc_bvh_node;
typedef struct orc_tlas_node { float aabbMin[3]; uint32_t leftRight; float aabbMax[3]; uint32_t BLAS; } orc_tlas_node;
typedef struct orc_tri {
    float vertex0[3], vertex1[3], vertex2[3];
    float normal0[3], normal1[3], normal2[3];
    float uv0[2], uv1[2], uv2[2];
    float centroid[3];
    int32_t objIdx;
} orc_tri;

typedef struct orc_ray_in { float O[3]; float D[3]; int32_t inside; } orc_ray_in;
typedef struct orc_hit {
    float t; float u, v; int32_t objIdx; int32_t triIdx;
    int32_t traversed;   /* node iterations (TLAS + BLAS), reference `ray.traversed` */
    int32_t tested;      /* triangle tests summed over the whole query (NOT the reference's per-BLAS-reset `ray.tested`) */
} orc_hit;

typedef struct orc_counters {
    uint64_t rays;            /* FindNearest calls (primary + secondary) */
    uint64_t primary;         /* primary rays */
    uint64_t interior_iters;  /* I: interior-node iterations of mesh BVHs */
    uint64_t leaf_iters;      /* leaf iterations of mesh BVHs */
    uint64_t tri_tests;       /* T */
    uint64_t tlas_iters;      /* TLAS node iterations (interior + leaf) */
    uint64_t blas_visits;     /* V */
    uint64_t mesh_hits;       /* H: rays whose nearest hit is a mesh triangle */
} orc_counters;

/* kind: 0 = FileScene (one BVH over all world-space triangles), 1 = TLASFileScene (BLAS per object + TLAS) */
orc_ctx* orc_create(int kind);
void     orc_destroy(orc_ctx*);
const char* orc_last_error(orc_ctx*);

/* scene description (what LoadSceneFile + tinyobj + stb_image hand to the scene ctor) */
int orc_set_light_position(orc_ctx*, const float pos[3]);
int orc_set_floor_texture(orc_ctx*, const uint32_t* rgb, int w, int h);   /* 0x00RRGGBB */
int orc_set_skydome(orc_ctx*, const uint32_t* rgb, int w, int h);
int orc_add_material(orc_ctx*, float reflectivity, float refractivity, const float absorption[3],
                     const uint32_t* tex_rgb, int w, int h);              /* tex_rgb NULL = no texture */
/* corners: n_corners (multiple of 3) un-indexed vertices in face order, as tinyobj's shape.mesh.indices resolve them;
 * nrm/uv may be NULL (-> zeros, as `Vertex vertex{}`) */
int orc_add_object(orc_ctx*, const float* pos, const float* nrm, const float* uv, int n_corners,
                   const float position[3], const float rotation_deg[3], const float scale[3], int material_idx);
int orc_build(orc_ctx*);

/* introspection of the built structures */
int orc_bvh_count(orc_ctx*);
int orc_bvh_info(orc_ctx*, int bvh, uint32_t* nodesUsed, uint32_t* triCount, uint32_t* maxDepth);
int orc_bvh_copy(orc_ctx*, int bvh, orc_bvh_node* nodes /*nodesUsed*/, uint32_t* triIndices /*triCount*/, orc_tri* tris /*triCount*/);
int orc_set_blas_transform(orc_ctx*, int bvh, const float T[16]);   /* BLASBVH::SetTransform + TLASBVH::Build */
int orc_bvh_move_and_refit(orc_ctx*, int bvh, const float* positions /* 9 floats per triangle */, uint32_t triCount);   /* BVH::Refit, bvh.cpp:26-43 */
int orc_blas_transform(orc_ctx*, int bvh, float T[16], float invT[16], float worldMin[3], float worldMax[3]);
int orc_tlas_copy(orc_ctx*, orc_tlas_node* nodes /*2*blasCount*/, uint32_t* nodesUsed);

/* renderer */
int orc_renderer_init(orc_ctx*, int width, int height);        /* Renderer::Init + default Camera() */
int orc_set_camera_state(orc_ctx*, const float pos[3], const float target[3]);
int orc_get_camera(orc_ctx*, float camPos[3], float topLeft[3], float topRight[3], float bottomLeft[3]);
int orc_primary_rays(orc_ctx*, const float* xy /* 2n: pixel coordinates incl. jitter */, size_t n, float* O /* 3n */, float* D /* 3n */);   /* Camera::GetPrimaryRay */
int orc_texture_sample(const uint32_t* texels, int w, int h, const float* uv /* 2n */, size_t n, float* rgb /* 3n */);                        /* Texture::Sample */
int orc_set_params(orc_ctx*, int depthLimit, int passes);
int orc_clear(orc_ctx*);                                       /* ClearAccumulator + spp = 1 */
int orc_set_spp(orc_ctx*, int spp);
int orc_get_spp(orc_ctx*);
int orc_set_tile_range(orc_ctx*, int first_tile, int tile_count); /* multi-GPU parity: only these tiles are rendered (count<0 = all) */
int orc_tick(orc_ctx*, int n_threads);                         /* one Renderer::Tick */
int orc_render(orc_ctx*, int frames, int n_threads);           /* frames x Tick */
const float* orc_accumulator(orc_ctx*);                        /* float4[W*H] */
const uint32_t* orc_screen(orc_ctx*);                          /* 0x00RRGGBB [W*H] */
float orc_energy(orc_ctx*);
int orc_get_counters(orc_ctx*, orc_counters*);
int orc_reset_counters(orc_ctx*);
int orc_tile_seed_after_frame(orc_ctx*, int spp, int tile, uint32_t* seed_out); /* RNG state after ProcessTile(tile) at `spp` */

/* query entry = scene.FindNearest */
int orc_find_nearest(orc_ctx*, const orc_ray_in* rays, orc_hit* hits, size_t n);
/* Renderer::Sample for one ray; seed is in/out */
int orc_sample(orc_ctx*, const orc_ray_in* ray, uint32_t* seed, float rgb[3]);

/* Whitted integrator (2. WhittedStyle/renderer.cpp) — CPU plumbing for BASELINE config #1 */
int orc_whitted_render(orc_ctx*, int n_threads);               /* one frame into accumulator (xyz) */

/* helper for the oracle-side PNG reader (oracle/orc.py) */
int orc_png_unfilter(const uint8_t* raw, uint8_t* out, int stride, int h, int fb);

/* deterministic math used for absorption / skydome lookups (see DESIGN.md "numerics") */
/* alternative accelerators of FileScene (infra/kdtree.cpp, infra/grid.cpp), standalone over a triangle array; flat pre-order KD nodes */
typedef struct orc_kd_node { float aabbMin[3]; int32_t left; float aabbMax[3]; int32_t right; float splitDistance; int32_t splitAxis; uint32_t firstTri, triCount; } orc_kd_node;   /* left < 0: leaf */
void* orc_kd_build(const orc_tri* tris, uint32_t n);
void orc_kd_info(void* h, uint32_t* nodes, uint32_t* refs, uint32_t* maxDepth, uint32_t* nodesUsed);
void orc_kd_dump(void* h, orc_kd_node* nodes, uint32_t* refs);
void orc_kd_intersect(void* h, const float* O, const float* D, uint32_t n, orc_hit* out);
void orc_kd_free(void* h);
void* orc_grid_build(const orc_tri* tris, uint32_t n);
void orc_grid_info(void* h, int32_t res[3], float cell[3], float lo[3], float hi[3], uint32_t* refs);
void orc_grid_dump(void* h, uint32_t* cellStart, int32_t* refs);
void orc_grid_intersect(void* h, const float* O, const float* D, uint32_t n, orc_hit* out);
void orc_grid_free(void* h);
/* Sample / Trace through the KD-tree (1) or grid (2) of a FileScene instead of its BVH (file_scene.h:10-12): h = the structure built over the scene's triangles */
int orc_set_render_accel(orc_ctx*, int kind, void* h);
/* PrimitiveScene: orc_create(2); the wall images; animation time; state dump (6 matrices, sphere position, torus radii, cube box: 108 floats) */
int orc_prim_setup(orc_ctx*, const uint32_t* red512, const uint32_t* blue512);
int orc_prim_set_time(orc_ctx*, float t);
int orc_prim_state(orc_ctx*, float* out108);
double orc_det_acos(double x);
double orc_det_cos(double x);
void orc_math_probe(const float* in12, uint32_t n, float* out120);
uint32_t orc_vertex_dedup(const float* v8, uint32_t n, uint32_t* idx, float* unique8);
float orc_expf(float x);
float orc_atan2f(float y, float x);
float orc_acosf(float x);
/* RNG primitives */
uint32_t orc_init_seed(uint32_t base);
uint32_t orc_random_uint(uint32_t* seed);

#ifdef __cplusplus
}
#endif
#endif
