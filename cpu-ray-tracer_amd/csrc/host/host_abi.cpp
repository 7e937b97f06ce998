// host_abi.cpp — C wrappers (include/crt_host.h) around the C++ host front.  Exceptions stop here.
#include "../../../include/crt_host.h"
#include "scene.h"
#include "accel_alt.h"

#include <cstdlib>
#include <string>

using namespace crt;

namespace { thread_local std::string g_err; }

struct crt_host_scene { BaseScene* scene = nullptr; FileScene* file = nullptr; TLASFileScene* tlas = nullptr; KDTree* kd = nullptr; Grid* grid = nullptr; };
struct crt_host_renderer { Renderer* r = nullptr; };

#define GUARD_BEGIN try {
#define GUARD_END(code) } catch (const std::exception& e) { g_err = e.what(); return code; } catch (...) { g_err = "unknown exception"; return code; }

extern "C" {

const char* crt_host_last_error(void) { return g_err.c_str(); }
void crt_host_set_error(const char* msg) { g_err = msg ? msg : ""; }      // (other translation units of the host front: primitive_scene.cpp)

int crt_host_scene_load(const char* xml, int kind, const char* base, crt_host_scene** out)
{
    if (!xml || !out) { g_err = "null argument"; return CRT_ERR_INVALID; }
    *out = nullptr;
    if (kind != CRT_SCENE_FILE && kind != CRT_SCENE_TLAS) { g_err = "unknown scene kind"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    crt_host_scene* h = new crt_host_scene();
    try {
        const std::string b = base ? base : "";
        if (kind == CRT_SCENE_FILE) { h->file = new FileScene(xml, b); h->scene = h->file; }
        else { h->tlas = new TLASFileScene(xml, b); h->scene = h->tlas; }
    } catch (...) { delete h; throw; }
    *out = h;
    return CRT_OK;
    GUARD_END(CRT_ERR_IO)
}
void crt_host_scene_free(crt_host_scene* s) { if (s) { delete s->scene; delete s->kd; delete s->grid; delete s; } }
int crt_host_scene_upload(crt_host_scene* s, crt_ctx* ctx)
{
    if (!s || !ctx) { g_err = "null argument"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    const int rc = s->scene->Upload(ctx);
    if (rc != CRT_OK) g_err = crt_last_error(ctx);
    return rc;
    GUARD_END(CRT_ERR_INVALID)
}
int crt_host_scene_kind(crt_host_scene* s) { return s ? s->scene->Kind() : CRT_ERR_INVALID; }
int crt_host_scene_triangle_count(crt_host_scene* s) { return s ? s->scene->GetTriangleCount() : CRT_ERR_INVALID; }
int crt_host_scene_bvh_count(crt_host_scene* s) { return !s ? CRT_ERR_INVALID : (s->file ? 1 : (int)s->tlas->tlas.blas.size()); }

static bool pick(crt_host_scene* s, int i, const std::vector<BVHNode>** nodes, const std::vector<Tri>** tris, const std::vector<uint32_t>** idx, uint32_t* used, uint32_t* depth)
{
    if (!s) return false;
    if (s->file) { if (i != 0) return false; *nodes = &s->file->acc.bvhNodes; *tris = &s->file->acc.triangles; *idx = &s->file->acc.triangleIndices; *used = s->file->acc.nodesUsed; *depth = s->file->acc.maxDepth; return true; }
    if (i < 0 || i >= (int)s->tlas->tlas.blas.size()) return false;
    const BLASBVH* b = s->tlas->tlas.blas[(size_t)i];
    *nodes = &b->bvhNodes; *tris = &b->triangles; *idx = &b->triangleIndices; *used = b->nodesUsed; *depth = b->maxDepth; return true;
}
int crt_host_scene_bvh_info(crt_host_scene* s, int i, uint32_t* nodesUsed, uint32_t* triCount, uint32_t* maxDepth)
{
    const std::vector<BVHNode>* n; const std::vector<Tri>* t; const std::vector<uint32_t>* x; uint32_t u, d;
    if (!pick(s, i, &n, &t, &x, &u, &d)) { g_err = "bvh index out of range"; return CRT_ERR_INVALID; }
    if (nodesUsed) *nodesUsed = u; if (triCount) *triCount = (uint32_t)t->size(); if (maxDepth) *maxDepth = d;
    return CRT_OK;
}
int crt_host_scene_bvh_copy(crt_host_scene* s, int i, crt_bvh_node* nodes, uint32_t* idx, crt_tri* tris)
{
    const std::vector<BVHNode>* n; const std::vector<Tri>* t; const std::vector<uint32_t>* x; uint32_t u, d;
    if (!pick(s, i, &n, &t, &x, &u, &d)) { g_err = "bvh index out of range"; return CRT_ERR_INVALID; }
    if (nodes) memcpy(nodes, n->data(), sizeof(BVHNode) * u);
    if (idx) memcpy(idx, x->data(), 4 * x->size());
    if (tris) memcpy(tris, t->data(), sizeof(Tri) * t->size());
    return CRT_OK;
}
int crt_host_scene_bvh_move_and_refit(crt_host_scene* s, int i, const float* positions, uint32_t triCount)
{
    if (!s || !positions) { g_err = "null argument"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    std::vector<Tri>* tris = nullptr;
    BLASBVH* blas = nullptr;
    if (s->file) { if (i != 0) { g_err = "bvh index out of range"; return CRT_ERR_INVALID; } tris = &s->file->acc.triangles; }
    else { if (i < 0 || i >= (int)s->tlas->tlas.blas.size()) { g_err = "bvh index out of range"; return CRT_ERR_INVALID; } blas = s->tlas->tlas.blas[(size_t)i]; tris = &blas->triangles; }
    if (triCount != tris->size()) { g_err = "triangle count does not match the built BVH (Refit keeps the topology)"; return CRT_ERR_INVALID; }
    for (uint32_t t = 0; t < triCount; t++) {
        memcpy((*tris)[t].vertex0, positions + 9 * (size_t)t, 12); memcpy((*tris)[t].vertex1, positions + 9 * (size_t)t + 3, 12);
        memcpy((*tris)[t].vertex2, positions + 9 * (size_t)t + 6, 12);
    }
    if (s->file) s->file->acc.Refit();
    else { blas->Refit(); blas->SetTransform(blas->T); s->tlas->tlas.Build(); }      // world bounds + TLAS follow the refitted BLAS, as a per-frame animation loop does
    return CRT_OK;
    GUARD_END(CRT_ERR_INVALID)
}
// BLASBVH::SetTransform(T) of instance i (blas_bvh.cpp:363-374: T, invT = FastInvertedTransformNoScale, world bounds of the 8 root-box corners)
// followed by TLASBVH::Build (tlas_bvh.cpp:17-55), as an animation loop does per frame; crt_host_scene_update then moves it to the device
int crt_host_scene_set_transform(crt_host_scene* s, int i, const float T[16])
{
    if (!s || !T) { g_err = "null argument"; return CRT_ERR_INVALID; }
    if (!s->tlas || i < 0 || i >= (int)s->tlas->tlas.blas.size()) { g_err = "not a TLAS scene / index out of range (a FileScene bakes its transforms into the triangles)"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    mat4 m; memcpy(m.cell, T, 64);
    s->tlas->tlas.blas[(size_t)i]->SetTransform(m);
    s->tlas->tlas.Build();
    return CRT_OK;
    GUARD_END(CRT_ERR_INVALID)
}
int crt_host_scene_update(crt_host_scene* s, crt_ctx* ctx, uint32_t what)
{
    if (!s || !ctx) { g_err = "null argument"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    const int rc = s->scene->Update(ctx, what);
    if (rc != CRT_OK) g_err = crt_last_error(ctx);
    return rc;
    GUARD_END(CRT_ERR_DEVICE)
}
int crt_host_scene_blas_transform(crt_host_scene* s, int i, float T[16], float invT[16], float lo[3], float hi[3])
{
    if (!s || !s->tlas || i < 0 || i >= (int)s->tlas->tlas.blas.size()) { g_err = "not a TLAS scene / index out of range"; return CRT_ERR_INVALID; }
    const BLASBVH* b = s->tlas->tlas.blas[(size_t)i];
    memcpy(T, b->T.cell, 64); memcpy(invT, b->invT.cell, 64);
    lo[0] = b->worldBounds.bmin3.x; lo[1] = b->worldBounds.bmin3.y; lo[2] = b->worldBounds.bmin3.z;
    hi[0] = b->worldBounds.bmax3.x; hi[1] = b->worldBounds.bmax3.y; hi[2] = b->worldBounds.bmax3.z;
    return CRT_OK;
}
int crt_host_scene_tlas_copy(crt_host_scene* s, crt_tlas_node* nodes, uint32_t* nodesUsed)
{
    if (!s || !s->tlas) { g_err = "not a TLAS scene"; return CRT_ERR_INVALID; }
    if (nodes) memcpy(nodes, s->tlas->tlas.tlasNode.data(), sizeof(TLASBVHNode) * s->tlas->tlas.tlasNode.size());
    if (nodesUsed) *nodesUsed = s->tlas->tlas.nodesUsed;
    return CRT_OK;
}

int crt_host_camera_state(int w, int h, const float p[3], const float t[3], float camPos[3], float tl[3], float tr[3], float bl[3])
{
    if (w <= 0 || h <= 0 || !p || !t) { g_err = "bad argument"; return CRT_ERR_INVALID; }
    Camera c(w, h);
    c.SetCameraState(float3(p[0], p[1], p[2]), float3(t[0], t[1], t[2]));
    camPos[0] = c.camPos.x; camPos[1] = c.camPos.y; camPos[2] = c.camPos.z;
    tl[0] = c.topLeft.x; tl[1] = c.topLeft.y; tl[2] = c.topLeft.z;
    tr[0] = c.topRight.x; tr[1] = c.topRight.y; tr[2] = c.topRight.z;
    bl[0] = c.bottomLeft.x; bl[1] = c.bottomLeft.y; bl[2] = c.bottomLeft.z;
    return CRT_OK;
}

int crt_host_renderer_create(crt_host_scene* s, int w, int h, int device, crt_host_renderer** out)
{
    if (!s || !out || w < 16 || h < 16) { g_err = "bad argument"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    crt_host_renderer* r = new crt_host_renderer();
    r->r = new Renderer(s->scene, w, h, device);
    *out = r;
    return CRT_OK;
    GUARD_END(CRT_ERR_INVALID)
}
void crt_host_renderer_destroy(crt_host_renderer* r) { if (r) { delete r->r; delete r; } }
int crt_host_renderer_init(crt_host_renderer* r) { if (!r) return CRT_ERR_INVALID; GUARD_BEGIN r->r->Init(); return CRT_OK; GUARD_END(CRT_ERR_DEVICE) }
int crt_host_renderer_set_camera(crt_host_renderer* r, const float p[3], const float t[3])
{
    if (!r || !p || !t) return CRT_ERR_INVALID;
    r->r->camera.SetCameraState(float3(p[0], p[1], p[2]), float3(t[0], t[1], t[2]));
    return CRT_OK;
}
int crt_host_renderer_set_passes(crt_host_renderer* r, int passes) { if (!r || passes < 1 || passes > 4) { g_err = "passes must be 1..4"; return CRT_ERR_INVALID; } r->r->passes = passes; return CRT_OK; }
int crt_host_renderer_clear(crt_host_renderer* r) { if (!r) return CRT_ERR_INVALID; GUARD_BEGIN r->r->ClearAccumulator(); return CRT_OK; GUARD_END(CRT_ERR_DEVICE) }
int crt_host_renderer_tick(crt_host_renderer* r, float dt) { if (!r) return CRT_ERR_INVALID; GUARD_BEGIN r->r->Tick(dt); return CRT_OK; GUARD_END(CRT_ERR_DEVICE) }
int crt_host_renderer_tick_whitted(crt_host_renderer* r) { if (!r) return CRT_ERR_INVALID; GUARD_BEGIN r->r->TickWhitted(); return CRT_OK; GUARD_END(CRT_ERR_DEVICE) }
int crt_host_renderer_render(crt_host_renderer* r, int frames) { if (!r) return CRT_ERR_INVALID; GUARD_BEGIN r->r->Render(frames); return CRT_OK; GUARD_END(CRT_ERR_DEVICE) }
int crt_host_renderer_spp(crt_host_renderer* r) { return r ? r->r->spp : CRT_ERR_INVALID; }
float crt_host_renderer_energy(crt_host_renderer* r) { return r ? r->r->energy : 0.0f; }
const float* crt_host_renderer_accumulator(crt_host_renderer* r) { return r ? r->r->accumulator : nullptr; }
const uint32_t* crt_host_renderer_screen(crt_host_renderer* r) { return (r && r->r->screen) ? r->r->screen->pixels.data() : nullptr; }
crt_ctx* crt_host_renderer_ctx(crt_host_renderer* r) { return r ? r->r->ctx : nullptr; }

int crt_host_obj_load(const char* path, uint32_t* corners, float** pos, float** nrm, float** uv)
{
    if (!path || !corners || !pos || !nrm || !uv) { g_err = "null argument"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    const MeshCorners m = LoadObj(path);
    const size_t n = m.count();
    *corners = (uint32_t)n;
    *pos = (float*)malloc(n * 12); *nrm = (float*)malloc(n * 12); *uv = (float*)malloc(n * 8);
    memcpy(*pos, m.pos.data(), n * 12); memcpy(*nrm, m.nrm.data(), n * 12); memcpy(*uv, m.uv.data(), n * 8);
    return CRT_OK;
    GUARD_END(CRT_ERR_IO)
}
int crt_host_image_load(const char* path, int* w, int* h, uint32_t** pixels)
{
    if (!path || !w || !h || !pixels) { g_err = "null argument"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    const Image img = LoadImage(path);
    *w = img.width; *h = img.height;
    *pixels = (uint32_t*)malloc(img.pixels.size() * 4);
    memcpy(*pixels, img.pixels.data(), img.pixels.size() * 4);
    return CRT_OK;
    GUARD_END(CRT_ERR_IO)
}
void crt_host_free(void* p) { free(p); }

// FileScene's alternative accelerators (file_scene.h:10-12: USE_KDTree is what the reference ships): built over the scene's triangle array on the host
int crt_host_scene_build_alt(crt_host_scene* s, int kind)
{
    if (!s || !s->file) { g_err = "alternative accelerators belong to a FileScene"; return CRT_ERR_INVALID; }
    GUARD_BEGIN
    if (kind == CRT_ACCEL_KDTREE) { delete s->kd; s->kd = new KDTree(); s->kd->triangles = s->file->acc.triangles; s->kd->Build(); }
    else if (kind == CRT_ACCEL_GRID) { delete s->grid; s->grid = new Grid(); s->grid->triangles = s->file->acc.triangles; s->grid->Build(); }
    else { g_err = "unknown accelerator kind"; return CRT_ERR_INVALID; }
    return CRT_OK;
    GUARD_END(CRT_ERR_INVALID)
}
static int describe_alt(crt_host_scene* s, int kind, crt_alt_accel& a)
{
    memset(&a, 0, sizeof(a)); a.kind = kind;
    if (kind == CRT_ACCEL_KDTREE && s->kd) {
        a.triangles = s->kd->triangles.data(); a.triCount = (uint32_t)s->kd->triangles.size();
        a.kdNodes = s->kd->nodes.data(); a.kdNodeCount = (uint32_t)s->kd->nodes.size(); a.kdTriIndices = s->kd->leafTriIndices.data(); a.kdTriIndexCount = (uint32_t)s->kd->leafTriIndices.size();
        return CRT_OK;
    }
    if (kind == CRT_ACCEL_GRID && s->grid) {
        const Grid& g = *s->grid;
        a.triangles = g.triangles.data(); a.triCount = (uint32_t)g.triangles.size();
        for (int k = 0; k < 3; k++) { a.gridResolution[k] = g.resolution[k]; a.gridCellSize[k] = g.cellSize[k]; a.gridMin[k] = g.localBounds.bmin3[k]; a.gridMax[k] = g.localBounds.bmax3[k]; }
        a.gridCellStart = g.cellStart.data(); a.gridCellTris = g.cellTris.data(); a.gridCellTriCount = (uint32_t)g.cellTris.size();
        return CRT_OK;
    }
    g_err = "accelerator not built (crt_host_scene_build_alt)"; return CRT_ERR_STATE;
}
int crt_host_scene_upload_alt(crt_host_scene* s, crt_ctx* ctx, int kind)
{
    if (!s || !ctx) { g_err = "null argument"; return CRT_ERR_INVALID; }
    crt_alt_accel a; int rc = describe_alt(s, kind, a); if (rc) return rc;
    rc = crt_upload_alt_accel(ctx, &a);
    if (rc != CRT_OK) g_err = crt_last_error(ctx);
    return rc;
}
// sizes: KD-tree -> {nodes, leaf triangle indices, maxDepth, nodesUsed}; grid -> {rx, ry, rz, cell triangle references}
int crt_host_scene_alt_info(crt_host_scene* s, int kind, uint32_t out[4])
{
    if (!s || !out) { g_err = "null argument"; return CRT_ERR_INVALID; }
    crt_alt_accel a; int rc = describe_alt(s, kind, a); if (rc) return rc;
    if (kind == CRT_ACCEL_KDTREE) { out[0] = a.kdNodeCount; out[1] = a.kdTriIndexCount; out[2] = s->kd->maxDepth; out[3] = s->kd->nodesUsed; }
    else { out[0] = (uint32_t)a.gridResolution[0]; out[1] = (uint32_t)a.gridResolution[1]; out[2] = (uint32_t)a.gridResolution[2]; out[3] = a.gridCellTriCount; }
    return CRT_OK;
}
// copies: KD-tree -> nodes (48 B each) + leaf triangle indices; grid -> f[0..2] cellSize, f[3..5] bounds min, f[6..8] bounds max, cellStart (cells + 1), cell triangle references
int crt_host_scene_alt_copy(crt_host_scene* s, int kind, void* nodesOrCellStart, void* refs, float* f9)
{
    if (!s) { g_err = "null argument"; return CRT_ERR_INVALID; }
    crt_alt_accel a; int rc = describe_alt(s, kind, a); if (rc) return rc;
    if (kind == CRT_ACCEL_KDTREE) { if (nodesOrCellStart) memcpy(nodesOrCellStart, a.kdNodes, (size_t)a.kdNodeCount * 48); if (refs) memcpy(refs, a.kdTriIndices, (size_t)a.kdTriIndexCount * 4); }
    else {
        const size_t cells = (size_t)a.gridResolution[0] * a.gridResolution[1] * a.gridResolution[2];
        if (nodesOrCellStart) memcpy(nodesOrCellStart, a.gridCellStart, (cells + 1) * 4); if (refs) memcpy(refs, a.gridCellTris, (size_t)a.gridCellTriCount * 4);
        if (f9) for (int k = 0; k < 3; k++) { f9[k] = a.gridCellSize[k]; f9[3 + k] = a.gridMin[k]; f9[6 + k] = a.gridMax[k]; }
    }
    return CRT_OK;
}

// test entries: the host front's restatements of the tmplmath.h inlines / the Vertex table, same layout as the real-reference harness's probes (tests/golden/make_golden.py)
void crt_host_math_probe(const float* in, uint32_t n, float* out)
{
    using namespace crt;
    for (uint32_t i = 0; i < n; i++, in += 12, out += 120) {
        const float3 a(in[0], in[1], in[2]), b(in[3], in[4], in[5]), ang(in[6], in[7], in[8]), sc(in[9], in[10], in[11]);
        float* o = out;
        const float3 na = normalize(a), rf = a - 2.0f * b * dot(b, a) /* reflect, tmplmath.h:506 */, cr = cross(a, b);
        o[0] = na.x; o[1] = na.y; o[2] = na.z; o[3] = rf.x; o[4] = rf.y; o[5] = rf.z; o[6] = cr.x; o[7] = cr.y; o[8] = cr.z; o[9] = dot(a, b); o += 10;
        const mat4 mt = mat4::Translate(a), rx = mat4::RotateX(ang.x), ry = mat4::RotateY(ang.y), rz = mat4::RotateZ(ang.z), ms = mat4::Scale(sc);
        memcpy(o, mt.cell, 64); memcpy(o + 16, rx.cell, 64); memcpy(o + 32, ry.cell, 64); memcpy(o + 48, rz.cell, 64); memcpy(o + 64, ms.cell, 64); o += 80;
        mat4 m = ry; m.cell[3] = a.x; m.cell[7] = a.y; m.cell[11] = a.z;
        const mat4 inv = m.FastInvertedTransformNoScale();
        memcpy(o, inv.cell, 64); o += 16;
        aabb bb; bb.Grow(a); bb.Grow(b); bb.Grow(sc);
        o[0] = bb.bmin3.x; o[1] = bb.bmin3.y; o[2] = bb.bmin3.z; o[3] = bb.bmax3.x; o[4] = bb.bmax3.y; o[5] = bb.bmax3.z; o[6] = bb.Area(); o += 7;
        aabb b1, b2; b1.Grow(a); b1.Grow(b); b2.Grow(ang); b2.Grow(sc); b1.Grow(b2);
        o[0] = b1.bmin3.x; o[1] = b1.bmin3.y; o[2] = b1.bmin3.z; o[3] = b1.bmax3.x; o[4] = b1.bmax3.y; o[5] = b1.bmax3.z; o[6] = b1.Area();
    }
}
uint32_t crt_host_vertex_dedup(const float* v8, uint32_t n, uint32_t* idx, float* unique8)
{
    crt::MeshCorners m;
    for (uint32_t i = 0; i < n; i++) { m.pos.insert(m.pos.end(), v8 + 8 * i, v8 + 8 * i + 3); m.nrm.insert(m.nrm.end(), v8 + 8 * i + 3, v8 + 8 * i + 6); m.uv.insert(m.uv.end(), v8 + 8 * i + 6, v8 + 8 * i + 8); }
    std::vector<float> P, N, U; std::vector<uint32_t> ix;
    crt::DedupVertices(m, P, N, U, ix);
    memcpy(idx, ix.data(), ix.size() * 4);
    for (size_t k = 0; k < P.size() / 3; k++) { memcpy(unique8 + 8 * k, &P[3 * k], 12); memcpy(unique8 + 8 * k + 3, &N[3 * k], 12); memcpy(unique8 + 8 * k + 6, &U[2 * k], 8); }
    return (uint32_t)(P.size() / 3);
}

} // extern "C"
