/*
 * ref_harness.cpp — C entry points around the REAL reference code, compiled where it lies (oracle/_ref only;
 * never shipped, never loaded by the product).  Everything below the includes is glue: it fills the reference's
 * own containers, calls the reference's own functions and copies their results out.
 *   infra/bvh.cpp            -> BVH::Build, BVH::Refit, BVH::Intersect (IntersectBVH / IntersectAABB / IntersectTri)
 *   infra/kdtree.cpp         -> KDTree::Build (spatial median, depth <= 20), KDTree::Intersect (recursive front-to-back traversal)
 *   infra/grid.cpp           -> Grid::Build (uniform grid, <= 128^3 cells), Grid::Intersect (3D-DDA)
 *   lib/tiny_obj_loader.h    -> tinyobj::LoadObj (float parsing, quad / polygon triangulation)
 *   lib/stb_image.h          -> stbi_load (PNG / JPG / TGA decode)
 *   template/camera.h        -> Camera(): default frustum, GetPrimaryRay, SetCameraState (compiled for its fixed SCRWIDTH x SCRHEIGHT = 1024 x 640)
 *   template/texture.h       -> Texture::LoadFromFile (0x00RRGGBB packing), Texture::Sample
 *   template/material.h      -> Material::GetAlbedo
 *   template/tmplmath.h      -> the inline functions on the path: normalize, reflect, cross, dot, mat4::Translate / RotateX / RotateY / RotateZ / Scale,
 *                               mat4::FastInvertedTransformNoScale, aabb::Grow / Area   (ref_math_probe)
 *   infra/helper.h           -> Vertex::operator== and std::hash<Vertex> inside a real std::unordered_map<Vertex, uint32_t>   (ref_vertex_dedup)
 * The three headers need three names that template/precomp.h / template/opengl.h declare and that live in translation units this
 * image cannot build (template.cpp, opencl.cpp: <windows.h>, OpenGL, OpenCL): IsKeyDown (precomp.h:143), WindowHasFocus (opengl.h:23)
 * and FatalError (precomp.h:203).  They are DECLARED below exactly as there and never defined: only Camera::HandleInput and the
 * Texture(path) constructor use them, neither is instantiated here, so nothing is emitted that would need them.  Build flags:
 * -fno-access-control (the harness fills Texture's private texel vector), -Wno-non-pod-varargs (texture.h:45 passes a std::string through
 * FatalError's varargs), -DGLFW_INCLUDE_NONE (the reference's own lib/GLFW/include/GLFW/glfw3.h provides GLFW_KEY_*; no system GL headers).
 */
#include "precomp.h"
#include "bvh.cpp"                     /* /root/reference/infra/bvh.cpp, unmodified */
#include "kdtree.cpp"                  /* /root/reference/infra/kdtree.cpp, unmodified (the FileScene default accelerator, file_scene.h:10-12) */
#include "grid.cpp"                    /* /root/reference/infra/grid.cpp, unmodified */

#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_PSD
#define STBI_NO_PIC
#define STBI_NO_PNM
#include "stb_image.h"                 /* /root/reference/lib/stb_image.h */
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"           /* /root/reference/lib/tiny_obj_loader.h */

#include "GLFW/glfw3.h"                /* /root/reference/lib/GLFW/include/GLFW/glfw3.h (GLFW_KEY_* used by camera.h) */
bool IsKeyDown( const uint key );      /* template/precomp.h:143 (declaration only) */
bool WindowHasFocus();                 /* template/opengl.h:23   (declaration only) */
void FatalError( const char* fmt, ... ); /* template/precomp.h:203 (declaration only) */
#include "camera.h"                    /* /root/reference/template/camera.h */
#include "texture.h"                   /* /root/reference/template/texture.h */
#include "material.h"                  /* /root/reference/template/material.h */

#include <stdint.h>

static_assert(SCRWIDTH == 1024 && SCRHEIGHT == 640, "camera.h resolution");
static_assert(sizeof(Tri) == 112, "Tri layout");
static_assert(sizeof(BVHNode) == 32, "BVHNode layout");

extern "C" {

struct ref_hit { float t, u, v; int32_t objIdx, triIdx, traversed, tested; };

void* ref_bvh_build(const void* tris112, uint32_t n)
{
    BVH* b = new BVH();
    b->triangles.resize(n);
    memcpy((void*)b->triangles.data(), tris112, (size_t)n * sizeof(Tri));
    b->Build();
    return b;
}
void ref_bvh_free(void* h) { delete (BVH*)h; }
void ref_bvh_info(void* h, uint32_t* nodesUsed, uint32_t* maxDepth) { BVH* b = (BVH*)h; *nodesUsed = b->nodesUsed; *maxDepth = b->maxDepth; }
void ref_bvh_copy(void* h, void* nodes32, uint32_t* triIdx)
{
    BVH* b = (BVH*)h;
    memcpy(nodes32, b->bvhNodes.data(), (size_t)b->nodesUsed * sizeof(BVHNode));
    memcpy(triIdx, b->triangleIndices.data(), b->triangleIndices.size() * 4);
}
/* BVH::Refit of the reference (infra/bvh.cpp:26-43) after moving the vertices: positions = 9 floats per triangle */
void ref_bvh_move_and_refit(void* h, const float* positions, uint32_t n)
{
    BVH* b = (BVH*)h;
    for (uint32_t i = 0; i < n && i < b->triangles.size(); i++) {
        Tri& t = b->triangles[i];
        t.vertex0 = float3(positions[9 * i], positions[9 * i + 1], positions[9 * i + 2]);
        t.vertex1 = float3(positions[9 * i + 3], positions[9 * i + 4], positions[9 * i + 5]);
        t.vertex2 = float3(positions[9 * i + 6], positions[9 * i + 7], positions[9 * i + 8]);
    }
    b->Refit();
}
void ref_bvh_intersect(void* h, const float* O, const float* D, uint32_t n, ref_hit* out)
{
    BVH* b = (BVH*)h;
    for (uint32_t i = 0; i < n; i++) {
        Ray r(float3(O[3 * i], O[3 * i + 1], O[3 * i + 2]), float3(D[3 * i], D[3 * i + 1], D[3 * i + 2]));
        b->Intersect(r);
        out[i].t = r.t; out[i].u = r.barycentric.x; out[i].v = r.barycentric.y;
        out[i].objIdx = r.objIdx; out[i].triIdx = r.triIdx; out[i].traversed = r.traversed; out[i].tested = r.tested;
    }
}

/* ---- KDTree (infra/kdtree.cpp): flattened in PRE-ORDER (node, left subtree, right subtree); leaves list their triangle indices in a shared array ---- */
void* ref_kd_build(const void* tris112, uint32_t n)
{
    KDTree* k = new KDTree();
    k->triangles.resize(n);
    memcpy((void*)k->triangles.data(), tris112, (size_t)n * sizeof(Tri));
    k->Build();
    return k;
}
static void kd_count(const KDTreeNode* nd, uint32_t& nodes, uint32_t& refs) { nodes++; if (nd->isLeaf) { refs += (uint32_t)nd->triIndices.size(); return; } kd_count(nd->left, nodes, refs); kd_count(nd->right, nodes, refs); }
void ref_kd_info(void* h, uint32_t* nodes, uint32_t* refs, uint32_t* maxDepth, uint32_t* nodesUsed)
{
    KDTree* k = (KDTree*)h; *nodes = 0; *refs = 0; kd_count(k->rootNode, *nodes, *refs); *maxDepth = k->maxDepth; *nodesUsed = k->nodesUsed;
}
struct ref_kd_node { float aabbMin[3]; int32_t left; float aabbMax[3]; int32_t right; float splitDistance; int32_t splitAxis; uint32_t firstTri, triCount; };
static uint32_t kd_flatten(const KDTreeNode* nd, ref_kd_node* out, uint32_t& next, uint32_t* refs, uint32_t& nrefs)
{
    const uint32_t me = next++;
    ref_kd_node& o = out[me];
    o.aabbMin[0] = nd->aabbMin.x; o.aabbMin[1] = nd->aabbMin.y; o.aabbMin[2] = nd->aabbMin.z; o.aabbMax[0] = nd->aabbMax.x; o.aabbMax[1] = nd->aabbMax.y; o.aabbMax[2] = nd->aabbMax.z;
    o.splitDistance = nd->splitDistance; o.splitAxis = nd->splitAxis; o.firstTri = nrefs; o.triCount = 0; o.left = o.right = -1;
    if (nd->isLeaf) { o.triCount = (uint32_t)nd->triIndices.size(); for (uint t : nd->triIndices) refs[nrefs++] = t; return me; }
    const uint32_t l = kd_flatten(nd->left, out, next, refs, nrefs), r = kd_flatten(nd->right, out, next, refs, nrefs);
    out[me].left = (int32_t)l; out[me].right = (int32_t)r;
    return me;
}
void ref_kd_dump(void* h, void* nodes48, uint32_t* refs) { uint32_t next = 0, nrefs = 0; kd_flatten(((KDTree*)h)->rootNode, (ref_kd_node*)nodes48, next, refs, nrefs); }
void ref_kd_intersect(void* h, const float* O, const float* D, uint32_t n, ref_hit* out)
{
    KDTree* k = (KDTree*)h;
    for (uint32_t i = 0; i < n; i++) {
        Ray r(float3(O[3 * i], O[3 * i + 1], O[3 * i + 2]), float3(D[3 * i], D[3 * i + 1], D[3 * i + 2]));
        k->Intersect(r);
        out[i].t = r.t; out[i].u = r.barycentric.x; out[i].v = r.barycentric.y;
        out[i].objIdx = r.objIdx; out[i].triIdx = r.triIdx; out[i].traversed = r.traversed; out[i].tested = r.tested;
    }
}
void ref_kd_free(void* h) { delete (KDTree*)h; }      /* (the reference never frees the nodes either) */

/* ---- Grid (infra/grid.cpp): cells in x-major order (ix + iy * rx + iz * rx * ry), each with its triangle list ---- */
void* ref_grid_build(const void* tris112, uint32_t n)
{
    Grid* g = new Grid();
    g->triangles.resize(n);
    memcpy((void*)g->triangles.data(), tris112, (size_t)n * sizeof(Tri));
    g->Build();
    return g;
}
void ref_grid_info(void* h, int32_t res[3], float cell[3], float lo[3], float hi[3], uint32_t* refs)
{
    Grid* g = (Grid*)h;
    res[0] = g->resolution.x; res[1] = g->resolution.y; res[2] = g->resolution.z; cell[0] = g->cellSize.x; cell[1] = g->cellSize.y; cell[2] = g->cellSize.z;
    for (int k = 0; k < 3; k++) { lo[k] = g->localBounds.bmin[k]; hi[k] = g->localBounds.bmax[k]; }
    uint32_t r = 0; for (const GridCell& c : g->gridCells) r += (uint32_t)c.triIndices.size(); *refs = r;
}
void ref_grid_dump(void* h, uint32_t* cellStart /* cells + 1 */, int32_t* refs)
{
    Grid* g = (Grid*)h; uint32_t r = 0, i = 0;
    for (const GridCell& c : g->gridCells) { cellStart[i++] = r; for (int t : c.triIndices) refs[r++] = t; }
    cellStart[i] = r;
}
void ref_grid_intersect(void* h, const float* O, const float* D, uint32_t n, ref_hit* out)
{
    Grid* g = (Grid*)h;
    for (uint32_t i = 0; i < n; i++) {
        Ray r(float3(O[3 * i], O[3 * i + 1], O[3 * i + 2]), float3(D[3 * i], D[3 * i + 1], D[3 * i + 2]));
        g->Intersect(r);
        out[i].t = r.t; out[i].u = r.barycentric.x; out[i].v = r.barycentric.y;
        out[i].objIdx = r.objIdx; out[i].triIdx = r.triIdx; out[i].traversed = r.traversed; out[i].tested = r.tested;
    }
}
void ref_grid_free(void* h) { delete (Grid*)h; }

/* tinyobj: corners resolved exactly as infra/model.cpp:16-54 does (missing normal/uv index -> zeros) */
struct ref_obj { std::vector<float> pos, nrm, uv; };
void* ref_obj_load(const char* path, uint32_t* nCorners)
{
    tinyobj::attrib_t attrib; std::vector<tinyobj::shape_t> shapes; std::vector<tinyobj::material_t> materials;
    std::string warn, err;
    if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, path)) return nullptr;
    ref_obj* o = new ref_obj();
    for (const auto& shape : shapes) for (const auto& index : shape.mesh.indices) {
        float p[3] = {0, 0, 0}, nn[3] = {0, 0, 0}, t[2] = {0, 0};
        if (index.vertex_index >= 0) for (int k = 0; k < 3; k++) p[k] = attrib.vertices[3 * index.vertex_index + k];
        if (index.normal_index >= 0) for (int k = 0; k < 3; k++) nn[k] = attrib.normals[3 * index.normal_index + k];
        if (index.texcoord_index >= 0) for (int k = 0; k < 2; k++) t[k] = attrib.texcoords[2 * index.texcoord_index + k];
        o->pos.insert(o->pos.end(), p, p + 3); o->nrm.insert(o->nrm.end(), nn, nn + 3); o->uv.insert(o->uv.end(), t, t + 2);
    }
    *nCorners = (uint32_t)(o->pos.size() / 3);
    return o;
}
void ref_obj_copy(void* h, float* pos, float* nrm, float* uv)
{
    ref_obj* o = (ref_obj*)h;
    memcpy(pos, o->pos.data(), o->pos.size() * 4); memcpy(nrm, o->nrm.data(), o->nrm.size() * 4); memcpy(uv, o->uv.data(), o->uv.size() * 4);
}
void ref_obj_free(void* h) { delete (ref_obj*)h; }

/* stb_image: raw decoded bytes + channel count, as Texture::LoadFromFile receives them (template/texture.h:18) */
unsigned char* ref_image_load(const char* path, int* w, int* h, int* n) { return stbi_load(path, w, h, n, 0); }
void ref_image_free(unsigned char* p) { stbi_image_free(p); }

/* Camera (template/camera.h:14-30, 61-73): default frustum or SetCameraState(pos, target); rays for n pixel coordinates */
void ref_camera_rays(const float* pos_target /* 6 floats or NULL = default Camera() */, const float* xy, uint32_t n, float* corners /* camPos, TL, TR, BL */, float* O, float* D)
{
    Camera cam;
    if (pos_target) cam.SetCameraState(float3(pos_target[0], pos_target[1], pos_target[2]), float3(pos_target[3], pos_target[4], pos_target[5]));
    const float3 c[4] = {cam.camPos, cam.topLeft, cam.topRight, cam.bottomLeft};
    for (int k = 0; k < 4; k++) { corners[3 * k] = c[k].x; corners[3 * k + 1] = c[k].y; corners[3 * k + 2] = c[k].z; }
    for (uint32_t i = 0; i < n; i++) {
        Ray r = cam.GetPrimaryRay(xy[2 * i], xy[2 * i + 1]);
        O[3 * i] = r.O.x; O[3 * i + 1] = r.O.y; O[3 * i + 2] = r.O.z; D[3 * i] = r.D.x; D[3 * i + 1] = r.D.y; D[3 * i + 2] = r.D.z;
    }
}
/* Texture::LoadFromFile's packing (template/texture.h:15-39): returns width * height texels, 0x00RRGGBB */
uint32_t* ref_texture_load(const char* path, int* w, int* h)
{
    Texture t; t.LoadFromFile(path);
    if (t.pixels.empty()) return nullptr;
    *w = t.width; *h = t.height;
    uint32_t* out = (uint32_t*)malloc(t.pixels.size() * 4);
    memcpy(out, t.pixels.data(), t.pixels.size() * 4);
    return out;
}
void ref_free(void* p) { free(p); }
/* Texture::Sample (template/texture.h:61-96) and Material::GetAlbedo (template/material.h:28-35) on caller-provided texels */
void ref_texture_sample(const uint32_t* px, int w, int h, const float* uv, uint32_t n, float* rgb, float* albedo)
{
    Material m;
    m.textureDiffuse = std::make_unique<Texture>();
    Texture& t = *m.textureDiffuse;
    t.pixels.assign(px, px + (size_t)w * h); t.width = w; t.height = h;
    for (uint32_t i = 0; i < n; i++) {
        const float3 c = t.Sample(uv[2 * i], uv[2 * i + 1]);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
        const float3 a = m.GetAlbedo(float2(uv[2 * i], uv[2 * i + 1]));
        albedo[3 * i] = a.x; albedo[3 * i + 1] = a.y; albedo[3 * i + 2] = a.z;
    }
}

/* The inline functions of template/tmplmath.h that the path uses (file:line in the reference): normalize :480, reflect :506, cross :512, dot :458,
 * mat4::Translate :735, RotateX/Y/Z :673-675, Scale :677, FastInvertedTransformNoScale :745-768 (the non-MSVC branch is what compiles here),
 * aabb::Grow(float3) / Grow(aabb) / Area :580-598.  mat4 PRODUCTS (operator*, tmplmath.cpp:109-122) are not inline and cannot be built here.
 * in: n x 12 floats (a[3], b[3], angles[3], s[3]); out: n x 120 floats, layout stated in tests/golden/make_golden.py. */
void ref_math_probe(const float* in, uint32_t n, float* out)
{
    for (uint32_t i = 0; i < n; i++, in += 12, out += 120) {
        const float3 a(in[0], in[1], in[2]), b(in[3], in[4], in[5]), ang(in[6], in[7], in[8]), sc(in[9], in[10], in[11]);
        float* o = out;
        const float3 na = normalize(a), rf = reflect(a, b), cr = cross(a, b);
        o[0] = na.x; o[1] = na.y; o[2] = na.z; o[3] = rf.x; o[4] = rf.y; o[5] = rf.z; o[6] = cr.x; o[7] = cr.y; o[8] = cr.z; o[9] = dot(a, b); o += 10;
        const mat4 mt = mat4::Translate(a), rx = mat4::RotateX(ang.x), ry = mat4::RotateY(ang.y), rz = mat4::RotateZ(ang.z), ms = mat4::Scale(sc);
        memcpy(o, mt.cell, 64); memcpy(o + 16, rx.cell, 64); memcpy(o + 32, ry.cell, 64); memcpy(o + 48, rz.cell, 64); memcpy(o + 64, ms.cell, 64); o += 80;
        mat4 m = ry;                                       /* a rigid transform assembled without operator*: rotation block of RotateY, translation a */
        m.cell[3] = a.x; m.cell[7] = a.y; m.cell[11] = a.z;
        const mat4 inv = m.FastInvertedTransformNoScale();
        memcpy(o, inv.cell, 64); o += 16;
        aabb bb; bb.Grow(a); bb.Grow(b); bb.Grow(sc);
        o[0] = bb.bmin[0]; o[1] = bb.bmin[1]; o[2] = bb.bmin[2]; o[3] = bb.bmax[0]; o[4] = bb.bmax[1]; o[5] = bb.bmax[2]; o[6] = bb.Area(); o += 7;
        aabb b1, b2; b1.Grow(a); b1.Grow(b); b2.Grow(ang); b2.Grow(sc); b1.Grow(b2);
        o[0] = b1.bmin[0]; o[1] = b1.bmin[1]; o[2] = b1.bmin[2]; o[3] = b1.bmax[0]; o[4] = b1.bmax[1]; o[5] = b1.bmax[2]; o[6] = b1.Area();
    }
}
/* The three statements of infra/model.cpp:44-50 / infra/blas_bvh.cpp:46-52 around the REAL Vertex (operator== = float equality, infra/helper.h:34-37)
 * and the REAL std::hash<Vertex> (helper.h:40-86) in a real std::unordered_map: idx[i] = index of corner i, unique = the vertices array (8 floats each).
 * Returns the number of unique vertices; hash[i] = std::hash<Vertex> of corner i. */
uint32_t ref_vertex_dedup(const float* v8, uint32_t n, uint32_t* idx, float* unique8, uint64_t* hash)
{
    std::unordered_map<Vertex, uint32_t> uniqueVertices{};
    std::vector<Vertex> vertices;
    for (uint32_t i = 0; i < n; i++) {
        Vertex vertex{};
        vertex.position = {v8[8 * i], v8[8 * i + 1], v8[8 * i + 2]};
        vertex.normal = {v8[8 * i + 3], v8[8 * i + 4], v8[8 * i + 5]};
        vertex.uv = {v8[8 * i + 6], v8[8 * i + 7]};
        if (uniqueVertices.count(vertex) == 0) { uniqueVertices[vertex] = static_cast<uint32_t>(vertices.size()); vertices.push_back(vertex); }
        idx[i] = uniqueVertices[vertex];
        hash[i] = (uint64_t)std::hash<Vertex>()(vertex);
    }
    for (size_t k = 0; k < vertices.size(); k++) {
        const Vertex& v = vertices[k]; float* u = unique8 + 8 * k;
        u[0] = v.position.x; u[1] = v.position.y; u[2] = v.position.z; u[3] = v.normal.x; u[4] = v.normal.y; u[5] = v.normal.z; u[6] = v.uv.x; u[7] = v.uv.y;
    }
    return (uint32_t)vertices.size();
}

} // extern "C"
