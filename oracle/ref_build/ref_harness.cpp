/*
 * ref_harness.cpp — C entry points around the REAL reference code, compiled where it lies (oracle/_ref only;
 * never shipped, never loaded by the product).  Everything below the includes is glue: it fills the reference's
 * own containers, calls the reference's own functions and copies their results out.
 *   infra/bvh.cpp            -> BVH::Build, BVH::Refit, BVH::Intersect (IntersectBVH / IntersectAABB / IntersectTri)
 *   lib/tiny_obj_loader.h    -> tinyobj::LoadObj (float parsing, quad / polygon triangulation)
 *   lib/stb_image.h          -> stbi_load (PNG / JPG / TGA decode)
 *   template/camera.h        -> Camera(): default frustum, GetPrimaryRay, SetCameraState (compiled for its fixed SCRWIDTH x SCRHEIGHT = 1024 x 640)
 *   template/texture.h       -> Texture::LoadFromFile (0x00RRGGBB packing), Texture::Sample
 *   template/material.h      -> Material::GetAlbedo
 * The three headers need three names that template/precomp.h / template/opengl.h declare and that live in translation units this
 * image cannot build (template.cpp, opencl.cpp: <windows.h>, OpenGL, OpenCL): IsKeyDown (precomp.h:143), WindowHasFocus (opengl.h:23)
 * and FatalError (precomp.h:203).  They are DECLARED below exactly as there and never defined: only Camera::HandleInput and the
 * Texture(path) constructor use them, neither is instantiated here, so nothing is emitted that would need them.  Build flags:
 * -fno-access-control (the harness fills Texture's private texel vector), -Wno-non-pod-varargs (texture.h:45 passes a std::string through
 * FatalError's varargs), -DGLFW_INCLUDE_NONE (the reference's own lib/GLFW/include/GLFW/glfw3.h provides GLFW_KEY_*; no system GL headers).
 */
#include "precomp.h"
#include "bvh.cpp"                     /* /root/reference/infra/bvh.cpp, unmodified */

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

} // extern "C"
