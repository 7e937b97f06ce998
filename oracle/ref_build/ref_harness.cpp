/*
 * ref_harness.cpp — C entry points around the REAL reference code, compiled where it lies (oracle/_ref only;
 * never shipped, never loaded by the product).  Everything below the includes is glue: it fills the reference's
 * own containers, calls the reference's own functions and copies their results out.
 *   infra/bvh.cpp            -> BVH::Build, BVH::Refit, BVH::Intersect (IntersectBVH / IntersectAABB / IntersectTri)
 *   lib/tiny_obj_loader.h    -> tinyobj::LoadObj (float parsing, quad / polygon triangulation)
 *   lib/stb_image.h          -> stbi_load (PNG / JPG / TGA decode)
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

#include <stdint.h>

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

} // extern "C"
