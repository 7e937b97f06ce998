// accel.h — CPU-side acceleration structures of the host front: BVH (single level), BLASBVH (per model, instanced)
// and TLASBVH (top level).  Same public surface as the reference classes (infra/bvh.h:28-43, infra/blas_bvh.h:36-57,
// infra/tlas_bvh.h:22-31): Build(), public bvhNodes / triangles / triangleIndices / nodesUsed / T / invT / blas.
// The build runs on the CPU and must reproduce the reference's node numbering and triangle order exactly — the GPU
// layout is derived from these arrays (crt_upload_scene).  Traversal lives on the GPU; Intersect() is not provided here.
#pragma once
#include "../../../include/crt_abi.h"
#include "hmath.h"

#include <string>
#include <vector>

namespace crt {

typedef crt_tri Tri;           // 112-byte reference layout
typedef crt_bvh_node BVHNode;  // 32-byte reference layout
typedef crt_tlas_node TLASBVHNode;

struct MeshCorners {           // tinyobj-resolved corners, three per triangle (positions, normals, uvs; zeros when absent)
    std::vector<float> pos, nrm, uv;
    size_t count() const { return pos.size() / 3; }
};

// unique-vertex table of Model / BLASBVH (model.cpp:16-54): index of every corner + the unique vertices, first occurrence wins (test entry)
void DedupVertices(const MeshCorners& m, std::vector<float>& P, std::vector<float>& N, std::vector<float>& U, std::vector<uint32_t>& indices);
// binned-SAH builder shared by BVH and BLASBVH (the reference duplicates the code: bvh.cpp:4-178, blas_bvh.cpp:82-256)
void BuildSAH(std::vector<Tri>& triangles, std::vector<BVHNode>& nodes, std::vector<uint32_t>& triangleIndices, uint32_t& nodesUsed, uint32_t& maxDepth);
// BVH::Refit / BLASBVH::Refit (bvh.cpp:26-43, blas_bvh.cpp:104-121): bounds of every node recomputed bottom-up for moved vertices,
// topology unchanged.  Node 1 is skipped exactly as the reference does (its loop tests `i != 1`, a left-over of the tutorial layout
// that keeps node 1 unused; here node 1 is the root's left child, so its box stays as built — bug-compatible).
void RefitSAH(const std::vector<Tri>& triangles, std::vector<BVHNode>& nodes, const std::vector<uint32_t>& triangleIndices, uint32_t nodesUsed);

class BVH {
public:
    void Build() { BuildSAH(triangles, bvhNodes, triangleIndices, nodesUsed, maxDepth); }
    void Refit() { RefitSAH(triangles, bvhNodes, triangleIndices, nodesUsed); }
    int GetTriangleCount() const { return (int)triangles.size(); }
    int objIdx = -1;
    std::vector<BVHNode> bvhNodes;
    std::vector<Tri> triangles;
    std::vector<uint32_t> triangleIndices;
    uint32_t rootNodeIdx = 0, nodesUsed = 1;
    uint32_t maxDepth = 0;
};

class BLASBVH {
public:
    BLASBVH() = default;
    // infra/blas_bvh.cpp:4-80: de-duplicated vertices -> triangles with the scale baked in, Build(), SetTransform(T)
    BLASBVH(int idx, const MeshCorners& mesh, const mat4& transform, const mat4& scaleMat);
    void Build() { BuildSAH(triangles, bvhNodes, triangleIndices, nodesUsed, maxDepth); }
    void Refit() { RefitSAH(triangles, bvhNodes, triangleIndices, nodesUsed); }
    void SetTransform(const mat4& transform);
    int GetTriangleCount() const { return (int)triangles.size(); }
    int objIdx = -1, matIdx = -1;
    std::vector<BVHNode> bvhNodes;
    std::vector<Tri> triangles;
    std::vector<uint32_t> triangleIndices;
    uint32_t rootNodeIdx = 0, nodesUsed = 1;
    aabb worldBounds;
    mat4 T, invT;
    uint32_t maxDepth = 0;
};

class TLASBVH {
public:
    TLASBVH() = default;
    explicit TLASBVH(const std::vector<BLASBVH*>& bvhList);
    void Build();
    std::vector<TLASBVHNode> tlasNode;     // 2 * blasCount entries (reference keeps this private; exposed for the upload)
    uint32_t nodesUsed = 0, blasCount = 0;
    std::vector<BLASBVH*> blas;
};

// Model (infra/model.{h,cpp}): de-duplicated vertices of one OBJ + its world transform, for the single-BVH FileScene
class Model {
public:
    Model() = default;
    Model(int idx, const MeshCorners& mesh, const mat4& transform);
    void AppendTriangles(std::vector<Tri>& triangles) const;
    int objIdx = -1, matIdx = -1;
    std::vector<float> positions, normals, uvs;     // unique vertices (xyz, xyz, uv)
    std::vector<uint32_t> indices;
    mat4 T, invT;
};

} // namespace crt
