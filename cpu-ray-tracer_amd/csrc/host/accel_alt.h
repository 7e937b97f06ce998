// accel_alt.h — FileScene's alternative acceleration structures on the host front: KDTree (infra/kdtree.{h,cpp}; what the reference ships
// enabled, infra/scene/file_scene.h:10-12) and Grid (infra/grid.{h,cpp}).  Same members as there (triangles, rootNode / gridCells flattened, Build()); the
// build runs on the CPU exactly as in the reference and the result is handed to crt_upload_alt_accel; Intersect() lives on the GPU (crt_find_nearest_alt).
#pragma once
#include "accel.h"

namespace crt {

class KDTree {                                   // infra/kdtree.h:14-38
public:
    void Build();                                // kdtree.cpp:4-107: spatial median of the longest axis, depth <= 20, leaves of <= 2 triangles
    int GetTriangleCount() const { return (int)triangles.size(); }
    std::vector<Tri> triangles;
    std::vector<aabb> triangleBounds;
    // the pointer-linked KDTreeNode tree (blas_kdtree.h:15-24) in pre-order; leaves name a range of leafTriIndices
    std::vector<crt_kd_node> nodes;
    std::vector<uint32_t> leafTriIndices;
    uint32_t nodesUsed = 1, maxDepth = 0;
    aabb localBounds;
private:
    int m_maxBuildDepth = 20;
};

class Grid {                                     // infra/grid.h:10-34
public:
    void Build();                                // grid.cpp:4-50: resolution from 5 triangles per cell on average, clamped to 1..128 per axis
    int GetTriangleCount() const { return (int)triangles.size(); }
    std::vector<Tri> triangles;
    int resolution[3] = {0, 0, 0};
    float3 cellSize{0, 0, 0};
    aabb localBounds;
    std::vector<uint32_t> cellStart;             // gridCells[c].triIndices = cellTris[cellStart[c] .. cellStart[c + 1])
    std::vector<int32_t> cellTris;
};

} // namespace crt
