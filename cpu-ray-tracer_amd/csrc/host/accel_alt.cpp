// accel_alt.cpp — host builds of KDTree and Grid (see accel_alt.h).  The node order / cell contents must equal the reference's because
// Ray::traversed / Ray::tested and, through the visiting order, ties between equal hits depend on them.
#include "accel_alt.h"

#include <cmath>

namespace crt {

static aabb tri_bounds(const Tri& t)             // Tri::GetBounds, infra/helper.h:18-25
{
    aabb b;
    b.Grow(float3(t.vertex0[0], t.vertex0[1], t.vertex0[2])); b.Grow(float3(t.vertex1[0], t.vertex1[1], t.vertex1[2])); b.Grow(float3(t.vertex2[0], t.vertex2[1], t.vertex2[2]));
    return b;
}

void KDTree::Build()
{
    triangleBounds.resize(triangles.size());
    aabb all;                                                        // UpdateBounds, kdtree.cpp:28-42
    for (size_t i = 0; i < triangles.size(); i++) { const aabb tb = tri_bounds(triangles[i]); all.Grow(tb); triangleBounds[i] = tb; }
    localBounds = all;
    nodes.clear(); leafTriIndices.clear(); nodesUsed = 1; maxDepth = 0;
    // Subdivide (kdtree.cpp:45-107) is depth-first, left before right; here as a work list that emits the nodes in pre-order.  A pending item carries
    // the triangle list the reference keeps in node->triIndices until the node is split.
    struct Item { uint32_t node; int depth; std::vector<uint32_t> tris; };
    std::vector<Item> work;
    auto emit = [&](const float3& lo, const float3& hi) {
        crt_kd_node n{}; n.aabbMin[0] = lo.x; n.aabbMin[1] = lo.y; n.aabbMin[2] = lo.z; n.aabbMax[0] = hi.x; n.aabbMax[1] = hi.y; n.aabbMax[2] = hi.z;
        n.left = n.right = -1; n.firstTri = (uint32_t)leafTriIndices.size();
        nodes.push_back(n);
        return (uint32_t)nodes.size() - 1u;
    };
    {
        Item root; root.depth = 0; root.tris.resize(triangles.size());
        for (size_t i = 0; i < triangles.size(); i++) root.tris[i] = (uint32_t)i;
        root.node = emit(all.bmin3, all.bmax3);
        work.push_back(std::move(root));
    }
    // pre-order with an explicit stack: the right child is created (numbered) only after the whole left subtree, so an interior node parks
    // its right list until then
    struct Pending { uint32_t parent; int depth; float3 lo, hi; std::vector<uint32_t> tris; };
    std::vector<Pending> rights;
    while (!work.empty() || !rights.empty()) {
        if (work.empty()) {
            Pending p = std::move(rights.back()); rights.pop_back();
            Item it; it.depth = p.depth; it.tris = std::move(p.tris); it.node = emit(p.lo, p.hi);
            nodes[p.parent].right = (int32_t)it.node;
            work.push_back(std::move(it));
            continue;
        }
        Item it = std::move(work.back()); work.pop_back();
        const uint32_t triCount = (uint32_t)it.tris.size();
        if (it.depth >= m_maxBuildDepth || triCount <= 2) {                     // leaf: keeps its list
            nodes[it.node].firstTri = (uint32_t)leafTriIndices.size(); nodes[it.node].triCount = triCount;
            leafTriIndices.insert(leafTriIndices.end(), it.tris.begin(), it.tris.end());
            continue;
        }
        if ((uint32_t)it.depth > maxDepth) maxDepth = (uint32_t)it.depth;
        const float3 lo(nodes[it.node].aabbMin[0], nodes[it.node].aabbMin[1], nodes[it.node].aabbMin[2]), hi(nodes[it.node].aabbMax[0], nodes[it.node].aabbMax[1], nodes[it.node].aabbMax[2]);
        const float3 extent = hi - lo;
        int axis = 0;
        if (extent.y > extent.x) axis = 1;
        if (extent.z > extent[axis]) axis = 2;
        const float distance = extent[axis] * 0.5f;
        const float splitPos = lo[axis] + distance;
        std::vector<uint32_t> L, R;
        for (uint32_t i = 0; i < triCount; i++) {
            const uint32_t idx = it.tris[i];
            if (triangleBounds[idx].bmax3[axis] < splitPos) L.push_back(idx);
            else if (triangleBounds[idx].bmin3[axis] > splitPos - 0.001) R.push_back(idx);           // float vs. double expression, as written in the reference
            else { L.push_back(idx); R.push_back(idx); }
        }
        nodesUsed += 2;
        nodes[it.node].splitAxis = axis; nodes[it.node].splitDistance = distance;
        float3 lhi = hi, rlo = lo; lhi[axis] = splitPos; rlo[axis] = splitPos;
        Pending p; p.parent = it.node; p.depth = it.depth + 1; p.lo = rlo; p.hi = hi; p.tris = std::move(R);
        rights.push_back(std::move(p));
        Item l; l.depth = it.depth + 1; l.tris = std::move(L); l.node = emit(lo, lhi);
        nodes[it.node].left = (int32_t)l.node;
        work.push_back(std::move(l));
    }
}

void Grid::Build()
{
    for (const Tri& t : triangles) localBounds.Grow(tri_bounds(t));
    const float3 gridSize = localBounds.bmax3 - localBounds.bmin3;
    const float cubeRoot = powf(5 * GetTriangleCount() / (gridSize.x * gridSize.y * gridSize.z), 1 / 3.f);
    for (int i = 0; i < 3; i++) {
        int r = static_cast<int>(floorf(gridSize[i] * cubeRoot));
        r = r < 128 ? r : 128;                                                  // max(1, min(r, 128))
        resolution[i] = r > 1 ? r : 1;
    }
    cellSize = float3(gridSize.x / resolution[0], gridSize.y / resolution[1], gridSize.z / resolution[2]);
    const size_t cells = (size_t)resolution[0] * resolution[1] * resolution[2];
    // two passes instead of a std::vector per cell: count, prefix, fill — triangles are visited in index order in both, so every cell lists them
    // in the reference's push_back order
    auto range = [&](const Tri& t, int mn[3], int mx[3]) {
        const aabb b = tri_bounds(t);
        for (int k = 0; k < 3; k++) {
            const int lo = static_cast<int>((b.bmin3[k] - localBounds.bmin3[k]) / cellSize[k]), hi = static_cast<int>((b.bmax3[k] - localBounds.bmin3[k]) / cellSize[k]);
            const int top = resolution[k] - 1;
            mn[k] = lo < 0 ? 0 : (lo > top ? top : lo); mx[k] = hi < 0 ? 0 : (hi > top ? top : hi);
        }
    };
    cellStart.assign(cells + 1, 0u);
    for (const Tri& t : triangles) {
        int mn[3], mx[3]; range(t, mn, mx);
        for (int iz = mn[2]; iz <= mx[2]; ++iz) for (int iy = mn[1]; iy <= mx[1]; ++iy) for (int ix = mn[0]; ix <= mx[0]; ++ix)
            cellStart[(size_t)ix + (size_t)iy * resolution[0] + (size_t)iz * resolution[0] * resolution[1] + 1]++;
    }
    for (size_t c = 0; c < cells; c++) cellStart[c + 1] += cellStart[c];
    cellTris.assign(cellStart[cells], 0);
    std::vector<uint32_t> fill(cellStart.begin(), cellStart.end() - 1);
    for (size_t ti = 0; ti < triangles.size(); ti++) {
        int mn[3], mx[3]; range(triangles[ti], mn, mx);
        for (int iz = mn[2]; iz <= mx[2]; ++iz) for (int iy = mn[1]; iy <= mx[1]; ++iy) for (int ix = mn[0]; ix <= mx[0]; ++ix)
            cellTris[fill[(size_t)ix + (size_t)iy * resolution[0] + (size_t)iz * resolution[0] * resolution[1]]++] = (int32_t)ti;
    }
}

} // namespace crt
