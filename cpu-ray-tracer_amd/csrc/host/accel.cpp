// accel.cpp — CPU construction of the BVH / BLAS / TLAS that the GPU traverses.
#include "accel.h"

#include <stdexcept>
#include <unordered_map>

namespace crt {

namespace {

constexpr int kBins = 8;    // BVH_BINS (infra/bvh.h:7, infra/blas_bvh.h:5)

inline float3 V(const float* p) { return {p[0], p[1], p[2]}; }
inline void S(float* p, const float3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

struct Builder {
    std::vector<Tri>& tris; std::vector<BVHNode>& nodes; std::vector<uint32_t>& idx;
    uint32_t used = 1, deepest = 0;

    // tight bounds over a node's triangles (bvh.cpp:45-61); sentinels are 1e30, not the aabb class's 1e34
    void fit(uint32_t n)
    {
        BVHNode& node = nodes[n];
        float3 lo(1e30f), hi(-1e30f);
        for (uint32_t i = 0; i < node.triCount; i++) {
            const Tri& t = tris[idx[node.leftFirst + i]];
            lo = fminf3(lo, V(t.vertex0)); lo = fminf3(lo, V(t.vertex1)); lo = fminf3(lo, V(t.vertex2));
            hi = fmaxf3(hi, V(t.vertex0)); hi = fmaxf3(hi, V(t.vertex1)); hi = fmaxf3(hi, V(t.vertex2));
        }
        S(node.aabbMin, lo); S(node.aabbMax, hi);
    }

    // best of the 3 x 7 candidate planes (bvh.cpp:124-178); std::min / std::max operand order kept
    float choosePlane(const BVHNode& node, int& axis, float& pos) const
    {
        float best = 1e30f;
        for (int a = 0; a < 3; a++) {
            float cmin = 1e30f, cmax = -1e30f;
            for (uint32_t i = 0; i < node.triCount; i++) {
                const float c = tris[idx[node.leftFirst + i]].centroid[a];
                cmin = (c < cmin) ? c : cmin;
                cmax = (cmax < c) ? c : cmax;
            }
            if (cmin == cmax) continue;
            aabb box[kBins]; int count[kBins] = {};
            float scale = kBins / (cmax - cmin);
            for (uint32_t i = 0; i < node.triCount; i++) {
                const Tri& t = tris[idx[node.leftFirst + i]];
                int b = (int)((t.centroid[a] - cmin) * scale);
                if (b > kBins - 1) b = kBins - 1;
                count[b]++;
                box[b].Grow(V(t.vertex0)); box[b].Grow(V(t.vertex1)); box[b].Grow(V(t.vertex2));
            }
            float areaL[kBins - 1], areaR[kBins - 1]; int nL[kBins - 1], nR[kBins - 1];
            aabb accL, accR; int sumL = 0, sumR = 0;
            for (int i = 0; i < kBins - 1; i++) {
                sumL += count[i]; nL[i] = sumL; accL.Grow(box[i]); areaL[i] = accL.Area();
                sumR += count[kBins - 1 - i]; nR[kBins - 2 - i] = sumR; accR.Grow(box[kBins - 1 - i]); areaR[kBins - 2 - i] = accR.Area();
            }
            scale = (cmax - cmin) / kBins;
            for (int i = 0; i < kBins - 1; i++) {
                const float cost = nL[i] * areaL[i] + nR[i] * areaR[i];
                if (cost < best) { axis = a; pos = cmin + scale * (i + 1); best = cost; }
            }
        }
        return best;
    }

    void run()
    {
        // explicit work list instead of recursion; children are numbered when their parent splits and the left
        // subtree is completed before the right one, which reproduces the reference's depth-first numbering (bvh.cpp:63-115)
        struct Item { uint32_t node, depth; };
        std::vector<Item> work; work.push_back({0u, 0u});
        while (!work.empty()) {
            const Item it = work.back(); work.pop_back();
            BVHNode& node = nodes[it.node];
            if (node.triCount <= 2) continue;
            int axis = 0; float pos = 0;
            const float splitCost = choosePlane(node, axis, pos);
            const float e0 = node.aabbMax[0] - node.aabbMin[0], e1 = node.aabbMax[1] - node.aabbMin[1], e2 = node.aabbMax[2] - node.aabbMin[2];
            const float leafCost = node.triCount * (e0 * e1 + e1 * e2 + e2 * e0);        // bvh.cpp:117-122
            if (splitCost >= leafCost) continue;
            int i = (int)node.leftFirst, j = i + (int)node.triCount - 1;
            while (i <= j) {
                if (tris[idx[i]].centroid[axis] < pos) i++;
                else { std::swap(idx[i], idx[j]); j--; }
            }
            const int leftCount = i - (int)node.leftFirst;
            if (leftCount == 0 || leftCount == (int)node.triCount) continue;
            const uint32_t L = used++, R = used++;
            nodes[L].leftFirst = node.leftFirst; nodes[L].triCount = (uint32_t)leftCount;
            nodes[R].leftFirst = (uint32_t)i; nodes[R].triCount = node.triCount - (uint32_t)leftCount;
            node.leftFirst = L; node.triCount = 0;
            fit(L); fit(R);
            if (it.depth > deepest) deepest = it.depth;
            work.push_back({R, it.depth + 1});
            work.push_back({L, it.depth + 1});
        }
    }
};

struct Key { uint32_t w[8]; bool operator==(const Key& o) const { return memcmp(w, o.w, sizeof(w)) == 0; } };
struct KeyHash { size_t operator()(const Key& k) const { uint64_t h = 0xcbf29ce484222325ull; for (uint32_t x : k.w) { h = (h ^ x) * 0x100000001b3ull; } return (size_t)h; } };

// unique-vertex table keyed on float equality (so -0 and +0 collapse to the first one seen) — model.cpp:16-54, blas_bvh.cpp:16-56
void Dedup(const MeshCorners& m, std::vector<float>& P, std::vector<float>& N, std::vector<float>& U, std::vector<uint32_t>& indices)
{
    std::unordered_map<Key, uint32_t, KeyHash> table;
    const size_t n = m.count();
    table.reserve(n);
    for (size_t c = 0; c < n; c++) {
        const float v[8] = {m.pos[3 * c], m.pos[3 * c + 1], m.pos[3 * c + 2], m.nrm[3 * c], m.nrm[3 * c + 1], m.nrm[3 * c + 2], m.uv[2 * c], m.uv[2 * c + 1]};
        Key k; bool nan = false;
        for (int i = 0; i < 8; i++) { uint32_t b; memcpy(&b, &v[i], 4); k.w[i] = (b == 0x80000000u) ? 0u : b; nan = nan || v[i] != v[i]; }
        if (nan) {      // a NaN vertex equals nothing: model.cpp:46-48 appends it, and model.cpp:50's second lookup misses again and yields a value-initialised 0
            P.insert(P.end(), v, v + 3); N.insert(N.end(), v + 3, v + 6); U.insert(U.end(), v + 6, v + 8);
            indices.push_back(0u);
            continue;
        }
        auto it = table.find(k);
        if (it == table.end()) {
            const uint32_t id = (uint32_t)(P.size() / 3);
            table.emplace(k, id);
            P.insert(P.end(), v, v + 3); N.insert(N.end(), v + 3, v + 6); U.insert(U.end(), v + 6, v + 8);
            indices.push_back(id);
        } else indices.push_back(it->second);
    }
}

} // namespace

void DedupVertices(const MeshCorners& m, std::vector<float>& P, std::vector<float>& N, std::vector<float>& U, std::vector<uint32_t>& indices) { Dedup(m, P, N, U, indices); }

void RefitSAH(const std::vector<Tri>& tris, std::vector<BVHNode>& nodes, const std::vector<uint32_t>& idx, uint32_t nodesUsed)
{
    for (int i = (int)nodesUsed - 1; i >= 0; i--) {
        if (i == 1) continue;                                              // bvh.cpp:28
        BVHNode& node = nodes[(size_t)i];
        if (node.triCount > 0) {                                           // leaf: tight bounds of its triangles (UpdateNodeBounds)
            float3 lo(1e30f), hi(-1e30f);
            for (uint32_t k = 0; k < node.triCount; k++) {
                const Tri& t = tris[idx[node.leftFirst + k]];
                lo = fminf3(lo, V(t.vertex0)); lo = fminf3(lo, V(t.vertex1)); lo = fminf3(lo, V(t.vertex2));
                hi = fmaxf3(hi, V(t.vertex0)); hi = fmaxf3(hi, V(t.vertex1)); hi = fmaxf3(hi, V(t.vertex2));
            }
            S(node.aabbMin, lo); S(node.aabbMax, hi);
        } else {                                                           // interior: union of the two children
            const BVHNode& l = nodes[node.leftFirst]; const BVHNode& r = nodes[node.leftFirst + 1];
            S(node.aabbMin, fminf3(V(l.aabbMin), V(r.aabbMin))); S(node.aabbMax, fmaxf3(V(l.aabbMax), V(r.aabbMax)));
        }
    }
}

void BuildSAH(std::vector<Tri>& triangles, std::vector<BVHNode>& nodes, std::vector<uint32_t>& triangleIndices, uint32_t& nodesUsed, uint32_t& maxDepth)
{
    if (triangles.empty()) throw std::runtime_error("BVH::Build: no triangles");
    triangleIndices.resize(triangles.size());
    for (size_t i = 0; i < triangles.size(); i++) triangleIndices[i] = (uint32_t)i;
    BVHNode zero; memset(&zero, 0, sizeof(zero));
    nodes.assign(triangles.size() * 2 - 1, zero);
    nodes[0].leftFirst = 0; nodes[0].triCount = (uint32_t)triangles.size();
    Builder b{triangles, nodes, triangleIndices};
    b.fit(0);
    b.run();
    nodesUsed = b.used; maxDepth = b.deepest;
}

BLASBVH::BLASBVH(int idx, const MeshCorners& mesh, const mat4& transform, const mat4& scaleMat)
{
    std::vector<float> P, N, U; std::vector<uint32_t> ind;
    Dedup(mesh, P, N, U, ind);
    objIdx = idx;
    for (size_t i = 0; i + 2 < ind.size(); i += 3) {
        Tri t; memset(&t, 0, sizeof(t));
        const uint32_t a = ind[i], b = ind[i + 1], c = ind[i + 2];
        S(t.vertex0, TransformPosition(V(&P[3 * a]), scaleMat));
        S(t.vertex1, TransformPosition(V(&P[3 * b]), scaleMat));
        S(t.vertex2, TransformPosition(V(&P[3 * c]), scaleMat));
        memcpy(t.normal0, &N[3 * a], 12); memcpy(t.normal1, &N[3 * b], 12); memcpy(t.normal2, &N[3 * c], 12);
        memcpy(t.uv0, &U[2 * a], 8); memcpy(t.uv1, &U[2 * b], 8); memcpy(t.uv2, &U[2 * c], 8);
        S(t.centroid, (V(t.vertex0) + V(t.vertex1) + V(t.vertex2)) * 0.3333f);       // blas_bvh.cpp:74
        t.objIdx = objIdx;
        triangles.push_back(t);
    }
    Build();
    SetTransform(transform);
}

void BLASBVH::SetTransform(const mat4& transform)   // blas_bvh.cpp:363-374
{
    T = transform;
    invT = transform.FastInvertedTransformNoScale();
    const float3 lo = V(bvhNodes[0].aabbMin), hi = V(bvhNodes[0].aabbMax);
    worldBounds = aabb();
    for (int i = 0; i < 8; i++)
        worldBounds.Grow(TransformPosition(float3(i & 1 ? hi.x : lo.x, i & 2 ? hi.y : lo.y, i & 4 ? hi.z : lo.z), transform));
}

TLASBVH::TLASBVH(const std::vector<BLASBVH*>& bvhList)
{
    blasCount = (uint32_t)bvhList.size();
    blas = bvhList;
    Build();
}

void TLASBVH::Build()   // tlas_bvh.cpp:17-70 — agglomerative clustering on merged half-area
{
    if (blasCount == 0) throw std::runtime_error("TLASBVH::Build: no BLAS");
    if (blasCount > 256) throw std::runtime_error("TLASBVH::Build: more than 256 BLAS (tlas_bvh.cpp:21)");
    TLASBVHNode zero; memset(&zero, 0, sizeof(zero));
    tlasNode.assign(2 * (size_t)blasCount, zero);
    int open[256], openCount = (int)blasCount;
    nodesUsed = 1;
    for (uint32_t i = 0; i < blasCount; i++) {
        open[i] = (int)nodesUsed;
        S(tlasNode[nodesUsed].aabbMin, blas[i]->worldBounds.bmin3);
        S(tlasNode[nodesUsed].aabbMax, blas[i]->worldBounds.bmax3);
        tlasNode[nodesUsed].BLAS = i;
        tlasNode[nodesUsed].leftRight = 0;
        nodesUsed++;
    }
    auto partner = [&](int A) {
        float smallest = 1e30f; int best = -1;
        for (int B = 0; B < openCount; B++) if (B != A) {
            const float3 hi = fmaxf3(V(tlasNode[open[A]].aabbMax), V(tlasNode[open[B]].aabbMax));
            const float3 lo = fminf3(V(tlasNode[open[A]].aabbMin), V(tlasNode[open[B]].aabbMin));
            const float3 e = hi - lo;
            const float area = e.x * e.y + e.y * e.z + e.z * e.x;
            if (area < smallest) { smallest = area; best = B; }
        }
        return best;
    };
    int A = 0, B = partner(A);
    while (openCount > 1) {
        const int C = partner(B);
        if (A == C) {
            const int ia = open[A], ib = open[B];
            TLASBVHNode& n = tlasNode[nodesUsed];
            n.leftRight = (uint32_t)ia + ((uint32_t)ib << 16);
            S(n.aabbMin, fminf3(V(tlasNode[ia].aabbMin), V(tlasNode[ib].aabbMin)));
            S(n.aabbMax, fmaxf3(V(tlasNode[ia].aabbMax), V(tlasNode[ib].aabbMax)));
            open[A] = (int)nodesUsed++;
            open[B] = open[openCount - 1];
            openCount--;
            B = partner(A);
        } else { A = B; B = C; }
    }
    tlasNode[0] = tlasNode[open[A]];
}

Model::Model(int idx, const MeshCorners& mesh, const mat4& transform)
{
    Dedup(mesh, positions, normals, uvs, indices);
    T = transform;
    invT = transform.FastInvertedTransformNoScale();
    objIdx = idx;
}

void Model::AppendTriangles(std::vector<Tri>& triangles) const   // model.cpp:62-80
{
    for (size_t i = 0; i + 2 < indices.size(); i += 3) {
        Tri t; memset(&t, 0, sizeof(t));
        const uint32_t a = indices[i], b = indices[i + 1], c = indices[i + 2];
        S(t.vertex0, TransformPosition(V(&positions[3 * a]), T));
        S(t.vertex1, TransformPosition(V(&positions[3 * b]), T));
        S(t.vertex2, TransformPosition(V(&positions[3 * c]), T));
        S(t.normal0, normalize(TransformVector(V(&normals[3 * a]), invT)));      // bug-compatible: invT, although T carries the scale
        S(t.normal1, normalize(TransformVector(V(&normals[3 * b]), invT)));
        S(t.normal2, normalize(TransformVector(V(&normals[3 * c]), invT)));
        memcpy(t.uv0, &uvs[2 * a], 8); memcpy(t.uv1, &uvs[2 * b], 8); memcpy(t.uv2, &uvs[2 * c], 8);
        S(t.centroid, (V(t.vertex0) + V(t.vertex1) + V(t.vertex2)) * 0.3333f);   // model.cpp:77
        t.objIdx = objIdx;
        triangles.push_back(t);
    }
}

} // namespace crt
