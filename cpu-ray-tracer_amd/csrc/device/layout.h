// layout.h — device-side (HBM) scene layout shared by the uploader (host) and the kernels.
//
// ONE geometry buffer holds every record the traversal touches, so a record address is `geom + 32-bit byte offset`
// (scalar base register + per-lane offset: no 64-bit address arithmetic in the hot loop).  Records are sized and
// aligned for 16-byte vector loads (global_load_dwordx4); a traversal step always fetches the 64 bytes at its record.
//   NodePair  64 B  both children of an interior node in one aligned 64-byte line
//   LeafTri   48 B  Möller–Trumbore operands in leaf order (the triangleIndices indirection is resolved at upload)
//   TlasNode  32 B  reference TLASBVHNode layout, leftRight/BLAS replaced by the node's packed reference; the TLAS child pairs section holds, for every
//                   TLAS interior node, its two child TlasNodes side by side (NodePair layout)
//   Instance 128 B  per-BLAS: invT rows + ids in the first 64 B (what entering the BLAS needs), then T rows
//   ShadeTri  64 B  normals + uvs + material of a triangle, addressed by the global shade index carried in LeafTri
#pragma once
#include <stdint.h>

namespace crt {

// packed node reference carried in registers / on the traversal stack (32 bit)
//   bits 31..30 = 10 : BVH / BLAS interior, bits 0..29 = offset of its NodePair in the geometry buffer, in 16-byte units
//   bits 31..30 = 00 : BVH / BLAS leaf,     bits 0..29 = offset of its first LeafTri, in 16-byte units (never 0);  0 = "traversal done"
//   bits 31..30 = 01 : TLAS interior,       bits 0..14 = left child TlasNode index, bits 15..29 = right child
//   bits 31..30 = 11 : TLAS leaf,           bits 0..15 = BLAS (Instance) index;  0xFFFFFFFF = "return to TLAS level" stack marker
constexpr uint32_t kRefInterior = 0x80000000u;
constexpr uint32_t kRefTlasBit = 0x40000000u;
constexpr uint32_t kRefTlasInterior = 0x40000000u;
constexpr uint32_t kRefTlasLeaf = 0xC0000000u;
constexpr uint32_t kRefReturn = 0xFFFFFFFFu;
constexpr uint32_t kRefDone = 0u;
constexpr uint32_t kRefOffsetMask = 0x3fffffffu;
constexpr uint64_t kMaxGeomBytes = 1ull << 32;     // 32-bit byte offsets
// render_narrow_kernel only (FileScene): a reference inside the treetop copy (Scene::topOff) — bits 31..30 = 01 (the TLAS-interior tag, unused in a single-level
// scene), bits 0..29 = index of the treetop entry
constexpr uint32_t kRefTop = 0x40000000u;
constexpr uint64_t kTreetopMaxPairs = 512;         // 32 KB of LDS per workgroup of four narrow wavefronts

// 16-bit form of the same reference (`ref16`), used by render_pool_kernel for `cur` and for its traversal stack (2 bytes per entry: LDS is
// what limits the waves per SIMD): the same tags in bits 15..14 and a record INDEX in bits 13..0 instead of an offset
//   10 : BVH / BLAS interior, index of its NodePair                 (record at            index * 64)
//   00 : BVH / BLAS leaf,     index of its first LeafTri + 1; 0 = "traversal done"   (record at leafOff + (index - 1) * 48; the next triangle of the leaf is ref16 + 1)
//   01 : TLAS interior,       index of its TLAS child pair         (record at tlasPairOff + index * 64)
//   11 : TLAS leaf,           BLAS (Instance) index                (record at instOff + index * 128);  0xFFFF = "return to TLAS level" marker
// Scene::ref16ok tells whether the scene fits (<= 16383 node pairs and < 16383 triangles over all BVHs); otherwise the host launches render_tiles_kernel.
constexpr uint32_t kRef16Interior = 0x8000u, kRef16TlasBit = 0x4000u, kRef16TlasLeaf = 0xC000u, kRef16TagMask = 0xC000u, kRef16IndexMask = 0x3fffu;
constexpr uint32_t kRef16Return = 0xFFFFu, kRef16MaxIndex = 0x3ffeu;

struct alignas(16) NodeChild { float lo[3]; uint32_t ref; float hi[3]; uint32_t ref16; };   // 32 B; ref / ref16 = the child's packed reference, 32- and 16-bit form
struct alignas(64) NodePair { NodeChild c[2]; };                                           // 64 B

struct alignas(16) LeafTri {              // 48 B = 3 offset units
    float v0[3]; uint32_t shadeIdx;       // vertex0; global index of the triangle's ShadeTri (= shadeBase of its BVH + reference triIdx)
    float e1[3]; int32_t objIdx;          // vertex1 - vertex0; hit object id (tri.objIdx or BLASBVH::objIdx)
    float e2[3]; uint32_t remain;         // vertex2 - vertex0; triangles left in this leaf including this one (>= 1)
};

struct alignas(16) ShadeTri {             // 64 B
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    int32_t mat;                          // index into Scene::mats ([0] light, [1] floor, 2.. scene materials)
};

struct alignas(16) TlasNode { float lo[3]; uint32_t ref; float hi[3]; uint32_t ref16; };       // 32 B; ref / ref16 = packed reference of THIS node

struct alignas(16) Instance {             // 128 B
    float invT[12];                       // rows 0..2 of BLASBVH::invT (ray -> object space)
    uint32_t shadeBase;                   // first ShadeTri of this BLAS (find_nearest reports triIdx = shadeIdx - shadeBase)
    uint32_t rootRef16;                   // 16-bit reference of the BLAS's node 0
    uint32_t rootRef;                     // packed reference of the BLAS's node 0
    int32_t objIdx;
    float T[12];                          // rows 0..2 of BLASBVH::T    (normal -> world space)
    uint32_t pad[4];
};

struct alignas(16) Material {             // 32 B: material + its texture descriptor in one record
    float reflectivity, refractivity;
    float absorption[3];
    uint32_t texOffset;                   // first texel in the pooled texel array
    int32_t texW, texH;                   // texW == 0: untextured (albedo 1,1,1)
};

struct Scene {                            // passed to the kernels BY VALUE (kernel argument segment -> scalar loads, global pointers)
    int32_t kind;                         // 0 FileScene, 1 TLASFileScene
    int32_t depthLimit;
    // camera (template/camera.h)
    float camPos[3], topLeft[3], topRight[3], bottomLeft[3];
    float invW, invH;                     // 1.0f / SCRWIDTH, 1.0f / SCRHEIGHT
    int32_t W, H;
    // light quad / floor plane
    float lightInvT[12]; float lightNrm[3]; float lightSize;
    float lightPos[3];                    // GetLightPos() (file_scene.cpp:156-162): the Whitted integrator's point light
    float floorN[3]; float floorD; float floorInvto;
    Material floorMat;                    // primitiveMaterials[1]: diffuse, textured
    uint32_t skyOffset; int32_t skyW, skyH;
    // pools
    const char* geom;                     // pairs | leaf tris | TLAS nodes | instances | shade records
    uint32_t tlasOff, instOff, shadeOff;  // byte offsets of those sections inside geom
    uint32_t leafOff, tlasPairOff;        // ... of the leaf triangles and of the TLAS child pairs (NodePair layout: the two child TlasNodes of every TLAS interior node side by side)
    uint32_t rootRef16, ref16ok;          // 16-bit form of rootRef; 1: every reference of the scene has a 16-bit form (render_pool_kernel can run)
    const uint32_t* texels;
    const Material* mats;
    uint32_t rootRef;                     // packed reference of the root (BVH node 0 / TLAS node 0)
    uint32_t stackDepth;                  // dwords per lane of the LDS traversal stack (BVH height + TLAS height + 1 marker + slack)
    uint32_t bvhStack;                    // of which the BVH part (find_nearest_kernel keeps the TLAS entries above it)
    uint32_t lightAxis, floorAxisY;       // 1: light invT has an identity rotation block / floor normal is exactly (0,1,0): short quad / plane tests (kernels.hip)
    uint32_t topOff, topCount;            // FileScene: the treetop (first child pairs in breadth-first order, references rewritten: kRefTop) inside geom; 0 pairs: none
    uint32_t rootIsPair;                  // 1: rootPair holds the root's two children (always, unless the root itself is a leaf)
    float rootPair[16];                   // NodePair of the root (BVH: its child pair; TLAS: its two child TlasNodes, same 2 x {lo, ref, hi, -} layout)
};

struct Counters { unsigned long long v[8]; };   // order = crt_counters

} // namespace crt
