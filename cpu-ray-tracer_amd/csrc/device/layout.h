// layout.h — device-side (HBM) scene layout shared by the uploader (host) and the kernels.
//
// All records are sized and aligned for 16-byte vector loads (global_load_dwordx4):
//   NodePair  64 B  both children of an interior node in one aligned 64-byte line
//   LeafTri   48 B  Möller–Trumbore operands in leaf order (the triangleIndices indirection is resolved at upload)
//   ShadeTri  64 B  normals + uvs of a triangle, addressed by the reference's triIdx
//   TlasNode  32 B  reference TLASBVHNode layout
//   Instance 128 B  per-BLAS transform rows + array bases
#pragma once
#include <stdint.h>

namespace crt {

// packed node reference carried in registers / on the traversal stack (32 bit); all indices are GLOBAL (the
// uploader folds each BLAS's array bases in), so a reference alone identifies the record to fetch.
//   bits 31..30 = 10 : BVH / BLAS interior, bits 0..29 = NodePair index
//   bits 31..30 = 00 : BVH / BLAS leaf,     bits 24..29 = triCount (1..63), bits 0..23 = first LeafTri slot;  0 = "traversal done"
//   bits 31..30 = 01 : TLAS interior,       bits 0..14 = left child TlasNode index, bits 15..29 = right child
//   bits 31..30 = 11 : TLAS leaf,           bits 0..15 = BLAS (Instance) index;  0xFFFFFFFF = "return to TLAS level" stack marker
constexpr uint32_t kRefInterior = 0x80000000u;
constexpr uint32_t kRefTlasBit = 0x40000000u;
constexpr uint32_t kRefTlasInterior = 0x40000000u;
constexpr uint32_t kRefTlasLeaf = 0xC0000000u;
constexpr uint32_t kRefReturn = 0xFFFFFFFFu;
constexpr uint32_t kRefDone = 0u;
constexpr uint32_t kMaxLeafTris = 63u;
constexpr uint32_t kMaxLeafSlots = 1u << 24;
constexpr uint32_t kMaxPairs = 1u << 30;

struct alignas(16) NodeChild { float lo[3]; uint32_t ref; float hi[3]; uint32_t pad; };   // 32 B
struct alignas(64) NodePair { NodeChild c[2]; };                                           // 64 B

struct alignas(16) LeafTri {              // 48 B
    float v0[3]; uint32_t triIdx;         // vertex0, reference triangle index (written to Ray::triIdx)
    float e1[3]; int32_t objIdx;          // vertex1 - vertex0, hit object id (tri.objIdx or BLASBVH::objIdx)
    float e2[3]; uint32_t pad;            // vertex2 - vertex0
};

struct alignas(16) ShadeTri {             // 64 B
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    int32_t objIdx;
};

struct alignas(16) TlasNode { float lo[3]; uint32_t ref; float hi[3]; uint32_t pad; };         // 32 B; ref = packed reference of THIS node

struct alignas(16) Instance {             // 128 B; the first 64 B are what entering the BLAS needs (one record fetch)
    float invT[12];                       // rows 0..2 of BLASBVH::invT (ray -> object space)
    uint32_t shadeBase;
    int32_t matIdx;
    uint32_t rootRef;                     // packed (global) reference of the BLAS's node 0
    int32_t objIdx;
    float T[12];                          // rows 0..2 of BLASBVH::T    (normal -> world space)
    uint32_t pad[4];
};

struct alignas(16) Material {             // 32 B
    float reflectivity, refractivity;
    float absorption[3];
    int32_t tex;                          // texture id or -1
    int32_t isLight;
    uint32_t pad;
};

struct TexDesc { uint32_t offset; int32_t w, h; uint32_t pad; };   // offset in texels into the texel pool

struct Scene {                            // passed to the kernels BY VALUE (kernel argument segment -> scalar loads, global pointers)
    int32_t kind;                         // 0 FileScene, 1 TLASFileScene
    int32_t depthLimit;
    // camera (template/camera.h)
    float camPos[3], topLeft[3], topRight[3], bottomLeft[3];
    float invW, invH;                     // 1.0f / SCRWIDTH, 1.0f / SCRHEIGHT
    int32_t W, H;
    // light quad / floor plane
    float lightInvT[12]; float lightNrm[3]; float lightSize;
    float floorN[3]; float floorD; float floorInvto;
    int32_t floorTex, skyTex;
    // pools
    const uint32_t* texels; const TexDesc* tex;
    const Material* mats;
    const NodePair* pairs; const LeafTri* leaf; const ShadeTri* shade;
    const int32_t* objMat;                // FileScene: object id - 2 -> material
    uint32_t rootRef;                     // packed reference of the root (BVH node 0 / TLAS node 0)
    const TlasNode* tlas; const Instance* inst;
    uint32_t stackDepth;                  // dwords per lane of the LDS traversal stack (BVH height + TLAS height + 1 marker + slack)
    uint32_t bvhStack;                    // of which the BVH part (find_nearest_kernel keeps the TLAS entries above it)
};

struct Counters { unsigned long long v[8]; };   // order = crt_counters

} // namespace crt
