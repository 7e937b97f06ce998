// primitive_scene.cpp — host side of PrimitiveScene (infra/scene/primitive_scene.cpp:4-68): the constructor's constants and SetTime(t)'s animation, producing the
// crt_primitive_scene the C ABI takes.  Same names and members as the reference class; the matrices are built with hmath.h (factories and
// FastInvertedTransformNoScale pinned to the real tmplmath.h, products restated from tmplmath.cpp:109-122) and mat4::Inverted (tmplmath.h:769-813: the MESA
// cofactor formula, its 96 triple products in the reference's order).  PARITY UNPINNED like the device side (render_prim.hip).
#include "../../../include/crt_host.h"
#include "hmath.h"
#include "loaders.h"

#include <cmath>
#include <cstring>
#include <string>

namespace crt {

static mat4 Inverted(const mat4& m)
{
    static const signed char T[16][6][4] = {
        {{1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {1, 9, 7, 14}, {1, 13, 6, 11}, {-1, 13, 7, 10}}, {{-1, 1, 10, 15}, {1, 1, 11, 14}, {1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {1, 13, 3, 10}},
        {{1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {1, 5, 3, 14}, {1, 13, 2, 7}, {-1, 13, 3, 6}}, {{-1, 1, 6, 11}, {1, 1, 7, 10}, {1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {1, 9, 3, 6}},
        {{-1, 4, 10, 15}, {1, 4, 11, 14}, {1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {1, 12, 7, 10}}, {{1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {1, 8, 3, 14}, {1, 12, 2, 11}, {-1, 12, 3, 10}},
        {{-1, 0, 6, 15}, {1, 0, 7, 14}, {1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {1, 12, 3, 6}}, {{1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {1, 4, 3, 10}, {1, 8, 2, 7}, {-1, 8, 3, 6}},
        {{1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {1, 8, 7, 13}, {1, 12, 5, 11}, {-1, 12, 7, 9}}, {{-1, 0, 9, 15}, {1, 0, 11, 13}, {1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {1, 12, 3, 9}},
        {{1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {1, 4, 3, 13}, {1, 12, 1, 7}, {-1, 12, 3, 5}}, {{-1, 0, 5, 11}, {1, 0, 7, 9}, {1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {1, 8, 3, 5}},
        {{-1, 4, 9, 14}, {1, 4, 10, 13}, {1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {1, 12, 6, 9}}, {{1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {1, 8, 2, 13}, {1, 12, 1, 10}, {-1, 12, 2, 9}},
        {{-1, 0, 5, 14}, {1, 0, 6, 13}, {1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {1, 12, 2, 5}}, {{1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {1, 4, 2, 9}, {1, 8, 1, 6}, {-1, 8, 2, 5}}};
    float inv[16];
    for (int i = 0; i < 16; i++) {
        float acc = 0;
        for (int k = 0; k < 6; k++) {
            const float t = m.cell[T[i][k][1]] * m.cell[T[i][k][2]] * m.cell[T[i][k][3]];
            if (k == 0) acc = T[i][k][0] < 0 ? -t : t; else acc = T[i][k][0] < 0 ? acc - t : acc + t;
        }
        inv[i] = acc;
    }
    const float det = m.cell[0] * inv[0] + m.cell[1] * inv[4] + m.cell[2] * inv[8] + m.cell[3] * inv[12];
    mat4 r;
    if (det != 0) { const float invdet = 1.0f / det; for (int i = 0; i < 16; i++) r.cell[i] = inv[i] * invdet; }
    return r;
}

static const float kPI = 3.14159265358979323846264f;       // template/common.h:8

class PrimitiveScene {
public:
    crt_primitive_scene s{};
    Image red, blue;
    explicit PrimitiveScene(const std::string& assets)        // primitive_scene.cpp:4-42 (+ the images Plane::GetAlbedo loads lazily, primitives.h:147-166)
    {
        memset(&s, 0, sizeof(s));
        s.quadSize = 1 * 0.5f;                                  // Quad(0, 1)
        const float3 size{1.15f, 1.15f, 1.15f}, pos{0, 0, 0};
        const float3 lo = pos - 0.5f * size, hi = pos + 0.5f * size;
        s.cubeMin[0] = lo.x; s.cubeMin[1] = lo.y; s.cubeMin[2] = lo.z; s.cubeMax[0] = hi.x; s.cubeMax[1] = hi.y; s.cubeMax[2] = hi.z;
        const float a = 0.8f, b = 0.25f;                        // Torus(10, 0.8f, 0.25f)
        s.torusRc2 = a * a; s.torusRt2 = b * b; s.torusR2 = (a + b) * (a + b);
        const mat4 T = mat4::Translate({-0.25f, 0, 2}) * mat4::RotateX(kPI / 4), invT = Inverted(T);
        memcpy(s.torusT, T.cell, 64); memcpy(s.torusInvT, invT.cell, 64);
        s.reflectivity[1] = 1.0f; s.refractivity[3] = 1.0f; s.absorption[3][0] = 0.5f; s.absorption[3][2] = 0.5f; s.reflectivity[6] = 0.3f; s.refractivity[10] = 1.0f;
        if (!assets.empty()) {
            red = LoadImage(assets + "/red.png"); blue = LoadImage(assets + "/blue.png");
            s.red.pixels = red.pixels.data(); s.red.width = red.width; s.red.height = red.height;
            s.blue.pixels = blue.pixels.data(); s.blue.width = blue.width; s.blue.height = blue.height;
        }
        SetTime(0);
    }
    void SetTime(float t)                                      // primitive_scene.cpp:44-68
    {
        const mat4 M1base = mat4::Translate({0, 2.6f, 2});
        const mat4 M1 = M1base * mat4::RotateZ(sinf(t * 0.6f) * 0.1f) * mat4::Translate({0, -0.9f, 0});
        const mat4 invM1 = M1.FastInvertedTransformNoScale();
        memcpy(s.quadT, M1.cell, 64); memcpy(s.quadInvT, invM1.cell, 64);
        const mat4 M2base = mat4::RotateX(kPI / 4) * mat4::RotateZ(kPI / 4);
        const mat4 M2 = mat4::Translate({1.8f, 0, 2.5f}) * mat4::RotateY(t * 0.5f) * M2base;
        const mat4 invM2 = M2.FastInvertedTransformNoScale();
        memcpy(s.cubeM, M2.cell, 64); memcpy(s.cubeInvM, invM2.cell, 64);
        const float f = fmodf(t, 2.0f) - 1, tm = 1 - f * f;
        s.spherePos[0] = -1.8f; s.spherePos[1] = -0.4f + tm; s.spherePos[2] = 1;
    }
};

} // namespace crt

extern "C" {
extern void crt_host_set_error(const char* msg);

int crt_host_primitive_scene_create(const char* assetsDir, void** out)
{
    if (!out) return CRT_ERR_INVALID;
    try { *out = new crt::PrimitiveScene(assetsDir ? assetsDir : ""); return CRT_OK; }
    catch (const std::exception& e) { crt_host_set_error(e.what()); *out = nullptr; return CRT_ERR_IO; }
}
void crt_host_primitive_scene_free(void* h) { delete static_cast<crt::PrimitiveScene*>(h); }
int crt_host_primitive_scene_set_time(void* h, float t) { if (!h) return CRT_ERR_INVALID; static_cast<crt::PrimitiveScene*>(h)->SetTime(t); return CRT_OK; }
int crt_host_primitive_scene_desc(void* h, crt_primitive_scene* out) { if (!h || !out) return CRT_ERR_INVALID; *out = static_cast<crt::PrimitiveScene*>(h)->s; return CRT_OK; }
int crt_host_primitive_scene_upload(void* h, crt_ctx* ctx) { if (!h || !ctx) return CRT_ERR_INVALID; return crt_upload_primitive_scene(ctx, &static_cast<crt::PrimitiveScene*>(h)->s); }
}
