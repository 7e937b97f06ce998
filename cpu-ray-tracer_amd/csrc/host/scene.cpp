// scene.cpp — scene assembly on the CPU (as the reference's scene constructors do) and the Renderer facade.
#include "scene.h"
#include <chrono>

#include <stdexcept>

namespace crt {

namespace {

std::string resolve(const std::string& baseDir, const std::string& p)
{
    if (baseDir.empty() || (!p.empty() && p[0] == '/')) return p;       // the reference opens paths relative to the process CWD
    return baseDir + "/" + p;
}

void check(crt_ctx* ctx, int rc, const char* what)
{
    if (rc != CRT_OK) throw std::runtime_error(std::string(what) + ": " + crt_last_error(ctx));
}

} // namespace

// file_scene.cpp:4-43 / tlas_file_scene.cpp:4-38: light quad, floor plane, textures, materials
void BaseScene::LoadCommon(const SceneData& sd, const std::string& baseDir)
{
    textures.clear();
    textures.push_back(LoadImage(resolve(baseDir, sd.planeTextureLocation)));      // primitiveMaterials[1].textureDiffuse
    textures.push_back(LoadImage(resolve(baseDir, sd.skydomeLocation)));           // skydome
    objIdUsed = 2;
    lightSize = 1 * 0.5f;                                                            // Quad(0, 1): size = s * 0.5f
    floorN = float3(0, 1, 0); floorD = 1;
    floorInvto = 1.f / (float)(textures[0].width / 100);                            // integer division, file_scene.cpp:16
    const mat4 M1base = mat4::Translate(sd.lightPos);
    lightT = M1base; lightInvT = M1base.FastInvertedTransformNoScale();
    sceneName = sd.name;
    objCount = (int)sd.objects.size();
    materialCount = (int)sd.materials.size();
    for (Material* m : materials) delete m;
    materials.assign((size_t)materialCount, nullptr);
    for (int i = 0; i < materialCount; i++) {
        materials[i] = new Material();
        materials[i]->reflectivity = sd.materials[i].reflectivity;
        materials[i]->refractivity = sd.materials[i].refractivity;
        materials[i]->absorption = sd.materials[i].absorption;
        if (!sd.materials[i].textureLocation.empty()) {
            textures.push_back(LoadImage(resolve(baseDir, sd.materials[i].textureLocation)));
            materials[i]->texture = (int)textures.size() - 1;
        }
    }
    for (const ObjectData& o : sd.objects)
        if (o.materialIdx < 0 || o.materialIdx >= materialCount) throw std::runtime_error("scene file: material_idx out of range");
}

float3 BaseScene::GetLightPos() const   // file_scene.cpp:156-162
{
    const float3 c1 = TransformPosition(float3(-0.5f, 0, -0.5f), lightT), c2 = TransformPosition(float3(0.5f, 0, 0.5f), lightT);
    return (c1 + c2) * 0.5f - float3(0, 0.01f, 0);
}

// crt_scene_desc of the built scene (borrowed pointers into this object and into `k`, which must outlive the call that uses the description)
void BaseScene::BuildDesc(crt_scene_desc& d, DescKeep& k)
{
    memset(&d, 0, sizeof(d));
    Describe(d, k.bvhs, k.objMat);
    d.kind = Kind();
    d.bvhs = k.bvhs.data(); d.bvhCount = (uint32_t)k.bvhs.size();
    d.objMatIdx = k.objMat.empty() ? nullptr : k.objMat.data(); d.objCount = (uint32_t)k.objMat.size();
    k.mats.resize(materials.size());
    for (size_t i = 0; i < materials.size(); i++) {
        k.mats[i].reflectivity = materials[i]->reflectivity; k.mats[i].refractivity = materials[i]->refractivity;
        k.mats[i].absorption[0] = materials[i]->absorption.x; k.mats[i].absorption[1] = materials[i]->absorption.y; k.mats[i].absorption[2] = materials[i]->absorption.z;
        k.mats[i].texture = materials[i]->texture;
    }
    d.materials = k.mats.data(); d.materialCount = (uint32_t)k.mats.size();
    k.tex.resize(textures.size());
    for (size_t i = 0; i < textures.size(); i++) { k.tex[i].pixels = textures[i].pixels.data(); k.tex[i].width = textures[i].width; k.tex[i].height = textures[i].height; }
    d.textures = k.tex.data(); d.textureCount = (uint32_t)k.tex.size();
    d.floorTexture = 0; d.skyTexture = 1;
    memcpy(d.lightT, lightT.cell, 64); memcpy(d.lightInvT, lightInvT.cell, 64); d.lightSize = lightSize;
    d.floorN[0] = floorN.x; d.floorN[1] = floorN.y; d.floorN[2] = floorN.z; d.floorD = floorD; d.floorInvto = floorInvto;
}

int BaseScene::Upload(crt_ctx* ctx)
{
    crt_scene_desc d; DescKeep k;
    BuildDesc(d, k);
    const int rc = crt_upload_scene(ctx, &d);
    if (rc == CRT_OK) bound = ctx;
    return rc;
}

// after BLASBVH::SetTransform + TLASBVH::Build (CRT_UPDATE_TRANSFORMS) or Refit (CRT_UPDATE_BOUNDS): rewrites only those sections on the device
int BaseScene::Update(crt_ctx* ctx, uint32_t what)
{
    crt_scene_desc d; DescKeep k;
    BuildDesc(d, k);
    return crt_update_scene(ctx, &d, what);
}

void BaseScene::FindNearest(Ray& ray)
{
    if (!bound) throw std::runtime_error("FindNearest: scene not uploaded to a device context");
    crt_ray r; crt_hit h;
    r.O[0] = ray.O.x; r.O[1] = ray.O.y; r.O[2] = ray.O.z; r.D[0] = ray.D.x; r.D[1] = ray.D.y; r.D[2] = ray.D.z; r.inside = ray.inside;
    check(bound, crt_find_nearest(bound, &r, &h, 1), "crt_find_nearest");
    // the device query starts from t = 1e34 like Ray(origin, direction); a shorter incoming t only ever keeps the old hit
    if (h.t < ray.t) { ray.t = h.t; ray.objIdx = h.objIdx; ray.triIdx = h.triIdx; ray.barycentric.x = h.u; ray.barycentric.y = h.v; }
    ray.traversed += h.traversed; ray.tested += h.tested;
}

// ---- FileScene ---------------------------------------------------------------------------------------
FileScene::FileScene(const std::string& filePath, const std::string& baseDir)
{
    const SceneData sd = LoadSceneFile(filePath);
    LoadCommon(sd, baseDir);
    models.resize((size_t)objCount);
    for (int i = 0; i < objCount; i++) {
        const ObjectData& o = sd.objects[i];
        const mat4 T = mat4::Translate(o.position) * mat4::RotateX(o.rotation.x * Deg2Red) * mat4::RotateY(o.rotation.y * Deg2Red)
                     * mat4::RotateZ(o.rotation.z * Deg2Red) * mat4::Scale(o.scale);                       // file_scene.cpp:45-48
        models[i] = new Model(objIdUsed, LoadObj(resolve(baseDir, o.modelLocation)), T);
        models[i]->matIdx = o.materialIdx;
        objIdUsed++;
    }
    for (int i = 0; i < objCount; i++) models[i]->AppendTriangles(acc.triangles);
    acc.Build();
}
FileScene::~FileScene() { for (Model* m : models) delete m; for (Material* m : materials) delete m; }

void FileScene::Describe(crt_scene_desc&, std::vector<crt_bvh>& bvhs, std::vector<int32_t>& objMat)
{
    crt_bvh b; memset(&b, 0, sizeof(b));
    b.nodes = acc.bvhNodes.data(); b.nodesUsed = acc.nodesUsed;
    b.triangles = acc.triangles.data(); b.triCount = (uint32_t)acc.triangles.size();
    b.triangleIndices = acc.triangleIndices.data();
    b.objIdx = -1; b.matIdx = -1;
    bvhs.push_back(b);
    for (const Model* m : models) objMat.push_back(m->matIdx);       // materials[models[tri.objIdx - 2]->matIdx], file_scene.cpp:207
}

// ---- TLASFileScene -----------------------------------------------------------------------------------
TLASFileScene::TLASFileScene(const std::string& filePath, const std::string& baseDir)
{
    const SceneData sd = LoadSceneFile(filePath);
    LoadCommon(sd, baseDir);
    std::vector<BLASBVH*> blas((size_t)objCount);
    for (int i = 0; i < objCount; i++) {
        const ObjectData& o = sd.objects[i];
        const mat4 T = mat4::Translate(o.position) * mat4::RotateX(o.rotation.x * Deg2Red) * mat4::RotateY(o.rotation.y * Deg2Red)
                     * mat4::RotateZ(o.rotation.z * Deg2Red);                                               // tlas_file_scene.cpp:46-49
        const mat4 Sc = mat4::Scale(o.scale);
        blas[i] = new BLASBVH(objIdUsed, LoadObj(resolve(baseDir, o.modelLocation)), T, Sc);
        blas[i]->matIdx = o.materialIdx;
        objIdUsed++;
    }
    tlas = TLASBVH(blas);
}
TLASFileScene::~TLASFileScene() { for (BLASBVH* b : tlas.blas) delete b; for (Material* m : materials) delete m; }

void TLASFileScene::Describe(crt_scene_desc& d, std::vector<crt_bvh>& bvhs, std::vector<int32_t>&)
{
    for (const BLASBVH* bl : tlas.blas) {
        crt_bvh b; memset(&b, 0, sizeof(b));
        b.nodes = bl->bvhNodes.data(); b.nodesUsed = bl->nodesUsed;
        b.triangles = bl->triangles.data(); b.triCount = (uint32_t)bl->triangles.size();
        b.triangleIndices = bl->triangleIndices.data();
        b.objIdx = bl->objIdx; b.matIdx = bl->matIdx;
        memcpy(b.T, bl->T.cell, 64); memcpy(b.invT, bl->invT.cell, 64);
        bvhs.push_back(b);
    }
    d.tlasNodes = tlas.tlasNode.data(); d.tlasNodeCount = (uint32_t)tlas.tlasNode.size();
}

// ---- Camera ------------------------------------------------------------------------------------------
void Camera::SetCameraState(const float3& position, const float3& target)   // camera.h:61-73
{
    camPos = position; camTarget = target;
    const float3 ahead = normalize(camTarget - camPos);
    const float3 tmpUp(0, 1, 0);
    float3 right = normalize(cross(tmpUp, ahead));
    const float3 up = normalize(cross(ahead, right));
    right = normalize(cross(up, ahead));
    topLeft = camPos + 2 * ahead - aspect * right + up;
    topRight = camPos + 2 * ahead + aspect * right + up;
    bottomLeft = camPos + 2 * ahead - aspect * right - up;
}

// ---- Renderer ----------------------------------------------------------------------------------------
Renderer::Renderer(BaseScene* s, int w, int h, int dev) : scene(s), camera(w, h), width(w), height(h), device(dev) {}
Renderer::~Renderer() { if (ctx) crt_destroy(ctx); }

void Renderer::Init()   // renderer.cpp:8-13 + device context + one-time scene upload
{
    if (ctx) { crt_destroy(ctx); ctx = nullptr; }
    crt_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.width = width; cfg.height = height; cfg.depthLimit = depthLimit; cfg.device = device; cfg.tileCount = -1; cfg.tileStride = 1;
    const int rc = crt_create(&ctx, &cfg);
    if (rc != CRT_OK) { ctx = nullptr; throw std::runtime_error(std::string("crt_create: ") + crt_last_error(nullptr)); }
    check(ctx, scene->Upload(ctx), "crt_upload_scene");
    accumulatorStorage.assign((size_t)width * height * 4, 0.0f);
    accumulator = accumulatorStorage.data();
    if (!screen) { ownScreen.width = width; ownScreen.height = height; ownScreen.pixels.assign((size_t)width * height, 0u); screen = &ownScreen; }
    ClearAccumulator();
}

void Renderer::ClearAccumulator()   // renderer.cpp:15-18 (spp is NOT reset by the reference; callers that moved the camera rely on that)
{
    if (!ctx) return;
    check(ctx, crt_clear(ctx), "crt_clear");
    std::fill(accumulatorStorage.begin(), accumulatorStorage.end(), 0.0f);
}

void Renderer::PushCamera()
{
    const float p[3] = {camera.camPos.x, camera.camPos.y, camera.camPos.z}, tl[3] = {camera.topLeft.x, camera.topLeft.y, camera.topLeft.z};
    const float tr[3] = {camera.topRight.x, camera.topRight.y, camera.topRight.z}, bl[3] = {camera.bottomLeft.x, camera.bottomLeft.y, camera.bottomLeft.z};
    check(ctx, crt_set_camera(ctx, p, tl, tr, bl), "crt_set_camera");
}

void Renderer::Render(int frames)
{
    if (!ctx) throw std::runtime_error("Renderer::Render before Init");
    if (frames <= 0) return;
    if (animating) { scene->SetTime(anim_time); ClearAccumulator(); }
    PushCamera();
    check(ctx, crt_render(ctx, (uint32_t)spp, (uint32_t)frames, (uint32_t)passes), "crt_render");
    const int lastSpp = spp + (frames - 1) * passes;
    const float scale = 1.0f / (lastSpp + passes);                                  // renderer.cpp:119
    check(ctx, crt_read_accumulator(ctx, accumulator), "crt_read_accumulator");
    check(ctx, crt_resolve_screen(ctx, scale, screen ? screen->pixels.data() : nullptr, &energy), "crt_resolve_screen");
    spp += frames * passes;                                                          // renderer.cpp:167 (camera input belongs to the shell)
}

void Renderer::TickWhitted()
{
    if (!ctx) throw std::runtime_error("Renderer::TickWhitted before Init");
    PushCamera();
    check(ctx, crt_whitted_tick(ctx, screen ? screen->pixels.data() : nullptr), "crt_whitted_tick");
    check(ctx, crt_read_accumulator(ctx, accumulator), "crt_read_accumulator");
}

void Renderer::Tick(float deltaTime)   // renderer.cpp:144-168
{
    if (animating) anim_time += deltaTime * 0.002f;                                              // :147 (Render applies SetTime + ClearAccumulator)
    const auto t0 = std::chrono::steady_clock::now();
    Render(1);                                                                                  // the tile jobs (:149-153) + energy (:155-157)
    // performance report - running average - ms, frames/s, primary rays per ms (:159-161; the UI labels the last one "Mrays/s")
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m_avg = (1 - m_alpha) * m_avg + m_alpha * ms;
    if (m_alpha > 0.05f) m_alpha *= 0.75f;
    m_fps = 1000.0f / m_avg; m_rps = (float)(width * height) / m_avg;
}

} // namespace crt
