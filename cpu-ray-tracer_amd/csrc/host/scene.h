// scene.h — C++ front of the MI355X back end: the reference's scene / camera / renderer surface re-implemented on
// top of the C ABI (include/crt_abi.h).  A host application written against the reference's classes
//   FileScene / TLASFileScene   (infra/scene/file_scene.h, tlas_file_scene.h: ctor(path), FindNearest, members)
//   Camera                      (template/camera.h: GetPrimaryRay inputs, SetCameraState)
//   Renderer                    ("3. PathTracer/renderer.h": Init, Tick, ClearAccumulator, accumulator, spp, passes, energy)
// can switch to these with the same calls; construction (XML + OBJ + textures + SAH-BVH / TLAS build) stays on the CPU,
// FindNearest and the whole tile loop of Tick run on the GPU.
#pragma once
#include "accel.h"
#include "loaders.h"

#include <memory>
#include <string>
#include <vector>

namespace crt {

struct Ray {                       // template/ray.h subset that crosses the FindNearest seam
    Ray() = default;
    Ray(const float3& origin, const float3& direction, float distance = 1e34f, int idx = -1) : O(origin), D(direction), t(distance), objIdx(idx) {}
    float3 O, D;
    float t = 1e34f;
    float2 barycentric;
    int objIdx = -1, triIdx = -1;
    int traversed = 0, tested = 0;
    bool inside = false;
};

struct Material {                  // template/material.h
    bool isLight = false;
    float reflectivity = 0, refractivity = 0;
    float3 absorption;
    int texture = -1;              // index into BaseScene::textures (the reference owns a Texture through unique_ptr)
};

class BaseScene {                  // infra/scene/base_scene.h
public:
    virtual ~BaseScene() = default;
    // flattens and uploads the built scene to the device context (the one call the reference's classes lack)
    int Upload(crt_ctx* ctx);
    // in-place device update after instance motion (BLASBVH::SetTransform + TLASBVH::Build) / Refit: crt_update_scene with this scene's description
    int Update(crt_ctx* ctx, uint32_t what);
    // scene.FindNearest(ray): one ray through crt_find_nearest (use the batch ABI for many)
    void FindNearest(Ray& ray);
    float3 GetLightPos() const;
    float3 GetLightColor() const { return float3(24, 24, 22); }
    virtual int GetTriangleCount() const = 0;
    virtual int Kind() const = 0;
    // filled by the subclasses' constructors
    std::string sceneName;
    std::vector<Image> textures;   // [0] floor, [1] skydome, then material textures
    std::vector<Material*> materials;
    int objCount = 0, materialCount = 0, objIdUsed = 2;
    mat4 lightT, lightInvT; float lightSize = 0.5f;          // Quad light(0, 1)
    float3 floorN{0, 1, 0}; float floorD = 1, floorInvto = 1; // Plane floor(1, (0,1,0), 1, texW/100)
    float animTime = 0;
    void SetTime(float t) { animTime = t; }
protected:
    void LoadCommon(const SceneData& sd, const std::string& baseDir);
    virtual void Describe(crt_scene_desc& d, std::vector<crt_bvh>& bvhs, std::vector<int32_t>& objMat) = 0;
    struct DescKeep { std::vector<crt_bvh> bvhs; std::vector<int32_t> objMat; std::vector<crt_material> mats; std::vector<crt_texture> tex; };
    void BuildDesc(crt_scene_desc& d, DescKeep& keep);
    crt_ctx* bound = nullptr;
};

class FileScene : public BaseScene {      // infra/scene/file_scene.{h,cpp} with USE_BVH
public:
    explicit FileScene(const std::string& filePath, const std::string& baseDir = "");
    ~FileScene() override;
    int GetTriangleCount() const override { return acc.GetTriangleCount() * objCount; }   // file_scene.cpp:222-230 (UI figure, bug-compatible)
    int Kind() const override { return CRT_SCENE_FILE; }
    uint32_t GetMaxTreeDepth() const { return acc.maxDepth; }
    BVH acc;
    std::vector<Model*> models;
protected:
    void Describe(crt_scene_desc& d, std::vector<crt_bvh>& bvhs, std::vector<int32_t>& objMat) override;
};

class TLASFileScene : public BaseScene {  // infra/scene/tlas_file_scene.{h,cpp} with TLAS_USE_BVH
public:
    explicit TLASFileScene(const std::string& filePath, const std::string& baseDir = "");
    ~TLASFileScene() override;
    int GetTriangleCount() const override { int n = 0; for (auto* b : tlas.blas) n += b->GetTriangleCount(); return n; }
    int Kind() const override { return CRT_SCENE_TLAS; }
    TLASBVH tlas;
protected:
    void Describe(crt_scene_desc& d, std::vector<crt_bvh>& bvhs, std::vector<int32_t>& objMat) override;
};

class Camera {                     // template/camera.h:14-30, 61-73 (keyboard handling is the shell's business)
public:
    Camera(int width, int height) : aspect((float)width / (float)height)
    {
        camPos = float3(0, 0, -2); camTarget = float3(0, 0, -1);
        topLeft = float3(-aspect, 1, 0); topRight = float3(aspect, 1, 0); bottomLeft = float3(-aspect, -1, 0);
    }
    void SetCameraState(const float3& position, const float3& target);
    float aspect;
    float3 camPos, camTarget, topLeft, topRight, bottomLeft;
};

struct Surface { std::vector<uint32_t> pixels; int width = 0, height = 0; };   // template/surface.h: only `pixels` is touched by the path

class Renderer {                   // "3. PathTracer/renderer.{h,cpp}" — TheApp::Init / Tick
public:
    Renderer(BaseScene* scene, int width, int height, int device = 0);
    ~Renderer();
    void Init();
    void ClearAccumulator();
    void Tick(float deltaTime);
    void Render(int frames);       // `frames` Ticks in one submission (no per-frame read-back)
    void TickWhitted();            // one Tick of the Whitted-style renderer ("2. WhittedStyle/renderer.cpp":131-157): accumulator = Trace(primary)
    void Shutdown() {}
    // data members the shell / UI reads (renderer.h:46-53)
    std::vector<float> accumulatorStorage; float* accumulator = nullptr;   // float4[W*H], refreshed by Tick
    BaseScene* scene;
    Camera camera;
    int spp = 1, passes = 1;
    bool animating = false;
    float energy = 0, anim_time = 0;
    // performance report of Tick (renderer.cpp:159-161): running average of the frame time in ms, frames per second, primary rays per ms
    float m_avg = 10, m_fps = 0, m_rps = 0, m_alpha = 1;
    int depthLimit = 5;
    Surface* screen = nullptr; Surface ownScreen;
    crt_ctx* ctx = nullptr;
    int width, height, device;
private:
    void PushCamera();
};

} // namespace crt
