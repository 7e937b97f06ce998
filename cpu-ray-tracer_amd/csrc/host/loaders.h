// loaders.h — host-side asset readers of the C++ front: Wavefront OBJ, scene XML, textures.
// They replace the three third-party parsers the reference vendors (lib/tiny_obj_loader.h, lib/rapidxml-1.13,
// lib/stb_image.h) for exactly the subset the path tracer's scenes use.  Errors are reported by exception
// (std::runtime_error) inside the C++ front and converted to CRT_ERR_IO at the C ABI.
#pragma once
#include "accel.h"

#include <string>
#include <vector>

namespace crt {

// OBJ -> triangle corners with tinyobjloader v2.0 semantics (triangulate = true): v / vn / vt / f, relative indices,
// quads split along the shorter diagonal, larger polygons ear-clipped.  Groups, materials and smoothing are ignored
// (the reference never reads them: infra/model.cpp:16-54).
MeshCorners LoadObj(const std::string& path);

// 8-bit image -> 0x00RRGGBB texels, top row first (Texture::LoadFromFile, template/texture.h:15-39).
// Supported: PNG (8/16-bit, grey / RGB / palette, +alpha, interlaced or not), TGA (true-colour / grey, raw or RLE),
// JPEG (Huffman sequential and progressive, 8 bit, grey / YCbCr / RGB, any sampling factors; decoded with stb_image's arithmetic so the
// texels are the reference's), binary PPM (P6) / PGM (P5).
struct Image { int width = 0, height = 0; std::vector<uint32_t> pixels; };
Image LoadImage(const std::string& path);

// scene file schema of LoadSceneFile (infra/scene/file_scene.cpp:64-135, tlas_file_scene.h:18-40)
struct MaterialData { float reflectivity = 0, refractivity = 0; float3 absorption; std::string textureLocation; };
struct ObjectData { std::string modelLocation; int materialIdx = 0; float3 position, rotation, scale; };
struct SceneData {
    std::string name; float3 lightPos; std::string planeTextureLocation, skydomeLocation;
    std::vector<ObjectData> objects; std::vector<MaterialData> materials;
};
SceneData LoadSceneFile(const std::string& path);

} // namespace crt
