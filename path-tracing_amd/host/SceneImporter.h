// SceneImporter.h -- host mirror of Path-Tracing/SceneImporter.{h,cpp} (row N2) for glTF 2.0, binary FBX (FbxReader.h) and Wavefront OBJ (ObjReader.h).
//
// The reference hands every format to assimp and then walks the aiScene (SceneImporter.cpp:1048-1114).  assimp is
// not available, so this importer reads glTF 2.0 itself (.gltf with external / data-URI buffers, .glb) and then
// follows the reference's pipeline step for step:
//   LoadSceneNodes   pre-order flatten through an explicit stack (children come out in reverse order, :671-706)
//   LoadMaterials    material model choice (:300-319), texture slots (:33-102) incl. the ORCA slot remaps
//                    (TextureMapping), emissive / transmission blocks (:104-167)
//   LoadMeshes       one geometry per primitive, identical index ranges shared (:402-413), tangent-space repair
//                    (:520-528), skinned primitives into the animated vertex / index arrays, bones (:420-453)
//   LoadModels       everything below an instance root (the scene root or an animated node) merges into ONE model
//                    with baked mesh transforms; skinned meshes become their own instance (:708-837)
//   LoadAnimations / LoadLights / LoadCameras (:840-1040; KHR_lights_punctual, perspective cameras)
// with the conventions assimp's glTF2 importer applies on the way (v -> 1 - v from aiProcess_FlipUVs, key times in
// milliseconds at 1000 ticks per second, the metallic-roughness texture bound to both the roughness and the
// metalness slot, attenuation (0, 0, 1) for point lights).
#pragma once

#include <filesystem>
#include <variant>

#include "Scene.h"

namespace PathTracing
{

// SceneImporter.h:11-36
struct MetallicRoughnessTextureMapping
{
    TextureType ColorTexture;
    TextureType NormalTexture;
    TextureType RoughnessTexture;
    TextureType MetallicTexture;
};

struct SpecularGlossinessTextureMapping
{
    TextureType ColorTexture;
    TextureType NormalTexture;
    TextureType SpecularTexture;
    TextureType GlossinessTexture;
};

struct PhongTextureMapping
{
    TextureType ColorTexture;
    TextureType NormalTexture;
    TextureType SpecularTexture;
    TextureType ShininessTexture;
};

using TextureMapping = std::variant<std::monostate, MetallicRoughnessTextureMapping, SpecularGlossinessTextureMapping, PhongTextureMapping>;

class SceneImporter
{
public:
    static void Init() {}
    static void Shutdown() {}

    static SceneBuilder &AddFile(SceneBuilder &builder, const std::filesystem::path &path, TextureMapping mapping = std::monostate());
};

}
