// SceneManager.h -- host mirror of the scene-description part of Path-Tracing/SceneManager.{h,cpp} (row N2): how the
// reference turns "these files + this skybox + these flags" into a Scene.
//
//   SceneLoader            SceneManager.h:16-23    something that fills a SceneBuilder
//   CombinedSceneLoader    SceneManager.h:25-46, SceneManager.cpp:11-64    any number of model files imported into ONE
//                          builder (SceneImporter::AddFile each, sharing the texture-slot mapping), an equirectangular
//                          skybox image, HasDxNormalTextures, ForceFullTextureSize
//   SceneDescription       SceneManager.h:48-57, SceneManager.cpp:66-94    the aggregate the scene registry is written in
//                          (ExampleScenes.cpp:87-236: Intel Sponza = three glTF components + an .hdr sky + DX normal
//                          maps; the ORCA scenes = one file + NVIDIAOrcaTextureMapping + both flags); ToLoader() drops
//                          components and skies that do not exist on disk
// The registry itself (scene groups, the active scene, background loading) belongs to the application and is not
// mirrored.
#pragma once

#include <filesystem>
#include <memory>
#include <optional>
#include <span>
#include <vector>

#include "Scene.h"
#include "SceneImporter.h"

namespace PathTracing
{

class SceneLoader
{
public:
    virtual ~SceneLoader() = default;
    virtual void Load(SceneBuilder &sceneBuilder) = 0;
};

class CombinedSceneLoader : public SceneLoader
{
public:
    ~CombinedSceneLoader() override = default;

    void AddTextureMapping(TextureMapping mapping);
    void AddComponent(const std::filesystem::path &path);
    void AddComponents(std::span<const std::filesystem::path> paths);
    void AddSkybox2D(const std::filesystem::path &path);
    void SetDxNormalTextures();
    void ForceFullTextureSize();

    [[nodiscard]] bool HasContent() const;

    void Load(SceneBuilder &sceneBuilder) override;

private:
    TextureMapping m_TextureMapping;
    std::vector<std::filesystem::path> m_ComponentPaths;
    std::optional<std::filesystem::path> m_SkyboxPath;
    bool m_HasDxNormalTextures = false;
    bool m_ForceFullTextureSize = false;
};

struct SceneDescription
{
    std::vector<std::filesystem::path> ComponentPaths;
    std::optional<std::filesystem::path> SkyboxPath;
    TextureMapping Mapping;
    bool HasDxNormalTextures = false;
    bool ForceFullTextureSize = false;

    [[nodiscard]] std::unique_ptr<CombinedSceneLoader> ToLoader() const;

    // {"components": ["a.gltf", ...], "skybox": "sky.hdr", "mapping": "orca" | "none", "dxNormalTextures": true,
    //  "forceFullTextureSize": true}; relative paths are taken from `base`.  (The reference writes its descriptions as
    // C++ aggregates; this is the same aggregate for callers behind the C-ABI.)
    static SceneDescription FromJson(const std::string &text, const std::filesystem::path &base = {});
};

// ExampleScenes.cpp:113-118: the slot remap of the NVIDIA ORCA assets (Sun Temple, Bistro, Emerald Square, Zero Day)
MetallicRoughnessTextureMapping NVIDIAOrcaTextureMapping();

}
