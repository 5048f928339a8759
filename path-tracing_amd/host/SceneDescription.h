// SceneDescription.h -- "these model files + this sky image + these flags" as ONE scene, for callers behind the C-ABI
// (scene names "description:<json>" / "description:@file.json", include/ptx_host.h).
//
// The reference writes such aggregates in C++ in its scene registry (ExampleScenes.cpp:87-236: Intel Sponza = three glTF
// components + an .hdr sky + DX normal maps; the NVIDIA ORCA scenes = one file + a texture-slot remap + both flags) and
// hands them to its scene manager, which belongs to the application (registry, scene groups, background loading) and is
// NOT mirrored here (out of scope).  What the hot path needs from it is only the outcome -- every component
// imported into one SceneBuilder under one texture mapping, the equirectangular sky, the two flags -- which is the one
// function below.
#pragma once

#include <filesystem>
#include <string>
#include <vector>

#include "Scene.h"
#include "SceneImporter.h"

namespace PathTracing
{

struct SceneDescription
{
    std::vector<std::filesystem::path> Components; // model files (.gltf / .glb / .fbx / .obj), imported in this order
    std::filesystem::path Sky;                     // equirectangular image; empty: the clear colour
    TextureMapping Mapping;                        // which imported texture slot feeds which material slot
    bool DxNormalTextures = false;                 // normal maps in the DirectX convention (green flipped)
    bool FullSizeTextures = false;                 // exempt the scene from the texture memory budget

    // {"components": ["a.gltf", ...], "skybox": "sky.hdr", "mapping": "orca" | "none", "dxNormalTextures": true,
    //  "forceFullTextureSize": true}; relative paths are taken from `base`.
    static SceneDescription Parse(const std::string &json, const std::filesystem::path &base = {});

    // Imports what exists on disk into `builder` and returns the paths that did not (the reference warns and carries on;
    // a description of which NOTHING exists is its "Entire scene not found", ExampleScenes.cpp:76-85 -- thrown here too).
    std::vector<std::filesystem::path> Build(SceneBuilder &builder) const;
};

// ExampleScenes.cpp:113-118: the slot remap of the NVIDIA ORCA assets (Sun Temple, Bistro, Emerald Square, Zero Day):
// roughness and metalness both read the texture assimp files under "specular"
MetallicRoughnessTextureMapping NVIDIAOrcaTextureMapping();

}
