// ExampleScenes.h -- scene registry (mirror of Path-Tracing/ExampleScenes.h plus the
// procedural stand-ins of SURVEY.md 8d).
#pragma once

#include <memory>
#include <string>

#include "Scene.h"

namespace PathTracing::ExampleScenes
{

void CreateDefaultScene(SceneBuilder &sceneBuilder);             // ExampleScenes.cpp:320-545
void CreateRoughnessTestCubesScene(SceneBuilder &sceneBuilder);  // ExampleScenes.cpp:755-842

// detail in (0, 1] scales tessellation (1 = the triangle counts of SURVEY.md 8d)
void CreateAttenuationBlobScene(SceneBuilder &sceneBuilder, float detail, uint32_t seed);
void CreateChessLikeScene(SceneBuilder &sceneBuilder, float detail, uint32_t seed);
void CreateTempleLikeScene(SceneBuilder &sceneBuilder, float detail, uint32_t seed);
void CreateAtriumLikeScene(SceneBuilder &sceneBuilder, float detail, uint32_t seed);
void CreateStreetLikeScene(SceneBuilder &sceneBuilder, float detail, uint32_t seed);
void CreateTextureTestScene(SceneBuilder &sceneBuilder, uint32_t seed); // sampler test content (row N1)
// every sampleMaterial branch: MetallicRoughness / SpecularGlossiness / Phong, textured and plain, an unknown type, DX normal maps
void CreateMaterialsTestScene(SceneBuilder &sceneBuilder, float detail, uint32_t seed);

const char *GetSceneNames();
std::shared_ptr<Scene> CreateScene(const std::string &name, float detail, uint32_t seed);

}
