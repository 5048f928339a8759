// RendererHip.h -- adapter with the reference's `Renderer` method names
// (Path-Tracing/Renderer/Renderer.h:42-85) on top of the C-ABI in include/ptx.h, so
// host code written against the static Renderer class keeps its call sequence:
//   Init -> UpdateSceneData -> OnResize -> [SetSettings] -> Render ... -> Shutdown.
// Only the path-tracing pass is implemented; post-processing, UI and output saving
// stay with the reference's Vulkan renderer (out of scope).
#pragma once

#include <memory>
#include <vector>

#include "Scene.h"

namespace PathTracing
{

class RendererHip
{
public:
    // Renderer::PathTracingSettings (Renderer.h:61-66)
    struct PathTracingSettings
    {
        uint32_t BounceCount = 4;
        float LensRadius = 0.0f;
        float FocalDistance = 10.0f;
    };

    static void Init(int deviceIndex = 0, void *stream = nullptr);
    static void Shutdown();

    static void UpdateSceneData(const std::shared_ptr<Scene> &scene, bool updated);
    static void OnResize(uint32_t width, uint32_t height);
    static void SetSettings(const PathTracingSettings &settings);
    static void SetSamplesPerFrame(uint32_t samples); // s_RefreshRate.SamplesPerFrame (Renderer.cpp:1615-1657)
    static void SetTileShard(uint32_t rank, uint32_t worldSize, uint32_t tileSize);

    // Renderer::Render (Renderer.cpp:1659-1809): uniform fill + one path-tracing launch.
    static void Render();
    static void ResetAccumulationImage();

    static uint32_t GetTotalSamples();
    static std::vector<float> ReadAccumulationImage(); // RGBA32F running sum
    static PtxRenderer *GetHandle();

private:
    static void Check(int status);
};

}
