#include "RendererHip.h"

namespace PathTracing
{

namespace
{
PtxRenderer *s_Renderer = nullptr;
std::shared_ptr<Scene> s_Scene; // SceneData::Handle (Renderer.h:187)
RendererHip::PathTracingSettings s_PathTracingSettings;
uint32_t s_SamplesPerFrame = 1;
uint32_t s_TotalSamples = 0;
uint32_t s_Width = 0, s_Height = 0;
}

void RendererHip::Check(int status)
{
    if (status != PTX_OK)
        throw error(std::string("RendererHip: ") + (s_Renderer ? ptx_last_error(s_Renderer) : "no renderer"));
}

void RendererHip::Init(int deviceIndex, void *stream)
{
    PtxDeviceDesc desc = { deviceIndex, PTX_BACKEND_WAVEFRONT, stream, 0u, 0u };
    if (ptx_abi_version() != PTX_ABI_VERSION)
        throw error("RendererHip: libptx_hip.so was built against another ptx.h (ABI " + std::to_string(ptx_abi_version()) + ", expected " +
                    std::to_string(PTX_ABI_VERSION) + ")");
    if (ptx_create(&desc, &s_Renderer) != PTX_OK)
        throw error("RendererHip: ptx_create failed (no HIP device?)");
}

void RendererHip::Shutdown()
{
    ptx_destroy(s_Renderer);
    s_Renderer = nullptr;
    s_Scene.reset();
}

// Renderer.cpp:238-439
void RendererHip::UpdateSceneData(const std::shared_ptr<Scene> &scene, bool updated)
{
    if (s_Scene == scene)
    {
        if (updated)
            ResetAccumulationImage();
        return;
    }
    s_Scene = scene;
    const PtxSceneDesc desc = scene->GetDesc();
    Check(ptx_scene_upload(s_Renderer, &desc));
    Check(ptx_build_accel(s_Renderer));
    ResetAccumulationImage();
}

void RendererHip::OnResize(uint32_t width, uint32_t height)
{
    s_Width = width;
    s_Height = height;
    Check(ptx_resize(s_Renderer, width, height));
    ResetAccumulationImage();
}

void RendererHip::SetSettings(const PathTracingSettings &settings)
{
    s_PathTracingSettings = settings;
    ResetAccumulationImage();
}

void RendererHip::SetSamplesPerFrame(uint32_t samples)
{
    s_SamplesPerFrame = samples ? samples : 1;
}

void RendererHip::SetTileShard(uint32_t rank, uint32_t worldSize, uint32_t tileSize)
{
    const PtxTileShard shard = { rank, worldSize, tileSize };
    Check(ptx_set_tile_shard(s_Renderer, &shard));
}

// Renderer.cpp:801-808
void RendererHip::ResetAccumulationImage()
{
    s_TotalSamples = 0;
    if (s_Renderer && s_Width)
        Check(ptx_reset_accumulation(s_Renderer));
}

// Renderer.cpp:1686-1726
void RendererHip::Render()
{
    Camera &camera = s_Scene->GetActiveCamera();
    camera.OnResize(s_Width, s_Height);
    PtxRaygenUniformData rgenData;
    ToColumnMajor(camera.GetInvViewMatrix(), rgenData.ViewInverse);
    ToColumnMajor(camera.GetInvProjectionMatrix(), rgenData.ProjInverse);
    rgenData.BounceCount = s_PathTracingSettings.BounceCount;
    rgenData.LensRadius = s_PathTracingSettings.LensRadius;
    rgenData.FocalDistance = s_PathTracingSettings.FocalDistance;
    rgenData.SampleCount = s_SamplesPerFrame;
    rgenData.TotalSamples = s_TotalSamples;
    s_TotalSamples += s_SamplesPerFrame;
    const PtxLightsUbo lights = s_Scene->GetLightsUbo();
    Check(ptx_render(s_Renderer, &rgenData, &lights));
}

uint32_t RendererHip::GetTotalSamples()
{
    return s_TotalSamples;
}

static RendererHip::PostProcessSettings s_PostProcessSettings;

void RendererHip::SetPostProcessSettings(const PostProcessSettings &settings)
{
    s_PostProcessSettings = settings;
}

void RendererHip::SaveOutput(const OutputInfo &info)
{
    if (info.Extent.width != s_Width || info.Extent.height != s_Height)
        throw error("SaveOutput: the output extent must equal the render extent");
    const PtxPostProcessingUniformData u = { GetTotalSamples(), s_PostProcessSettings.Exposure, s_PostProcessSettings.BloomThreshold,
                                             s_PostProcessSettings.BloomIntensity };
    Check(ptx_postprocess(s_Renderer, &u, s_PostProcessSettings.Hdr ? PTX_TONE_MAPPING_HDR : PTX_TONE_MAPPING_SDR));
    const uint32_t format = OutputSaver::SelectImageFormat(info.Format);
    std::vector<std::byte> bytes(static_cast<size_t>(s_Width) * s_Height * (format == PTX_OUTPUT_RGBA32F ? 16 : 4));
    Check(ptx_read_output(s_Renderer, format, bytes.data(), bytes.size()));
    if (!OutputSaver::WriteImage(info, bytes))
        throw error("SaveOutput: cannot write " + info.Path.string());
}

std::vector<float> RendererHip::ReadAccumulationImage()
{
    std::vector<float> image(static_cast<size_t>(s_Width) * s_Height * 4);
    Check(ptx_readback(s_Renderer, image.data(), image.size() * sizeof(float)));
    return image;
}

PtxRenderer *RendererHip::GetHandle()
{
    return s_Renderer;
}

}
