// render_scene.cpp -- the backend driven from C++ exactly as the reference drives its Renderer
// (Application::Run, Application.cpp:328-351): scene->Update -> UpdateSceneData -> Render ... and
// then the equivalent of its offline "Render" dialog (UserInterface.cpp:1076-1090): accumulate N
// samples, run the post-process chain (postprocess.comp -> bloom -> composition.comp -> toneMapping.comp) and
// hand the output image to the OutputSaver (PNG / TGA / HDR by file extension).
//
//   g++ -std=c++20 -O2 examples/render_scene.cpp path-tracing_amd/host/{Scene,Camera,ExampleScenes,OutputSaver,TextureImporter,JpegDecoder,SceneImporter,SceneDescription,FbxReader,ObjReader,RendererHip}.cpp \
//       -Ipath-tracing_amd/host -Lpath-tracing_amd -lptx_hip -Wl,-rpath,'$ORIGIN/../path-tracing_amd' -o examples/render_scene
//   examples/render_scene default 640 360 16 4 out.png
#include <cmath>
#include <cstdio>
#include <filesystem>
#include <cstdlib>
#include <string>

#include "ExampleScenes.h"
#include "RendererHip.h"

using namespace PathTracing;

int main(int argc, char **argv)
{
    const std::string name = argc > 1 ? argv[1] : "default";
    const uint32_t width = argc > 2 ? std::atoi(argv[2]) : 640, height = argc > 3 ? std::atoi(argv[3]) : 360;
    const uint32_t spp = argc > 4 ? std::atoi(argv[4]) : 16, bounces = argc > 5 ? std::atoi(argv[5]) : 4;
    const char *out = argc > 6 ? argv[6] : "render.png";
    // the host's job, before the process first uses HIP (INTEGRATION.md "Frames in flight"): one hardware queue per stream
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    try
    {
        std::shared_ptr<Scene> scene = ExampleScenes::CreateScene(name, 0.25f, 0);
        RendererHip::Init(0);
        scene->Update(0.0f);
        RendererHip::UpdateSceneData(scene, true);
        RendererHip::OnResize(width, height);
        RendererHip::PathTracingSettings settings;
        settings.BounceCount = bounces;
        RendererHip::SetSettings(settings);
        for (uint32_t i = 0; i < spp; i++) // one sample per frame, like a Profile/Debug build (Config.h:34-36)
            RendererHip::Render();
        const std::vector<float> acc = RendererHip::ReadAccumulationImage();
        const float inv = 1.0f / static_cast<float>(RendererHip::GetTotalSamples());
        double sum = 0.0;
        for (size_t p = 0; p < static_cast<size_t>(width) * height; p++)
            for (int c = 0; c < 3; c++)
                sum += acc[p * 4 + c] * inv;
        // post-process chain + OutputSaver: the format follows the file extension like UserInterface.cpp:1060-1074
        const std::string ext = std::filesystem::path(out).extension().string();
        const OutputFormat format = ext == ".hdr" ? OutputFormat::Hdr : ext == ".tga" ? OutputFormat::Tga : OutputFormat::Png;
        RendererHip::SaveOutput({ out, { width, height }, 0, format });
        std::printf("scene %s %ux%u %u spp depth %u: mean radiance %.9g -> %s\n", name.c_str(), width, height, spp, bounces,
                    sum / (3.0 * width * height), out);
        RendererHip::Shutdown();
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
