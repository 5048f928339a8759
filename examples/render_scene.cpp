// render_scene.cpp -- the backend driven from C++ exactly as the reference drives its Renderer
// (Application::Run, Application.cpp:328-351): scene->Update -> UpdateSceneData -> Render ... and
// then the equivalent of its offline "Render" dialog (UserInterface.cpp:1076-1090): accumulate N
// samples, divide by TotalSamples (postprocess.comp:22), tone map with 1 - exp(-c)
// (toneMapping.comp:13-24) and write the image.
//
//   g++ -std=c++20 -O2 examples/render_scene.cpp path-tracing_amd/host/{Scene,Camera,ExampleScenes,RendererHip}.cpp \
//       -Ipath-tracing_amd/host -Lpath-tracing_amd -lptx_hip -Wl,-rpath,'$ORIGIN/../path-tracing_amd' -o examples/render_scene
//   examples/render_scene default 640 360 16 4 out.ppm
#include <cmath>
#include <cstdio>
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
    const char *out = argc > 6 ? argv[6] : "render.ppm";
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
        FILE *f = std::fopen(out, "wb");
        if (!f)
            throw error(std::string("cannot open ") + out);
        std::fprintf(f, "P6\n%u %u\n255\n", width, height);
        for (size_t p = 0; p < static_cast<size_t>(width) * height; p++)
        {
            unsigned char rgb[3];
            for (int c = 0; c < 3; c++)
            {
                const float linear = acc[p * 4 + c] * inv;
                sum += linear;
                const float mapped = 1.0f - std::exp(-linear);                                     // toneMapping.comp SDR curve
                const float srgb = mapped <= 0.0031308f ? 12.92f * mapped : 1.055f * std::pow(mapped, 1.0f / 2.4f) - 0.055f;
                const float q = srgb < 0.0f ? 0.0f : (srgb > 1.0f ? 1.0f : srgb);
                rgb[c] = static_cast<unsigned char>(q * 255.0f + 0.5f);
            }
            std::fwrite(rgb, 1, 3, f);
        }
        std::fclose(f);
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
