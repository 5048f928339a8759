// host_capi.cpp -- include/ptx_host.h over the C++ host mirror.
#include "../../include/ptx_host.h"

#include <string>

#include "ExampleScenes.h"
#include "OutputSaver.h"
#include "TextureImporter.h"

using namespace PathTracing;

struct PthScene
{
    std::shared_ptr<Scene> scene;
};

static thread_local std::string g_error;

extern "C" {

const char *pth_scene_names(void)
{
    return ExampleScenes::GetSceneNames();
}

const char *pth_last_error(void)
{
    return g_error.c_str();
}

PthScene *pth_scene_create(const char *name, float detail, uint32_t seed)
{
    try
    {
        if (!(detail > 0.0f))
            detail = 1.0f;
        auto *s = new PthScene;
        s->scene = ExampleScenes::CreateScene(name ? name : "default", detail, seed);
        return s;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return nullptr;
    }
}

void pth_scene_destroy(PthScene *s)
{
    delete s;
}

int pth_scene_desc(PthScene *s, PtxSceneDesc *out)
{
    if (!s || !out)
        return PTX_ERROR_INVALID_ARGUMENT;
    *out = s->scene->GetDesc();
    return PTX_OK;
}

int pth_scene_lights(PthScene *s, PtxLightsUbo *out)
{
    if (!s || !out)
        return PTX_ERROR_INVALID_ARGUMENT;
    *out = s->scene->GetLightsUbo();
    return PTX_OK;
}

uint64_t pth_scene_triangle_count(PthScene *s)
{
    if (!s)
        return 0;
    uint64_t n = 0;
    const auto models = s->scene->GetModels();
    const auto geometries = s->scene->GetGeometries();
    for (const auto &instance : s->scene->GetModelInstances())
        for (const auto &mesh : models[instance.ModelIndex].Meshes)
            n += geometries[mesh.GeometryIndex].IndexLength / 3;
    return n;
}

int pth_scene_raygen_uniform(PthScene *s, uint32_t width, uint32_t height, uint32_t bounceCount, float lensRadius,
                             float focalDistance, uint32_t sampleCount, uint32_t totalSamples, PtxRaygenUniformData *out)
{
    if (!s || !out || !width || !height)
        return PTX_ERROR_INVALID_ARGUMENT;
    Camera &camera = s->scene->GetActiveCamera();
    camera.OnResize(width, height);
    ToColumnMajor(camera.GetInvViewMatrix(), out->ViewInverse);
    ToColumnMajor(camera.GetInvProjectionMatrix(), out->ProjInverse);
    out->BounceCount = bounceCount;
    out->LensRadius = lensRadius;
    out->FocalDistance = focalDistance;
    out->SampleCount = sampleCount;
    out->TotalSamples = totalSamples;
    return PTX_OK;
}

int pth_scene_set_active_camera(PthScene *s, int32_t cameraId)
{
    if (!s || cameraId < -1 || cameraId >= static_cast<int32_t>(s->scene->GetSceneCamerasCount()))
        return PTX_ERROR_INVALID_ARGUMENT;
    s->scene->SetActiveCamera(cameraId);
    s->scene->Update(0.0f);
    return PTX_OK;
}

int pth_scene_set_camera_pose(PthScene *s, const float position[3], const float direction[3])
{
    if (!s || !position || !direction)
        return PTX_ERROR_INVALID_ARGUMENT;
    s->scene->SetActiveCamera(Scene::g_InputCameraId);
    s->scene->GetActiveCamera().SetPose(Vec3(position[0], position[1], position[2]), Vec3(direction[0], direction[1], direction[2]));
    return PTX_OK;
}

int pth_scene_update(PthScene *s, float timeStep)
{
    if (!s)
        return -1;
    return s->scene->Update(timeStep) ? 1 : 0;
}

uint32_t pth_scene_bone_count(PthScene *s)
{
    return s ? static_cast<uint32_t>(s->scene->GetBoneTransforms().size()) : 0;
}

int pth_scene_animation_state(PthScene *s, PtxTransform *instanceTransforms, uint32_t instanceCount, PtxTransform *boneTransforms, uint32_t boneCount)
{
    if (!s)
        return PTX_ERROR_INVALID_ARGUMENT;
    const auto instances = s->scene->GetModelInstances();
    const auto bones = s->scene->GetBoneTransforms();
    if ((instanceTransforms && instanceCount != instances.size()) || (boneTransforms && boneCount != bones.size()))
        return PTX_ERROR_INVALID_ARGUMENT;
    if (instanceTransforms)
        for (size_t i = 0; i < instances.size(); i++)
            std::memcpy(instanceTransforms[i].m, &instances[i].Transform.m[0][0], sizeof(float) * 12);
    if (boneTransforms && !bones.empty())
        std::memcpy(boneTransforms, bones.data(), bones.size() * sizeof(PtxTransform));
    return PTX_OK;
}

int pth_decode_image(const void *file, size_t fileBytes, uint32_t info[4], void *pixels, size_t bytes)
{
    if (!file || !info)
        return PTX_ERROR_INVALID_ARGUMENT;
    try
    {
        const DecodedImage img = TextureImporter::Decode(std::span<const uint8_t>(static_cast<const uint8_t *>(file), fileBytes));
        info[0] = img.Width; info[1] = img.Height; info[2] = img.Channels; info[3] = img.IsFloat ? 1u : 0u;
        if (pixels)
        {
            const size_t level0 = static_cast<size_t>(img.Width) * img.Height * (img.IsFloat ? 16 : 4);
            if (bytes != img.Pixels.size() && bytes != level0)
                return PTX_ERROR_INVALID_ARGUMENT;
            std::memcpy(pixels, img.Pixels.data(), bytes);
        }
        return PTX_OK;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return PTX_ERROR_INVALID_ARGUMENT;
    }
}

uint32_t pth_decode_image_levels(const void *file, size_t fileBytes)
{
    if (!file)
        return 0;
    try
    {
        return TextureImporter::Decode(std::span<const uint8_t>(static_cast<const uint8_t *>(file), fileBytes)).Levels;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 0;
    }
}

int pth_write_image(const char *path, uint32_t format, uint32_t width, uint32_t height, const void *data, size_t bytes)
{
    if (!path || !data || !width || !height || format > 3)
        return PTX_ERROR_INVALID_ARGUMENT;
    const OutputInfo info = { path, { width, height }, 0, static_cast<OutputFormat>(format) };
    return OutputSaver::WriteImage(info, std::span<const std::byte>(static_cast<const std::byte *>(data), bytes)) ? PTX_OK : PTX_ERROR_INVALID_ARGUMENT;
}

int pth_save_checkpoint(const char *path, uint32_t width, uint32_t height, uint32_t totalSamples, const float *rgba)
{
    if (!path || !rgba || !width || !height)
        return PTX_ERROR_INVALID_ARGUMENT;
    return SaveCheckpoint(path, width, height, totalSamples, rgba) ? PTX_OK : PTX_ERROR_INVALID_ARGUMENT;
}

int pth_load_checkpoint(const char *path, uint32_t *width, uint32_t *height, uint32_t *totalSamples, float *rgba, size_t bytes)
{
    if (!path || !width || !height || !totalSamples)
        return PTX_ERROR_INVALID_ARGUMENT;
    std::vector<float> data;
    if (!LoadCheckpoint(path, *width, *height, *totalSamples, data))
        return PTX_ERROR_INVALID_ARGUMENT;
    if (rgba) // rgba == NULL: header query only
    {
        if (bytes != data.size() * sizeof(float))
            return PTX_ERROR_INVALID_ARGUMENT;
        std::memcpy(rgba, data.data(), bytes);
    }
    return PTX_OK;
}
}
