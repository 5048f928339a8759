/*
 * ptx_host.h -- small C-ABI over the C++ host mirror (path-tracing_amd/host/: Scene,
 * SceneBuilder, Camera, ExampleScenes) so that tests, bench.py and non-C++ callers can
 * obtain the same PtxSceneDesc / RaygenUniformData / lights UBO the reference's host
 * side would hand to its Renderer (Scene.h:182-207, Renderer.cpp:1686-1726).
 * No GPU is needed for anything here.
 */
#ifndef PTX_HOST_H
#define PTX_HOST_H

#include "ptx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PthScene PthScene;

/* comma-separated registry: default, roughness_cubes, attenuation_blob, chess_like, ... */
PTX_API const char *pth_scene_names(void);
/* detail in (0,1] scales tessellation of the procedural stand-ins; seed 0 = scene default.
 * Besides the registry names:
 *   "file:<path>"            one glTF 2.0 / GLB / FBX / OBJ asset through SceneImporter::AddFile (the registry's
 *                            one-component file scenes, ExampleScenes.cpp:41-66)
 *   "description:<json>"     a SceneDescription (host/SceneDescription.h; the aggregates of ExampleScenes.cpp:87-236):
 *   "description:@<file>"    {"components": [...], "skybox": "x.hdr", "mapping": "orca" | "none",
 *                            "dxNormalTextures": bool, "forceFullTextureSize": bool}; components and skybox that do not
 *                            exist are dropped (the reference warns and carries on, SceneManager.cpp:66-94), none left = error */
PTX_API PthScene *pth_scene_create(const char *name, float detail, uint32_t seed);
PTX_API void pth_scene_destroy(PthScene *s);
PTX_API const char *pth_last_error(void);

/* Scene getters as one POD; pointers stay valid until pth_scene_destroy */
PTX_API int pth_scene_desc(PthScene *s, PtxSceneDesc *out);
PTX_API int pth_scene_lights(PthScene *s, PtxLightsUbo *out);
PTX_API uint64_t pth_scene_triangle_count(PthScene *s); /* instanced (flattened) count */

/* Camera::OnResize + Renderer::Render's uniform fill (Renderer.cpp:1684-1694) */
PTX_API int pth_scene_raygen_uniform(PthScene *s, uint32_t width, uint32_t height, uint32_t bounceCount,
                                     float lensRadius, float focalDistance, uint32_t sampleCount,
                                     uint32_t totalSamples, PtxRaygenUniformData *out);
/* -1 = the InputCamera (Scene.h:259-260), >= 0 = scene camera */
PTX_API int pth_scene_set_active_camera(PthScene *s, int32_t cameraId);
PTX_API int pth_scene_set_camera_pose(PthScene *s, const float position[3], const float direction[3]);

/* Animation (row N3).  pth_scene_update = Scene::Update(timeStep) (Scene.cpp:52-83): advances the keyframe
 * animations, the scene graph, instance transforms, bone matrices and light positions; returns 1 if the
 * renderer has to refresh (Renderer.cpp:1750-1754), 0 if nothing moved, < 0 on error.  The state to hand to
 * ptx_update_animation is read back with pth_scene_animation_state (either pointer may be NULL). */
PTX_API int pth_scene_update(PthScene *s, float timeStep);
PTX_API uint32_t pth_scene_bone_count(PthScene *s);
PTX_API int pth_scene_animation_state(PthScene *s, PtxTransform *instanceTransforms, uint32_t instanceCount, PtxTransform *boneTransforms,
                                      uint32_t boneCount);

/* TextureImporter (rows N1 / N2): decode an image file held in memory (PNG, baseline JPG, TGA, Radiance HDR, DDS
 * BC1 / BC3 / BC5).  info = { width, height, channels in the file, isFloat }.  pixels may be NULL (query only);
 * otherwise bytes must be width * height * (isFloat ? 16 : 4). */
PTX_API int pth_decode_image(const void *file, size_t fileBytes, uint32_t info[4], void *pixels, size_t bytes);
/* TextureInfo::Levels of the same file: 1, or the number of mip levels a DDS file carries (0 = not decodable).  With more
 * than one level pth_decode_image also takes `bytes` = the size of the whole chain (level 0 first, max(w >> l, 1) x
 * max(h >> l, 1) texels per level, tightly packed) and returns all of it. */
PTX_API uint32_t pth_decode_image_levels(const void *file, size_t fileBytes);

/* Output stage (row N4): OutputSaver::WriteImage (OutputSaver.cpp:227-257) for one image.  format: 0 Png, 1 Jpg,
 * 2 Tga, 3 Hdr.  data: RGBA8 (Png / Jpg / Tga) or RGBA32F (Hdr), top row first. */
PTX_API int pth_write_image(const char *path, uint32_t format, uint32_t width, uint32_t height, const void *data, size_t bytes);
/* checkpoint / resume of the running sum: { magic, width, height, totalSamples } + W*H*4 floats */
PTX_API int pth_save_checkpoint(const char *path, uint32_t width, uint32_t height, uint32_t totalSamples, const float *rgba);
PTX_API int pth_load_checkpoint(const char *path, uint32_t *width, uint32_t *height, uint32_t *totalSamples, float *rgba, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
