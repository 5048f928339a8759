/*
 * ptx.h -- C-ABI of the MI355X wavefront path-tracing backend.
 *
 * This is the drop-in boundary for ONE hot path of piotrprzybyszdev/Path-Tracing:
 * the vkCmdTraceRaysKHR(W,H,1) pass recorded by Renderer::RecordPathTracingCommands
 * (Path-Tracing/Renderer/Renderer.cpp:892-926) and the scene upload that feeds it
 * (Renderer::UpdateSceneData, Renderer.cpp:238-439).  The reference has no FFI; the
 * seam is the static C++ class `Renderer` (Renderer/Renderer.h:39-85).  Every entry
 * point below names the reference call it stands in for.
 *
 * Rules of the ABI: plain pointers and sizes, no C++ types, no exceptions; every
 * function returns a PtxStatus (0 = OK) and ptx_last_error() gives the message
 * (the reference throws PathTracing::error, Core/Core.h:117-122).  A handle is NOT
 * thread-safe; one host thread per handle, one handle per GPU.
 *
 * All structs in the "data contract" block are byte-compatible with the reference's
 * host/device shared headers (Path-Tracing/Shaders/ShaderTypes.incl and
 * ShaderRendererTypes.incl); sizes are checked by PTX_STATIC_ASSERTs (the
 * equivalent of the reference's PaddingTest.cpp).
 */
#ifndef PTX_H
#define PTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#define PTX_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define PTX_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

#if defined(_WIN32)
#define PTX_API
#else
#define PTX_API __attribute__((visibility("default")))
#endif

/* Bumped whenever a struct of this header changes size or layout or an entry point changes meaning; ptx_abi_version()
 * returns the value the loaded library was built with.  3: PtxSceneDesc.textureMemoryBudget (round 2),
 * PtxStats.hardwareQueues (round 3).  4: PtxStats.treeTriangles / treeReferences (round 4).  5: every `/` of the shader path
 * became a * rcp(b) (round 5: ptx_test_eval and rendered images changed meaning in the last bits) and ptx_unpack_shard_host was
 * added; ptx_unpack_shards, ptx_bind_shard_accumulation; repeat addressing of the sampler takes the exact floor(x) mod n for extents
 * that are not powers of two; normalize() / inversesqrt() go through the specified rsq (PTX_FN_RSQ) instead of rcp(sqrt());
 * PtxDeviceDesc.flags (round 6). */
#define PTX_ABI_VERSION 5u

/* ------------------------------------------------------------------------- */
/* Data contract                                                             */
/* ------------------------------------------------------------------------- */

/* ShaderTypes.incl:18-33 */
enum {
    PTX_DEFAULT_COLOR_TEXTURE_INDEX = 0,
    PTX_DEFAULT_NORMAL_TEXTURE_INDEX = 1,
    PTX_DEFAULT_ROUGHNESS_TEXTURE_INDEX = 2,
    PTX_DEFAULT_METALLIC_TEXTURE_INDEX = 3,
    PTX_DEFAULT_EMISSIVE_TEXTURE_INDEX = 4,
    PTX_DEFAULT_SPECULAR_TEXTURE_INDEX = 5,
    PTX_DEFAULT_GLOSSINESS_TEXTURE_INDEX = 6,
    PTX_DEFAULT_SHININESS_TEXTURE_INDEX = 7,
    PTX_PLACEHOLDER_TEXTURE_INDEX = 8,
    PTX_SCENE_TEXTURE_OFFSET = 9,
    PTX_MAX_TEXTURE_COUNT = 1024,
    PTX_MAX_LIGHT_COUNT = 64
};

/* ShaderTypes.incl:143-145 */
enum {
    PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS = 0,
    PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS = 1,
    PTX_MATERIAL_TYPE_PHONG = 2
};

/* ShaderTypes.incl:41-48 -- 14 floats, unpadded, read by GLSL as vec2[7] (common.glsl:27-46) */
typedef struct PtxVertex {
    float Position[3];
    float TexCoords[2];
    float Normal[3];
    float Tangent[3];
    float Bitangent[3];
} PtxVertex;
PTX_STATIC_ASSERT(sizeof(PtxVertex) == 56, "Vertex is 56 B");

/* ShaderTypes.incl:61-80 */
typedef struct PtxMetallicRoughnessMaterial {
    float EmissiveColor[3];
    float EmissiveIntensity;
    float Color[4];
    float Roughness;
    float Metalness;
    float Ior;
    float Transmission;
    float AttenuationColor[3];
    float AttenuationDistance;
    float pad0, pad1, pad2;
    uint32_t EmissiveIdx;
    uint32_t ColorIdx;
    uint32_t NormalIdx;
    uint32_t RoughnessIdx;
    uint32_t MetallicIdx;
} PtxMetallicRoughnessMaterial;
PTX_STATIC_ASSERT(sizeof(PtxMetallicRoughnessMaterial) == 96, "MR material is 96 B");

/* ShaderTypes.incl:82-99 */
typedef struct PtxSpecularGlossinessMaterial {
    float EmissiveColor[3];
    float EmissiveIntensity;
    float Color[4];
    float Specular[3];
    float Glossiness;
    float AttenuationColor[3];
    float AttenuationDistance;
    float Ior;
    float Transmission;
    uint32_t EmissiveIdx;
    uint32_t ColorIdx;
    uint32_t NormalIdx;
    uint32_t SpecularIdx;
    uint32_t GlossinessIdx;
    float pad0;
} PtxSpecularGlossinessMaterial;
PTX_STATIC_ASSERT(sizeof(PtxSpecularGlossinessMaterial) == 96, "SG material is 96 B");

/* ShaderTypes.incl:101-118 */
typedef struct PtxPhongMaterial {
    float EmissiveColor[3];
    float EmissiveIntensity;
    float Color[4];
    float Specular[3];
    float Shininess;
    float AttenuationColor[3];
    float AttenuationDistance;
    float Ior;
    float Transmission;
    uint32_t EmissiveIdx;
    uint32_t ColorIdx;
    uint32_t NormalIdx;
    uint32_t SpecularIdx;
    uint32_t ShininessIdx;
    float pad0;
} PtxPhongMaterial;
PTX_STATIC_ASSERT(sizeof(PtxPhongMaterial) == 96, "Phong material is 96 B");

/* ShaderTypes.incl:120-126 */
typedef struct PtxDirectionalLight {
    float Color[3];
    float pad0;
    float Direction[3];
    float pad1;
} PtxDirectionalLight;
PTX_STATIC_ASSERT(sizeof(PtxDirectionalLight) == 32, "DirectionalLight is 32 B");

/* ShaderTypes.incl:128-138 */
typedef struct PtxPointLight {
    float Color[3];
    float pad0;
    float Position[3];
    float pad1;
    float AttenuationConstant;
    float AttenuationLinear;
    float AttenuationQuadratic;
    float pad2;
} PtxPointLight;
PTX_STATIC_ASSERT(sizeof(PtxPointLight) == 48, "PointLight is 48 B");

/* closestHit.rchit:32-36 with the offsets of Renderer.h:152-156:
 * uint count @0, DirectionalLight @16, PointLight[64] @48. */
typedef struct PtxLightsUbo {
    uint32_t LightCount;
    uint32_t pad[3];
    PtxDirectionalLight Directional;
    PtxPointLight Lights[PTX_MAX_LIGHT_COUNT];
} PtxLightsUbo;
PTX_STATIC_ASSERT(sizeof(PtxLightsUbo) == 48 + 64 * 48, "lights UBO is 3120 B");

/* ShaderRendererTypes.incl:26-34 (Camera = ShaderTypes.incl:35-39).  Matrices are
 * glm column-major: element [col*4 + row]. */
typedef struct PtxRaygenUniformData {
    float ViewInverse[16];
    float ProjInverse[16];
    uint32_t BounceCount;
    float LensRadius;
    float FocalDistance;
    uint32_t SampleCount;  /* samples added by this launch                      */
    uint32_t TotalSamples; /* samples accumulated BEFORE this launch = RNG frame */
} PtxRaygenUniformData;
PTX_STATIC_ASSERT(sizeof(PtxRaygenUniformData) == 148, "RaygenUniformData is 148 B");

/* Scene.h:63-71 */
typedef struct PtxGeometry {
    uint32_t VertexOffset;
    uint32_t VertexLength;
    uint32_t IndexOffset;
    uint32_t IndexLength;
    uint8_t IsOpaque;
    uint8_t IsAnimated;
    uint8_t pad[2];
} PtxGeometry;
PTX_STATIC_ASSERT(sizeof(PtxGeometry) == 20, "Geometry is 20 B");

/* ShaderRendererTypes.incl:42-47 (SBTBuffer); one per mesh, in model-then-mesh order
 * (Renderer.cpp:378-399).  Record index = Model.MeshOffset + geometry index inside
 * the model's BLAS (AccelerationStructure.cpp:270-274). */
typedef struct PtxMeshRecord {
    uint32_t GeometryIndex;
    uint32_t MaterialId; /* (index << 8) | type, ShaderTypes.incl:155-168 */
    uint32_t TransformIndex;
} PtxMeshRecord;
PTX_STATIC_ASSERT(sizeof(PtxMeshRecord) == 12, "SBTBuffer is 12 B");

/* glm::mat3x4 as used for Scene::GetTransforms() and VkTransformMatrixKHR
 * (closestHit.rchit:12-14): 3 rows of the affine matrix, 4 floats each. */
typedef struct PtxTransform {
    float m[12];
} PtxTransform;
PTX_STATIC_ASSERT(sizeof(PtxTransform) == 48, "mat3x4 is 48 B");

/* Scene.h:96-100 flattened: the meshes of model i are records
 * [MeshOffset, MeshOffset + MeshCount). */
typedef struct PtxModel {
    uint32_t MeshOffset;
    uint32_t MeshCount;
} PtxModel;

/* Scene.h:102-107; Transform = first 3 rows of the instance's affine matrix
 * (AccelerationStructure.cpp:271). */
typedef struct PtxModelInstance {
    uint32_t ModelIndex;
    PtxTransform Transform;
} PtxModelInstance;

/* Scene textures (row N1).  Texture i of the scene has shader index PTX_SCENE_TEXTURE_OFFSET + i
 * (Scene.cpp:125-141).  8-bit data is RGBA8; the image format follows the texture TYPE as in
 * TextureUploader::GetImageFormat (TextureUploader.cpp:571-594): sRGB for Color / Specular /
 * Emissive / Skybox, UNORM otherwise.  The full mip chain (floor(log2(max(w,h))) + 1 levels,
 * Image.cpp:14-17) is generated level by level with a linear 2:1 blit (Image.cpp:264-300). */
typedef enum PtxTextureFormat {
    PTX_TEXTURE_RGBA8_UNORM = 0,
    PTX_TEXTURE_RGBA8_SRGB = 1,
    PTX_TEXTURE_RGBA32F = 2
} PtxTextureFormat;

/* ShaderTypes.incl:50-59 (scalar block layout), the input of skinning.comp */
typedef struct PtxAnimatedVertex {
    float Position[3];
    float TexCoords[2];
    float Normal[3];
    float Tangent[3];
    float Bitangent[3];
    uint32_t BoneIndices[4]; /* MaxBonesPerVertex */
    float BoneWeights[4];
} PtxAnimatedVertex;

typedef struct PtxTextureDesc {
    uint32_t width, height;
    uint32_t format;   /* PtxTextureFormat */
    uint32_t levels;   /* TextureInfo::Levels: mip levels in `data` (0 or 1: level 0 only).  A full chain (floor(log2(max(w, h))) + 1
                        * levels) is uploaded as it is; from any other count only level 0 is used and the chain is generated by
                        * linear blits (TextureUploader.cpp:440-456).  Skybox images: level 0 only */
    const void *data;  /* level 0 first, then max(w >> l, 1) x max(h >> l, 1) texels per level, row-major, tightly packed */
} PtxTextureDesc;

enum {
    PTX_SKYBOX_CLEAR_COLOR = 0, /* miss.rmiss:37: constant (0.08, 0.09, 0.10) */
    PTX_SKYBOX_2D = 1,          /* MissFlagsSkybox2D: miss.rmiss:18-30, skybox[0] is the equirectangular image */
    PTX_SKYBOX_CUBE = 2         /* MissFlagsSkyboxCube: miss.rmiss:31-34, skybox[0..5] = +X -X +Y -Y +Z -Z
                                 * (the layer order of TextureUploader.cpp:234-235: Front Back Up Down Left Right) */
};

/* What Renderer::UpdateSceneData pulls through the Scene getters
 * (Scene.h:182-207).  The caller keeps ownership; ptx_scene_upload copies. */
typedef struct PtxSceneDesc {
    const PtxVertex *vertices;
    uint64_t vertexCount;
    const uint32_t *indices; /* relative to the geometry's VertexOffset */
    uint64_t indexCount;
    const PtxTransform *transforms; /* [0] = identity (Scene.h:306,312) */
    uint32_t transformCount;
    const PtxGeometry *geometries;
    uint32_t geometryCount;
    const PtxMetallicRoughnessMaterial *metallicRoughnessMaterials;
    uint32_t metallicRoughnessMaterialCount;
    const PtxSpecularGlossinessMaterial *specularGlossinessMaterials;
    uint32_t specularGlossinessMaterialCount;
    const PtxPhongMaterial *phongMaterials;
    uint32_t phongMaterialCount;
    const PtxMeshRecord *meshes;
    uint32_t meshCount;
    const PtxModel *models;
    uint32_t modelCount;
    const PtxModelInstance *instances;
    uint32_t instanceCount;
    uint32_t skyboxKind;       /* PTX_SKYBOX_*; PathTracingPipelineConfig.MissFlags */
    uint32_t dxNormalTextures; /* HitFlagsDxNormalTextures, ShaderRendererTypes.incl:96-99 */
    const PtxTextureDesc *textures; /* Scene::GetTextures(); may be NULL: indices >= 9 then sample the white placeholder */
    uint32_t textureCount;
    uint32_t forceFullTextureSize; /* Scene::GetForceFullTextureSize(): TextureUploader::DetermineMaxTextureSizes never halves
                                    * (TextureUploader.cpp:551-569): every texture keeps its size up to 4096 x 4096 */
    const PtxTextureDesc *skybox; /* Scene::GetSkybox(): 1 (2D) or 6 (cube, equal square faces) images, one level each
                                   * (TextureUploader.cpp:203-262); ignored for PTX_SKYBOX_CLEAR_COLOR */
    /* Scene::GetAnimatedVertices() / GetAnimatedIndices(): geometries with IsAnimated index THESE arrays
     * (Renderer.cpp:262-265, :280-312).  Every instanced animated mesh gets its own skinned copy, in bind pose
     * until the first ptx_update_animation. */
    const PtxAnimatedVertex *animatedVertices;
    uint64_t animatedVertexCount;
    const uint32_t *animatedIndices;
    uint64_t animatedIndexCount;
    /* Config MaxTextureMemoryBudgetAbsolute / ...VramPercent (Config.h:63-64,162-163; TextureUploader.cpp:29-37): bytes the scene
     * textures may take together.  Each gets budget / textureCount; a larger one is scaled down by an integer factor on upload
     * (TextureUploader.cpp:409-415,479-501).  0 = the reference's default, min(80 % of the device memory, 1 GiB);
     * ~0 = no limit.  Ignored with forceFullTextureSize.
     * The budget prices the texels in the IMAGE format, as the reference does (4 B per texel for the 8-bit formats); what this
     * library keeps resident is every level decoded to four floats (16 B per texel) plus, for the colour textures of non-opaque
     * geometry, one more float4 per base-level texel (the alpha footprints of the any-hit stages): about 4x the budgeted bytes
     * for 8-bit textures, and about 5x at the peak of the upload, while the encoded pools and the decoded pool coexist.  A host
     * that passes its whole VRAM share here leaves too little for that; the default (at most 1 GiB) does not come near it. */
    uint64_t textureMemoryBudget;
} PtxSceneDesc;

/* ------------------------------------------------------------------------- */
/* Renderer                                                                  */
/* ------------------------------------------------------------------------- */

typedef enum PtxStatus {
    PTX_OK = 0,
    PTX_ERROR_INVALID_ARGUMENT = 1,
    PTX_ERROR_NO_DEVICE = 2,
    PTX_ERROR_OUT_OF_MEMORY = 3,
    PTX_ERROR_DEVICE = 4,
    PTX_ERROR_NOT_READY = 5
} PtxStatus;

typedef enum PtxBackend {
    PTX_BACKEND_WAVEFRONT = 0,  /* queue-per-stage kernels (production path)            */
    PTX_BACKEND_MEGAKERNEL = 1  /* one thread per pixel running raygen.rgen's loop 1:1  */
} PtxBackend;

enum {
    /* ONE stream per handle: the shadow and tail kernels of a frame ride on its main stream instead of an auxiliary one -- no overlap
     * inside a frame, but a frame in flight then takes one hardware queue, not two, and twice as many frames fit the queues.  Pays
     * for THIN frames (a rank's tile shard of an N-GPU job: 16 single-stream frames in flight against 8 two-stream ones, a 1 / 8
     * shard step of chess_like 1.05 -> 0.95 ms, the heavier stand-ins -1 ... -4 %); a whole frame on one GPU does better with two
     * streams (profiles/r06_single_stream*.txt). */
    PTX_DEVICE_SINGLE_STREAM = 1u
};

typedef struct PtxDeviceDesc {
    int32_t deviceIndex;  /* HIP device ordinal                                        */
    uint32_t backend;     /* PtxBackend                                                */
    void *stream;         /* hipStream_t to launch on, or NULL for an internal stream  */
    uint32_t flags;       /* PTX_DEVICE_*                                              */
    uint32_t reserved;    /* 0                                                         */
} PtxDeviceDesc;

/* Pixel-tile shard of a frame: the image is cut into tileSize x tileSize
 * tiles, numbered row-major; this renderer owns tiles with (tile % worldSize) == rank.
 * rank 0 / worldSize 1 = whole frame. */
typedef struct PtxTileShard {
    uint32_t rank;
    uint32_t worldSize;
    uint32_t tileSize;
} PtxTileShard;

/* Counters of the last ptx_render() (all deterministic given the RNG schedule). */
typedef struct PtxStats {
    uint64_t pathSamples;    /* pixel-samples completed (incl. NaN/Inf retries)     */
    uint64_t segments;       /* closest-hit queries traced                          */
    uint64_t shadowRays;     /* occlusion queries traced                            */
    uint64_t retries;        /* NaN/Inf sample restarts (raygen.rgen:99-112)        */
    uint64_t triangles;      /* flattened world-space triangles in the LBVH         */
    uint64_t bvhNodes;       /* nodes of the 4-wide tree (reachable from the root)  */
    double lastRenderMs;     /* device time of the last ptx_render (HIP events)     */
    double lastTraceMs;      /* ... spent in k_trace_closest (HIP events on the stream) */
    double lastBuildMs;      /* device time of the last ptx_build_accel (every candidate tree it built) */
    uint64_t traceLaunches;  /* number of traversal kernel launches in last render  */
    double lastShadeMs;      /* ... spent in k_shade                                */
    double lastShadowMs;     /* ... spent in k_trace_shadow                         */
    double lastTailMs;       /* ... spent in k_tail (fused late bounces)            */
    uint64_t tracedRays;     /* closest-hit queries carried by the k_trace_closest launches timed in lastTraceMs
                                (segments also counts the ones k_tail traces itself) */
    uint64_t hardwareQueues; /* hardware queues the environment grants the process's HIP streams (GPU_MAX_HW_QUEUES when
                                the handle was created, 4 = the runtime's default when unset): below two per handle the
                                frames in flight run one after the other */
    uint64_t treeTriangles;  /* triangles in the tree: `triangles` minus the zero-area ones, which no ray can hit (a full
                                build leaves them out; bvhNodes is no measure of this -- the collapse into 4-wide nodes is
                                driven by the boxes' areas, so fewer triangles can make more nodes) */
    uint64_t treeReferences; /* leaves of the tree: treeTriangles plus the extra references of triangles that were split before
                                the build (a large triangle may hang from several leaves, each with a tighter box) */
} PtxStats;

typedef struct PtxRenderer PtxRenderer;

/* Renderer::Init / Renderer::Shutdown (Renderer.cpp:77-218).  One handle = one frame in flight (the reference's per-frame
 * rendering resources, Renderer.cpp:1454-1460): two HIP streams each.  The HIP runtime maps a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues (4 by default, which serialises frames in flight): a host that keeps several handles
 * exports GPU_MAX_HW_QUEUES=16 before its first HIP call.  The library does not touch the environment; it
 * reports what it finds (PtxStats::hardwareQueues) and the first handle whose streams no longer fit leaves a note in
 * ptx_last_error although ptx_create returned PTX_OK. */
PTX_API int ptx_create(const PtxDeviceDesc *desc, PtxRenderer **out);
PTX_API void ptx_destroy(PtxRenderer *r);
PTX_API const char *ptx_last_error(const PtxRenderer *r);
PTX_API int ptx_device_count(void);
/* PTX_ABI_VERSION of the library that was loaded: the structs of this header carry no size fields, so a binding compares
 * this with the header it was written against before the first call (the Python package and RendererHip do). */
PTX_API uint32_t ptx_abi_version(void);

/* Renderer::UpdateSceneData (Renderer.cpp:238-439): copy the scene to HBM. */
PTX_API int ptx_scene_upload(PtxRenderer *r, const PtxSceneDesc *scene);
/* AccelerationStructure::Build (AccelerationStructure.cpp:26-46; BLAS :64-247, TLAS
 * :250-301), replaced by a software LBVH over the flattened world-space triangles.
 * The reference asks its driver for ePreferFastTrace (AccelerationStructure.cpp:319-324); this build spends time the same way:
 * seven candidate trees (clustering radius, merge metric, Morton cells), each re-optimised by two passes of parallel reinsertion
 * and collapsed to 4-wide nodes by a cost-driven rule, are priced on a sample of path-like rays (node visits + triangle tests), and
 * the cheapest is built once more with 32 reinsertion passes.  Results never depend on the tree; its quality moves frame times by
 * several per cent, and no one setting is best for every scene.  PtxStats::lastBuildMs is the time of all of it (2 M triangles:
 * ~0.8 s, 4 M: ~1.7 s; PTX_REINSERT=0 in the environment: 0.14 / 0.27 s).  PTX_PLOC_RADIUS / PTX_PLOC_SHAPE build one tree with
 * those parameters instead; the per-frame rebuilds of ptx_update_animation use the parameters chosen here with two reinsertion
 * passes. */
PTX_API int ptx_build_accel(PtxRenderer *r);
/* Frames in flight share ONE scene: the reference keeps GetInFlightCount() sets of per-frame rendering resources
 * (Renderer.cpp:1454-1460) over one set of scene buffers and one acceleration structure (s_StaticSceneData / s_SceneData,
 * Renderer.cpp:238-439).  `r` from now on renders the scene and the tree of `owner` (same device, uploaded and built)
 * instead of holding copies; its own scene, if any, is released.  Lights, camera, image, path state stay per renderer.
 * The borrowing ends with ptx_scene_upload on `r`, or when either renderer is destroyed (a borrower whose owner is gone
 * reports PTX_ERROR_NOT_READY).  ptx_build_accel / ptx_update_animation are the owner's calls; they, and a new
 * ptx_scene_upload on the owner, first wait for the borrowers' frames in flight.  Not thread-safe across the two
 * renderers, like the rest of the interface (the reference's Renderer is driven from one thread). */
PTX_API int ptx_share_scene(PtxRenderer *r, PtxRenderer *owner);

/* Renderer::OnResize / CreateSceneRenderingResources: (re)allocate the RGBA32F
 * accumulation image (Renderer.cpp:1284-1287) and the wavefront path state. */
PTX_API int ptx_resize(PtxRenderer *r, uint32_t width, uint32_t height);
PTX_API int ptx_set_tile_shard(PtxRenderer *r, const PtxTileShard *shard);
PTX_API int ptx_set_backend(PtxRenderer *r, uint32_t backend);

/* Renderer::ResetAccumulationImage (Renderer.cpp:801-808, clear at :1734-1748). */
PTX_API int ptx_reset_accumulation(PtxRenderer *r);

/* Renderer::Render's uniform fill + RecordPathTracingCommands (Renderer.cpp:1686-1726,
 * 892-926): adds uniform->SampleCount samples per owned pixel, RNG frame =
 * uniform->TotalSamples, into the accumulation image.  Asynchronous on the stream. */
PTX_API int ptx_render(PtxRenderer *r, const PtxRaygenUniformData *uniform, const PtxLightsUbo *lights);
/* Convenience for the canonical schedule: `frames` launches with
 * SampleCount = 1 and TotalSamples = firstFrame .. firstFrame+frames-1, issued as ONE
 * wavefront batch; the result is bit-identical to `frames` ptx_render calls. */
PTX_API int ptx_render_frames(PtxRenderer *r, const PtxRaygenUniformData *uniform, const PtxLightsUbo *lights,
                              uint32_t firstFrame, uint32_t frames);

PTX_API int ptx_synchronize(PtxRenderer *r);
/* imageLoad of the accumulation image: device -> host, W*H*4 floats (running SUM). */
PTX_API int ptx_readback(PtxRenderer *r, float *rgba, size_t bytes);
/* The same read-back, overlapped with whatever is launched next: _begin snapshots the image on the render stream and
 * starts the copy into `pinnedHost` (page-locked memory, width*height*16 bytes) on the renderer's auxiliary stream -- a
 * one-workgroup copy kernel when the device can address the buffer (hipHostMalloc / hipHostRegister memory; gentle on the
 * PCIe link the other frames' dispatches share), the runtime's copy otherwise; _end waits for it.
 * A second _begin before _end queues behind the first.  (OutputSaver reads its output back a frame late in the same
 * way, OutputSaver.cpp:120-199.) */
PTX_API int ptx_readback_begin(PtxRenderer *r, float *pinnedHost, size_t bytes);
PTX_API int ptx_readback_end(PtxRenderer *r);
/* Device pointer of the accumulation image (for the RCCL gather) and its size. */
PTX_API void *ptx_device_accum_ptr(PtxRenderer *r);
PTX_API size_t ptx_accum_bytes(const PtxRenderer *r);
/* Pack / unpack this shard's tiles to/from a dense tile-major buffer of
 * ptx_shard_bytes() bytes (the message of the single gather). */
PTX_API size_t ptx_shard_bytes(const PtxRenderer *r, uint32_t rank);
PTX_API int ptx_pack_shard(PtxRenderer *r, void *devDst);
PTX_API int ptx_unpack_shard(PtxRenderer *r, uint32_t rank, const void *devSrc);
/* The same, and the shard's pixels also go straight into `pinnedHost` (page-locked, device-addressable, width*height*16 bytes):
 * the rank that owns the gathered frame hands it to the host while it unpacks it -- no snapshot and no second pass over the image,
 * which is what bounds a step once the ranks render faster than the PCIe link reads back.  Asynchronous on
 * the render stream; the buffer is complete once every rank's shard has been unpacked into it and ptx_readback_end() has returned
 * (it waits for the last such call). */
PTX_API int ptx_unpack_shard_host(PtxRenderer *r, uint32_t rank, const void *devSrc, float *pinnedHost, size_t bytes);
/* The whole gathered frame in ONE launch: `devSrc` holds the shards of ranks 0 .. worldSize-1 (ptx_pack_shard's layout each),
 * `strideBytes` apart (a multiple of 16, at least the largest ptx_shard_bytes) -- the receive buffer of the gather as it is.
 * Targets: this renderer's device image (toDeviceImage != 0), the host's page-locked frame (pinnedHost != NULL, width*height*16
 * bytes; complete after ptx_readback_end), or both.  A rank that only hands the frame to the host -- OutputSaver's role
 * (OutputSaver.cpp:120-199) -- passes toDeviceImage = 0 and never rewrites its device image.  One thread per pixel in row-major
 * order: the stores are one contiguous stream over the frame.  Because the RNG is a pure function of (pixel, width, frame)
 * (common.glsl:143-147) ANY rank may own any frame: a job rotates the owner over the ranks with its frames in flight
 * (bench.py), so that no rank carries the read-back of every step. */
PTX_API int ptx_unpack_shards(PtxRenderer *r, const void *devSrc, size_t strideBytes, int toDeviceImage, float *pinnedHost, size_t bytes);
/* Accumulate the samples of this renderer's tile shard IN `devShard` (at least ptx_shard_bytes(r, own rank) bytes of device memory,
 * e.g. the send buffer of the gather) in ptx_pack_shard's layout, instead of in the row-major accumulation image:
 * raygen.rgen:115-117's imageLoad / imageStore go to entry = slot inside the frame.  ptx_reset_accumulation clears it,
 * ptx_pack_shard becomes a no-op (or a device copy if asked for another buffer); the calls that need the row-major frame
 * (ptx_readback*, ptx_postprocess, ptx_write_accumulation) return PTX_ERROR_NOT_READY until NULL is bound again.  ptx_resize and a
 * ptx_set_tile_shard that changes the shard unbind it. */
PTX_API int ptx_bind_shard_accumulation(PtxRenderer *r, void *devShard, size_t bytes);

PTX_API int ptx_get_stats(PtxRenderer *r, PtxStats *stats);

/* ------------------------------------------------------------------------- */
/* Animation (row N3)                                                        */
/* ------------------------------------------------------------------------- */

typedef enum PtxAccelUpdate {
    PTX_ACCEL_REFIT = 0,   /* keep the tree topology of the last full build: new triangle positions, boxes refitted bottom-up
                            * (the reference's BLAS update + TLAS rebuild, AccelerationStructure.cpp:48-57) */
    PTX_ACCEL_REBUILD = 1  /* full LBVH build */
} PtxAccelUpdate;

/* What Renderer::Render does when Scene::Update reported a change (Renderer.cpp:1750-1754, :854-890):
 * new ModelInstance transforms (instanceCount must match the uploaded scene; NULL = unchanged), new bone
 * matrices for skinning.comp (Scene::GetBoneTransforms(): boneCount x mat3x4; NULL = unchanged), the skinning
 * pass over every animated mesh, then the acceleration-structure update.  Results never depend on the tree:
 * a refit and a rebuild render identical images. */
PTX_API int ptx_update_animation(PtxRenderer *r, const PtxTransform *instanceTransforms, uint32_t instanceCount,
                                 const PtxTransform *boneTransforms, uint32_t boneCount, uint32_t accelUpdate);

/* ------------------------------------------------------------------------- */
/* Output stage (row N4): what turns the running sum into a displayable image */
/* ------------------------------------------------------------------------- */

/* ShaderRendererTypes.incl:81-87 (uniform of postprocess.comp / composition.comp) */
typedef struct PtxPostProcessingUniformData {
    uint32_t TotalSamples;
    float Exposure;
    float BloomThreshold;
    float BloomIntensity;
} PtxPostProcessingUniformData;

typedef enum PtxToneMappingMode {
    PTX_TONE_MAPPING_SDR = 0, /* ToneMappingModeSDR: 1 - exp(-c), toneMapping.comp:22 */
    PTX_TONE_MAPPING_HDR = 1  /* ToneMappingModeHDR: pass through                      */
} PtxToneMappingMode;

typedef enum PtxOutputFormat {
    PTX_OUTPUT_RGBA8_SRGB = 0, /* OutputSaver::SelectImageFormat: Png / Jpg / Tga / Mp4 -> eR8G8B8A8Srgb */
    PTX_OUTPUT_RGBA32F = 1     /* Hdr -> eR32G32B32A32Sfloat                                            */
} PtxOutputFormat;

/* Renderer::RecordPostProcessCommands (Renderer.cpp:928-1085) followed by RecordSaveOutputCommands
 * (:1204-1246) on the accumulation image: postprocess.comp (sum / TotalSamples * Exposure, NaN / Inf
 * markers, bloom prefilter) -> bloomDownsample.comp / bloomUpsample.comp over min(levels - 3, 12) mips
 * -> composition.comp -> toneMapping.comp.  Every intermediate image of the reference is rgba16f and so
 * is every intermediate value here.  Images whose larger side is below 16 pixels skip the bloom chain.
 * The result stays on the device until ptx_read_output. */
PTX_API int ptx_postprocess(PtxRenderer *r, const PtxPostProcessingUniformData *uniform, uint32_t toneMappingMode);
/* OutputSaver's blit of the tone-mapped image into its sRGB8 / RGBA32F output image + readback
 * (OutputSaver.cpp:64-86, :120-199): W*H*4 bytes or W*H*16 bytes, row-major, top row first. */
PTX_API int ptx_read_output(PtxRenderer *r, uint32_t outputFormat, void *host, size_t bytes);
/* Restores the accumulation image from a host copy of the running sum (checkpoint / resume). */
PTX_API int ptx_write_accumulation(PtxRenderer *r, const float *rgba, size_t bytes);

/* Use caller-owned device memory (width*height*16 bytes, e.g. a torch tensor that takes
 * part in the RCCL gather) as the accumulation image; NULL returns to the internal one. */
PTX_API int ptx_bind_accumulation(PtxRenderer *r, void *devPtr, size_t bytes);

/* traceRayEXT stand-in on explicit rays, 8 floats each (ox,oy,oz,tmin, dx,dy,dz,tmax):
 * closest hit (anyHit = 0, gl_RayFlagsNoneEXT, raygen.rgen:68) or occlusion (anyHit = 1,
 * TerminateOnFirstHit, raygen.rgen:31).  hits: 4 floats per ray (t, u, v, hit ? 1 : 0);
 * ids: 2 uints per ray ((instance,mesh) pair index, primitive index; 0xffffffff = miss). */
PTX_API int ptx_trace_rays(PtxRenderer *r, const float *rays, uint32_t n, int anyHit, float *hits, uint32_t *ids);

/* ------------------------------------------------------------------------- */
/* Function-level entry (mirrors Path-Tracing-Tests/TestRenderer.cpp:79-105:   */
/* run one production shading function over n packed inputs on the device).    */
/* ------------------------------------------------------------------------- */
typedef enum PtxTestFunction {
    /* ShadingTestShaderTypes.incl:19-27 */
    PTX_FN_GGX_DISTRIBUTION = 0,    /* in: H.xyz, alpha            out: result                  */
    PTX_FN_LAMBDA = 1,              /* in: V.xyz, alpha            out: result                  */
    PTX_FN_GGX_SMITH = 2,           /* in: V.xyz, alpha            out: result                  */
    PTX_FN_DIELECTRIC_FRESNEL = 3,  /* in: VdotH, eta              out: result                  */
    PTX_FN_SCHLICK_FRESNEL = 4,     /* in: VdotH                   out: result                  */
    PTX_FN_EVALUATE_REFLECTION = 5, /* in: V, L, F, alpha (10)     out: result.xyz, pdf         */
    PTX_FN_EVALUATE_REFRACTION = 6, /* in: V, L, F, alpha, eta(11) out: result.xyz, pdf         */
    PTX_FN_SAMPLE_GGX = 7,          /* in: u.xy, V.xyz, alpha      out: result.xyz              */
    /* BsdfTestShaderTypes.incl:13 */
    PTX_FN_SAMPLE_LOBE_PDFS = 8,    /* in: metalness, transmission, F   out: 4 lobe weights     */
    /* additional coverage of bsdf.glsl / common.glsl / sampling.glsl / ray.glsl */
    PTX_FN_EVALUATE_BSDF = 9,       /* in: material(8) V L (14)    out: bsdf.xyz, pdf           */
    PTX_FN_SAMPLE_BSDF = 10,        /* in: material(8) V rng (12)  out: dir.xyz pdf color.xyz rng (8) */
    PTX_FN_RNG = 11,                /* in: px py w frame (as u32)  out: state0, 4 draws (5)     */
    PTX_FN_DISK = 12,               /* in: u.xy                    out: d.xy                    */
    PTX_FN_COS_HEMISPHERE = 13,     /* in: u.xy                    out: d.xyz                   */
    PTX_FN_TANGENT_SPACE = 14,      /* in: n.xyz                   out: 9 (columns T,B,N)       */
    PTX_FN_OFFSET_SELF_INTERSECTION = 15, /* in: origin.xyz normal.xyz  out: p.xyz              */
    PTX_FN_PRIMARY_RAY = 16,        /* in: px py w h u.xy + 32 matrix floats (38)  out: o d rx.o rx.d ry.o ry.d (18) */
    PTX_FN_SINCOS = 17,             /* in: x                       out: sin, cos                */
    PTX_FN_POW = 18,                /* in: x, y                    out: pow(x, y)               */
    PTX_FN_SAMPLE_LIGHT = 19,       /* in: u.xyz pos.xyz count(u32) dirColor dirDir 2x(color pos att) (31)
                                       out: dir.xyz dist color.xyz atten pdf (9)                 */
    PTX_FN_SHADOW_TERMINATOR = 20,  /* in: P, (P,N)x3, bary.xyz, isRefracted (25)  out: origin.xyz */
    PTX_FN_PRIMARY_RAY_LENS = 21,   /* in: px py w h u.xy u2.xy lensRadius focalDistance + 32 (42) out: as 16 (18) */
    /* tracing.glsl (ray differentials -> texture footprint) */
    PTX_FN_DPN_DUV = 22,            /* in: (P,N,uv)x3 (24), vertex T,B (6) = 30   out: dpdu dpdv dndu dndv (12)  */
    PTX_FN_DP_DXY = 23,             /* in: p o d rxO rxD ryO ryD n (24)           out: dpdx dpdy (6)             */
    PTX_FN_DERIVATIVES = 24,        /* in: dpdx dpdy dpdu dpdv (12)               out: dudx dvdx dudy dvdy (4)   */
    PTX_FN_REFLECTED_DIFFERENTIALS = 25, /* in: deriv(4) n p wo wi dndu dndv rxO rxD ryO ryD (34)  out: rxO rxD ryO ryD (12) */
    PTX_FN_REFRACTED_DIFFERENTIALS = 26, /* in: same + eta (35)                   out: rxO rxD ryO ryD (12)      */
    PTX_FN_COMPUTE_LOD = 27,        /* in: derivatives (4)                        out: lod                       */
    PTX_FN_SKYBOX_TEXCOORDS = 28,   /* in: ray direction (3)                      out: uv (2)   miss.rmiss:20-25 */
    PTX_FN_HDR_TO_LDR = 29,         /* in: rgb (3)                                out: rgb (3)  common.glsl:17-20 */
    PTX_FN_ATAN_ASIN = 30,          /* in: y x (2)                                out: atan(y,x) asin(y) kernels */
    PTX_FN_POSTPROCESS_PIXEL = 31,  /* in: acc.rgb TotalSamples(bits) Exposure BloomThreshold (6) out: color bloom (6) postprocess.comp:22-37 */
    PTX_FN_COMPOSITION_PIXEL = 32,  /* in: post.rgb bloom.rgb BloomIntensity (7)    out: rgb (3)  composition.comp:23 */
    PTX_FN_TONEMAP_PIXEL = 33,      /* in: rgb (3)                                  out: rgb (3)  toneMapping.comp:20-22, SDR */
    PTX_FN_SAMPLE_MATERIAL = 34,    /* material.glsl:62-171 on explicit texels.  in (47): type, isHitFromInside, flipNormalY (u32 bits), the 96-byte
                                       material record of that type (24 dwords; its texture indices are ignored), the five textureGrad
                                       results in slot order (emissive, colour, normal, 4th, 5th; rgba each)
                                       out (17): EmissiveColor Color Normal Roughness Metalness Transmission Eta AttenuationColor AttenuationDistance */
    PTX_FN_DIVIDE = 35,             /* in: a, b (2)   out: rcp(b), a / b (2) -- the specified division every `/` of the shader path goes
                                       through: a * rcp(b), rcp correctly rounded on [2^-126, 2^126], +-inf below, +-0 above */
    PTX_FN_SQRT = 36,               /* in: x          out: sqrt(x), correctly rounded                                          */
    PTX_FN_RSQ = 37,                /* in: x          out: rsq(x) -- the specified reciprocal square root of normalize() and inversesqrt():
                                       RN(1 / sqrt(x)) for positive normal x, +-inf for +-0 / denormals, +0 for +inf, NaN otherwise  */
    PTX_FN_COUNT = 38
} PtxTestFunction;

/* Material block used by PTX_FN_EVALUATE_BSDF / PTX_FN_SAMPLE_BSDF:
 * color.rgb, roughness, metalness, transmission, eta (7 floats + 1 pad). */

/* textureGrad (implicitLod = 0; material.glsl:72-76) or texture() at LOD 0 (implicitLod = 1, the
 * any-hit shaders: anyhit.rahit:48) of the uploaded scene's textures.  in: 7 floats per sample
 * (texture index as uint bits, u, v, dudx, dvdx, dudy, dvdy); out: rgba. */
PTX_API int ptx_test_texture(PtxRenderer *r, const float *in, float *out, uint32_t n, int implicitLod);

PTX_API int ptx_test_input_stride(uint32_t fn);
PTX_API int ptx_test_output_stride(uint32_t fn);
PTX_API int ptx_test_eval(PtxRenderer *r, uint32_t fn, const float *in, float *out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* PTX_H */
