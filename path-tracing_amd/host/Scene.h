// Scene.h -- host-side mirror of the reference's Scene / SceneBuilder
// (Path-Tracing/Scene.h:63-361, Scene.cpp) for the part the path-tracing pass consumes.
// Same class and method names, same argument meaning, so scene-construction code
// written against the reference (ExampleScenes.cpp) reads the same here.  Differences:
// no glm (Math.h), textures carry their decoded level-0 texels, and the SceneGraph with its
// keyframe animations (SceneGraph.h / SceneGraph.cpp) lives inside Scene instead of a separate class.
#pragma once

#include <memory>
#include <span>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

#include "../../include/ptx.h"

#include "Camera.h"
#include "Math.h"

namespace PathTracing
{

// Core/Core.h:117-122
class error : public std::runtime_error
{
public:
    explicit error(const std::string &message) : std::runtime_error(message) {}
};

namespace Shaders
{
using Vertex = PtxVertex;
using MetallicRoughnessMaterial = PtxMetallicRoughnessMaterial;
using SpecularGlossinessMaterial = PtxSpecularGlossinessMaterial;
using PhongMaterial = PtxPhongMaterial;
using DirectionalLight = PtxDirectionalLight;
using PointLight = PtxPointLight;
using AnimatedVertex = PtxAnimatedVertex;
using MaterialId = uint32_t;

inline constexpr uint32_t SceneTextureOffset = PTX_SCENE_TEXTURE_OFFSET;
inline constexpr uint32_t MaxTextureCount = PTX_MAX_TEXTURE_COUNT;
inline constexpr uint32_t MaxLightCount = PTX_MAX_LIGHT_COUNT;
inline constexpr uint32_t MaxMaterialCount = 1u << 24;
inline constexpr uint32_t MaterialTypeMetallicRoughness = PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS;
inline constexpr uint32_t MaterialTypeSpecularGlossiness = PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS;
inline constexpr uint32_t MaterialTypePhong = PTX_MATERIAL_TYPE_PHONG;

// ShaderTypes.incl:150-158
inline uint32_t GetSceneTextureIndex(uint32_t textureIndex) { return SceneTextureOffset + textureIndex; }
inline uint32_t CreateMaterialId(uint32_t materialIndex, uint32_t materialType) { return (materialIndex << 8) | materialType; }
}

// Scene.h:22-33
enum class TextureType : uint8_t
{
    Emisive,
    Color,
    Normal,
    Roughness,
    Metallic,
    Specular,
    Glossiness,
    Shininess,
    Skybox,
};

// Scene.h:35-42 (the block-compressed formats arrive with the importer, row N2)
enum class TextureFormat : uint8_t
{
    RGBAU8,
    RGBAF32,
};

// Scene.h:48-59.  The reference keeps a file / memory source and decodes on a loader thread
// (TextureImporter.cpp); here the decoded level-0 texels ride along.  An empty Pixels vector is a
// texture whose data is not available: it samples as the white placeholder, which is what the
// reference shows until a texture has finished loading (Renderer.cpp:421-429).
struct TextureInfo
{
    TextureType Type;
    uint32_t Width = 1, Height = 1;
    std::string Name;
    TextureFormat Format = TextureFormat::RGBAU8;
    std::vector<uint8_t> Pixels; // RGBAU8: 4 bytes per texel; RGBAF32: 16 bytes per texel
    uint32_t Levels = 1;         // mip levels in Pixels (a DDS file's own chain), level 0 first, tightly packed
};

using Geometry = PtxGeometry; // Scene.h:63-71

// Scene.h:73-78
enum class MaterialType : uint8_t
{
    MetallicRoughness,
    SpecularGlossiness,
    Phong,
};

// Scene.h:80-86; Transform = glm::mat3x4 = 3 rows of the affine matrix
struct MeshInfo
{
    uint32_t GeometryIndex;
    uint32_t MaterialIndex; // a MaterialId as returned by AddMaterial
    MaterialType ShaderMaterialType;
    PtxTransform Transform;
};

// Scene.h:88-100
struct Mesh
{
    uint32_t GeometryIndex;
    uint32_t MaterialIndex;
    MaterialType ShaderMaterialType;
    uint32_t TransformBufferOffset;
};
struct Model
{
    std::vector<Mesh> Meshes;
    uint32_t MeshOffset;
};

// Scene.h:102-107
struct ModelInstance
{
    uint32_t ModelIndex;
    uint32_t SceneNodeIndex;
    Mat4 Transform;
};

// SceneGraph.h:13-18
struct SceneNode
{
    uint32_t Parent;
    Mat4 Transform;
    Mat4 CurrentTransform;
};

// SceneGraph.h:20-49: per-node TRS keyframe tracks
struct AnimationNode
{
    template<typename T> struct Sequence
    {
        struct Key
        {
            T Value;
            float Tick;
        };

        std::vector<Key> Keys;
        uint32_t Index = 0;

        T Update(float currentTick);
        T Interpolate(float ratio);
    };

    uint32_t SceneNodeIndex;

    Sequence<Vec3> Positions;
    Sequence<Quat> Rotations;
    Sequence<Vec3> Scales;
};

// Keyframe evaluation (behaviour of SceneGraph.h:51-76, written for this mirror; tests/test_animation.py pins it).
// A track is sampled at non-decreasing ticks between two wraps of its animation, so `Index` is a cursor that only moves forward:
// it rests on the last key whose SUCCESSOR is not strictly behind the tick (a tick exactly on a key still blends the segment that
// ends there, with ratio 1).  Before the first key and from the last key on the track is constant.
inline Vec3 BlendKeys(const Vec3 &from, const Vec3 &to, float ratio) { return Mix(from, to, ratio); }
inline Quat BlendKeys(const Quat &from, const Quat &to, float ratio) { return Slerp(from, to, ratio); } // rotations: along the arc

template<typename T> inline T AnimationNode::Sequence<T>::Interpolate(float ratio)
{
    return BlendKeys(Keys[Index].Value, Keys[Index + 1].Value, ratio);
}

template<typename T> inline T AnimationNode::Sequence<T>::Update(float currentTick)
{
    const Key &first = Keys.front();
    if (currentTick < first.Tick)
        return first.Value;
    const uint32_t lastKey = static_cast<uint32_t>(Keys.size()) - 1u;
    for (; Index < lastKey && Keys[Index + 1].Tick < currentTick; ++Index)
    {
    }
    if (Index == lastKey)
        return Keys[lastKey].Value;
    const Key &from = Keys[Index], &to = Keys[Index + 1];
    return Interpolate((currentTick - from.Tick) / (to.Tick - from.Tick));
}

// SceneGraph.h:78-86
struct Animation
{
    std::vector<AnimationNode> Nodes;
    float TickPerSecond;
    float Duration;
    float CurrentTick = 0;

    void Update(float timeStep, std::span<SceneNode> nodes);
};

// Scene.h:109-113
struct Bone
{
    uint32_t SceneNodeIndex;
    Mat4 Offset;
};

struct LightInfo
{
    uint32_t SceneNodeIndex;
    Vec3 Position;
};
struct DirectionalLightInfo
{
    uint32_t SceneNodeIndex;
    Vec3 Direction;
};

// Scene.h:146-155
struct CameraInfo
{
    float VerticalFOV;
    float NearClip;
    float FarClip;
    Vec3 Position;
    Vec3 Direction;
    Vec3 UpDirection;
    uint32_t SceneNodeIndex;
};

// Scene.h:127-157
struct SkyboxClearColor
{
};

struct Skybox2D
{
    TextureInfo Content;
};

struct SkyboxCube
{
    TextureInfo Front;
    TextureInfo Back;
    TextureInfo Up;
    TextureInfo Down;
    TextureInfo Left;
    TextureInfo Right;
};

using SkyboxVariant = std::variant<SkyboxClearColor, Skybox2D, SkyboxCube>;
using CameraId = int32_t;

inline PtxTransform IdentityTransform()
{
    PtxTransform t = { { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 } };
    return t;
}

class Scene
{
public:
    Scene() = default;

    // Scene::Update (Scene.cpp:52-83): advances the scene graph (static here) and
    // refreshes instance transforms and light positions from it.
    bool Update(float timeStep);

    [[nodiscard]] const std::string &GetName() const { return m_Name; }

    [[nodiscard]] std::span<const Shaders::Vertex> GetVertices() const { return m_Vertices; }
    [[nodiscard]] std::span<const uint32_t> GetIndices() const { return m_Indices; }
    [[nodiscard]] std::span<const Shaders::AnimatedVertex> GetAnimatedVertices() const { return m_AnimatedVertices; }
    [[nodiscard]] std::span<const uint32_t> GetAnimatedIndices() const { return m_AnimatedIndices; }
    [[nodiscard]] std::span<const PtxTransform> GetBoneTransforms() const { return m_Graph.BoneTransforms; } // glm::mat3x4 each
    [[nodiscard]] bool HasAnimations() const { return !m_Graph.Animations.empty(); }
    [[nodiscard]] bool HasSkeletalAnimations() const { return m_Graph.HasSkinnedGeometry; }
    void SetAnimationPaused(bool paused) { m_Graph.Paused = paused; }
    [[nodiscard]] std::span<const PtxTransform> GetTransforms() const { return m_Transforms; }
    [[nodiscard]] std::span<const Geometry> GetGeometries() const { return m_Geometries; }
    [[nodiscard]] std::span<const Shaders::MetallicRoughnessMaterial> GetMetallicRoughnessMaterials() const { return m_Materials.MetallicRoughness; }
    [[nodiscard]] std::span<const Shaders::SpecularGlossinessMaterial> GetSpecularGlossinessMaterials() const { return m_Materials.SpecularGlossiness; }
    [[nodiscard]] std::span<const Shaders::PhongMaterial> GetPhongMaterials() const { return m_Materials.Phong; }
    [[nodiscard]] std::span<const TextureInfo> GetTextures() const { return m_Textures; }
    [[nodiscard]] std::span<const Model> GetModels() const { return m_Models; }
    [[nodiscard]] std::span<const ModelInstance> GetModelInstances() const { return m_ModelInstances; }
    [[nodiscard]] bool HasDxNormalTextures() const { return m_HasDxNormalTextures; }
    [[nodiscard]] bool GetForceFullTextureSize() const { return m_ForceFullTextureSize; } // Scene.h: TextureUploader's downscale off
    // Application::GetConfig().MaxTextureMemoryBudget* (TextureUploader.cpp:29-37) as a property of the scene handed to the
    // backend: 0 = the reference's default (min(80 % of the device memory, 1 GiB)), ~0 = no limit
    void SetTextureMemoryBudget(uint64_t bytes) { m_TextureMemoryBudget = bytes; }
    [[nodiscard]] uint64_t GetTextureMemoryBudget() const { return m_TextureMemoryBudget; }
    [[nodiscard]] std::span<const Shaders::PointLight> GetPointLights() const { return m_Lights.Point; }
    [[nodiscard]] const Shaders::DirectionalLight &GetDirectionalLight() const { return m_Lights.Directional; }
    [[nodiscard]] const SkyboxVariant &GetSkybox() const { return m_Skybox; }

    [[nodiscard]] uint32_t GetSceneCamerasCount() const { return static_cast<uint32_t>(m_SceneCameras.size()); }
    [[nodiscard]] CameraId GetActiveCameraId() const { return m_ActiveCameraId; }
    [[nodiscard]] Camera &GetActiveCamera();
    void SetActiveCamera(CameraId id);

    inline static const CameraId g_InputCameraId = -1;

    [[nodiscard]] static uint32_t GetDefaultTextureIndex(TextureType type);

    // What Renderer::UpdateSceneData pulls through the getters above, as the POD the
    // C-ABI takes (include/ptx.h PtxSceneDesc).  Pointers stay valid while the Scene lives.
    [[nodiscard]] PtxSceneDesc GetDesc() const;
    // Lights UBO image as uploaded every frame (Renderer.cpp:1719-1726).
    [[nodiscard]] PtxLightsUbo GetLightsUbo() const;

private:
    friend class SceneBuilder;

    std::string m_Name;
    // geometry pools and what indexes them
    std::vector<Shaders::Vertex> m_Vertices;
    std::vector<uint32_t> m_Indices;
    std::vector<Shaders::AnimatedVertex> m_AnimatedVertices;
    std::vector<uint32_t> m_AnimatedIndices;
    std::vector<PtxTransform> m_Transforms;
    std::vector<Geometry> m_Geometries;
    std::vector<Model> m_Models;
    std::vector<ModelInstance> m_ModelInstances;
    struct MaterialTables // one table per shading model; a MaterialId is (index << 8) | type
    {
        std::vector<Shaders::MetallicRoughnessMaterial> MetallicRoughness;
        std::vector<Shaders::SpecularGlossinessMaterial> SpecularGlossiness;
        std::vector<Shaders::PhongMaterial> Phong;
    } m_Materials;
    std::vector<TextureInfo> m_Textures;
    bool m_HasDxNormalTextures = false;
    bool m_ForceFullTextureSize = false;
    uint64_t m_TextureMemoryBudget = 0;
    // The scene graph (the reference's SceneGraph class, SceneGraph.h:88-110, as a part of the scene): nodes in pre-order, which
    // of them compose with their parent, the clips that drive them, and the skeleton's bones with the matrices skinning.comp reads.
    struct Graph
    {
        std::vector<SceneNode> Nodes;
        std::vector<bool> IsRelative;
        std::vector<Animation> Animations;
        std::vector<Bone> Bones;
        std::vector<PtxTransform> BoneTransforms;
        bool MovesInstances = false;     // an instance, light or bone hangs below an animated node: Update() reports a change
        bool HasSkinnedGeometry = false; // some geometry is skinned (Scene.cpp:47-48)
        bool Paused = false;
    } m_Graph;
    // lights as the UBO holds them, and where each rests in its node's frame (they follow their nodes, Scene.cpp:73-80)
    struct Lights
    {
        std::vector<LightInfo> PointRest;
        std::vector<Shaders::PointLight> Point;
        DirectionalLightInfo DirectionalRest;
        Shaders::DirectionalLight Directional;
    } m_Lights;
    SkyboxVariant m_Skybox = SkyboxClearColor {};
    mutable std::vector<PtxTextureDesc> m_SkyboxRecords;

    // flattened views handed to the C-ABI
    std::vector<PtxMeshRecord> m_MeshRecords;
    std::vector<PtxModel> m_ModelRanges;
    mutable std::vector<PtxModelInstance> m_InstanceRecords;
    mutable std::vector<PtxTextureDesc> m_TextureRecords;

    // Scene.h:259-260
    InputCamera m_InputCamera = InputCamera(45.0f, 100.0f, 0.1f, Vec3(3.0f, 1.0f, 0.0f), Vec3(-1.0f, 0.0f, 0.0f));
    std::vector<AnimatedCamera> m_SceneCameras;
    CameraId m_ActiveCameraId = g_InputCameraId;

    void UpdateTransforms(); // SceneGraph.cpp:36-60
};

class SceneBuilder
{
public:
    /* SceneNodes have to be added in pre-order sequence */
    uint32_t AddSceneNode(SceneNode &&node);

    uint32_t AddGeometry(Geometry &&geometry);
    uint32_t AddModel(std::span<const MeshInfo> meshInfos);
    uint32_t AddModelInstance(uint32_t modelIndex, uint32_t sceneNodeIndex);

    uint32_t AddTexture(TextureInfo &&texture);
    Shaders::MaterialId AddMaterial(std::string name, Shaders::MetallicRoughnessMaterial material);
    Shaders::MaterialId AddMaterial(std::string name, Shaders::SpecularGlossinessMaterial material);
    Shaders::MaterialId AddMaterial(std::string name, Shaders::PhongMaterial material);

    std::vector<Shaders::Vertex> &GetVertices() { return m_Scene->m_Vertices; }
    std::vector<uint32_t> &GetIndices() { return m_Scene->m_Indices; }
    std::vector<Shaders::AnimatedVertex> &GetAnimatedVertices() { return m_Scene->m_AnimatedVertices; }
    std::vector<uint32_t> &GetAnimatedIndices() { return m_Scene->m_AnimatedIndices; }

    void AddAnimation(Animation &&animation) { m_Scene->m_Graph.Animations.push_back(std::move(animation)); }
    uint32_t AddBone(Bone &&bone)
    {
        m_Scene->m_Graph.Bones.emplace_back(std::move(bone));
        return static_cast<uint32_t>(m_Scene->m_Graph.Bones.size() - 1);
    }

    void SetAbsoluteTransform(uint32_t sceneNodeIndex);

    void AddLight(Shaders::PointLight &&light, uint32_t sceneNodeIndex);
    void SetDirectionalLight(Shaders::DirectionalLight &&light, uint32_t sceneNodeIndex);

    void SetSkybox(Skybox2D &&skybox) { m_Scene->m_Skybox = std::move(skybox); }
    void SetSkybox(SkyboxCube &&skybox) { m_Scene->m_Skybox = std::move(skybox); }

    void AddCamera(CameraInfo &&camera);

    void SetDxNormalTextures() { m_Scene->m_HasDxNormalTextures = true; }
    void ForceFullTextureSize() { m_Scene->m_ForceFullTextureSize = true; }
    [[nodiscard]] std::shared_ptr<Scene> CreateSceneShared(const std::string &name);

public:
    static inline constexpr uint32_t IdentityTransformIndex = 0;
    static inline constexpr uint32_t RootNodeIndex = 0;

    SceneBuilder();

private:
    // The builder fills the Scene it will hand out IN PLACE (it is a friend of Scene): geometry, materials, textures, the scene
    // graph, lights and the skybox go straight into `m_Scene`'s members, so nothing is copied or cleared member by member when the
    // scene is finished.  What is the builder's own is only what a finished Scene has no use for: the name -> id tables that
    // deduplicate textures and materials, the (model, node) pairs and camera descriptions that become instances and cameras once
    // the graph's transforms are known, and the running mesh-record offset.
    std::shared_ptr<Scene> m_Scene;
    struct IdsByName
    {
        std::unordered_map<std::string, uint32_t> Textures, MetallicRoughness, SpecularGlossiness, Phong;
    } m_Ids;
    std::vector<std::pair<uint32_t, uint32_t>> m_PendingInstances; // (model index, scene node index)
    std::vector<CameraInfo> m_PendingCameras;
    uint32_t m_NextMeshRecord = 0;

    void Reset();
    Model CreateModel(std::span<const MeshInfo> meshInfos);
};

}
