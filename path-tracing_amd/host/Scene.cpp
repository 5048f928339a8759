// Scene.cpp -- see Scene.h.  Follows Path-Tracing/Scene.cpp line by line where the
// behaviour matters to the path-tracing pass (deduplication by name, identity-transform
// sharing, running MeshOffset, default light).
#include "Scene.h"

#include <cassert>

namespace PathTracing
{

namespace
{
// Scene.h:352-355
const Shaders::DirectionalLight g_DefaultLight = { { 10.0f, 10.0f, 10.0f }, 0.0f, { -0.4f, -1.0f, -0.2f }, 0.0f };

bool IsIdentity(const PtxTransform &t)
{
    const PtxTransform id = IdentityTransform();
    for (int i = 0; i < 12; i++)
        if (t.m[i] != id.m[i])
            return false;
    return true;
}

Vec3 TransformPoint(const Mat4 &t, Vec3 p)
{
    return Vec3(t.m[0][0] * p.x + t.m[0][1] * p.y + t.m[0][2] * p.z + t.m[0][3],
                t.m[1][0] * p.x + t.m[1][1] * p.y + t.m[1][2] * p.z + t.m[1][3],
                t.m[2][0] * p.x + t.m[2][1] * p.y + t.m[2][2] * p.z + t.m[2][3]);
}
Vec3 TransformVector(const Mat4 &t, Vec3 p)
{
    return Vec3(t.m[0][0] * p.x + t.m[0][1] * p.y + t.m[0][2] * p.z, t.m[1][0] * p.x + t.m[1][1] * p.y + t.m[1][2] * p.z,
                t.m[2][0] * p.x + t.m[2][1] * p.y + t.m[2][2] * p.z);
}
}

// ---------------------------------------------------------------------------
// Scene
// ---------------------------------------------------------------------------

// SceneGraph.cpp:36-60.  Stored matrices are math matrices here, so
// "node.Transform * parent.CurrentTransform" of the transposed glm form becomes
// parent.CurrentTransform * node.Transform.
void Scene::UpdateTransforms()
{
    m_Graph.Nodes[0].CurrentTransform = m_Graph.Nodes[0].Transform;
    for (size_t i = 1; i < m_Graph.Nodes.size(); i++)
    {
        SceneNode &node = m_Graph.Nodes[i];
        const SceneNode &parent = m_Graph.Nodes[node.Parent];
        if (m_Graph.IsRelative[i])
            node.CurrentTransform = parent.CurrentTransform * node.Transform;
        else
            node.CurrentTransform = node.Transform;
    }
}

// One step of an animation clip (behaviour of SceneGraph.cpp:8-34): the clock advances by timeStep * TickPerSecond and loops at
// Duration; a loop sends every track's cursor back to its first key (cursors only move forward); every animated node's local
// transform becomes T * R * S of the three sampled tracks -- glm::transpose(scale(translate(I, p) * mat4(r), s)) in the
// reference's transposed storage is this math matrix.
void Animation::Update(float timeStep, std::span<SceneNode> nodes)
{
    float tick = CurrentTick + timeStep * TickPerSecond;
    const bool looped = tick >= Duration;
    while (tick >= Duration) // (repeated subtraction, not fmod: the float the clock lands on is part of the behaviour)
        tick -= Duration;
    CurrentTick = tick;

    for (AnimationNode &track : Nodes)
    {
        if (looped)
            track.Positions.Index = track.Rotations.Index = track.Scales.Index = 0;
        const Mat4 translation = Translate(Mat4::Identity(), track.Positions.Update(tick));
        const Mat4 rotation = ToMat4(track.Rotations.Update(tick));
        nodes[track.SceneNodeIndex].Transform = Scale(translation * rotation, track.Scales.Update(tick));
    }
}

// Scene.cpp:52-83 with SceneGraph::Update (SceneGraph.cpp:86-92) inlined
bool Scene::Update(float timeStep)
{
    bool updated = GetActiveCamera().OnUpdate(timeStep);
    if (m_Graph.Paused)
        return updated;
    updated |= m_Graph.MovesInstances;

    for (Animation &animation : m_Graph.Animations)
        animation.Update(timeStep, m_Graph.Nodes);
    UpdateTransforms();
    for (auto &instance : m_ModelInstances)
        instance.Transform = m_Graph.Nodes[instance.SceneNodeIndex].CurrentTransform;
    for (size_t i = 0; i < m_Graph.Bones.size(); i++) // "Offset * node" of the transposed glm form = node * Offset
    {
        const Mat4 t = m_Graph.Nodes[m_Graph.Bones[i].SceneNodeIndex].CurrentTransform * m_Graph.Bones[i].Offset;
        std::memcpy(m_Graph.BoneTransforms[i].m, &t.m[0][0], sizeof(float) * 12);
    }
    for (size_t i = 0; i < m_Lights.PointRest.size(); i++)
    {
        const Vec3 p = TransformPoint(m_Graph.Nodes[m_Lights.PointRest[i].SceneNodeIndex].CurrentTransform, m_Lights.PointRest[i].Position);
        m_Lights.Point[i].Position[0] = p.x;
        m_Lights.Point[i].Position[1] = p.y;
        m_Lights.Point[i].Position[2] = p.z;
    }
    const Vec3 d = TransformVector(m_Graph.Nodes[m_Lights.DirectionalRest.SceneNodeIndex].CurrentTransform, m_Lights.DirectionalRest.Direction);
    m_Lights.Directional.Direction[0] = d.x;
    m_Lights.Directional.Direction[1] = d.y;
    m_Lights.Directional.Direction[2] = d.z;
    return updated;
}

Camera &Scene::GetActiveCamera()
{
    if (m_ActiveCameraId == g_InputCameraId)
        return m_InputCamera;
    return m_SceneCameras[m_ActiveCameraId];
}

// Scene.cpp:468-482
void Scene::SetActiveCamera(CameraId id)
{
    if (m_ActiveCameraId == id)
        return;
    Camera *camera = &m_InputCamera;
    if (id != g_InputCameraId)
        camera = &m_SceneCameras[id];
    auto [width, height] = GetActiveCamera().GetExtent();
    if (width && height)
        camera->OnResize(width, height);
    m_ActiveCameraId = id;
}

// Scene.cpp:489-512
uint32_t Scene::GetDefaultTextureIndex(TextureType type)
{
    switch (type)
    {
    case TextureType::Color: return PTX_DEFAULT_COLOR_TEXTURE_INDEX;
    case TextureType::Normal: return PTX_DEFAULT_NORMAL_TEXTURE_INDEX;
    case TextureType::Roughness: return PTX_DEFAULT_ROUGHNESS_TEXTURE_INDEX;
    case TextureType::Metallic: return PTX_DEFAULT_METALLIC_TEXTURE_INDEX;
    case TextureType::Emisive: return PTX_DEFAULT_EMISSIVE_TEXTURE_INDEX;
    case TextureType::Specular: return PTX_DEFAULT_SPECULAR_TEXTURE_INDEX;
    case TextureType::Glossiness: return PTX_DEFAULT_GLOSSINESS_TEXTURE_INDEX;
    case TextureType::Shininess: return PTX_DEFAULT_SHININESS_TEXTURE_INDEX;
    default: throw error("Unsupported Texture type " + std::to_string(static_cast<int>(type)));
    }
}

PtxSceneDesc Scene::GetDesc() const
{
    m_InstanceRecords.resize(m_ModelInstances.size());
    for (size_t i = 0; i < m_ModelInstances.size(); i++)
    {
        m_InstanceRecords[i].ModelIndex = m_ModelInstances[i].ModelIndex;
        // TrivialCopy<glm::mat3x4, vk::TransformMatrixKHR>(instance.Transform), AccelerationStructure.cpp:271
        std::memcpy(m_InstanceRecords[i].Transform.m, &m_ModelInstances[i].Transform.m[0][0], sizeof(float) * 12);
    }
    PtxSceneDesc d;
    std::memset(&d, 0, sizeof(d));
    d.vertices = m_Vertices.data();
    d.vertexCount = m_Vertices.size();
    d.indices = m_Indices.data();
    d.indexCount = m_Indices.size();
    d.transforms = m_Transforms.data();
    d.transformCount = static_cast<uint32_t>(m_Transforms.size());
    d.geometries = m_Geometries.data();
    d.geometryCount = static_cast<uint32_t>(m_Geometries.size());
    d.metallicRoughnessMaterials = m_Materials.MetallicRoughness.data();
    d.metallicRoughnessMaterialCount = static_cast<uint32_t>(m_Materials.MetallicRoughness.size());
    d.specularGlossinessMaterials = m_Materials.SpecularGlossiness.data();
    d.specularGlossinessMaterialCount = static_cast<uint32_t>(m_Materials.SpecularGlossiness.size());
    d.phongMaterials = m_Materials.Phong.data();
    d.phongMaterialCount = static_cast<uint32_t>(m_Materials.Phong.size());
    d.meshes = m_MeshRecords.data();
    d.meshCount = static_cast<uint32_t>(m_MeshRecords.size());
    d.models = m_ModelRanges.data();
    d.modelCount = static_cast<uint32_t>(m_ModelRanges.size());
    d.instances = m_InstanceRecords.data();
    d.instanceCount = static_cast<uint32_t>(m_InstanceRecords.size());
    // TextureUploader::GetImageFormat (TextureUploader.cpp:571-594): colour-like types are sRGB
    auto describe = [](const TextureInfo &t, PtxTextureDesc &rec) {
        const bool isColor = t.Type == TextureType::Color || t.Type == TextureType::Specular || t.Type == TextureType::Emisive ||
                             t.Type == TextureType::Skybox;
        rec.width = t.Width;
        rec.height = t.Height;
        rec.format = t.Format == TextureFormat::RGBAF32 ? PTX_TEXTURE_RGBA32F : (isColor ? PTX_TEXTURE_RGBA8_SRGB : PTX_TEXTURE_RGBA8_UNORM);
        rec.levels = t.Pixels.empty() ? 1u : t.Levels;
        rec.data = t.Pixels.empty() ? nullptr : t.Pixels.data();
        if (t.Pixels.empty()) // no data: a 1x1 white placeholder
        {
            rec.width = rec.height = 1;
            static const uint32_t white = 0xffffffffu;
            rec.format = PTX_TEXTURE_RGBA8_UNORM;
            rec.data = &white;
        }
    };
    m_TextureRecords.resize(m_Textures.size());
    for (size_t i = 0; i < m_Textures.size(); i++)
        describe(m_Textures[i], m_TextureRecords[i]);
    d.textures = m_TextureRecords.data();
    d.textureCount = static_cast<uint32_t>(m_TextureRecords.size());
    // Renderer.cpp:402-412, :679-690: the skybox variant picks the miss-shader flag and its images
    d.skyboxKind = PTX_SKYBOX_CLEAR_COLOR;
    m_SkyboxRecords.clear();
    if (const Skybox2D *sky = std::get_if<Skybox2D>(&m_Skybox))
    {
        m_SkyboxRecords.resize(1);
        describe(sky->Content, m_SkyboxRecords[0]);
        d.skyboxKind = PTX_SKYBOX_2D;
    }
    else if (const SkyboxCube *cube = std::get_if<SkyboxCube>(&m_Skybox))
    {
        const TextureInfo *faces[6] = { &cube->Front, &cube->Back, &cube->Up, &cube->Down, &cube->Left, &cube->Right };
        m_SkyboxRecords.resize(6);
        for (int i = 0; i < 6; i++)
            describe(*faces[i], m_SkyboxRecords[i]);
        d.skyboxKind = PTX_SKYBOX_CUBE;
    }
    d.skybox = m_SkyboxRecords.empty() ? nullptr : m_SkyboxRecords.data();
    d.animatedVertices = m_AnimatedVertices.empty() ? nullptr : m_AnimatedVertices.data();
    d.animatedVertexCount = m_AnimatedVertices.size();
    d.animatedIndices = m_AnimatedIndices.empty() ? nullptr : m_AnimatedIndices.data();
    d.animatedIndexCount = m_AnimatedIndices.size();
    d.dxNormalTextures = m_HasDxNormalTextures ? 1u : 0u;
    d.forceFullTextureSize = m_ForceFullTextureSize ? 1u : 0u;
    d.textureMemoryBudget = m_TextureMemoryBudget;
    return d;
}

// Renderer.cpp:1719-1726 with the offsets of Renderer.h:152-156
PtxLightsUbo Scene::GetLightsUbo() const
{
    PtxLightsUbo ubo;
    std::memset(&ubo, 0, sizeof(ubo));
    ubo.LightCount = static_cast<uint32_t>(m_Lights.Point.size());
    ubo.Directional = m_Lights.Directional;
    for (size_t i = 0; i < m_Lights.Point.size(); i++)
        ubo.Lights[i] = m_Lights.Point[i];
    return ubo;
}

// ---------------------------------------------------------------------------
// SceneBuilder
// ---------------------------------------------------------------------------

SceneBuilder::SceneBuilder()
{
    Reset();
}

// A fresh scene under construction: transform slot 0 is the identity every untransformed mesh shares (Scene.h:312), node 0 is the
// root of the graph (Scene.h:336), the light is the reference's default.
void SceneBuilder::Reset()
{
    m_Scene = std::make_shared<Scene>();
    Scene &s = *m_Scene;
    s.m_Transforms.push_back(IdentityTransform());
    s.m_Graph.Nodes.push_back(SceneNode { RootNodeIndex, Mat4::Identity(), Mat4::Identity() });
    s.m_Graph.IsRelative.push_back(true);
    s.m_Lights.Directional = g_DefaultLight;
    s.m_Lights.DirectionalRest = { RootNodeIndex, Vec3(-0.4f, -1.0f, -0.2f) };
    m_Ids = IdsByName();
    m_PendingInstances.clear();
    m_PendingCameras.clear();
    m_NextMeshRecord = 0;
}

uint32_t SceneBuilder::AddSceneNode(SceneNode &&node)
{
    m_Scene->m_Graph.Nodes.push_back(node);
    m_Scene->m_Graph.IsRelative.push_back(true);
    return static_cast<uint32_t>(m_Scene->m_Graph.Nodes.size() - 1);
}

uint32_t SceneBuilder::AddGeometry(Geometry &&geometry)
{
    m_Scene->m_Geometries.push_back(geometry);
    return static_cast<uint32_t>(m_Scene->m_Geometries.size() - 1);
}

uint32_t SceneBuilder::AddModel(std::span<const MeshInfo> meshInfos)
{
    m_Scene->m_Models.push_back(CreateModel(meshInfos));
    return static_cast<uint32_t>(m_Scene->m_Models.size() - 1);
}

uint32_t SceneBuilder::AddModelInstance(uint32_t modelIndex, uint32_t sceneNodeIndex)
{
    m_PendingInstances.emplace_back(modelIndex, sceneNodeIndex);
    return static_cast<uint32_t>(m_PendingInstances.size() - 1);
}

namespace
{
// One entry per distinct name (the reference deduplicates textures and materials by name, Scene.cpp:125-194): the id the first
// caller was given, or `make()`'s for a new name.
template<typename Make> uint32_t IdForName(std::unordered_map<std::string, uint32_t> &ids, const std::string &name, Make make)
{
    const auto known = ids.find(name);
    if (known != ids.end())
        return known->second;
    const uint32_t id = make();
    ids.emplace(name, id);
    return id;
}
}

uint32_t SceneBuilder::AddTexture(TextureInfo &&texture)
{
    const std::string name = texture.Name;
    return IdForName(m_Ids.Textures, name, [&] {
        std::vector<TextureInfo> &textures = m_Scene->m_Textures;
        assert(textures.size() < Shaders::MaxTextureCount);
        textures.push_back(std::move(texture));
        return Shaders::GetSceneTextureIndex(static_cast<uint32_t>(textures.size() - 1));
    });
}

Shaders::MaterialId SceneBuilder::AddMaterial(std::string name, Shaders::MetallicRoughnessMaterial material)
{
    return IdForName(m_Ids.MetallicRoughness, name, [&] {
        auto &materials = m_Scene->m_Materials.MetallicRoughness;
        assert(material.Ior >= 1.0f); // Scene.cpp:147
        assert(materials.size() < Shaders::MaxMaterialCount);
        materials.push_back(material);
        return Shaders::CreateMaterialId(static_cast<uint32_t>(materials.size() - 1), Shaders::MaterialTypeMetallicRoughness);
    });
}

Shaders::MaterialId SceneBuilder::AddMaterial(std::string name, Shaders::SpecularGlossinessMaterial material)
{
    return IdForName(m_Ids.SpecularGlossiness, name, [&] {
        auto &materials = m_Scene->m_Materials.SpecularGlossiness;
        materials.push_back(material);
        return Shaders::CreateMaterialId(static_cast<uint32_t>(materials.size() - 1), Shaders::MaterialTypeSpecularGlossiness);
    });
}

Shaders::MaterialId SceneBuilder::AddMaterial(std::string name, Shaders::PhongMaterial material)
{
    return IdForName(m_Ids.Phong, name, [&] {
        auto &materials = m_Scene->m_Materials.Phong;
        materials.push_back(material);
        return Shaders::CreateMaterialId(static_cast<uint32_t>(materials.size() - 1), Shaders::MaterialTypePhong);
    });
}

void SceneBuilder::SetAbsoluteTransform(uint32_t sceneNodeIndex)
{
    m_Scene->m_Graph.IsRelative[sceneNodeIndex] = false;
}

// a point light follows its scene node (Scene.cpp:229-234, :73-75): its rest position is kept beside the UBO record
void SceneBuilder::AddLight(Shaders::PointLight &&light, uint32_t sceneNodeIndex)
{
    assert(m_Scene->m_Lights.PointRest.size() < Shaders::MaxLightCount);
    m_Scene->m_Lights.PointRest.push_back({ sceneNodeIndex, Vec3(light.Position[0], light.Position[1], light.Position[2]) });
    m_Scene->m_Lights.Point.push_back(light);
}

void SceneBuilder::SetDirectionalLight(Shaders::DirectionalLight &&light, uint32_t sceneNodeIndex)
{
    m_Scene->m_Lights.DirectionalRest = { sceneNodeIndex, Vec3(light.Direction[0], light.Direction[1], light.Direction[2]) };
    m_Scene->m_Lights.Directional = light;
}

void SceneBuilder::AddCamera(CameraInfo &&camera)
{
    m_PendingCameras.push_back(camera);
}

// Scene.cpp:337-355: meshes without a transform of their own share transform slot 0; MeshOffset is the running index of the
// model's first mesh record, which becomes instanceShaderBindingTableRecordOffset.
Model SceneBuilder::CreateModel(std::span<const MeshInfo> meshInfos)
{
    std::vector<PtxTransform> &transforms = m_Scene->m_Transforms;
    Model model = { {}, m_NextMeshRecord };
    model.Meshes.reserve(meshInfos.size());
    for (const MeshInfo &info : meshInfos)
    {
        uint32_t slot = IdentityTransformIndex;
        if (!IsIdentity(info.Transform))
        {
            slot = static_cast<uint32_t>(transforms.size());
            transforms.push_back(info.Transform);
        }
        model.Meshes.push_back({ info.GeometryIndex, info.MaterialIndex, info.ShaderMaterialType, slot });
    }
    m_NextMeshRecord += static_cast<uint32_t>(meshInfos.size());
    return model;
}

// Finish the scene in place (behaviour of Scene.cpp:267-335) and start a fresh one.
std::shared_ptr<Scene> SceneBuilder::CreateSceneShared(const std::string &name)
{
    std::shared_ptr<Scene> scene = std::move(m_Scene);
    Scene &s = *scene;
    s.m_Name = name;

    // what moves: every node an animation drives, and everything below such a node (nodes are stored in pre-order, so one pass)
    std::vector<bool> moves(s.m_Graph.Nodes.size(), false);
    for (const Animation &animation : s.m_Graph.Animations)
        for (const AnimationNode &track : animation.Nodes)
            moves[track.SceneNodeIndex] = true;
    for (size_t i = 0; i < s.m_Graph.Nodes.size(); i++)
        moves[i] = moves[i] || moves[s.m_Graph.Nodes[i].Parent];
    s.m_Graph.MovesInstances = !s.m_Graph.Bones.empty();
    for (const LightInfo &light : s.m_Lights.PointRest)
        s.m_Graph.MovesInstances = s.m_Graph.MovesInstances || moves[light.SceneNodeIndex];
    for (const auto &[modelIndex, sceneNodeIndex] : m_PendingInstances)
        s.m_Graph.MovesInstances = s.m_Graph.MovesInstances || moves[sceneNodeIndex];

    s.m_Graph.BoneTransforms.assign(s.m_Graph.Bones.size(), IdentityTransform());
    s.m_Graph.HasSkinnedGeometry = false;
    for (const Geometry &g : s.m_Geometries)
        s.m_Graph.HasSkinnedGeometry = s.m_Graph.HasSkinnedGeometry || g.IsAnimated != 0; // Scene.cpp:47-48

    for (const auto &[modelIndex, sceneNodeIndex] : m_PendingInstances)
        s.m_ModelInstances.push_back({ modelIndex, sceneNodeIndex, s.m_Graph.Nodes[sceneNodeIndex].Transform });

    // flattened SBT-record table in model-then-mesh order (Renderer.cpp:378-399; static scenes: geometryIndexMap is the identity,
    // Renderer.cpp:333-350)
    for (const Model &model : s.m_Models)
    {
        s.m_ModelRanges.push_back({ model.MeshOffset, static_cast<uint32_t>(model.Meshes.size()) });
        for (const Mesh &mesh : model.Meshes)
            s.m_MeshRecords.push_back({ mesh.GeometryIndex, mesh.MaterialIndex, mesh.TransformBufferOffset });
    }

    s.UpdateTransforms();
    for (const CameraInfo &info : m_PendingCameras)
        s.m_SceneCameras.emplace_back(info.VerticalFOV, info.NearClip, info.FarClip, info.Position, info.Direction, info.UpDirection,
                                      s.m_Graph.Nodes[info.SceneNodeIndex].CurrentTransform);
    // the application's frame loop runs Scene::Update before the first upload (Application.cpp:328-351): do the same so that the
    // Scene is consumable right away
    s.Update(0.0f);

    Reset();
    return scene;
}

}
