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
    m_SceneNodes[0].CurrentTransform = m_SceneNodes[0].Transform;
    for (size_t i = 1; i < m_SceneNodes.size(); i++)
    {
        SceneNode &node = m_SceneNodes[i];
        const SceneNode &parent = m_SceneNodes[node.Parent];
        if (m_IsRelativeTransform[i])
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
    if (m_IsAnimationPaused)
        return updated;
    updated |= m_HasAnimatedInstances;

    for (Animation &animation : m_Animations)
        animation.Update(timeStep, m_SceneNodes);
    UpdateTransforms();
    for (auto &instance : m_ModelInstances)
        instance.Transform = m_SceneNodes[instance.SceneNodeIndex].CurrentTransform;
    for (size_t i = 0; i < m_Bones.size(); i++) // "Offset * node" of the transposed glm form = node * Offset
    {
        const Mat4 t = m_SceneNodes[m_Bones[i].SceneNodeIndex].CurrentTransform * m_Bones[i].Offset;
        std::memcpy(m_BoneTransforms[i].m, &t.m[0][0], sizeof(float) * 12);
    }
    for (size_t i = 0; i < m_LightInfos.size(); i++)
    {
        const Vec3 p = TransformPoint(m_SceneNodes[m_LightInfos[i].SceneNodeIndex].CurrentTransform, m_LightInfos[i].Position);
        m_PointLights[i].Position[0] = p.x;
        m_PointLights[i].Position[1] = p.y;
        m_PointLights[i].Position[2] = p.z;
    }
    const Vec3 d = TransformVector(m_SceneNodes[m_DirectionalLightInfo.SceneNodeIndex].CurrentTransform, m_DirectionalLightInfo.Direction);
    m_DirectionalLight.Direction[0] = d.x;
    m_DirectionalLight.Direction[1] = d.y;
    m_DirectionalLight.Direction[2] = d.z;
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
    d.metallicRoughnessMaterials = m_MetallicRoughnessMaterials.data();
    d.metallicRoughnessMaterialCount = static_cast<uint32_t>(m_MetallicRoughnessMaterials.size());
    d.specularGlossinessMaterials = m_SpecularGlossinessMaterials.data();
    d.specularGlossinessMaterialCount = static_cast<uint32_t>(m_SpecularGlossinessMaterials.size());
    d.phongMaterials = m_PhongMaterials.data();
    d.phongMaterialCount = static_cast<uint32_t>(m_PhongMaterials.size());
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
    ubo.LightCount = static_cast<uint32_t>(m_PointLights.size());
    ubo.Directional = m_DirectionalLight;
    for (size_t i = 0; i < m_PointLights.size(); i++)
        ubo.Lights[i] = m_PointLights[i];
    return ubo;
}

// ---------------------------------------------------------------------------
// SceneBuilder
// ---------------------------------------------------------------------------

SceneBuilder::SceneBuilder()
{
    Reset();
}

void SceneBuilder::Reset()
{
    m_MeshOffset = 0;
    m_Vertices.clear();
    m_Indices.clear();
    m_Transforms = { IdentityTransform() }; // Scene.h:312
    m_Geometries.clear();
    m_MetallicRoughnessMaterials.clear();
    m_MetallicRoughnessMaterialIds.clear();
    m_SpecularGlossinessMaterials.clear();
    m_SpecularGlossinessMaterialIds.clear();
    m_PhongMaterials.clear();
    m_PhongMaterialIds.clear();
    m_Textures.clear();
    m_TextureIndices.clear();
    m_Models.clear();
    m_ModelInstanceInfos.clear();
    m_SceneNodes.clear();
    m_SceneNodes.push_back(SceneNode { RootNodeIndex, Mat4::Identity(), Mat4::Identity() }); // Scene.h:336
    m_IsRelativeTransform.clear();
    m_IsRelativeTransform.push_back(true);
    m_Animations.clear();
    m_AnimatedVertices.clear();
    m_AnimatedIndices.clear();
    m_Bones.clear();
    m_LightInfos.clear();
    m_PointLights.clear();
    m_DirectionalLight = g_DefaultLight;
    m_Skybox = SkyboxClearColor {};
    m_DirectionalLightInfo = { RootNodeIndex, Vec3(-0.4f, -1.0f, -0.2f) };
    m_CameraInfos.clear();
    m_HasDxNormalTextures = false;
    m_ForceFullTextureSize = false;
}

uint32_t SceneBuilder::AddSceneNode(SceneNode &&node)
{
    m_SceneNodes.push_back(node);
    m_IsRelativeTransform.push_back(true);
    return static_cast<uint32_t>(m_SceneNodes.size() - 1);
}

uint32_t SceneBuilder::AddGeometry(Geometry &&geometry)
{
    m_Geometries.push_back(geometry);
    return static_cast<uint32_t>(m_Geometries.size() - 1);
}

uint32_t SceneBuilder::AddModel(std::span<const MeshInfo> meshInfos)
{
    m_Models.push_back(CreateModel(meshInfos));
    return static_cast<uint32_t>(m_Models.size() - 1);
}

uint32_t SceneBuilder::AddModelInstance(uint32_t modelIndex, uint32_t sceneNodeIndex)
{
    m_ModelInstanceInfos.emplace_back(modelIndex, sceneNodeIndex);
    return static_cast<uint32_t>(m_ModelInstanceInfos.size() - 1);
}

// Scene.cpp:125-141: deduplicated by name
uint32_t SceneBuilder::AddTexture(TextureInfo &&texture)
{
    auto it = m_TextureIndices.find(texture.Name);
    if (it != m_TextureIndices.end())
        return it->second;
    assert(m_Textures.size() < Shaders::MaxTextureCount);
    m_Textures.push_back(std::move(texture));
    const uint32_t textureIndex = Shaders::GetSceneTextureIndex(static_cast<uint32_t>(m_Textures.size() - 1));
    m_TextureIndices[m_Textures.back().Name] = textureIndex;
    return textureIndex;
}

// Scene.cpp:143-160
Shaders::MaterialId SceneBuilder::AddMaterial(std::string name, Shaders::MetallicRoughnessMaterial material)
{
    auto it = m_MetallicRoughnessMaterialIds.find(name);
    if (it != m_MetallicRoughnessMaterialIds.end())
        return it->second;
    assert(material.Ior >= 1.0f);
    assert(m_MetallicRoughnessMaterials.size() < Shaders::MaxMaterialCount);
    m_MetallicRoughnessMaterials.push_back(material);
    const Shaders::MaterialId materialId = Shaders::CreateMaterialId(
        static_cast<uint32_t>(m_MetallicRoughnessMaterials.size() - 1), Shaders::MaterialTypeMetallicRoughness);
    m_MetallicRoughnessMaterialIds[std::move(name)] = materialId;
    return materialId;
}

// Scene.cpp:162-177
Shaders::MaterialId SceneBuilder::AddMaterial(std::string name, Shaders::SpecularGlossinessMaterial material)
{
    auto it = m_SpecularGlossinessMaterialIds.find(name);
    if (it != m_SpecularGlossinessMaterialIds.end())
        return it->second;
    m_SpecularGlossinessMaterials.push_back(material);
    const Shaders::MaterialId materialId = Shaders::CreateMaterialId(
        static_cast<uint32_t>(m_SpecularGlossinessMaterials.size() - 1), Shaders::MaterialTypeSpecularGlossiness);
    m_SpecularGlossinessMaterialIds[std::move(name)] = materialId;
    return materialId;
}

// Scene.cpp:179-194
Shaders::MaterialId SceneBuilder::AddMaterial(std::string name, Shaders::PhongMaterial material)
{
    auto it = m_PhongMaterialIds.find(name);
    if (it != m_PhongMaterialIds.end())
        return it->second;
    m_PhongMaterials.push_back(material);
    const Shaders::MaterialId materialId =
        Shaders::CreateMaterialId(static_cast<uint32_t>(m_PhongMaterials.size() - 1), Shaders::MaterialTypePhong);
    m_PhongMaterialIds[std::move(name)] = materialId;
    return materialId;
}

void SceneBuilder::SetAbsoluteTransform(uint32_t sceneNodeIndex)
{
    m_IsRelativeTransform[sceneNodeIndex] = false;
}

// Scene.cpp:229-234
void SceneBuilder::AddLight(Shaders::PointLight &&light, uint32_t sceneNodeIndex)
{
    assert(m_LightInfos.size() < Shaders::MaxLightCount);
    m_LightInfos.push_back({ sceneNodeIndex, Vec3(light.Position[0], light.Position[1], light.Position[2]) });
    m_PointLights.push_back(light);
}

// Scene.cpp:236-240
void SceneBuilder::SetDirectionalLight(Shaders::DirectionalLight &&light, uint32_t sceneNodeIndex)
{
    m_DirectionalLightInfo = { sceneNodeIndex, Vec3(light.Direction[0], light.Direction[1], light.Direction[2]) };
    m_DirectionalLight = light;
}

void SceneBuilder::AddCamera(CameraInfo &&camera)
{
    m_CameraInfos.push_back(camera);
}

// Scene.cpp:337-355: identity mesh transforms share slot 0; MeshOffset is the running
// record index that becomes instanceShaderBindingTableRecordOffset.
Model SceneBuilder::CreateModel(std::span<const MeshInfo> meshInfos)
{
    Model model = { {}, m_MeshOffset };
    for (const MeshInfo &meshInfo : meshInfos)
    {
        const bool isIdentity = IsIdentity(meshInfo.Transform);
        model.Meshes.push_back({ meshInfo.GeometryIndex, meshInfo.MaterialIndex, meshInfo.ShaderMaterialType,
                                 isIdentity ? IdentityTransformIndex : static_cast<uint32_t>(m_Transforms.size()) });
        if (!isIdentity)
            m_Transforms.push_back(meshInfo.Transform);
    }
    m_MeshOffset += static_cast<uint32_t>(meshInfos.size());
    return model;
}

// Scene.cpp:267-335
std::shared_ptr<Scene> SceneBuilder::CreateSceneShared(const std::string &name)
{
    // Scene.cpp:269-288: anything below an animated node moves
    std::vector<bool> isAnimated(m_SceneNodes.size(), false);
    for (const Animation &animation : m_Animations)
        for (const AnimationNode &node : animation.Nodes)
            isAnimated[node.SceneNodeIndex] = true;
    for (size_t i = 0; i < m_SceneNodes.size(); i++)
        if (isAnimated[m_SceneNodes[i].Parent])
            isAnimated[i] = true;
    bool hasAnimatedInstances = !m_Bones.empty();
    for (const LightInfo &light : m_LightInfos)
        hasAnimatedInstances |= isAnimated[light.SceneNodeIndex];
    for (auto [modelIndex, sceneNodeIndex] : m_ModelInstanceInfos)
        hasAnimatedInstances |= isAnimated[sceneNodeIndex];

    auto scene = std::make_shared<Scene>();
    scene->m_Name = name;
    scene->m_HasAnimatedInstances = hasAnimatedInstances;
    scene->m_Vertices = std::move(m_Vertices);
    scene->m_Indices = std::move(m_Indices);
    scene->m_Transforms = std::move(m_Transforms);
    scene->m_Geometries = std::move(m_Geometries);
    scene->m_MetallicRoughnessMaterials = std::move(m_MetallicRoughnessMaterials);
    scene->m_SpecularGlossinessMaterials = std::move(m_SpecularGlossinessMaterials);
    scene->m_PhongMaterials = std::move(m_PhongMaterials);
    scene->m_Textures = std::move(m_Textures);
    scene->m_HasDxNormalTextures = m_HasDxNormalTextures;
    scene->m_ForceFullTextureSize = m_ForceFullTextureSize;
    scene->m_Models = std::move(m_Models);
    scene->m_SceneNodes = std::move(m_SceneNodes);
    scene->m_IsRelativeTransform = std::move(m_IsRelativeTransform);
    scene->m_Animations = std::move(m_Animations);
    scene->m_AnimatedVertices = std::move(m_AnimatedVertices);
    scene->m_AnimatedIndices = std::move(m_AnimatedIndices);
    scene->m_Bones = std::move(m_Bones);
    scene->m_BoneTransforms.assign(scene->m_Bones.size(), IdentityTransform());
    scene->m_HasSkeletalAnimations = false;
    for (const Geometry &g : scene->m_Geometries)
        scene->m_HasSkeletalAnimations |= g.IsAnimated != 0; // Scene.cpp:47-48
    scene->m_LightInfos = std::move(m_LightInfos);
    scene->m_PointLights = std::move(m_PointLights);
    scene->m_DirectionalLightInfo = m_DirectionalLightInfo;
    scene->m_DirectionalLight = m_DirectionalLight;
    scene->m_Skybox = std::move(m_Skybox);

    for (auto [modelIndex, sceneNodeIndex] : m_ModelInstanceInfos)
        scene->m_ModelInstances.push_back({ modelIndex, sceneNodeIndex, scene->m_SceneNodes[sceneNodeIndex].Transform });

    // flattened SBT-record table in model-then-mesh order (Renderer.cpp:378-399; static
    // scenes: geometryIndexMap is the identity, Renderer.cpp:333-350)
    for (const Model &model : scene->m_Models)
    {
        scene->m_ModelRanges.push_back({ model.MeshOffset, static_cast<uint32_t>(model.Meshes.size()) });
        for (const Mesh &mesh : model.Meshes)
            scene->m_MeshRecords.push_back({ mesh.GeometryIndex, mesh.MaterialIndex, mesh.TransformBufferOffset });
    }

    scene->UpdateTransforms();
    for (const auto &info : m_CameraInfos)
        scene->m_SceneCameras.emplace_back(info.VerticalFOV, info.NearClip, info.FarClip, info.Position, info.Direction,
                                           info.UpDirection, scene->m_SceneNodes[info.SceneNodeIndex].CurrentTransform);
    // the application's frame loop runs Scene::Update before the first upload
    // (Application.cpp:328-351): do the same so the Scene is consumable right away
    scene->Update(0.0f);

    Reset();
    return scene;
}

}
