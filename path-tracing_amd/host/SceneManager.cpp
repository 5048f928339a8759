// SceneManager.cpp -- see SceneManager.h
#include "SceneManager.h"

#include "Json.h"
#include "TextureImporter.h"

namespace PathTracing
{

void CombinedSceneLoader::AddTextureMapping(TextureMapping mapping)
{
    m_TextureMapping = mapping;
}

void CombinedSceneLoader::AddComponent(const std::filesystem::path &path)
{
    m_ComponentPaths.push_back(path);
}

void CombinedSceneLoader::AddComponents(std::span<const std::filesystem::path> paths)
{
    for (const auto &path : paths)
        m_ComponentPaths.push_back(path);
}

void CombinedSceneLoader::AddSkybox2D(const std::filesystem::path &path)
{
    m_SkyboxPath = path;
}

void CombinedSceneLoader::SetDxNormalTextures()
{
    m_HasDxNormalTextures = true;
}

void CombinedSceneLoader::ForceFullTextureSize()
{
    m_ForceFullTextureSize = true;
}

bool CombinedSceneLoader::HasContent() const
{
    return m_SkyboxPath.has_value() || !m_ComponentPaths.empty();
}

// SceneManager.cpp:47-64
void CombinedSceneLoader::Load(SceneBuilder &sceneBuilder)
{
    for (const auto &path : m_ComponentPaths)
        SceneImporter::AddFile(sceneBuilder, path, m_TextureMapping);

    if (m_SkyboxPath.has_value())
    {
        TextureInfo info = TextureImporter::GetTextureInfo(m_SkyboxPath.value(), TextureType::Skybox, "Skybox");
        sceneBuilder.SetSkybox(Skybox2D { std::move(info) });
    }

    if (m_HasDxNormalTextures)
        sceneBuilder.SetDxNormalTextures();

    if (m_ForceFullTextureSize)
        sceneBuilder.ForceFullTextureSize();
}

// SceneManager.cpp:66-94; the reference logs what it drops, here the caller can ask HasContent()
std::unique_ptr<CombinedSceneLoader> SceneDescription::ToLoader() const
{
    auto loader = std::make_unique<CombinedSceneLoader>();
    loader->AddTextureMapping(Mapping);

    for (const auto &path : ComponentPaths)
        if (std::filesystem::exists(path))
            loader->AddComponent(path);

    if (SkyboxPath.has_value() && std::filesystem::exists(SkyboxPath.value()))
        loader->AddSkybox2D(SkyboxPath.value());

    if (HasDxNormalTextures)
        loader->SetDxNormalTextures();

    if (ForceFullTextureSize)
        loader->ForceFullTextureSize();

    return loader;
}

MetallicRoughnessTextureMapping NVIDIAOrcaTextureMapping()
{
    return { TextureType::Color, TextureType::Normal, TextureType::Specular, TextureType::Specular };
}

SceneDescription SceneDescription::FromJson(const std::string &text, const std::filesystem::path &base)
{
    const Json doc = Json::Parse(text);
    auto resolve = [&](const std::string &p) {
        const std::filesystem::path path(p);
        return path.is_absolute() || base.empty() ? path : base / path;
    };
    SceneDescription d;
    for (size_t i = 0; i < doc["components"].Size(); i++)
        d.ComponentPaths.push_back(resolve(doc["components"][i].Str()));
    if (doc.Has("skybox"))
        d.SkyboxPath = resolve(doc["skybox"].Str());
    if (doc.Has("mapping"))
    {
        const std::string &m = doc["mapping"].Str();
        if (m == "orca")
            d.Mapping = NVIDIAOrcaTextureMapping();
        else if (m != "none" && !m.empty())
            throw error("SceneDescription: unknown texture mapping '" + m + "'");
    }
    auto flag = [&](const char *key) { return doc.Has(key) && doc[key].kind == Json::Kind::Bool && doc[key].boolean; };
    d.HasDxNormalTextures = flag("dxNormalTextures");
    d.ForceFullTextureSize = flag("forceFullTextureSize");
    return d;
}

}
