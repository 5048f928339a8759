// SceneDescription.cpp -- see SceneDescription.h
#include "SceneDescription.h"

#include "Json.h"
#include "TextureImporter.h"

namespace PathTracing
{

MetallicRoughnessTextureMapping NVIDIAOrcaTextureMapping()
{
    return { TextureType::Color, TextureType::Normal, TextureType::Specular, TextureType::Specular };
}

SceneDescription SceneDescription::Parse(const std::string &json, const std::filesystem::path &base)
{
    const Json doc = Json::Parse(json);
    auto resolve = [&](const std::string &p) {
        const std::filesystem::path path(p);
        return path.is_absolute() || base.empty() ? path : base / path;
    };
    auto flag = [&](const char *key) { return doc.Has(key) && doc[key].kind == Json::Kind::Bool && doc[key].boolean; };
    SceneDescription d;
    for (size_t i = 0; i < doc["components"].Size(); i++)
        d.Components.push_back(resolve(doc["components"][i].Str()));
    if (doc.Has("skybox"))
        d.Sky = resolve(doc["skybox"].Str());
    if (doc.Has("mapping"))
    {
        const std::string &m = doc["mapping"].Str();
        if (m == "orca")
            d.Mapping = NVIDIAOrcaTextureMapping();
        else if (m != "none" && !m.empty())
            throw error("SceneDescription: unknown texture mapping '" + m + "'");
    }
    d.DxNormalTextures = flag("dxNormalTextures");
    d.FullSizeTextures = flag("forceFullTextureSize");
    return d;
}

std::vector<std::filesystem::path> SceneDescription::Build(SceneBuilder &builder) const
{
    std::vector<std::filesystem::path> missing;
    size_t found = 0;
    for (const std::filesystem::path &file : Components)
    {
        if (!std::filesystem::exists(file))
        {
            missing.push_back(file);
            continue;
        }
        SceneImporter::AddFile(builder, file, Mapping);
        found++;
    }
    if (!Sky.empty())
    {
        if (std::filesystem::exists(Sky))
        {
            builder.SetSkybox(Skybox2D { TextureImporter::GetTextureInfo(Sky, TextureType::Skybox, "Skybox") });
            found++;
        }
        else
            missing.push_back(Sky);
    }
    if (found == 0)
        throw error("Entire scene not found");
    if (DxNormalTextures)
        builder.SetDxNormalTextures();
    if (FullSizeTextures)
        builder.ForceFullTextureSize();
    return missing;
}

}
