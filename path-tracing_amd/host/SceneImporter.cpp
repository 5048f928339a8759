#include "SceneImporter.h"
#include "FbxReader.h"
#include "ObjReader.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>
#include <stack>
#include <tuple>
#include <unordered_map>

#include "Json.h"
#include "TextureImporter.h"

namespace PathTracing
{

namespace
{

// ---------------------------------------------------------------------------------------------------------
// glTF container: JSON + buffers
// ---------------------------------------------------------------------------------------------------------

std::vector<uint8_t> DecodeBase64(std::string_view s)
{
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (char c : s)
    {
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else continue; // padding / whitespace
        acc = (acc << 6) | static_cast<uint32_t>(v);
        bits += 6;
        if (bits >= 8)
        {
            bits -= 8;
            out.push_back(static_cast<uint8_t>((acc >> bits) & 0xff));
        }
    }
    return out;
}

std::string UriDecode(const std::string &uri)
{
    std::string s;
    for (size_t i = 0; i < uri.size(); i++)
        if (uri[i] == '%' && i + 2 < uri.size())
        {
            s.push_back(static_cast<char>(std::strtoul(uri.substr(i + 1, 2).c_str(), nullptr, 16)));
            i += 2;
        }
        else
            s.push_back(uri[i]);
    return s;
}

struct Gltf
{
    Json json;
    std::filesystem::path base;
    std::vector<std::vector<uint8_t>> buffers;

    // an accessor as floats: count x components (normalised integers are scaled to [0, 1] / [-1, 1])
    std::vector<float> ReadFloats(int64_t accessorIndex, int &components) const
    {
        std::vector<double> d = Read(accessorIndex, components, true);
        return std::vector<float>(d.begin(), d.end());
    }
    std::vector<uint32_t> ReadUints(int64_t accessorIndex, int &components) const
    {
        std::vector<double> d = Read(accessorIndex, components, false);
        std::vector<uint32_t> u(d.size());
        for (size_t i = 0; i < d.size(); i++)
            u[i] = static_cast<uint32_t>(d[i]);
        return u;
    }
    std::span<const uint8_t> BufferView(int64_t viewIndex) const
    {
        const Json &view = json["bufferViews"][static_cast<size_t>(viewIndex)];
        const int64_t buffer = view["buffer"].Int(), offset = view["byteOffset"].Int(0), length = view["byteLength"].Int(0);
        if (viewIndex < 0 || view.IsNull() || buffer < 0 || static_cast<size_t>(buffer) >= buffers.size() || offset < 0 || length < 0)
            throw error("glTF: buffer view out of range");
        // unsigned, overflow-free: offset <= size and length <= size - offset
        const uint64_t size = buffers[static_cast<size_t>(buffer)].size(), uo = static_cast<uint64_t>(offset), ul = static_cast<uint64_t>(length);
        if (uo > size || ul > size - uo)
            throw error("glTF: buffer view out of range");
        return { buffers[static_cast<size_t>(buffer)].data() + uo, static_cast<size_t>(ul) };
    }

private:
    std::vector<double> Read(int64_t accessorIndex, int &components, bool normalise) const
    {
        const Json &acc = json["accessors"][static_cast<size_t>(accessorIndex)];
        if (accessorIndex < 0 || acc.IsNull())
            throw error("glTF: accessor index out of range");
        if (acc.Has("sparse"))
            throw error("glTF: sparse accessors are not supported");
        const std::string &type = acc["type"].Str();
        components = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : type == "MAT3" ? 9 : type == "MAT2" ? 4 : 0;
        const int64_t ct = acc["componentType"].Int(), count = acc["count"].Int(0);
        const size_t cs = (ct == 5120 || ct == 5121) ? 1 : (ct == 5122 || ct == 5123) ? 2 : (ct == 5125 || ct == 5126) ? 4 : 0;
        if (!components || !cs || count < 0)
            throw error("glTF: unsupported accessor type");
        if (acc["bufferView"].IsNull())
        {
            if (count > (1 << 28))
                throw error("glTF: accessor without a buffer view is too large");
            return std::vector<double>(static_cast<size_t>(count) * components, 0.0); // all zeros by definition
        }
        const std::span<const uint8_t> view = BufferView(acc["bufferView"].Int());
        const int64_t accOffset = acc["byteOffset"].Int(0);
        const int64_t declaredStride = json["bufferViews"][static_cast<size_t>(acc["bufferView"].Int())]["byteStride"].Int(0);
        if (accOffset < 0 || declaredStride < 0 || declaredStride > 0xffff)
            throw error("glTF: negative accessor offset or bad byteStride");
        const size_t offset = static_cast<size_t>(accOffset);
        const size_t stride = declaredStride > 0 ? static_cast<size_t>(declaredStride) : cs * components;
        // count elements of `stride` bytes, the last one cs * components long, all inside the view (no wrap-around:
        // every term is checked against what is left)
        const uint64_t elem = cs * static_cast<uint64_t>(components);
        if (count && (offset > view.size() || elem > view.size() - offset ||
                      (static_cast<uint64_t>(count) - 1) > (view.size() - offset - elem) / stride))
            throw error("glTF: accessor reads past its buffer view");
        std::vector<double> out(static_cast<size_t>(count) * components, 0.0); // only now: the count has been checked against the view
        const bool norm = normalise && acc["normalized"].kind == Json::Kind::Bool && acc["normalized"].boolean;
        for (size_t i = 0; i < static_cast<size_t>(count); i++)
            for (int k = 0; k < components; k++)
            {
                const uint8_t *p = view.data() + offset + i * stride + static_cast<size_t>(k) * cs;
                double v = 0;
                switch (ct)
                {
                case 5120: { int8_t x; std::memcpy(&x, p, 1); v = norm ? std::max(x / 127.0, -1.0) : x; break; }
                case 5121: { v = norm ? *p / 255.0 : *p; break; }
                case 5122: { int16_t x; std::memcpy(&x, p, 2); v = norm ? std::max(x / 32767.0, -1.0) : x; break; }
                case 5123: { uint16_t x; std::memcpy(&x, p, 2); v = norm ? x / 65535.0 : x; break; }
                case 5125: { uint32_t x; std::memcpy(&x, p, 4); v = x; break; }
                default: { float x; std::memcpy(&x, p, 4); v = x; break; }
                }
                out[i * components + k] = v;
            }
        return out;
    }
};

Gltf LoadGltf(const std::filesystem::path &path)
{
    Gltf g;
    g.base = path.parent_path();
    const std::vector<uint8_t> file = ReadFileBytes(path);
    std::vector<uint8_t> glbBin;
    std::string text;
    std::string extension = path.extension().string();
    std::transform(extension.begin(), extension.end(), extension.begin(), [](unsigned char c) { return static_cast<char>(std::tolower(c)); });
    if (extension == ".obj") // ObjReader.h
    {
        g.buffers.emplace_back();
        ConvertObjToGltf(file, g.base, g.json, g.buffers.back());
        return g;
    }
    if (IsBinaryFbx(file)) // FbxReader.h: the same document, built from the FBX records
    {
        g.buffers.emplace_back();
        ConvertFbxToGltf(file, g.json, g.buffers.back());
        return g;
    }
    if (file.size() >= 20 && !std::memcmp(file.data(), "glTF", 4))
    {
        auto le32 = [&](size_t o) { return uint32_t(file[o]) | (uint32_t(file[o + 1]) << 8) | (uint32_t(file[o + 2]) << 16) | (uint32_t(file[o + 3]) << 24); };
        if (le32(4) != 2)
            throw error("glTF: only GLB version 2 is supported");
        size_t pos = 12;
        while (pos + 8 <= file.size())
        {
            const uint32_t len = le32(pos), type = le32(pos + 4);
            if (pos + 8 + len > file.size())
                throw error("glTF: truncated GLB chunk");
            if (type == 0x4e4f534a) // "JSON"
                text.assign(reinterpret_cast<const char *>(&file[pos + 8]), len);
            else if (type == 0x004e4942 && glbBin.empty()) // "BIN\0"
                glbBin.assign(file.begin() + static_cast<ptrdiff_t>(pos + 8), file.begin() + static_cast<ptrdiff_t>(pos + 8 + len));
            pos += 8 + ((len + 3) & ~3u);
        }
    }
    else
        text.assign(reinterpret_cast<const char *>(file.data()), file.size());
    g.json = Json::Parse(text);
    if (g.json["asset"]["version"].Str().rfind("2.", 0) != 0)
        throw error("glTF: only version 2.x assets are supported");
    for (size_t i = 0; i < g.json["buffers"].Size(); i++)
    {
        const Json &b = g.json["buffers"][i];
        if (!b.Has("uri"))
            g.buffers.push_back(glbBin);
        else if (b["uri"].Str().rfind("data:", 0) == 0)
        {
            const size_t comma = b["uri"].Str().find(',');
            g.buffers.push_back(DecodeBase64(std::string_view(b["uri"].Str()).substr(comma == std::string::npos ? 0 : comma + 1)));
        }
        else
            g.buffers.push_back(ReadFileBytes(g.base / UriDecode(b["uri"].Str())));
    }
    return g;
}

// ---------------------------------------------------------------------------------------------------------
// small math helpers in the host's math-matrix convention
// ---------------------------------------------------------------------------------------------------------

Mat4 NodeLocalTransform(const Json &node)
{
    if (node["matrix"].Size() == 16) // column-major in the file
    {
        Mat4 m;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++)
                m.m[r][c] = static_cast<float>(node["matrix"][static_cast<size_t>(c * 4 + r)].Num());
        return m;
    }
    Vec3 t(0.0f), s(1.0f);
    Quat q;
    if (node["translation"].Size() == 3)
        t = Vec3(static_cast<float>(node["translation"][0].Num()), static_cast<float>(node["translation"][1].Num()), static_cast<float>(node["translation"][2].Num()));
    if (node["scale"].Size() == 3)
        s = Vec3(static_cast<float>(node["scale"][0].Num()), static_cast<float>(node["scale"][1].Num()), static_cast<float>(node["scale"][2].Num()));
    if (node["rotation"].Size() == 4) // glTF order x y z w
        q = { static_cast<float>(node["rotation"][3].Num()), static_cast<float>(node["rotation"][0].Num()), static_cast<float>(node["rotation"][1].Num()),
              static_cast<float>(node["rotation"][2].Num()) };
    return Scale(Translate(Mat4::Identity(), t) * ToMat4(q), s);
}

PtxTransform ToTransform34(const Mat4 &m)
{
    PtxTransform t;
    std::memcpy(t.m, &m.m[0][0], sizeof(t.m));
    return t;
}

// SceneImporter.cpp:508-517
std::pair<Vec3, Vec3> ComputeTangentSpace(Vec3 normal)
{
    const Vec3 t1 = Cross(normal, Vec3(1.0f, 0.0f, 0.0f)), t2 = Cross(normal, Vec3(0.0f, 1.0f, 0.0f));
    const Vec3 tangent = Length(t1) > Length(t2) ? t1 : t2;
    const Vec3 bitangent = Cross(normal, tangent);
    return { Normalize(tangent), Normalize(bitangent) };
}

bool Same(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

Vec3 SafeNormalize(Vec3 v, Vec3 fallback)
{
    const float l = Length(v);
    return l > 0.0f ? v * (1.0f / l) : fallback;
}

// ---------------------------------------------------------------------------------------------------------
// materials
// ---------------------------------------------------------------------------------------------------------

struct MaterialInfo
{
    uint32_t MaterialIndex;
    MaterialType Type;
    bool IsOpaque;
};

// which glTF texture reference feeds a TextureType slot (assimp's glTF2 importer: baseColor -> BASE_COLOR,
// metallicRoughness -> METALNESS and DIFFUSE_ROUGHNESS, specularGlossiness -> SPECULAR (and SHININESS for gloss))
const Json &TextureRef(const Json &material, TextureType type, bool specGloss)
{
    static const Json none;
    const Json &pbr = material["pbrMetallicRoughness"];
    const Json &sg = material["extensions"]["KHR_materials_pbrSpecularGlossiness"];
    if (material["extras"].Has("assimp")) // an FBX material: the aiTextureType lists of GetTextureTypes (SceneImporter.cpp:33-66)
    {
        const Json &slots = material["extras"]["assimp"]["textures"];
        switch (type)
        {
        case TextureType::Color: return slots["DIFFUSE"];
        case TextureType::Normal: return slots["NORMALS"];
        case TextureType::Emisive: return slots["EMISSIVE"];
        case TextureType::Specular: return slots["SPECULAR"];
        case TextureType::Glossiness:
        case TextureType::Shininess: return slots["SHININESS"];
        default: return none; // DIFFUSE_ROUGHNESS / METALNESS: the FBX converter never fills them
        }
    }
    switch (type)
    {
    case TextureType::Color: return specGloss ? sg["diffuseTexture"] : pbr["baseColorTexture"];
    case TextureType::Normal: return material["normalTexture"];
    case TextureType::Roughness:
    case TextureType::Metallic: return pbr["metallicRoughnessTexture"];
    case TextureType::Emisive: return material["emissiveTexture"];
    case TextureType::Specular:
    case TextureType::Glossiness:
    case TextureType::Shininess: return sg["specularGlossinessTexture"];
    default: return none;
    }
}

// SceneImporter.cpp:68-102: a slot's texture, or the slot's default when there is none / it cannot be loaded
uint32_t AddTexture(SceneBuilder &sb, const Gltf &g, const Json &material, TextureType slot, TextureType source, bool specGloss, bool *isTransparent = nullptr)
{
    if (isTransparent)
        *isTransparent = false;
    const Json &ref = TextureRef(material, source, specGloss);
    const int64_t textureIndex = ref["index"].Int();
    if (textureIndex < 0)
        return Scene::GetDefaultTextureIndex(slot);
    const int64_t imageIndex = g.json["textures"][static_cast<size_t>(textureIndex)]["source"].Int();
    const Json &image = g.json["images"][static_cast<size_t>(imageIndex)];
    if (imageIndex < 0 || image.IsNull())
        return Scene::GetDefaultTextureIndex(slot);
    try
    {
        std::string name = image.Has("uri") && image["uri"].Str().rfind("data:", 0) != 0 ? UriDecode(image["uri"].Str())
                                                                                          : "image " + std::to_string(imageIndex);
        TextureInfo info;
        if (!image["bufferView"].IsNull())
            info = TextureImporter::GetTextureInfo(g.BufferView(image["bufferView"].Int()), slot, std::move(name), isTransparent);
        else if (image["uri"].Str().rfind("data:", 0) == 0)
        {
            const size_t comma = image["uri"].Str().find(',');
            const std::vector<uint8_t> bytes = DecodeBase64(std::string_view(image["uri"].Str()).substr(comma == std::string::npos ? 0 : comma + 1));
            info = TextureImporter::GetTextureInfo(bytes, slot, std::move(name), isTransparent);
        }
        else
            info = TextureImporter::GetTextureInfo((g.base / UriDecode(image["uri"].Str())).lexically_normal(), slot, std::move(name), isTransparent);
        return sb.AddTexture(std::move(info));
    }
    catch (const error &)
    {
        return Scene::GetDefaultTextureIndex(slot);
    }
}

void Copy3(float *dst, const Json &a, float fallback)
{
    for (size_t k = 0; k < 3; k++)
        dst[k] = a.Size() > k ? static_cast<float>(a[k].Num()) : fallback;
}

// SceneImporter.cpp:300-319 on what assimp would expose for a glTF material
MaterialType ChooseMaterialType(const Json &material)
{
    if (material["extras"].Has("assimp")) // what assimp's FBX converter sets: never the metallic / roughness factors
    {
        const Json &a = material["extras"]["assimp"];
        if (a.Has("shininess"))
            return MaterialType::Phong;
        if (a.Has("specularFactor"))
            return MaterialType::SpecularGlossiness;
        return MaterialType::MetallicRoughness;
    }
    if (material.Has("pbrMetallicRoughness"))
        return MaterialType::MetallicRoughness;
    if (material["extensions"].Has("KHR_materials_pbrSpecularGlossiness"))
        return MaterialType::SpecularGlossiness;
    return MaterialType::MetallicRoughness;
}

std::vector<MaterialInfo> LoadMaterials(SceneBuilder &sb, const Gltf &g, const TextureMapping &textureMapping)
{
    // SceneImporter.cpp:327-347
    static const MetallicRoughnessTextureMapping defaultMr = { TextureType::Color, TextureType::Normal, TextureType::Roughness, TextureType::Metallic };
    static const SpecularGlossinessTextureMapping defaultSg = { TextureType::Color, TextureType::Normal, TextureType::Specular, TextureType::Glossiness };
    std::vector<MaterialInfo> infos;
    const Json &materials = g.json["materials"];
    for (size_t i = 0; i <= materials.Size(); i++)
    {
        // the extra pass (i == Size) is assimp's default material for primitives without one
        static const Json defaultMaterial = Json::Parse("{\"pbrMetallicRoughness\":{}}");
        const Json &m = i < materials.Size() ? materials[i] : defaultMaterial;
        std::string name = m["name"].Str().empty() ? "Unnamed Material at index " + std::to_string(i) : m["name"].Str();
        MaterialType type = ChooseMaterialType(m);
        if (std::holds_alternative<MetallicRoughnessTextureMapping>(textureMapping)) type = MaterialType::MetallicRoughness;
        if (std::holds_alternative<SpecularGlossinessTextureMapping>(textureMapping)) type = MaterialType::SpecularGlossiness;
        if (std::holds_alternative<PhongTextureMapping>(textureMapping)) type = MaterialType::Phong;
        // SceneImporter.cpp:390-393: the Phong case has no `break` and runs into `default: throw` -- a material that assimp
        // describes by a shininess (every classic FBX / OBJ material) loads only under a mapping that forces another model,
        // which is how the reference's own FBX scenes are set up (ExampleScenes.cpp:96-141).  Reproduced, not fixed.
        if (type == MaterialType::Phong)
            throw error("Unsupported material type");

        // LoadEmissive (:104-141) / LoadTransmission (:151-167)
        float emissiveColor[3] = { 0, 0, 0 }, emissiveIntensity = 1.0f;
        const Json &ext = m["extensions"];
        const bool specGloss = type == MaterialType::SpecularGlossiness;
        const uint32_t emissiveIdx = AddTexture(sb, g, m, TextureType::Emisive, TextureType::Emisive, specGloss);
        if (ext["KHR_materials_emissive_strength"].Has("emissiveStrength"))
            emissiveIntensity = static_cast<float>(ext["KHR_materials_emissive_strength"]["emissiveStrength"].Num(1.0));
        if (emissiveIdx == Scene::GetDefaultTextureIndex(TextureType::Emisive))
        {
            if (m.Has("emissiveFactor"))
                Copy3(emissiveColor, m["emissiveFactor"], 0.0f);
            else
                emissiveIntensity = 1.0f;
        }
        const float ior = static_cast<float>(ext["KHR_materials_ior"]["ior"].Num(1.5));
        const float transmission = static_cast<float>(ext["KHR_materials_transmission"]["transmissionFactor"].Num(0.0));
        float attenuationColor[3] = { 1, 1, 1 };
        if (ext["KHR_materials_volume"].Has("attenuationColor"))
            Copy3(attenuationColor, ext["KHR_materials_volume"]["attenuationColor"], 1.0f);
        const float attenuationDistance = static_cast<float>(ext["KHR_materials_volume"]["attenuationDistance"].Num(1e32));

        bool hasTransparency = false;
        if (!specGloss)
        {
            const MetallicRoughnessTextureMapping map = std::holds_alternative<MetallicRoughnessTextureMapping>(textureMapping)
                                                            ? std::get<MetallicRoughnessTextureMapping>(textureMapping) : defaultMr;
            const Json &pbr = m["pbrMetallicRoughness"];
            Shaders::MetallicRoughnessMaterial out;
            std::memset(&out, 0, sizeof(out));
            std::memcpy(out.EmissiveColor, emissiveColor, 12);
            out.EmissiveIntensity = emissiveIntensity;
            for (size_t k = 0; k < 4; k++)
                out.Color[k] = pbr["baseColorFactor"].Size() > k ? static_cast<float>(pbr["baseColorFactor"][k].Num()) : 1.0f;
            out.Roughness = static_cast<float>(pbr["roughnessFactor"].Num(1.0));
            out.Metalness = static_cast<float>(pbr["metallicFactor"].Num(1.0));
            out.Ior = ior;
            out.Transmission = transmission;
            std::memcpy(out.AttenuationColor, attenuationColor, 12);
            out.AttenuationDistance = attenuationDistance;
            out.EmissiveIdx = emissiveIdx;
            out.ColorIdx = AddTexture(sb, g, m, TextureType::Color, map.ColorTexture, false, &hasTransparency);
            out.NormalIdx = AddTexture(sb, g, m, TextureType::Normal, map.NormalTexture, false);
            out.RoughnessIdx = AddTexture(sb, g, m, TextureType::Roughness, map.RoughnessTexture, false);
            out.MetallicIdx = AddTexture(sb, g, m, TextureType::Metallic, map.MetallicTexture, false);
            infos.push_back({ sb.AddMaterial(name, out), MaterialType::MetallicRoughness, !hasTransparency });
        }
        else
        {
            const SpecularGlossinessTextureMapping map = std::holds_alternative<SpecularGlossinessTextureMapping>(textureMapping)
                                                             ? std::get<SpecularGlossinessTextureMapping>(textureMapping) : defaultSg;
            const Json &sg = ext["KHR_materials_pbrSpecularGlossiness"];
            Shaders::SpecularGlossinessMaterial out;
            std::memset(&out, 0, sizeof(out));
            std::memcpy(out.EmissiveColor, emissiveColor, 12);
            out.EmissiveIntensity = emissiveIntensity;
            const Json &fbx = m["extras"]["assimp"]; // AI_MATKEY_COLOR_DIFFUSE / SPECULAR_FACTOR of an FBX material
            const Json &diffuse = fbx.Has("diffuse") ? fbx["diffuse"] : sg["diffuseFactor"];
            for (size_t k = 0; k < 4; k++)
                out.Color[k] = diffuse.Size() > k ? static_cast<float>(diffuse[k].Num()) : 1.0f;
            Copy3(out.Specular, sg["specularFactor"], 1.0f);
            if (fbx.Has("specularFactor")) // a float read as a colour: only the first component arrives
                out.Specular[0] = static_cast<float>(fbx["specularFactor"].Num(1.0));
            out.Glossiness = static_cast<float>(sg["glossinessFactor"].Num(1.0));
            std::memcpy(out.AttenuationColor, attenuationColor, 12);
            out.AttenuationDistance = attenuationDistance;
            out.Ior = ior;
            out.Transmission = transmission;
            out.EmissiveIdx = emissiveIdx;
            out.ColorIdx = AddTexture(sb, g, m, TextureType::Color, map.ColorTexture, true, &hasTransparency);
            out.NormalIdx = AddTexture(sb, g, m, TextureType::Normal, map.NormalTexture, true);
            out.SpecularIdx = AddTexture(sb, g, m, TextureType::Specular, map.SpecularTexture, true);
            out.GlossinessIdx = AddTexture(sb, g, m, TextureType::Glossiness, map.GlossinessTexture, true);
            infos.push_back({ sb.AddMaterial(name, out), MaterialType::SpecularGlossiness, !hasTransparency });
        }
    }
    return infos;
}

// ---------------------------------------------------------------------------------------------------------
// meshes (one glTF primitive = one aiMesh)
// ---------------------------------------------------------------------------------------------------------

struct Primitive
{
    size_t mesh, index;     // position in json["meshes"][mesh]["primitives"][index]
    int64_t material;       // -1 = default
    int64_t skin = -1;      // skin of the node that instantiates it (skinned only)
    bool animated = false;
    uint32_t geometry = 0;
};

}

SceneBuilder &SceneImporter::AddFile(SceneBuilder &sb, const std::filesystem::path &path, TextureMapping textureMapping)
{
    const Gltf g = LoadGltf(path);
    const Json &J = g.json;

    // ---- scene nodes: a synthetic root (assimp's ROOT) above the scene's root nodes, pre-order through a stack
    const int64_t sceneIndex = J["scene"].Int(0);
    const Json &roots = J["scenes"][static_cast<size_t>(sceneIndex)]["nodes"];
    std::vector<int64_t> order;                   // glTF node index per imported node, -1 = the synthetic root
    std::vector<uint32_t> sceneNodeOf;            // imported position -> SceneBuilder node index
    std::unordered_map<int64_t, uint32_t> nodeToScene; // glTF node -> SceneBuilder node index
    std::unordered_map<int64_t, int64_t> parentOf;
    {
        std::stack<std::tuple<int64_t, uint32_t>> stack;
        stack.emplace(-1, SceneBuilder::RootNodeIndex);
        while (!stack.empty())
        {
            auto [node, parentSceneNode] = stack.top();
            stack.pop();
            const Mat4 local = node < 0 ? Mat4::Identity() : NodeLocalTransform(J["nodes"][static_cast<size_t>(node)]);
            const uint32_t sceneNode = sb.AddSceneNode({ parentSceneNode, local, Mat4::Identity() });
            order.push_back(node);
            sceneNodeOf.push_back(sceneNode);
            nodeToScene[node] = sceneNode;
            const Json &children = node < 0 ? roots : J["nodes"][static_cast<size_t>(node)]["children"];
            for (size_t i = 0; i < children.Size(); i++)
            {
                parentOf[children[i].Int()] = node;
                stack.emplace(children[i].Int(), sceneNode);
            }
        }
    }

    const std::vector<MaterialInfo> materialInfos = LoadMaterials(sb, g, textureMapping);
    const size_t defaultMaterial = materialInfos.size() - 1;

    // ---- which meshes are skinned: a mesh instantiated by a node with a skin
    std::unordered_map<size_t, int64_t> meshSkin;
    for (int64_t node : order)
        if (node >= 0)
        {
            const Json &n = J["nodes"][static_cast<size_t>(node)];
            if (!n["mesh"].IsNull() && !n["skin"].IsNull())
                meshSkin[static_cast<size_t>(n["mesh"].Int())] = n["skin"].Int();
        }

    // ---- geometry (LoadMeshes, SceneImporter.cpp:455-624)
    auto &vertices = sb.GetVertices();
    auto &indices = sb.GetIndices();
    auto &animatedVertices = sb.GetAnimatedVertices();
    auto &animatedIndices = sb.GetAnimatedIndices();
    std::vector<std::vector<Primitive>> meshPrimitives(J["meshes"].Size());
    std::map<std::tuple<int64_t, int64_t, int64_t, int64_t, int64_t>, uint32_t> sameGeometry; // (indices, POSITION, NORMAL, TEXCOORD_0, skin) -> geometry
    std::set<int64_t> armatureRoots;
    for (size_t mi = 0; mi < J["meshes"].Size(); mi++)
    {
        const Json &prims = J["meshes"][mi]["primitives"];
        for (size_t pi = 0; pi < prims.Size(); pi++)
        {
            const Json &prim = prims[pi];
            if (prim["mode"].Int(4) != 4 || prim["attributes"]["POSITION"].IsNull())
                continue; // points / lines / strips: not imported
            Primitive out { mi, pi, prim["material"].Int(-1) };
            const auto skinIt = meshSkin.find(mi);
            out.animated = skinIt != meshSkin.end() && !prim["attributes"]["JOINTS_0"].IsNull() && !prim["attributes"]["WEIGHTS_0"].IsNull();
            out.skin = out.animated ? skinIt->second : -1;
            const size_t materialSlot = out.material >= 0 && static_cast<size_t>(out.material) < defaultMaterial ? static_cast<size_t>(out.material) : defaultMaterial;

            // FindSameGeometry (:402-413): primitives that differ only in material share one geometry
            const auto key = std::make_tuple(prim["indices"].Int(), prim["attributes"]["POSITION"].Int(), prim["attributes"]["NORMAL"].Int(),
                                             prim["attributes"]["TEXCOORD_0"].Int(), out.skin);
            const auto same = sameGeometry.find(key);
            if (same != sameGeometry.end())
            {
                out.geometry = same->second;
                meshPrimitives[mi].push_back(out);
                continue;
            }

            int comps = 0;
            const std::vector<float> pos = g.ReadFloats(prim["attributes"]["POSITION"].Int(), comps);
            if (comps != 3)
                throw error("glTF: POSITION must be VEC3");
            const uint32_t vertexCount = static_cast<uint32_t>(pos.size() / 3);
            std::vector<uint32_t> idx;
            if (!prim["indices"].IsNull())
                idx = g.ReadUints(prim["indices"].Int(), comps);
            else
                for (uint32_t v = 0; v < vertexCount; v++)
                    idx.push_back(v);
            idx.resize(idx.size() / 3 * 3);
            for (uint32_t v : idx)
                if (v >= vertexCount)
                    throw error("glTF: index beyond the vertex count");
            std::vector<float> nrm, uv, tan;
            if (!prim["attributes"]["NORMAL"].IsNull())
                nrm = g.ReadFloats(prim["attributes"]["NORMAL"].Int(), comps);
            if (!prim["attributes"]["TEXCOORD_0"].IsNull())
            {
                uv = g.ReadFloats(prim["attributes"]["TEXCOORD_0"].Int(), comps);
                if (comps != 2) uv.clear();
            }
            if (!prim["attributes"]["TANGENT"].IsNull())
            {
                tan = g.ReadFloats(prim["attributes"]["TANGENT"].Int(), comps);
                if (comps != 4) tan.clear();
            }
            if (nrm.size() != pos.size()) // aiProcess_GenNormals stand-in: area-weighted vertex normals
            {
                nrm.assign(pos.size(), 0.0f);
                for (size_t t = 0; t + 2 < idx.size(); t += 3)
                {
                    const Vec3 a(pos[idx[t] * 3], pos[idx[t] * 3 + 1], pos[idx[t] * 3 + 2]), b(pos[idx[t + 1] * 3], pos[idx[t + 1] * 3 + 1], pos[idx[t + 1] * 3 + 2]),
                        c(pos[idx[t + 2] * 3], pos[idx[t + 2] * 3 + 1], pos[idx[t + 2] * 3 + 2]);
                    const Vec3 fn = Cross(b - a, c - a);
                    for (int k = 0; k < 3; k++)
                    {
                        nrm[idx[t + k] * 3] += fn.x; nrm[idx[t + k] * 3 + 1] += fn.y; nrm[idx[t + k] * 3 + 2] += fn.z;
                    }
                }
                for (uint32_t v = 0; v < vertexCount; v++)
                {
                    const Vec3 n = SafeNormalize(Vec3(nrm[v * 3], nrm[v * 3 + 1], nrm[v * 3 + 2]), Vec3(0, 1, 0));
                    nrm[v * 3] = n.x; nrm[v * 3 + 1] = n.y; nrm[v * 3 + 2] = n.z;
                }
            }

            // aiProcess_CalcTangentSpace (the reference's import flags, SceneImporter.cpp:1066-1068) for a mesh that has texture
            // coordinates but no tangents -- every FBX mesh, glTF primitives without TANGENT: per triangle the directions of
            // increasing u and v (on the coordinates aiProcess_FlipUVs has already turned), summed per vertex, then made
            // orthogonal to the normal.  assimp joins the per-face results of vertices within 45 degrees; the shared indexed
            // vertex does the same job here.
            std::vector<float> faceT, faceB;
            if (tan.empty() && !uv.empty())
            {
                faceT.assign(pos.size(), 0.0f);
                faceB.assign(pos.size(), 0.0f);
                for (size_t f = 0; f + 2 < idx.size(); f += 3)
                {
                    const uint32_t i0 = idx[f], i1 = idx[f + 1], i2 = idx[f + 2];
                    const Vec3 p0(pos[i0 * 3], pos[i0 * 3 + 1], pos[i0 * 3 + 2]);
                    const Vec3 e1 = Vec3(pos[i1 * 3], pos[i1 * 3 + 1], pos[i1 * 3 + 2]) - p0, e2 = Vec3(pos[i2 * 3], pos[i2 * 3 + 1], pos[i2 * 3 + 2]) - p0;
                    float sx = uv[i1 * 2] - uv[i0 * 2], sy = (1.0f - uv[i1 * 2 + 1]) - (1.0f - uv[i0 * 2 + 1]);
                    float tx = uv[i2 * 2] - uv[i0 * 2], ty = (1.0f - uv[i2 * 2 + 1]) - (1.0f - uv[i0 * 2 + 1]);
                    const float dir = (tx * sy - ty * sx) < 0.0f ? -1.0f : 1.0f;
                    if (sx * ty == sy * tx) // degenerate mapping: assimp substitutes an axis-aligned one
                    {
                        sx = 0.0f; sy = 1.0f; tx = 1.0f; ty = 0.0f;
                    }
                    const Vec3 ft = (e2 * sy - e1 * ty) * dir, fb = (e1 * tx - e2 * sx) * dir; // towards growing u, growing v
                    for (uint32_t v : { i0, i1, i2 })
                    {
                        faceT[v * 3] += ft.x; faceT[v * 3 + 1] += ft.y; faceT[v * 3 + 2] += ft.z;
                        faceB[v * 3] += fb.x; faceB[v * 3 + 1] += fb.y; faceB[v * 3 + 2] += fb.z;
                    }
                }
            }

            const uint32_t vo = static_cast<uint32_t>(out.animated ? animatedVertices.size() : vertices.size());
            const uint32_t io = static_cast<uint32_t>(out.animated ? animatedIndices.size() : indices.size());
            for (uint32_t v = 0; v < vertexCount; v++)
            {
                const Vec3 n(nrm[v * 3], nrm[v * 3 + 1], nrm[v * 3 + 2]);
                Vec3 t, b;
                if (!tan.empty())
                {
                    t = Vec3(tan[v * 4], tan[v * 4 + 1], tan[v * 4 + 2]);
                    b = Cross(n, t) * tan[v * 4 + 3];
                    if (Same(n, t) || Same(n, b) || Same(t, b)) // :520-528
                        std::tie(t, b) = ComputeTangentSpace(n);
                }
                else if (!faceT.empty())
                {
                    const Vec3 st(faceT[v * 3], faceT[v * 3 + 1], faceT[v * 3 + 2]), sb(faceB[v * 3], faceB[v * 3 + 1], faceB[v * 3 + 2]);
                    Vec3 lt = st - n * Dot(n, st);
                    const float tl = std::sqrt(Dot(lt, lt));
                    if (tl > 0.0f && std::isfinite(tl))
                    {
                        lt = lt * (1.0f / tl);
                        Vec3 lb = sb - n * Dot(n, sb) - lt * Dot(lt, sb);
                        const float bl = std::sqrt(Dot(lb, lb));
                        t = lt;
                        b = (bl > 0.0f && std::isfinite(bl)) ? lb * (1.0f / bl) : Cross(n, lt);
                    }
                    else
                        std::tie(t, b) = ComputeTangentSpace(n);
                    if (Same(n, t) || Same(n, b) || Same(t, b)) // :520-528
                        std::tie(t, b) = ComputeTangentSpace(n);
                }
                else
                    std::tie(t, b) = ComputeTangentSpace(n);
                const float tu = uv.empty() ? 0.0f : uv[v * 2], tv = uv.empty() ? 0.0f : 1.0f - uv[v * 2 + 1]; // aiProcess_FlipUVs
                if (out.animated)
                {
                    Shaders::AnimatedVertex a;
                    std::memset(&a, 0, sizeof(a));
                    a.Position[0] = pos[v * 3]; a.Position[1] = pos[v * 3 + 1]; a.Position[2] = pos[v * 3 + 2];
                    a.TexCoords[0] = tu; a.TexCoords[1] = tv;
                    a.Normal[0] = n.x; a.Normal[1] = n.y; a.Normal[2] = n.z;
                    a.Tangent[0] = t.x; a.Tangent[1] = t.y; a.Tangent[2] = t.z;
                    a.Bitangent[0] = b.x; a.Bitangent[1] = b.y; a.Bitangent[2] = b.z;
                    animatedVertices.push_back(a);
                }
                else
                    vertices.push_back({ { pos[v * 3], pos[v * 3 + 1], pos[v * 3 + 2] }, { tu, tv }, { n.x, n.y, n.z }, { t.x, t.y, t.z }, { b.x, b.y, b.z } });
            }
            (out.animated ? animatedIndices : indices).insert((out.animated ? animatedIndices : indices).end(), idx.begin(), idx.end());

            if (out.animated) // LoadBones (:420-453): one Bone per joint of the skin, for this mesh
            {
                const Json &skin = J["skins"][static_cast<size_t>(out.skin)];
                const Json &joints = skin["joints"];
                std::vector<float> ibm;
                if (!skin["inverseBindMatrices"].IsNull())
                    ibm = g.ReadFloats(skin["inverseBindMatrices"].Int(), comps);
                std::vector<uint32_t> boneOfJoint(joints.Size());
                for (size_t j = 0; j < joints.Size(); j++)
                {
                    Mat4 offset = Mat4::Identity();
                    if (ibm.size() >= (j + 1) * 16)
                        for (int c = 0; c < 4; c++)
                            for (int r = 0; r < 4; r++)
                                offset.m[r][c] = ibm[j * 16 + static_cast<size_t>(c * 4 + r)];
                    const auto it = nodeToScene.find(joints[j].Int());
                    if (it == nodeToScene.end())
                        throw error("glTF: a skin joint is not part of the scene");
                    boneOfJoint[j] = sb.AddBone({ it->second, offset });
                    // the armature: the top-most ancestor below the synthetic root
                    int64_t top = joints[j].Int();
                    while (parentOf.count(top) && parentOf[top] >= 0)
                        top = parentOf[top];
                    armatureRoots.insert(top);
                }
                const std::vector<uint32_t> jointIdx = g.ReadUints(prim["attributes"]["JOINTS_0"].Int(), comps);
                const int jointComps = comps;
                const std::vector<float> weights = g.ReadFloats(prim["attributes"]["WEIGHTS_0"].Int(), comps);
                for (uint32_t v = 0; v < vertexCount && jointComps == 4 && comps == 4; v++)
                {
                    Shaders::AnimatedVertex &a = animatedVertices[vo + v];
                    int slot = 0;
                    for (int k = 0; k < 4; k++)
                    {
                        const float w = weights[v * 4 + static_cast<size_t>(k)];
                        const uint32_t joint = jointIdx[v * 4 + static_cast<size_t>(k)];
                        if (w <= 0.0f || joint >= boneOfJoint.size())
                            continue;
                        a.BoneIndices[slot] = boneOfJoint[joint];
                        a.BoneWeights[slot] = w;
                        slot++;
                    }
                }
            }

            const bool isOpaque = materialInfos[materialSlot].IsOpaque;
            out.geometry = sb.AddGeometry({ vo, vertexCount, io, static_cast<uint32_t>(idx.size()), isOpaque, out.animated, { 0, 0 } });
            sameGeometry[key] = out.geometry;
            meshPrimitives[mi].push_back(out);
        }
    }

    // ---- dynamic nodes (FindDynamicNodes, :626-669): every node an animation channel targets
    std::set<int64_t> dynamicNodes;
    for (size_t a = 0; a < J["animations"].Size(); a++)
        for (size_t c = 0; c < J["animations"][a]["channels"].Size(); c++)
        {
            const Json &target = J["animations"][a]["channels"][c]["target"];
            if (!target["node"].IsNull() && nodeToScene.count(target["node"].Int()) && target["path"].Str() != "weights")
                dynamicNodes.insert(target["node"].Int());
        }

    // ---- models (LoadModels, :708-837)
    {
        auto isInstanceRoot = [&](int64_t node) { return node < 0 || dynamicNodes.count(node) != 0; };
        std::vector<std::vector<MeshInfo>> modelToMeshInfos, modelToAnimatedMeshInfos;
        std::vector<uint32_t> modelToSceneNode;
        std::unordered_map<int64_t, uint32_t> nodeToModel;
        std::unordered_map<int64_t, Mat4> nodeToMeshTransform;
        auto newModel = [&](uint32_t sceneNode) {
            modelToMeshInfos.emplace_back();
            modelToAnimatedMeshInfos.emplace_back();
            modelToSceneNode.push_back(sceneNode);
            return static_cast<uint32_t>(modelToSceneNode.size() - 1);
        };
        for (size_t k = 0; k < order.size(); k++)
        {
            const int64_t node = order[k];
            uint32_t modelIndex;
            Mat4 total;
            if (isInstanceRoot(node))
            {
                modelIndex = newModel(sceneNodeOf[k]);
                total = Mat4::Identity();
            }
            else
            {
                const int64_t parent = parentOf[node];
                modelIndex = nodeToModel[parent];
                // "node.Transform * parent total" of the transposed glm form = parent total * node local here
                total = nodeToMeshTransform[parent] * NodeLocalTransform(J["nodes"][static_cast<size_t>(node)]);
            }
            nodeToModel[node] = modelIndex;
            nodeToMeshTransform[node] = total;
            if (node < 0 || J["nodes"][static_cast<size_t>(node)]["mesh"].IsNull())
                continue;
            const size_t mesh = static_cast<size_t>(J["nodes"][static_cast<size_t>(node)]["mesh"].Int());
            if (mesh >= meshPrimitives.size())
                continue;
            bool hasAnimated = false;
            for (const Primitive &p : meshPrimitives[mesh])
            {
                if (p.animated)
                {
                    hasAnimated = true;
                    continue;
                }
                const MaterialInfo &mat = materialInfos[p.material >= 0 && static_cast<size_t>(p.material) < defaultMaterial ? static_cast<size_t>(p.material) : defaultMaterial];
                modelToMeshInfos[modelIndex].push_back({ p.geometry, mat.MaterialIndex, mat.Type, ToTransform34(total) });
            }
            if (hasAnimated)
            {
                // a skinned mesh is its own instance attached to the node's parent; the bones of that subtree are
                // then expressed relative to it (SetAbsoluteTransform on the parent's children, :789-803)
                const int64_t ancestor = parentOf.count(node) ? parentOf[node] : -1;
                const uint32_t animatedModel = newModel(nodeToScene[ancestor]);
                const Json &siblings = ancestor < 0 ? roots : J["nodes"][static_cast<size_t>(ancestor)]["children"];
                for (size_t c = 0; c < siblings.Size(); c++)
                    sb.SetAbsoluteTransform(nodeToScene[siblings[c].Int()]);
                for (const Primitive &p : meshPrimitives[mesh])
                    if (p.animated)
                    {
                        const MaterialInfo &mat = materialInfos[p.material >= 0 && static_cast<size_t>(p.material) < defaultMaterial ? static_cast<size_t>(p.material) : defaultMaterial];
                        modelToAnimatedMeshInfos[animatedModel].push_back({ p.geometry, mat.MaterialIndex, mat.Type, IdentityTransform() });
                    }
            }
        }
        for (size_t i = 0; i < modelToSceneNode.size(); i++)
        {
            if (!modelToMeshInfos[i].empty())
                sb.AddModelInstance(sb.AddModel(modelToMeshInfos[i]), modelToSceneNode[i]);
            if (!modelToAnimatedMeshInfos[i].empty())
                sb.AddModelInstance(sb.AddModel(modelToAnimatedMeshInfos[i]), modelToSceneNode[i]);
        }
    }

    // ---- animations (LoadAnimations, :840-917): assimp's glTF2 importer reports milliseconds at 1000 ticks per second
    for (size_t a = 0; a < J["animations"].Size(); a++)
    {
        const Json &anim = J["animations"][a];
        std::map<int64_t, AnimationNode> perNode;
        float duration = 0.0f;
        for (size_t c = 0; c < anim["channels"].Size(); c++)
        {
            const Json &channel = anim["channels"][c];
            const int64_t node = channel["target"]["node"].Int();
            const std::string &pathName = channel["target"]["path"].Str();
            const Json &sampler = anim["samplers"][static_cast<size_t>(channel["sampler"].Int())];
            if (!nodeToScene.count(node) || sampler.IsNull() || pathName == "weights")
                continue;
            int comps = 0;
            const std::vector<float> times = g.ReadFloats(sampler["input"].Int(), comps);
            const std::vector<float> values = g.ReadFloats(sampler["output"].Int(), comps);
            const bool cubic = sampler["interpolation"].Str() == "CUBICSPLINE"; // in-tangent, value, out-tangent: keep the value
            AnimationNode &out = perNode[node];
            out.SceneNodeIndex = nodeToScene[node];
            for (size_t k = 0; k < times.size(); k++)
            {
                const size_t e = (cubic ? 3 * k + 1 : k) * static_cast<size_t>(comps);
                if (e + static_cast<size_t>(comps) > values.size())
                    break;
                const float tick = times[k] * 1000.0f;
                duration = std::max(duration, tick);
                if (pathName == "translation" && comps == 3)
                    out.Positions.Keys.push_back({ Vec3(values[e], values[e + 1], values[e + 2]), tick });
                else if (pathName == "scale" && comps == 3)
                    out.Scales.Keys.push_back({ Vec3(values[e], values[e + 1], values[e + 2]), tick });
                else if (pathName == "rotation" && comps == 4)
                    out.Rotations.Keys.push_back({ Quat { values[e + 3], values[e], values[e + 1], values[e + 2] }, tick });
            }
        }
        Animation outAnimation { {}, 1000.0f, duration };
        for (auto &[node, animNode] : perNode)
        {
            // a channel the file does not animate keeps the node's own TRS as its only key (assimp does the same)
            const Json &n = J["nodes"][static_cast<size_t>(node)];
            if (animNode.Positions.Keys.empty())
                animNode.Positions.Keys.push_back({ n["translation"].Size() == 3 ? Vec3(static_cast<float>(n["translation"][0].Num()), static_cast<float>(n["translation"][1].Num()),
                                                                                       static_cast<float>(n["translation"][2].Num())) : Vec3(0.0f), 0.0f });
            if (animNode.Rotations.Keys.empty())
                animNode.Rotations.Keys.push_back({ n["rotation"].Size() == 4 ? Quat { static_cast<float>(n["rotation"][3].Num()), static_cast<float>(n["rotation"][0].Num()),
                                                                                        static_cast<float>(n["rotation"][1].Num()), static_cast<float>(n["rotation"][2].Num()) } : Quat(), 0.0f });
            if (animNode.Scales.Keys.empty())
                animNode.Scales.Keys.push_back({ n["scale"].Size() == 3 ? Vec3(static_cast<float>(n["scale"][0].Num()), static_cast<float>(n["scale"][1].Num()),
                                                                               static_cast<float>(n["scale"][2].Num())) : Vec3(1.0f), 0.0f });
            outAnimation.Nodes.push_back(std::move(animNode));
        }
        if (!outAnimation.Nodes.empty() && duration > 0.0f)
            sb.AddAnimation(std::move(outAnimation));
    }

    // ---- lights (LoadLights, :919-995) from KHR_lights_punctual, cameras (LoadCameras, :997-1029)
    bool hasDirectionalLight = false;
    const Json &lightDefs = J["extensions"]["KHR_lights_punctual"]["lights"];
    for (size_t k = 0; k < order.size(); k++)
    {
        if (order[k] < 0)
            continue;
        const Json &n = J["nodes"][static_cast<size_t>(order[k])];
        const int64_t lightIndex = n["extensions"]["KHR_lights_punctual"]["light"].Int();
        if (lightIndex >= 0 && !lightDefs[static_cast<size_t>(lightIndex)].IsNull())
        {
            const Json &l = lightDefs[static_cast<size_t>(lightIndex)];
            const float intensity = static_cast<float>(l["intensity"].Num(1.0));
            float color[3];
            Copy3(color, l["color"], 1.0f);
            for (float &c : color)
                c *= intensity;
            if (color[0] == 0.0f && color[1] == 0.0f && color[2] == 0.0f)
                color[0] = color[1] = color[2] = 10.0f; // :957-959
            if (l["type"].Str() == "directional")
            {
                if (!hasDirectionalLight)
                {
                    Shaders::DirectionalLight d;
                    std::memset(&d, 0, sizeof(d));
                    std::memcpy(d.Color, color, 12);
                    d.Direction[2] = -1.0f; // a glTF light shines along its local -z
                    sb.SetDirectionalLight(std::move(d), sceneNodeOf[k]);
                    hasDirectionalLight = true;
                }
            }
            else // point, and spot treated as point (:953-955)
            {
                Shaders::PointLight p;
                std::memset(&p, 0, sizeof(p));
                std::memcpy(p.Color, color, 12);
                p.AttenuationQuadratic = 1.0f;
                sb.AddLight(std::move(p), sceneNodeOf[k]);
            }
        }
        const int64_t cameraIndex = n["camera"].Int();
        const Json &cam = J["cameras"][static_cast<size_t>(cameraIndex >= 0 ? cameraIndex : 0)];
        if (cameraIndex >= 0 && cam["type"].Str() == "perspective")
        {
            const Json &p = cam["perspective"];
            const float yfov = static_cast<float>(p["yfov"].Num(0.7853981633974483));
            sb.AddCamera({ yfov * 57.29577951308232f, static_cast<float>(p["znear"].Num(0.1)), static_cast<float>(p["zfar"].Num(1000.0)), Vec3(0.0f),
                           Vec3(0.0f, 0.0f, -1.0f), Vec3(0.0f, -1.0f, 0.0f), sceneNodeOf[k] }); // up.y flipped, :1015-1016
        }
    }
    return sb;
}

}
