#include "ObjReader.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <tuple>

#include "TextureImporter.h"

namespace PathTracing
{

namespace
{

Json Num(double v) { Json j; j.kind = Json::Kind::Number; j.number = v; return j; }
Json Str(const std::string &s) { Json j; j.kind = Json::Kind::String; j.string = s; return j; }
Json Obj() { Json j; j.kind = Json::Kind::Object; return j; }
Json Arr() { Json j; j.kind = Json::Kind::Array; return j; }
Json Nums(const double *v, size_t n)
{
    Json a = Arr();
    for (size_t k = 0; k < n; k++)
        a.array.push_back(Num(v[k]));
    return a;
}

struct MtlMaterial
{
    std::string name;
    double kd[3] = { 0.6, 0.6, 0.6 }, ke[3] = { 0, 0, 0 }; // assimp's defaults for a material without the statement
    bool hasKe = false;
    double ns = 0.0, d = 1.0;
    std::map<std::string, std::string> maps; // aiTextureType name -> file
};

std::string Trim(const std::string &s)
{
    const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// the file name of a map statement: the last token (options like -bm 1.0 come before it), backslashes to slashes
std::string MapFile(const std::string &rest)
{
    std::string name = Trim(rest);
    const size_t cut = name.find_last_of(" \t");
    if (cut != std::string::npos && (name[0] == '-' || name.find(" -") != std::string::npos))
        name = name.substr(cut + 1);
    std::replace(name.begin(), name.end(), '\\', '/');
    return name;
}

std::vector<MtlMaterial> ReadMtl(const std::filesystem::path &path)
{
    std::vector<MtlMaterial> out;
    std::vector<uint8_t> bytes;
    try
    {
        bytes = ReadFileBytes(path);
    }
    catch (const error &)
    {
        return out; // a missing library leaves the materials at their defaults, as assimp does
    }
    std::istringstream in(std::string(bytes.begin(), bytes.end()));
    std::string line;
    while (std::getline(in, line))
    {
        line = Trim(line);
        if (line.empty() || line[0] == '#')
            continue;
        const size_t sp = line.find_first_of(" \t");
        const std::string key = line.substr(0, sp), rest = sp == std::string::npos ? std::string() : Trim(line.substr(sp));
        if (key == "newmtl")
        {
            out.emplace_back();
            out.back().name = rest;
            continue;
        }
        if (out.empty())
            continue;
        MtlMaterial &m = out.back();
        auto three = [&](double *v) {
            std::istringstream s(rest);
            double a = 0, b = 0, c = 0;
            if (s >> a)
            {
                b = c = a;
                if (s >> b)
                    s >> c;
                v[0] = a; v[1] = b; v[2] = c;
            }
        };
        if (key == "Kd") three(m.kd);
        else if (key == "Ke") { three(m.ke); m.hasKe = true; }
        else if (key == "Ns") m.ns = std::atof(rest.c_str());
        else if (key == "d") m.d = std::atof(rest.c_str());
        else if (key == "Tr") m.d = 1.0 - std::atof(rest.c_str());
        else if (key == "map_Kd") m.maps["DIFFUSE"] = MapFile(rest);
        else if (key == "map_Ks") m.maps["SPECULAR"] = MapFile(rest);
        else if (key == "map_Ns") m.maps["SHININESS"] = MapFile(rest);
        else if (key == "map_Ke") m.maps["EMISSIVE"] = MapFile(rest);
        else if (key == "norm" || key == "map_Kn") m.maps["NORMALS"] = MapFile(rest);
    }
    return out;
}

struct Part // one mesh of the document: an object / group cut by material
{
    std::string name;
    int64_t material = -1;
    std::vector<float> position, normal, uv;
    std::vector<uint32_t> index;
    bool anyNormal = false, anyUv = false;
    std::map<std::tuple<int64_t, int64_t, int64_t>, uint32_t> seen;
};

}

void ConvertObjToGltf(std::span<const uint8_t> file, const std::filesystem::path &directory, Json &json, std::vector<uint8_t> &buffer)
{
    std::vector<double> v, vt, vn;
    std::vector<MtlMaterial> materials;
    std::map<std::string, int64_t> materialByName;
    std::vector<Part> parts;
    std::string groupName = "default";
    int64_t currentMaterial = -1;
    auto partFor = [&]() -> Part & {
        if (parts.empty() || parts.back().name != groupName || parts.back().material != currentMaterial)
        {
            for (Part &p : parts) // the same (group, material) again: continue that mesh
                if (p.name == groupName && p.material == currentMaterial)
                    return p;
            parts.emplace_back();
            parts.back().name = groupName;
            parts.back().material = currentMaterial;
        }
        return parts.back();
    };
    std::istringstream in(std::string(file.begin(), file.end()));
    std::string line;
    size_t lineNo = 0;
    while (std::getline(in, line))
    {
        lineNo++;
        line = Trim(line);
        while (!line.empty() && line.back() == '\\') // continuation
        {
            std::string next;
            if (!std::getline(in, next))
                break;
            line.pop_back();
            line += " " + Trim(next);
        }
        if (line.empty() || line[0] == '#')
            continue;
        const size_t sp = line.find_first_of(" \t");
        const std::string key = line.substr(0, sp), rest = sp == std::string::npos ? std::string() : Trim(line.substr(sp));
        if (key == "v" || key == "vn" || key == "vt")
        {
            std::istringstream s(rest);
            double a = 0, b = 0, c = 0;
            s >> a >> b >> c;
            std::vector<double> &dst = key == "v" ? v : key == "vn" ? vn : vt;
            dst.push_back(a);
            dst.push_back(b);
            if (key != "vt")
                dst.push_back(c);
        }
        else if (key == "o" || key == "g")
            groupName = rest.empty() ? "default" : rest;
        else if (key == "mtllib")
        {
            std::string name = rest;
            std::replace(name.begin(), name.end(), '\\', '/');
            for (MtlMaterial &m : ReadMtl((directory / name).lexically_normal()))
                if (!materialByName.count(m.name))
                {
                    materialByName[m.name] = static_cast<int64_t>(materials.size());
                    materials.push_back(std::move(m));
                }
        }
        else if (key == "usemtl")
        {
            const auto it = materialByName.find(rest);
            if (it == materialByName.end()) // named but not defined: a default material of that name
            {
                materialByName[rest] = static_cast<int64_t>(materials.size());
                materials.emplace_back();
                materials.back().name = rest;
                currentMaterial = static_cast<int64_t>(materials.size() - 1);
            }
            else
                currentMaterial = it->second;
        }
        else if (key == "f")
        {
            Part &part = partFor();
            std::istringstream s(rest);
            std::string corner;
            std::vector<uint32_t> polygon;
            while (s >> corner)
            {
                int64_t idx[3] = { 0, 0, 0 }; // v / vt / vn, 1-based, negative = relative to the end, 0 = absent
                size_t start = 0;
                for (int k = 0; k < 3 && start <= corner.size(); k++)
                {
                    const size_t slash = corner.find('/', start);
                    const std::string field = corner.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (!field.empty())
                        idx[k] = std::strtoll(field.c_str(), nullptr, 10);
                    if (slash == std::string::npos)
                        break;
                    start = slash + 1;
                }
                const int64_t counts[3] = { static_cast<int64_t>(v.size() / 3), static_cast<int64_t>(vt.size() / 2), static_cast<int64_t>(vn.size() / 3) };
                for (int k = 0; k < 3; k++)
                {
                    if (idx[k] < 0)
                        idx[k] = counts[k] + idx[k] + 1;
                    if (idx[k] < 0 || idx[k] > counts[k])
                        throw error("OBJ: line " + std::to_string(lineNo) + ": index out of range");
                }
                if (idx[0] == 0)
                    throw error("OBJ: line " + std::to_string(lineNo) + ": a face corner without a vertex");
                const auto keyOf = std::make_tuple(idx[0], idx[1], idx[2]);
                auto it = part.seen.find(keyOf);
                if (it == part.seen.end())
                {
                    const uint32_t id = static_cast<uint32_t>(part.position.size() / 3);
                    for (int c = 0; c < 3; c++)
                        part.position.push_back(static_cast<float>(v[static_cast<size_t>(idx[0] - 1) * 3 + static_cast<size_t>(c)]));
                    for (int c = 0; c < 2; c++)
                        part.uv.push_back(idx[1] ? static_cast<float>(vt[static_cast<size_t>(idx[1] - 1) * 2 + static_cast<size_t>(c)]) : 0.0f);
                    for (int c = 0; c < 3; c++)
                        part.normal.push_back(idx[2] ? static_cast<float>(vn[static_cast<size_t>(idx[2] - 1) * 3 + static_cast<size_t>(c)]) : 0.0f);
                    part.anyUv = part.anyUv || idx[1] != 0;
                    part.anyNormal = part.anyNormal || idx[2] != 0;
                    it = part.seen.emplace(keyOf, id).first;
                }
                polygon.push_back(it->second);
            }
            for (size_t c = 2; c < polygon.size(); c++) // aiProcess_Triangulate on a convex polygon: a fan
            {
                part.index.push_back(polygon[0]);
                part.index.push_back(polygon[c - 1]);
                part.index.push_back(polygon[c]);
            }
        }
    }

    json = Obj();
    json.object["asset"] = Obj();
    json.object["asset"].object["version"] = Str("2.0");
    json.object["asset"].object["generator"] = Str("ObjReader");
    Json views = Arr(), accessors = Arr(), meshes = Arr(), nodes = Arr(), roots = Arr(), mats = Arr(), textures = Arr(), images = Arr();
    auto add = [&](const void *data, size_t byteLength, size_t count, int componentType, const char *type) {
        while (buffer.size() % 4)
            buffer.push_back(0);
        Json view = Obj();
        view.object["buffer"] = Num(0);
        view.object["byteOffset"] = Num(static_cast<double>(buffer.size()));
        view.object["byteLength"] = Num(static_cast<double>(byteLength));
        buffer.insert(buffer.end(), static_cast<const uint8_t *>(data), static_cast<const uint8_t *>(data) + byteLength);
        views.array.push_back(std::move(view));
        Json acc = Obj();
        acc.object["bufferView"] = Num(static_cast<double>(views.array.size() - 1));
        acc.object["componentType"] = Num(componentType);
        acc.object["count"] = Num(static_cast<double>(count));
        acc.object["type"] = Str(type);
        accessors.array.push_back(std::move(acc));
        return Num(static_cast<double>(accessors.array.size() - 1));
    };
    for (const MtlMaterial &m : materials)
    {
        Json out = Obj(), assimp = Obj(), slots = Obj();
        out.object["name"] = Str(m.name);
        const double rgba[4] = { m.kd[0], m.kd[1], m.kd[2], m.d };
        assimp.object["diffuse"] = Nums(rgba, 4);
        assimp.object["shininess"] = Num(m.ns); // always present: every OBJ material is "Phong" to ChooseMaterialType
        if (m.hasKe)
            out.object["emissiveFactor"] = Nums(m.ke, 3);
        for (const auto &[slot, fileName] : m.maps)
        {
            Json image = Obj();
            image.object["uri"] = Str(fileName);
            images.array.push_back(std::move(image));
            Json tex = Obj();
            tex.object["source"] = Num(static_cast<double>(images.array.size() - 1));
            textures.array.push_back(std::move(tex));
            Json ref = Obj();
            ref.object["index"] = Num(static_cast<double>(textures.array.size() - 1));
            slots.object[slot] = std::move(ref);
        }
        assimp.object["textures"] = std::move(slots);
        out.object["extras"] = Obj();
        out.object["extras"].object["assimp"] = std::move(assimp);
        mats.array.push_back(std::move(out));
    }
    for (const Part &part : parts)
    {
        if (part.index.empty())
            continue;
        Json attributes = Obj();
        attributes.object["POSITION"] = add(part.position.data(), part.position.size() * 4, part.position.size() / 3, 5126, "VEC3");
        if (part.anyNormal)
            attributes.object["NORMAL"] = add(part.normal.data(), part.normal.size() * 4, part.normal.size() / 3, 5126, "VEC3");
        if (part.anyUv)
            attributes.object["TEXCOORD_0"] = add(part.uv.data(), part.uv.size() * 4, part.uv.size() / 2, 5126, "VEC2");
        Json prim = Obj();
        prim.object["attributes"] = std::move(attributes);
        prim.object["indices"] = add(part.index.data(), part.index.size() * 4, part.index.size(), 5125, "SCALAR");
        if (part.material >= 0)
            prim.object["material"] = Num(static_cast<double>(part.material));
        Json mesh = Obj();
        mesh.object["name"] = Str(part.name);
        mesh.object["primitives"] = Arr();
        mesh.object["primitives"].array.push_back(std::move(prim));
        meshes.array.push_back(std::move(mesh));
        Json node = Obj();
        node.object["name"] = Str(part.name);
        node.object["mesh"] = Num(static_cast<double>(meshes.array.size() - 1));
        nodes.array.push_back(std::move(node));
        roots.array.push_back(Num(static_cast<double>(nodes.array.size() - 1)));
    }
    if (meshes.array.empty())
        throw error("OBJ: no faces");
    Json scene = Obj();
    scene.object["nodes"] = std::move(roots);
    json.object["scene"] = Num(0);
    json.object["scenes"] = Arr();
    json.object["scenes"].array.push_back(std::move(scene));
    json.object["nodes"] = std::move(nodes);
    json.object["meshes"] = std::move(meshes);
    json.object["materials"] = std::move(mats);
    json.object["textures"] = std::move(textures);
    json.object["images"] = std::move(images);
    json.object["accessors"] = std::move(accessors);
    json.object["bufferViews"] = std::move(views);
    Json buf = Obj();
    buf.object["byteLength"] = Num(static_cast<double>(buffer.size()));
    json.object["buffers"] = Arr();
    json.object["buffers"].array.push_back(std::move(buf));
}

}
