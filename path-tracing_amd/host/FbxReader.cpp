#include "FbxReader.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <variant>

#include "Math.h"
#include "TextureImporter.h"

namespace PathTracing
{

namespace
{

// ---------------------------------------------------------------------------------------------------------
// the record tree
// ---------------------------------------------------------------------------------------------------------

struct FbxProperty
{
    char type = 0;               // Y C I F D L (scalars), f d l i b (arrays), S R (bytes)
    int64_t integer = 0;         // Y C I L
    double real = 0.0;           // F D
    std::vector<double> reals;   // f d
    std::vector<int64_t> ints;   // l i b
    std::string bytes;           // S R
    [[nodiscard]] double AsDouble() const { return (type == 'F' || type == 'D') ? real : static_cast<double>(integer); }
};

struct FbxNode
{
    std::string name;
    std::vector<FbxProperty> props;
    std::vector<FbxNode> children;
    [[nodiscard]] const FbxNode *Child(const char *n) const
    {
        for (const FbxNode &c : children)
            if (c.name == n)
                return &c;
        return nullptr;
    }
};

class Reader
{
public:
    Reader(std::span<const uint8_t> f, bool wide) : m_File(f), m_Wide(wide) {}

    // children of one list, up to `end` (a null record closes a nested list)
    void ReadList(size_t &pos, size_t end, std::vector<FbxNode> &out, int depth)
    {
        if (depth > 64)
            throw error("FBX: records nested too deeply");
        const size_t header = m_Wide ? 25 : 13;
        while (pos + header <= end)
        {
            const uint64_t endOffset = Offset(pos), numProps = Offset(pos + (m_Wide ? 8 : 4)), propBytes = Offset(pos + (m_Wide ? 16 : 8));
            const uint8_t nameLen = m_File[pos + header - 1];
            if (endOffset == 0) // null record
            {
                pos += header;
                return;
            }
            if (endOffset > end || endOffset < pos + header + nameLen || propBytes > endOffset - (pos + header + nameLen))
                throw error("FBX: record runs past its parent");
            FbxNode node;
            node.name.assign(reinterpret_cast<const char *>(&m_File[pos + header]), nameLen);
            size_t p = pos + header + nameLen;
            const size_t propEnd = p + static_cast<size_t>(propBytes);
            if (numProps > propBytes) // every property takes at least one byte
                throw error("FBX: property count exceeds the property list");
            node.props.reserve(static_cast<size_t>(numProps));
            for (uint64_t k = 0; k < numProps; k++)
                node.props.push_back(ReadProperty(p, propEnd));
            p = propEnd;
            if (p < endOffset)
                ReadList(p, static_cast<size_t>(endOffset), node.children, depth + 1);
            pos = static_cast<size_t>(endOffset);
            out.push_back(std::move(node));
        }
    }

private:
    std::span<const uint8_t> m_File;
    bool m_Wide;

    template <typename T> T Get(size_t pos, size_t end) const
    {
        if (pos > end || sizeof(T) > end - pos)
            throw error("FBX: truncated property");
        T v;
        std::memcpy(&v, &m_File[pos], sizeof(T));
        return v;
    }
    uint64_t Offset(size_t pos) const
    {
        if (m_Wide)
            return Get<uint64_t>(pos, m_File.size());
        return Get<uint32_t>(pos, m_File.size());
    }

    FbxProperty ReadProperty(size_t &p, size_t end) const
    {
        FbxProperty out;
        out.type = static_cast<char>(Get<uint8_t>(p, end));
        p += 1;
        switch (out.type)
        {
        case 'Y': out.integer = Get<int16_t>(p, end); p += 2; break;
        case 'C': out.integer = Get<uint8_t>(p, end) & 1; p += 1; break;
        case 'I': out.integer = Get<int32_t>(p, end); p += 4; break;
        case 'L': out.integer = Get<int64_t>(p, end); p += 8; break;
        case 'F': out.real = Get<float>(p, end); p += 4; break;
        case 'D': out.real = Get<double>(p, end); p += 8; break;
        case 'S':
        case 'R':
        {
            const uint32_t len = Get<uint32_t>(p, end);
            p += 4;
            if (len > end - p)
                throw error("FBX: truncated string property");
            out.bytes.assign(reinterpret_cast<const char *>(&m_File[p]), len);
            p += len;
            break;
        }
        case 'f': case 'd': case 'l': case 'i': case 'b':
        {
            const uint32_t count = Get<uint32_t>(p, end), encoding = Get<uint32_t>(p + 4, end), stored = Get<uint32_t>(p + 8, end);
            p += 12;
            if (stored > end - p)
                throw error("FBX: truncated array property");
            const size_t elem = out.type == 'f' || out.type == 'i' ? 4 : out.type == 'b' ? 1 : 8;
            std::vector<uint8_t> inflated;
            std::span<const uint8_t> raw(&m_File[p], stored);
            if (encoding == 1)
            {
                inflated = TextureImporter::Inflate(raw);
                raw = inflated;
            }
            else if (encoding != 0)
                throw error("FBX: unknown array encoding");
            if (raw.size() != static_cast<size_t>(count) * elem)
                throw error("FBX: array length does not match its data");
            if (out.type == 'f' || out.type == 'd')
            {
                out.reals.resize(count);
                for (uint32_t k = 0; k < count; k++)
                    if (out.type == 'f') { float v; std::memcpy(&v, &raw[k * 4], 4); out.reals[k] = v; }
                    else std::memcpy(&out.reals[k], &raw[static_cast<size_t>(k) * 8], 8);
            }
            else
            {
                out.ints.resize(count);
                for (uint32_t k = 0; k < count; k++)
                    if (out.type == 'i') { int32_t v; std::memcpy(&v, &raw[k * 4], 4); out.ints[k] = v; }
                    else if (out.type == 'l') std::memcpy(&out.ints[k], &raw[static_cast<size_t>(k) * 8], 8);
                    else out.ints[k] = raw[k];
            }
            p += stored;
            break;
        }
        default:
            throw error("FBX: unknown property type");
        }
        return out;
    }
};

// ---------------------------------------------------------------------------------------------------------
// helpers on the tree
// ---------------------------------------------------------------------------------------------------------

// Properties70 / P: name, type, label, flags, values ...
struct Props70
{
    std::map<std::string, const FbxNode *> byName;
    explicit Props70(const FbxNode &object)
    {
        if (const FbxNode *p70 = object.Child("Properties70"))
            for (const FbxNode &p : p70->children)
                if (p.name == "P" && !p.props.empty() && (p.props[0].type == 'S'))
                    byName[p.props[0].bytes] = &p;
    }
    [[nodiscard]] bool Has(const std::string &n) const { return byName.count(n) != 0; }
    [[nodiscard]] double Number(const std::string &n, double fallback) const
    {
        const auto it = byName.find(n);
        return it != byName.end() && it->second->props.size() > 4 ? it->second->props[4].AsDouble() : fallback;
    }
    bool Vector(const std::string &n, double out[3]) const
    {
        const auto it = byName.find(n);
        if (it == byName.end() || it->second->props.size() < 7)
            return false;
        for (int k = 0; k < 3; k++)
            out[k] = it->second->props[4 + static_cast<size_t>(k)].AsDouble();
        return true;
    }
};

std::string ObjectName(const FbxNode &object) // "name\0\1Class" -> name
{
    if (object.props.size() < 2 || object.props[1].type != 'S')
        return {};
    const std::string &s = object.props[1].bytes;
    const size_t cut = s.find(std::string("\0\1", 2));
    return cut == std::string::npos ? s : s.substr(0, cut);
}

const std::vector<double> *RealArray(const FbxNode *n)
{
    return n && !n->props.empty() && (n->props[0].type == 'd' || n->props[0].type == 'f') ? &n->props[0].reals : nullptr;
}
const std::vector<int64_t> *IntArray(const FbxNode *n)
{
    return n && !n->props.empty() && (n->props[0].type == 'i' || n->props[0].type == 'l') ? &n->props[0].ints : nullptr;
}
std::string StringOf(const FbxNode *n) { return n && !n->props.empty() && n->props[0].type == 'S' ? n->props[0].bytes : std::string(); }

struct DMat
{
    double m[4][4];
    static DMat Identity()
    {
        DMat r;
        std::memset(r.m, 0, sizeof(r.m));
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0;
        return r;
    }
    [[nodiscard]] bool IsIdentity() const
    {
        const DMat i = Identity();
        return std::memcmp(m, i.m, sizeof(m)) == 0;
    }
};
DMat operator*(const DMat &a, const DMat &b)
{
    DMat r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
DMat Translation(const double v[3])
{
    DMat r = DMat::Identity();
    r.m[0][3] = v[0]; r.m[1][3] = v[1]; r.m[2][3] = v[2];
    return r;
}
DMat Scaling(const double v[3])
{
    DMat r = DMat::Identity();
    r.m[0][0] = v[0]; r.m[1][1] = v[1]; r.m[2][2] = v[2];
    return r;
}
DMat EulerXyz(const double degrees[3]) // FBX eEulerXYZ: X first, then Y, then Z = Rz * Ry * Rx
{
    const double k = 3.14159265358979323846 / 180.0;
    const double cx = std::cos(degrees[0] * k), sx = std::sin(degrees[0] * k), cy = std::cos(degrees[1] * k), sy = std::sin(degrees[1] * k),
                 cz = std::cos(degrees[2] * k), sz = std::sin(degrees[2] * k);
    DMat rx = DMat::Identity(), ry = DMat::Identity(), rz = DMat::Identity();
    rx.m[1][1] = cx; rx.m[1][2] = -sx; rx.m[2][1] = sx; rx.m[2][2] = cx;
    ry.m[0][0] = cy; ry.m[0][2] = sy; ry.m[2][0] = -sy; ry.m[2][2] = cy;
    rz.m[0][0] = cz; rz.m[0][1] = -sz; rz.m[1][0] = sz; rz.m[1][1] = cz;
    return rz * ry * rx;
}
DMat TransposeRotation(const DMat &r) // inverse of a rotation
{
    DMat t = DMat::Identity();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t.m[i][j] = r.m[j][i];
    return t;
}

DMat InverseAffine(const DMat &a) // inverse of [ L t ; 0 1 ] by the cofactors of L
{
    const double (*m)[4] = a.m;
    const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2], c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    const double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    if (det == 0.0 || !std::isfinite(det))
        throw error("FBX: a cluster's bind pose is not invertible");
    DMat r = DMat::Identity();
    r.m[0][0] = c00 / det; r.m[1][0] = c01 / det; r.m[2][0] = c02 / det;
    r.m[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det;
    r.m[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det;
    r.m[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det;
    r.m[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
    r.m[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
    r.m[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
    for (int i = 0; i < 3; i++)
        r.m[i][3] = -(r.m[i][0] * m[0][3] + r.m[i][1] * m[1][3] + r.m[i][2] * m[2][3]);
    return r;
}

// FBX SDK: WorldTransform = ParentWorld * T * Roff * Rp * Rpre * R * Rpost^-1 * Rp^-1 * Soff * Sp * S * Sp^-1
DMat LocalTransform(const Props70 &p)
{
    double t[3] = { 0, 0, 0 }, r[3] = { 0, 0, 0 }, s[3] = { 1, 1, 1 }, pre[3] = { 0, 0, 0 }, post[3] = { 0, 0, 0 }, rp[3] = { 0, 0, 0 }, sp[3] = { 0, 0, 0 },
           roff[3] = { 0, 0, 0 }, soff[3] = { 0, 0, 0 };
    p.Vector("Lcl Translation", t);
    p.Vector("Lcl Rotation", r);
    p.Vector("Lcl Scaling", s);
    p.Vector("PreRotation", pre);
    p.Vector("PostRotation", post);
    p.Vector("RotationPivot", rp);
    p.Vector("ScalingPivot", sp);
    p.Vector("RotationOffset", roff);
    p.Vector("ScalingOffset", soff);
    if (p.Number("RotationOrder", 0.0) != 0.0)
        throw error("FBX: only the XYZ rotation order is supported");
    const double nrp[3] = { -rp[0], -rp[1], -rp[2] }, nsp[3] = { -sp[0], -sp[1], -sp[2] };
    return Translation(t) * Translation(roff) * Translation(rp) * EulerXyz(pre) * EulerXyz(r) * TransposeRotation(EulerXyz(post)) * Translation(nrp) *
           Translation(soff) * Translation(sp) * Scaling(s) * Translation(nsp);
}
DMat GeometricTransform(const Props70 &p)
{
    double t[3] = { 0, 0, 0 }, r[3] = { 0, 0, 0 }, s[3] = { 1, 1, 1 };
    p.Vector("GeometricTranslation", t);
    p.Vector("GeometricRotation", r);
    p.Vector("GeometricScaling", s);
    return Translation(t) * EulerXyz(r) * Scaling(s);
}

// ---------------------------------------------------------------------------------------------------------
// JSON building
// ---------------------------------------------------------------------------------------------------------

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

struct BufferWriter
{
    std::vector<uint8_t> &bytes;
    Json views = Arr(), accessors = Arr();
    int64_t Add(const void *data, size_t byteLength, size_t count, int componentType, const char *type)
    {
        while (bytes.size() % 4)
            bytes.push_back(0);
        Json view = Obj();
        view.object["buffer"] = Num(0);
        view.object["byteOffset"] = Num(static_cast<double>(bytes.size()));
        view.object["byteLength"] = Num(static_cast<double>(byteLength));
        bytes.insert(bytes.end(), static_cast<const uint8_t *>(data), static_cast<const uint8_t *>(data) + byteLength);
        views.array.push_back(std::move(view));
        Json acc = Obj();
        acc.object["bufferView"] = Num(static_cast<double>(views.array.size() - 1));
        acc.object["componentType"] = Num(componentType);
        acc.object["count"] = Num(static_cast<double>(count));
        acc.object["type"] = Str(type);
        accessors.array.push_back(std::move(acc));
        return static_cast<int64_t>(accessors.array.size() - 1);
    }
};

// one layer element (normals / UVs) resolved per polygon vertex
struct Layer
{
    const std::vector<double> *values = nullptr;
    const std::vector<int64_t> *index = nullptr;
    bool byPolygonVertex = true;
    int width = 3;
    [[nodiscard]] bool Fetch(size_t polygonVertex, int64_t controlPoint, float *out) const
    {
        if (!values)
            return false;
        int64_t i = byPolygonVertex ? static_cast<int64_t>(polygonVertex) : controlPoint;
        if (index)
        {
            if (i < 0 || static_cast<size_t>(i) >= index->size())
                return false;
            i = (*index)[static_cast<size_t>(i)];
        }
        if (i < 0 || (static_cast<size_t>(i) + 1) * static_cast<size_t>(width) > values->size())
            return false;
        for (int k = 0; k < width; k++)
            out[k] = static_cast<float>((*values)[static_cast<size_t>(i) * static_cast<size_t>(width) + static_cast<size_t>(k)]);
        return true;
    }
};

Layer ReadLayer(const FbxNode &geometry, const char *element, const char *valuesName, const char *indexName, int width)
{
    Layer l;
    l.width = width;
    const FbxNode *e = geometry.Child(element);
    if (!e)
        return l;
    const std::string mapping = StringOf(e->Child("MappingInformationType")), reference = StringOf(e->Child("ReferenceInformationType"));
    l.values = RealArray(e->Child(valuesName));
    if (mapping == "ByPolygonVertex")
        l.byPolygonVertex = true;
    else if (mapping == "ByVertice" || mapping == "ByVertex" || mapping == "ByControlPoint")
        l.byPolygonVertex = false;
    else
        l.values = nullptr; // ByPolygon / AllSame normals are not used by the reference's scenes
    if (reference == "IndexToDirect" || reference == "Index")
        l.index = IntArray(e->Child(indexName));
    return l;
}

}

bool IsBinaryFbx(std::span<const uint8_t> file)
{
    return file.size() >= 27 && std::memcmp(file.data(), "Kaydara FBX Binary  \0\x1a\0", 23) == 0;
}

void ConvertFbxToGltf(std::span<const uint8_t> file, Json &json, std::vector<uint8_t> &buffer)
{
    if (!IsBinaryFbx(file))
        throw error("FBX: not a binary FBX file (ASCII FBX is not supported)");
    uint32_t version;
    std::memcpy(&version, &file[23], 4);
    std::vector<FbxNode> top;
    {
        Reader reader(file, version >= 7500);
        size_t pos = 27;
        reader.ReadList(pos, file.size(), top, 0);
    }
    const FbxNode *objects = nullptr, *connections = nullptr;
    for (const FbxNode &n : top)
    {
        if (n.name == "Objects") objects = &n;
        if (n.name == "Connections") connections = &n;
    }
    if (!objects || !connections)
        throw error("FBX: no Objects / Connections section");

    // ---- objects by id, connections in file order
    std::unordered_map<int64_t, const FbxNode *> byId;
    std::vector<int64_t> models;
    for (const FbxNode &o : objects->children)
        if (!o.props.empty() && (o.props[0].type == 'L' || o.props[0].type == 'I'))
        {
            byId[o.props[0].integer] = &o;
            if (o.name == "Model")
                models.push_back(o.props[0].integer);
        }
    struct Link { int64_t child, parent; std::string property; };
    std::vector<Link> links;
    for (const FbxNode &c : connections->children)
        if (c.name == "C" && c.props.size() >= 3 && c.props[0].type == 'S')
            links.push_back({ c.props[1].integer, c.props[2].integer, c.props.size() > 3 && c.props[3].type == 'S' ? c.props[3].bytes : std::string() });
    auto childrenOf = [&](int64_t parent, const char *kind) {
        std::vector<const Link *> out;
        for (const Link &l : links)
            if (l.parent == parent)
            {
                const auto it = byId.find(l.child);
                if (it != byId.end() && it->second->name == kind)
                    out.push_back(&l);
            }
        return out;
    };

    json = Obj();
    json.object["asset"] = Obj();
    json.object["asset"].object["version"] = Str("2.0");
    json.object["asset"].object["generator"] = Str("FbxReader (binary FBX " + std::to_string(version) + ")");
    BufferWriter writer { buffer };
    Json nodes = Arr(), meshes = Arr(), materials = Arr(), textures = Arr(), images = Arr(), lights = Arr(), cameras = Arr(), skins = Arr();

    // ---- textures / materials (aiMaterial of assimp's FBX converter)
    std::unordered_map<int64_t, int64_t> textureIndexOf, materialIndexOf;
    auto textureIndex = [&](int64_t id) -> int64_t {
        const auto known = textureIndexOf.find(id);
        if (known != textureIndexOf.end())
            return known->second;
        const FbxNode *t = byId.at(id);
        std::string name = StringOf(t->Child("RelativeFilename"));
        if (name.empty())
            name = StringOf(t->Child("FileName"));
        std::replace(name.begin(), name.end(), '\\', '/');
        Json image = Obj();
        image.object["uri"] = Str(name);
        images.array.push_back(std::move(image));
        Json tex = Obj();
        tex.object["source"] = Num(static_cast<double>(images.array.size() - 1));
        textures.array.push_back(std::move(tex));
        return textureIndexOf[id] = static_cast<int64_t>(textures.array.size() - 1);
    };
    auto materialIndex = [&](int64_t id) -> int64_t {
        const auto known = materialIndexOf.find(id);
        if (known != materialIndexOf.end())
            return known->second;
        const FbxNode *m = byId.at(id);
        const Props70 p(*m);
        Json out = Obj(), assimp = Obj(), slots = Obj();
        out.object["name"] = Str(ObjectName(*m));
        double c[3];
        if (p.Vector("DiffuseColor", c) || p.Vector("Diffuse", c))
        {
            const double f = p.Number("DiffuseFactor", 1.0);
            const double rgba[4] = { c[0] * f, c[1] * f, c[2] * f, 1.0 };
            assimp.object["diffuse"] = Nums(rgba, 4);
        }
        if (p.Vector("EmissiveColor", c) || p.Vector("Emissive", c))
        {
            const double f = p.Number("EmissiveFactor", 1.0);
            const double rgb[3] = { c[0] * f, c[1] * f, c[2] * f };
            out.object["emissiveFactor"] = Nums(rgb, 3);
        }
        if (p.Has("ShininessExponent") || p.Has("Shininess"))
            assimp.object["shininess"] = Num(p.Number("ShininessExponent", p.Number("Shininess", 0.0)));
        if (p.Has("SpecularFactor"))
            assimp.object["specularFactor"] = Num(p.Number("SpecularFactor", 1.0));
        for (const Link *l : childrenOf(id, "Texture"))
        {
            const char *slot = l->property == "DiffuseColor" || l->property == "Diffuse" ? "DIFFUSE"
                               : l->property == "NormalMap" || l->property == "Bump" ? "NORMALS"
                               : l->property == "SpecularColor" || l->property == "Specular" ? "SPECULAR"
                               : l->property == "ShininessExponent" || l->property == "Shininess" ? "SHININESS"
                               : l->property == "EmissiveColor" || l->property == "Emissive" ? "EMISSIVE" : nullptr;
            if (slot && !slots.Has(slot))
            {
                Json ref = Obj();
                ref.object["index"] = Num(static_cast<double>(textureIndex(l->child)));
                slots.object[slot] = std::move(ref);
            }
        }
        assimp.object["textures"] = std::move(slots);
        out.object["extras"] = Obj();
        out.object["extras"].object["assimp"] = std::move(assimp);
        materials.array.push_back(std::move(out));
        return materialIndexOf[id] = static_cast<int64_t>(materials.array.size() - 1);
    };

    // ---- geometry: one mesh per (Geometry, material list of the model that uses it)
    // per control point: up to four (joint, weight) pairs, largest first, weights renormalised (aiProcess_LimitBoneWeights)
    struct Influence { uint16_t joint[4] = { 0, 0, 0, 0 }; float weight[4] = { 0, 0, 0, 0 }; };
    auto buildMesh = [&](const FbxNode &geometry, const std::vector<int64_t> &materialOfSlot, const std::vector<Influence> *skin) -> int64_t {
        const std::vector<double> *points = RealArray(geometry.Child("Vertices"));
        const std::vector<int64_t> *polygons = IntArray(geometry.Child("PolygonVertexIndex"));
        if (!points || !polygons || points->size() < 9)
            return -1;
        const size_t controlPoints = points->size() / 3;
        const Layer normals = ReadLayer(geometry, "LayerElementNormal", "Normals", "NormalsIndex", 3);
        const Layer uvs = ReadLayer(geometry, "LayerElementUV", "UV", "UVIndex", 2);
        const std::vector<int64_t> *polygonMaterial = nullptr;
        bool materialByPolygon = false;
        if (const FbxNode *e = geometry.Child("LayerElementMaterial"))
        {
            polygonMaterial = IntArray(e->Child("Materials"));
            materialByPolygon = StringOf(e->Child("MappingInformationType")) == "ByPolygon";
        }
        struct Part
        {
            std::vector<float> position, normal, uv;
            std::vector<uint32_t> index, controlPoint; // controlPoint: per vertex, for the skin weights
            std::map<std::tuple<int64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, uint32_t> seen;
        };
        std::map<int64_t, Part> parts; // by material slot
        const bool haveNormals = normals.values != nullptr, haveUvs = uvs.values != nullptr;
        size_t polygon = 0, first = 0;
        for (size_t k = 0; k < polygons->size(); k++)
        {
            if ((*polygons)[k] >= 0)
                continue;
            // polygon [first, k]: the last index is stored as ~index
            int64_t slot = 0;
            if (polygonMaterial && !polygonMaterial->empty())
                slot = materialByPolygon ? (polygon < polygonMaterial->size() ? (*polygonMaterial)[polygon] : 0) : (*polygonMaterial)[0];
            Part &part = parts[slot];
            std::vector<uint32_t> corner;
            for (size_t v = first; v <= k; v++)
            {
                const int64_t cp = v == k ? ~(*polygons)[v] : (*polygons)[v];
                if (cp < 0 || static_cast<size_t>(cp) >= controlPoints)
                    throw error("FBX: polygon refers to a control point that does not exist");
                float n[3] = { 0, 0, 0 }, t[2] = { 0, 0 };
                const bool hn = haveNormals && normals.Fetch(v, cp, n), ht = haveUvs && uvs.Fetch(v, cp, t);
                uint32_t bits[5] = { 0, 0, 0, 0, 0 };
                if (hn) std::memcpy(bits, n, 12);
                if (ht) std::memcpy(bits + 3, t, 8);
                const auto key = std::make_tuple(cp, bits[0], bits[1], bits[2], bits[3], bits[4]);
                auto it = part.seen.find(key);
                if (it == part.seen.end())
                {
                    const uint32_t id = static_cast<uint32_t>(part.position.size() / 3);
                    for (int c = 0; c < 3; c++)
                        part.position.push_back(static_cast<float>((*points)[static_cast<size_t>(cp) * 3 + static_cast<size_t>(c)]));
                    part.normal.insert(part.normal.end(), n, n + 3);
                    part.uv.insert(part.uv.end(), t, t + 2);
                    part.controlPoint.push_back(static_cast<uint32_t>(cp));
                    it = part.seen.emplace(key, id).first;
                }
                corner.push_back(it->second);
            }
            for (size_t c = 2; c < corner.size(); c++) // aiProcess_Triangulate on a convex polygon: a fan
            {
                part.index.push_back(corner[0]);
                part.index.push_back(corner[c - 1]);
                part.index.push_back(corner[c]);
            }
            first = k + 1;
            polygon++;
        }
        Json primitives = Arr();
        for (auto &[slot, part] : parts)
        {
            if (part.index.empty())
                continue;
            Json attributes = Obj();
            attributes.object["POSITION"] = Num(static_cast<double>(writer.Add(part.position.data(), part.position.size() * 4, part.position.size() / 3, 5126, "VEC3")));
            if (haveNormals)
                attributes.object["NORMAL"] = Num(static_cast<double>(writer.Add(part.normal.data(), part.normal.size() * 4, part.normal.size() / 3, 5126, "VEC3")));
            if (haveUvs)
                attributes.object["TEXCOORD_0"] = Num(static_cast<double>(writer.Add(part.uv.data(), part.uv.size() * 4, part.uv.size() / 2, 5126, "VEC2")));
            if (skin)
            {
                std::vector<uint16_t> joints;
                std::vector<float> weights;
                for (uint32_t cp : part.controlPoint)
                {
                    const Influence &in = cp < skin->size() ? (*skin)[cp] : Influence();
                    joints.insert(joints.end(), in.joint, in.joint + 4);
                    weights.insert(weights.end(), in.weight, in.weight + 4);
                }
                attributes.object["JOINTS_0"] = Num(static_cast<double>(writer.Add(joints.data(), joints.size() * 2, joints.size() / 4, 5123, "VEC4")));
                attributes.object["WEIGHTS_0"] = Num(static_cast<double>(writer.Add(weights.data(), weights.size() * 4, weights.size() / 4, 5126, "VEC4")));
            }
            Json prim = Obj();
            prim.object["attributes"] = std::move(attributes);
            prim.object["indices"] = Num(static_cast<double>(writer.Add(part.index.data(), part.index.size() * 4, part.index.size(), 5125, "SCALAR")));
            if (slot >= 0 && static_cast<size_t>(slot) < materialOfSlot.size())
                prim.object["material"] = Num(static_cast<double>(materialOfSlot[static_cast<size_t>(slot)]));
            primitives.array.push_back(std::move(prim));
        }
        if (primitives.array.empty())
            return -1;
        Json mesh = Obj();
        mesh.object["name"] = Str(ObjectName(geometry));
        mesh.object["primitives"] = std::move(primitives);
        meshes.array.push_back(std::move(mesh));
        return static_cast<int64_t>(meshes.array.size() - 1);
    };

    // ---- models -> nodes
    auto matrixJson = [](const DMat &m) {
        double col[16];
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++)
                col[c * 4 + r] = m.m[r][c];
        return Nums(col, 16);
    };
    std::unordered_map<int64_t, size_t> nodeOfModel;
    for (int64_t id : models)
    {
        const FbxNode *model = byId.at(id);
        const Props70 p(*model);
        Json node = Obj();
        node.object["name"] = Str(ObjectName(*model));
        node.object["matrix"] = matrixJson(LocalTransform(p));
        node.object["children"] = Arr();
        nodeOfModel[id] = nodes.array.size();
        nodes.array.push_back(std::move(node));
    }
    Json roots = Arr();
    for (int64_t id : models)
    {
        int64_t parent = 0;
        for (const Link &l : links)
            if (l.child == id && l.property.empty())
            {
                const auto it = byId.find(l.parent);
                if (l.parent == 0 || (it != byId.end() && it->second->name == "Model"))
                {
                    parent = l.parent;
                    break;
                }
            }
        const auto parentNode = nodeOfModel.find(parent);
        if (parent != 0 && parentNode != nodeOfModel.end() && parent != id)
            nodes.array[parentNode->second].object["children"].array.push_back(Num(static_cast<double>(nodeOfModel[id])));
        else
            roots.array.push_back(Num(static_cast<double>(nodeOfModel[id])));
    }
    std::unordered_map<int64_t, int64_t> parentModel;
    for (int64_t id : models)
        for (const Link &l : links)
            if (l.child == id && l.property.empty() && l.parent != id && nodeOfModel.count(l.parent))
            {
                parentModel[id] = l.parent;
                break;
            }
    auto globalOf = [&](int64_t id) {
        DMat g = DMat::Identity();
        int guard = 0;
        for (int64_t at = id; guard < 4096; guard++)
        {
            g = LocalTransform(Props70(*byId.at(at))) * g;
            const auto up = parentModel.find(at);
            if (up == parentModel.end())
                break;
            at = up->second;
        }
        return g;
    };
    for (int64_t id : models)
    {
        const FbxNode *model = byId.at(id);
        const Props70 p(*model);
        // what hangs below the model: its geometry (with the model's materials, in connection order) or a light
        std::vector<int64_t> materialOfSlot;
        for (const Link *l : childrenOf(id, "Material"))
            materialOfSlot.push_back(materialIndex(l->child));
        Json attach = Obj(); // mesh / light go onto the node itself, or onto a child when a geometric transform applies
        for (const Link *l : childrenOf(id, "Geometry"))
        {
            // a Skin deformer on the geometry: Cluster sub-deformers, each with the control points it moves, their weights,
            // the bone Model it follows and that bone's bind pose (TransformLink).  assimp: offset matrix =
            // TransformLink^-1 * the mesh node's global transform (FBXConverter::ConvertCluster).
            std::vector<Influence> influences;
            Json joints = Arr();
            std::vector<float> inverseBind;
            for (const Link *skinLink : childrenOf(l->child, "Deformer"))
                for (const Link *clusterLink : childrenOf(skinLink->child, "Deformer"))
                {
                    const FbxNode *cluster = byId.at(clusterLink->child);
                    const std::vector<int64_t> *indexes = IntArray(cluster->Child("Indexes"));
                    const std::vector<double> *weights = RealArray(cluster->Child("Weights")), *bindPose = RealArray(cluster->Child("TransformLink"));
                    const std::vector<const Link *> bones = childrenOf(clusterLink->child, "Model");
                    if (bones.empty() || !nodeOfModel.count(bones[0]->child) || !bindPose || bindPose->size() != 16)
                        continue;
                    DMat link;
                    for (int c = 0; c < 4; c++)
                        for (int r = 0; r < 4; r++)
                            link.m[r][c] = (*bindPose)[static_cast<size_t>(c * 4 + r)];
                    const DMat offset = InverseAffine(link) * globalOf(id);
                    for (int c = 0; c < 4; c++)
                        for (int r = 0; r < 4; r++)
                            inverseBind.push_back(static_cast<float>(offset.m[r][c]));
                    const uint16_t joint = static_cast<uint16_t>(joints.array.size());
                    joints.array.push_back(Num(static_cast<double>(nodeOfModel[bones[0]->child])));
                    if (!indexes || !weights)
                        continue;
                    for (size_t k = 0; k < indexes->size() && k < weights->size(); k++)
                    {
                        const int64_t cp = (*indexes)[k];
                        const float w = static_cast<float>((*weights)[k]);
                        if (cp < 0 || cp > (1 << 28) || !(w > 0.0f))
                            continue;
                        if (static_cast<size_t>(cp) >= influences.size())
                            influences.resize(static_cast<size_t>(cp) + 1);
                        Influence &in = influences[static_cast<size_t>(cp)];
                        int slot = 3; // keep the four largest, largest first
                        if (w <= in.weight[3])
                            continue;
                        while (slot > 0 && w > in.weight[slot - 1])
                        {
                            in.weight[slot] = in.weight[slot - 1];
                            in.joint[slot] = in.joint[slot - 1];
                            slot--;
                        }
                        in.weight[slot] = w;
                        in.joint[slot] = joint;
                    }
                }
            for (Influence &in : influences)
            {
                const float sum = in.weight[0] + in.weight[1] + in.weight[2] + in.weight[3];
                if (sum > 0.0f)
                    for (float &w : in.weight)
                        w /= sum;
            }
            const bool skinned = !joints.array.empty();
            const int64_t mesh = buildMesh(*byId.at(l->child), materialOfSlot, skinned ? &influences : nullptr);
            if (mesh >= 0 && !attach.Has("mesh"))
            {
                attach.object["mesh"] = Num(static_cast<double>(mesh));
                if (skinned)
                {
                    Json skin = Obj();
                    skin.object["joints"] = std::move(joints);
                    skin.object["inverseBindMatrices"] = Num(static_cast<double>(writer.Add(inverseBind.data(), inverseBind.size() * 4, inverseBind.size() / 16, 5126, "MAT4")));
                    skins.array.push_back(std::move(skin));
                    attach.object["skin"] = Num(static_cast<double>(skins.array.size() - 1));
                }
            }
        }
        for (const Link *l : childrenOf(id, "NodeAttribute"))
        {
            const FbxNode *attr = byId.at(l->child);
            if (attr->props.size() >= 3 && attr->props[2].bytes == "Camera")
            {
                // assimp's FBX converter: aspect = AspectWidth / AspectHeight, horizontal field of view = FieldOfView, the
                // camera at the node's origin looking along +X with +Y up; LoadCameras (SceneImporter.cpp:997-1020) turns
                // that into a vertical angle.  The document's cameras look along -Z: a child node turns -Z into +X.
                const Props70 cp(*attr);
                const double aspect = cp.Number("AspectWidth", 1.0) / std::max(cp.Number("AspectHeight", 1.0), 1e-9);
                const double horizontal = cp.Number("FieldOfView", 25.114999771118164) * 3.14159265358979323846 / 180.0;
                Json perspective = Obj();
                perspective.object["yfov"] = Num(2.0 * std::atan(std::tan(horizontal / 2.0) / (aspect == 0.0 ? 16.0 / 9.0 : aspect)));
                perspective.object["aspectRatio"] = Num(aspect);
                perspective.object["znear"] = Num(cp.Number("NearPlane", 10.0));
                perspective.object["zfar"] = Num(cp.Number("FarPlane", 4000.0));
                Json camera = Obj();
                camera.object["type"] = Str("perspective");
                camera.object["perspective"] = std::move(perspective);
                cameras.array.push_back(std::move(camera));
                Json child = Obj();
                child.object["name"] = Str(ObjectName(*model) + " (camera axis)");
                const double quarter[3] = { 0.0, -90.0, 0.0 };
                child.object["matrix"] = matrixJson(EulerXyz(quarter));
                child.object["camera"] = Num(static_cast<double>(cameras.array.size() - 1));
                nodes.array.push_back(std::move(child));
                nodes.array[nodeOfModel[id]].object["children"].array.push_back(Num(static_cast<double>(nodes.array.size() - 1)));
                continue;
            }
            if (attr->props.size() < 3 || attr->props[2].bytes != "Light")
                continue;
            const Props70 lp(*attr);
            double color[3] = { 1, 1, 1 };
            lp.Vector("Color", color);
            const int type = static_cast<int>(lp.Number("LightType", 0.0));
            Json light = Obj();
            light.object["type"] = Str(type == 1 ? "directional" : type == 2 ? "spot" : "point");
            light.object["color"] = Nums(color, 3);
            light.object["intensity"] = Num(lp.Number("Intensity", 100.0) / 100.0); // assimp: percent -> factor
            lights.array.push_back(std::move(light));
            // an FBX light shines along its local -Y, the document's lights along -Z: a child node turns one into the other
            Json ext = Obj(), ref = Obj();
            ref.object["light"] = Num(static_cast<double>(lights.array.size() - 1));
            ext.object["KHR_lights_punctual"] = std::move(ref);
            Json child = Obj();
            child.object["name"] = Str(ObjectName(*model) + " (light axis)");
            const double quarter[3] = { -90.0, 0.0, 0.0 };
            child.object["matrix"] = matrixJson(EulerXyz(quarter));
            child.object["extensions"] = std::move(ext);
            nodes.array.push_back(std::move(child));
            nodes.array[nodeOfModel[id]].object["children"].array.push_back(Num(static_cast<double>(nodes.array.size() - 1)));
        }
        if (attach.Has("mesh"))
        {
            const DMat geometric = GeometricTransform(p);
            if (geometric.IsIdentity())
            {
                nodes.array[nodeOfModel[id]].object["mesh"] = attach["mesh"];
                if (attach.Has("skin"))
                    nodes.array[nodeOfModel[id]].object["skin"] = attach["skin"];
            }
            else
            {
                Json child = Obj();
                child.object["name"] = Str(ObjectName(*model) + " (geometric transform)");
                child.object["matrix"] = matrixJson(geometric);
                child.object["mesh"] = attach["mesh"];
                if (attach.Has("skin"))
                    child.object["skin"] = attach["skin"];
                nodes.array.push_back(std::move(child));
                nodes.array[nodeOfModel[id]].object["children"].array.push_back(Num(static_cast<double>(nodes.array.size() - 1)));
            }
        }
    }

    // ---- animation: AnimationStack <- AnimationLayer <- AnimationCurveNode (-> a Model's Lcl Translation / Rotation / Scaling)
    // <- AnimationCurve per component (d|X, d|Y, d|Z; KeyTime in 1 / 46,186,158,000 s, KeyValueFloat).  A stack becomes one
    // animation of the document: per animated model and property a linear sampler over the union of its components' key
    // times.  Only models whose local transform is a plain T * R * S are animated (no pre- / post-rotation, pivots or
    // offsets: assimp spreads those over helper nodes); such a model is written as translation / rotation / scale.
    Json animations = Arr();
    {
        auto plainTrs = [&](const Props70 &p) {
            double v[3];
            for (const char *name : { "PreRotation", "PostRotation", "RotationPivot", "ScalingPivot", "RotationOffset", "ScalingOffset" })
                if (p.Vector(name, v) && (v[0] != 0.0 || v[1] != 0.0 || v[2] != 0.0))
                    return false;
            return true;
        };
        auto eulerQuaternion = [](const double degrees[3], double q[4]) { // x, y, z, w of Rz * Ry * Rx
            const double k = 3.14159265358979323846 / 360.0;
            const double cx = std::cos(degrees[0] * k), sx = std::sin(degrees[0] * k), cy = std::cos(degrees[1] * k), sy = std::sin(degrees[1] * k),
                         cz = std::cos(degrees[2] * k), sz = std::sin(degrees[2] * k);
            q[0] = sx * cy * cz - cx * sy * sz;
            q[1] = cx * sy * cz + sx * cy * sz;
            q[2] = cx * cy * sz - sx * sy * cz;
            q[3] = cx * cy * cz + sx * sy * sz;
        };
        struct Curve { std::vector<double> time, value; };
        auto readCurve = [&](int64_t id, Curve &c) {
            const FbxNode *n = byId.at(id);
            const std::vector<int64_t> *t = IntArray(n->Child("KeyTime"));
            const std::vector<double> *v = RealArray(n->Child("KeyValueFloat"));
            if (!t || !v || t->size() != v->size())
                return;
            for (size_t k = 0; k < t->size(); k++)
            {
                c.time.push_back(static_cast<double>((*t)[k]) / 46186158000.0);
                c.value.push_back((*v)[k]);
            }
        };
        auto evaluate = [](const Curve &c, double t, double fallback) {
            if (c.time.empty())
                return fallback;
            if (t <= c.time.front())
                return c.value.front();
            if (t >= c.time.back())
                return c.value.back();
            const size_t hi = static_cast<size_t>(std::upper_bound(c.time.begin(), c.time.end(), t) - c.time.begin());
            const double span = c.time[hi] - c.time[hi - 1], a = span > 0.0 ? (t - c.time[hi - 1]) / span : 0.0;
            return c.value[hi - 1] * (1.0 - a) + c.value[hi] * a;
        };
        std::unordered_map<int64_t, bool> writtenAsTrs;
        for (const FbxNode &stack : objects->children)
        {
            if (stack.name != "AnimationStack" || stack.props.empty())
                continue;
            Json samplers = Arr(), channels = Arr();
            for (const Link *layer : childrenOf(stack.props[0].integer, "AnimationLayer"))
                for (const Link *curveNode : childrenOf(layer->child, "AnimationCurveNode"))
                {
                    // which model property this curve node drives
                    int64_t model = 0;
                    std::string property;
                    for (const Link &l : links)
                        if (l.child == curveNode->child && !l.property.empty() && nodeOfModel.count(l.parent))
                        {
                            model = l.parent;
                            property = l.property;
                        }
                    const char *path = property == "Lcl Translation" ? "translation" : property == "Lcl Rotation" ? "rotation" : property == "Lcl Scaling" ? "scale" : nullptr;
                    if (!model || !path)
                        continue;
                    const Props70 mp(*byId.at(model));
                    if (!plainTrs(mp))
                        continue;
                    Curve xyz[3];
                    for (const Link *curve : childrenOf(curveNode->child, "AnimationCurve"))
                    {
                        const int axis = curve->property == "d|X" ? 0 : curve->property == "d|Y" ? 1 : curve->property == "d|Z" ? 2 : -1;
                        if (axis >= 0)
                            readCurve(curve->child, xyz[axis]);
                    }
                    std::vector<double> times;
                    for (const Curve &c : xyz)
                        times.insert(times.end(), c.time.begin(), c.time.end());
                    std::sort(times.begin(), times.end());
                    times.erase(std::unique(times.begin(), times.end()), times.end());
                    if (times.empty())
                        continue;
                    double rest[3] = { 0, 0, 0 };
                    if (std::strcmp(path, "scale") == 0)
                        rest[0] = rest[1] = rest[2] = 1.0;
                    mp.Vector(property, rest);
                    std::vector<float> input, output;
                    for (double t : times)
                    {
                        const double v[3] = { evaluate(xyz[0], t, rest[0]), evaluate(xyz[1], t, rest[1]), evaluate(xyz[2], t, rest[2]) };
                        input.push_back(static_cast<float>(t));
                        if (std::strcmp(path, "rotation") == 0)
                        {
                            double q[4];
                            eulerQuaternion(v, q);
                            for (double c : q)
                                output.push_back(static_cast<float>(c));
                        }
                        else
                            for (double c : v)
                                output.push_back(static_cast<float>(c));
                    }
                    Json sampler = Obj();
                    sampler.object["input"] = Num(static_cast<double>(writer.Add(input.data(), input.size() * 4, input.size(), 5126, "SCALAR")));
                    const bool rotation = std::strcmp(path, "rotation") == 0;
                    sampler.object["output"] = Num(static_cast<double>(writer.Add(output.data(), output.size() * 4, input.size(), 5126, rotation ? "VEC4" : "VEC3")));
                    sampler.object["interpolation"] = Str("LINEAR");
                    samplers.array.push_back(std::move(sampler));
                    Json target = Obj();
                    target.object["node"] = Num(static_cast<double>(nodeOfModel[model]));
                    target.object["path"] = Str(path);
                    Json channel = Obj();
                    channel.object["sampler"] = Num(static_cast<double>(samplers.array.size() - 1));
                    channel.object["target"] = std::move(target);
                    channels.array.push_back(std::move(channel));
                    if (!writtenAsTrs[model])
                    {
                        // the animated node carries translation / rotation / scale instead of a matrix
                        writtenAsTrs[model] = true;
                        Json &node = nodes.array[nodeOfModel[model]];
                        node.object.erase("matrix");
                        double t[3] = { 0, 0, 0 }, r[3] = { 0, 0, 0 }, sc[3] = { 1, 1, 1 }, q[4];
                        mp.Vector("Lcl Translation", t);
                        mp.Vector("Lcl Rotation", r);
                        mp.Vector("Lcl Scaling", sc);
                        eulerQuaternion(r, q);
                        node.object["translation"] = Nums(t, 3);
                        node.object["rotation"] = Nums(q, 4);
                        node.object["scale"] = Nums(sc, 3);
                    }
                }
            if (!channels.array.empty())
            {
                Json animation = Obj();
                animation.object["name"] = Str(ObjectName(stack));
                animation.object["samplers"] = std::move(samplers);
                animation.object["channels"] = std::move(channels);
                animations.array.push_back(std::move(animation));
            }
        }
    }

    Json scene = Obj();
    scene.object["nodes"] = std::move(roots);
    json.object["scene"] = Num(0);
    json.object["scenes"] = Arr();
    json.object["scenes"].array.push_back(std::move(scene));
    json.object["nodes"] = std::move(nodes);
    json.object["meshes"] = std::move(meshes);
    json.object["materials"] = std::move(materials);
    json.object["textures"] = std::move(textures);
    json.object["images"] = std::move(images);
    if (!cameras.array.empty())
        json.object["cameras"] = std::move(cameras);
    if (!animations.array.empty())
        json.object["animations"] = std::move(animations);
    if (!skins.array.empty())
        json.object["skins"] = std::move(skins);
    json.object["accessors"] = std::move(writer.accessors);
    json.object["bufferViews"] = std::move(writer.views);
    Json buf = Obj();
    buf.object["byteLength"] = Num(static_cast<double>(buffer.size()));
    json.object["buffers"] = Arr();
    json.object["buffers"].array.push_back(std::move(buf));
    if (!lights.array.empty())
    {
        Json ext = Obj(), punctual = Obj();
        punctual.object["lights"] = std::move(lights);
        ext.object["KHR_lights_punctual"] = std::move(punctual);
        json.object["extensions"] = std::move(ext);
    }
}

}
