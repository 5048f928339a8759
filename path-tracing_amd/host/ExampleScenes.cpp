// ExampleScenes.cpp -- scene registry of the backend.
//
//  * CreateDefaultScene / CreateRoughnessTestCubesScene restate the two procedural test
//    scenes that are fully specified in the reference's source
//    (Path-Tracing/ExampleScenes.cpp:264-298 AddCube, :320-545, :755-842); the embedded
//    PNG textures of Resources.cpp are referenced as texture descriptors only (software
//    texturing is the next row N1), so textured materials sample the white placeholder.
//  * attenuation_blob, chess_like, temple_like, atrium_like, street_like are the seeded
//    procedural STAND-INS for the asset scenes BASELINE.json names: the
//    assets are downloaded at CMake time by the reference (cmake/DownloadAssets.cmake)
//    and do not exist offline.
#include "ExampleScenes.h"
#include "SceneImporter.h"
#include "SceneDescription.h"

#include <fstream>
#include <sstream>

#include <array>
#include <cmath>
#include <functional>

namespace PathTracing::ExampleScenes
{

namespace
{

const PtxTransform kIdentity = IdentityTransform();

Shaders::MetallicRoughnessMaterial DefaultMaterialInfo()
{
    // ExampleScenes.cpp:322-334
    Shaders::MetallicRoughnessMaterial m;
    std::memset(&m, 0, sizeof(m));
    m.Color[0] = m.Color[1] = m.Color[2] = m.Color[3] = 1.0f;
    m.Roughness = 1.0f;
    m.Metalness = 0.0f;
    m.Ior = 1.5f;
    m.AttenuationColor[0] = m.AttenuationColor[1] = m.AttenuationColor[2] = 1.0f;
    m.AttenuationDistance = 1e32f;
    m.EmissiveIdx = Scene::GetDefaultTextureIndex(TextureType::Emisive);
    m.ColorIdx = Scene::GetDefaultTextureIndex(TextureType::Color);
    m.NormalIdx = Scene::GetDefaultTextureIndex(TextureType::Normal);
    m.RoughnessIdx = Scene::GetDefaultTextureIndex(TextureType::Roughness);
    m.MetallicIdx = Scene::GetDefaultTextureIndex(TextureType::Metallic);
    return m;
}

Shaders::Vertex V(std::array<float, 3> p, std::array<float, 2> uv, std::array<float, 3> n, std::array<float, 3> t,
                  std::array<float, 3> b)
{
    Shaders::Vertex v;
    for (int i = 0; i < 3; i++)
    {
        v.Position[i] = p[i];
        v.Normal[i] = n[i];
        v.Tangent[i] = t[i];
        v.Bitangent[i] = b[i];
    }
    v.TexCoords[0] = uv[0];
    v.TexCoords[1] = uv[1];
    return v;
}

// ExampleScenes.cpp:264-318
std::array<uint32_t, 6> AddCube(SceneBuilder &sceneBuilder)
{
    auto &vertices = sceneBuilder.GetVertices();
    uint32_t vertexOffset = static_cast<uint32_t>(vertices.size());
    const Shaders::Vertex cube[24] = {
        V({ -1, -1, 1 }, { 0, 1 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),
        V({ 1, -1, 1 }, { 1, 1 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),
        V({ 1, 1, 1 }, { 1, 0 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),
        V({ -1, 1, 1 }, { 0, 0 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),

        V({ 1, -1, -1 }, { 0, 1 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),
        V({ -1, -1, -1 }, { 1, 1 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),
        V({ -1, 1, -1 }, { 1, 0 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),
        V({ 1, 1, -1 }, { 0, 0 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),

        V({ -1, -1, -1 }, { 0, 1 }, { -1, 0, 0 }, { 0, 0, 1 }, { 0, 1, 0 }),
        V({ -1, -1, 1 }, { 1, 1 }, { -1, 0, 0 }, { 0, 0, 1 }, { 0, 1, 0 }),
        V({ -1, 1, 1 }, { 1, 0 }, { -1, 0, 0 }, { 0, 0, 1 }, { 0, 1, 0 }),
        V({ -1, 1, -1 }, { 0, 0 }, { -1, 0, 0 }, { 0, 0, 1 }, { 0, 1, 0 }),

        V({ 1, -1, 1 }, { 0, 1 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),
        V({ 1, -1, -1 }, { 1, 1 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),
        V({ 1, 1, -1 }, { 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),
        V({ 1, 1, 1 }, { 0, 0 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),

        V({ -1, 1, 1 }, { 0, 1 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),
        V({ 1, 1, 1 }, { 1, 1 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),
        V({ 1, 1, -1 }, { 1, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),
        V({ -1, 1, -1 }, { 0, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),

        V({ -1, -1, -1 }, { 0, 1 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
        V({ 1, -1, -1 }, { 1, 1 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
        V({ 1, -1, 1 }, { 1, 0 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
        V({ -1, -1, 1 }, { 0, 0 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
    };
    vertices.insert(vertices.end(), std::begin(cube), std::end(cube));

    auto &indices = sceneBuilder.GetIndices();
    uint32_t indexOffset = static_cast<uint32_t>(indices.size());
    for (int i = 0; i < 6; i++)
        for (uint32_t k : { 0u, 1u, 2u, 2u, 3u, 0u })
            indices.push_back(k);

    std::array<uint32_t, 6> geometryIndices = {};
    for (uint32_t i = 0; i < 6; i++)
    {
        geometryIndices[i] = sceneBuilder.AddGeometry({ vertexOffset, 4, indexOffset, 6, true, false, { 0, 0 } });
        vertexOffset += 4;
        indexOffset += 6;
    }
    return geometryIndices;
}

MeshInfo MI(uint32_t geometry, Shaders::MaterialId material, const PtxTransform &t = kIdentity)
{
    return MeshInfo { geometry, material, MaterialType::MetallicRoughness, t };
}

// ---------------------------------------------------------------------------
// procedural helpers for the stand-in scenes
// ---------------------------------------------------------------------------

struct Rng
{
    uint32_t s;
    explicit Rng(uint32_t seed) : s(seed * 747796405u + 2891336453u) {}
    uint32_t NextU()
    {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        return s;
    }
    float Next() { return static_cast<float>(NextU() >> 8) * (1.0f / 16777216.0f); }
    float Range(float a, float b) { return a + (b - a) * Next(); }
};

Vec3 SafeNormalize(Vec3 v, Vec3 fallback)
{
    const float l = Length(v);
    return l > 1e-20f ? v * (1.0f / l) : fallback;
}

// Parametric (nu x nv)-quad surface; i runs along u, j along v.  Triangles are
// (a, d, c), (a, c, b) with a=(i,j) b=(i+1,j) c=(i+1,j+1) d=(i,j+1), i.e. the
// geometric normal is cross(dP/dv, dP/du) (outward for a lathe with u = angle).
uint32_t AddGridSurface(SceneBuilder &sb, uint32_t nu, uint32_t nv, bool closedU,
                        const std::function<Vec3(float, float)> &f, bool flip = false, bool opaque = true)
{
    auto &vertices = sb.GetVertices();
    auto &indices = sb.GetIndices();
    const uint32_t vertexOffset = static_cast<uint32_t>(vertices.size());
    const uint32_t indexOffset = static_cast<uint32_t>(indices.size());
    const uint32_t cu = closedU ? nu : nu + 1, cv = nv + 1;

    std::vector<Vec3> P(static_cast<size_t>(cu) * cv), N(static_cast<size_t>(cu) * cv, Vec3(0.0f));
    for (uint32_t j = 0; j < cv; j++)
        for (uint32_t i = 0; i < cu; i++)
            P[static_cast<size_t>(j) * cu + i] = f(static_cast<float>(i) / nu, static_cast<float>(j) / nv);
    auto at = [&](uint32_t i, uint32_t j) { return static_cast<uint32_t>(j * cu + (closedU ? i % nu : i)); };

    for (uint32_t j = 0; j < nv; j++)
        for (uint32_t i = 0; i < nu; i++)
        {
            const uint32_t a = at(i, j), b = at(i + 1, j), c = at(i + 1, j + 1), d = at(i, j + 1);
            const uint32_t tri[2][3] = { { a, flip ? c : d, flip ? d : c }, { a, flip ? b : c, flip ? c : b } };
            for (auto &t : tri)
            {
                indices.push_back(t[0]);
                indices.push_back(t[1]);
                indices.push_back(t[2]);
                const Vec3 fn = Cross(P[t[1]] - P[t[0]], P[t[2]] - P[t[0]]);
                N[t[0]] = N[t[0]] + fn;
                N[t[1]] = N[t[1]] + fn;
                N[t[2]] = N[t[2]] + fn;
            }
        }

    for (uint32_t j = 0; j < cv; j++)
        for (uint32_t i = 0; i < cu; i++)
        {
            const uint32_t k = j * cu + i;
            const Vec3 n = SafeNormalize(N[k], Vec3(0, 1, 0));
            const uint32_t i0 = closedU ? (i + nu - 1) % nu : (i > 0 ? i - 1 : i);
            const uint32_t i1 = closedU ? (i + 1) % nu : (i + 1 < cu ? i + 1 : i);
            Vec3 t = P[j * cu + i1] - P[j * cu + i0];
            t = t - n * Dot(n, t);
            const Vec3 axis = std::fabs(n.x) < 0.9f ? Vec3(1, 0, 0) : Vec3(0, 0, 1);
            t = SafeNormalize(t, SafeNormalize(Cross(n, axis), Vec3(0, 0, 1)));
            const Vec3 b = SafeNormalize(Cross(n, t), Vec3(0, 1, 0));
            Shaders::Vertex v;
            v.Position[0] = P[k].x; v.Position[1] = P[k].y; v.Position[2] = P[k].z;
            v.TexCoords[0] = static_cast<float>(i) / nu;
            v.TexCoords[1] = static_cast<float>(j) / nv;
            v.Normal[0] = n.x; v.Normal[1] = n.y; v.Normal[2] = n.z;
            v.Tangent[0] = t.x; v.Tangent[1] = t.y; v.Tangent[2] = t.z;
            v.Bitangent[0] = b.x; v.Bitangent[1] = b.y; v.Bitangent[2] = b.z;
            vertices.push_back(v);
        }
    return sb.AddGeometry({ vertexOffset, cu * cv, indexOffset, nu * nv * 6, opaque, false, { 0, 0 } });
}

// Surface of revolution around +y through a piecewise-linear (radius, height) profile,
// resampled to `rings` rows with a smooth (Catmull-Rom) interpolation.
uint32_t AddLathe(SceneBuilder &sb, const std::vector<Vec2> &profile, uint32_t segments, uint32_t rings)
{
    const float twoPi = 6.283185307179586f;
    auto sample = [&](float v) {
        const float x = v * static_cast<float>(profile.size() - 1);
        int k = static_cast<int>(x);
        if (k >= static_cast<int>(profile.size()) - 1)
            k = static_cast<int>(profile.size()) - 2;
        const float t = x - static_cast<float>(k);
        auto P = [&](int q) { return profile[static_cast<size_t>(q < 0 ? 0 : (q >= static_cast<int>(profile.size()) ? static_cast<int>(profile.size()) - 1 : q))]; };
        const Vec2 p0 = P(k - 1), p1 = P(k), p2 = P(k + 1), p3 = P(k + 2);
        auto cr = [&](float a, float b, float c, float d) {
            return 0.5f * ((2 * b) + (-a + c) * t + (2 * a - 5 * b + 4 * c - d) * t * t + (-a + 3 * b - 3 * c + d) * t * t * t);
        };
        Vec2 r;
        r.x = cr(p0.x, p1.x, p2.x, p3.x);
        r.y = cr(p0.y, p1.y, p2.y, p3.y);
        if (r.x < 0)
            r.x = 0;
        return r;
    };
    return AddGridSurface(sb, segments, rings, true, [&](float u, float v) {
        const Vec2 p = sample(v);
        return Vec3(p.x * std::cos(twoPi * u), p.y, p.x * std::sin(twoPi * u));
    });
}

// Axis-aligned box made of six one-quad geometries merged into ONE geometry (12 tris)
uint32_t AddBox(SceneBuilder &sb, Vec3 c, Vec3 h)
{
    auto &vertices = sb.GetVertices();
    auto &indices = sb.GetIndices();
    const uint32_t vertexOffset = static_cast<uint32_t>(vertices.size());
    const uint32_t indexOffset = static_cast<uint32_t>(indices.size());
    const Vec3 nrm[6] = { { 0, 0, 1 }, { 0, 0, -1 }, { -1, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 } };
    const Vec3 tan[6] = { { 1, 0, 0 }, { -1, 0, 0 }, { 0, 0, 1 }, { 0, 0, -1 }, { 1, 0, 0 }, { 1, 0, 0 } };
    for (int f = 0; f < 6; f++)
    {
        const Vec3 n = nrm[f], t = tan[f], b = Cross(n, t);
        const float sx[4] = { -1, 1, 1, -1 }, sy[4] = { -1, -1, 1, 1 };
        for (int k = 0; k < 4; k++)
        {
            const Vec3 p = Vec3(c.x + h.x * (n.x + t.x * sx[k] + b.x * sy[k]), c.y + h.y * (n.y + t.y * sx[k] + b.y * sy[k]),
                                c.z + h.z * (n.z + t.z * sx[k] + b.z * sy[k]));
            vertices.push_back(V({ p.x, p.y, p.z }, { 0.5f + 0.5f * sx[k], 0.5f - 0.5f * sy[k] }, { n.x, n.y, n.z },
                                 { t.x, t.y, t.z }, { b.x, b.y, b.z }));
        }
        for (uint32_t k : { 0u, 1u, 2u, 2u, 3u, 0u })
            indices.push_back(static_cast<uint32_t>(f) * 4 + k);
    }
    return sb.AddGeometry({ vertexOffset, 24, indexOffset, 36, true, false, { 0, 0 } });
}

// `count` small randomly oriented quads ("leaf cards") inside a box
uint32_t AddCards(SceneBuilder &sb, Rng &rng, uint32_t count, Vec3 lo, Vec3 hi, float size, bool opaque = true)
{
    auto &vertices = sb.GetVertices();
    auto &indices = sb.GetIndices();
    const uint32_t vertexOffset = static_cast<uint32_t>(vertices.size());
    const uint32_t indexOffset = static_cast<uint32_t>(indices.size());
    for (uint32_t q = 0; q < count; q++)
    {
        const Vec3 c(rng.Range(lo.x, hi.x), rng.Range(lo.y, hi.y), rng.Range(lo.z, hi.z));
        Vec3 n = SafeNormalize(Vec3(rng.Range(-1, 1), rng.Range(-1, 1), rng.Range(-1, 1)), Vec3(0, 1, 0));
        const Vec3 axis = std::fabs(n.x) < 0.9f ? Vec3(1, 0, 0) : Vec3(0, 0, 1);
        const Vec3 t = Normalize(Cross(n, axis)), b = Cross(n, t);
        const float s = size * rng.Range(0.5f, 1.0f);
        const float sx[4] = { -1, 1, 1, -1 }, sy[4] = { -1, -1, 1, 1 };
        for (int k = 0; k < 4; k++)
        {
            const Vec3 p = c + t * (s * sx[k]) + b * (s * sy[k]);
            vertices.push_back(V({ p.x, p.y, p.z }, { 0.5f + 0.5f * sx[k], 0.5f - 0.5f * sy[k] }, { n.x, n.y, n.z },
                                 { t.x, t.y, t.z }, { b.x, b.y, b.z }));
        }
        for (uint32_t k : { 0u, 1u, 2u, 2u, 3u, 0u })
            indices.push_back(q * 4 + k);
    }
    return sb.AddGeometry({ vertexOffset, count * 4, indexOffset, count * 6, opaque, false, { 0, 0 } });
}

Shaders::MetallicRoughnessMaterial MakeMaterial(Vec3 color, float roughness, float metalness)
{
    Shaders::MetallicRoughnessMaterial m = DefaultMaterialInfo();
    m.Color[0] = color.x;
    m.Color[1] = color.y;
    m.Color[2] = color.z;
    m.Roughness = roughness;
    m.Metalness = metalness;
    return m;
}

PtxTransform ToTransform(const Mat4 &m)
{
    PtxTransform t;
    std::memcpy(t.m, &m.m[0][0], sizeof(t.m));
    return t;
}

uint32_t Scaled(uint32_t n, float detail, uint32_t minimum)
{
    const uint32_t v = static_cast<uint32_t>(static_cast<float>(n) * detail + 0.5f);
    return v < minimum ? minimum : v;
}

Shaders::PointLight MakePointLight(Vec3 color, Vec3 position)
{
    Shaders::PointLight l;
    std::memset(&l, 0, sizeof(l));
    l.Color[0] = color.x; l.Color[1] = color.y; l.Color[2] = color.z;
    l.Position[0] = position.x; l.Position[1] = position.y; l.Position[2] = position.z;
    l.AttenuationConstant = 1.0f;
    l.AttenuationLinear = 0.09f;
    l.AttenuationQuadratic = 0.032f;
    return l;
}

void AddViewCamera(SceneBuilder &sb, Vec3 position, Vec3 target, float fov = 45.0f)
{
    sb.AddCamera({ fov, 0.1f, 1000.0f, position, Normalize(target - position), Vec3(0.0f, -1.0f, 0.0f), SceneBuilder::RootNodeIndex });
}

}

// ---------------------------------------------------------------------------
// Reference test scenes
// ---------------------------------------------------------------------------

// ExampleScenes.cpp:320-545
void CreateDefaultScene(SceneBuilder &sceneBuilder)
{
    const Shaders::MetallicRoughnessMaterial defaultMaterialInfo = DefaultMaterialInfo();

    const uint32_t logoColorTexture = sceneBuilder.AddTexture({ TextureType::Color, 1, 1, "Logo Color Texture", TextureFormat::RGBAU8, {} });
    const uint32_t vulkanPathTracingTexture = sceneBuilder.AddTexture({ TextureType::Color, 1, 1, "Vulkan Path-Tracing Texture", TextureFormat::RGBAU8, {} });
    const uint32_t authorsTexture = sceneBuilder.AddTexture({ TextureType::Color, 1, 1, "Authors Texture", TextureFormat::RGBAU8, {} });
    const uint32_t pressSpaceTexture = sceneBuilder.AddTexture({ TextureType::Color, 1, 1, "Press Space Texture", TextureFormat::RGBAU8, {} });

    auto whiteMaterialInfo = defaultMaterialInfo;
    auto greenMaterialInfo = defaultMaterialInfo;
    greenMaterialInfo.Color[0] = 0.0f; greenMaterialInfo.Color[1] = 1.0f; greenMaterialInfo.Color[2] = 0.0f;
    auto redMaterialInfo = defaultMaterialInfo;
    redMaterialInfo.Color[0] = 1.0f; redMaterialInfo.Color[1] = 0.0f; redMaterialInfo.Color[2] = 0.0f;
    auto logoMaterialInfo = defaultMaterialInfo;
    logoMaterialInfo.ColorIdx = logoColorTexture;
    auto lightMaterialInfo = defaultMaterialInfo;
    lightMaterialInfo.EmissiveColor[0] = lightMaterialInfo.EmissiveColor[1] = lightMaterialInfo.EmissiveColor[2] = 1.0f;
    lightMaterialInfo.EmissiveIntensity = 1.0f;
    auto glassMaterialInfo = defaultMaterialInfo;
    glassMaterialInfo.Color[0] = 0.70f; glassMaterialInfo.Color[1] = 0.81f; glassMaterialInfo.Color[2] = 0.85f;
    glassMaterialInfo.Roughness = 0.0f;
    glassMaterialInfo.Transmission = 1.0f;
    glassMaterialInfo.Ior = 1.5f;
    auto glassTexturedMaterialInfo = glassMaterialInfo;
    glassTexturedMaterialInfo.ColorIdx = authorsTexture;
    auto mirrorMaterialInfo = defaultMaterialInfo;
    mirrorMaterialInfo.Roughness = 0.0f;
    mirrorMaterialInfo.Metalness = 1.0f;
    auto mirrorTexturedMaterialInfo = mirrorMaterialInfo;
    mirrorTexturedMaterialInfo.ColorIdx = vulkanPathTracingTexture;
    auto floorMaterialInfo = defaultMaterialInfo;
    floorMaterialInfo.ColorIdx = pressSpaceTexture;

    Shaders::MaterialId whiteMaterial = sceneBuilder.AddMaterial("White Material", whiteMaterialInfo);
    Shaders::MaterialId greenMaterial = sceneBuilder.AddMaterial("Green Material", greenMaterialInfo);
    Shaders::MaterialId redMaterial = sceneBuilder.AddMaterial("Red Material", redMaterialInfo);
    Shaders::MaterialId logoMaterial = sceneBuilder.AddMaterial("Logo Material", logoMaterialInfo);
    Shaders::MaterialId lightMaterial = sceneBuilder.AddMaterial("Light Material", lightMaterialInfo);
    Shaders::MaterialId glassMaterial = sceneBuilder.AddMaterial("Glass Material", glassMaterialInfo);
    Shaders::MaterialId glassTexturedMaterial = sceneBuilder.AddMaterial("Glass Textured Material", glassTexturedMaterialInfo);
    Shaders::MaterialId mirrorMaterial = sceneBuilder.AddMaterial("Mirror Material", mirrorMaterialInfo);
    Shaders::MaterialId mirrorTexturedMaterial = sceneBuilder.AddMaterial("Mirror Textured Material", mirrorTexturedMaterialInfo);
    Shaders::MaterialId floorMaterial = sceneBuilder.AddMaterial("Floor Material", floorMaterialInfo);

    auto &vertices = sceneBuilder.GetVertices();
    vertices = {
        V({ -1.1f, -1.1f, -1 }, { 0, 1 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),
        V({ 1.1f, -1.1f, -1 }, { 1, 1 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),
        V({ 1.1f, 1.1f, -1 }, { 1, 0 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),
        V({ -1.1f, 1.1f, -1 }, { 0, 0 }, { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }),

        V({ 1.1f, -1.1f, 1 }, { 0, 1 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),
        V({ -1.1f, -1.1f, 1 }, { 1, 1 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),
        V({ -1.1f, 1.1f, 1 }, { 1, 0 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),
        V({ 1.1f, 1.1f, 1 }, { 0, 0 }, { 0, 0, -1 }, { -1, 0, 0 }, { 0, 1, 0 }),

        V({ -1.1f, -1.1f, 1 }, { 0, 1 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),
        V({ -1.1f, -1.1f, -1 }, { 1, 1 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),
        V({ -1.1f, 1.1f, -1 }, { 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),
        V({ -1.1f, 1.1f, 1 }, { 0, 0 }, { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 }),

        V({ -1.1f, -1.1f, 1 }, { 0, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),
        V({ 1.1f, -1.1f, 1 }, { 0, 1 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),
        V({ 1.1f, -1.1f, -1 }, { 1, 1 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),
        V({ -1.1f, -1.1f, -1 }, { 1, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }),

        V({ -1.1f, 1.1f, -1 }, { 0, 1 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
        V({ 1.1f, 1.1f, -1 }, { 1, 1 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
        V({ 1.1f, 1.1f, 1 }, { 1, 0 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
        V({ -1.1f, 1.1f, 1 }, { 0, 0 }, { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }),
    };

    auto &indices = sceneBuilder.GetIndices();
    for (int i = 0; i < 5; i++)
        for (uint32_t k : { 0u, 1u, 2u, 2u, 3u, 0u })
            indices.push_back(k);

    uint32_t vertexOffset = 0, indexOffset = 0;
    for (uint32_t i = 0; i < 5; i++)
    {
        sceneBuilder.AddGeometry({ vertexOffset, 4, indexOffset, 6, true, false, { 0, 0 } });
        vertexOffset += 4;
        indexOffset += 6;
    }

    std::array<MeshInfo, 5> meshes = { { MI(0, redMaterial), MI(1, greenMaterial), MI(2, logoMaterial), MI(3, floorMaterial),
                                         MI(4, whiteMaterial) } };

    std::array<uint32_t, 6> geometryIndices = AddCube(sceneBuilder);

    std::array<MeshInfo, 6> glassCubeMeshes = { { MI(geometryIndices[0], glassMaterial), MI(geometryIndices[1], glassMaterial),
                                                  MI(geometryIndices[2], glassMaterial), MI(geometryIndices[3], glassTexturedMaterial),
                                                  MI(geometryIndices[4], glassMaterial), MI(geometryIndices[5], glassMaterial) } };
    std::array<MeshInfo, 6> metallicCubeMeshes = { { MI(geometryIndices[0], mirrorMaterial), MI(geometryIndices[1], mirrorMaterial),
                                                     MI(geometryIndices[2], mirrorMaterial), MI(geometryIndices[3], mirrorTexturedMaterial),
                                                     MI(geometryIndices[4], mirrorMaterial), MI(geometryIndices[5], mirrorMaterial) } };

    const uint32_t lightVertexOffset = static_cast<uint32_t>(vertices.size());
    const uint32_t lightIndexOffset = static_cast<uint32_t>(indices.size());
    vertices.push_back(V({ 0.2f, 0.0f, 0.2f }, { 1.0f, 1.0f }, { 0.0f, -1.0f, 0.0f }, { 1, 0, 0 }, { 0, 0, 1 }));
    vertices.push_back(V({ -0.2f, 0.0f, 0.2f }, { 0.0f, 1.0f }, { 0.0f, -1.0f, 0.0f }, { 1, 0, 0 }, { 0, 0, 1 }));
    vertices.push_back(V({ -0.2f, 0.0f, -0.2f }, { 0.0f, 1.0f }, { 0.0f, -1.0f, 0.0f }, { 1, 0, 0 }, { 0, 0, 1 }));
    vertices.push_back(V({ 0.2f, 0.0f, -0.2f }, { 1.0f, 0.0f }, { 0.0f, -1.0f, 0.0f }, { 1, 0, 0 }, { 0, 0, 1 }));
    for (uint32_t k : { 0u, 1u, 2u, 2u, 3u, 0u })
        indices.push_back(k);

    const uint32_t lightGeometry = sceneBuilder.AddGeometry({ lightVertexOffset, 4, lightIndexOffset, 6, true, false, { 0, 0 } });
    const std::array<MeshInfo, 1> lightMeshes = { MI(lightGeometry, lightMaterial) };

    const uint32_t box = sceneBuilder.AddModel(meshes);
    const uint32_t metallicCube = sceneBuilder.AddModel(metallicCubeMeshes);
    const uint32_t glassCube = sceneBuilder.AddModel(glassCubeMeshes);
    const uint32_t light = sceneBuilder.AddModel(lightMeshes);

    const Mat4 boxTransform = Translate(Scale(Mat4::Identity(), Vec3(2.0f)), Vec3(-2.25f, 0.5f, 0.0f));

    const uint32_t rootNode = sceneBuilder.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const uint32_t boxNode = sceneBuilder.AddSceneNode({ rootNode, boxTransform, Mat4::Identity() });
    sceneBuilder.AddModelInstance(box, boxNode);

    const Mat4 leftCubeTransform =
        Scale(Rotate(Translate(Mat4::Identity(), Vec3(-0.4f, -0.795f, 0.5f)), Radians(25.0f), Vec3(0.0f, 1.0f, 0.0f)), Vec3(0.3f));
    const uint32_t leftCubeNode = sceneBuilder.AddSceneNode({ boxNode, leftCubeTransform, Mat4::Identity() });

    const Mat4 rightCubeTransform =
        Scale(Rotate(Translate(Mat4::Identity(), Vec3(0.2f, -0.795f, -0.6f)), Radians(-20.0f), Vec3(0.0f, 1.0f, 0.0f)), Vec3(0.3f));
    const uint32_t rightCubeNode = sceneBuilder.AddSceneNode({ boxNode, rightCubeTransform, Mat4::Identity() });

    sceneBuilder.AddModelInstance(metallicCube, leftCubeNode);
    sceneBuilder.AddModelInstance(glassCube, rightCubeNode);

    const Mat4 lightTransform = Translate(Mat4::Identity(), Vec3(0.0f, 1.099f, 0.0f));
    const uint32_t lightNode = sceneBuilder.AddSceneNode({ boxNode, lightTransform, Mat4::Identity() });
    sceneBuilder.AddModelInstance(light, lightNode);

    Shaders::DirectionalLight dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.Direction[1] = -1.0f;
    sceneBuilder.SetDirectionalLight(std::move(dl), rootNode);
}

TextureInfo MakeTexture(TextureType type, const std::string &name, uint32_t w, uint32_t h,
                        const std::function<void(uint32_t, uint32_t, uint8_t *)> &texel);
uint32_t HashU(uint32_t x, uint32_t y, uint32_t seed);

// Procedural daylight sky, the stand-in for the reference's sky_42 skybox images (downloaded assets):
// blue zenith, pale horizon, grey-brown ground, a sun disc, thin cloud streaks.  8-bit sRGB like the PNGs.
void SkyTexel(Vec3 d, uint8_t *p)
{
    d = Normalize(d);
    const Vec3 sun = Normalize(Vec3(0.45f, 0.55f, -0.7f));
    float r, g, b;
    if (d.y >= 0.0f)
    {
        const float h = std::pow(1.0f - d.y, 3.0f);
        r = 0.25f + 0.6f * h; g = 0.45f + 0.45f * h; b = 0.85f + 0.1f * h;
        const float streak = std::sin(9.0f * d.x + 4.0f * d.z) * std::sin(13.0f * d.z - 3.0f * d.x);
        const float cloud = streak > 0.55f ? (streak - 0.55f) * 1.6f * d.y : 0.0f;
        r += cloud * (1.0f - r); g += cloud * (1.0f - g); b += cloud * (1.0f - b);
        const float c = Dot(d, sun);
        if (c > 0.9985f)
            r = g = b = 1.0f;
        else if (c > 0.96f)
        {
            const float halo = (c - 0.96f) / 0.0385f;
            r += halo * (1.0f - r); g += halo * 0.9f * (1.0f - g); b += halo * 0.6f * (1.0f - b);
        }
    }
    else
    {
        const float h = std::pow(1.0f + d.y, 6.0f);
        r = 0.22f + 0.5f * h; g = 0.2f + 0.5f * h; b = 0.18f + 0.55f * h;
    }
    auto q = [](float v) { return static_cast<uint8_t>(std::fmin(std::fmax(v, 0.0f), 1.0f) * 255.0f + 0.5f); };
    p[0] = q(r); p[1] = q(g); p[2] = q(b); p[3] = 255;
}

// equirectangular image addressed as miss.rmiss:20-25 does: u = atan(z, x) / 2pi + 0.5, v = asin(-y) / pi + 0.5
Skybox2D MakeSkybox2D(uint32_t width, uint32_t height)
{
    const float pi = 3.14159265358979f;
    return Skybox2D { MakeTexture(TextureType::Skybox, "Skybox", width, height, [=](uint32_t x, uint32_t y, uint8_t *p) {
        const float lon = ((static_cast<float>(x) + 0.5f) / static_cast<float>(width) - 0.5f) * 2.0f * pi;
        const float lat = ((static_cast<float>(y) + 0.5f) / static_cast<float>(height) - 0.5f) * pi;
        SkyTexel(Vec3(std::cos(lat) * std::cos(lon), -std::sin(lat), std::cos(lat) * std::sin(lon)), p);
    }) };
}

// six faces in the layer order +X -X +Y -Y +Z -Z (px nx py ny pz nz of ExampleScenes.cpp:745-752)
SkyboxCube MakeSkyboxCube(uint32_t size)
{
    const char *names[6] = { "Skybox px", "Skybox nx", "Skybox py", "Skybox ny", "Skybox pz", "Skybox nz" };
    TextureInfo faces[6];
    for (int f = 0; f < 6; f++)
        faces[f] = MakeTexture(TextureType::Skybox, names[f], size, size, [=](uint32_t x, uint32_t y, uint8_t *p) {
            const float sc = 2.0f * (static_cast<float>(x) + 0.5f) / static_cast<float>(size) - 1.0f;
            const float tc = 2.0f * (static_cast<float>(y) + 0.5f) / static_cast<float>(size) - 1.0f;
            const Vec3 dirs[6] = { Vec3(1, -tc, -sc), Vec3(-1, -tc, sc), Vec3(sc, 1, tc), Vec3(sc, -1, -tc), Vec3(sc, -tc, 1), Vec3(-sc, -tc, -1) };
            SkyTexel(dirs[f], p);
        });
    return SkyboxCube { std::move(faces[0]), std::move(faces[1]), std::move(faces[2]), std::move(faces[3]), std::move(faces[4]), std::move(faces[5]) };
}

// ExampleScenes.cpp:755-842; the 2-D skybox is the procedural stand-in above
void CreateRoughnessTestCubesScene(SceneBuilder &sceneBuilder)
{
    std::array<std::array<Shaders::MaterialId, 6>, 6> whiteMaterials;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++)
        {
            Shaders::MetallicRoughnessMaterial m;
            std::memset(&m, 0, sizeof(m)); // aggregate init in the reference: unnamed members are zero
            m.Color[0] = m.Color[1] = m.Color[2] = m.Color[3] = 1.0f;
            m.Roughness = static_cast<float>(i) * 0.2f;
            m.Metalness = static_cast<float>(j) * 0.2f;
            m.Ior = 1.5f;
            m.EmissiveIdx = Scene::GetDefaultTextureIndex(TextureType::Emisive);
            m.ColorIdx = Scene::GetDefaultTextureIndex(TextureType::Color);
            m.NormalIdx = Scene::GetDefaultTextureIndex(TextureType::Normal);
            m.RoughnessIdx = Scene::GetDefaultTextureIndex(TextureType::Roughness);
            m.MetallicIdx = Scene::GetDefaultTextureIndex(TextureType::Metallic);
            whiteMaterials[i][j] = sceneBuilder.AddMaterial("White Material " + std::to_string(i) + "_" + std::to_string(j), m);
        }

    std::array<uint32_t, 6> geometryIndices = AddCube(sceneBuilder);

    std::array<uint32_t, 36> cubeModels;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++)
        {
            std::array<MeshInfo, 6> cubeMeshes;
            for (int k = 0; k < 6; k++)
                cubeMeshes[k] = MI(geometryIndices[k], whiteMaterials[i][j]);
            cubeModels[i * 6 + j] = sceneBuilder.AddModel(cubeMeshes);
        }

    const uint32_t rootNode = sceneBuilder.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++)
        {
            const Mat4 t = Translate(Mat4::Identity(), Vec3(static_cast<float>(j) * -4.0f, 0.0f, static_cast<float>(i) * -4.0f));
            const uint32_t cubeNode = sceneBuilder.AddSceneNode({ rootNode, t, Mat4::Identity() });
            sceneBuilder.AddModelInstance(cubeModels[i * 6 + j], cubeNode);
        }

    sceneBuilder.SetSkybox(MakeSkybox2D(1024, 512)); // :839-841
}

// ExampleScenes.cpp:658-753 "Reuse Mesh Cubes": ONE cube model assembled from three quad geometries, each used
// twice through a baked mesh transform (the half-turn about x / y / z), three PBR materials with colour, normal,
// roughness and metallic maps, and a cube-map skybox.  The downloaded JPG material sets (Metal062C,
// PavingStones142, Logs001) are replaced by procedural 256^2 maps of the same roles.
void CreateReuseMeshCubesScene(SceneBuilder &sb, uint32_t seed)
{
    const char *names[3] = { "Metal", "PavingStones", "Logs" };
    Shaders::MaterialId materialIds[3];
    for (int i = 0; i < 3; i++)
    {
        const uint32_t salt = seed * 16u + static_cast<uint32_t>(i);
        auto pattern = [i, salt](uint32_t x, uint32_t y) { // 0..1 height-like value per material
            if (i == 0)
                return 0.5f + 0.5f * std::sin(0.35f * static_cast<float>(x) + 0.02f * static_cast<float>(HashU(0, y / 3, salt) & 63));
            if (i == 1)
            {
                const uint32_t bx = x % 64, by = (y + (x / 64 % 2) * 32) % 64;
                const bool joint = bx < 4 || by < 4;
                return joint ? 0.1f : 0.6f + 0.4f * static_cast<float>(HashU(x / 64, (y + (x / 64 % 2) * 32) / 64, salt) & 255) / 255.0f;
            }
            const float ring = std::sin(0.18f * std::sqrt(static_cast<float>((x % 128) * (x % 128) + ((y + 40) % 128) * ((y + 40) % 128))));
            return 0.5f + 0.5f * ring;
        };
        const Vec3 tint = i == 0 ? Vec3(0.78f, 0.8f, 0.84f) : i == 1 ? Vec3(0.62f, 0.58f, 0.52f) : Vec3(0.5f, 0.33f, 0.18f);
        const std::string base = names[i];
        auto m = DefaultMaterialInfo();
        m.Color[0] = m.Color[1] = m.Color[2] = m.Color[3] = 1.0f;
        m.Roughness = 1.0f;
        m.Metalness = 1.0f;
        m.Ior = 1.5f;
        m.ColorIdx = sb.AddTexture(MakeTexture(TextureType::Color, base + "_Color", 256, 256, [=](uint32_t x, uint32_t y, uint8_t *p) {
            const float v = 0.55f + 0.45f * pattern(x, y);
            p[0] = static_cast<uint8_t>(255.0f * tint.x * v); p[1] = static_cast<uint8_t>(255.0f * tint.y * v);
            p[2] = static_cast<uint8_t>(255.0f * tint.z * v); p[3] = 255;
        }));
        m.NormalIdx = sb.AddTexture(MakeTexture(TextureType::Normal, base + "_NormalGL", 256, 256, [=](uint32_t x, uint32_t y, uint8_t *p) {
            const float dx = pattern((x + 1) % 256, y) - pattern((x + 255) % 256, y), dy = pattern(x, (y + 1) % 256) - pattern(x, (y + 255) % 256);
            p[0] = static_cast<uint8_t>(128.0f - 100.0f * std::fmax(-1.0f, std::fmin(1.0f, dx)));
            p[1] = static_cast<uint8_t>(128.0f - 100.0f * std::fmax(-1.0f, std::fmin(1.0f, dy)));
            p[2] = 255; p[3] = 255;
        }));
        // the reference binds the same *_Roughness image to both slots: roughness reads .g, metalness .b
        const float metal = i == 0 ? 1.0f : 0.0f;
        auto roughTexel = [=](uint32_t x, uint32_t y, uint8_t *p) {
            const float rough = i == 0 ? 0.2f + 0.3f * pattern(x, y) : 0.55f + 0.4f * pattern(x, y);
            p[0] = p[1] = static_cast<uint8_t>(255.0f * rough); p[2] = static_cast<uint8_t>(255.0f * metal); p[3] = 255;
        };
        m.RoughnessIdx = sb.AddTexture(MakeTexture(TextureType::Roughness, base + "_Roughness", 256, 256, roughTexel));
        m.MetallicIdx = sb.AddTexture(MakeTexture(TextureType::Metallic, base + "_Roughness (metallic)", 256, 256, roughTexel));
        materialIds[i] = sb.AddMaterial(names[i], m);
    }

    // three quads of the cube [-1,1]^3: +z, -x and +y faces
    auto &vertices = sb.GetVertices();
    auto &indices = sb.GetIndices();
    const Vec3 origin[3] = { Vec3(-1, -1, 1), Vec3(-1, -1, -1), Vec3(-1, 1, 1) };
    const Vec3 du[3] = { Vec3(2, 0, 0), Vec3(0, 0, 2), Vec3(2, 0, 0) }, dv[3] = { Vec3(0, 2, 0), Vec3(0, 2, 0), Vec3(0, 0, -2) };
    const Vec3 nrm[3] = { Vec3(0, 0, 1), Vec3(-1, 0, 0), Vec3(0, 1, 0) };
    uint32_t geometryIndices[3];
    for (int f = 0; f < 3; f++)
    {
        const uint32_t vertexOffset = static_cast<uint32_t>(vertices.size()), indexOffset = static_cast<uint32_t>(indices.size());
        const float cu[4] = { 0, 1, 1, 0 }, cv[4] = { 0, 0, 1, 1 };
        const Vec3 t = Normalize(du[f]), b = Normalize(dv[f]);
        for (int k = 0; k < 4; k++)
        {
            const Vec3 p = origin[f] + du[f] * cu[k] + dv[f] * cv[k];
            vertices.push_back(V({ p.x, p.y, p.z }, { cu[k], 1.0f - cv[k] }, { nrm[f].x, nrm[f].y, nrm[f].z }, { t.x, t.y, t.z }, { b.x, b.y, b.z }));
        }
        for (uint32_t k : { 0u, 1u, 2u, 2u, 3u, 0u })
            indices.push_back(k);
        geometryIndices[f] = sb.AddGeometry({ vertexOffset, 4, indexOffset, 6, true, false, { 0, 0 } });
    }

    const float halfTurn = 3.14159265358979f;
    const Vec3 axes[3] = { Vec3(1, 0, 0), Vec3(0, 1, 0), Vec3(0, 0, 1) };
    const Shaders::MaterialId faceMaterials[6] = { materialIds[1], materialIds[1], materialIds[1], materialIds[2], materialIds[2], materialIds[2] };
    std::array<MeshInfo, 6> meshes;
    for (int f = 0; f < 3; f++)
    {
        meshes[2 * f] = MI(geometryIndices[f], faceMaterials[2 * f]);
        meshes[2 * f + 1] = MI(geometryIndices[f], faceMaterials[2 * f + 1], ToTransform(Rotate(Mat4::Identity(), halfTurn, axes[f])));
    }
    const uint32_t cube = sb.AddModel(meshes);
    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    sb.AddModelInstance(cube, sb.AddSceneNode({ root, Mat4::Identity(), Mat4::Identity() }));
    // a second, metal copy beside it so all three materials are on screen
    const std::array<MeshInfo, 6> metalMeshes = { MI(geometryIndices[0], materialIds[0]), meshes[1], MI(geometryIndices[1], materialIds[0]), meshes[3],
                                                  MI(geometryIndices[2], materialIds[0]), meshes[5] };
    const Mat4 t2 = Rotate(Translate(Mat4::Identity(), Vec3(-3.0f, 0.0f, 0.5f)), 0.6f, Vec3(0, 1, 0));
    sb.AddModelInstance(sb.AddModel(metalMeshes), sb.AddSceneNode({ root, t2, Mat4::Identity() }));

    sb.SetSkybox(MakeSkyboxCube(256));
    AddViewCamera(sb, Vec3(2.6f, 2.2f, -4.6f), Vec3(-1.2f, 0.0f, 0.0f));
}

// ---------------------------------------------------------------------------
// Stand-in scenes
// ---------------------------------------------------------------------------

// C1 "DragonAttenuation": closed displaced sphere, Transmission 1, Ior 1.5, coloured
// Beer-Lambert attenuation, on a diffuse ground quad, default directional light.
void CreateAttenuationBlobScene(SceneBuilder &sb, float detail, uint32_t seed)
{
    Rng rng(seed);
    const float ph[6] = { rng.Range(0, 6.28f), rng.Range(0, 6.28f), rng.Range(0, 6.28f),
                          rng.Range(0, 6.28f), rng.Range(0, 6.28f), rng.Range(0, 6.28f) };
    const uint32_t segs = Scaled(320, detail, 12), rings = Scaled(160, detail, 6); // ~102k tris at detail 1
    const float pi = 3.14159265358979f;
    const uint32_t blob = AddGridSurface(sb, segs, rings, true, [&](float u, float v) {
        const float th = pi * (1.0f - v), phi = 2 * pi * u; // v=0 south pole ... v=1 north pole
        const Vec3 d(std::sin(th) * std::cos(phi), std::cos(th), std::sin(th) * std::sin(phi));
        const float r = 1.0f + 0.12f * std::sin(3 * d.x + ph[0]) * std::sin(4 * d.y + ph[1]) +
                        0.08f * std::sin(5 * d.z + ph[2]) * std::sin(2 * d.x + ph[3]) + 0.05f * std::sin(7 * d.y + ph[4]) * std::sin(6 * d.z + ph[5]);
        return d * r;
    });
    const uint32_t ground = AddBox(sb, Vec3(0, -0.05f, 0), Vec3(8, 0.05f, 8));

    auto glass = DefaultMaterialInfo();
    glass.Roughness = 0.0f;
    glass.Transmission = 1.0f;
    glass.Ior = 1.5f;
    glass.AttenuationColor[0] = 0.9f; glass.AttenuationColor[1] = 0.4f; glass.AttenuationColor[2] = 0.2f;
    glass.AttenuationDistance = 0.5f;
    const auto glassId = sb.AddMaterial("Blob Glass", glass);
    const auto groundId = sb.AddMaterial("Ground", MakeMaterial(Vec3(0.8f, 0.8f, 0.8f), 1.0f, 0.0f));

    const std::array<MeshInfo, 1> blobMesh = { MI(blob, glassId) };
    const std::array<MeshInfo, 1> groundMesh = { MI(ground, groundId) };
    const uint32_t blobModel = sb.AddModel(blobMesh), groundModel = sb.AddModel(groundMesh);
    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const uint32_t blobNode = sb.AddSceneNode({ root, Translate(Mat4::Identity(), Vec3(0, 1.3f, 0)), Mat4::Identity() });
    sb.AddModelInstance(groundModel, root);
    sb.AddModelInstance(blobModel, blobNode);
    AddViewCamera(sb, Vec3(3.2f, 2.4f, -3.0f), Vec3(0, 1.2f, 0));
}

// C2 "ABeautifulGame": 32 instanced lathe pieces on an 8x8 board, ~2 M triangles through
// instancing, MR materials roughness U[0.05,0.8], metalness in {0,1}, two glass pieces.
void CreateChessLikeScene(SceneBuilder &sb, float detail, uint32_t seed)
{
    Rng rng(seed);
    const uint32_t segs = Scaled(256, detail, 8), rings = Scaled(122, detail, 6); // 62,464 tris per piece at detail 1
    const std::vector<std::vector<Vec2>> profiles = {
        // pawn, rook, knight-ish, bishop, queen, king: (radius, height)
        { { 0.0f, 0.0f }, { 0.36f, 0.0f }, { 0.38f, 0.08f }, { 0.22f, 0.2f }, { 0.14f, 0.5f }, { 0.2f, 0.62f }, { 0.21f, 0.75f }, { 0.12f, 0.88f }, { 0.0f, 0.92f } },
        { { 0.0f, 0.0f }, { 0.4f, 0.0f }, { 0.42f, 0.1f }, { 0.27f, 0.25f }, { 0.25f, 0.8f }, { 0.34f, 0.9f }, { 0.34f, 1.1f }, { 0.2f, 1.1f }, { 0.0f, 1.0f } },
        { { 0.0f, 0.0f }, { 0.4f, 0.0f }, { 0.42f, 0.1f }, { 0.24f, 0.3f }, { 0.3f, 0.7f }, { 0.36f, 0.95f }, { 0.24f, 1.2f }, { 0.1f, 1.3f }, { 0.0f, 1.32f } },
        { { 0.0f, 0.0f }, { 0.4f, 0.0f }, { 0.42f, 0.1f }, { 0.2f, 0.3f }, { 0.14f, 0.9f }, { 0.24f, 1.05f }, { 0.2f, 1.3f }, { 0.06f, 1.45f }, { 0.0f, 1.5f } },
        { { 0.0f, 0.0f }, { 0.44f, 0.0f }, { 0.46f, 0.12f }, { 0.24f, 0.35f }, { 0.16f, 1.1f }, { 0.3f, 1.3f }, { 0.34f, 1.5f }, { 0.12f, 1.62f }, { 0.0f, 1.7f } },
        { { 0.0f, 0.0f }, { 0.46f, 0.0f }, { 0.48f, 0.12f }, { 0.26f, 0.4f }, { 0.18f, 1.2f }, { 0.32f, 1.45f }, { 0.3f, 1.65f }, { 0.1f, 1.8f }, { 0.0f, 1.9f } },
    };
    std::vector<uint32_t> pieceGeometry;
    for (const auto &p : profiles)
        pieceGeometry.push_back(AddLathe(sb, p, segs, rings));

    // board: light and dark squares as two geometries of 32 quads each + a frame box
    uint32_t squares[2];
    for (int colour = 0; colour < 2; colour++)
    {
        auto &vertices = sb.GetVertices();
        auto &indices = sb.GetIndices();
        const uint32_t vo = static_cast<uint32_t>(vertices.size()), io = static_cast<uint32_t>(indices.size());
        uint32_t q = 0;
        for (int z = 0; z < 8; z++)
            for (int x = 0; x < 8; x++)
            {
                if (((x + z) & 1) != colour)
                    continue;
                const float x0 = static_cast<float>(x) - 4.0f, z0 = static_cast<float>(z) - 4.0f;
                vertices.push_back(V({ x0, 0, z0 + 1 }, { 0, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }));
                vertices.push_back(V({ x0 + 1, 0, z0 + 1 }, { 0, 1 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }));
                vertices.push_back(V({ x0 + 1, 0, z0 }, { 1, 1 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }));
                vertices.push_back(V({ x0, 0, z0 }, { 1, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }));
                for (uint32_t k : { 0u, 1u, 2u, 2u, 3u, 0u })
                    indices.push_back(q * 4 + k);
                q++;
            }
        squares[colour] = sb.AddGeometry({ vo, q * 4, io, q * 6, true, false, { 0, 0 } });
    }
    const uint32_t frame = AddBox(sb, Vec3(0, -0.26f, 0), Vec3(4.6f, 0.25f, 4.6f));
    const uint32_t table = AddBox(sb, Vec3(0, -0.6f, 0), Vec3(30.0f, 0.08f, 30.0f));

    const auto lightSq = sb.AddMaterial("Light Square", MakeMaterial(Vec3(0.85f, 0.8f, 0.7f), 0.35f, 0.0f));
    const auto darkSq = sb.AddMaterial("Dark Square", MakeMaterial(Vec3(0.12f, 0.1f, 0.09f), 0.25f, 0.0f));
    const auto frameMat = sb.AddMaterial("Frame", MakeMaterial(Vec3(0.3f, 0.18f, 0.1f), 0.6f, 0.0f));
    const auto tableMat = sb.AddMaterial("Table", MakeMaterial(Vec3(0.55f, 0.55f, 0.6f), 0.9f, 0.0f));
    const std::array<MeshInfo, 4> boardMeshes = { MI(squares[0], lightSq), MI(squares[1], darkSq), MI(frame, frameMat), MI(table, tableMat) };
    const uint32_t boardModel = sb.AddModel(boardMeshes);

    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    sb.AddModelInstance(boardModel, root);

    const int backRank[8] = { 1, 2, 3, 4, 5, 3, 2, 1 };
    int pieceNo = 0;
    for (int side = 0; side < 2; side++)
        for (int rank = 0; rank < 2; rank++)
            for (int file = 0; file < 8; file++, pieceNo++)
            {
                const int type = rank == 0 ? backRank[file] : 0;
                Shaders::MetallicRoughnessMaterial m;
                if (pieceNo == 3 || pieceNo == 20) // two glass pieces
                {
                    m = DefaultMaterialInfo();
                    m.Color[0] = 0.9f; m.Color[1] = 0.95f; m.Color[2] = 1.0f;
                    m.Roughness = 0.0f;
                    m.Transmission = 1.0f;
                }
                else
                {
                    const float metal = rng.Next() < 0.5f ? 1.0f : 0.0f;
                    const Vec3 base = side == 0 ? Vec3(0.9f, 0.87f, 0.8f) : Vec3(0.2f, 0.16f, 0.14f);
                    const Vec3 tint(rng.Range(0.85f, 1.0f), rng.Range(0.85f, 1.0f), rng.Range(0.85f, 1.0f));
                    m = MakeMaterial(Vec3(base.x * tint.x, base.y * tint.y, base.z * tint.z), rng.Range(0.05f, 0.8f), metal);
                }
                const auto matId = sb.AddMaterial("Piece " + std::to_string(pieceNo), m);
                const std::array<MeshInfo, 1> mesh = { MI(pieceGeometry[static_cast<size_t>(type)], matId) };
                const uint32_t model = sb.AddModel(mesh);
                const float z = side == 0 ? (rank == 0 ? -3.5f : -2.5f) : (rank == 0 ? 3.5f : 2.5f);
                const Mat4 t = Rotate(Translate(Mat4::Identity(), Vec3(static_cast<float>(file) - 3.5f, 0.0f, z)), rng.Range(0, 6.28f), Vec3(0, 1, 0));
                const uint32_t node = sb.AddSceneNode({ root, t, Mat4::Identity() });
                sb.AddModelInstance(model, node);
            }
    AddViewCamera(sb, Vec3(6.5f, 5.0f, -7.5f), Vec3(0.0f, 0.4f, 0.0f));
}

// C3 "Sun Temple": closed interior with instanced columns and beams, 12 materials,
// 8 point lights and emissive braziers.
void CreateTempleLikeScene(SceneBuilder &sb, float detail, uint32_t seed)
{
    Rng rng(seed);
    const uint32_t segs = Scaled(160, detail, 8), rings = Scaled(72, detail, 6); // 23,040 tris per column
    const std::vector<Vec2> columnProfile = { { 0.0f, 0.0f }, { 0.7f, 0.0f }, { 0.7f, 0.3f }, { 0.5f, 0.5f }, { 0.45f, 3.0f }, { 0.42f, 5.4f }, { 0.6f, 5.7f }, { 0.7f, 6.0f }, { 0.0f, 6.0f } };
    const uint32_t column = AddLathe(sb, columnProfile, segs, rings);
    const std::vector<Vec2> bowlProfile = { { 0.0f, 0.0f }, { 0.3f, 0.0f }, { 0.12f, 0.3f }, { 0.12f, 0.9f }, { 0.5f, 1.2f }, { 0.55f, 1.3f }, { 0.0f, 1.25f } };
    const uint32_t bowl = AddLathe(sb, bowlProfile, Scaled(64, detail, 8), Scaled(32, detail, 6));
    const uint32_t flame = AddBox(sb, Vec3(0, 1.4f, 0), Vec3(0.18f, 0.12f, 0.18f));

    const uint32_t floor = AddBox(sb, Vec3(0, -0.1f, 0), Vec3(16, 0.1f, 10));
    const uint32_t ceiling = AddBox(sb, Vec3(0, 6.6f, 0), Vec3(16, 0.1f, 10));
    const uint32_t wallN = AddBox(sb, Vec3(0, 3.25f, 10.1f), Vec3(16, 3.45f, 0.1f));
    const uint32_t wallS = AddBox(sb, Vec3(0, 3.25f, -10.1f), Vec3(16, 3.45f, 0.1f));
    const uint32_t wallE = AddBox(sb, Vec3(16.1f, 3.25f, 0), Vec3(0.1f, 3.45f, 10.2f));
    const uint32_t wallW = AddBox(sb, Vec3(-16.1f, 3.25f, 0), Vec3(0.1f, 3.45f, 10.2f));
    const uint32_t beam = AddBox(sb, Vec3(0, 6.25f, 0), Vec3(0.5f, 0.25f, 10));
    // relief panels: finely tessellated wavy sheets along the walls
    const uint32_t relief = AddGridSurface(sb, Scaled(240, detail, 8), Scaled(120, detail, 4), false, [&](float u, float v) {
        return Vec3(-15.0f + 30.0f * u, 0.5f + 5.5f * v, 9.9f - 0.15f * std::sin(40 * u) * std::sin(25 * v));
    });

    Shaders::MaterialId mats[12];
    for (int i = 0; i < 12; i++)
    {
        const Vec3 c(rng.Range(0.35f, 0.9f), rng.Range(0.3f, 0.8f), rng.Range(0.25f, 0.7f));
        mats[i] = sb.AddMaterial("Temple " + std::to_string(i), MakeMaterial(c, rng.Range(0.15f, 1.0f), i % 4 == 3 ? 1.0f : 0.0f));
    }
    auto emissive = DefaultMaterialInfo();
    emissive.EmissiveColor[0] = 1.0f; emissive.EmissiveColor[1] = 0.55f; emissive.EmissiveColor[2] = 0.2f;
    emissive.EmissiveIntensity = 6.0f;
    const auto flameMat = sb.AddMaterial("Flame", emissive);

    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const std::array<MeshInfo, 8> shell = { MI(floor, mats[0]), MI(ceiling, mats[1]), MI(wallN, mats[2]), MI(wallS, mats[2]),
                                            MI(wallE, mats[4]), MI(wallW, mats[4]), MI(relief, mats[5]), MI(relief, mats[6], ToTransform(Scale(Mat4::Identity(), Vec3(1, 1, -1)))) };
    sb.AddModelInstance(sb.AddModel(shell), root);

    uint32_t columnModels[3];
    for (int k = 0; k < 3; k++)
    {
        const std::array<MeshInfo, 1> m = { MI(column, mats[7 + k]) };
        columnModels[k] = sb.AddModel(m);
    }
    const std::array<MeshInfo, 1> beamMesh = { MI(beam, mats[10]) };
    const uint32_t beamModel = sb.AddModel(beamMesh);
    const std::array<MeshInfo, 2> brazierMeshes = { MI(bowl, mats[11]), MI(flame, flameMat) };
    const uint32_t brazierModel = sb.AddModel(brazierMeshes);

    int n = 0;
    for (int i = 0; i < 12; i++)
        for (int side = 0; side < 2; side++, n++)
        {
            const float x = -13.75f + 2.5f * static_cast<float>(i), z = side ? 5.0f : -5.0f;
            const uint32_t node = sb.AddSceneNode({ root, Translate(Mat4::Identity(), Vec3(x, 0, z)), Mat4::Identity() });
            sb.AddModelInstance(columnModels[n % 3], node);
        }
    for (int i = 0; i < 12; i++)
    {
        const uint32_t node = sb.AddSceneNode({ root, Translate(Mat4::Identity(), Vec3(-13.75f + 2.5f * static_cast<float>(i), 0, 0)), Mat4::Identity() });
        sb.AddModelInstance(beamModel, node);
    }
    for (int i = 0; i < 8; i++)
    {
        const Vec3 p(-12.5f + 25.0f * static_cast<float>(i % 4) / 3.0f, 0.0f, i < 4 ? -2.0f : 2.0f);
        const uint32_t node = sb.AddSceneNode({ root, Translate(Mat4::Identity(), p), Mat4::Identity() });
        sb.AddModelInstance(brazierModel, node);
        sb.AddLight(MakePointLight(Vec3(5.0f, 3.0f, 1.25f), Vec3(p.x, 2.2f, p.z)), root);
    }
    Shaders::DirectionalLight dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.Direction[1] = -1.0f;
    sb.SetDirectionalLight(std::move(dl), root);
    AddViewCamera(sb, Vec3(-14.0f, 2.2f, -1.5f), Vec3(6.0f, 2.6f, 1.0f), 60.0f);
}

// C4 "Intel Sponza": two-storey arcade around an open atrium, curtain sheets and ivy
// leaf cards (opaque until alpha testing lands with N1), ~4 M triangles, one
// directional light through the open roof.
// multiplies the texture coordinates of the vertices added since `firstVertex` (tiling of a procedural texture)
void ScaleTexCoords(SceneBuilder &sb, size_t firstVertex, float su, float sv)
{
    auto &vertices = sb.GetVertices();
    for (size_t k = firstVertex; k < vertices.size(); k++)
    {
        vertices[k].TexCoords[0] *= su;
        vertices[k].TexCoords[1] *= sv;
    }
}

void CreateAtriumLikeScene(SceneBuilder &sb, float detail, uint32_t seed)
{
    Rng rng(seed);
    const uint32_t segs = Scaled(192, detail, 8), rings = Scaled(96, detail, 6); // 36,864 tris per column
    const std::vector<Vec2> columnProfile = { { 0.0f, 0.0f }, { 0.55f, 0.0f }, { 0.55f, 0.25f }, { 0.4f, 0.4f }, { 0.36f, 2.0f }, { 0.33f, 3.5f }, { 0.5f, 3.8f }, { 0.55f, 4.0f }, { 0.0f, 4.0f } };
    const uint32_t column = AddLathe(sb, columnProfile, segs, rings);
    size_t mark = sb.GetVertices().size();
    const uint32_t floor = AddGridSurface(sb, Scaled(600, detail, 4), Scaled(300, detail, 4), false, [&](float u, float v) {
        return Vec3(-20.0f + 40.0f * u, 0.02f * std::sin(60 * u) * std::sin(45 * v), -10.0f + 20.0f * v);
    });
    ScaleTexCoords(sb, mark, 10.0f, 5.0f); // 4 x 4 tiles per texture repeat, 1 m tiles
    const uint32_t gallery = AddBox(sb, Vec3(0, 4.2f, 0), Vec3(20, 0.2f, 2.5f));
    const uint32_t wall = AddBox(sb, Vec3(0, 5.0f, 0), Vec3(20.2f, 5.0f, 0.2f));
    const uint32_t endWall = AddBox(sb, Vec3(0, 5.0f, 0), Vec3(0.2f, 5.0f, 10.2f));
    mark = sb.GetVertices().size();
    const uint32_t curtain = AddGridSurface(sb, Scaled(250, detail, 4), Scaled(200, detail, 4), false, [&](float u, float v) {
        return Vec3(3.0f * u, 3.6f * v, 0.18f * std::sin(25 * u + 3 * v) * (0.3f + v)); // 50k quads each
    });
    ScaleTexCoords(sb, mark, 30.0f, 36.0f); // 1 cm threads
    // ivy: alpha-tested leaf cards (non-opaque geometry, anyhit.rahit), like the foliage of the Sponza / Bistro assets
    const uint32_t ivy = AddCards(sb, rng, Scaled(250000, detail, 16), Vec3(-19.5f, 0.2f, 9.2f), Vec3(19.5f, 9.5f, 9.75f), 0.09f, false);
    const uint32_t ivy2 = AddCards(sb, rng, Scaled(250000, detail, 16), Vec3(19.2f, 0.2f, -9.5f), Vec3(19.75f, 9.5f, 9.5f), 0.09f, false);

    const auto stone = sb.AddMaterial("Stone", MakeMaterial(Vec3(0.72f, 0.68f, 0.6f), 0.85f, 0.0f));
    const auto stone2 = sb.AddMaterial("Stone 2", MakeMaterial(Vec3(0.6f, 0.55f, 0.5f), 0.7f, 0.0f));
    auto floorM = MakeMaterial(Vec3(1.0f), 0.4f, 0.0f);
    floorM.ColorIdx = sb.AddTexture(MakeTexture(TextureType::Color, "Atrium Tiles", 256, 256, [seed](uint32_t x, uint32_t y, uint8_t *p) {
        const bool joint = x % 64 < 2 || y % 64 < 2;
        const uint32_t h = HashU(x / 64, y / 64, seed) & 31;
        const uint8_t v = joint ? 60 : static_cast<uint8_t>(150 + h);
        p[0] = v; p[1] = static_cast<uint8_t>(v * 9 / 10); p[2] = static_cast<uint8_t>(v * 8 / 10); p[3] = 255;
    }));
    const auto floorMat = sb.AddMaterial("Atrium Floor", floorM);
    auto leafM = MakeMaterial(Vec3(1.0f), 0.6f, 0.0f);
    leafM.ColorIdx = sb.AddTexture(MakeTexture(TextureType::Color, "Ivy Leaf", 64, 64, [](uint32_t x, uint32_t y, uint8_t *p) {
        const float dx = (static_cast<float>(x) - 31.5f) / 31.5f, dy = (static_cast<float>(y) - 31.5f) / 22.0f;
        const float r = dx * dx + dy * dy;
        const bool vein = (x % 10 == 4) || std::abs(static_cast<int>(y) - 32) < 2;
        p[0] = vein ? 70 : 38; p[1] = vein ? 150 : 115; p[2] = vein ? 50 : 30;
        p[3] = r < 0.75f ? 255 : (r < 1.0f ? 200 : 0);
    }));
    const auto leaf = sb.AddMaterial("Leaf", leafM);
    const Vec3 curtainColours[4] = { { 0.7f, 0.1f, 0.1f }, { 0.1f, 0.2f, 0.65f }, { 0.1f, 0.5f, 0.2f }, { 0.75f, 0.6f, 0.2f } };
    const uint32_t weave = sb.AddTexture(MakeTexture(TextureType::Color, "Weave", 32, 32, [](uint32_t x, uint32_t y, uint8_t *p) {
        const bool over = ((x / 4) + (y / 4)) % 2 == 0;
        const uint8_t v = static_cast<uint8_t>((over ? 235 : 170) - ((x % 4 == 0 || y % 4 == 0) ? 50 : 0));
        p[0] = p[1] = p[2] = v; p[3] = 255;
    }));
    Shaders::MaterialId curtainMats[4];
    for (int i = 0; i < 4; i++)
    {
        auto m = MakeMaterial(curtainColours[i], 0.9f, 0.0f);
        m.ColorIdx = weave;
        curtainMats[i] = sb.AddMaterial("Curtain " + std::to_string(i), m);
    }
    const auto bronze = sb.AddMaterial("Bronze", MakeMaterial(Vec3(0.8f, 0.5f, 0.25f), 0.3f, 1.0f));

    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const std::array<MeshInfo, 9> shell = {
        MI(floor, floorMat),
        MI(gallery, stone2, ToTransform(Translate(Mat4::Identity(), Vec3(0, 0, 7.5f)))),
        MI(gallery, stone2, ToTransform(Translate(Mat4::Identity(), Vec3(0, 0, -7.5f)))),
        MI(wall, stone, ToTransform(Translate(Mat4::Identity(), Vec3(0, 0, 10.2f)))),
        MI(wall, stone, ToTransform(Translate(Mat4::Identity(), Vec3(0, 0, -10.2f)))),
        MI(endWall, stone, ToTransform(Translate(Mat4::Identity(), Vec3(20.2f, 0, 0)))),
        MI(endWall, stone, ToTransform(Translate(Mat4::Identity(), Vec3(-20.2f, 0, 0)))),
        MI(ivy, leaf),
        MI(ivy2, leaf),
    };
    sb.AddModelInstance(sb.AddModel(shell), root);

    const std::array<MeshInfo, 1> colMesh = { MI(column, stone) }, colMeshB = { MI(column, bronze) };
    const uint32_t colModel = sb.AddModel(colMesh), colModelB = sb.AddModel(colMeshB);
    int n = 0;
    for (int storey = 0; storey < 2; storey++)
        for (int side = 0; side < 2; side++)
            for (int i = 0; i < 16; i++, n++)
            {
                const Vec3 p(-18.75f + 2.5f * static_cast<float>(i), storey ? 4.4f : 0.0f, side ? 5.2f : -5.2f);
                const uint32_t node = sb.AddSceneNode({ root, Translate(Mat4::Identity(), p), Mat4::Identity() });
                sb.AddModelInstance(n % 7 == 3 ? colModelB : colModel, node);
            }
    uint32_t curtainModels[4];
    for (int i = 0; i < 4; i++)
    {
        const std::array<MeshInfo, 1> m = { MI(curtain, curtainMats[i]) };
        curtainModels[i] = sb.AddModel(m);
    }
    for (int i = 0; i < 8; i++) // 8 x 50k quads = 400k quads... trimmed to the 200k-quad budget by 4 per side
    {
        if (i >= 4 && detail >= 1.0f)
            break;
        const Vec3 p(-15.0f + 8.0f * static_cast<float>(i % 4), 4.5f, (i & 1) ? 5.0f : -5.0f);
        const uint32_t node = sb.AddSceneNode({ root, Translate(Mat4::Identity(), p), Mat4::Identity() });
        sb.AddModelInstance(curtainModels[i % 4], node);
    }
    Shaders::DirectionalLight dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.Color[0] = 12.0f; dl.Color[1] = 11.0f; dl.Color[2] = 9.5f;
    dl.Direction[0] = -0.25f; dl.Direction[1] = -1.0f; dl.Direction[2] = 0.18f;
    sb.SetDirectionalLight(std::move(dl), root);
    AddViewCamera(sb, Vec3(-17.0f, 1.8f, -0.8f), Vec3(5.0f, 3.4f, 1.2f), 60.0f);
}

// C5 "Bistro night": street canyon with a tessellated cobble road, box buildings,
// 64 point lights (the MaxLightCount cap) and 300 small emissive quads; no sun.
void CreateStreetLikeScene(SceneBuilder &sb, float detail, uint32_t seed)
{
    Rng rng(seed);
    const uint32_t road = AddGridSurface(sb, Scaled(1400, detail, 8), Scaled(700, detail, 4), false, [&](float u, float v) {
        return Vec3(-40.0f + 80.0f * u, 0.03f * std::sin(220 * u) * std::sin(160 * v), -8.0f + 16.0f * v); // 1.96 M tris
    });
    const std::vector<Vec2> postProfile = { { 0.0f, 0.0f }, { 0.25f, 0.0f }, { 0.1f, 0.3f }, { 0.07f, 3.6f }, { 0.3f, 3.9f }, { 0.3f, 4.1f }, { 0.0f, 4.2f } };
    const uint32_t post = AddLathe(sb, postProfile, Scaled(96, detail, 8), Scaled(80, detail, 6)); // 15,360 tris
    const uint32_t awning = AddGridSurface(sb, Scaled(120, detail, 4), Scaled(60, detail, 4), false, [&](float u, float v) {
        return Vec3(4.0f * u, 3.0f - 0.8f * v + 0.05f * std::sin(30 * u), 1.6f * v);
    });
    const uint32_t unitBox = AddBox(sb, Vec3(0, 0.5f, 0), Vec3(0.5f, 0.5f, 0.5f));
    const uint32_t sign = AddCards(sb, rng, 1, Vec3(0, 0, 0), Vec3(0, 0, 0), 0.25f);

    Shaders::MaterialId facade[6];
    for (int i = 0; i < 6; i++)
        facade[i] = sb.AddMaterial("Facade " + std::to_string(i), MakeMaterial(Vec3(rng.Range(0.3f, 0.8f), rng.Range(0.3f, 0.7f), rng.Range(0.3f, 0.7f)), rng.Range(0.4f, 1.0f), 0.0f));
    const auto roadMat = sb.AddMaterial("Cobble", MakeMaterial(Vec3(0.25f, 0.25f, 0.27f), 0.35f, 0.0f));
    const auto metalMat = sb.AddMaterial("Post", MakeMaterial(Vec3(0.4f, 0.4f, 0.42f), 0.3f, 1.0f));
    const auto clothMat = sb.AddMaterial("Awning", MakeMaterial(Vec3(0.6f, 0.15f, 0.12f), 0.9f, 0.0f));
    auto glass = DefaultMaterialInfo();
    glass.Roughness = 0.05f;
    glass.Transmission = 1.0f;
    const auto windowMat = sb.AddMaterial("Window", glass);

    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const std::array<MeshInfo, 1> roadMesh = { MI(road, roadMat) };
    sb.AddModelInstance(sb.AddModel(roadMesh), root);

    uint32_t boxModels[7];
    for (int i = 0; i < 6; i++)
    {
        const std::array<MeshInfo, 1> m = { MI(unitBox, facade[i]) };
        boxModels[i] = sb.AddModel(m);
    }
    const std::array<MeshInfo, 1> wm = { MI(unitBox, windowMat) };
    boxModels[6] = sb.AddModel(wm);
    for (int side = 0; side < 2; side++)
    {
        float x = -40.0f;
        while (x < 40.0f)
        {
            const float w = rng.Range(4.0f, 9.0f), h = rng.Range(6.0f, 16.0f), d = rng.Range(4.0f, 8.0f);
            const float z = side ? 8.0f + 0.5f * d : -8.0f - 0.5f * d;
            Mat4 t = Scale(Translate(Mat4::Identity(), Vec3(x + 0.5f * w, 0, z)), Vec3(w, h, d));
            sb.AddModelInstance(boxModels[rng.NextU() % 6], sb.AddSceneNode({ root, t, Mat4::Identity() }));
            // protruding window boxes
            const int nw = static_cast<int>(w / 1.6f), nf = static_cast<int>(h / 3.0f);
            for (int f = 0; f < nf; f++)
                for (int k = 0; k < nw; k++)
                {
                    const Vec3 p(x + 0.8f + 1.6f * static_cast<float>(k), 1.2f + 3.0f * static_cast<float>(f), side ? 8.0f - 0.05f : -8.0f + 0.05f);
                    Mat4 wt = Scale(Translate(Mat4::Identity(), p), Vec3(0.9f, 1.4f, 0.12f));
                    sb.AddModelInstance(boxModels[6], sb.AddSceneNode({ root, wt, Mat4::Identity() }));
                }
            x += w + rng.Range(0.0f, 0.6f);
        }
    }
    const std::array<MeshInfo, 1> postMesh = { MI(post, metalMat) }, awningMesh = { MI(awning, clothMat) };
    const uint32_t postModel = sb.AddModel(postMesh), awningModel = sb.AddModel(awningMesh);
    for (int i = 0; i < 64; i++)
    {
        const Vec3 p(-39.0f + 78.0f * static_cast<float>(i / 2) / 31.0f, 0.0f, (i & 1) ? 6.5f : -6.5f);
        sb.AddModelInstance(postModel, sb.AddSceneNode({ root, Translate(Mat4::Identity(), p), Mat4::Identity() }));
        sb.AddLight(MakePointLight(Vec3(rng.Range(2.0f, 4.0f), rng.Range(1.5f, 3.0f), rng.Range(0.8f, 2.0f)), Vec3(p.x, 4.4f, p.z * 0.92f)), root);
    }
    for (int i = 0; i < 12; i++)
    {
        const Vec3 p(-36.0f + 6.5f * static_cast<float>(i), 0.0f, (i & 1) ? 6.3f : -7.9f);
        sb.AddModelInstance(awningModel, sb.AddSceneNode({ root, Translate(Mat4::Identity(), p), Mat4::Identity() }));
    }
    for (int i = 0; i < 300; i++) // small emissive signs / lit windows
    {
        auto em = DefaultMaterialInfo();
        em.EmissiveColor[0] = rng.Range(0.3f, 1.0f); em.EmissiveColor[1] = rng.Range(0.3f, 1.0f); em.EmissiveColor[2] = rng.Range(0.2f, 1.0f);
        em.EmissiveIntensity = rng.Range(2.0f, 8.0f);
        const auto id = sb.AddMaterial("Sign " + std::to_string(i % 24), em);
        const std::array<MeshInfo, 1> m = { MI(sign, id) };
        const uint32_t model = sb.AddModel(m);
        const bool side = rng.Next() < 0.5f;
        const Vec3 p(rng.Range(-39.0f, 39.0f), rng.Range(1.0f, 9.0f), side ? 7.8f : -7.8f);
        Mat4 t = Rotate(Translate(Mat4::Identity(), p), rng.Range(0, 6.28f), Vec3(rng.Range(-1, 1), rng.Range(-1, 1), rng.Range(-1, 1)));
        sb.AddModelInstance(model, sb.AddSceneNode({ root, t, Mat4::Identity() }));
    }
    Shaders::DirectionalLight dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.Direction[1] = -1.0f;
    sb.SetDirectionalLight(std::move(dl), root);
    AddViewCamera(sb, Vec3(-34.0f, 1.7f, 1.0f), Vec3(10.0f, 3.5f, -0.5f), 60.0f);
}

// ---------------------------------------------------------------------------
// Procedural textures (row N1): there is no image decoder in this tree yet (N2), so textured
// test content is generated.  TextureImporter.cpp:24-51: texels with alpha 0 get rgb 0.
// ---------------------------------------------------------------------------

TextureInfo MakeTexture(TextureType type, const std::string &name, uint32_t w, uint32_t h,
                        const std::function<void(uint32_t, uint32_t, uint8_t *)> &texel)
{
    TextureInfo t;
    t.Type = type;
    t.Width = w;
    t.Height = h;
    t.Name = name;
    t.Format = TextureFormat::RGBAU8;
    t.Pixels.resize(static_cast<size_t>(w) * h * 4);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
        {
            uint8_t *p = &t.Pixels[(static_cast<size_t>(y) * w + x) * 4];
            texel(x, y, p);
            if (p[3] == 0)
                p[0] = p[1] = p[2] = 0;
        }
    return t;
}

uint32_t HashU(uint32_t x, uint32_t y, uint32_t seed)
{
    uint32_t h = x * 374761393u + y * 668265263u + seed * 2246822519u;
    h = (h ^ (h >> 13)) * 1274126177u;
    return h ^ (h >> 16);
}

// small scene that exercises the sampler: square / non-square / non-power-of-two sizes, sRGB and
// UNORM types, a float texture, tiled and magnified uv ranges
void CreateTextureTestScene(SceneBuilder &sb, uint32_t seed)
{
    const uint32_t checker = sb.AddTexture(MakeTexture(TextureType::Color, "Checker", 64, 64, [](uint32_t x, uint32_t y, uint8_t *p) {
        const bool on = ((x / 8) + (y / 8)) & 1;
        p[0] = on ? 230 : 30; p[1] = on ? 200 : 40; p[2] = on ? 60 : 150; p[3] = 255;
    }));
    const uint32_t noise = sb.AddTexture(MakeTexture(TextureType::Color, "Noise", 37, 21, [seed](uint32_t x, uint32_t y, uint8_t *p) {
        const uint32_t h = HashU(x, y, seed);
        p[0] = h & 255; p[1] = (h >> 8) & 255; p[2] = (h >> 16) & 255; p[3] = 255;
    }));
    const uint32_t rough = sb.AddTexture(MakeTexture(TextureType::Roughness, "Roughness Stripes", 128, 8, [](uint32_t x, uint32_t, uint8_t *p) {
        p[0] = 0; p[1] = static_cast<uint8_t>(40 + (x * 200) / 127); p[2] = 0; p[3] = 255;
    }));
    const uint32_t metal = sb.AddTexture(MakeTexture(TextureType::Metallic, "Metal Dots", 32, 32, [](uint32_t x, uint32_t y, uint8_t *p) {
        const int dx = static_cast<int>(x % 16) - 8, dy = static_cast<int>(y % 16) - 8;
        p[0] = p[1] = 0; p[2] = (dx * dx + dy * dy < 30) ? 255 : 0; p[3] = 255;
    }));
    const uint32_t bump = sb.AddTexture(MakeTexture(TextureType::Normal, "Bumps", 64, 64, [](uint32_t x, uint32_t y, uint8_t *p) {
        const float fx = std::sin(static_cast<float>(x) * 0.3927f), fy = std::sin(static_cast<float>(y) * 0.3927f);
        p[0] = static_cast<uint8_t>(128.0f + 90.0f * fx); p[1] = static_cast<uint8_t>(128.0f + 90.0f * fy); p[2] = 255; p[3] = 255;
    }));
    const uint32_t glow = sb.AddTexture(MakeTexture(TextureType::Emisive, "Glow Grid", 16, 16, [](uint32_t x, uint32_t y, uint8_t *p) {
        const bool on = (x % 4 == 0) || (y % 4 == 0);
        p[0] = on ? 255 : 0; p[1] = on ? 120 : 0; p[2] = on ? 30 : 0; p[3] = 255;
    }));
    TextureInfo hdr; // RGBA32F
    hdr.Type = TextureType::Color;
    hdr.Name = "Float Ramp";
    hdr.Width = 16;
    hdr.Height = 4;
    hdr.Format = TextureFormat::RGBAF32;
    hdr.Pixels.resize(16 * 4 * 16);
    for (uint32_t y = 0; y < 4; y++)
        for (uint32_t x = 0; x < 16; x++)
        {
            const float v[4] = { 0.05f + 0.06f * static_cast<float>(x), 0.2f + 0.2f * static_cast<float>(y), 0.9f - 0.05f * static_cast<float>(x), 1.0f };
            std::memcpy(&hdr.Pixels[(static_cast<size_t>(y) * 16 + x) * 16], v, 16);
        }
    const uint32_t ramp = sb.AddTexture(std::move(hdr));

    auto textured = [&](const char *name, uint32_t colorIdx, float roughness, float metalness) {
        auto m = MakeMaterial(Vec3(1.0f), roughness, metalness);
        m.ColorIdx = colorIdx;
        return sb.AddMaterial(name, m);
    };
    const auto floorMat = textured("Checker Floor", checker, 0.9f, 0.0f);
    const auto noiseMat = textured("Noise", noise, 0.6f, 0.0f);
    const auto rampMat = textured("Ramp", ramp, 0.4f, 0.0f);
    auto pbr = MakeMaterial(Vec3(0.9f, 0.85f, 0.8f), 1.0f, 1.0f);
    pbr.RoughnessIdx = rough;
    pbr.MetallicIdx = metal;
    pbr.NormalIdx = bump;
    const auto pbrMat = sb.AddMaterial("PBR Maps", pbr);
    auto lamp = MakeMaterial(Vec3(0.2f), 0.8f, 0.0f);
    lamp.EmissiveIdx = glow;
    lamp.EmissiveIntensity = 4.0f;
    const auto lampMat = sb.AddMaterial("Lamp", lamp);

    // floor: uv tiled 6x (minification at the far end), built as a grid so uv varies per vertex
    const uint32_t floor = AddGridSurface(sb, 8, 8, false, [](float u, float v) { return Vec3(-6.0f + 12.0f * u, 0.0f, -6.0f + 12.0f * v); });
    {
        auto &vertices = sb.GetVertices();
        for (size_t k = vertices.size() - 81; k < vertices.size(); k++)
        {
            vertices[k].TexCoords[0] *= 6.0f;
            vertices[k].TexCoords[1] *= 6.0f;
        }
    }
    const std::array<uint32_t, 6> cube = AddCube(sb);
    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const std::array<MeshInfo, 1> floorMesh = { MI(floor, floorMat) };
    sb.AddModelInstance(sb.AddModel(floorMesh), root);
    const Shaders::MaterialId mats[4] = { noiseMat, rampMat, pbrMat, lampMat };
    for (int i = 0; i < 4; i++)
    {
        std::array<MeshInfo, 6> meshes;
        for (int k = 0; k < 6; k++)
            meshes[k] = MI(cube[k], mats[i]);
        const uint32_t model = sb.AddModel(meshes);
        const Mat4 t = Scale(Rotate(Translate(Mat4::Identity(), Vec3(-3.0f + 2.0f * static_cast<float>(i), 0.6f, 0.5f * static_cast<float>(i & 1))), 0.4f * static_cast<float>(i), Vec3(0, 1, 0)), Vec3(0.6f));
        sb.AddModelInstance(model, sb.AddSceneNode({ root, t, Mat4::Identity() }));
    }
    sb.AddLight(MakePointLight(Vec3(6.0f, 6.0f, 6.0f), Vec3(0.0f, 4.0f, -2.0f)), root);
    AddViewCamera(sb, Vec3(0.0f, 2.2f, -6.5f), Vec3(0.0f, 0.4f, 0.0f));
}

// Alpha-tested geometry and decals (anyhit.rahit / occlusionAnyhit.rahit): non-opaque geometries whose
// base-colour alpha decides, per candidate hit, whether the ray sees the surface.
//   leaves   alpha 1 inside, a soft 0.5..1 rim (visible to path rays, transparent to shadow rays), 0 outside
//   decal    alpha 0.3 rings / 0 elsewhere on a panel in front of the wall: never hit, tints what is behind
//   ghost    untextured material with colour alpha 0.4: an invisible tinting pane
//   film     untextured material with colour alpha 0.8: solid to path rays, casts no shadow
void CreateAlphaTestScene(SceneBuilder &sb, uint32_t seed)
{
    Rng rng(seed);
    const uint32_t checker = sb.AddTexture(MakeTexture(TextureType::Color, "Tiles", 32, 32, [](uint32_t x, uint32_t y, uint8_t *p) {
        const bool on = ((x / 4) + (y / 4)) & 1;
        p[0] = on ? 200 : 120; p[1] = on ? 200 : 120; p[2] = on ? 210 : 130; p[3] = 255;
    }));
    const uint32_t leafTex = sb.AddTexture(MakeTexture(TextureType::Color, "Leaf", 48, 48, [](uint32_t x, uint32_t y, uint8_t *p) {
        const float dx = (static_cast<float>(x) - 23.5f) / 23.5f, dy = (static_cast<float>(y) - 23.5f) / 16.0f;
        const float r = dx * dx + dy * dy;
        const bool vein = (x % 8 == 3) || std::abs(static_cast<int>(y) - 24) < 2;
        p[0] = vein ? 90 : 40; p[1] = vein ? 170 : 130; p[2] = vein ? 60 : 35;
        p[3] = r < 0.7f ? 255 : (r < 1.0f ? static_cast<uint8_t>(250.0f - 400.0f * (r - 0.7f)) : 0); // rim 250..130
    }));
    const uint32_t decalTex = sb.AddTexture(MakeTexture(TextureType::Color, "Rings", 64, 64, [](uint32_t x, uint32_t y, uint8_t *p) {
        const float dx = static_cast<float>(x) - 31.5f, dy = static_cast<float>(y) - 31.5f;
        const int ring = static_cast<int>(std::sqrt(dx * dx + dy * dy) / 5.0f);
        const bool on = ring < 6 && (ring & 1) == 0;
        p[0] = 220; p[1] = ring < 3 ? 30 : 120; p[2] = 40; p[3] = on ? 76 : 0;
    }));

    auto floorM = MakeMaterial(Vec3(1.0f), 0.8f, 0.0f);
    floorM.ColorIdx = checker;
    const auto floorMat = sb.AddMaterial("Tiles", floorM);
    const auto wallMat = sb.AddMaterial("Wall", MakeMaterial(Vec3(0.85f, 0.85f, 0.8f), 0.9f, 0.0f));
    auto leafM = MakeMaterial(Vec3(1.0f), 0.6f, 0.0f);
    leafM.ColorIdx = leafTex;
    const auto leafMat = sb.AddMaterial("Leaf", leafM);
    auto decalM = MakeMaterial(Vec3(1.0f), 0.5f, 0.0f);
    decalM.ColorIdx = decalTex;
    const auto decalMat = sb.AddMaterial("Decal", decalM);
    auto ghostM = MakeMaterial(Vec3(0.1f, 0.3f, 0.9f), 0.5f, 0.0f);
    ghostM.Color[3] = 0.4f;
    const auto ghostMat = sb.AddMaterial("Ghost", ghostM);
    auto filmM = MakeMaterial(Vec3(0.9f, 0.8f, 0.2f), 0.3f, 0.0f);
    filmM.Color[3] = 0.8f;
    const auto filmMat = sb.AddMaterial("Film", filmM);

    const uint32_t floor = AddGridSurface(sb, 4, 4, false, [](float u, float v) { return Vec3(-5.0f + 10.0f * u, 0.0f, -5.0f + 10.0f * v); });
    {
        auto &vertices = sb.GetVertices();
        for (size_t k = vertices.size() - 25; k < vertices.size(); k++)
        {
            vertices[k].TexCoords[0] *= 3.0f;
            vertices[k].TexCoords[1] *= 3.0f;
        }
    }
    const uint32_t wall = AddGridSurface(sb, 2, 2, false, [](float u, float v) { return Vec3(-5.0f + 10.0f * u, 5.0f * v, 3.0f); });
    const uint32_t decal = AddGridSurface(sb, 1, 1, false, [](float u, float v) { return Vec3(0.8f + 2.4f * u, 2.6f + 2.4f * v, 2.9f); }, false, false);
    const uint32_t ghost = AddGridSurface(sb, 1, 1, false, [](float u, float v) { return Vec3(0.5f + 2.0f * u, 0.2f + 2.5f * v, 1.5f); }, false, false);
    const uint32_t film = AddGridSurface(sb, 1, 1, false, [](float u, float v) { return Vec3(2.8f + 1.6f * u, 0.0f + 2.0f * v, 0.5f - 0.8f * u); }, false, false);
    const uint32_t leaves = AddCards(sb, rng, 300, Vec3(-3.5f, 0.3f, -1.5f), Vec3(0.5f, 3.2f, 1.5f), 0.35f, false);

    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const std::array<MeshInfo, 6> meshes = { MI(floor, floorMat), MI(wall, wallMat), MI(decal, decalMat), MI(ghost, ghostMat), MI(film, filmMat),
                                             MI(leaves, leafMat) };
    sb.AddModelInstance(sb.AddModel(meshes), root);

    sb.AddLight(MakePointLight(Vec3(9.0f, 8.5f, 8.0f), Vec3(-1.0f, 4.5f, -2.5f)), root);
    Shaders::DirectionalLight dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.Color[0] = 2.0f; dl.Color[1] = 1.9f; dl.Color[2] = 1.7f;
    dl.Direction[0] = 0.3f; dl.Direction[1] = -1.0f; dl.Direction[2] = 0.5f;
    sb.SetDirectionalLight(std::move(dl), root);
    AddViewCamera(sb, Vec3(0.0f, 2.0f, -6.0f), Vec3(0.0f, 1.6f, 0.0f));
}

// Every branch of sampleMaterial (material.glsl:62-171) in one scene: MetallicRoughness, SpecularGlossiness and Phong
// materials with and without their texture sets, a transmissive SpecularGlossiness "glass", a mesh whose material id
// carries an unknown type (the red default of material.glsl:161-165), and HasDxNormalTextures (the normal map's green
// channel is flipped, closestHit.rchit:101-102) -- what the ORCA scenes of BASELINE configs[2..4] bring through the
// importer (ExampleScenes.cpp:93,96-131: Phong / SpecularGlossiness materials, DirectX normal maps).
void CreateMaterialsTestScene(SceneBuilder &sb, float detail, uint32_t seed)
{
    const uint32_t tiles = sb.AddTexture(MakeTexture(TextureType::Color, "Mat Tiles", 64, 64, [](uint32_t x, uint32_t y, uint8_t *p) {
        const bool on = ((x / 8) + (y / 8)) & 1;
        p[0] = on ? 215 : 70; p[1] = on ? 205 : 75; p[2] = on ? 190 : 90; p[3] = 255;
    }));
    const uint32_t weave = sb.AddTexture(MakeTexture(TextureType::Color, "Mat Weave", 32, 48, [seed](uint32_t x, uint32_t y, uint8_t *p) {
        const uint32_t h = HashU(x / 2, y / 3, seed);
        p[0] = static_cast<uint8_t>(120 + (h & 63)); p[1] = static_cast<uint8_t>(60 + ((h >> 8) & 63)); p[2] = static_cast<uint8_t>(40 + ((h >> 16) & 31)); p[3] = 255;
    }));
    // a DirectX-convention normal map: green points DOWN the image, the scene flag flips it back
    const uint32_t dxBumps = sb.AddTexture(MakeTexture(TextureType::Normal, "Mat DX Bumps", 64, 64, [](uint32_t x, uint32_t y, uint8_t *p) {
        const float fx = std::cos(static_cast<float>(x) * 0.3927f), fy = std::cos(static_cast<float>(y) * 0.19635f);
        p[0] = static_cast<uint8_t>(128.0f + 70.0f * fx); p[1] = static_cast<uint8_t>(128.0f - 95.0f * fy); p[2] = 255; p[3] = 255;
    }));
    const uint32_t specMap = sb.AddTexture(MakeTexture(TextureType::Specular, "Mat Specular", 32, 32, [](uint32_t x, uint32_t y, uint8_t *p) {
        const bool stripe = ((x + y) / 6) & 1;
        p[0] = stripe ? 250 : 90; p[1] = stripe ? 210 : 90; p[2] = stripe ? 120 : 95; p[3] = 255;
    }));
    // glossiness / shininess live in the ALPHA channel (material.glsl:107,136)
    const uint32_t glossMap = sb.AddTexture(MakeTexture(TextureType::Glossiness, "Mat Gloss", 64, 16, [](uint32_t x, uint32_t, uint8_t *p) {
        p[0] = 255; p[1] = 0; p[2] = 255; p[3] = static_cast<uint8_t>(60 + (x * 190) / 63);
    }));
    const uint32_t shineMap = sb.AddTexture(MakeTexture(TextureType::Shininess, "Mat Shine", 16, 64, [](uint32_t, uint32_t y, uint8_t *p) {
        p[0] = 0; p[1] = 255; p[2] = 0; p[3] = static_cast<uint8_t>(250 - (y * 200) / 63);
    }));
    const uint32_t ember = sb.AddTexture(MakeTexture(TextureType::Emisive, "Mat Ember", 8, 8, [](uint32_t x, uint32_t y, uint8_t *p) {
        const bool on = ((x ^ y) & 3) == 0;
        p[0] = on ? 255 : 0; p[1] = on ? 90 : 0; p[2] = on ? 20 : 0; p[3] = 255;
    }));

    auto sgBase = [&]() {
        Shaders::SpecularGlossinessMaterial m;
        std::memset(&m, 0, sizeof(m));
        m.Color[0] = m.Color[1] = m.Color[2] = m.Color[3] = 1.0f;
        m.Specular[0] = m.Specular[1] = m.Specular[2] = 1.0f;
        m.Glossiness = 1.0f;
        m.AttenuationColor[0] = m.AttenuationColor[1] = m.AttenuationColor[2] = 1.0f;
        m.AttenuationDistance = 1e32f;
        m.Ior = 1.5f;
        m.EmissiveIdx = Scene::GetDefaultTextureIndex(TextureType::Emisive);
        m.ColorIdx = Scene::GetDefaultTextureIndex(TextureType::Color);
        m.NormalIdx = Scene::GetDefaultTextureIndex(TextureType::Normal);
        m.SpecularIdx = Scene::GetDefaultTextureIndex(TextureType::Specular);
        m.GlossinessIdx = Scene::GetDefaultTextureIndex(TextureType::Glossiness);
        return m;
    };
    auto phongBase = [&]() {
        Shaders::PhongMaterial m;
        std::memset(&m, 0, sizeof(m));
        m.Color[0] = m.Color[1] = m.Color[2] = m.Color[3] = 1.0f;
        m.Specular[0] = m.Specular[1] = m.Specular[2] = 1.0f;
        m.Shininess = 1.0f;
        m.AttenuationColor[0] = m.AttenuationColor[1] = m.AttenuationColor[2] = 1.0f;
        m.AttenuationDistance = 1e32f;
        m.Ior = 1.5f;
        m.EmissiveIdx = Scene::GetDefaultTextureIndex(TextureType::Emisive);
        m.ColorIdx = Scene::GetDefaultTextureIndex(TextureType::Color);
        m.NormalIdx = Scene::GetDefaultTextureIndex(TextureType::Normal);
        m.SpecularIdx = Scene::GetDefaultTextureIndex(TextureType::Specular);
        m.ShininessIdx = Scene::GetDefaultTextureIndex(TextureType::Shininess);
        return m;
    };

    std::vector<Shaders::MaterialId> mats;
    {
        auto m = MakeMaterial(Vec3(1.0f), 0.55f, 0.0f); // MetallicRoughness under the DX flag
        m.ColorIdx = tiles;
        m.NormalIdx = dxBumps;
        mats.push_back(sb.AddMaterial("Mat MR Bumped", m));
    }
    {
        auto m = sgBase(); // every SpecularGlossiness slot textured
        m.ColorIdx = weave;
        m.NormalIdx = dxBumps;
        m.SpecularIdx = specMap;
        m.GlossinessIdx = glossMap;
        m.Specular[0] = 0.9f; m.Specular[1] = 0.8f; m.Specular[2] = 0.7f;
        m.Glossiness = 0.85f;
        mats.push_back(sb.AddMaterial("Mat SG Textured", m));
    }
    {
        auto m = sgBase(); // factors only: default specular texel 1, default glossiness texel 0 -> roughness 1
        m.Color[0] = 0.25f; m.Color[1] = 0.5f; m.Color[2] = 0.8f;
        m.Specular[0] = m.Specular[1] = m.Specular[2] = 0.3f;
        m.EmissiveIdx = ember;
        m.EmissiveIntensity = 1.5f;
        mats.push_back(sb.AddMaterial("Mat SG Plain", m));
    }
    {
        auto m = sgBase(); // transmissive: Eta, attenuation and the refraction lobe through the SpecularGlossiness branch
        m.Color[0] = 0.95f; m.Color[1] = 0.97f; m.Color[2] = 1.0f;
        m.Specular[0] = m.Specular[1] = m.Specular[2] = 0.04f;
        m.GlossinessIdx = glossMap;
        m.Glossiness = 1.0f;
        m.Transmission = 0.9f;
        m.Ior = 1.45f;
        m.AttenuationColor[0] = 0.6f; m.AttenuationColor[1] = 0.9f; m.AttenuationColor[2] = 0.7f;
        m.AttenuationDistance = 0.8f;
        mats.push_back(sb.AddMaterial("Mat SG Glass", m));
    }
    {
        auto m = phongBase(); // every Phong slot textured
        m.ColorIdx = tiles;
        m.NormalIdx = dxBumps;
        m.SpecularIdx = specMap;
        m.ShininessIdx = shineMap;
        m.Color[0] = 0.9f; m.Color[1] = 0.6f; m.Color[2] = 0.5f;
        m.Shininess = 0.9f;
        mats.push_back(sb.AddMaterial("Mat Phong Textured", m));
    }
    {
        auto m = phongBase(); // factors only: default shininess texel 0 -> roughness 1
        m.Color[0] = 0.7f; m.Color[1] = 0.75f; m.Color[2] = 0.3f;
        m.Specular[0] = 0.5f; m.Specular[1] = 0.45f; m.Specular[2] = 0.2f;
        mats.push_back(sb.AddMaterial("Mat Phong Plain", m));
    }
    mats.push_back(Shaders::CreateMaterialId(0u, 7u)); // unknown material type: red, emissive red (material.glsl:161-165)

    auto floorM = MakeMaterial(Vec3(1.0f), 0.9f, 0.0f);
    floorM.ColorIdx = tiles;
    const auto floorMat = sb.AddMaterial("Mat Floor", floorM);
    const uint32_t floor = AddGridSurface(sb, 6, 6, false, [](float u, float v) { return Vec3(-7.0f + 14.0f * u, 0.0f, -4.0f + 9.0f * v); });
    {
        auto &vertices = sb.GetVertices();
        for (size_t k = vertices.size() - 49; k < vertices.size(); k++)
        {
            vertices[k].TexCoords[0] *= 5.0f;
            vertices[k].TexCoords[1] *= 3.0f;
        }
    }
    const uint32_t su = Scaled(48, detail, 8), svn = Scaled(24, detail, 4);
    const uint32_t sphere = AddGridSurface(sb, su, svn, true, [](float u, float v) {
        const float phi = 6.283185307179586f * u, theta = 3.14159265358979f * (0.02f + 0.96f * v);
        return Vec3(std::sin(theta) * std::cos(phi), -std::cos(theta), std::sin(theta) * std::sin(phi));
    });
    const std::array<uint32_t, 6> cube = AddCube(sb);

    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const std::array<MeshInfo, 1> floorMesh = { MI(floor, floorMat) };
    sb.AddModelInstance(sb.AddModel(floorMesh), root);
    for (size_t i = 0; i < mats.size(); i++)
    {
        const float x = -5.4f + 1.8f * static_cast<float>(i);
        const std::array<MeshInfo, 1> ball = { MI(sphere, mats[i]) };
        sb.AddModelInstance(sb.AddModel(ball), sb.AddSceneNode({ root, Scale(Translate(Mat4::Identity(), Vec3(x, 0.75f, 0.0f)), Vec3(0.72f)), Mat4::Identity() }));
        std::array<MeshInfo, 6> box;
        for (int k = 0; k < 6; k++)
            box[k] = MI(cube[k], mats[(i + 3) % mats.size()]);
        const Mat4 t = Scale(Rotate(Translate(Mat4::Identity(), Vec3(x + 0.3f, 0.4f, 2.4f)), 0.5f + 0.3f * static_cast<float>(i), Vec3(0, 1, 0)), Vec3(0.4f));
        sb.AddModelInstance(sb.AddModel(box), sb.AddSceneNode({ root, t, Mat4::Identity() }));
    }
    sb.SetDxNormalTextures();
    sb.AddLight(MakePointLight(Vec3(9.0f, 8.5f, 8.0f), Vec3(-2.0f, 4.0f, -2.5f)), root);
    sb.AddLight(MakePointLight(Vec3(3.0f, 4.0f, 6.0f), Vec3(4.0f, 2.5f, 3.5f)), root);
    Shaders::DirectionalLight dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.Color[0] = 1.6f; dl.Color[1] = 1.5f; dl.Color[2] = 1.4f;
    dl.Direction[0] = 0.35f; dl.Direction[1] = -1.0f; dl.Direction[2] = 0.45f;
    sb.SetDirectionalLight(std::move(dl), root);
    AddViewCamera(sb, Vec3(0.0f, 3.4f, -8.2f), Vec3(0.0f, 0.5f, 0.6f));
}

// Node animation, hierarchy and skinning in one small scene (row N3):
//   * a platform spinning about y with a cube riding on it (child node of an animated node) that also bobs,
//   * a skinned "tentacle": a 12-sided tube whose rings blend between consecutive bones of a 4-bone chain
//     (AnimatedVertex / Bone / skinning.comp), each joint swinging with a phase offset,
//   * a point light sliding along x (the animated light of the reference's textured-cubes scene,
//     ExampleScenes.cpp:636-651).
// TickPerSecond 30, duration 120 ticks (4 s loop).
void CreateAnimatedTestScene(SceneBuilder &sb, float detail, uint32_t seed)
{
    (void)seed;
    const auto floorMat = sb.AddMaterial("Anim Floor", MakeMaterial(Vec3(0.6f, 0.6f, 0.62f), 0.8f, 0.0f));
    const auto platformMat = sb.AddMaterial("Platform", MakeMaterial(Vec3(0.75f, 0.3f, 0.2f), 0.5f, 0.0f));
    const auto riderMat = sb.AddMaterial("Rider", MakeMaterial(Vec3(0.9f, 0.8f, 0.3f), 0.3f, 1.0f));
    const auto skinMat = sb.AddMaterial("Tentacle", MakeMaterial(Vec3(0.25f, 0.55f, 0.35f), 0.45f, 0.0f));

    const uint32_t floor = AddBox(sb, Vec3(0, -0.1f, 0), Vec3(8, 0.1f, 8));
    const uint32_t platform = AddBox(sb, Vec3(0, 0.15f, 0), Vec3(1.4f, 0.15f, 1.4f));
    const uint32_t rider = AddBox(sb, Vec3(0, 0, 0), Vec3(0.3f, 0.3f, 0.3f));

    const uint32_t root = sb.AddSceneNode({ 0u, Mat4::Identity(), Mat4::Identity() });
    const std::array<MeshInfo, 1> floorMesh = { MI(floor, floorMat) };
    sb.AddModelInstance(sb.AddModel(floorMesh), root);

    const uint32_t platformNode = sb.AddSceneNode({ root, Translate(Mat4::Identity(), Vec3(-2.0f, 0.0f, 0.5f)), Mat4::Identity() });
    const std::array<MeshInfo, 1> platformMesh = { MI(platform, platformMat) };
    sb.AddModelInstance(sb.AddModel(platformMesh), platformNode);
    const uint32_t riderNode = sb.AddSceneNode({ platformNode, Translate(Mat4::Identity(), Vec3(0.9f, 0.6f, 0.0f)), Mat4::Identity() });
    const std::array<MeshInfo, 1> riderMesh = { MI(rider, riderMat) };
    sb.AddModelInstance(sb.AddModel(riderMesh), riderNode);

    const float duration = 120.0f;
    std::vector<AnimationNode> nodes;
    {
        AnimationNode spin;
        spin.SceneNodeIndex = platformNode;
        spin.Positions.Keys = { { Vec3(-2.0f, 0.0f, 0.5f), 0.0f } };
        for (int k = 0; k <= 4; k++)
            spin.Rotations.Keys.push_back({ AngleAxis(1.5707963f * static_cast<float>(k), Vec3(0, 1, 0)), 30.0f * static_cast<float>(k) });
        spin.Scales.Keys = { { Vec3(1.0f), 0.0f } };
        nodes.push_back(std::move(spin));
        AnimationNode bob;
        bob.SceneNodeIndex = riderNode;
        bob.Positions.Keys = { { Vec3(0.9f, 0.6f, 0.0f), 0.0f }, { Vec3(0.9f, 1.3f, 0.0f), 60.0f }, { Vec3(0.9f, 0.6f, 0.0f), 120.0f } };
        bob.Rotations.Keys = { { Quat(), 0.0f }, { AngleAxis(3.0f, Vec3(1, 0, 0)), 120.0f } };
        bob.Scales.Keys = { { Vec3(1.0f), 0.0f }, { Vec3(1.4f, 0.7f, 1.4f), 60.0f }, { Vec3(1.0f), 120.0f } };
        nodes.push_back(std::move(bob));
    }

    // skinned tube along +y: 4 bones of length 0.8, ring weights blend linearly between neighbouring bones
    const uint32_t sides = Scaled(48, detail, 12), ringsPerBone = Scaled(24, detail, 4), boneCountLocal = 4;
    const float boneLength = 0.8f, radius = 0.22f, twoPi = 6.283185307179586f;
    const Vec3 base(2.0f, 0.0f, 0.0f);
    uint32_t boneNodes[4];
    uint32_t parent = root;
    for (uint32_t b = 0; b < boneCountLocal; b++)
    {
        const Vec3 local = b == 0 ? base : Vec3(0.0f, boneLength, 0.0f);
        boneNodes[b] = sb.AddSceneNode({ parent, Translate(Mat4::Identity(), local), Mat4::Identity() });
        parent = boneNodes[b];
        // inverse bind matrix: the joint sits at base + (0, b * boneLength, 0) in bind pose
        const Mat4 bind = Translate(Mat4::Identity(), base + Vec3(0.0f, boneLength * static_cast<float>(b), 0.0f));
        sb.AddBone({ boneNodes[b], Inverse(bind) });
        AnimationNode swing;
        swing.SceneNodeIndex = boneNodes[b];
        swing.Positions.Keys = { { local, 0.0f } };
        const float amp = 0.35f;
        for (int k = 0; k <= 8; k++)
        {
            const float phase = twoPi * static_cast<float>(k) / 8.0f - 0.9f * static_cast<float>(b);
            swing.Rotations.Keys.push_back({ AngleAxis(amp * std::sin(phase), Vec3(0, 0, 1)), duration * static_cast<float>(k) / 8.0f });
        }
        swing.Scales.Keys = { { Vec3(1.0f), 0.0f } };
        nodes.push_back(std::move(swing));
    }
    {
        auto &av = sb.GetAnimatedVertices();
        auto &ai = sb.GetAnimatedIndices();
        const uint32_t vertexOffset = static_cast<uint32_t>(av.size()), indexOffset = static_cast<uint32_t>(ai.size());
        const uint32_t rings = ringsPerBone * boneCountLocal;
        for (uint32_t j = 0; j <= rings; j++)
        {
            const float t = static_cast<float>(j) / static_cast<float>(ringsPerBone); // position along the chain in bones
            uint32_t b0 = static_cast<uint32_t>(t);
            if (b0 > boneCountLocal - 1)
                b0 = boneCountLocal - 1;
            const uint32_t b1 = b0 + 1 < boneCountLocal ? b0 + 1 : b0;
            const float frac = t - static_cast<float>(b0);
            const float w1 = b1 == b0 ? 0.0f : frac * frac * (3.0f - 2.0f * frac) * 0.5f; // the next joint takes over gradually
            const float taper = 1.0f - 0.55f * static_cast<float>(j) / static_cast<float>(rings);
            for (uint32_t i = 0; i < sides; i++)
            {
                const float a = twoPi * static_cast<float>(i) / static_cast<float>(sides);
                const Vec3 n(std::cos(a), 0.0f, std::sin(a));
                const Vec3 p = base + Vec3(0.0f, boneLength * t, 0.0f) + n * (radius * taper);
                Shaders::AnimatedVertex v;
                std::memset(&v, 0, sizeof(v));
                v.Position[0] = p.x; v.Position[1] = p.y; v.Position[2] = p.z;
                v.TexCoords[0] = static_cast<float>(i) / static_cast<float>(sides); v.TexCoords[1] = t;
                v.Normal[0] = n.x; v.Normal[1] = n.y; v.Normal[2] = n.z;
                v.Tangent[0] = -n.z; v.Tangent[1] = 0.0f; v.Tangent[2] = n.x;
                v.Bitangent[0] = 0.0f; v.Bitangent[1] = 1.0f; v.Bitangent[2] = 0.0f;
                v.BoneIndices[0] = b0; v.BoneIndices[1] = b1;
                v.BoneWeights[0] = 1.0f - w1; v.BoneWeights[1] = w1;
                av.push_back(v);
            }
        }
        for (uint32_t j = 0; j < rings; j++)
            for (uint32_t i = 0; i < sides; i++)
            {
                const uint32_t a = j * sides + i, b = j * sides + (i + 1) % sides, c = (j + 1) * sides + (i + 1) % sides, d = (j + 1) * sides + i;
                for (uint32_t k : { a, d, c, a, c, b }) // outward-facing
                    ai.push_back(k);
            }
        const uint32_t tube = sb.AddGeometry({ vertexOffset, (rings + 1) * sides, indexOffset, rings * sides * 6, true, true, { 0, 0 } });
        const std::array<MeshInfo, 1> tubeMesh = { MI(tube, skinMat) };
        sb.AddModelInstance(sb.AddModel(tubeMesh), root); // skinned meshes live in world space: the bones move them
    }

    const uint32_t lightNode = sb.AddSceneNode({ root, Translate(Mat4::Identity(), Vec3(-1.0f, 3.0f, -1.0f)), Mat4::Identity() });
    sb.AddLight(MakePointLight(Vec3(14.0f, 13.0f, 12.0f), Vec3(0.0f)), lightNode);
    {
        AnimationNode slide;
        slide.SceneNodeIndex = lightNode;
        slide.Positions.Keys = { { Vec3(-1.0f, 3.0f, -1.0f), 0.0f }, { Vec3(2.5f, 3.0f, -1.0f), 60.0f }, { Vec3(-1.0f, 3.0f, -1.0f), 120.0f } };
        slide.Rotations.Keys = { { Quat(), 0.0f } };
        slide.Scales.Keys = { { Vec3(1.0f), 0.0f } };
        nodes.push_back(std::move(slide));
    }
    sb.AddAnimation(Animation { std::move(nodes), 30.0f, duration });

    Shaders::DirectionalLight dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.Color[0] = dl.Color[1] = dl.Color[2] = 1.5f;
    dl.Direction[0] = -0.3f; dl.Direction[1] = -1.0f; dl.Direction[2] = 0.4f;
    sb.SetDirectionalLight(std::move(dl), root);
    AddViewCamera(sb, Vec3(0.3f, 3.0f, -6.5f), Vec3(0.0f, 1.2f, 0.0f));
}

// ---------------------------------------------------------------------------

const char *const kSceneNames = "default,roughness_cubes,attenuation_blob,chess_like,temple_like,atrium_like,street_like,texture_test,alpha_test,reuse_mesh_cubes,animated_test,materials_test";

const char *GetSceneNames()
{
    return kSceneNames;
}

std::shared_ptr<Scene> CreateScene(const std::string &name, float detail, uint32_t seed)
{
    SceneBuilder sb;
    bool useSceneCamera = true;
    if (name.rfind("file:", 0) == 0) // a glTF 2.0 asset through the importer (the registry's file scenes, ExampleScenes.cpp:41-66)
        SceneImporter::AddFile(sb, name.substr(5));
    else if (name.rfind("description:", 0) == 0)
    {
        // a SceneDescription (ExampleScenes.cpp:87-236): inline JSON, or "description:@file.json" with paths relative to it
        std::string text = name.substr(12);
        std::filesystem::path base;
        if (!text.empty() && text[0] == '@')
        {
            const std::filesystem::path file(text.substr(1));
            std::ifstream in(file);
            if (!in)
                throw error("Scene description not found: " + file.string());
            std::stringstream ss;
            ss << in.rdbuf();
            text = ss.str();
            base = file.parent_path();
        }
        SceneDescription::Parse(text, base).Build(sb);
    }
    else if (name == "default")
    {
        CreateDefaultScene(sb);
        useSceneCamera = false;
    }
    else if (name == "roughness_cubes")
    {
        CreateRoughnessTestCubesScene(sb);
        AddViewCamera(sb, Vec3(6.0f, 9.0f, 6.0f), Vec3(-10.0f, 0.0f, -10.0f));
    }
    else if (name == "attenuation_blob")
        CreateAttenuationBlobScene(sb, detail, seed ? seed : 1);
    else if (name == "chess_like")
        CreateChessLikeScene(sb, detail, seed ? seed : 2);
    else if (name == "temple_like")
        CreateTempleLikeScene(sb, detail, seed ? seed : 3);
    else if (name == "atrium_like")
        CreateAtriumLikeScene(sb, detail, seed ? seed : 4);
    else if (name == "street_like")
        CreateStreetLikeScene(sb, detail, seed ? seed : 5);
    else if (name == "texture_test")
        CreateTextureTestScene(sb, seed ? seed : 6);
    else if (name == "reuse_mesh_cubes")
        CreateReuseMeshCubesScene(sb, seed ? seed : 8);
    else if (name == "animated_test")
        CreateAnimatedTestScene(sb, detail, seed ? seed : 9);
    else if (name == "alpha_test")
        CreateAlphaTestScene(sb, seed ? seed : 7);
    else if (name == "materials_test")
        CreateMaterialsTestScene(sb, detail, seed ? seed : 10);
    else
        throw error("Unknown scene: " + name);
    auto scene = sb.CreateSceneShared(name);
    if (useSceneCamera && scene->GetSceneCamerasCount() > 0)
    {
        scene->SetActiveCamera(0);
        scene->Update(0.0f);
    }
    return scene;
}

}
