// ObjReader.h -- Wavefront OBJ (+ MTL) for the host importer (row N2), the third format the reference reaches through
// assimp (SceneImporter.cpp:1048-1114).  Like FbxReader, it produces the in-memory document the glTF pipeline works on.
//
// Read: v / vt / vn, f with any of the v, v/vt, v//vn, v/vt/vn forms, negative (relative) indices, polygons of any size
// (-> fans), o / g (one mesh per object or group), usemtl / mtllib; from the MTL: Kd, Ks, Ke, Ns, Ni, d / Tr and the maps
// map_Kd, map_Ks, map_Ns, map_Ke, norm / map_Kn (assimp files map_bump / bump under HEIGHT, which no slot of the
// reference reads).  Corners with equal (v, vt, vn) share a vertex.
//
// What assimp's OBJ importer puts into the aiMaterial rides in "extras.assimp" as for FBX.  It ALWAYS sets a shininess
// (ObjFileImporter: AI_MATKEY_SHININESS from Ns, 0 when the MTL has none), so ChooseMaterialType (SceneImporter.cpp:300-319)
// says Phong for every OBJ material, and the reference's LoadMaterials throws "Unsupported material type" for Phong
// (the missing `break`, :390-393): an OBJ file loads only under a TextureMapping that forces another model.
#pragma once

#include <cstdint>
#include <filesystem>
#include <span>
#include <vector>

#include "Json.h"

namespace PathTracing
{

// `file` is the .obj text; the material libraries it names are read from `directory`.  Throws PathTracing::error.
void ConvertObjToGltf(std::span<const uint8_t> file, const std::filesystem::path &directory, Json &json, std::vector<uint8_t> &buffer);

}
