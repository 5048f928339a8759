// FbxReader.h -- binary FBX (Kaydara FBX Binary, versions 7.1 - 7.7) for the host importer (row N2).
//
// The reference hands .fbx files (UE4 Sun Temple, Amazon Bistro: ExampleScenes.cpp:118-154) to assimp and walks the
// resulting aiScene (SceneImporter.cpp:1048-1114).  assimp is not available, so the file is read here and turned into the
// SAME in-memory document the glTF importer works on -- nodes with matrices, one primitive per (geometry, material),
// materials, images, KHR_lights_punctual lights -- so that one pipeline (LoadSceneNodes / LoadMaterials / LoadMeshes /
// LoadModels / LoadLights of SceneImporter.cpp) serves both formats.  What assimp's FBX converter would put into the
// aiMaterial rides in the material's "extras.assimp" object:
//   "textures"        { "DIFFUSE" | "NORMALS" | "SPECULAR" | "SHININESS" | "EMISSIVE": texture index }  (the aiTextureType
//                     of the FBX connection: DiffuseColor, NormalMap / Bump, SpecularColor, ShininessExponent, EmissiveColor)
//   "diffuse"         AI_MATKEY_COLOR_DIFFUSE (DiffuseColor x DiffuseFactor)
//   "shininess"       AI_MATKEY_SHININESS, present when the material has Shininess / ShininessExponent
//   "specularFactor"  AI_MATKEY_SPECULAR_FACTOR, present when the material has SpecularFactor
// and "emissiveFactor" (EmissiveColor x EmissiveFactor) in the usual glTF place.
//
// Read: the node records (32- and 64-bit offsets), every property type incl. zlib-deflated arrays; Objects: Geometry
// (control points, polygons of any size -> triangle fans, normals and UVs ByPolygonVertex / ByVertice, Direct /
// IndexToDirect, materials AllSame / ByPolygon), Model (Lcl translation / rotation / scaling, pre- and post-rotation,
// pivots and offsets, geometric transform; Euler order XYZ), Material, Texture, NodeAttribute lights and cameras, animation
// stacks (curves on Lcl Translation / Rotation / Scaling of models with a plain T * R * S transform, sampled linearly),
// Skin / Cluster deformers; Connections.  Not read: ASCII FBX, curves on models with pivots, embedded media.
#pragma once

#include <cstdint>
#include <span>
#include <vector>

#include "Json.h"

namespace PathTracing
{

bool IsBinaryFbx(std::span<const uint8_t> file);

// The file as a glTF 2.0 document: `json` + the one binary buffer its accessors refer to.  Image URIs are the files'
// relative names with forward slashes.  Throws PathTracing::error on a malformed file.
void ConvertFbxToGltf(std::span<const uint8_t> file, Json &json, std::vector<uint8_t> &buffer);

}
