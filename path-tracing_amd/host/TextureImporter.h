// TextureImporter.h -- host mirror of Path-Tracing/TextureImporter.{h,cpp} (rows N1 / N2): turns an image file
// (or an in-memory blob, e.g. a glTF buffer view) into a TextureInfo with decoded level-0 texels.
//
// The reference calls stb_image (PNG / JPG / TGA / HDR ...) and gli (DDS BC1 / BC3 / BC5); neither library is
// available here, so the decoders are written out:
//   PNG   all colour types, 1 - 16 bit (16 -> high byte, as stb), zlib inflate with stored / fixed / dynamic blocks,
//         Adam7 interlacing
//   JPG   baseline / extended sequential and progressive DCT (Huffman, 8-bit, 1 or 3 components, any sampling factors,
//         restart intervals); arithmetic-coded and lossless files are rejected
//   TGA   uncompressed and RLE true colour / greyscale, 8 / 24 / 32 bit
//   HDR   Radiance RGBE, run-length and flat scanlines -> RGBA32F
//   DDS   BC1 / BC3 / BC5 blocks decoded to RGBA8, with the file's own mip levels (a complete chain is uploaded as it
//         is, like the reference's; an incomplete one is regenerated from level 0: TextureUploader.cpp:440-456)
// A file that cannot be decoded throws PathTracing::error, which the importer turns into the default texture
// of the slot (SceneImporter.cpp:97-101).
//
// Reference semantics kept: *hasTransparency = the file has 4 channels (TextureImporter.cpp:300-301); colour
// textures with an alpha channel get their alpha-0 texels zeroed (PremultiplyTextureData, :24-51).
#pragma once

#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <span>
#include <string>
#include <vector>

#include "Scene.h"

namespace PathTracing
{

struct DecodedImage
{
    uint32_t Width = 0, Height = 0;
    uint32_t Channels = 0;      // channels in the file (1..4)
    bool IsFloat = false;       // RGBA32F (HDR) instead of RGBA8
    uint32_t Levels = 1;        // mip levels in Pixels (DDS files carry their own chain), level 0 first
    std::vector<uint8_t> Pixels; // RGBA8: 4 bytes per texel, RGBA32F: 16 bytes per texel
};

class TextureImporter
{
public:
    // TextureImporter::GetTextureInfo + LoadTextureData in one step
    static TextureInfo GetTextureInfo(const std::filesystem::path &path, TextureType type, std::string &&name, bool *hasTransparency = nullptr);
    static TextureInfo GetTextureInfo(std::span<const uint8_t> memory, TextureType type, std::string &&name, bool *hasTransparency = nullptr);

    // format sniffing + decode; throws PathTracing::error
    static DecodedImage Decode(std::span<const uint8_t> file);

    static DecodedImage DecodePng(std::span<const uint8_t> file);
    static DecodedImage DecodeJpeg(std::span<const uint8_t> file);
    static DecodedImage DecodeTga(std::span<const uint8_t> file);
    static DecodedImage DecodeHdr(std::span<const uint8_t> file);
    static DecodedImage DecodeDds(std::span<const uint8_t> file);

    // zlib stream -> bytes (RFC 1950 / 1951); also used by tests
    static std::vector<uint8_t> Inflate(std::span<const uint8_t> zlibStream);
};

std::vector<uint8_t> ReadFileBytes(const std::filesystem::path &path);

}
