// OutputSaver.h -- host mirror of Path-Tracing/Renderer/OutputSaver.h for the HIP backend (row N4).
//
// The reference's OutputSaver owns Vulkan images, a read-back buffer and a writer thread
// (RegisterOutput / StartOutputWait / EndOutput / CancelOutput, OutputSaver.cpp:64-225) and hands the
// bytes to stb_image_write or pipes them to an ffmpeg child (WriteImage, :227-257).  Here the GPU side is
// ptx_postprocess + ptx_read_output; this class keeps the same call sequence and the same formats, with
// the encoders written out (no stb): PNG (zlib stream, fixed-Huffman deflate + LZ77), baseline JPEG (quality 90,
// 4:2:0, per-image optimal Huffman tables), TGA, Radiance HDR, and the raw-RGBA pipe to `ffmpeg` for MP4.
//
// Also: checkpoint / resume of the running sum -- a raw dump with a 32-byte header.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <span>
#include <string>
#include <vector>

namespace PathTracing
{

enum class OutputFormat
{
    Png, Jpg, Tga, Hdr, Mp4
};

struct Extent2D
{
    uint32_t width, height;
};

struct OutputInfo
{
    std::filesystem::path Path;
    Extent2D Extent;
    uint32_t Framerate;
    OutputFormat Format;
};

/*
 * As a caller do (same protocol as the reference):
 * 1. RegisterOutput to open the sink (and the ffmpeg pipe for Mp4)
 * 2. render + ptx_postprocess
 * 3. SubmitFrame with the bytes of ptx_read_output every frame (StartOutputWait's job)
 * 4. EndOutput after the last frame
 */
class OutputSaver
{
public:
    OutputSaver();
    ~OutputSaver();

    [[nodiscard]] bool CanOutputVideo() const { return m_HasFFmpeg; }
    // PTX_OUTPUT_RGBA32F for Hdr, PTX_OUTPUT_RGBA8_SRGB otherwise (OutputSaver::SelectImageFormat, :259-274)
    [[nodiscard]] static uint32_t SelectImageFormat(OutputFormat format);

    void RegisterOutput(const OutputInfo &info);
    bool SubmitFrame(std::span<const std::byte> data);
    void EndOutput();
    void CancelOutput();

    // one image to one file (Png / Jpg / Tga: RGBA8, Hdr: RGBA32F), top row first
    static bool WriteImage(const OutputInfo &info, std::span<const std::byte> data, FILE *videoPipe = nullptr);

    // encoders, exposed for tests
    static std::vector<uint8_t> EncodePng(uint32_t width, uint32_t height, const uint8_t *rgba);
    static std::vector<uint8_t> EncodeTga(uint32_t width, uint32_t height, const uint8_t *rgba);
    static std::vector<uint8_t> EncodeJpg(uint32_t width, uint32_t height, const uint8_t *rgba, int quality = 90); // stb's default quality
    static std::vector<uint8_t> EncodeHdr(uint32_t width, uint32_t height, const float *rgba);

private:
    OutputInfo m_Info {};
    bool m_Registered = false;
    bool m_HasFFmpeg = false;
    FILE *m_FFmpegPipe = nullptr;
};

// running-sum checkpoint: header { "PTXACC1\0", width, height, totalSamples, reserved[3] } + W*H*4 floats
bool SaveCheckpoint(const std::filesystem::path &path, uint32_t width, uint32_t height, uint32_t totalSamples, const float *rgba);
bool LoadCheckpoint(const std::filesystem::path &path, uint32_t &width, uint32_t &height, uint32_t &totalSamples, std::vector<float> &rgba);

}
