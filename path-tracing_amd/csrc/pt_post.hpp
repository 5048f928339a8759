// pt_post.hpp -- the output stage (row N4) on the device: what Renderer::RecordPostProcessCommands
// (Renderer.cpp:928-1085) and RecordSaveOutputCommands (:1204-1246) dispatch after the path-tracing pass,
// and OutputSaver's conversion into its sRGB8 / RGBA32F image (OutputSaver.cpp:64-86).
//
//   k_postprocess      postprocess.comp:16-40      sum / TotalSamples * Exposure, NaN / Inf markers, bloom prefilter
//   k_bloom_downsample bloomDownsample.comp:18-62  13-tap filter, level i -> i + 1
//   k_bloom_upsample   bloomUpsample.comp:18-55    3x3 tent of level i added onto level i - 1
//   k_compose_tonemap  composition.comp:16-26 + toneMapping.comp:13-25 (two stores of the reference fused: the
//                      value is rounded to binary16 between them exactly as the rgba16f image would)
//   k_encode_srgb8     the blit into VK_FORMAT_R8G8B8A8_SRGB
//
// The reference's intermediate images are rgba16f; here they are float arrays whose values are rounded to
// binary16 at every store (f16Round), which keeps the arithmetic identical and the code free of half types.
// All of it is HBM-streaming work: one launch per pass, coalesced float4 / float3 rows.
#pragma once

#include "pt_device.hpp"

namespace ptd
{

// float -> IEEE binary16 (round to nearest even, overflow to infinity, subnormals kept) -> float
PT_DEV float f16Round(float f)
{
    const uint32_t x = __float_as_uint(f), sign = x & 0x80000000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u)
        return ax > 0x7f800000u ? __uint_as_float(sign | 0x7fc00000u) : f;
    if (ax >= 0x477ff000u) // >= 65520
        return __uint_as_float(sign | 0x7f800000u);
    if (ax < 0x38800000u) // below 2^-14: half subnormal, quantum 2^-24
    {
        const float q = __uint_as_float(ax) * 16777216.0f;
        const float rq = (q + 12582912.0f) - 12582912.0f;
        return __uint_as_float(sign | __float_as_uint(rq * (1.0f / 16777216.0f)));
    }
    const uint32_t lsb = (ax >> 13) & 1u;
    return __uint_as_float(sign | ((ax + 0x0fffu + lsb) & 0xffffe000u));
}
PT_DEV f3 f16Round(f3 c) { return F3(f16Round(c.x), f16Round(c.y), f16Round(c.z)); }

PT_DEV float exp_(float x) // exp through the fixed exp2 kernel
{
    if (x != x)
        return x;
    double t = (double)x * 1.4426950408889634;
    if (t > 300.0)
        t = 300.0;
    if (t < -300.0)
        t = -300.0;
    return (float)exp2_(t);
}

PT_DEV void postprocessPixel(f3 accColor, const PtxPostProcessingUniformData &u, f3 &colorOut, f3 &bloomOut) // postprocess.comp:22-36
{
    f3 color = (accColor / (float)u.TotalSamples) * u.Exposure;
    if (__builtin_isnan(color.x) || __builtin_isnan(color.y) || __builtin_isnan(color.z))
        color = F3(5000.0f, 0.0f, 0.0f);
    if (__builtin_isinf(color.x) || __builtin_isinf(color.y) || __builtin_isinf(color.z))
        color = F3(0.0f, 5000.0f, 0.0f);
    const float knee = 0.5f;
    const float threshold = u.BloomThreshold;
    const float br = fmax_(color.x, fmax_(color.y, color.z));
    const f3 curve = F3(threshold - knee, knee * 2.0f, div_(0.25f, knee));
    float rq = clamp_(br - curve.x, 0.0f, curve.y);
    rq = curve.z * rq * rq;
    bloomOut = color * (div_(fmax_(rq, br - threshold), fmax_(br, 0.0001f)));
    colorOut = color;
}

PT_DEV f3 compositionPixel(f3 postProcessColor, f3 bloomColor, const PtxPostProcessingUniformData &u) // composition.comp:22
{
    return bloomColor * (u.BloomIntensity * 0.1f) + postProcessColor * 1.0f;
}

PT_DEV f3 toneMapPixel(f3 color, uint32_t mode) // toneMapping.comp:19-21
{
    if (mode == PTX_TONE_MAPPING_HDR)
        return color;
    return F3(1.0f - exp_(-color.x), 1.0f - exp_(-color.y), 1.0f - exp_(-color.z));
}

struct BloomLevel
{
    float *rgb; // 3 floats per texel, binary16-valued
    uint32_t w, h;
};

// texture(u_BloomSampler[level], uv): bilinear, clamp to edge (Renderer.cpp:114-119)
PT_DEV f3 bloomTap(const BloomLevel &L, float u, float v)
{
    const float x = u * (float)L.w - 0.5f, y = v * (float)L.h - 0.5f;
    const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
    const float ax = x - x0, ay = y - y0;
    const float mx = (float)(L.w - 1), my = (float)(L.h - 1);
    const uint32_t ix0 = (uint32_t)clamp_(x0, 0.0f, mx), ix1 = (uint32_t)clamp_(x0 + 1.0f, 0.0f, mx);
    const uint32_t iy0 = (uint32_t)clamp_(y0, 0.0f, my), iy1 = (uint32_t)clamp_(y0 + 1.0f, 0.0f, my);
    const float *p00 = &L.rgb[((size_t)iy0 * L.w + ix0) * 3], *p10 = &L.rgb[((size_t)iy0 * L.w + ix1) * 3];
    const float *p01 = &L.rgb[((size_t)iy1 * L.w + ix0) * 3], *p11 = &L.rgb[((size_t)iy1 * L.w + ix1) * 3];
    f3 r;
    float top, bot;
    top = p00[0] * (1.0f - ax) + p10[0] * ax; bot = p01[0] * (1.0f - ax) + p11[0] * ax; r.x = top * (1.0f - ay) + bot * ay;
    top = p00[1] * (1.0f - ax) + p10[1] * ax; bot = p01[1] * (1.0f - ax) + p11[1] * ax; r.y = top * (1.0f - ay) + bot * ay;
    top = p00[2] * (1.0f - ax) + p10[2] * ax; bot = p01[2] * (1.0f - ax) + p11[2] * ax; r.z = top * (1.0f - ay) + bot * ay;
    return r;
}

PT_DEV f3 add4(f3 a, f3 b, f3 c, f3 d) { return ((a + b) + c) + d; }

__global__ void k_postprocess(const float4 *__restrict__ accum, uint32_t n, PtxPostProcessingUniformData u, float *__restrict__ post,
                              float *__restrict__ bloom0)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const float4 a = accum[i];
        f3 color, bloom;
        postprocessPixel(F3(a.x, a.y, a.z), u, color, bloom);
        color = f16Round(color);
        bloom = f16Round(bloom);
        post[3 * (size_t)i] = color.x; post[3 * (size_t)i + 1] = color.y; post[3 * (size_t)i + 2] = color.z;
        bloom0[3 * (size_t)i] = bloom.x; bloom0[3 * (size_t)i + 1] = bloom.y; bloom0[3 * (size_t)i + 2] = bloom.z;
    }
}

__global__ void k_bloom_downsample(BloomLevel src, BloomLevel dst)
{
    const uint32_t n = dst.w * dst.h;
    const float tx = div_(1.0f, (float)src.w), ty = div_(1.0f, (float)src.h);
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x)
    {
        const uint32_t xx = p % dst.w, yy = p / dst.w;
        const float u = div_((float)xx + 0.5f, (float)dst.w), v = div_((float)yy + 0.5f, (float)dst.h);
        const f3 a = bloomTap(src, u + -2.0f * tx, v + 2.0f * ty), b = bloomTap(src, u + 0.0f * tx, v + 2.0f * ty),
                 c = bloomTap(src, u + 2.0f * tx, v + 2.0f * ty);
        const f3 d = bloomTap(src, u + -2.0f * tx, v + 0.0f * ty), e = bloomTap(src, u + 0.0f * tx, v + 0.0f * ty),
                 f = bloomTap(src, u + 2.0f * tx, v + 0.0f * ty);
        const f3 g = bloomTap(src, u + -2.0f * tx, v + -2.0f * ty), h = bloomTap(src, u + 0.0f * tx, v + -2.0f * ty),
                 i = bloomTap(src, u + 2.0f * tx, v + -2.0f * ty);
        const f3 j = bloomTap(src, u + -1.0f * tx, v + 1.0f * ty), k = bloomTap(src, u + 1.0f * tx, v + 1.0f * ty);
        const f3 l = bloomTap(src, u + -1.0f * tx, v + -1.0f * ty), m = bloomTap(src, u + 1.0f * tx, v + -1.0f * ty);
        f3 down = e * 0.125f;
        down = down + add4(a, c, g, i) * 0.03125f;
        down = down + add4(b, d, f, h) * 0.0625f;
        down = down + add4(j, k, l, m) * 0.125f;
        down = f16Round(down);
        float *o = &dst.rgb[(size_t)p * 3];
        o[0] = down.x; o[1] = down.y; o[2] = down.z;
    }
}

__global__ void k_bloom_upsample(BloomLevel src, BloomLevel dst)
{
    const uint32_t n = dst.w * dst.h;
    const float x = div_(1.0f, (float)src.w), y = div_(1.0f, (float)src.h);
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x)
    {
        const uint32_t xx = p % dst.w, yy = p / dst.w;
        const float u = div_((float)xx + 0.5f, (float)dst.w), v = div_((float)yy + 0.5f, (float)dst.h);
        const f3 a = bloomTap(src, u + -x, v + y), b = bloomTap(src, u + 0.0f, v + y), c = bloomTap(src, u + x, v + y);
        const f3 d = bloomTap(src, u + -x, v + 0.0f), e = bloomTap(src, u + 0.0f, v + 0.0f), f = bloomTap(src, u + x, v + 0.0f);
        const f3 g = bloomTap(src, u + -x, v + -y), h = bloomTap(src, u + 0.0f, v + -y), i = bloomTap(src, u + x, v + -y);
        f3 up = e * 4.0f;
        up = up + add4(b, d, f, h) * 2.0f;
        up = up + add4(a, c, g, i);
        up = up * (1.0f / 16.0f);
        float *o = &dst.rgb[(size_t)p * 3];
        const f3 sum = f16Round(F3(o[0], o[1], o[2]) + up);
        o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
    }
}

__global__ void k_compose_tonemap(const float *__restrict__ post, const float *__restrict__ bloom0, uint32_t n, PtxPostProcessingUniformData u,
                                  uint32_t toneMode, float4 *__restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const f3 pp = F3(post[3 * (size_t)i], post[3 * (size_t)i + 1], post[3 * (size_t)i + 2]);
        const f3 bl = F3(bloom0[3 * (size_t)i], bloom0[3 * (size_t)i + 1], bloom0[3 * (size_t)i + 2]);
        const f3 c = f16Round(compositionPixel(pp, bl, u));
        const f3 t = f16Round(toneMapPixel(c, toneMode));
        out[i] = make_float4(t.x, t.y, t.z, 1.0f);
    }
}

__global__ void k_encode_srgb8(const float4 *__restrict__ linear, uint32_t n, uint32_t *__restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const float4 c = linear[i];
        out[i] = quantize8(linearToSrgb(c.x)) | quantize8(linearToSrgb(c.y)) << 8 | quantize8(linearToSrgb(c.z)) << 16 | quantize8(c.w) << 24;
    }
}

} // namespace ptd
