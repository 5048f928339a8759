/*
 * pt_oracle_post.c -- CPU ORACLE for the output stage (row N4).  TEST INFRASTRUCTURE ONLY, like
 * pt_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * Restates, pass by pass, what Renderer::RecordPostProcessCommands (Renderer.cpp:928-1085) and
 * RecordSaveOutputCommands (:1204-1246) run on the accumulation image:
 *   postprocess.comp:16-40, bloomDownsample.comp:18-62, bloomUpsample.comp:18-55,
 *   composition.comp:16-26, toneMapping.comp:13-25, then OutputSaver's blit into its sRGB8 or
 *   RGBA32F image (OutputSaver.cpp:64-86).
 * The reference's intermediate images are rgba16f: every store below rounds to binary16.
 *
 * PARITY PIN: the per-pixel arithmetic of postprocess.comp / composition.comp / toneMapping.comp is
 * pinned by golden vectors made from those shader lines (tools/gen_golden.py); the bloom taps and the
 * sRGB8 encode go through fixed-function hardware in the reference (sampler filtering, format
 * conversion in vkCmdBlitImage) and are restated from the Vulkan spec: unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pt_oracle.h"
#include "pt_oracle_math.h"

typedef struct Level
{
    uint32_t w, h;
    float *rgb; /* 3 floats per texel, binary16-valued */
} Level;

static inline v3 h3(v3 c) { return V3(pto_f16_round(c.x), pto_f16_round(c.y), pto_f16_round(c.z)); }

/* postprocess.comp:22-37: exposed colour and its soft-knee bloom prefilter */
void pto_postprocess_pixel(v3 accColor, const PtxPostProcessingUniformData *u, v3 *colorOut, v3 *bloomOut)
{
    v3 color = v_scale(v_div(accColor, (float)u->TotalSamples), u->Exposure);
    if (isnan(color.x) || isnan(color.y) || isnan(color.z))
        color = V3(5000.0f, 0.0f, 0.0f);
    if (isinf(color.x) || isinf(color.y) || isinf(color.z))
        color = V3(0.0f, 5000.0f, 0.0f);
    const float knee = 0.5f;
    const float threshold = u->BloomThreshold;
    const float br = f_max(color.x, f_max(color.y, color.z));
    const v3 curve = V3(threshold - knee, knee * 2.0f, pto_div(0.25f, knee));
    float rq = f_clamp(br - curve.x, 0.0f, curve.y);
    rq = curve.z * rq * rq;
    *bloomOut = v_scale(color, pto_div(f_max(rq, br - threshold), f_max(br, 0.0001f)));
    *colorOut = color;
}

/* composition.comp:23 */
v3 pto_composite_pixel(v3 postProcessColor, v3 bloomColor, const PtxPostProcessingUniformData *u)
{
    return v_add(v_scale(bloomColor, u->BloomIntensity * 0.1f), v_scale(postProcessColor, 1.0f));
}

/* toneMapping.comp:20-22 */
v3 pto_tonemap_pixel(v3 color, uint32_t mode)
{
    if (mode == PTX_TONE_MAPPING_HDR)
        return color;
    return V3(1.0f - pto_expf(-color.x), 1.0f - pto_expf(-color.y), 1.0f - pto_expf(-color.z));
}

/* texture(u_BloomSampler[level], uv): bilinear, clamp to edge (Renderer.cpp:114-119) */
static v3 bloomTap(const Level *L, float u, float v)
{
    const float x = u * (float)L->w - 0.5f, y = v * (float)L->h - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float ax = x - x0, ay = y - y0;
    const float mx = (float)(L->w - 1), my = (float)(L->h - 1);
    const uint32_t ix0 = (uint32_t)f_clamp(x0, 0.0f, mx), ix1 = (uint32_t)f_clamp(x0 + 1.0f, 0.0f, mx);
    const uint32_t iy0 = (uint32_t)f_clamp(y0, 0.0f, my), iy1 = (uint32_t)f_clamp(y0 + 1.0f, 0.0f, my);
    const float *p00 = &L->rgb[((size_t)iy0 * L->w + ix0) * 3], *p10 = &L->rgb[((size_t)iy0 * L->w + ix1) * 3];
    const float *p01 = &L->rgb[((size_t)iy1 * L->w + ix0) * 3], *p11 = &L->rgb[((size_t)iy1 * L->w + ix1) * 3];
    v3 r;
    float top, bot;
    top = p00[0] * (1.0f - ax) + p10[0] * ax; bot = p01[0] * (1.0f - ax) + p11[0] * ax; r.x = top * (1.0f - ay) + bot * ay;
    top = p00[1] * (1.0f - ax) + p10[1] * ax; bot = p01[1] * (1.0f - ax) + p11[1] * ax; r.y = top * (1.0f - ay) + bot * ay;
    top = p00[2] * (1.0f - ax) + p10[2] * ax; bot = p01[2] * (1.0f - ax) + p11[2] * ax; r.z = top * (1.0f - ay) + bot * ay;
    return r;
}

static inline v3 add4(v3 a, v3 b, v3 c, v3 d) { return v_add(v_add(v_add(a, b), c), d); }

/* bloomDownsample.comp:18-62 */
static void bloomDownsample(const Level *src, Level *dst)
{
    const float tx = pto_div(1.0f, (float)src->w), ty = pto_div(1.0f, (float)src->h);
#pragma omp parallel for schedule(static)
    for (int64_t yy = 0; yy < (int64_t)dst->h; yy++)
        for (uint32_t xx = 0; xx < dst->w; xx++)
        {
            const float u = pto_div((float)xx + 0.5f, (float)dst->w), v = pto_div((float)yy + 0.5f, (float)dst->h);
            const v3 a = bloomTap(src, u + -2.0f * tx, v + 2.0f * ty), b = bloomTap(src, u + 0.0f * tx, v + 2.0f * ty),
                     c = bloomTap(src, u + 2.0f * tx, v + 2.0f * ty);
            const v3 d = bloomTap(src, u + -2.0f * tx, v + 0.0f * ty), e = bloomTap(src, u + 0.0f * tx, v + 0.0f * ty),
                     f = bloomTap(src, u + 2.0f * tx, v + 0.0f * ty);
            const v3 g = bloomTap(src, u + -2.0f * tx, v + -2.0f * ty), h = bloomTap(src, u + 0.0f * tx, v + -2.0f * ty),
                     i = bloomTap(src, u + 2.0f * tx, v + -2.0f * ty);
            const v3 j = bloomTap(src, u + -1.0f * tx, v + 1.0f * ty), k = bloomTap(src, u + 1.0f * tx, v + 1.0f * ty);
            const v3 l = bloomTap(src, u + -1.0f * tx, v + -1.0f * ty), m = bloomTap(src, u + 1.0f * tx, v + -1.0f * ty);
            v3 down = v_scale(e, 0.125f);
            down = v_add(down, v_scale(add4(a, c, g, i), 0.03125f));
            down = v_add(down, v_scale(add4(b, d, f, h), 0.0625f));
            down = v_add(down, v_scale(add4(j, k, l, m), 0.125f));
            down = h3(down);
            float *o = &dst->rgb[((size_t)yy * dst->w + xx) * 3];
            o[0] = down.x; o[1] = down.y; o[2] = down.z;
        }
}

/* bloomUpsample.comp:18-55: 3x3 tent of the smaller level added onto the larger one */
static void bloomUpsample(const Level *src, Level *dst)
{
    const float x = pto_div(1.0f, (float)src->w), y = pto_div(1.0f, (float)src->h);
#pragma omp parallel for schedule(static)
    for (int64_t yy = 0; yy < (int64_t)dst->h; yy++)
        for (uint32_t xx = 0; xx < dst->w; xx++)
        {
            const float u = pto_div((float)xx + 0.5f, (float)dst->w), v = pto_div((float)yy + 0.5f, (float)dst->h);
            const v3 a = bloomTap(src, u + -x, v + y), b = bloomTap(src, u + 0.0f, v + y), c = bloomTap(src, u + x, v + y);
            const v3 d = bloomTap(src, u + -x, v + 0.0f), e = bloomTap(src, u + 0.0f, v + 0.0f), f = bloomTap(src, u + x, v + 0.0f);
            const v3 g = bloomTap(src, u + -x, v + -y), h = bloomTap(src, u + 0.0f, v + -y), i = bloomTap(src, u + x, v + -y);
            v3 up = v_scale(e, 4.0f);
            up = v_add(up, v_scale(add4(b, d, f, h), 2.0f));
            up = v_add(up, add4(a, c, g, i));
            up = v_scale(up, 1.0f / 16.0f);
            float *o = &dst->rgb[((size_t)yy * dst->w + xx) * 3];
            const v3 sum = h3(v_add(V3(o[0], o[1], o[2]), up));
            o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
        }
}

/* Renderer.cpp:928-1085 + :1204-1246.  out: W*H*4 floats (binary16-valued rgb, alpha 1). */
int pto_postprocess(const float *accum, uint32_t W, uint32_t H, const PtxPostProcessingUniformData *u, uint32_t toneMode, float *out)
{
    if (!accum || !u || !out || !W || !H)
        return 1;
    const size_t n = (size_t)W * H;
    float *post = (float *)malloc(n * 3 * sizeof(float));
    uint32_t levels = 1;
    for (uint32_t m = W > H ? W : H; m > 1; m >>= 1)
        levels++;
    Level L[13];
    uint32_t used = 0; /* mips 0 .. maxMipLevel-1 take part (Renderer.cpp:955-956) */
    if (levels >= 5)
        used = levels - 3 < 12 ? levels - 3 : 12;
    if (used == 0)
        used = 1; /* level 0 alone: the prefiltered colour is composited unblurred */
    for (uint32_t l = 0; l < used; l++)
    {
        L[l].w = (W >> l) ? (W >> l) : 1;
        L[l].h = (H >> l) ? (H >> l) : 1;
        L[l].rgb = (float *)malloc((size_t)L[l].w * L[l].h * 3 * sizeof(float));
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++)
    {
        v3 color, bloom;
        pto_postprocess_pixel(V3(accum[i * 4], accum[i * 4 + 1], accum[i * 4 + 2]), u, &color, &bloom);
        color = h3(color);
        bloom = h3(bloom);
        post[i * 3] = color.x; post[i * 3 + 1] = color.y; post[i * 3 + 2] = color.z;
        L[0].rgb[i * 3] = bloom.x; L[0].rgb[i * 3 + 1] = bloom.y; L[0].rgb[i * 3 + 2] = bloom.z;
    }
    for (uint32_t i = 0; i + 1 < used; i++)
        bloomDownsample(&L[i], &L[i + 1]);
    for (uint32_t i = used - 1; i > 0; i--)
        bloomUpsample(&L[i], &L[i - 1]);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++)
    {
        const v3 c = h3(pto_composite_pixel(V3(post[i * 3], post[i * 3 + 1], post[i * 3 + 2]),
                                            V3(L[0].rgb[i * 3], L[0].rgb[i * 3 + 1], L[0].rgb[i * 3 + 2]), u));
        const v3 t = h3(pto_tonemap_pixel(c, toneMode));
        out[i * 4] = t.x; out[i * 4 + 1] = t.y; out[i * 4 + 2] = t.z; out[i * 4 + 3] = 1.0f;
    }
    for (uint32_t l = 0; l < used; l++)
        free(L[l].rgb);
    free(post);
    return 0;
}

/* OutputSaver.cpp:64-86: the tone-mapped rgba16f image blitted into an sRGB8 or RGBA32F image */
int pto_encode_output(const float *linear, uint32_t W, uint32_t H, uint32_t format, void *out)
{
    if (!linear || !out)
        return 1;
    const size_t n = (size_t)W * H;
    if (format == PTX_OUTPUT_RGBA32F)
    {
        memcpy(out, linear, n * 16);
        return 0;
    }
    if (format != PTX_OUTPUT_RGBA8_SRGB)
        return 1;
    uint8_t *o = (uint8_t *)out;
    for (size_t i = 0; i < n; i++)
    {
        o[i * 4] = (uint8_t)quantize8(linearToSrgb(linear[i * 4]));
        o[i * 4 + 1] = (uint8_t)quantize8(linearToSrgb(linear[i * 4 + 1]));
        o[i * 4 + 2] = (uint8_t)quantize8(linearToSrgb(linear[i * 4 + 2]));
        o[i * 4 + 3] = (uint8_t)quantize8(linear[i * 4 + 3]);
    }
    return 0;
}

/* function-level entry for the golden vectors: which = 0 postprocess (in: acc.rgb, TotalSamples bits, Exposure,
 * BloomThreshold; out: color, bloom), 1 composition (in: post.rgb, bloom.rgb, BloomIntensity; out rgb),
 * 2 tone mapping SDR (in rgb; out rgb) */
int pto_test_post(uint32_t which, const float *in, float *out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        PtxPostProcessingUniformData u;
        memset(&u, 0, sizeof(u));
        if (which == 0)
        {
            const float *a = &in[(size_t)i * 6];
            u.TotalSamples = f2u(a[3]); u.Exposure = a[4]; u.BloomThreshold = a[5];
            v3 c, b;
            pto_postprocess_pixel(V3(a[0], a[1], a[2]), &u, &c, &b);
            float *o = &out[(size_t)i * 6];
            o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = b.x; o[4] = b.y; o[5] = b.z;
        }
        else if (which == 1)
        {
            const float *a = &in[(size_t)i * 7];
            u.BloomIntensity = a[6];
            const v3 c = pto_composite_pixel(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), &u);
            float *o = &out[(size_t)i * 3];
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
        }
        else if (which == 2)
        {
            const float *a = &in[(size_t)i * 3];
            const v3 c = pto_tonemap_pixel(V3(a[0], a[1], a[2]), PTX_TONE_MAPPING_SDR);
            float *o = &out[(size_t)i * 3];
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
        }
        else
            return 1;
    }
    return 0;
}
