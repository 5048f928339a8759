// pt_aux_kernels.hpp -- kernels beside the render path: texture upload (sRGB table, mip chain by linear blits, decode into the
// pool the render kernels sample) and the test entries (ptx_test_texture, ptx_test_eval: one production shading function on
// packed inputs, the equivalent of Path-Tracing-Tests/TestRenderer.cpp:79-105).
#pragma once

#include "pt_post.hpp"
#include "pt_wavefront.hpp"

// ---- textures (row N1): sRGB table and the mip chain are produced on the device ------------------
__global__ void k_build_srgb_lut(float *lut)
{
    const uint32_t c = threadIdx.x;
    if (c < 256)
        lut[c] = srgbToLinear((float)c / 255.0f);
}

// vkCmdBlitImage with a linear filter (Image.cpp:264-300, TextureUploader.cpp:479-490): every texel of level `dstLevel` of
// texture `dst` = the source level decoded and filtered bilinearly at the destination texel centre with clamp-to-edge,
// re-encoded in the image format.  One level of a mip chain is the blit from the level above it (src == dst).
__global__ void k_blit_level(TextureView tv, uint32_t src, uint32_t srcLevel, uint32_t dst, uint32_t dstLevel, uint32_t *texels8, float4 *texelsF)
{
    const DevTexture ts = tv.textures[src], td = tv.textures[dst];
    const uint32_t sw = levelDim(ts.width, srcLevel), sh = levelDim(ts.height, srcLevel);
    const uint32_t dw = levelDim(td.width, dstLevel), dh = levelDim(td.height, dstLevel);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= dw * dh)
        return;
    const uint32_t i = k % dw, j = k / dw;
    const float x = ((float)i + 0.5f) * ((float)sw / (float)dw) - 0.5f, y = ((float)j + 0.5f) * ((float)sh / (float)dh) - 0.5f;
    const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y), ax = x - x0, ay = y - y0;
    const float cx0 = clamp_(x0, 0.0f, (float)(sw - 1)), cx1 = clamp_(x0 + 1.0f, 0.0f, (float)(sw - 1));
    const float cy0 = clamp_(y0, 0.0f, (float)(sh - 1)), cy1 = clamp_(y0 + 1.0f, 0.0f, (float)(sh - 1));
    const f4 top = lerp4(fetchTexelEncoded(tv, ts, srcLevel, (uint32_t)cx0, (uint32_t)cy0), fetchTexelEncoded(tv, ts, srcLevel, (uint32_t)cx1, (uint32_t)cy0), ax);
    const f4 bot = lerp4(fetchTexelEncoded(tv, ts, srcLevel, (uint32_t)cx0, (uint32_t)cy1), fetchTexelEncoded(tv, ts, srcLevel, (uint32_t)cx1, (uint32_t)cy1), ax);
    const f4 c = lerp4(top, bot, ay);
    const size_t idx = (size_t)td.levelOffset[dstLevel] + (size_t)j * dw + i;
    if (td.format == PTX_TEXTURE_RGBA32F)
        texelsF[idx] = make_float4(c.x, c.y, c.z, c.w);
    else if (td.format == PTX_TEXTURE_RGBA8_SRGB)
        texels8[idx] = quantize8(linearToSrgb(c.x)) | quantize8(linearToSrgb(c.y)) << 8 | quantize8(linearToSrgb(c.z)) << 16 | quantize8(c.w) << 24;
    else
        texels8[idx] = quantize8(c.x) | quantize8(c.y) << 8 | quantize8(c.z) << 16 | quantize8(c.w) << 24;
}

// All levels of one 8-bit texture (a contiguous run of its pool) into the decoded pool the render kernels sample.
__global__ void k_decode_texels(const uint32_t *__restrict__ texels8, const float *__restrict__ srgbLut, uint32_t first, uint32_t count, uint32_t format,
                                float4 *__restrict__ decoded)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count)
        return;
    const uint32_t p = texels8[first + k];
    float4 r;
    if (format == PTX_TEXTURE_RGBA8_SRGB)
    {
        r.x = srgbLut[p & 255u]; r.y = srgbLut[(p >> 8) & 255u]; r.z = srgbLut[(p >> 16) & 255u];
    }
    else
    {
        r.x = (float)(p & 255u) / 255.0f; r.y = (float)((p >> 8) & 255u) / 255.0f; r.z = (float)((p >> 16) & 255u) / 255.0f;
    }
    r.w = (float)(p >> 24) / 255.0f;
    decoded[k] = r;
}

__global__ void k_test_texture(TextureView tv, const float *__restrict__ in, float *__restrict__ out, uint32_t n, int implicitLod)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float *a = in + (size_t)i * 7;
    const uint32_t idx = __float_as_uint(a[0]);
    f4 r;
    r.x = r.y = r.z = r.w = 1.0f;
    if (idx >= PTX_SCENE_TEXTURE_OFFSET && idx - PTX_SCENE_TEXTURE_OFFSET < tv.textureCount)
    {
        const DevTexture t = tv.textures[idx - PTX_SCENE_TEXTURE_OFFSET];
        r = implicitLod ? sampleLevel(tv, t, 0, a[1], a[2]) : textureGradSample(tv, t, a[1], a[2], a[3], a[4], a[5], a[6]);
    }
    out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
}

// function-level entry (Path-Tracing-Tests/TestRenderer.cpp:79-105 dispatches a compute
// shader that calls the production functions; packing documented in include/ptx.h)
__constant__ int c_inStride[PTX_FN_COUNT] = { 4, 4, 4, 2, 1, 10, 11, 6, 3, 14, 12, 4, 2, 2, 3, 6, 38, 1, 2, 31, 25, 42, 30, 24, 12, 34, 35, 4, 3, 3, 2, 6, 7, 3, 47, 2, 1, 1 };
__constant__ int c_outStride[PTX_FN_COUNT] = { 1, 1, 1, 1, 1, 4, 4, 3, 4, 4, 8, 5, 2, 3, 9, 3, 18, 2, 1, 9, 3, 18, 12, 6, 4, 12, 12, 1, 2, 3, 2, 6, 3, 3, 17, 2, 1, 1 };
static const int h_inStride[PTX_FN_COUNT] = { 4, 4, 4, 2, 1, 10, 11, 6, 3, 14, 12, 4, 2, 2, 3, 6, 38, 1, 2, 31, 25, 42, 30, 24, 12, 34, 35, 4, 3, 3, 2, 6, 7, 3, 47, 2, 1, 1 };
static const int h_outStride[PTX_FN_COUNT] = { 1, 1, 1, 1, 1, 4, 4, 3, 4, 4, 8, 5, 2, 3, 9, 3, 18, 2, 1, 9, 3, 18, 12, 6, 4, 12, 12, 1, 2, 3, 2, 6, 3, 3, 17, 2, 1, 1 };

PT_DEV MaterialSample unpackMaterial(const float *p)
{
    MaterialSample m;
    m.EmissiveColor = m.Normal = m.AttenuationColor = F3s(0.0f);
    m.AttenuationDistance = 0.0f;
    m.Color = F3(p[0], p[1], p[2]);
    m.Roughness = p[3];
    m.Metalness = p[4];
    m.Transmission = p[5];
    m.Eta = p[6];
    return m;
}

__global__ void k_test_eval(uint32_t fn, const float *__restrict__ in, float *__restrict__ out, uint32_t n, PtxLightsUbo *scratchUbo)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float *a = in + (size_t)i * c_inStride[fn];
    float *o = out + (size_t)i * c_outStride[fn];
    switch (fn)
    {
    case PTX_FN_GGX_DISTRIBUTION: o[0] = GGXDistribution(F3(a[0], a[1], a[2]), a[3]); break;
    case PTX_FN_LAMBDA: o[0] = Lambda(F3(a[0], a[1], a[2]), a[3]); break;
    case PTX_FN_GGX_SMITH: o[0] = GGXSmith(F3(a[0], a[1], a[2]), a[3]); break;
    case PTX_FN_DIELECTRIC_FRESNEL: o[0] = DielectricFresnel(a[0], a[1]); break;
    case PTX_FN_SCHLICK_FRESNEL: o[0] = SchlickFresnel(a[0]); break;
    case PTX_FN_EVALUATE_REFLECTION: {
        float pdf;
        const f3 r = EvaluateReflection(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), F3(a[6], a[7], a[8]), a[9], pdf);
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf;
        break;
    }
    case PTX_FN_EVALUATE_REFRACTION: {
        float pdf;
        const f3 r = EvaluateRefraction(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), F3(a[6], a[7], a[8]), a[9], a[10], pdf);
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf;
        break;
    }
    case PTX_FN_SAMPLE_GGX: {
        f2 u; u.x = a[0]; u.y = a[1];
        const f3 r = SampleGGX(u, F3(a[2], a[3], a[4]), a[5]);
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PTX_FN_SAMPLE_LOBE_PDFS: { // bsdf.glsl:62-70
        const float metal = a[0], trans = a[1], F = a[2];
        o[0] = (1.0f - metal) * (1.0f - F) * (1.0f - trans);
        o[1] = (1.0f - metal) * F;
        o[2] = metal;
        o[3] = (1.0f - metal) * (1.0f - F) * trans;
        break;
    }
    case PTX_FN_EVALUATE_BSDF: {
        const MaterialSample m = unpackMaterial(a);
        float pdf;
        const f3 r = evaluateBSDF(m, F3(a[8], a[9], a[10]), F3(a[11], a[12], a[13]), pdf);
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf;
        break;
    }
    case PTX_FN_SAMPLE_BSDF: {
        const MaterialSample m = unpackMaterial(a);
        uint32_t rng = __float_as_uint(a[11]);
        const BSDFSample r = sampleBSDF(m, F3(a[8], a[9], a[10]), rng);
        o[0] = r.Direction.x; o[1] = r.Direction.y; o[2] = r.Direction.z; o[3] = r.Pdf;
        o[4] = r.Color.x; o[5] = r.Color.y; o[6] = r.Color.z; o[7] = __uint_as_float(rng);
        break;
    }
    case PTX_FN_RNG: {
        uint32_t stt = initRng(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]));
        o[0] = __uint_as_float(stt);
        for (int k = 0; k < 4; k++)
            o[1 + k] = rnd(stt);
        break;
    }
    case PTX_FN_DISK: {
        f2 u; u.x = a[0]; u.y = a[1];
        const f2 d = sampleUniformDiskConcentric(u);
        o[0] = d.x; o[1] = d.y;
        break;
    }
    case PTX_FN_COS_HEMISPHERE: {
        f2 u; u.x = a[0]; u.y = a[1];
        const f3 d = sampleCosineHemisphere(u);
        o[0] = d.x; o[1] = d.y; o[2] = d.z;
        break;
    }
    case PTX_FN_TANGENT_SPACE: {
        const mat3 m = computeTangentSpace(F3(a[0], a[1], a[2]));
        o[0] = m.c0.x; o[1] = m.c0.y; o[2] = m.c0.z;
        o[3] = m.c1.x; o[4] = m.c1.y; o[5] = m.c1.z;
        o[6] = m.c2.x; o[7] = m.c2.y; o[8] = m.c2.z;
        break;
    }
    case PTX_FN_OFFSET_SELF_INTERSECTION: {
        const f3 r = offsetRayOriginSelfIntersection(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]));
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PTX_FN_PRIMARY_RAY: {
        f2 u; u.x = a[4]; u.y = a[5];
        f3 ro, rd, rx, ry;
        constructPrimaryRay<true>(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]), &a[6], &a[22], u, ro, rd,
                                  rx, ry);
        const f3 v[6] = { ro, rd, ro, rx, ro, ry };
        for (int k = 0; k < 6; k++) { o[3 * k] = v[k].x; o[3 * k + 1] = v[k].y; o[3 * k + 2] = v[k].z; }
        break;
    }
    case PTX_FN_SINCOS: sincos_(a[0], o[0], o[1]); break;
    case PTX_FN_POW: o[0] = pow_(a[0], a[1]); break;
    case PTX_FN_SAMPLE_LIGHT: {
        PtxLightsUbo *ubo = &scratchUbo[i];
        ubo->LightCount = __float_as_uint(a[6]);
        for (int k = 0; k < 3; k++)
        {
            ubo->Directional.Color[k] = a[7 + k];
            ubo->Directional.Direction[k] = a[10 + k];
        }
        for (int l = 0; l < 2; l++)
        {
            for (int k = 0; k < 3; k++)
            {
                ubo->Lights[l].Color[k] = a[13 + 9 * l + k];
                ubo->Lights[l].Position[k] = a[16 + 9 * l + k];
            }
            ubo->Lights[l].AttenuationConstant = a[19 + 9 * l];
            ubo->Lights[l].AttenuationLinear = a[20 + 9 * l];
            ubo->Lights[l].AttenuationQuadratic = a[21 + 9 * l];
        }
        float pdf;
        const LightSample ls = sampleLight(ubo, F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), pdf);
        o[0] = ls.Direction.x; o[1] = ls.Direction.y; o[2] = ls.Direction.z; o[3] = ls.Distance;
        o[4] = ls.Color.x; o[5] = ls.Color.y; o[6] = ls.Color.z; o[7] = ls.Attenuation; o[8] = pdf;
        break;
    }
    case PTX_FN_SHADOW_TERMINATOR: {
        const f3 r = offsetRayOriginShadowTerminator(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), F3(a[6], a[7], a[8]), F3(a[9], a[10], a[11]),
                                                     F3(a[12], a[13], a[14]), F3(a[15], a[16], a[17]), F3(a[18], a[19], a[20]),
                                                     F3(a[21], a[22], a[23]), a[24] != 0.0f);
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PTX_FN_PRIMARY_RAY_LENS: {
        f2 u, u2; u.x = a[4]; u.y = a[5]; u2.x = a[6]; u2.y = a[7];
        f3 ro, rd, rx, ry;
        constructPrimaryRayLens<true>(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]), &a[10], &a[26],
                                      u, u2, a[8], a[9], ro, rd, rx, ry);
        const f3 v[6] = { ro, rd, ro, rx, ro, ry };
        for (int k = 0; k < 6; k++) { o[3 * k] = v[k].x; o[3 * k + 1] = v[k].y; o[3 * k + 2] = v[k].z; }
        break;
    }
    case PTX_FN_DPN_DUV: {
        f3 P[3], N[3];
        f2 UV[3];
        for (int k = 0; k < 3; k++)
        {
            P[k] = F3(a[8 * k], a[8 * k + 1], a[8 * k + 2]);
            N[k] = F3(a[8 * k + 3], a[8 * k + 4], a[8 * k + 5]);
            UV[k].x = a[8 * k + 6];
            UV[k].y = a[8 * k + 7];
        }
        f3 r0, r1, r2, r3;
        computeDpnDuv(P, N, UV, F3(a[24], a[25], a[26]), F3(a[27], a[28], a[29]), r0, r1, r2, r3);
        o[0] = r0.x; o[1] = r0.y; o[2] = r0.z; o[3] = r1.x; o[4] = r1.y; o[5] = r1.z;
        o[6] = r2.x; o[7] = r2.y; o[8] = r2.z; o[9] = r3.x; o[10] = r3.y; o[11] = r3.z;
        break;
    }
    case PTX_FN_DP_DXY: {
        f3 dx, dy;
        computeDpDxy(F3(a[0], a[1], a[2]), F3(a[9], a[10], a[11]), F3(a[12], a[13], a[14]), F3(a[15], a[16], a[17]), F3(a[18], a[19], a[20]),
                     F3(a[21], a[22], a[23]), dx, dy);
        o[0] = dx.x; o[1] = dx.y; o[2] = dx.z; o[3] = dy.x; o[4] = dy.y; o[5] = dy.z;
        break;
    }
    case PTX_FN_DERIVATIVES: {
        const f4 r = computeDerivatives(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), F3(a[6], a[7], a[8]), F3(a[9], a[10], a[11]));
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
        break;
    }
    case PTX_FN_REFLECTED_DIFFERENTIALS:
    case PTX_FN_REFRACTED_DIFFERENTIALS: {
        f4 dv; dv.x = a[0]; dv.y = a[1]; dv.z = a[2]; dv.w = a[3];
        DiffRays r;
        r.rxOrigin = F3(a[22], a[23], a[24]); r.rxDirection = F3(a[25], a[26], a[27]);
        r.ryOrigin = F3(a[28], a[29], a[30]); r.ryDirection = F3(a[31], a[32], a[33]);
        if (fn == PTX_FN_REFLECTED_DIFFERENTIALS)
            computeReflectedDifferentialRays(dv, F3(a[4], a[5], a[6]), F3(a[7], a[8], a[9]), F3(a[10], a[11], a[12]), F3(a[13], a[14], a[15]),
                                             F3(a[16], a[17], a[18]), F3(a[19], a[20], a[21]), r);
        else
            computeRefractedDifferentialRays(dv, F3(a[4], a[5], a[6]), F3(a[7], a[8], a[9]), F3(a[10], a[11], a[12]), F3(a[13], a[14], a[15]),
                                             F3(a[16], a[17], a[18]), F3(a[19], a[20], a[21]), a[34], r);
        o[0] = r.rxOrigin.x; o[1] = r.rxOrigin.y; o[2] = r.rxOrigin.z; o[3] = r.rxDirection.x; o[4] = r.rxDirection.y; o[5] = r.rxDirection.z;
        o[6] = r.ryOrigin.x; o[7] = r.ryOrigin.y; o[8] = r.ryOrigin.z; o[9] = r.ryDirection.x; o[10] = r.ryDirection.y; o[11] = r.ryDirection.z;
        break;
    }
    case PTX_FN_SKYBOX_TEXCOORDS: {
        const f2 uv = missSkyboxTexCoords(F3(a[0], a[1], a[2]));
        o[0] = uv.x; o[1] = uv.y;
        break;
    }
    case PTX_FN_HDR_TO_LDR: {
        const f3 r = hdrToLdr(F3(a[0], a[1], a[2]));
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PTX_FN_ATAN_ASIN:
        o[0] = atan2_(a[0], a[1]);
        o[1] = asin_(a[0]);
        break;
    case PTX_FN_DIVIDE:
        o[0] = rcp_(a[1]);
        o[1] = div_(a[0], a[1]);
        break;
    case PTX_FN_SQRT: o[0] = sqrt_(a[0]); break;
    case PTX_FN_RSQ: o[0] = rsq_(a[0]); break;
    case PTX_FN_POSTPROCESS_PIXEL: {
        PtxPostProcessingUniformData u;
        u.TotalSamples = __float_as_uint(a[3]); u.Exposure = a[4]; u.BloomThreshold = a[5]; u.BloomIntensity = 0.0f;
        f3 c, b;
        postprocessPixel(F3(a[0], a[1], a[2]), u, c, b);
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = b.x; o[4] = b.y; o[5] = b.z;
        break;
    }
    case PTX_FN_COMPOSITION_PIXEL: {
        PtxPostProcessingUniformData u;
        u.TotalSamples = 1u; u.Exposure = u.BloomThreshold = 0.0f; u.BloomIntensity = a[6];
        const f3 c = compositionPixel(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), u);
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
        break;
    }
    case PTX_FN_TONEMAP_PIXEL: {
        const f3 c = toneMapPixel(F3(a[0], a[1], a[2]), PTX_TONE_MAPPING_SDR);
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
        break;
    }
    case PTX_FN_COMPUTE_LOD: {
        f4 dv; dv.x = a[0]; dv.y = a[1]; dv.z = a[2]; dv.w = a[3];
        o[0] = computeLod(dv);
        break;
    }
    case PTX_FN_SAMPLE_MATERIAL: {
        const uint32_t type = __float_as_uint(a[0]);
        const bool inside = __float_as_uint(a[1]) != 0u, flip = __float_as_uint(a[2]) != 0u;
        MaterialTexels t;
        f4 *tx[5] = { &t.emissive, &t.color, &t.normal, &t.a, &t.b };
        for (int k = 0; k < 5; k++)
        {
            tx[k]->x = a[27 + 4 * k]; tx[k]->y = a[28 + 4 * k]; tx[k]->z = a[29 + 4 * k]; tx[k]->w = a[30 + 4 * k];
        }
        MaterialSample m;
        if (type == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS)
            m = sampleMaterial(reinterpret_cast<const PtxMetallicRoughnessMaterial *>(a + 3), t, inside);
        else if (type == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS)
            m = sampleMaterial(reinterpret_cast<const PtxSpecularGlossinessMaterial *>(a + 3), t, inside);
        else if (type == PTX_MATERIAL_TYPE_PHONG)
            m = sampleMaterial(reinterpret_cast<const PtxPhongMaterial *>(a + 3), t, inside);
        else
            m = unknownMaterial();
        if (flip)
            m.Normal.y *= -1;
        o[0] = m.EmissiveColor.x; o[1] = m.EmissiveColor.y; o[2] = m.EmissiveColor.z; o[3] = m.Color.x; o[4] = m.Color.y; o[5] = m.Color.z;
        o[6] = m.Normal.x; o[7] = m.Normal.y; o[8] = m.Normal.z; o[9] = m.Roughness; o[10] = m.Metalness; o[11] = m.Transmission; o[12] = m.Eta;
        o[13] = m.AttenuationColor.x; o[14] = m.AttenuationColor.y; o[15] = m.AttenuationColor.z; o[16] = m.AttenuationDistance;
        break;
    }
    default: break;
    }
}

