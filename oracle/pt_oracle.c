/*
 * pt_oracle.c -- CPU ORACLE for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY
 * (see pt_oracle.h for the rules, the parity pin and the arithmetic conventions).
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "pt_oracle.h"
#include "pt_oracle_math.h"

#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* common.glsl                                                              */
/* ======================================================================== */

/* common.glsl:12-15 */
static inline float maxComponent(v3 rgb) { return f_max(rgb.x, f_max(rgb.y, rgb.z)); }

/* common.glsl:133-141 */
static inline uint32_t jenkinsHash(uint32_t x)
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}

/* common.glsl:143-147 -- dot(uvec2, uvec2) goes through float */
static inline uint32_t initRng(uint32_t px, uint32_t py, uint32_t resX, uint32_t frame)
{
    const float d = (float)px * 1.0f + (float)py * (float)resX;
    const uint32_t rngState = (uint32_t)d ^ jenkinsHash(frame);
    return jenkinsHash(rngState);
}

/* common.glsl:149-152 */
static inline float uintToFloat(uint32_t x) { return u2f(0x3f800000u | (x >> 9)) - 1.0f; }

/* common.glsl:154-160 */
static inline uint32_t xorshift(uint32_t *rngState)
{
    *rngState ^= *rngState << 13;
    *rngState ^= *rngState >> 17;
    *rngState ^= *rngState << 5;
    return *rngState;
}

/* common.glsl:162-165 */
static inline float rnd(uint32_t *rngState) { return uintToFloat(xorshift(rngState)); }

/* common.glsl:168-184 */
static inline v2 sampleUniformDiskConcentric(v2 u)
{
    v2 offset = { 2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f };
    v2 r = { 0.0f, 0.0f };
    if (offset.x == 0.0f && offset.y == 0.0f)
        return r;
    float s, c;
    if (fabsf(offset.x) > fabsf(offset.y))
    {
        const float theta = (pto_div(PTO_PI, 4)) * (pto_div(offset.y, offset.x));
        pto_sincosf(theta, &s, &c);
        r.x = offset.x * c;
        r.y = offset.x * s;
    }
    else
    {
        const float theta = pto_div(PTO_PI, 2) - (pto_div(PTO_PI, 4)) * (pto_div(offset.x, offset.y));
        pto_sincosf(theta, &s, &c);
        r.x = offset.y * c;
        r.y = offset.y * s;
    }
    return r;
}

/* common.glsl:186-191 */
static inline v3 sampleCosineHemisphere(v2 u)
{
    const v2 d = sampleUniformDiskConcentric(u);
    const float z = sqrtf(1 - d.x * d.x - d.y * d.y);
    return V3(d.x, d.y, z);
}

/* common.glsl:193-202 */
static inline m3 computeTangentSpace(v3 normal)
{
    const v3 t1 = v_cross(normal, V3(1.0f, 0.0f, 0.0f));
    const v3 t2 = v_cross(normal, V3(0.0f, 1.0f, 0.0f));
    const v3 tangent = v_length(t1) > v_length(t2) ? t1 : t2;
    const v3 bitangent = v_cross(normal, tangent);
    m3 m;
    m.c0 = v_normalize(tangent);
    m.c1 = v_normalize(bitangent);
    m.c2 = normal;
    return m;
}

/* ======================================================================== */
/* shading.glsl                                                             */
/* ======================================================================== */

/* shading.glsl:3-14 -- note the max(denom, 1): D is clamped to <= 1 (kept quirk) */
static inline float GGXDistribution(v3 H, float alpha)
{
    const float Hx2 = H.x * H.x;
    const float Hy2 = H.y * H.y;
    const float Hz2 = H.z * H.z;
    const float alpha2 = alpha * alpha;
    const float b = pto_div(Hx2, alpha2) + pto_div(Hy2, alpha2) + Hz2;
    const float denom = PTO_PI * alpha2 * (b * b);
    return pto_div(1.0f, f_max(denom, 1.0f));
}

/* shading.glsl:16-27 */
static inline float Lambda(v3 V, float alpha)
{
    const float Vx2 = V.x * V.x;
    const float Vy2 = V.y * V.y;
    const float Vz2 = fabsf(V.z) * fabsf(V.z);
    const float alpha2 = alpha * alpha;
    const float nom = sqrtf(1.0f + pto_div(alpha2 * Vx2 + alpha2 * Vy2, Vz2)) - 1.0f;
    return pto_div(nom, 2.0f);
}

/* shading.glsl:29-32 */
static inline float GGXSmith(v3 V, float alpha) { return pto_div(1.0f, 1.0f + Lambda(V, alpha)); }

/* shading.glsl:34-48 */
static inline float DielectricFresnel(float VdotH, float eta)
{
    const float cosThetaI = VdotH;
    const float sinThetaT2 = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaT2 > 1.0f)
        return 1.0f;
    const float cosThetaT = sqrtf(f_max(1.0f - sinThetaT2, 0.0f));
    const float rs = pto_div(eta * cosThetaT - cosThetaI, eta * cosThetaT + cosThetaI);
    const float rp = pto_div(eta * cosThetaI - cosThetaT, eta * cosThetaI + cosThetaT);
    return pto_div(rs * rs + rp * rp, 2.0f);
}

/* shading.glsl:50-53 -- pow(x, 5) = x2*x2*x */
static inline float SchlickFresnel(float VdotH)
{
    const float x = f_clamp(1.0f - VdotH, 0.0f, 1.0f);
    const float x2 = x * x;
    return x2 * x2 * x;
}

/* shading.glsl:56-77 */
static inline v3 EvaluateReflection(v3 V, v3 L, v3 F, float alpha, float *pdf)
{
    if (L.z < 0.00001f)
    {
        *pdf = 0.0f;
        return v3s(0.0f);
    }
    const v3 H = v_normalize(v_add(V, L));
    const float VdotH = v_dot(V, H);
    const float D = GGXDistribution(H, alpha);
    const float Gv = GGXSmith(V, alpha);
    const float Gl = GGXSmith(L, alpha);
    const float G = Gv * Gl;
    const float Dv = pto_div(Gv * f_max(VdotH, 0.0f) * D, V.z);
    *pdf = pto_div(Dv, 4.0f * VdotH);
    return v_div(v_scale(F, D * G), 4.0f * V.z);
}

/* shading.glsl:80-108 -- pow(x, 2) = x*x */
static inline v3 EvaluateRefraction(v3 V, v3 L, v3 F, float alpha, float eta, float *pdf)
{
    if (L.z > -0.00001f)
    {
        *pdf = 0.0f;
        return v3s(0.0f);
    }
    v3 H = v_normalize(v_add(v_scale(V, eta), L));
    if (H.z < 0.0f)
        H = v_neg(H);
    const float VdotH = v_dot(V, H);
    const float LdotH = v_dot(L, H);
    const float D = GGXDistribution(H, alpha);
    const float Gv = GGXSmith(V, alpha);
    const float Gl = GGXSmith(L, alpha);
    const float G = Gv * Gl;
    const float Dv = pto_div(Gv * fabsf(VdotH) * D, V.z);
    const float denominator = LdotH + eta * VdotH;
    const float jacobian = pto_div((eta * eta) * fabsf(LdotH), denominator * denominator);
    *pdf = Dv * jacobian;
    return v_scale(v_scale(v_scale(F, D * G), pto_div(fabsf(VdotH), fabsf(V.z))), jacobian);
}

/* shading.glsl:111-129 */
static inline v3 SampleGGX(v2 u, v3 V, float alpha)
{
    const v3 Vh = v_normalize(V3(alpha * V.x, alpha * V.y, fabsf(V.z)));
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    const v3 T1 = lensq > 0 ? v_scale(V3(-Vh.y, Vh.x, 0), pto_rsq(lensq)) : V3(1, 0, 0); /* inversesqrt */
    const v3 T2 = v_cross(Vh, T1);
    const float r = sqrtf(u.x);
    const float phi = 2.0f * PTO_PI * u.y;
    float sn, cs;
    pto_sincosf(phi, &sn, &cs);
    const float t1 = r * cs;
    float t2 = r * sn;
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    const v3 Nh =
        v_add(v_add(v_scale(T1, t1), v_scale(T2, t2)), v_scale(Vh, sqrtf(f_max(0.0f, 1.0f - t1 * t1 - t2 * t2))));
    return v_normalize(V3(alpha * Nh.x, alpha * Nh.y, f_max(0.0f, Nh.z)));
}

/* ======================================================================== */
/* bsdf.glsl                                                                */
/* ======================================================================== */

/* ShaderRendererTypes.incl:129-140 */
typedef struct MaterialSample
{
    v3 EmissiveColor;
    v3 Color;
    v3 Normal;
    float Roughness;
    float Metalness;
    float Transmission;
    float Eta;
    v3 AttenuationColor;
    float AttenuationDistance;
} MaterialSample;

typedef struct BSDFSample
{
    v3 Direction;
    float Pdf;
    v3 Color;
} BSDFSample;

typedef struct LobePdfs
{
    float Diffuse, Glossy, Metallic, Transmissive;
} LobePdfs;

/* bsdf.glsl:11-15 */
static inline v3 evaluateDiffuseBRDF(const MaterialSample *m, v3 V, v3 L, float *pdf)
{
    (void)V;
    *pdf = pto_div(L.z * 1.0f, PTO_PI);
    return v_div(v_scale(m->Color, L.z), PTO_PI);
}

/* bsdf.glsl:22-25 */
static inline v3 evaluateGlossyBSDF(const MaterialSample *m, v3 V, v3 L, float *pdf)
{
    return EvaluateReflection(V, L, v3s(1.0f), m->Roughness * m->Roughness, pdf);
}

/* bsdf.glsl:32-37 */
static inline v3 evaluateMetallicBRDF(const MaterialSample *m, v3 V, v3 L, float *pdf)
{
    const v3 H = v_normalize(v_add(V, L));
    const v3 F0 = v_mix(m->Color, v3s(1.0f), SchlickFresnel(v_dot(V, H)));
    return EvaluateReflection(V, L, F0, m->Roughness * m->Roughness, pdf);
}

/* bsdf.glsl:44-47 */
static inline v3 evaluateBTDF(const MaterialSample *m, v3 V, v3 L, float *pdf)
{
    return EvaluateRefraction(V, L, m->Color, m->Roughness * m->Roughness, m->Eta, pdf);
}

/* bsdf.glsl:62-70 */
static inline LobePdfs sampleLobePdfs(const MaterialSample *m, float F)
{
    LobePdfs p;
    p.Diffuse = (1.0f - m->Metalness) * (1.0f - F) * (1.0f - m->Transmission);
    p.Glossy = (1.0f - m->Metalness) * F;
    p.Metallic = m->Metalness;
    p.Transmissive = (1.0f - m->Metalness) * (1.0f - F) * m->Transmission;
    return p;
}

/* bsdf.glsl:72-103 */
static inline v3 evaluateBSDF(const MaterialSample *m, v3 V, v3 L, float *outPdf)
{
    const int isReflection = L.z > 0.0f;
    const v3 H = isReflection ? v_normalize(v_add(V, L)) : v_normalize(v_add(v_scale(V, m->Eta), L));
    const float FD = DielectricFresnel(fabsf(v_dot(V, H)), m->Eta);
    const LobePdfs pdfs = sampleLobePdfs(m, FD);

    v3 bsdf = v3s(0.0f);
    *outPdf = 0.0f;
    float pdf;
    if (isReflection)
    {
        bsdf = v_add(bsdf, v_scale(evaluateDiffuseBRDF(m, V, L, &pdf), pdfs.Diffuse));
        *outPdf += pdf * pdfs.Diffuse;
        bsdf = v_add(bsdf, v_scale(evaluateGlossyBSDF(m, V, L, &pdf), pdfs.Glossy));
        *outPdf += pdf * pdfs.Glossy;
        bsdf = v_add(bsdf, v_scale(evaluateMetallicBRDF(m, V, L, &pdf), pdfs.Metallic));
        *outPdf += pdf * pdfs.Metallic;
    }
    else
    {
        bsdf = v_add(bsdf, v_scale(evaluateBTDF(m, V, L, &pdf), pdfs.Transmissive));
        *outPdf += pdf * pdfs.Transmissive;
    }
    return bsdf;
}

/* bsdf.glsl:105-132 -- nested conditional draws; the draw order is part of the contract */
static inline BSDFSample sampleBSDF(const MaterialSample *m, v3 V, uint32_t *rngState)
{
    const float alpha = m->Roughness * m->Roughness;
    v2 u;
    u.x = rnd(rngState);
    u.y = rnd(rngState);
    const v3 H = SampleGGX(u, V, alpha);
    const float FD = DielectricFresnel(fabsf(v_dot(V, H)), m->Eta);

    v3 L;
    if (rnd(rngState) < m->Metalness)
        L = v_normalize(v_reflect(v_neg(V), H)); /* bsdf.glsl:39-42 */
    else
    {
        if (rnd(rngState) < FD)
            L = v_normalize(v_reflect(v_neg(V), H)); /* bsdf.glsl:27-30 */
        else
        {
            if (rnd(rngState) < m->Transmission)
                L = v_normalize(v_refract(v_neg(V), H, m->Eta)); /* bsdf.glsl:49-52 */
            else
            {
                v2 u2;
                u2.x = rnd(rngState);
                u2.y = rnd(rngState);
                L = sampleCosineHemisphere(u2); /* bsdf.glsl:17-20 */
            }
        }
    }

    BSDFSample ret;
    ret.Direction = L;
    ret.Color = evaluateBSDF(m, V, L, &ret.Pdf);
    return ret;
}

/* ======================================================================== */
/* ray.glsl                                                                 */
/* ======================================================================== */

typedef struct Ray
{
    v3 Origin;
    float tmin;
    v3 Direction;
    float tmax;
} Ray;

/* direction through pixel-centre coordinates (pcx, pcy): ray.glsl:64-65,72-73 */
static inline v3 pinholeDirection(float pcx, float pcy, uint32_t resX, uint32_t resY, const float *ViewInverse, const float *ProjInverse)
{
    const float inUVx = pto_div(pcx, (float)resX);
    const float inUVy = pto_div(pcy, (float)resY);
    const float dx = inUVx * 2.0f - 1.0f;
    const float dy = inUVy * 2.0f - 1.0f;
    const v4 target = m4_mul(ProjInverse, dx, dy, 1, 1);
    const v3 nt = v_normalize(V3(target.x, target.y, target.z));
    const v4 direction = m4_mul(ViewInverse, nt.x, nt.y, nt.z, 0);
    return V3(direction.x, direction.y, direction.z);
}

/* ray.glsl:58-85 (pinhole); rx / ry are the rays through the pixels one to the right / below */
static inline Ray constructPrimaryRayD(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                       const float *ProjInverse, v2 u, Ray *rx, Ray *ry)
{
    const float pcx = (float)px + u.x;
    const float pcy = (float)py + u.y;
    const v4 origin = m4_mul(ViewInverse, 0, 0, 0, 1);
    Ray r;
    r.Origin = V3(origin.x, origin.y, origin.z);
    r.tmin = 0.00001f;
    r.Direction = pinholeDirection(pcx, pcy, resX, resY, ViewInverse, ProjInverse);
    r.tmax = 10000.0f;
    if (rx)
    {
        *rx = r;
        rx->Direction = pinholeDirection(pcx + 1.0f, pcy + 0.0f, resX, resY, ViewInverse, ProjInverse);
        *ry = r;
        ry->Direction = pinholeDirection(pcx + 0.0f, pcy + 1.0f, resX, resY, ViewInverse, ProjInverse);
    }
    return r;
}
static inline Ray constructPrimaryRay(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                      const float *ProjInverse, v2 u)
{
    return constructPrimaryRayD(px, py, resX, resY, ViewInverse, ProjInverse, u, NULL, NULL);
}

/* thin-lens direction through (pcx, pcy): ray.glsl:24-25,35-38 */
static inline v3 lensDirection(float pcx, float pcy, uint32_t resX, uint32_t resY, const float *ViewInverse, const float *ProjInverse,
                               v3 originCameraSpace, float focalDistance)
{
    const float inUVx = pto_div(pcx, (float)resX);
    const float inUVy = pto_div(pcy, (float)resY);
    const float dx = inUVx * 2.0f - 1.0f;
    const float dy = inUVy * 2.0f - 1.0f;
    const v4 target = m4_mul(ProjInverse, dx, dy, 1, 1);
    const float ft = pto_div(focalDistance, target.z);
    const v3 pFocus = v_scale(V3(target.x, target.y, target.z), ft);
    const v3 nd = v_normalize(v_sub(pFocus, originCameraSpace));
    const v4 direction = m4_mul(ViewInverse, nd.x, nd.y, nd.z, 0);
    return V3(direction.x, direction.y, direction.z);
}

/* ray.glsl:16-56 (thin lens) */
static inline Ray constructPrimaryRayLensD(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                           const float *ProjInverse, v2 u, v2 u2, float lensRadius, float focalDistance, Ray *rx,
                                           Ray *ry)
{
    const float pcx = (float)px + u.x;
    const float pcy = (float)py + u.y;
    const v2 disk = sampleUniformDiskConcentric(u2);
    const v2 pLens = { lensRadius * disk.x, lensRadius * disk.y };
    const v3 originCameraSpace = V3(pLens.x, pLens.y, 0);
    const v4 origin = m4_mul(ViewInverse, originCameraSpace.x, originCameraSpace.y, originCameraSpace.z, 1);
    Ray r;
    r.Origin = V3(origin.x, origin.y, origin.z);
    r.tmin = 0.00001f;
    r.Direction = lensDirection(pcx, pcy, resX, resY, ViewInverse, ProjInverse, originCameraSpace, focalDistance);
    r.tmax = 10000.0f;
    if (rx)
    {
        *rx = r;
        rx->Direction = lensDirection(pcx + 1.0f, pcy + 0.0f, resX, resY, ViewInverse, ProjInverse, originCameraSpace, focalDistance);
        *ry = r;
        ry->Direction = lensDirection(pcx + 0.0f, pcy + 1.0f, resX, resY, ViewInverse, ProjInverse, originCameraSpace, focalDistance);
    }
    return r;
}
static inline Ray constructPrimaryRayLens(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                          const float *ProjInverse, v2 u, v2 u2, float lensRadius, float focalDistance)
{
    return constructPrimaryRayLensD(px, py, resX, resY, ViewInverse, ProjInverse, u, u2, lensRadius, focalDistance, NULL, NULL);
}

/* ray.glsl:93-106 (Waechter-Binder) */
static inline float offsetComponent(float o, float n)
{
    const float origin_const = 1.0f / 32.0f;
    const float float_scale = 1.0f / 65536.0f;
    const float int_scale = 256.0f;
    const int32_t of_i = (int32_t)(int_scale * n);
    const uint32_t bits = f2u(o) + (uint32_t)((o < 0) ? -of_i : of_i);
    const float p_i = u2f(bits);
    return (fabsf(o) < origin_const) ? o + float_scale * n : p_i;
}
static inline v3 offsetRayOriginSelfIntersection(v3 origin, v3 normal)
{
    return V3(offsetComponent(origin.x, normal.x), offsetComponent(origin.y, normal.y),
              offsetComponent(origin.z, normal.z));
}

typedef struct Vtx
{
    v3 Position;
    v2 TexCoords;
    v3 Normal, Tangent, Bitangent;
} Vtx;

/* ray.glsl:109-131 (Hanika shadow terminator) */
static inline v3 offsetRayOriginShadowTerminator(const Vtx *vertex, const Vtx *v0, const Vtx *v1, const Vtx *v2,
                                                 v3 bary, int isRefracted)
{
    v3 tmpu = v_sub(vertex->Position, v0->Position);
    v3 tmpv = v_sub(vertex->Position, v1->Position);
    v3 tmpw = v_sub(vertex->Position, v2->Position);
    v3 n0 = v0->Normal, n1 = v1->Normal, n2 = v2->Normal;
    if (isRefracted)
    {
        n0 = v_neg(n0);
        n1 = v_neg(n1);
        n2 = v_neg(n2);
    }
    const float dotu = f_min(0.0f, v_dot(tmpu, n0));
    const float dotv = f_min(0.0f, v_dot(tmpv, n1));
    const float dotw = f_min(0.0f, v_dot(tmpw, n2));
    tmpu = v_sub(tmpu, v_scale(n0, dotu));
    tmpv = v_sub(tmpv, v_scale(n1, dotv));
    tmpw = v_sub(tmpw, v_scale(n2, dotw));
    return v_add(v_add(v_add(vertex->Position, v_scale(tmpu, bary.x)), v_scale(tmpv, bary.y)), v_scale(tmpw, bary.z));
}

/* ======================================================================== */
/* sampling.glsl                                                            */
/* ======================================================================== */

typedef struct LightSample
{
    v3 Direction;
    float Distance;
    v3 Color;
    float Attenuation;
} LightSample;

/* sampling.glsl:25-56 */
static inline LightSample sampleLight(const PtxLightsUbo *ubo, v3 u, v3 position, float *pdf)
{
    const uint32_t lightCount = ubo->LightCount;
    const uint32_t lightIndex = (uint32_t)(u.x * (float)(lightCount + 1));
    *pdf = pto_div(1.0f, (float)(lightCount + 1));
    LightSample ret;
    v2 uyz = { u.y, u.z };

    if (lightIndex >= lightCount)
    {
        const v2 d2 = sampleUniformDiskConcentric(uyz);
        const v3 diskPoint = v_scale(V3(d2.x, d2.y, 0.0f), 0.001f);
        const v3 direction =
            v_normalize(V3(ubo->Directional.Direction[0], ubo->Directional.Direction[1], ubo->Directional.Direction[2]));
        ret.Direction = v_normalize(v_add(direction, m3_mul(computeTangentSpace(direction), diskPoint)));
        ret.Color = V3(ubo->Directional.Color[0], ubo->Directional.Color[1], ubo->Directional.Color[2]);
        ret.Distance = 100000.0f; /* sampling.glsl:3 */
        ret.Attenuation = 1.0f;
        return ret;
    }

    const PtxPointLight *light = &ubo->Lights[lightIndex];
    const v3 lpos = V3(light->Position[0], light->Position[1], light->Position[2]);
    const v2 d2 = sampleUniformDiskConcentric(uyz);
    const v3 diskPoint = v_scale(V3(d2.x, d2.y, 0.0f), 0.1f);
    const v3 direction = v_normalize(v_sub(position, lpos));
    const v3 newPosition = v_add(lpos, m3_mul(computeTangentSpace(direction), diskPoint));

    ret.Distance = v_length(v_sub(position, newPosition));
    ret.Direction = v_normalize(v_sub(position, newPosition));
    ret.Color = V3(light->Color[0], light->Color[1], light->Color[2]);
    const float attenuation = pto_div(1.0f, light->AttenuationConstant + ret.Distance * light->AttenuationLinear +
                                            ret.Distance * ret.Distance * light->AttenuationQuadratic);
    ret.Attenuation = f_clamp(attenuation, 0.0f, 1.0f);
    return ret;
}

/* ======================================================================== */
/* tracing.glsl -- ray differentials and texture footprint                  */
/* (feed only textureGrad; not carried by the render path while every        */
/* texture is 1x1; restated and tested for row N1)       */
/* ======================================================================== */

/* tracing.glsl:2-28 */
static inline void computeDpnDuv(const v3 p[3], const v3 n[3], const v2 uv[3], v3 vtxTangent, v3 vtxBitangent, v3 *dpdu,
                                 v3 *dpdv, v3 *dndu, v3 *dndv)
{
    const v3 e1 = v_sub(p[1], p[0]), e2 = v_sub(p[2], p[0]);
    const v3 en1 = v_sub(n[1], n[0]), en2 = v_sub(n[2], n[0]);
    const v2 duv1 = { uv[1].x - uv[0].x, uv[1].y - uv[0].y }, duv2 = { uv[2].x - uv[0].x, uv[2].y - uv[0].y };
    const float det = duv1.x * duv2.y - duv2.x * duv1.y;
    if (fabsf(det) < 1e-8f)
    {
        *dpdu = vtxTangent;
        *dpdv = vtxBitangent;
        *dndu = v3s(0.0f);
        *dndv = v3s(0.0f);
    }
    else
    {
        const float invDet = pto_div(1.0f, det);
        *dpdu = v_scale(v_sub(v_scale(e1, duv2.y), v_scale(e2, duv1.y)), invDet);
        *dpdv = v_scale(v_add(v_scale(e1, -duv2.x), v_scale(e2, duv1.x)), invDet);
        *dndu = v_scale(v_sub(v_scale(en1, duv2.y), v_scale(en2, duv1.y)), invDet);
        *dndv = v_scale(v_add(v_scale(en1, -duv2.x), v_scale(en2, duv1.x)), invDet);
    }
}

/* tracing.glsl:31-41 */
static inline void computeDpDxy(v3 p, v3 origin, v3 direction, v3 rxOrigin, v3 rxDirection, v3 ryOrigin, v3 ryDirection, v3 n,
                                v3 *dpdx, v3 *dpdy)
{
    (void)origin;
    (void)direction;
    const float d = -v_dot(n, p);
    const float tx = pto_div(-v_dot(n, rxOrigin) - d, v_dot(n, rxDirection));
    const v3 px = v_add(rxOrigin, v_scale(rxDirection, tx));
    const float ty = pto_div(-v_dot(n, ryOrigin) - d, v_dot(n, ryDirection));
    const v3 py = v_add(ryOrigin, v_scale(ryDirection, ty));
    *dpdx = v_sub(px, p);
    *dpdy = v_sub(py, p);
}

/* tracing.glsl:44-50 -- the GLSL calls fma() explicitly: a real fused op on both sides */
static inline float differenceOfProducts(float a, float b, float c, float d)
{
    const float cd = c * d;
    const float dop = fmaf(a, b, -cd);
    const float error = fmaf(-c, d, cd);
    return dop + error;
}

static inline float clampInf(float x) { return isinf(x) ? 0.0f : f_clamp(x, -1e8f, 1e8f); }

/* tracing.glsl:53-78 */
static inline v4 computeDerivatives(v3 dpdx, v3 dpdy, v3 dpdu, v3 dpdv)
{
    const float ata00 = v_dot(dpdu, dpdu);
    const float ata01 = v_dot(dpdu, dpdv);
    const float ata11 = v_dot(dpdv, dpdv);
    float invDet = pto_div(1, differenceOfProducts(ata00, ata11, ata01, ata01));
    invDet = isinf(invDet) ? 0.0f : invDet;
    const float atb0x = v_dot(dpdu, dpdx);
    const float atb1x = v_dot(dpdv, dpdx);
    const float atb0y = v_dot(dpdu, dpdy);
    const float atb1y = v_dot(dpdv, dpdy);
    v4 r;
    r.x = clampInf(differenceOfProducts(ata11, atb0x, ata01, atb1x) * invDet);
    r.y = clampInf(differenceOfProducts(ata00, atb1x, ata01, atb0x) * invDet);
    r.z = clampInf(differenceOfProducts(ata11, atb0y, ata01, atb1y) * invDet);
    r.w = clampInf(differenceOfProducts(ata00, atb1y, ata01, atb0y) * invDet);
    return r;
}

typedef v2 texcoord2; /* closestHit names its vertices v0..v2, which hides the type there */
typedef struct DiffRays
{
    v3 rxOrigin, rxDirection, ryOrigin, ryDirection;
} DiffRays;

/* tracing.glsl:81-108 */
static inline void computeReflectedDifferentialRays(v4 derivatives, v3 n, v3 p, v3 viewDir, v3 reflectedDir, v3 dndu, v3 dndv,
                                                    DiffRays *r)
{
    const float dudx = derivatives.x, dvdx = derivatives.y, dudy = derivatives.z, dvdy = derivatives.w;
    const v3 dndx = v_add(v_scale(dndu, dudx), v_scale(dndv, dvdx));
    const v3 dndy = v_add(v_scale(dndu, dudy), v_scale(dndv, dvdy));
    const float d = -v_dot(n, p);
    const float tx = pto_div(-v_dot(n, r->rxOrigin) - d, v_dot(n, r->rxDirection));
    const v3 px = v_add(r->rxOrigin, v_scale(r->rxDirection, tx));
    const float ty = pto_div(-v_dot(n, r->ryOrigin) - d, v_dot(n, r->ryDirection));
    const v3 py = v_add(r->ryOrigin, v_scale(r->ryDirection, ty));
    const v3 dwodx = v_sub(v_neg(r->rxDirection), viewDir);
    const v3 dwody = v_sub(v_neg(r->ryDirection), viewDir);
    r->rxOrigin = px;
    r->ryOrigin = py;
    const float dwoDotn_dx = v_dot(dwodx, n) + v_dot(viewDir, dndx);
    const float dwoDotn_dy = v_dot(dwody, n) + v_dot(viewDir, dndy);
    const float vn = v_dot(viewDir, n);
    r->rxDirection = v_normalize(v_add(v_sub(reflectedDir, dwodx), v_scale(v_add(v_scale(dndx, vn), v_scale(n, dwoDotn_dx)), 2)));
    r->ryDirection = v_normalize(v_add(v_sub(reflectedDir, dwody), v_scale(v_add(v_scale(dndy, vn), v_scale(n, dwoDotn_dy)), 2)));
}

/* tracing.glsl:111-148 */
static inline void computeRefractedDifferentialRays(v4 derivatives, v3 n, v3 p, v3 viewDir, v3 refractedDir, v3 dndu, v3 dndv,
                                                    float eta, DiffRays *r)
{
    const float dudx = derivatives.x, dvdx = derivatives.y, dudy = derivatives.z, dvdy = derivatives.w;
    v3 dndx = v_add(v_scale(dndu, dudx), v_scale(dndv, dvdx));
    v3 dndy = v_add(v_scale(dndu, dudy), v_scale(dndv, dvdy));
    const float d = -v_dot(n, p);
    const float tx = pto_div(-v_dot(n, r->rxOrigin) - d, v_dot(n, r->rxDirection));
    const v3 px = v_add(r->rxOrigin, v_scale(r->rxDirection, tx));
    const float ty = pto_div(-v_dot(n, r->ryOrigin) - d, v_dot(n, r->ryDirection));
    const v3 py = v_add(r->ryOrigin, v_scale(r->ryDirection, ty));
    const v3 dwodx = v_sub(v_neg(r->rxDirection), viewDir);
    const v3 dwody = v_sub(v_neg(r->ryDirection), viewDir);
    r->rxOrigin = px;
    r->ryOrigin = py;
    if (v_dot(viewDir, n) < 0.0f)
    {
        n = v_neg(n);
        dndx = v_neg(dndx);
        dndy = v_neg(dndy);
    }
    const float dwoDotn_dx = v_dot(dwodx, n) + v_dot(viewDir, dndx);
    const float dwoDotn_dy = v_dot(dwody, n) + v_dot(viewDir, dndy);
    const float mu = pto_div(v_dot(viewDir, n), eta) - fabsf(v_dot(refractedDir, n));
    const float dmudx = dwoDotn_dx * (pto_div(1.0f, eta) + pto_div(pto_div(1.0f, eta * eta) * v_dot(viewDir, n), v_dot(refractedDir, n)));
    const float dmudy = dwoDotn_dy * (pto_div(1.0f, eta) + pto_div(pto_div(1.0f, eta * eta) * v_dot(viewDir, n), v_dot(refractedDir, n)));
    r->rxDirection = v_normalize(v_add(v_sub(refractedDir, v_scale(dwodx, eta)), v_add(v_scale(dndx, mu), v_scale(n, dmudx))));
    r->ryDirection = v_normalize(v_add(v_sub(refractedDir, v_scale(dwody, eta)), v_add(v_scale(dndy, mu), v_scale(n, dmudy))));
}

/* tracing.glsl:151-161 -- log2 through the fixed kernel */
static inline float computeLod(v4 derivatives)
{
    const float sx = sqrtf(derivatives.x * derivatives.x + derivatives.y * derivatives.y);
    const float sy = sqrtf(derivatives.z * derivatives.z + derivatives.w * derivatives.w);
    const float smax = f_max(sx, sy);
    return smax == 0.0f ? 0.0f : (float)pto_log2((double)smax);
}

/* ======================================================================== */
/* Scene: flattened world-space triangles (stand-in for BLAS/TLAS)          */
/* ======================================================================== */

typedef struct Pair /* one (instance, mesh) */
{
    float M[12];   /* A_instance * A_mesh, 3 rows x 4 (sampling.glsl:5-15 derivation) */
    m3 Rinv;       /* inverse of the linear part, for the normal transform             */
    const PtxVertex *vb; /* first vertex of the mesh: the scene's vertices, or this pair's skinned copy */
    const uint32_t *ib;  /* first index of the mesh */
    uint32_t materialId, firstTri;
    uint32_t nonOpaque; /* geometry without VK_GEOMETRY_OPAQUE_BIT: any-hit shaders run (AccelerationStructure.cpp:94-97) */
} Pair;

typedef struct BvhNode
{
    float lo[3], hi[3];
    uint32_t left;  /* internal: left child index, right = left + 1; leaf: first tri slot */
    uint32_t count; /* 0 = internal                                                       */
} BvhNode;

struct PtoScene
{
    PtxSceneDesc d; /* deep copies */
    Pair *pairs;
    uint32_t pairCount;
    uint64_t triCount;
    float *v0, *e1, *e2; /* 3 floats per triangle, world space */
    uint32_t *triPair, *triPrim;
    BvhNode *nodes;
    uint32_t nodeCount;
    uint32_t *bvhTris; /* triangle ids in leaf order */
    /* scene textures (row N1) */
    struct OTexture *textures;
    uint32_t textureCount;
    PtxVertex *skinned; /* skinning.comp output: one copy per instanced animated mesh, in pair order */
    uint32_t skyKind; /* PTX_SKYBOX_*; its 1 / 6 images sit at textures[textureCount ...] */
    uint32_t *texels8; /* RGBA8 pool, all levels of all 8-bit textures */
    float *texelsF;    /* RGBA32F pool */
    float srgbLut[256];
};

typedef struct OTexture
{
    uint32_t width, height, levels, format;
    size_t levelOffset[16]; /* in texels, into the pool of its format */
} OTexture;

static void *dupmem(const void *p, size_t n)
{
    if (!n)
        return NULL;
    void *q = malloc(n);
    memcpy(q, p, n);
    return q;
}

/* world = A_instance * A_mesh * x (sampling.glsl:7) */
static void composeTransform(const float *Ai, const float *Am, float *M)
{
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++)
            M[r * 4 + c] = (Ai[r * 4 + 0] * Am[0 * 4 + c] + Ai[r * 4 + 1] * Am[1 * 4 + c]) + Ai[r * 4 + 2] * Am[2 * 4 + c];
        M[r * 4 + 3] =
            ((Ai[r * 4 + 0] * Am[0 * 4 + 3] + Ai[r * 4 + 1] * Am[1 * 4 + 3]) + Ai[r * 4 + 2] * Am[2 * 4 + 3]) + Ai[r * 4 + 3];
    }
}

static inline v3 xformPoint(const float *M, v3 p)
{
    return V3(((p.x * M[0] + p.y * M[1]) + p.z * M[2]) + M[3], ((p.x * M[4] + p.y * M[5]) + p.z * M[6]) + M[7],
              ((p.x * M[8] + p.y * M[9]) + p.z * M[10]) + M[11]);
}
static inline v3 xformVector(const float *M, v3 p)
{
    return V3((p.x * M[0] + p.y * M[1]) + p.z * M[2], (p.x * M[4] + p.y * M[5]) + p.z * M[6],
              (p.x * M[8] + p.y * M[9]) + p.z * M[10]);
}

static void buildBvh(PtoScene *s);
static void uploadTextures(PtoScene *s, const PtxSceneDesc *desc);

/* skinning.comp:21-50 for one vertex.  bones = mat3x4[]: vec4 * mat3x4 takes the dot product with each of the three
 * stored rows, i.e. bones[b] is the affine matrix of the bone in 3 rows x 4.  The normal goes through the inverse
 * transpose of the linear part ("transpose(inverse(mat4(transform)))", fixed here as the cofactor inverse). */
static PtxVertex skinVertex(const PtxAnimatedVertex *a, const PtxTransform *bones, uint32_t boneCount)
{
    v3 P = v3s(0.0f), N = v3s(0.0f), T = v3s(0.0f), B = v3s(0.0f);
    float totalWeight = 0;
    for (int i = 0; i < 4 && totalWeight < 1.0f; i++)
    {
        const uint32_t boneIndex = a->BoneIndices[i];
        const float w = a->BoneWeights[i];
        PtxTransform idt = { { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 } };
        const float *M = (bones && boneIndex < boneCount) ? bones[boneIndex].m : idt.m;
        const v3 p = xformPoint(M, V3(a->Position[0], a->Position[1], a->Position[2]));
        P = v_add(P, v_scale(p, w));
        T = v_add(T, v_scale(v_normalize(xformVector(M, V3(a->Tangent[0], a->Tangent[1], a->Tangent[2]))), w));
        B = v_add(B, v_scale(v_normalize(xformVector(M, V3(a->Bitangent[0], a->Bitangent[1], a->Bitangent[2]))), w));
        m3 R;
        R.c0 = V3(M[0], M[4], M[8]);
        R.c1 = V3(M[1], M[5], M[9]);
        R.c2 = V3(M[2], M[6], M[10]);
        const m3 Ri = m3_inverse(R);
        const v3 n = V3(a->Normal[0], a->Normal[1], a->Normal[2]);
        N = v_add(N, v_scale(v_normalize(V3(v_dot(n, Ri.c0), v_dot(n, Ri.c1), v_dot(n, Ri.c2))), w));
        totalWeight += w;
    }
    PtxVertex o;
    memset(&o, 0, sizeof(o));
    o.Position[0] = P.x; o.Position[1] = P.y; o.Position[2] = P.z;
    o.TexCoords[0] = a->TexCoords[0]; o.TexCoords[1] = a->TexCoords[1];
    o.Normal[0] = N.x; o.Normal[1] = N.y; o.Normal[2] = N.z;
    o.Tangent[0] = T.x; o.Tangent[1] = T.y; o.Tangent[2] = T.z;
    o.Bitangent[0] = B.x; o.Bitangent[1] = B.y; o.Bitangent[2] = B.z;
    return o;
}

PtoScene *pto_scene_create(const PtxSceneDesc *desc, int wantBvh)
{
    return pto_scene_create_posed(desc, NULL, NULL, 0, wantBvh);
}

/* The scene as Renderer::Render sees it after Scene::Update: instance transforms replaced (NULL = those of the
 * desc), animated meshes skinned with `bones` (NULL = bind pose, Renderer.cpp:296-303). */
PtoScene *pto_scene_create_posed(const PtxSceneDesc *desc, const PtxTransform *instanceTransforms, const PtxTransform *bones,
                                 uint32_t boneCount, int wantBvh)
{
    PtoScene *s = (PtoScene *)calloc(1, sizeof(PtoScene));
    s->d = *desc;
    s->d.vertices = (const PtxVertex *)dupmem(desc->vertices, desc->vertexCount * sizeof(PtxVertex));
    s->d.indices = (const uint32_t *)dupmem(desc->indices, desc->indexCount * 4);
    s->d.animatedVertices = NULL;
    s->d.animatedIndices = (const uint32_t *)dupmem(desc->animatedIndices, desc->animatedIndexCount * 4);
    s->d.transforms = (const PtxTransform *)dupmem(desc->transforms, desc->transformCount * sizeof(PtxTransform));
    s->d.geometries = (const PtxGeometry *)dupmem(desc->geometries, desc->geometryCount * sizeof(PtxGeometry));
    s->d.metallicRoughnessMaterials = (const PtxMetallicRoughnessMaterial *)dupmem(
        desc->metallicRoughnessMaterials, desc->metallicRoughnessMaterialCount * sizeof(PtxMetallicRoughnessMaterial));
    s->d.specularGlossinessMaterials = (const PtxSpecularGlossinessMaterial *)dupmem(
        desc->specularGlossinessMaterials, desc->specularGlossinessMaterialCount * sizeof(PtxSpecularGlossinessMaterial));
    s->d.phongMaterials =
        (const PtxPhongMaterial *)dupmem(desc->phongMaterials, desc->phongMaterialCount * sizeof(PtxPhongMaterial));
    s->d.meshes = (const PtxMeshRecord *)dupmem(desc->meshes, desc->meshCount * sizeof(PtxMeshRecord));
    s->d.models = (const PtxModel *)dupmem(desc->models, desc->modelCount * sizeof(PtxModel));
    s->d.instances = (const PtxModelInstance *)dupmem(desc->instances, desc->instanceCount * sizeof(PtxModelInstance));

    /* pairs in instance-then-mesh order; global triangle id = running primitive count */
    uint32_t pairCount = 0;
    uint64_t triCount = 0;
    for (uint32_t i = 0; i < desc->instanceCount; i++)
    {
        const PtxModel *m = &desc->models[desc->instances[i].ModelIndex];
        pairCount += m->MeshCount;
        for (uint32_t k = 0; k < m->MeshCount; k++)
            triCount += desc->geometries[desc->meshes[m->MeshOffset + k].GeometryIndex].IndexLength / 3;
    }
    s->pairCount = pairCount;
    s->triCount = triCount;
    s->pairs = (Pair *)calloc(pairCount ? pairCount : 1, sizeof(Pair));
    size_t skinnedCount = 0; /* one output range per instanced animated mesh, in pair order */
    for (uint32_t i = 0; i < desc->instanceCount; i++)
    {
        const PtxModel *m = &desc->models[desc->instances[i].ModelIndex];
        for (uint32_t k = 0; k < m->MeshCount; k++)
        {
            const PtxGeometry *g = &desc->geometries[desc->meshes[m->MeshOffset + k].GeometryIndex];
            if (g->IsAnimated)
                skinnedCount += g->VertexLength;
        }
    }
    s->skinned = (PtxVertex *)calloc(skinnedCount ? skinnedCount : 1, sizeof(PtxVertex));
    size_t skinnedCursor = 0;
    s->v0 = (float *)malloc((triCount ? triCount : 1) * 12);
    s->e1 = (float *)malloc((triCount ? triCount : 1) * 12);
    s->e2 = (float *)malloc((triCount ? triCount : 1) * 12);
    s->triPair = (uint32_t *)malloc((triCount ? triCount : 1) * 4);
    s->triPrim = (uint32_t *)malloc((triCount ? triCount : 1) * 4);

    uint32_t p = 0;
    uint64_t t = 0;
    for (uint32_t i = 0; i < desc->instanceCount; i++)
    {
        const PtxModelInstance *inst = &desc->instances[i];
        const PtxModel *m = &desc->models[inst->ModelIndex];
        for (uint32_t k = 0; k < m->MeshCount; k++, p++)
        {
            const PtxMeshRecord *rec = &desc->meshes[m->MeshOffset + k];
            const PtxGeometry *g = &desc->geometries[rec->GeometryIndex];
            Pair *pr = &s->pairs[p];
            composeTransform(instanceTransforms ? instanceTransforms[i].m : inst->Transform.m, desc->transforms[rec->TransformIndex].m, pr->M);
            m3 R;
            R.c0 = V3(pr->M[0], pr->M[4], pr->M[8]);
            R.c1 = V3(pr->M[1], pr->M[5], pr->M[9]);
            R.c2 = V3(pr->M[2], pr->M[6], pr->M[10]);
            pr->Rinv = m3_inverse(R);
            if (g->IsAnimated)
            {
                PtxVertex *out = &s->skinned[skinnedCursor];
                for (uint32_t v = 0; v < g->VertexLength; v++)
                {
                    const PtxAnimatedVertex *a = &desc->animatedVertices[g->VertexOffset + v];
                    if (bones)
                        out[v] = skinVertex(a, bones, boneCount);
                    else /* bind pose: the attributes as authored (OutBindPoseAnimatedVertices, Renderer.cpp:296-303) */
                    {
                        memcpy(out[v].Position, a->Position, 12); memcpy(out[v].TexCoords, a->TexCoords, 8);
                        memcpy(out[v].Normal, a->Normal, 12); memcpy(out[v].Tangent, a->Tangent, 12);
                        memcpy(out[v].Bitangent, a->Bitangent, 12);
                    }
                }
                skinnedCursor += g->VertexLength;
                pr->vb = out;
                pr->ib = &s->d.animatedIndices[g->IndexOffset];
            }
            else
            {
                pr->vb = &s->d.vertices[g->VertexOffset];
                pr->ib = &s->d.indices[g->IndexOffset];
            }
            pr->materialId = rec->MaterialId;
            pr->firstTri = (uint32_t)t;
            pr->nonOpaque = g->IsOpaque ? 0u : 1u;
            const uint32_t nprim = g->IndexLength / 3;
            for (uint32_t q = 0; q < nprim; q++, t++)
            {
                v3 w[3];
                for (int c = 0; c < 3; c++)
                {
                    const PtxVertex *vx = &pr->vb[pr->ib[q * 3 + c]];
                    w[c] = xformPoint(pr->M, V3(vx->Position[0], vx->Position[1], vx->Position[2]));
                }
                v3 a = v_sub(w[1], w[0]), b = v_sub(w[2], w[0]);
                /* A zero-area triangle (repeated or collinear vertices with an exactly vanishing edge cross product) is
                 * never hit -- as in Vulkan, where degenerate triangles generate no intersections.  Without this,
                 * det = e1 . (d x e2) is a rounding residue instead of 0 and the test "hits" it at a meaningless t
                 * (atrium_like triangle 3608410 with e1 == e2: t = 16, u = v = -0 for a ray passing its vertex at
                 * t = 29.65).  Zero edges make det exactly 0. */
                {
                    const v3 n = v_cross(a, b);
                    if (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f)
                        a = b = v3s(0.0f);
                }
                s->v0[t * 3 + 0] = w[0].x; s->v0[t * 3 + 1] = w[0].y; s->v0[t * 3 + 2] = w[0].z;
                s->e1[t * 3 + 0] = a.x; s->e1[t * 3 + 1] = a.y; s->e1[t * 3 + 2] = a.z;
                s->e2[t * 3 + 0] = b.x; s->e2[t * 3 + 1] = b.y; s->e2[t * 3 + 2] = b.z;
                s->triPair[t] = p;
                s->triPrim[t] = q;
            }
        }
    }
    uploadTextures(s, desc);
    s->d.textures = NULL; /* the caller's arrays are not retained */
    s->d.skybox = NULL;
    if (wantBvh && triCount)
        buildBvh(s);
    return s;
}

void pto_scene_destroy(PtoScene *s)
{
    if (!s)
        return;
    free((void *)s->d.vertices); free((void *)s->d.indices); free((void *)s->d.transforms);
    free((void *)s->d.animatedIndices); free(s->skinned);
    free((void *)s->d.geometries); free((void *)s->d.metallicRoughnessMaterials);
    free((void *)s->d.specularGlossinessMaterials); free((void *)s->d.phongMaterials);
    free((void *)s->d.meshes); free((void *)s->d.models); free((void *)s->d.instances);
    free(s->pairs); free(s->v0); free(s->e1); free(s->e2); free(s->triPair); free(s->triPrim);
    free(s->nodes); free(s->bvhTris);
    free(s->textures); free(s->texels8); free(s->texelsF);
    free(s);
}

uint64_t pto_scene_triangle_count(const PtoScene *s) { return s->triCount; }

/* ======================================================================== */
/* Ray / triangle (Moeller-Trumbore on v0,e1,e2) and the two queries        */
/* ======================================================================== */

/* The driver's ray-triangle test is unspecified; this is the fixed stand-in used by
 * both the oracle and the HIP kernels.  Hit iff det != 0, 0<=u<=1, v>=0, u+v<=1 and
 * tmin < t < tmax.  Returns barycentrics in Vulkan's hitAttribute convention. */
static inline int intersectTri(const float *v0, const float *e1, const float *e2, v3 o, v3 d, float tmin, float tmax,
                               float *t, float *u, float *v)
{
    const v3 E1 = V3(e1[0], e1[1], e1[2]), E2 = V3(e2[0], e2[1], e2[2]), V0 = V3(v0[0], v0[1], v0[2]);
    const v3 pvec = v_cross(d, E2);
    const float det = v_dot(E1, pvec);
    if (!(det != 0.0f))
        return 0;
    const float inv = pto_div(1.0f, det);
    /* Two passes (see pt_device.hpp): plain Moeller-Trumbore from the ray origin loses (|o - v0| / size)^2 ulps in the
     * barycentrics but its t is good; pass 1 solves from o with loose bounds on (u, v), pass 2 again from o + t1 d. */
    v3 tvec = v_sub(o, V0);
    float uu = v_dot(tvec, pvec) * inv;
    if (!(uu >= -0.25f && uu <= 1.25f))
        return 0;
    v3 qvec = v_cross(tvec, E1);
    float vv = v_dot(d, qvec) * inv;
    if (!(vv >= -0.25f && uu + vv <= 1.25f))
        return 0;
    const float t1 = v_dot(E2, qvec) * inv;
    if (!(t1 > -1e30f && t1 < 1e30f))
        return 0;
    /* pass 2 from o + t1 d: the offset to v0 is at most the triangle's size, the correction to t is tiny */
    tvec = v_sub(v_add(o, v_scale(d, t1)), V0);
    uu = v_dot(tvec, pvec) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f))
        return 0;
    qvec = v_cross(tvec, E1);
    vv = v_dot(d, qvec) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f))
        return 0;
    const float tt = t1 + v_dot(E2, qvec) * inv;
    if (!(tt > tmin && tt < tmax))
        return 0;
    *t = tt;
    *u = uu;
    *v = vv;
    return 1;
}

/* What anyhit.rahit leaves in the payload: the nearest ignored (alpha < 0.5) hit.  The any-hit
 * invocation order is the driver's; the nearest one (ties: smaller triangle id) is the only
 * order-independent reading of anyhit.rahit:54-61 that closestHit.rchit:105-106 can observe. */
typedef struct Decal
{
    float dist; /* -1 = none (payload.DirectLightPdf) */
    v3 color;   /* payload.LightDirection             */
    float alpha; /* payload.LightDistance             */
    uint32_t tri;
} Decal;

static v4 hitBaseColor(const PtoScene *s, uint32_t tri, float u, float v);

/* anyhit.rahit:36-64 for one candidate of a non-opaque geometry: 1 = ignoreIntersectionEXT (alpha < 0.5), having
 * remembered the candidate as the decal if it is nearer than the one the payload holds */
static inline int anyHitIgnores(const PtoScene *s, uint32_t tri, float t, float u, float v, Decal *decal)
{
    const v4 color = hitBaseColor(s, tri, u, v);
    if (!(color.w < 0.5f))
        return 0;
    if (decal && (decal->dist == -1.0f || t < decal->dist || (t == decal->dist && tri < decal->tri)))
    {
        decal->dist = t;
        decal->color = V3(color.x, color.y, color.z);
        decal->alpha = color.w;
        decal->tri = tri;
    }
    return 1;
}

/* closest hit: min t, ties broken by the smaller global triangle id */
static inline void considerTri(const PtoScene *s, uint32_t tri, v3 o, v3 d, float tmin, float tmax, PtoHit *best, Decal *decal)
{
    float t, u, v;
    const float lim = best->tri == 0xffffffffu ? tmax : best->t;
    /* accept t == best.t only for a smaller id: test against nextafter(lim) via <= below */
    if (!intersectTri(&s->v0[tri * 3], &s->e1[tri * 3], &s->e2[tri * 3], o, d, tmin, tmax, &t, &u, &v))
        return;
    if (s->pairs[s->triPair[tri]].nonOpaque && anyHitIgnores(s, tri, t, u, v, decal))
        return; /* ignoreIntersectionEXT */
    if (best->tri == 0xffffffffu || t < lim || (t == lim && tri < best->tri))
    {
        best->t = t;
        best->u = u;
        best->v = v;
        best->tri = tri;
    }
}

/* ---- binned-SAH BVH (CPU baseline accelerator) ---- */

typedef struct BuildCtx
{
    PtoScene *s;
    float *blo, *bhi, *cen; /* per-triangle bounds and centroid */
    uint32_t *ids;
    uint32_t nodeCap;
} BuildCtx;

static void triBounds(const PtoScene *s, uint32_t t, float *lo, float *hi)
{
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++)
    {
        const float p0 = s->v0[t * 3 + a], p1 = p0 + s->e1[t * 3 + a], p2 = p0 + s->e2[t * 3 + a];
        mn[a] = f_min(p0, f_min(p1, p2));
        mx[a] = f_max(p0, f_max(p1, p2));
    }
    for (int a = 0; a < 3; a++)
    {
        /* Padding so that the slab test does not reject a ray the triangle test accepts: 1e-5 of the coordinates + 5e-4
         * of the triangle's extent per axis (the two-pass triangle test is good to ~1e-4 of the size; the error moves the
         * point within the triangle's plane).  Same rule as on the HIP side. */
        const float pad = 1e-5f * f_max(fabsf(mn[a]), fabsf(mx[a])) + 5e-4f * (mx[a] - mn[a]) + 1e-7f;
        lo[a] = mn[a] - pad;
        hi[a] = mx[a] + pad;
    }
}

static float areaOf(const float *lo, const float *hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

static void buildRec(BuildCtx *c, uint32_t node, uint32_t first, uint32_t count)
{
    PtoScene *s = c->s;
    BvhNode *n = &s->nodes[node];
    float clo[3] = { 1e30f, 1e30f, 1e30f }, chi[3] = { -1e30f, -1e30f, -1e30f };
    for (int a = 0; a < 3; a++)
    {
        n->lo[a] = 1e30f;
        n->hi[a] = -1e30f;
    }
    for (uint32_t i = first; i < first + count; i++)
    {
        const uint32_t t = c->ids[i];
        for (int a = 0; a < 3; a++)
        {
            n->lo[a] = f_min(n->lo[a], c->blo[t * 3 + a]);
            n->hi[a] = f_max(n->hi[a], c->bhi[t * 3 + a]);
            clo[a] = f_min(clo[a], c->cen[t * 3 + a]);
            chi[a] = f_max(chi[a], c->cen[t * 3 + a]);
        }
    }
    if (count <= 4)
    {
        n->left = first;
        n->count = count;
        return;
    }
    enum { NB = 16 };
    int bestAxis = -1, bestSplit = 0;
    float bestCost = 1e30f;
    for (int a = 0; a < 3; a++)
    {
        const float ext = chi[a] - clo[a];
        if (!(ext > 0.0f))
            continue;
        uint32_t cnt[NB] = { 0 };
        float lo[NB][3], hi[NB][3];
        for (int b = 0; b < NB; b++)
            for (int k = 0; k < 3; k++)
            {
                lo[b][k] = 1e30f;
                hi[b][k] = -1e30f;
            }
        const float sc = (float)NB / ext;
        for (uint32_t i = first; i < first + count; i++)
        {
            const uint32_t t = c->ids[i];
            int b = (int)((c->cen[t * 3 + a] - clo[a]) * sc);
            if (b >= NB) b = NB - 1;
            if (b < 0) b = 0;
            cnt[b]++;
            for (int k = 0; k < 3; k++)
            {
                lo[b][k] = f_min(lo[b][k], c->blo[t * 3 + k]);
                hi[b][k] = f_max(hi[b][k], c->bhi[t * 3 + k]);
            }
        }
        float rightArea[NB];
        uint32_t rightCnt[NB];
        float rl[3] = { 1e30f, 1e30f, 1e30f }, rh[3] = { -1e30f, -1e30f, -1e30f };
        uint32_t rc = 0;
        for (int b = NB - 1; b > 0; b--)
        {
            for (int k = 0; k < 3; k++)
            {
                rl[k] = f_min(rl[k], lo[b][k]);
                rh[k] = f_max(rh[k], hi[b][k]);
            }
            rc += cnt[b];
            rightArea[b] = rc ? areaOf(rl, rh) : 0.0f;
            rightCnt[b] = rc;
        }
        float ll[3] = { 1e30f, 1e30f, 1e30f }, lh[3] = { -1e30f, -1e30f, -1e30f };
        uint32_t lc = 0;
        for (int b = 0; b < NB - 1; b++)
        {
            for (int k = 0; k < 3; k++)
            {
                ll[k] = f_min(ll[k], lo[b][k]);
                lh[k] = f_max(lh[k], hi[b][k]);
            }
            lc += cnt[b];
            if (!lc || !rightCnt[b + 1])
                continue;
            const float cost = areaOf(ll, lh) * (float)lc + rightArea[b + 1] * (float)rightCnt[b + 1];
            if (cost < bestCost)
            {
                bestCost = cost;
                bestAxis = a;
                bestSplit = b;
            }
        }
    }
    uint32_t mid;
    if (bestAxis < 0)
        mid = first + count / 2; /* all centroids coincide: split in the middle */
    else
    {
        const float ext = chi[bestAxis] - clo[bestAxis];
        const float sc = (float)NB / ext;
        uint32_t i = first, j = first + count;
        while (i < j)
        {
            const uint32_t t = c->ids[i];
            int b = (int)((c->cen[t * 3 + bestAxis] - clo[bestAxis]) * sc);
            if (b >= NB) b = NB - 1;
            if (b < 0) b = 0;
            if (b <= bestSplit)
                i++;
            else
            {
                j--;
                c->ids[i] = c->ids[j];
                c->ids[j] = t;
            }
        }
        mid = i;
        if (mid == first || mid == first + count)
            mid = first + count / 2;
    }
    const uint32_t l = s->nodeCount;
    s->nodeCount += 2;
    n = &s->nodes[node]; /* (nodes is preallocated; pointer stays valid) */
    n->left = l;
    n->count = 0;
    buildRec(c, l, first, mid - first);
    buildRec(c, l + 1, mid, first + count - mid);
}

static void buildBvh(PtoScene *s)
{
    const uint32_t n = (uint32_t)s->triCount;
    BuildCtx c;
    c.s = s;
    c.blo = (float *)malloc((size_t)n * 12);
    c.bhi = (float *)malloc((size_t)n * 12);
    c.cen = (float *)malloc((size_t)n * 12);
    c.ids = (uint32_t *)malloc((size_t)n * 4);
    for (uint32_t t = 0; t < n; t++)
    {
        triBounds(s, t, &c.blo[t * 3], &c.bhi[t * 3]);
        for (int a = 0; a < 3; a++)
            c.cen[t * 3 + a] = 0.5f * (c.blo[t * 3 + a] + c.bhi[t * 3 + a]);
        c.ids[t] = t;
    }
    s->nodes = (BvhNode *)malloc((size_t)(2 * n + 2) * sizeof(BvhNode));
    s->nodeCount = 1;
    buildRec(&c, 0, 0, n);
    s->bvhTris = c.ids;
    free(c.blo);
    free(c.bhi);
    free(c.cen);
}

static inline int slab(const BvhNode *n, v3 o, v3 id, float tmin, float tmax, float *tnear)
{
    float t0 = (n->lo[0] - o.x) * id.x, t1 = (n->hi[0] - o.x) * id.x;
    float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
    t0 = (n->lo[1] - o.y) * id.y;
    t1 = (n->hi[1] - o.y) * id.y;
    lo = fmaxf(lo, fminf(t0, t1));
    hi = fminf(hi, fmaxf(t0, t1));
    t0 = (n->lo[2] - o.z) * id.z;
    t1 = (n->hi[2] - o.z) * id.z;
    lo = fmaxf(lo, fminf(t0, t1));
    hi = fminf(hi, fmaxf(t0, t1));
    lo = fmaxf(lo, tmin);
    hi = fminf(hi, tmax);
    *tnear = lo;
    return lo <= hi * 1.0000004f;
}

static PtoHit traceClosest(const PtoScene *s, v3 o, v3 d, float tmin, float tmax, int brute, PtoStats *st, Decal *decal)
{
    PtoHit best;
    if (decal)
        decal->dist = -1.0f;
    best.t = tmax;
    best.u = best.v = 0.0f;
    best.tri = 0xffffffffu;
    if (brute || !s->nodes)
    {
        for (uint32_t t = 0; t < s->triCount; t++)
            considerTri(s, t, o, d, tmin, tmax, &best, decal);
        if (st)
            st->trisTested += s->triCount;
        return best;
    }
    const v3 id = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp)
    {
        const BvhNode *n = &s->nodes[stack[--sp]];
        float tn;
        /* Culling against the current best must leave room for the triangle test's own error in t: Moeller-Trumbore
         * from a far origin is good to ~1e-5 relative, so a second triangle in the same plane can report the SAME t
         * while the point o + t d lies a few 1e-5 outside that triangle's padded box (found on street_like at
         * 1920x1080: two overlapping coplanar triangles, bit-identical t, box entry 3e-5 later than t -- the BVH walk
         * kept the larger id where brute force keeps the smaller).  Equal-t candidates must be visited for the id
         * tie-break anyway; the limit is widened by 1e-4 relative. */
        const float limit = best.tri == 0xffffffffu ? tmax : f_min(best.t * 1.0001f, tmax);
        if (!slab(n, o, id, tmin, limit, &tn))
            continue;
        if (st)
            st->nodesVisited++;
        if (n->count)
        {
            for (uint32_t i = 0; i < n->count; i++)
                considerTri(s, s->bvhTris[n->left + i], o, d, tmin, tmax, &best, decal);
            if (st)
                st->trisTested += n->count;
        }
        else
        {
            stack[sp++] = n->left;
            stack[sp++] = n->left + 1;
        }
    }
    return best;
}

/* debugging aid: which nodes on the way to the leaf of triangle `tri` pass the slab test of this ray? */
PTX_API void pto_debug_path(const PtoScene *s, const float *ray, uint32_t tri)
{
    const v3 o = V3(ray[0], ray[1], ray[2]), d = V3(ray[4], ray[5], ray[6]);
    const v3 id = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float tmin = ray[3], tmax = ray[7];
    /* parent links by one pass */
    int32_t *parent = (int32_t *)malloc((size_t)s->nodeCount * 4);
    int32_t leaf = -1;
    parent[0] = -1;
    for (uint32_t i = 0; i < s->nodeCount; i++)
    {
        const BvhNode *n = &s->nodes[i];
        if (n->count)
        {
            for (uint32_t k = 0; k < n->count; k++)
                if (s->bvhTris[n->left + k] == tri)
                    leaf = (int32_t)i;
        }
        else
        {
            parent[n->left] = (int32_t)i;
            parent[n->left + 1] = (int32_t)i;
        }
    }
    int depth = 0;
    for (int32_t i = leaf; i >= 0; i = parent[i], depth++)
    {
        const BvhNode *n = &s->nodes[i];
        float tn;
        const int ok = slab(n, o, id, tmin, tmax, &tn);
        fprintf(stderr, "[pto] node %d (depth-from-leaf %d, count %u) slab %d tn %.9g lo (%.9g %.9g %.9g) hi (%.9g %.9g %.9g)\n", i, depth, n->count, ok, tn,
                n->lo[0], n->lo[1], n->lo[2], n->hi[0], n->hi[1], n->hi[2]);
    }
    fprintf(stderr, "[pto] depth %d, id (%.9g %.9g %.9g)\n", depth, id.x, id.y, id.z);
    free(parent);
}

/* occlusionAnyhit.rahit:35-53: a shadow ray passes through any hit whose alpha is below 1 */
static inline int occlusionIgnores(const PtoScene *s, uint32_t tri, float u, float v)
{
    if (!s->pairs[s->triPair[tri]].nonOpaque)
        return 0;
    return hitBaseColor(s, tri, u, v).w < 1.0f;
}

static int traceAny(const PtoScene *s, v3 o, v3 d, float tmin, float tmax, int brute, PtoStats *st)
{
    float t, u, v;
    if (brute || !s->nodes)
    {
        for (uint32_t i = 0; i < s->triCount; i++)
            if (intersectTri(&s->v0[i * 3], &s->e1[i * 3], &s->e2[i * 3], o, d, tmin, tmax, &t, &u, &v) && !occlusionIgnores(s, i, u, v))
                return 1;
        return 0;
    }
    const v3 id = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp)
    {
        const BvhNode *n = &s->nodes[stack[--sp]];
        float tn;
        if (!slab(n, o, id, tmin, tmax, &tn))
            continue;
        if (st)
            st->nodesVisited++;
        if (n->count)
        {
            for (uint32_t i = 0; i < n->count; i++)
            {
                const uint32_t tr = s->bvhTris[n->left + i];
                if (intersectTri(&s->v0[tr * 3], &s->e1[tr * 3], &s->e2[tr * 3], o, d, tmin, tmax, &t, &u, &v) && !occlusionIgnores(s, tr, u, v))
                    return 1;
            }
        }
        else
        {
            stack[sp++] = n->left;
            stack[sp++] = n->left + 1;
        }
    }
    return 0;
}

void pto_trace_closest(const PtoScene *s, const float *rays, uint32_t n, PtoHit *hits, int brute)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; i++)
    {
        const float *r = &rays[i * 8];
        hits[i] = traceClosest(s, V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7], brute, NULL, NULL);
    }
}

void pto_trace_any(const PtoScene *s, const float *rays, uint32_t n, uint32_t *occluded, int brute)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; i++)
    {
        const float *r = &rays[i * 8];
        occluded[i] = (uint32_t)traceAny(s, V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7], brute, NULL);
    }
}

/* ======================================================================== */
/* Software sampler: what the Vulkan sampler of Renderer.cpp:103-112 does     */
/* (linear min/mag/mip, repeat addressing) restated with fixed arithmetic.    */
/* Anisotropic filtering (implementation-defined in Vulkan) is NOT modelled:  */
/* textureGrad is isotropic trilinear.                                        */
/* ======================================================================== */

static inline uint32_t levelDim(uint32_t d, uint32_t level) { const uint32_t v = d >> level; return v ? v : 1u; }

/* sRGB EOTF / inverse through the fixed pow kernel (shared definition with the HIP kernels) */

static inline v4 fetchTexel(const PtoScene *s, const OTexture *t, uint32_t level, uint32_t x, uint32_t y)
{
    const size_t idx = t->levelOffset[level] + (size_t)y * levelDim(t->width, level) + x;
    v4 r;
    if (t->format == PTX_TEXTURE_RGBA32F)
    {
        r.x = s->texelsF[idx * 4]; r.y = s->texelsF[idx * 4 + 1]; r.z = s->texelsF[idx * 4 + 2]; r.w = s->texelsF[idx * 4 + 3];
        return r;
    }
    const uint32_t p = s->texels8[idx];
    if (t->format == PTX_TEXTURE_RGBA8_SRGB)
    {
        r.x = s->srgbLut[p & 255u]; r.y = s->srgbLut[(p >> 8) & 255u]; r.z = s->srgbLut[(p >> 16) & 255u];
    }
    else
    {
        r.x = (float)(p & 255u) / 255.0f; r.y = (float)((p >> 8) & 255u) / 255.0f; r.z = (float)((p >> 16) & 255u) / 255.0f;
    }
    r.w = (float)(p >> 24) / 255.0f;
    return r;
}

static inline v4 v4_lerp(v4 a, v4 b, float t)
{
    v4 r = { a.x * (1.0f - t) + b.x * t, a.y * (1.0f - t) + b.y * t, a.z * (1.0f - t) + b.z * t, a.w * (1.0f - t) + b.w * t };
    return r;
}

/* repeat addressing: floor(x) mod n for an integer-valued x0, in float so that CPU and GPU agree for any finite x.  Sampler
 * addressing, not a shader `/`: x0 * rcp(n) carries two roundings, so for an n that is not a power of two floor() may be one
 * off either way (x0 = n = 41: 41 * RN(1/41) < 1); the remainder is corrected by one period, not clamped.  For |x0| < 2^22
 * every step is exact integer arithmetic and the result is the mathematical floor(x0) mod n (tests/test_textures.py, against
 * np.mod); beyond it the guards only keep the index inside the level. */
static inline uint32_t wrapRepeat(float x0, uint32_t n)
{
    const float fn = (float)n;
    float m = x0 - floorf(pto_div(x0, fn)) * fn;
    if (m < 0.0f) m += fn;
    else if (m >= fn) m -= fn;
    if (!(m >= 0.0f)) m = 0.0f;
    uint32_t i = (uint32_t)m;
    return i >= n ? n - 1 : i;
}

/* bilinear sample of one level, normalised coordinates */
static v4 sampleLevel(const PtoScene *s, const OTexture *t, uint32_t level, float u, float v)
{
    const uint32_t w = levelDim(t->width, level), h = levelDim(t->height, level);
    if (w == 1 && h == 1) /* exact for 1x1 (hardware weights are fixed point and sum to 1) */
        return fetchTexel(s, t, level, 0, 0);
    if (!(fabsf(u) < 1e9f)) u = 0.0f;
    if (!(fabsf(v) < 1e9f)) v = 0.0f;
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float ax = x - x0, ay = y - y0;
    const uint32_t ix0 = wrapRepeat(x0, w), ix1 = wrapRepeat(x0 + 1.0f, w), iy0 = wrapRepeat(y0, h), iy1 = wrapRepeat(y0 + 1.0f, h);
    const v4 top = v4_lerp(fetchTexel(s, t, level, ix0, iy0), fetchTexel(s, t, level, ix1, iy0), ax);
    const v4 bot = v4_lerp(fetchTexel(s, t, level, ix0, iy1), fetchTexel(s, t, level, ix1, iy1), ax);
    return v4_lerp(top, bot, ay);
}

/* textureGrad with the reference's sampler (trilinear, anisotropy enabled at the device maximum, Renderer.cpp:103-110).
 * Anisotropic filtering is implementation-defined; this follows the scheme of the Vulkan / EXT_texture_filter_anisotropic
 * specifications: rho_x, rho_y = lengths of the two gradients in texels, eta = min(rho_max / rho_min, 16), N = ceil(eta)
 * trilinear taps at LOD log2(rho_max / eta), spaced along the longer gradient at (i / (N + 1) - 1/2), averaged.  With N = 1
 * this is the isotropic lookup. */
#define PTO_MAX_ANISOTROPY 16.0f
static v4 trilinearSample(const PtoScene *s, const OTexture *t, float lod, float u, float v)
{
    const float q = (float)(t->levels - 1);
    if (!(lod >= 0.0f)) lod = 0.0f;
    if (lod > q) lod = q;
    const float d0 = floorf(lod), f = lod - d0;
    const uint32_t l0 = (uint32_t)d0, l1 = l0 + 1 < t->levels ? l0 + 1 : t->levels - 1;
    const v4 c0 = sampleLevel(s, t, l0, u, v);
    if (f == 0.0f || l1 == l0)
        return c0;
    return v4_lerp(c0, sampleLevel(s, t, l1, u, v), f);
}

static v4 textureGradSample(const PtoScene *s, const OTexture *t, float u, float v, float dudx, float dvdx, float dudy, float dvdy)
{
    if (t->levels <= 1)
        return sampleLevel(s, t, 0, u, v);
    const float mux = dudx * (float)t->width, mvx = dvdx * (float)t->height;
    const float muy = dudy * (float)t->width, mvy = dvdy * (float)t->height;
    const float rx = sqrtf(mux * mux + mvx * mvx), ry = sqrtf(muy * muy + mvy * mvy);
    const float rmax = f_max(rx, ry), rmin = f_min(rx, ry);
    float eta = 1.0f;
    if (rmax > 0.0f && rmax < 3.0e38f) /* finite footprint: otherwise a single tap */
        eta = rmin > 0.0f ? f_min(pto_div(rmax, rmin), PTO_MAX_ANISOTROPY) : PTO_MAX_ANISOTROPY;
    if (!(eta >= 1.0f)) eta = 1.0f;
    const float n = ceilf(eta);
    const float rho = pto_div(rmax, eta);
    const float lod = rho > 0.0f ? (float)pto_log2((double)rho) : 0.0f;
    if (n <= 1.0f || lod >= (float)(t->levels - 1)) /* every tap would read the 1x1 top level: one tap */
        return trilinearSample(s, t, lod, u, v);
    const float du = rx >= ry ? dudx : dudy, dv = rx >= ry ? dvdx : dvdy;
    v4 sum = { 0.0f, 0.0f, 0.0f, 0.0f };
    const int taps = (int)n;
    for (int i = 1; i <= taps; i++)
    {
        const float w = pto_div((float)i, n + 1.0f) - 0.5f;
        const v4 c = trilinearSample(s, t, lod, u + du * w, v + dv * w);
        sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
    }
    const float rn = pto_rcp(n);
    sum.x *= rn; sum.y *= rn; sum.z *= rn; sum.w *= rn;
    return sum;
}

/* vkCmdBlitImage with a linear filter (Image.cpp:264-300, TextureUploader.cpp:479-490): level dstLevel of td = level
 * srcLevel of ts decoded, filtered bilinearly at the destination texel centre with clamp-to-edge, re-encoded in the
 * image format.  One level of a mip chain is the blit from the level above it (ts == td). */
static void blitLevel(PtoScene *s, const OTexture *ts, uint32_t srcLevel, const OTexture *td, uint32_t dstLevel)
{
    const uint32_t sw = levelDim(ts->width, srcLevel), sh = levelDim(ts->height, srcLevel);
    const uint32_t dw = levelDim(td->width, dstLevel), dh = levelDim(td->height, dstLevel);
    for (uint32_t j = 0; j < dh; j++)
        for (uint32_t i = 0; i < dw; i++)
        {
            const float x = ((float)i + 0.5f) * ((float)sw / (float)dw) - 0.5f, y = ((float)j + 0.5f) * ((float)sh / (float)dh) - 0.5f;
            const float x0 = floorf(x), y0 = floorf(y), ax = x - x0, ay = y - y0;
            const float cx0 = f_clamp(x0, 0.0f, (float)(sw - 1)), cx1 = f_clamp(x0 + 1.0f, 0.0f, (float)(sw - 1));
            const float cy0 = f_clamp(y0, 0.0f, (float)(sh - 1)), cy1 = f_clamp(y0 + 1.0f, 0.0f, (float)(sh - 1));
            const v4 top = v4_lerp(fetchTexel(s, ts, srcLevel, (uint32_t)cx0, (uint32_t)cy0), fetchTexel(s, ts, srcLevel, (uint32_t)cx1, (uint32_t)cy0), ax);
            const v4 bot = v4_lerp(fetchTexel(s, ts, srcLevel, (uint32_t)cx0, (uint32_t)cy1), fetchTexel(s, ts, srcLevel, (uint32_t)cx1, (uint32_t)cy1), ax);
            const v4 c = v4_lerp(top, bot, ay);
            const size_t idx = td->levelOffset[dstLevel] + (size_t)j * dw + i;
            if (td->format == PTX_TEXTURE_RGBA32F)
            {
                s->texelsF[idx * 4] = c.x; s->texelsF[idx * 4 + 1] = c.y; s->texelsF[idx * 4 + 2] = c.z; s->texelsF[idx * 4 + 3] = c.w;
            }
            else if (td->format == PTX_TEXTURE_RGBA8_SRGB)
                s->texels8[idx] = quantize8(linearToSrgb(c.x)) | quantize8(linearToSrgb(c.y)) << 8 | quantize8(linearToSrgb(c.z)) << 16 | quantize8(c.w) << 24;
            else
                s->texels8[idx] = quantize8(c.x) | quantize8(c.y) << 8 | quantize8(c.z) << 16 | quantize8(c.w) << 24;
        }
}

static uint32_t fullLevels(uint32_t w, uint32_t h)
{
    uint32_t m = w > h ? w : h, levels = 1;
    while (m > 1) { m >>= 1; levels++; } /* floor(log2(max)) + 1, Image.cpp:14-17 */
    return levels > 16 ? 16 : levels;
}

/* TextureUploader::UploadTexturesBlocking / UploadTexture (TextureUploader.cpp:400-511) and DetermineMaxTextureSizes
 * (:551-569): the per-texture share of the budget gives a maximal square extent per format; a larger texture is scaled
 * down by an integer factor -- from the file's own chain when it has one, else by linear blits that halve it --; a
 * texture that fits keeps a complete file chain and gets a generated one otherwise. */
static void uploadTextures(PtoScene *s, const PtxSceneDesc *desc)
{
    for (int c = 0; c < 256; c++)
        s->srgbLut[c] = srgbToLinear((float)c / 255.0f);
    s->textureCount = desc->textures ? desc->textureCount : 0;
    /* the skybox images follow the scene textures in the table, one level each (TextureUploader.cpp:203-262) */
    s->skyKind = desc->skybox ? desc->skyboxKind : PTX_SKYBOX_CLEAR_COLOR;
    const uint32_t skyCount = s->skyKind == PTX_SKYBOX_2D ? 1u : s->skyKind == PTX_SKYBOX_CUBE ? 6u : 0u;
    const uint32_t total = s->textureCount + skyCount;
    if (!total)
        return;
    uint32_t maxExtent[3] = { 4096u, 4096u, 4096u }; /* TextureUploader.h:74 */
    if (!desc->forceFullTextureSize && s->textureCount && desc->textureMemoryBudget != ~0ull)
    {
        /* Config.h:63-64: 1 GiB absolute; the 80 %-of-device-memory term (Config.h:163) is the larger one on any device this
         * runs beside, and the host has no device memory to ask about */
        const uint64_t budget = desc->textureMemoryBudget ? desc->textureMemoryBudget : (1024ull << 20);
        const uint64_t perTexture = budget / s->textureCount;
        for (uint32_t f = 0; f <= PTX_TEXTURE_RGBA32F; f++)
            while (maxExtent[f] > 1u)
            {
                uint64_t texels = 0;
                for (uint32_t e = maxExtent[f]; e; e >>= 1)
                    texels += (uint64_t)e * e;
                if (texels * (f == PTX_TEXTURE_RGBA32F ? 16u : 4u) <= perTexture)
                    break;
                maxExtent[f] >>= 1;
            }
    }
    s->textures = (OTexture *)calloc(total, sizeof(OTexture));
    uint32_t *srcW = (uint32_t *)calloc(total, 4), *srcH = (uint32_t *)calloc(total, 4), *firstFile = (uint32_t *)calloc(total, 4);
    uint32_t *halvings = (uint32_t *)calloc(total, 4);
    uint8_t *useFile = (uint8_t *)calloc(total, 1), *scaled = (uint8_t *)calloc(total, 1);
    size_t n8 = 0, nf = 0, scratch8 = 0, scratchF = 0;
    for (uint32_t i = 0; i < total; i++)
    {
        const PtxTextureDesc *d = i < s->textureCount ? &desc->textures[i] : &desc->skybox[i - s->textureCount];
        OTexture *t = &s->textures[i];
        srcW[i] = d->width ? d->width : 1;
        srcH[i] = d->height ? d->height : 1;
        t->width = srcW[i];
        t->height = srcH[i];
        t->format = d->format;
        t->levels = 1;
        if (i < s->textureCount)
        {
            const uint32_t fileLevels = d->levels ? d->levels : 1u;
            const uint32_t mx = maxExtent[d->format];
            const uint32_t sx = (srcW[i] + mx - 1) / mx, sy = (srcH[i] + mx - 1) / mx, scale = sx > sy ? sx : sy; /* :409-415 */
            t->width = srcW[i] / scale ? srcW[i] / scale : 1;
            t->height = srcH[i] / scale ? srcH[i] / scale : 1;
            t->levels = fullLevels(t->width, t->height);
            if (scale == 1)
                useFile[i] = fileLevels == t->levels && t->levels > 1; /* :440 */
            else
            {
                const uint32_t skip = fileLevels > t->levels ? fileLevels - t->levels : 0u; /* :492-501 */
                if (skip && levelDim(srcW[i], skip) == t->width && levelDim(srcH[i], skip) == t->height)
                {
                    useFile[i] = 1;
                    firstFile[i] = skip;
                }
                else
                {
                    scaled[i] = 1;
                    while (levelDim(srcW[i], halvings[i] + 1) >= t->width && levelDim(srcH[i], halvings[i] + 1) >= t->height &&
                           (levelDim(srcW[i], halvings[i]) > t->width || levelDim(srcH[i], halvings[i]) > t->height))
                        halvings[i]++;
                    size_t need = 0;
                    for (uint32_t l = 0; l <= halvings[i]; l++)
                        need += (size_t)levelDim(srcW[i], l) * levelDim(srcH[i], l);
                    size_t *sc = d->format == PTX_TEXTURE_RGBA32F ? &scratchF : &scratch8;
                    if (need > *sc)
                        *sc = need;
                }
            }
        }
        size_t *cursor = t->format == PTX_TEXTURE_RGBA32F ? &nf : &n8;
        for (uint32_t l = 0; l < t->levels; l++)
        {
            t->levelOffset[l] = *cursor;
            *cursor += (size_t)levelDim(t->width, l) * levelDim(t->height, l);
        }
    }
    s->texels8 = (uint32_t *)calloc(n8 + scratch8 ? n8 + scratch8 : 1, 4);
    s->texelsF = (float *)calloc(nf + scratchF ? nf + scratchF : 1, 16);
    for (uint32_t i = 0; i < total; i++)
    {
        const PtxTextureDesc *d = i < s->textureCount ? &desc->textures[i] : &desc->skybox[i - s->textureCount];
        OTexture *t = &s->textures[i];
        const int isFloat = t->format == PTX_TEXTURE_RGBA32F;
        const size_t texel = isFloat ? 16 : 4;
        uint8_t *pool = isFloat ? (uint8_t *)s->texelsF : (uint8_t *)s->texels8;
        if (d->data && useFile[i])
        {
            const uint8_t *p = (const uint8_t *)d->data;
            for (uint32_t l = 0; l < firstFile[i]; l++)
                p += (size_t)levelDim(srcW[i], l) * levelDim(srcH[i], l) * texel;
            for (uint32_t l = 0; l < t->levels; l++)
            {
                const size_t nl = (size_t)levelDim(t->width, l) * levelDim(t->height, l);
                memcpy(pool + t->levelOffset[l] * texel, p, nl * texel);
                p += nl * texel;
            }
            continue;
        }
        if (d->data && scaled[i])
        {
            OTexture tmp; /* the scratch chain behind the textures of the pool */
            memset(&tmp, 0, sizeof(tmp));
            tmp.width = srcW[i];
            tmp.height = srcH[i];
            tmp.format = t->format;
            tmp.levels = halvings[i] + 1;
            size_t cursor = isFloat ? nf : n8;
            for (uint32_t l = 0; l < tmp.levels; l++)
            {
                tmp.levelOffset[l] = cursor;
                cursor += (size_t)levelDim(tmp.width, l) * levelDim(tmp.height, l);
            }
            memcpy(pool + tmp.levelOffset[0] * texel, d->data, (size_t)srcW[i] * srcH[i] * texel);
            for (uint32_t l = 1; l <= halvings[i]; l++)
                blitLevel(s, &tmp, l - 1, &tmp, l);
            if (levelDim(srcW[i], halvings[i]) == t->width && levelDim(srcH[i], halvings[i]) == t->height)
                memcpy(pool + t->levelOffset[0] * texel, pool + tmp.levelOffset[halvings[i]] * texel, (size_t)t->width * t->height * texel);
            else
                blitLevel(s, &tmp, halvings[i], t, 0);
        }
        else if (d->data)
            memcpy(pool + t->levelOffset[0] * texel, d->data, (size_t)t->width * t->height * texel);
        for (uint32_t l = 1; l < t->levels; l++)
            blitLevel(s, t, l - 1, t, l);
    }
    free(srcW); free(srcH); free(firstFile); free(halvings); free(useFile); free(scaled);
}

int pto_test_texture(const PtoScene *s, const float *in, float *out, uint32_t n, int implicitLod)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float *a = &in[(size_t)i * 7];
        const uint32_t idx = f2u(a[0]);
        v4 r = { 1.0f, 1.0f, 1.0f, 1.0f };
        if (idx >= PTX_SCENE_TEXTURE_OFFSET && idx - PTX_SCENE_TEXTURE_OFFSET < s->textureCount)
        {
            const OTexture *t = &s->textures[idx - PTX_SCENE_TEXTURE_OFFSET];
            r = implicitLod ? sampleLevel(s, t, 0, a[1], a[2]) : textureGradSample(s, t, a[1], a[2], a[3], a[4], a[5], a[6]);
        }
        out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
    }
    return 0;
}

/* ======================================================================== */
/* material.glsl with the 1x1 default textures                              */
/* ======================================================================== */

/* Texels of the fixed slots 0..8 after format decode (ShaderRendererTypes.incl:49-56,
 * formats TextureUploader.cpp:571-594: Color/Specular/Emissive sRGB, others UNORM).
 * Scene textures (index >= 9) go through the software sampler; an index past the uploaded table
 * samples as the white placeholder, like a texture that has not finished loading
 * (Renderer.cpp:421-429). */
static inline v4 sampleTexture(const PtoScene *s, uint32_t idx, v2 uv, v4 dv)
{
    v4 w = { 1.0f, 1.0f, 1.0f, 1.0f };
    if (idx >= PTX_SCENE_TEXTURE_OFFSET && idx - PTX_SCENE_TEXTURE_OFFSET < s->textureCount) /* textureGrad(textures[idx], uv, dv.xy, dv.zw) */
        return textureGradSample(s, &s->textures[idx - PTX_SCENE_TEXTURE_OFFSET], uv.x, uv.y, dv.x, dv.y, dv.z, dv.w);
    switch (idx)
    {
    case PTX_DEFAULT_NORMAL_TEXTURE_INDEX: /* 0xffff8080 UNORM */
        w.x = 128.0f / 255.0f;
        w.y = 128.0f / 255.0f;
        return w;
    case PTX_DEFAULT_EMISSIVE_TEXTURE_INDEX: /* 0x00000000 */
    case PTX_DEFAULT_GLOSSINESS_TEXTURE_INDEX:
    case PTX_DEFAULT_SHININESS_TEXTURE_INDEX:
        w.x = w.y = w.z = w.w = 0.0f;
        return w;
    default:
        return w;
    }
}

/* anyhit.rahit:38-52 / occlusionAnyhit.rahit:37-50: texture(textures[colorIdx], uv) * colorFactor at
 * the candidate hit, with getColorTextureIdx / getColorFactor of material.glsl:25-54.  texture() in a
 * ray-tracing stage has no implicit derivatives: base level. */
static v4 hitBaseColor(const PtoScene *s, uint32_t tri, float u, float v)
{
    const Pair *pr = &s->pairs[s->triPair[tri]];
    const uint32_t prim = s->triPrim[tri];
    const v3 bary = V3(1.0f - u - v, u, v);
    const uint32_t *ix = &pr->ib[prim * 3];
    const PtxVertex *vb = pr->vb;
    v2 uv;
    uv.x = vb[ix[0]].TexCoords[0] * bary.x + vb[ix[1]].TexCoords[0] * bary.y + vb[ix[2]].TexCoords[0] * bary.z;
    uv.y = vb[ix[0]].TexCoords[1] * bary.x + vb[ix[1]].TexCoords[1] * bary.y + vb[ix[2]].TexCoords[1] * bary.z;
    const uint32_t materialType = pr->materialId & 0xffu, materialIndex = pr->materialId >> 8;
    uint32_t idx = 0;
    v4 factor = { 1.0f, 0.0f, 0.0f, 1.0f };
    const float *c = NULL;
    switch (materialType)
    {
    case PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS:
        idx = s->d.metallicRoughnessMaterials[materialIndex].ColorIdx;
        c = s->d.metallicRoughnessMaterials[materialIndex].Color;
        break;
    case PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS:
        idx = s->d.specularGlossinessMaterials[materialIndex].ColorIdx;
        c = s->d.specularGlossinessMaterials[materialIndex].Color;
        break;
    case PTX_MATERIAL_TYPE_PHONG:
        idx = s->d.phongMaterials[materialIndex].ColorIdx;
        c = s->d.phongMaterials[materialIndex].Color;
        break;
    default:
        break;
    }
    if (c)
    {
        factor.x = c[0]; factor.y = c[1]; factor.z = c[2]; factor.w = c[3];
    }
    v4 t;
    if (idx >= PTX_SCENE_TEXTURE_OFFSET && idx - PTX_SCENE_TEXTURE_OFFSET < s->textureCount)
        t = sampleLevel(s, &s->textures[idx - PTX_SCENE_TEXTURE_OFFSET], 0, uv.x, uv.y);
    else
    {
        const v4 zero = { 0.0f, 0.0f, 0.0f, 0.0f };
        t = sampleTexture(s, idx, uv, zero);
    }
    t.x *= factor.x; t.y *= factor.y; t.z *= factor.z; t.w *= factor.w;
    return t;
}

/* material.glsl:55-60 */
static inline v3 ReconstructNormalFromXY(v3 n)
{
    n = V3(2.0f * n.x - 1.0f, 2.0f * n.y - 1.0f, 2.0f * n.z - 1.0f);
    return V3(n.x, n.y, sqrtf(f_max(1 - n.x * n.x - n.y * n.y, 0.0f)));
}

static inline v3 rgb(v4 t) { return V3(t.x, t.y, t.z); }

/* The five textureGrad results a material branch consumes, in the slot order of its struct: emissive, colour, normal,
 * then roughness + metallic (MetallicRoughness), specular + glossiness (SpecularGlossiness) or specular + shininess (Phong). */
typedef struct MaterialTexels
{
    v4 emissive, color, normal, a, b;
} MaterialTexels;

static inline v3 specGlossMetalness(v3 specular, v3 color) /* material.glsl:109-110, :138-139 */
{
    return V3(pto_div(f_max(specular.x - 0.04f, 0.0f), (color.x - 0.04f) + 0.00001f), pto_div(f_max(specular.y - 0.04f, 0.0f), (color.y - 0.04f) + 0.00001f),
              pto_div(f_max(specular.z - 0.04f, 0.0f), (color.z - 0.04f) + 0.00001f));
}

/* material.glsl:62-84 as a function of the texels its textureGrad calls return */
static MaterialSample sampleMaterialMR(const PtxMetallicRoughnessMaterial *m, const MaterialTexels *t, int isHitFromInside)
{
    MaterialSample ret;
    memset(&ret, 0, sizeof(ret));
    ret.EmissiveColor = v_scale(v_add(rgb(t->emissive), V3(m->EmissiveColor[0], m->EmissiveColor[1], m->EmissiveColor[2])), m->EmissiveIntensity);
    ret.Color = v_mul(rgb(t->color), V3(m->Color[0], m->Color[1], m->Color[2]));
    ret.Normal = ReconstructNormalFromXY(rgb(t->normal));
    ret.Roughness = t->a.y * m->Roughness;
    ret.Metalness = t->b.z * m->Metalness;
    ret.Transmission = m->Transmission;
    ret.AttenuationColor = V3(m->AttenuationColor[0], m->AttenuationColor[1], m->AttenuationColor[2]);
    ret.AttenuationDistance = m->AttenuationDistance;
    ret.Eta = isHitFromInside ? m->Ior : (pto_div(1.0f, m->Ior));
    return ret;
}

/* material.glsl:86-113 */
static MaterialSample sampleMaterialSG(const PtxSpecularGlossinessMaterial *m, const MaterialTexels *t, int isHitFromInside)
{
    MaterialSample ret;
    memset(&ret, 0, sizeof(ret));
    ret.EmissiveColor = v_scale(v_add(rgb(t->emissive), V3(m->EmissiveColor[0], m->EmissiveColor[1], m->EmissiveColor[2])), m->EmissiveIntensity);
    ret.Color = v_mul(rgb(t->color), V3(m->Color[0], m->Color[1], m->Color[2]));
    ret.Normal = ReconstructNormalFromXY(rgb(t->normal));
    ret.Transmission = m->Transmission;
    ret.AttenuationColor = V3(m->AttenuationColor[0], m->AttenuationColor[1], m->AttenuationColor[2]);
    ret.AttenuationDistance = m->AttenuationDistance;
    ret.Eta = isHitFromInside ? m->Ior : (pto_div(1.0f, m->Ior));
    const v3 specular = v_mul(rgb(t->a), V3(m->Specular[0], m->Specular[1], m->Specular[2]));
    const float glossiness = t->b.w * m->Glossiness;
    ret.Roughness = 1.0f - glossiness;
    const v3 diff = specGlossMetalness(specular, ret.Color);
    ret.Metalness = pto_div(diff.x + diff.y + diff.z, 3.0f);
    return ret;
}

/* material.glsl:115-142 */
static MaterialSample sampleMaterialPhong(const PtxPhongMaterial *m, const MaterialTexels *t, int isHitFromInside)
{
    MaterialSample ret;
    memset(&ret, 0, sizeof(ret));
    ret.EmissiveColor = v_scale(v_add(rgb(t->emissive), V3(m->EmissiveColor[0], m->EmissiveColor[1], m->EmissiveColor[2])), m->EmissiveIntensity);
    ret.Color = v_mul(rgb(t->color), V3(m->Color[0], m->Color[1], m->Color[2]));
    ret.Normal = ReconstructNormalFromXY(rgb(t->normal));
    ret.Transmission = m->Transmission;
    ret.AttenuationColor = V3(m->AttenuationColor[0], m->AttenuationColor[1], m->AttenuationColor[2]);
    ret.AttenuationDistance = m->AttenuationDistance;
    ret.Eta = isHitFromInside ? m->Ior : (pto_div(1.0f, m->Ior));
    const v3 specular = v_mul(rgb(t->a), V3(m->Specular[0], m->Specular[1], m->Specular[2]));
    const float shininess = t->b.w * m->Shininess;
    ret.Roughness = 1.0f - shininess;
    const v3 diff = specGlossMetalness(specular, ret.Color);
    ret.Metalness = pto_div(diff.x + diff.y + diff.z, 3.0f);
    return ret;
}

/* material.glsl:161-166: unknown material type; the fields the GLSL leaves undefined are zero here */
static MaterialSample unknownMaterial(void)
{
    MaterialSample ret;
    memset(&ret, 0, sizeof(ret));
    ret.Color = V3(1.0f, 0.0f, 0.0f);
    ret.EmissiveColor = V3(1.0f, 0.0f, 0.0f);
    return ret;
}

/* material.glsl:144-171 dispatching to :62-142 */
static MaterialSample sampleMaterial(const PtoScene *s, uint32_t materialId, v2 texCoords, v4 derivatives, int isHitFromInside,
                                     int flipNormalY)
{
#define sampleTexture(idx) sampleTexture(s, (idx), texCoords, derivatives)
    const uint32_t materialType = materialId & 0xffu; /* ShaderTypes.incl:164-168 */
    const uint32_t materialIndex = materialId >> 8;
    MaterialSample ret;
    MaterialTexels t;
    switch (materialType)
    {
    case PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS: {
        const PtxMetallicRoughnessMaterial *m = &s->d.metallicRoughnessMaterials[materialIndex];
        t.emissive = sampleTexture(m->EmissiveIdx); t.color = sampleTexture(m->ColorIdx); t.normal = sampleTexture(m->NormalIdx);
        t.a = sampleTexture(m->RoughnessIdx); t.b = sampleTexture(m->MetallicIdx);
        ret = sampleMaterialMR(m, &t, isHitFromInside);
        break;
    }
    case PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS: {
        const PtxSpecularGlossinessMaterial *m = &s->d.specularGlossinessMaterials[materialIndex];
        t.emissive = sampleTexture(m->EmissiveIdx); t.color = sampleTexture(m->ColorIdx); t.normal = sampleTexture(m->NormalIdx);
        t.a = sampleTexture(m->SpecularIdx); t.b = sampleTexture(m->GlossinessIdx);
        ret = sampleMaterialSG(m, &t, isHitFromInside);
        break;
    }
    case PTX_MATERIAL_TYPE_PHONG: {
        const PtxPhongMaterial *m = &s->d.phongMaterials[materialIndex];
        t.emissive = sampleTexture(m->EmissiveIdx); t.color = sampleTexture(m->ColorIdx); t.normal = sampleTexture(m->NormalIdx);
        t.a = sampleTexture(m->SpecularIdx); t.b = sampleTexture(m->ShininessIdx);
        ret = sampleMaterialPhong(m, &t, isHitFromInside);
        break;
    }
    default: /* material.glsl:163-166 */
        ret = unknownMaterial();
        break;
    }
    if (flipNormalY)
        ret.Normal.y *= -1;
    return ret;
#undef sampleTexture
}

/* ======================================================================== */
/* closestHit.rchit / miss.rmiss                                            */
/* ======================================================================== */

/* ShaderRendererTypes.incl:101-118 */
typedef struct Payload
{
    DiffRays diff; /* RayDifferentials0..2 */
    v3 Position;
    v3 Direction;
    float MaxRoughness;
    v3 Bsdf;
    float Pdf;
    v3 Emissive;
    uint32_t RngState;
    v3 DirectLight;
    float DirectLightPdf;
    v3 LightDirection;
    float LightDistance;
} Payload;

/* common.glsl:27-46 */
static inline Vtx getVertex(const PtoScene *s, const Pair *pr, uint32_t offset)
{
    (void)s;
    const PtxVertex *p = &pr->vb[pr->ib[offset]];
    Vtx v;
    v.Position = V3(p->Position[0], p->Position[1], p->Position[2]);
    v.TexCoords.x = p->TexCoords[0];
    v.TexCoords.y = p->TexCoords[1];
    v.Normal = V3(p->Normal[0], p->Normal[1], p->Normal[2]);
    v.Tangent = V3(p->Tangent[0], p->Tangent[1], p->Tangent[2]);
    v.Bitangent = V3(p->Bitangent[0], p->Bitangent[1], p->Bitangent[2]);
    return v;
}

/* common.glsl:107-110 */
static inline v3 interp3(v3 a, v3 b, v3 c, v3 bc)
{
    return v_add(v_add(v_scale(a, bc.x), v_scale(b, bc.y)), v_scale(c, bc.z));
}

/* sampling.glsl:5-15: M = A_instance * A_mesh; the normal goes through inverse-transpose */
static inline Vtx transformVertex(const Pair *pr, Vtx v)
{
    v.Position = xformPoint(pr->M, v.Position);
    v.Tangent = v_normalize(xformVector(pr->M, v.Tangent));
    v.Bitangent = v_normalize(xformVector(pr->M, v.Bitangent));
    v.Normal = v_normalize(V3(v_dot(v.Normal, pr->Rinv.c0), v_dot(v.Normal, pr->Rinv.c1), v_dot(v.Normal, pr->Rinv.c2)));
    return v;
}

/* PTO_DEBUG_PIXEL="x,y": print the light-sampling and BSDF inputs / outputs of every hit of that pixel (bit patterns),
 * to bisect a mismatch against the HIP path with ptx_test_eval */
static __thread int g_dbgPixel;
static int g_dbgX = -1, g_dbgY = -1;
static void dbg3(const char *name, v3 a)
{
    fprintf(stderr, "[pto] %s %08x %08x %08x (%.9g %.9g %.9g)\n", name, f2u(a.x), f2u(a.y), f2u(a.z), a.x, a.y, a.z);
}
static void dbg1(const char *name, float a) { fprintf(stderr, "[pto] %s %08x (%.9g)\n", name, f2u(a), a); }

/* closestHit.rchit:52-161 */
static void closestHit(const PtoScene *s, const PtxLightsUbo *lights, v3 rayOriginW, v3 rayDirW, const PtoHit *hit,
                       Payload *payload)
{
    const v3 bary = V3(1.0f - hit->u - hit->v, hit->u, hit->v); /* common.glsl:22-25 */
    const Pair *pr = &s->pairs[s->triPair[hit->tri]];
    const uint32_t prim = s->triPrim[hit->tri];

    const Vtx o0 = getVertex(s, pr, prim * 3), o1 = getVertex(s, pr, prim * 3 + 1), o2 = getVertex(s, pr, prim * 3 + 2);
    Vtx ov; /* common.glsl:112-130 */
    ov.Position = interp3(o0.Position, o1.Position, o2.Position, bary);
    ov.TexCoords.x = o0.TexCoords.x * bary.x + o1.TexCoords.x * bary.y + o2.TexCoords.x * bary.z;
    ov.TexCoords.y = o0.TexCoords.y * bary.x + o1.TexCoords.y * bary.y + o2.TexCoords.y * bary.z;
    ov.Normal = interp3(o0.Normal, o1.Normal, o2.Normal, bary);
    ov.Tangent = interp3(o0.Tangent, o1.Tangent, o2.Tangent, bary);
    ov.Bitangent = interp3(o0.Bitangent, o1.Bitangent, o2.Bitangent, bary);
    Vtx vertex = transformVertex(pr, ov);

    const Vtx v0 = transformVertex(pr, o0), v1 = transformVertex(pr, o1), v2 = transformVertex(pr, o2);

    const v3 edge1 = v_sub(v1.Position, v0.Position);
    const v3 edge2 = v_sub(v2.Position, v0.Position);
    v3 geometricNormal = v_normalize(v_cross(edge1, edge2));

    const int isHitFromInside = v_dot(geometricNormal, rayDirW) > 0.0f;
    if (isHitFromInside)
    {
        geometricNormal = v_neg(geometricNormal);
        vertex.Normal = v_neg(vertex.Normal);
        vertex.Tangent = v_neg(vertex.Tangent);
        vertex.Bitangent = v_neg(vertex.Bitangent);
    }

    /* :88-99 texture footprint from the ray differentials */
    v3 dpdu, dpdv, dndu, dndv;
    {
        const v3 P3[3] = { v0.Position, v1.Position, v2.Position }, N3[3] = { v0.Normal, v1.Normal, v2.Normal };
        const texcoord2 UV3[3] = { o0.TexCoords, o1.TexCoords, o2.TexCoords };
        computeDpnDuv(P3, N3, UV3, vertex.Tangent, vertex.Bitangent, &dpdu, &dpdv, &dndu, &dndv);
    }
    v3 dpdx, dpdy;
    computeDpDxy(vertex.Position, rayOriginW, v_normalize(rayDirW), payload->diff.rxOrigin, payload->diff.rxDirection,
                 payload->diff.ryOrigin, payload->diff.ryDirection, vertex.Normal, &dpdx, &dpdy);
    const v4 derivatives = computeDerivatives(dpdx, dpdy, dpdu, dpdv);

    MaterialSample material = sampleMaterial(s, pr->materialId, ov.TexCoords, derivatives, isHitFromInside, s->d.dxNormalTextures != 0);

    /* :105-106 decals: an ignored alpha < 0.5 hit in front of this one tints the base colour */
    if (payload->DirectLightPdf != -1.0f && hit->t > payload->DirectLightPdf)
        material.Color = v_mix(material.Color, payload->LightDirection, payload->LightDistance);

    payload->MaxRoughness = f_max(material.Roughness, payload->MaxRoughness); /* :109 */
    material.Roughness = f_max(payload->MaxRoughness, 0.01f);                 /* :112 */

    m3 geometryTBN;
    geometryTBN.c0 = vertex.Tangent;
    geometryTBN.c1 = vertex.Bitangent;
    geometryTBN.c2 = vertex.Normal;
    const v3 N = v_normalize(v_add(vertex.Normal, m3_mul(geometryTBN, material.Normal)));
    const m3 TBN = computeTangentSpace(N);
    const m3 invTBN = m3_inverse(TBN);
    const v3 V = v_normalize(m3_mul(invTBN, v_normalize(v_neg(rayDirW))));

    uint32_t rngState = payload->RngState;
    BSDFSample bsdf = sampleBSDF(&material, V, &rngState);

    if (isHitFromInside) /* :123-128 Beer-Lambert over this segment */
    {
        const float e = pto_div(hit->t, material.AttenuationDistance);
        bsdf.Color.x *= pto_powf(material.AttenuationColor.x, e);
        bsdf.Color.y *= pto_powf(material.AttenuationColor.y, e);
        bsdf.Color.z *= pto_powf(material.AttenuationColor.z, e);
    }

    const int isRefracted = bsdf.Direction.z < 0.0f;
    const v3 rayOrigin = offsetRayOriginShadowTerminator(&vertex, &v0, &v1, &v2, bary, isRefracted);

    float lightPdf, lightSmplPdf;
    v3 u3;
    u3.x = rnd(&rngState);
    u3.y = rnd(&rngState);
    u3.z = rnd(&rngState);
    const LightSample light = sampleLight(lights, u3, rayOrigin, &lightPdf);
    const v3 L = v_normalize(m3_mul(invTBN, v_neg(light.Direction)));
    const v3 lightBsdf = evaluateBSDF(&material, V, L, &lightSmplPdf);
    if (g_dbgPixel)
    {
        dbg3("rayOriginW", rayOriginW); dbg3("rayDirW", rayDirW); dbg1("hit.t", hit->t);
        fprintf(stderr, "[pto] hit.tri %u\n", hit->tri);
        dbg3("tri.v0", V3(s->v0[hit->tri * 3], s->v0[hit->tri * 3 + 1], s->v0[hit->tri * 3 + 2]));
        dbg3("tri.e1", V3(s->e1[hit->tri * 3], s->e1[hit->tri * 3 + 1], s->e1[hit->tri * 3 + 2]));
        dbg3("tri.e2", V3(s->e2[hit->tri * 3], s->e2[hit->tri * 3 + 1], s->e2[hit->tri * 3 + 2]));
        dbg3("u3", u3); dbg3("rayOrigin", rayOrigin); dbg3("light.Direction", light.Direction); dbg1("light.Distance", light.Distance);
        dbg3("light.Color", light.Color); dbg1("light.Attenuation", light.Attenuation); dbg1("lightPdf", lightPdf);
        dbg3("V", V); dbg3("L", L); dbg3("lightBsdf", lightBsdf); dbg1("lightSmplPdf", lightSmplPdf);
        dbg3("material.Color", material.Color); dbg1("material.Roughness", material.Roughness); dbg1("material.Metalness", material.Metalness);
        dbg1("material.Eta", material.Eta); dbg1("material.Transmission", material.Transmission); dbg3("material.EmissiveColor", material.EmissiveColor);
        dbg3("bsdf.Direction", bsdf.Direction); dbg3("bsdf.Color", bsdf.Color); dbg1("bsdf.Pdf", bsdf.Pdf);
        dbg3("N", N); dbg3("vertex.Position", vertex.Position); dbg3("vertex.Normal", vertex.Normal);
        dbg3("invTBN.c0", invTBN.c0); dbg3("invTBN.c1", invTBN.c1); dbg3("invTBN.c2", invTBN.c2);
        dbg3("v0.P", v0.Position); dbg3("v1.P", v1.Position); dbg3("v2.P", v2.Position); dbg3("v0.N", v0.Normal); dbg3("v1.N", v1.Normal); dbg3("v2.N", v2.Normal);
        dbg3("geometricNormal", geometricNormal); dbg3("bary", bary); dbg3("vertex.Tangent", vertex.Tangent); dbg3("vertex.Bitangent", vertex.Bitangent);
    }

    payload->Direction = v_normalize(m3_mul(TBN, bsdf.Direction));
    if (isRefracted)
        payload->Position = offsetRayOriginSelfIntersection(vertex.Position, v_neg(geometricNormal));
    else
        payload->Position = rayOrigin;
    payload->Bsdf = bsdf.Color;
    payload->Pdf = bsdf.Pdf;
    payload->Emissive = material.EmissiveColor;
    payload->RngState = rngState;
    payload->DirectLight = v_mul(v_scale(light.Color, light.Attenuation), lightBsdf);
    payload->DirectLightPdf = lightPdf;
    payload->LightDirection = light.Direction;
    payload->LightDistance = light.Distance;
    if (g_dbgPixel)
    {
        dbg3("payload.Position", payload->Position); dbg3("payload.Direction", payload->Direction);
        dbg1("hit.u", hit->u); dbg1("hit.v", hit->v);
    }

    /* :150-160 differentials of the continuation ray */
    if (isRefracted)
        computeRefractedDifferentialRays(derivatives, vertex.Normal, rayOrigin, v_neg(rayDirW), payload->Direction, dndu, dndv, material.Eta,
                                         &payload->diff);
    else
        computeReflectedDifferentialRays(derivatives, vertex.Normal, rayOrigin, v_neg(rayDirW), payload->Direction, dndu, dndv, &payload->diff);
}

/* common.glsl:17-20 */
static inline v3 hdrToLdr(v3 rgb) { return v_div(rgb, 1.0f + maxComponent(rgb)); }

/* miss.rmiss:20-25: equirectangular coordinates of a direction */
static inline v2 missSkyboxTexCoords(v3 dir)
{
    const float PI = 3.14159265359f; /* common.glsl:3 */
    const float longitude = pto_atan2f(dir.z, dir.x);
    const float latitude = pto_asinf(-dir.y);
    v2 uv;
    uv.x = pto_div(pto_div(longitude, 2.0f), PI) + 0.5f;
    uv.y = pto_div(latitude, PI) + 0.5f;
    return uv;
}

/* Cube map face selection of the Vulkan specification (largest magnitude, z before y before x on ties): face and the
 * face coordinates (sc, tc) with major axis length ma. */
static inline void cubeFace(v3 r, uint32_t *face, float *sc, float *tc, float *ma)
{
    const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
    if (az >= ax && az >= ay)
    {
        *face = r.z < 0.0f ? 5u : 4u;
        *sc = r.z < 0.0f ? -r.x : r.x;
        *tc = -r.y;
        *ma = az;
    }
    else if (ay >= ax)
    {
        *face = r.y < 0.0f ? 3u : 2u;
        *sc = r.x;
        *tc = r.y < 0.0f ? -r.z : r.z;
        *ma = ay;
    }
    else
    {
        *face = r.x < 0.0f ? 1u : 0u;
        *sc = r.x < 0.0f ? r.z : -r.z;
        *tc = -r.y;
        *ma = ax;
    }
}

/* Texel (ix, iy) of a face, where ONE of the indices may lie one step outside 0 .. n-1: the texel across that edge.
 * (Seamless cube maps: Vulkan samples "Cube Map Edge Handling" -- the reference's sampler filters across face borders.)
 * The texel centre, folded over the edge onto the cube's surface, is looked up again through the face selection: the
 * index along the edge is kept, the index across it becomes the neighbour's border row. */
static v4 cubeTexel(const PtoScene *s, const OTexture *faces, uint32_t face, int ix, int iy)
{
    const int n = (int)faces[face].width;
    if (ix >= 0 && ix < n && iy >= 0 && iy < n)
        return fetchTexel(s, &faces[face], 0, (uint32_t)ix, (uint32_t)iy);
    float sc = 2.0f * (pto_div((float)ix + 0.5f, (float)n)) - 1.0f, tc = 2.0f * (pto_div((float)iy + 0.5f, (float)n)) - 1.0f, ma = 1.0f;
    if (ix < 0 || ix >= n)
    {
        ma = 1.0f - (fabsf(sc) - 1.0f);
        sc = sc < 0.0f ? -1.0f : 1.0f;
    }
    else
    {
        ma = 1.0f - (fabsf(tc) - 1.0f);
        tc = tc < 0.0f ? -1.0f : 1.0f;
    }
    v3 r;
    switch (face) /* inverse of the selection table */
    {
    case 0: r = V3(ma, -tc, -sc); break;
    case 1: r = V3(-ma, -tc, sc); break;
    case 2: r = V3(sc, ma, tc); break;
    case 3: r = V3(sc, -ma, -tc); break;
    case 4: r = V3(sc, -tc, ma); break;
    default: r = V3(-sc, -tc, -ma); break;
    }
    uint32_t f2;
    float s2, t2, m2;
    cubeFace(r, &f2, &s2, &t2, &m2);
    const float mx = (float)(n - 1);
    const uint32_t jx = (uint32_t)f_clamp(floorf((0.5f * (pto_div(s2, m2)) + 0.5f) * (float)n), 0.0f, mx);
    const uint32_t jy = (uint32_t)f_clamp(floorf((0.5f * (pto_div(t2, m2)) + 0.5f) * (float)n), 0.0f, mx);
    return fetchTexel(s, &faces[f2], 0, jx, jy);
}

/* bilinear lookup at (u, v) of a face with the footprint continuing on the neighbouring faces; at a corner of the cube,
 * where three faces meet and the fourth texel does not exist, it is the mean of the other three (Vulkan "Cube Map
 * Corner Handling") */
static v4 sampleFaceSeamless(const PtoScene *s, const OTexture *faces, uint32_t face, float u, float v)
{
    const int n = (int)faces[face].width;
    if (!(fabsf(u) < 1e9f)) u = 0.0f;
    if (!(fabsf(v) < 1e9f)) v = 0.0f;
    const float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float ax = x - x0, ay = y - y0;
    const int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
    const int ox0 = ix0 < 0, ox1 = ix1 >= n, oy0 = iy0 < 0, oy1 = iy1 >= n;
    v4 c00, c10, c01, c11;
    const v4 zero = { 0.0f, 0.0f, 0.0f, 0.0f };
    c00 = (ox0 && oy0) ? zero : cubeTexel(s, faces, face, ix0, iy0);
    c10 = (ox1 && oy0) ? zero : cubeTexel(s, faces, face, ix1, iy0);
    c01 = (ox0 && oy1) ? zero : cubeTexel(s, faces, face, ix0, iy1);
    c11 = (ox1 && oy1) ? zero : cubeTexel(s, faces, face, ix1, iy1);
    if ((ox0 || ox1) && (oy0 || oy1))
    {
        const float third = 1.0f / 3.0f;
        const v4 sum = { (c00.x + c10.x) + (c01.x + c11.x), (c00.y + c10.y) + (c01.y + c11.y), (c00.z + c10.z) + (c01.z + c11.z),
                         (c00.w + c10.w) + (c01.w + c11.w) };
        const v4 mean = { sum.x * third, sum.y * third, sum.z * third, sum.w * third };
        if (ox0 && oy0) c00 = mean;
        else if (ox1 && oy0) c10 = mean;
        else if (ox0 && oy1) c01 = mean;
        else c11 = mean;
    }
    const v4 top = v4_lerp(c00, c10, ax);
    const v4 bot = v4_lerp(c01, c11, ax);
    return v4_lerp(top, bot, ay);
}

/* texture(samplerCube, dir): face selection and (s, t) of the Vulkan spec's cube map face selection
 * tables; bilinear filtering with seamless edges. */
static v4 sampleCube(const PtoScene *s, const OTexture *faces, v3 r)
{
    uint32_t face;
    float sc, tc, ma;
    cubeFace(r, &face, &sc, &tc, &ma);
    const float u = 0.5f * (pto_div(sc, ma)) + 0.5f, v = 0.5f * (pto_div(tc, ma)) + 0.5f;
    return sampleFaceSeamless(s, faces, face, u, v);
}

/* miss.rmiss:16-39 */
static inline void missShader(const PtoScene *s, v3 rayDir, Payload *payload)
{
    if (s->skyKind == PTX_SKYBOX_2D)
    {
        const v2 uv = missSkyboxTexCoords(rayDir);
        const v4 c = sampleLevel(s, &s->textures[s->textureCount], 0, uv.x, uv.y);
        payload->Emissive = hdrToLdr(V3(c.x, c.y, c.z));
    }
    else if (s->skyKind == PTX_SKYBOX_CUBE)
    {
        const v4 c = sampleCube(s, &s->textures[s->textureCount], rayDir);
        payload->Emissive = V3(c.x, c.y, c.z);
    }
    else
        payload->Emissive = V3(0.08f, 0.09f, 0.1f);
    payload->Pdf = -1.0f;
}

/* ======================================================================== */
/* raygen.rgen                                                              */
/* ======================================================================== */

/* Scripted trace calls (pto_test_raygen): the stage-level golden vectors of tests/golden/golden_stage_*.json run the
 * reference's raygen.rgen main() with traceRayEXT replaced by a script -- record k of the script is what the closest-hit /
 * miss stage leaves in the payload on the k-th primary trace (and whether the shadow query of that bounce is occluded);
 * past the end of the script a trace misses into black.  The same script drives raygenPixel here, so the loop itself
 * (RNG draws, the order of the radiance / throughput updates, roulette, the NaN / inf restart) is compared with the
 * reference's text, not with a second restatement.  Every ray handed to a trace call is folded into a hash. */
#define PTO_SCRIPT_RECORDS 12
#define PTO_SCRIPT_RECORD_WORDS 23
typedef struct
{
    const uint32_t *records; /* PTO_SCRIPT_RECORDS x PTO_SCRIPT_RECORD_WORDS */
    uint32_t calls0, calls1, hash;
} TraceScript;
static __thread TraceScript *g_script;

static void scriptHashRay(v3 o, float tmin, v3 d, float tmax)
{
    const float f[8] = { o.x, o.y, o.z, tmin, d.x, d.y, d.z, tmax };
    for (int k = 0; k < 8; k++)
        g_script->hash = (g_script->hash ^ f2u(f[k])) * 16777619u;
}

static void scriptedClosest(Payload *payload, Ray ray)
{
    scriptHashRay(ray.Origin, ray.tmin, ray.Direction, ray.tmax);
    const uint32_t k = g_script->calls0++;
    if (k >= PTO_SCRIPT_RECORDS)
    {
        payload->Emissive = v3s(0.0f);
        payload->Pdf = -1.0f;
        return;
    }
    const uint32_t *r = &g_script->records[k * PTO_SCRIPT_RECORD_WORDS];
    payload->Position = V3(u2f(r[0]), u2f(r[1]), u2f(r[2]));
    payload->Direction = V3(u2f(r[3]), u2f(r[4]), u2f(r[5]));
    payload->Emissive = V3(u2f(r[6]), u2f(r[7]), u2f(r[8]));
    payload->Bsdf = V3(u2f(r[9]), u2f(r[10]), u2f(r[11]));
    payload->Pdf = u2f(r[12]);
    payload->DirectLight = V3(u2f(r[13]), u2f(r[14]), u2f(r[15]));
    payload->DirectLightPdf = u2f(r[16]);
    payload->LightDirection = V3(u2f(r[17]), u2f(r[18]), u2f(r[19]));
    payload->LightDistance = u2f(r[20]);
    for (uint32_t d = 0; d < r[21]; d++) /* the draws the hit stage would have made */
        (void)rnd(&payload->RngState);
}

static int scriptedOccluded(v3 position, v3 direction, float tmin, float tmax)
{
    scriptHashRay(position, tmin, direction, tmax);
    g_script->calls1++;
    const uint32_t k = g_script->calls0 - 1; /* the bounce that asked */
    return k < PTO_SCRIPT_RECORDS ? g_script->records[k * PTO_SCRIPT_RECORD_WORDS + 22] != 0u : 1;
}

/* raygen.rgen:22-34 */
static inline int checkOccluded(const PtoScene *s, v3 lightDir, v3 position, float dist, int brute, PtoStats *st)
{
    const v3 direction = v_neg(v_normalize(lightDir));
    st->shadowRays++;
    if (g_script)
        return scriptedOccluded(position, direction, 0.00001f, dist);
    return traceAny(s, position, direction, 0.00001f, dist, brute, st);
}

/* raygen.rgen:36-118 for one pixel */
static void raygenPixel(const PtoScene *s, const PtxRaygenUniformData *U, const PtxLightsUbo *lights, uint32_t px,
                        uint32_t py, uint32_t W, uint32_t H, float *accum, int brute, PtoStats *st)
{
    uint32_t rngState = initRng(px, py, W, U->TotalSamples);
    v3 radiance = v3s(0.0f);
    Payload payload;
    memset(&payload, 0, sizeof(payload));
    g_dbgPixel = (int)px == g_dbgX && (int)py == g_dbgY;

    for (int smpl = 0; smpl < (int)U->SampleCount; smpl++)
    {
        v3 throughput = v3s(1.0f);
        v2 u;
        u.x = rnd(&rngState);
        u.y = rnd(&rngState);
        Ray ray, rx, ry;
        if (U->LensRadius > 0)
        {
            v2 u2;
            u2.x = rnd(&rngState);
            u2.y = rnd(&rngState);
            ray = constructPrimaryRayLensD(px, py, W, H, U->ViewInverse, U->ProjInverse, u, u2, U->LensRadius, U->FocalDistance, &rx, &ry);
        }
        else
            ray = constructPrimaryRayD(px, py, W, H, U->ViewInverse, U->ProjInverse, u, &rx, &ry);
        payload.diff.rxOrigin = rx.Origin; /* raygen.rgen:56-58 */
        payload.diff.rxDirection = rx.Direction;
        payload.diff.ryOrigin = ry.Origin;
        payload.diff.ryDirection = ry.Direction;

        payload.MaxRoughness = 0.0f;

        for (uint32_t bounce = 0; bounce < U->BounceCount; bounce++)
        {
            payload.RngState = rngState;
            payload.DirectLightPdf = -1.0f;
            payload.LightDirection = v3s(0.0f);
            payload.LightDistance = 0.0f;
            st->segments++;
            if (g_script)
                scriptedClosest(&payload, ray);
            else
            {
                Decal decal;
                const PtoHit hit = traceClosest(s, ray.Origin, ray.Direction, ray.tmin, ray.tmax, brute, st, &decal);
                if (decal.dist != -1.0f) /* anyhit.rahit:57-59 */
                {
                    payload.LightDirection = decal.color;
                    payload.LightDistance = decal.alpha;
                    payload.DirectLightPdf = decal.dist;
                }
                if (hit.tri == 0xffffffffu)
                    missShader(s, ray.Direction, &payload);
                else
                    closestHit(s, lights, ray.Origin, ray.Direction, &hit, &payload);
            }
            rngState = payload.RngState;

            if (payload.Pdf == -1.0f)
            {
                radiance = v_add(radiance, v_mul(throughput, payload.Emissive));
                break;
            }

            radiance = v_add(radiance, v_mul(throughput, payload.Emissive));

            if (payload.DirectLightPdf > 0.0f)
                if (!checkOccluded(s, payload.LightDirection, payload.Position, payload.LightDistance, brute, st))
                    radiance = v_add(radiance, v_div(v_mul(throughput, payload.DirectLight), payload.DirectLightPdf));

            if (payload.Pdf > 0.001f)
                throughput = v_mul(throughput, v_div(payload.Bsdf, payload.Pdf));

            const float prob = f_min(maxComponent(throughput), 1.0f);
            if (prob < 0.001f)
                break;
            if (prob < rnd(&rngState))
                break;
            throughput = v_div(throughput, prob);

            ray.Origin = payload.Position;
            ray.Direction = payload.Direction;
        }

        st->pathSamples++;
        /* :99-112 NaN / Inf sample rejection restarts the whole sample loop */
        if (isnan(radiance.x) || isnan(radiance.y) || isnan(radiance.z) || isinf(radiance.x) || isinf(radiance.y) ||
            isinf(radiance.z))
        {
            radiance = v3s(0.0f);
            smpl = -1;
            st->retries++;
            continue;
        }
    }

    float *p = &accum[((size_t)py * W + px) * 4]; /* :115-117 */
    p[0] = radiance.x + p[0];
    p[1] = radiance.y + p[1];
    p[2] = radiance.z + p[2];
    p[3] = 1.0f;
}

int pto_render(const PtoScene *s, const PtxRaygenUniformData *U, const PtxLightsUbo *lights, uint32_t W, uint32_t H,
               uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const PtxTileShard *shard, float *accum, int threads,
               int brute, PtoStats *stats)
{
    if (!s || !U || !lights || !accum || x1 > W || y1 > H)
        return 1;
    PtoStats total;
    memset(&total, 0, sizeof(total));
    total.triangles = s->triCount;
    g_dbgX = g_dbgY = -1;
    if (getenv("PTO_DEBUG_PIXEL"))
        sscanf(getenv("PTO_DEBUG_PIXEL"), "%d,%d", &g_dbgX, &g_dbgY);
#ifdef _OPENMP
    if (threads > 0)
        omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    const uint32_t ts = shard && shard->tileSize ? shard->tileSize : 16;
    const uint32_t tilesX = (W + ts - 1) / ts, tilesY = (H + ts - 1) / ts;
    const int64_t ntiles = (int64_t)tilesX * tilesY;
#pragma omp parallel
    {
        PtoStats st;
        memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 1)
        for (int64_t tile = 0; tile < ntiles; tile++)
        {
            if (shard && shard->worldSize > 1 && (uint32_t)(tile % shard->worldSize) != shard->rank)
                continue;
            const uint32_t tx = (uint32_t)(tile % tilesX), ty = (uint32_t)(tile / tilesX);
            for (uint32_t y = ty * ts; y < (ty + 1) * ts && y < H; y++)
                for (uint32_t x = tx * ts; x < (tx + 1) * ts && x < W; x++)
                    if (x >= x0 && x < x1 && y >= y0 && y < y1)
                        raygenPixel(s, U, lights, x, y, W, H, accum, brute, &st);
        }
#pragma omp critical
        {
            total.pathSamples += st.pathSamples;
            total.segments += st.segments;
            total.shadowRays += st.shadowRays;
            total.retries += st.retries;
            total.nodesVisited += st.nodesVisited;
            total.trisTested += st.trisTested;
        }
    }
    if (stats)
        *stats = total;
    return 0;
}

/* ======================================================================== */
/* Function-level entry (packing documented in include/ptx.h)               */
/* ======================================================================== */

static const int kInStride[PTX_FN_COUNT] = { 4, 4, 4, 2, 1, 10, 11, 6, 3, 14, 12, 4, 2, 2, 3, 6, 38, 1, 2, 31, 25, 42, 30, 24, 12, 34, 35, 4, 3, 3, 2, 6, 7, 3, 47, 2, 1, 1 };
static const int kOutStride[PTX_FN_COUNT] = { 1, 1, 1, 1, 1, 4, 4, 3, 4, 4, 8, 5, 2, 3, 9, 3, 18, 2, 1, 9, 3, 18, 12, 6, 4, 12, 12, 1, 2, 3, 2, 6, 3, 3, 17, 2, 1, 1 };

static MaterialSample unpackMaterial(const float *p)
{
    MaterialSample m;
    memset(&m, 0, sizeof(m));
    m.Color = V3(p[0], p[1], p[2]);
    m.Roughness = p[3];
    m.Metalness = p[4];
    m.Transmission = p[5];
    m.Eta = p[6];
    return m;
}

/* r == RN(1 / sqrt(x)) exactly?  x = X 2^ex, r = R 2^er (24-bit integers): 1 / sqrt(x) lies between the midpoints m_lo < r < m_hi
 * iff x m_hi^2 >= 1 >= x m_lo^2 (equality only where 1 / sqrt(x) is a float itself). */
static int rsqExactlyRounded(float x, float r)
{
    int ex = 0, er = 0;
    const float fx = frexpf(x, &ex), fr = frexpf(r, &er);
    const uint64_t X = (uint64_t)ldexpf(fx, 24), R = (uint64_t)ldexpf(fr, 24);
    ex -= 24;
    er -= 24;
    const unsigned __int128 mhi = 2 * R + 1; /* m_hi = (2 R + 1) 2^(er - 1) */
    const int ehi = er - 1;
    unsigned __int128 mlo;                   /* below a power of two the spacing halves */
    int elo;
    if (R == (1u << 23)) { mlo = 4 * R - 1; elo = er - 2; } else { mlo = 2 * R - 1; elo = er - 1; }
    const unsigned __int128 phi = (unsigned __int128)X * mhi * mhi, plo = (unsigned __int128)X * mlo * mlo;
    const int shi = -(ex + 2 * ehi), slo = -(ex + 2 * elo); /* X m^2 against 2^s */
    const int okhi = shi < 0 ? 1 : (shi > 126 ? 0 : phi >= ((unsigned __int128)1 << shi));
    const int oklo = slo < 0 ? 0 : (slo > 126 ? 1 : plo <= ((unsigned __int128)1 << slo));
    return okhi && oklo;
}

uint64_t pto_rsq_selfcheck(uint64_t *seedDependent)
{
    uint64_t wrong = 0, dep = 0;
    for (uint32_t parity = 0; parity < 2; parity++)
        for (uint32_t m = 0; m < (1u << 23); m++)
        {
            const float x = u2f(((127u + parity) << 23) | m);
            const float want = pto_rsq(x);
            int bad = !rsqExactlyRounded(x, want), differs = 0;
            float first = 0.0f;
            for (int d = -1; d <= 1; d++) /* the device's sequence (pt_device.hpp rsq_) from every seed v_rsq_f32's 1 ULP allows */
            {
                const float y0 = u2f(f2u(want) + (uint32_t)d);
                const float h = x * y0;
                const float l = fmaf(x, y0, -h);
                float e = fmaf(-h, y0, 1.0f);
                e = fmaf(-l, y0, e);
                const float p = fmaf(0.375f, e, 0.5f);
                const float y1 = fmaf(y0 * e, p, y0);
                if (d == -1)
                    first = y1;
                else if (y1 != first)
                    differs = 1;
                if (y1 != want)
                    bad = 1;
            }
            wrong += (uint64_t)bad;
            dep += (uint64_t)differs;
        }
    if (seedDependent)
        *seedDependent = dep;
    return wrong;
}

int pto_test_eval(uint32_t fn, const float *in, float *out, uint32_t n)
{
    if (fn >= PTX_FN_COUNT)
        return 1;
    if (fn >= PTX_FN_POSTPROCESS_PIXEL && fn <= PTX_FN_TONEMAP_PIXEL) /* output stage: pt_oracle_post.c */
        return pto_test_post(fn - PTX_FN_POSTPROCESS_PIXEL, in, out, n);
    for (uint32_t i = 0; i < n; i++)
    {
        const float *a = &in[(size_t)i * kInStride[fn]];
        float *o = &out[(size_t)i * kOutStride[fn]];
        switch (fn)
        {
        case PTX_FN_GGX_DISTRIBUTION: o[0] = GGXDistribution(V3(a[0], a[1], a[2]), a[3]); break;
        case PTX_FN_LAMBDA: o[0] = Lambda(V3(a[0], a[1], a[2]), a[3]); break;
        case PTX_FN_GGX_SMITH: o[0] = GGXSmith(V3(a[0], a[1], a[2]), a[3]); break;
        case PTX_FN_DIELECTRIC_FRESNEL: o[0] = DielectricFresnel(a[0], a[1]); break;
        case PTX_FN_SCHLICK_FRESNEL: o[0] = SchlickFresnel(a[0]); break;
        case PTX_FN_EVALUATE_REFLECTION: {
            float pdf;
            const v3 r = EvaluateReflection(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]), a[9], &pdf);
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf;
            break;
        }
        case PTX_FN_EVALUATE_REFRACTION: {
            float pdf;
            const v3 r = EvaluateRefraction(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]), a[9], a[10], &pdf);
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf;
            break;
        }
        case PTX_FN_SAMPLE_GGX: {
            v2 u = { a[0], a[1] };
            const v3 r = SampleGGX(u, V3(a[2], a[3], a[4]), a[5]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case PTX_FN_SAMPLE_LOBE_PDFS: {
            MaterialSample m;
            memset(&m, 0, sizeof(m));
            m.Metalness = a[0];
            m.Transmission = a[1];
            const LobePdfs p = sampleLobePdfs(&m, a[2]);
            o[0] = p.Diffuse; o[1] = p.Glossy; o[2] = p.Metallic; o[3] = p.Transmissive;
            break;
        }
        case PTX_FN_EVALUATE_BSDF: {
            const MaterialSample m = unpackMaterial(a);
            float pdf;
            const v3 r = evaluateBSDF(&m, V3(a[8], a[9], a[10]), V3(a[11], a[12], a[13]), &pdf);
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf;
            break;
        }
        case PTX_FN_SAMPLE_BSDF: {
            const MaterialSample m = unpackMaterial(a);
            uint32_t rng = f2u(a[11]);
            const BSDFSample r = sampleBSDF(&m, V3(a[8], a[9], a[10]), &rng);
            o[0] = r.Direction.x; o[1] = r.Direction.y; o[2] = r.Direction.z; o[3] = r.Pdf;
            o[4] = r.Color.x; o[5] = r.Color.y; o[6] = r.Color.z; o[7] = u2f(rng);
            break;
        }
        case PTX_FN_RNG: {
            uint32_t st = initRng(f2u(a[0]), f2u(a[1]), f2u(a[2]), f2u(a[3]));
            o[0] = u2f(st);
            for (int k = 0; k < 4; k++)
                o[1 + k] = rnd(&st);
            break;
        }
        case PTX_FN_DISK: {
            v2 u = { a[0], a[1] };
            const v2 d = sampleUniformDiskConcentric(u);
            o[0] = d.x; o[1] = d.y;
            break;
        }
        case PTX_FN_COS_HEMISPHERE: {
            v2 u = { a[0], a[1] };
            const v3 d = sampleCosineHemisphere(u);
            o[0] = d.x; o[1] = d.y; o[2] = d.z;
            break;
        }
        case PTX_FN_TANGENT_SPACE: {
            const m3 m = computeTangentSpace(V3(a[0], a[1], a[2]));
            o[0] = m.c0.x; o[1] = m.c0.y; o[2] = m.c0.z;
            o[3] = m.c1.x; o[4] = m.c1.y; o[5] = m.c1.z;
            o[6] = m.c2.x; o[7] = m.c2.y; o[8] = m.c2.z;
            break;
        }
        case PTX_FN_OFFSET_SELF_INTERSECTION: {
            const v3 r = offsetRayOriginSelfIntersection(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]));
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case PTX_FN_PRIMARY_RAY: {
            v2 u = { a[4], a[5] };
            Ray rx, ry;
            const Ray r = constructPrimaryRayD(f2u(a[0]), f2u(a[1]), f2u(a[2]), f2u(a[3]), &a[6], &a[22], u, &rx, &ry);
            const Ray *rs[3] = { &r, &rx, &ry };
            for (int k = 0; k < 3; k++)
            {
                o[6 * k] = rs[k]->Origin.x; o[6 * k + 1] = rs[k]->Origin.y; o[6 * k + 2] = rs[k]->Origin.z;
                o[6 * k + 3] = rs[k]->Direction.x; o[6 * k + 4] = rs[k]->Direction.y; o[6 * k + 5] = rs[k]->Direction.z;
            }
            break;
        }
        case PTX_FN_SINCOS: pto_sincosf(a[0], &o[0], &o[1]); break;
        case PTX_FN_POW: o[0] = pto_powf(a[0], a[1]); break;
        case PTX_FN_SAMPLE_LIGHT: {
            PtxLightsUbo ubo;
            memset(&ubo, 0, sizeof(ubo));
            ubo.LightCount = f2u(a[6]);
            memcpy(ubo.Directional.Color, &a[7], 12);
            memcpy(ubo.Directional.Direction, &a[10], 12);
            for (int k = 0; k < 2; k++)
            {
                memcpy(ubo.Lights[k].Color, &a[13 + 9 * k], 12);
                memcpy(ubo.Lights[k].Position, &a[16 + 9 * k], 12);
                ubo.Lights[k].AttenuationConstant = a[19 + 9 * k];
                ubo.Lights[k].AttenuationLinear = a[20 + 9 * k];
                ubo.Lights[k].AttenuationQuadratic = a[21 + 9 * k];
            }
            float pdf;
            const LightSample l = sampleLight(&ubo, V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), &pdf);
            o[0] = l.Direction.x; o[1] = l.Direction.y; o[2] = l.Direction.z; o[3] = l.Distance;
            o[4] = l.Color.x; o[5] = l.Color.y; o[6] = l.Color.z; o[7] = l.Attenuation; o[8] = pdf;
            break;
        }
        case PTX_FN_SHADOW_TERMINATOR: {
            Vtx vx, q0, q1, q2;
            memset(&vx, 0, sizeof(vx)); memset(&q0, 0, sizeof(q0)); memset(&q1, 0, sizeof(q1)); memset(&q2, 0, sizeof(q2));
            vx.Position = V3(a[0], a[1], a[2]);
            q0.Position = V3(a[3], a[4], a[5]); q0.Normal = V3(a[6], a[7], a[8]);
            q1.Position = V3(a[9], a[10], a[11]); q1.Normal = V3(a[12], a[13], a[14]);
            q2.Position = V3(a[15], a[16], a[17]); q2.Normal = V3(a[18], a[19], a[20]);
            const v3 r = offsetRayOriginShadowTerminator(&vx, &q0, &q1, &q2, V3(a[21], a[22], a[23]), a[24] != 0.0f);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case PTX_FN_PRIMARY_RAY_LENS: {
            v2 u = { a[4], a[5] }, u2 = { a[6], a[7] };
            Ray rx, ry;
            const Ray r = constructPrimaryRayLensD(f2u(a[0]), f2u(a[1]), f2u(a[2]), f2u(a[3]), &a[10], &a[26], u, u2, a[8], a[9], &rx, &ry);
            const Ray *rs[3] = { &r, &rx, &ry };
            for (int k = 0; k < 3; k++)
            {
                o[6 * k] = rs[k]->Origin.x; o[6 * k + 1] = rs[k]->Origin.y; o[6 * k + 2] = rs[k]->Origin.z;
                o[6 * k + 3] = rs[k]->Direction.x; o[6 * k + 4] = rs[k]->Direction.y; o[6 * k + 5] = rs[k]->Direction.z;
            }
            break;
        }
        case PTX_FN_DPN_DUV: {
            v3 P[3], N[3];
            v2 UV[3];
            for (int k = 0; k < 3; k++)
            {
                P[k] = V3(a[8 * k], a[8 * k + 1], a[8 * k + 2]);
                N[k] = V3(a[8 * k + 3], a[8 * k + 4], a[8 * k + 5]);
                UV[k].x = a[8 * k + 6];
                UV[k].y = a[8 * k + 7];
            }
            v3 r[4];
            computeDpnDuv(P, N, UV, V3(a[24], a[25], a[26]), V3(a[27], a[28], a[29]), &r[0], &r[1], &r[2], &r[3]);
            for (int k = 0; k < 4; k++) { o[3 * k] = r[k].x; o[3 * k + 1] = r[k].y; o[3 * k + 2] = r[k].z; }
            break;
        }
        case PTX_FN_DP_DXY: {
            v3 dx, dy;
            computeDpDxy(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]), V3(a[9], a[10], a[11]), V3(a[12], a[13], a[14]),
                         V3(a[15], a[16], a[17]), V3(a[18], a[19], a[20]), V3(a[21], a[22], a[23]), &dx, &dy);
            o[0] = dx.x; o[1] = dx.y; o[2] = dx.z; o[3] = dy.x; o[4] = dy.y; o[5] = dy.z;
            break;
        }
        case PTX_FN_DERIVATIVES: {
            const v4 r = computeDerivatives(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]), V3(a[9], a[10], a[11]));
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
            break;
        }
        case PTX_FN_REFLECTED_DIFFERENTIALS:
        case PTX_FN_REFRACTED_DIFFERENTIALS: {
            v4 dv = { a[0], a[1], a[2], a[3] };
            DiffRays r;
            r.rxOrigin = V3(a[22], a[23], a[24]); r.rxDirection = V3(a[25], a[26], a[27]);
            r.ryOrigin = V3(a[28], a[29], a[30]); r.ryDirection = V3(a[31], a[32], a[33]);
            if (fn == PTX_FN_REFLECTED_DIFFERENTIALS)
                computeReflectedDifferentialRays(dv, V3(a[4], a[5], a[6]), V3(a[7], a[8], a[9]), V3(a[10], a[11], a[12]), V3(a[13], a[14], a[15]),
                                                 V3(a[16], a[17], a[18]), V3(a[19], a[20], a[21]), &r);
            else
                computeRefractedDifferentialRays(dv, V3(a[4], a[5], a[6]), V3(a[7], a[8], a[9]), V3(a[10], a[11], a[12]), V3(a[13], a[14], a[15]),
                                                 V3(a[16], a[17], a[18]), V3(a[19], a[20], a[21]), a[34], &r);
            o[0] = r.rxOrigin.x; o[1] = r.rxOrigin.y; o[2] = r.rxOrigin.z; o[3] = r.rxDirection.x; o[4] = r.rxDirection.y; o[5] = r.rxDirection.z;
            o[6] = r.ryOrigin.x; o[7] = r.ryOrigin.y; o[8] = r.ryOrigin.z; o[9] = r.ryDirection.x; o[10] = r.ryDirection.y; o[11] = r.ryDirection.z;
            break;
        }
        case PTX_FN_SKYBOX_TEXCOORDS: {
            const v2 uv = missSkyboxTexCoords(V3(a[0], a[1], a[2]));
            o[0] = uv.x; o[1] = uv.y;
            break;
        }
        case PTX_FN_HDR_TO_LDR: {
            const v3 r = hdrToLdr(V3(a[0], a[1], a[2]));
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case PTX_FN_ATAN_ASIN:
            o[0] = pto_atan2f(a[0], a[1]);
            o[1] = pto_asinf(a[0]);
            break;
        case PTX_FN_DIVIDE:
            o[0] = pto_rcp(a[1]);
            o[1] = pto_div(a[0], a[1]);
            break;
        case PTX_FN_SQRT: o[0] = sqrtf(a[0]); break;
        case PTX_FN_RSQ: o[0] = pto_rsq(a[0]); break;
        case PTX_FN_COMPUTE_LOD: {
            v4 dv = { a[0], a[1], a[2], a[3] };
            o[0] = computeLod(dv);
            break;
        }
        case PTX_FN_SAMPLE_MATERIAL: {
            uint32_t hdr[3];
            memcpy(hdr, a, sizeof(hdr));
            union { PtxMetallicRoughnessMaterial mr; PtxSpecularGlossinessMaterial sg; PtxPhongMaterial ph; } rec;
            memcpy(&rec, a + 3, 96);
            MaterialTexels t;
            memcpy(&t, a + 27, sizeof(t));
            MaterialSample m;
            if (hdr[0] == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS)
                m = sampleMaterialMR(&rec.mr, &t, hdr[1] != 0u);
            else if (hdr[0] == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS)
                m = sampleMaterialSG(&rec.sg, &t, hdr[1] != 0u);
            else if (hdr[0] == PTX_MATERIAL_TYPE_PHONG)
                m = sampleMaterialPhong(&rec.ph, &t, hdr[1] != 0u);
            else
                m = unknownMaterial();
            if (hdr[2])
                m.Normal.y *= -1;
            o[0] = m.EmissiveColor.x; o[1] = m.EmissiveColor.y; o[2] = m.EmissiveColor.z; o[3] = m.Color.x; o[4] = m.Color.y; o[5] = m.Color.z;
            o[6] = m.Normal.x; o[7] = m.Normal.y; o[8] = m.Normal.z; o[9] = m.Roughness; o[10] = m.Metalness; o[11] = m.Transmission; o[12] = m.Eta;
            o[13] = m.AttenuationColor.x; o[14] = m.AttenuationColor.y; o[15] = m.AttenuationColor.z; o[16] = m.AttenuationDistance;
            break;
        }
        default: return 1;
        }
    }
    return 0;
}

/* raygen.rgen:36-118 with scripted trace calls.  One case = 44 + 12 x 23 words:
 *   [0] pixel x, [1] pixel y, [2] width, [3] height, [4] TotalSamples, [5] SampleCount, [6] BounceCount, [7] LensRadius,
 *   [8] FocalDistance, [9..24] ViewInverse, [25..40] ProjInverse (column-major), [41..43] the pixel's previous sum,
 *   then 12 records { Position, Direction, Emissive, Bsdf, Pdf, DirectLight, DirectLightPdf, LightDirection,
 *   LightDistance, draws, occluded }.
 * Output, 7 words: the stored pixel (rgba), primary trace calls, shadow trace calls, hash of every traced ray. */
int pto_test_raygen(const uint32_t *in, uint32_t *out, uint32_t n)
{
    if (!in || !out)
        return 1;
    const uint32_t stride = 44 + PTO_SCRIPT_RECORDS * PTO_SCRIPT_RECORD_WORDS;
    for (uint32_t i = 0; i < n; i++)
    {
        const uint32_t *a = &in[(size_t)i * stride];
        PtxRaygenUniformData U;
        memset(&U, 0, sizeof(U));
        for (int k = 0; k < 16; k++)
        {
            U.ViewInverse[k] = u2f(a[9 + k]);
            U.ProjInverse[k] = u2f(a[25 + k]);
        }
        U.TotalSamples = a[4];
        U.SampleCount = a[5];
        U.BounceCount = a[6];
        U.LensRadius = u2f(a[7]);
        U.FocalDistance = u2f(a[8]);
        const uint32_t W = a[2], H = a[3];
        if (a[0] >= W || a[1] >= H || (uint64_t)W * H > (1u << 24))
            return 1;
        TraceScript script = { &a[44], 0u, 0u, 2166136261u };
        PtoStats st;
        memset(&st, 0, sizeof(st));
        /* raygenPixel addresses the image by pixel: give it a one-pixel window */
        float *accum = (float *)calloc((size_t)W * H, 16);
        if (!accum)
            return 1;
        float *p = &accum[((size_t)a[1] * W + a[0]) * 4];
        p[0] = u2f(a[41]); p[1] = u2f(a[42]); p[2] = u2f(a[43]);
        g_script = &script;
        raygenPixel(NULL, &U, NULL, a[0], a[1], W, H, accum, 0, &st);
        g_script = NULL;
        uint32_t *o = &out[(size_t)i * 7];
        for (int k = 0; k < 4; k++)
            o[k] = f2u(p[k]);
        o[4] = script.calls0;
        o[5] = script.calls1;
        o[6] = script.hash;
        free(accum);
    }
    return 0;
}

/* closestHit.rchit:52-161 on triangle 0 of the scene.  One case = 28 words: ray origin, direction, hit distance, the two
 * hit attributes (barycentrics of vertex 1 and 2), then the payload as raygen / the any-hit stage leave it: RngState,
 * MaxRoughness, DirectLightPdf (decal distance or -1), LightDirection (decal colour), LightDistance (decal alpha) and the
 * two differential rays (origin, direction each).  Output, 35 words: Position, Direction, MaxRoughness, Bsdf, Pdf,
 * Emissive, RngState, DirectLight, DirectLightPdf, LightDirection, LightDistance, the differential rays.
 * Checked against tests/golden/golden_stage_*.json (the reference's closestHit.rchit text, tools/gen_golden.py). */
int pto_test_closest_hit(const PtoScene *s, const PtxLightsUbo *lights, const uint32_t *in, uint32_t *out, uint32_t n)
{
    if (!s || !lights || !in || !out || s->triCount == 0)
        return 1;
    for (uint32_t i = 0; i < n; i++)
    {
        const uint32_t *a = &in[(size_t)i * 28];
        const v3 origin = V3(u2f(a[0]), u2f(a[1]), u2f(a[2])), dir = V3(u2f(a[3]), u2f(a[4]), u2f(a[5]));
        PtoHit hit;
        memset(&hit, 0, sizeof(hit));
        hit.t = u2f(a[6]);
        hit.u = u2f(a[7]);
        hit.v = u2f(a[8]);
        hit.tri = 0u;
        Payload p;
        memset(&p, 0, sizeof(p));
        p.RngState = a[9];
        p.MaxRoughness = u2f(a[10]);
        p.DirectLightPdf = u2f(a[11]);
        p.LightDirection = V3(u2f(a[12]), u2f(a[13]), u2f(a[14]));
        p.LightDistance = u2f(a[15]);
        p.diff.rxOrigin = V3(u2f(a[16]), u2f(a[17]), u2f(a[18]));
        p.diff.rxDirection = V3(u2f(a[19]), u2f(a[20]), u2f(a[21]));
        p.diff.ryOrigin = V3(u2f(a[22]), u2f(a[23]), u2f(a[24]));
        p.diff.ryDirection = V3(u2f(a[25]), u2f(a[26]), u2f(a[27]));
        closestHit(s, lights, origin, dir, &hit, &p);
        const float o[35] = { p.Position.x, p.Position.y, p.Position.z, p.Direction.x, p.Direction.y, p.Direction.z, p.MaxRoughness,
                              p.Bsdf.x, p.Bsdf.y, p.Bsdf.z, p.Pdf, p.Emissive.x, p.Emissive.y, p.Emissive.z, 0.0f,
                              p.DirectLight.x, p.DirectLight.y, p.DirectLight.z, p.DirectLightPdf,
                              p.LightDirection.x, p.LightDirection.y, p.LightDirection.z, p.LightDistance,
                              p.diff.rxOrigin.x, p.diff.rxOrigin.y, p.diff.rxOrigin.z, p.diff.rxDirection.x, p.diff.rxDirection.y, p.diff.rxDirection.z,
                              p.diff.ryOrigin.x, p.diff.ryOrigin.y, p.diff.ryOrigin.z, p.diff.ryDirection.x, p.diff.ryDirection.y, p.diff.ryDirection.z };
        for (int k = 0; k < 35; k++)
            out[(size_t)i * 35 + (size_t)k] = f2u(o[k]);
        out[(size_t)i * 35 + 14] = p.RngState;
    }
    return 0;
}

/* anyhit.rahit:36-64 and occlusionAnyhit.rahit:35-53 for one candidate hit on triangle 0 (which must belong to a
 * non-opaque geometry).  One case = 8 words: hit attributes u, v, the candidate's distance, then the decal the payload
 * holds: DirectLightPdf (distance or -1), LightDirection (colour), LightDistance (alpha).  Output, 7 words: ignored by the
 * closest-hit query (0 / 1), ignored by the shadow query, and the payload's decal fields afterwards. */
int pto_test_any_hit(const PtoScene *s, const uint32_t *in, uint32_t *out, uint32_t n)
{
    if (!s || !in || !out || s->triCount == 0)
        return 1;
    for (uint32_t i = 0; i < n; i++)
    {
        const uint32_t *a = &in[(size_t)i * 8];
        Decal d;
        d.dist = u2f(a[3]);
        d.color = V3(u2f(a[4]), u2f(a[5]), u2f(a[6]));
        d.alpha = u2f(a[7]);
        d.tri = 0u; /* not above the candidate's id: an equal distance keeps the payload's decal, as `dist < payload.DirectLightPdf` does */
        uint32_t *o = &out[(size_t)i * 7];
        o[0] = (uint32_t)anyHitIgnores(s, 0u, u2f(a[2]), u2f(a[0]), u2f(a[1]), &d);
        o[1] = (uint32_t)occlusionIgnores(s, 0u, u2f(a[0]), u2f(a[1]));
        o[2] = f2u(d.dist);
        o[3] = f2u(d.color.x); o[4] = f2u(d.color.y); o[5] = f2u(d.color.z);
        o[6] = f2u(d.alpha);
    }
    return 0;
}

/* miss.rmiss:16-39 for given ray directions (3 words each).  Output, 4 words: payload.Emissive, payload.Pdf. */
int pto_test_miss(const PtoScene *s, const uint32_t *in, uint32_t *out, uint32_t n)
{
    if (!s || !in || !out)
        return 1;
    for (uint32_t i = 0; i < n; i++)
    {
        Payload p;
        memset(&p, 0, sizeof(p));
        p.Pdf = 7.0f;
        missShader(s, V3(u2f(in[i * 3]), u2f(in[i * 3 + 1]), u2f(in[i * 3 + 2])), &p);
        out[i * 4] = f2u(p.Emissive.x); out[i * 4 + 1] = f2u(p.Emissive.y); out[i * 4 + 2] = f2u(p.Emissive.z);
        out[i * 4 + 3] = f2u(p.Pdf);
    }
    return 0;
}
