// pt_device.hpp -- device-side shading arithmetic of the HIP path tracer (gfx950).
//
// Hand-written HIP restatement of the reference's GLSL headers
//   Path-Tracing/Shaders/{common,shading,bsdf,sampling,ray,material}.glsl
// and of closestHit.rchit / miss.rmiss, used by both backends (wavefront kernels and
// the bring-up megakernel) in pt_kernels.hip.  Each function cites the GLSL it follows.
//
// Arithmetic conventions (fixed so results are reproducible bit for bit; GLSL leaves
// them implementation-defined): IEEE binary32 round-to-nearest, no contraction
// (compiled with -ffp-contract=off), fma only where the GLSL writes fma();
// dot = (x*x' + y*y') + z*z'; normalize(v) = v * rsq(dot(v,v)), rsq the correctly rounded 1/sqrt; mat3*vec3 =
// (c0*x + c1*y) + c2*z; inverse(mat3) by cofactors * (1/det); min/max as the GLSL
// select forms (NaN behaviour of "y < x ? y : x"); pow(x,2) = x*x, pow(x,5) = x2*x2*x;
// sin/cos/pow by the fixed polynomial kernels below (no ocml calls: their results are
// not specified to the bit).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptx.h"

#define PT_DEV __device__ __forceinline__
#define PT_PI 3.14159265359f // common.glsl:3

namespace ptd
{

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct mat3 { f3 c0, c1, c2; }; // columns

PT_DEV f2 F2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
PT_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 F3s(float s) { return F3(s, s, s); }
PT_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
// The specified division (arithmetic conventions): a / b := a * rcp_(b), rcp_ = the correctly rounded reciprocal
// on [2^-126, 2^126], +-inf below (zero and denormal divisors), +-0 above.  v_rcp_f32 (1 ULP, flushes denormals both ways) + one FMA
// Newton step = that definition on all 2^32 inputs (profiles/r05_rcp_sqrt_exhaustive.txt, test_specified_reciprocal_on_all_inputs);
// the select keeps the seed where the step would produce NaN (x = 0, inf, denormal).  5 VALU; hipcc's IEEE quotient is 11.
PT_DEV float rcp_(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    const float r1 = __builtin_fmaf(r0, e, r0);
    return __builtin_fabsf(e) < 1.0f ? r1 : r0;
}
PT_DEV float div_(float a, float b) { return a * rcp_(b); }
PT_DEV f3 operator/(f3 a, float s) { const float r = rcp_(s); return F3(a.x * r, a.y * r, a.z * r); }
PT_DEV f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }

PT_DEV float fmin_(float a, float b) { return (b < a) ? b : a; } // GLSL min
PT_DEV float fmax_(float a, float b) { return (a < b) ? b : a; } // GLSL max
PT_DEV float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
PT_DEV float abs_(float x) { return __builtin_fabsf(x); }
// Correctly rounded square root (IEEE): hipcc's expansion, 16 VALU -- v_sqrt_f32, the two-neighbour residual test, and a
// power-of-two scaling for |x| < 2^-96 where the residuals would underflow.  A hand-written form without the scaling behind a
// wave-uniform branch (10 VALU on the fast path; equal on all 2^32 inputs: tools/experiments/rcp_sqrt_exhaustive.hip) was measured
// and is NOT used: k_shade alone 0.555 -> 0.586 ms, chess_like 2,656 -> 2,636 Msamples/s on one box (profiles/r05_ab_sqrt.txt) --
// 62 extra branch sites cost the scheduler more than the six instructions they save.  -DPTX_HANDWRITTEN_SQRT builds it.
PT_DEV float sqrt_(float x)
{
#ifndef PTX_HANDWRITTEN_SQRT
    return __builtin_sqrtf(x);
#else
    if (__builtin_amdgcn_ballot_w64(__builtin_fabsf(x) < 1.262177448e-29f) != 0) // 2^-96
        return __builtin_sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = rm <= 0.0f ? sm : s;
    r = rp > 0.0f ? sp : r;
    return r;
#endif
}

// The specified reciprocal square root (arithmetic conventions): RN(1 / sqrt(x)) for positive normal x,
// the hardware's own answer elsewhere (+-inf for +-0 and denormals, which v_rsq_f32 flushes; +0 for +inf; NaN for negative numbers).
// v_rsq_f32 (1 ULP) + ONE Newton step whose residual 1 - x y^2 is taken in two pieces (h + l = x y exactly) and whose correction
// carries the second-order term: y (1 + e / 2 + 3 e^2 / 8).  For every seed within 1 ULP of the true value this rounds to the
// correctly rounded result on all 2^24 (mantissa, exponent parity) classes (proof by enumeration on the CPU, in the tests) and it equals the
// definition on all 2^32 inputs on the hardware (tools/experiments/rcp_sqrt_exhaustive.hip).  8 VALU + the class test and select;
// rcp_(sqrt_(x)), which normalize() was until round 6, is 21.
PT_DEV float rsq_(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    const float h = x * y0;
    float e = __builtin_fmaf(-h, y0, 1.0f);
    const float nl = __builtin_fmaf(-x, y0, h); // -(x y0 - h): the low half of the product, negated (fma(a, b, c) = -fma(-a, b, -c) exactly)
    e = __builtin_fmaf(nl, y0, e);
    const float p = __builtin_fmaf(0.375f, e, 0.5f);
    const float y1 = __builtin_fmaf(y0 * e, p, y0);
    return __builtin_amdgcn_classf(x, 0x100) ? y1 : y0; // 0x100: positive normal
}

PT_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_DEV f3 cross(f3 a, f3 b) { return F3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
PT_DEV float length(f3 a) { return sqrt_(dot(a, a)); }
PT_DEV f3 normalize(f3 a) { return a * rsq_(dot(a, a)); }
PT_DEV f3 reflect(f3 I, f3 N) { return I - N * (2.0f * dot(N, I)); }
PT_DEV f3 refract(f3 I, f3 N, float eta)
{
    const float d = dot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f)
        return F3s(0.0f);
    return I * eta - N * (eta * d + sqrt_(k));
}
PT_DEV f3 mix(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }

PT_DEV f3 mul(const mat3 &m, f3 v)
{
    return F3((m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z, (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z,
              (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z);
}

PT_DEV mat3 inverse(const mat3 &m)
{
    const float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z;
    const float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z;
    const float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z;
    const float det = (m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02)) + m20 * (m01 * m12 - m11 * m02);
    const float id = rcp_(det);
    mat3 r;
    r.c0.x = (m11 * m22 - m21 * m12) * id;
    r.c1.x = -(m10 * m22 - m20 * m12) * id;
    r.c2.x = (m10 * m21 - m20 * m11) * id;
    r.c0.y = -(m01 * m22 - m21 * m02) * id;
    r.c1.y = (m00 * m22 - m20 * m02) * id;
    r.c2.y = -(m00 * m21 - m20 * m01) * id;
    r.c0.z = (m01 * m12 - m11 * m02) * id;
    r.c1.z = -(m00 * m12 - m10 * m02) * id;
    r.c2.z = (m00 * m11 - m10 * m01) * id;
    return r;
}

// mat4 (glm column-major [c*4+r]) * vec4
PT_DEV f4 mul4(const float *m, float x, float y, float z, float w)
{
    f4 r;
    r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
    r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
    r.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
    return r;
}

// ---- fixed transcendental kernels -------------------------------------------------

// sin & cos on roughly [-pi/4, 2pi]: 3-piece Cody-Waite reduction by pi/2, then the
// degree-7 / degree-8 minimax polynomials.
PT_DEV void sincos_(float x, float &s, float &c)
{
    const float fk = __builtin_floorf(x * 0.636619772f + 0.5f);
    const int k = (int)fk;
    float r = x - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188216e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    const int q = k & 3;
    s = (q == 0) ? ps : (q == 1) ? pc : (q == 2) ? -ps : -pc;
    c = (q == 0) ? pc : (q == 1) ? -ps : (q == 2) ? -pc : ps;
}

PT_DEV double log2_(double x) // positive, finite, normal
{
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)bits);
    if (m > 1.4142135623730951)
    {
        m = m * 0.5;
        e = e + 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double ln = 2.0 * s * p;
    return (double)e + ln * 1.4426950408889634;
}

PT_DEV double exp2_(double t) // |t| <= 300
{
    const double k = __builtin_floor(t + 0.5);
    const double r = (t - k) * 0.6931471805599453;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const unsigned long long bits = (unsigned long long)((long long)k + 1023) << 52;
    return p * __longlong_as_double((long long)bits);
}

PT_DEV float pow_(float x, float y) // x >= 0
{
    if (y == 0.0f || x == 1.0f)
        return 1.0f;
    if (x != x || y != y || x < 0.0f)
        return __uint_as_float(0x7fc00000u);
    if (x == 0.0f)
        return y > 0.0f ? 0.0f : __uint_as_float(0x7f800000u);
    if (x == __uint_as_float(0x7f800000u))
        return y > 0.0f ? __uint_as_float(0x7f800000u) : 0.0f;
    double t = (double)y * log2_((double)x);
    if (t > 300.0)
        t = 300.0;
    if (t < -300.0)
        t = -300.0;
    return (float)exp2_(t);
}

// atan(y, x) over the full circle and asin(x): fixed double-precision kernels, the same operation
// sequence as the CPU check library (|t| <= tan(pi/8) after the reductions, odd Taylor series to t^23).
PT_DEV float atan2_(float yf, float xf)
{
    const double y = (double)yf, x = (double)xf;
    const double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    const double hi = ax < ay ? ay : ax, lo = ax < ay ? ax : ay;
    if (hi == 0.0)
        return 0.0f;
    const double a = lo / hi;
    const int reduce = a > 0.4142135623730951;
    const double t = reduce ? (a - 1.0) / (a + 1.0) : a;
    const double z = t * t;
    double p = -1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z - 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z - 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z - 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z - 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z - 1.0 / 3.0;
    p = p * z + 1.0;
    double r = t * p;
    if (reduce)
        r = 0.7853981633974483 + r;
    if (ay > ax)
        r = 1.5707963267948966 - r;
    if (x < 0.0)
        r = 3.141592653589793 - r;
    if (y < 0.0)
        r = -r;
    return (float)r;
}

PT_DEV float asin_(float xf)
{
    double x = (double)xf;
    if (x > 1.0)
        x = 1.0;
    if (x < -1.0)
        x = -1.0;
    const double c = __builtin_sqrt((1.0 - x) * (1.0 + x));
    const double ax = c, ay = x < 0.0 ? -x : x; // atan2(x, c) with c >= 0, in double throughout
    const double hi = ax < ay ? ay : ax, lo = ax < ay ? ax : ay;
    if (!(hi > 0.0))
        return xf != xf ? xf : 0.0f;
    const double a = lo / hi;
    const int reduce = a > 0.4142135623730951;
    const double t = reduce ? (a - 1.0) / (a + 1.0) : a;
    const double z = t * t;
    double p = -1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z - 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z - 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z - 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z - 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z - 1.0 / 3.0;
    p = p * z + 1.0;
    double r = t * p;
    if (reduce)
        r = 0.7853981633974483 + r;
    if (ay > ax)
        r = 1.5707963267948966 - r;
    if (x < 0.0)
        r = -r;
    return (float)r;
}

// ---- common.glsl --------------------------------------------------------------------

PT_DEV float maxComponent(f3 rgb) { return fmax_(rgb.x, fmax_(rgb.y, rgb.z)); } // :12-15

PT_DEV uint32_t jenkinsHash(uint32_t x) // :133-141
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}

PT_DEV uint32_t initRng(uint32_t px, uint32_t py, uint32_t resX, uint32_t frame) // :143-147
{
    const float d = (float)px * 1.0f + (float)py * (float)resX;
    const uint32_t rngState = (uint32_t)d ^ jenkinsHash(frame);
    return jenkinsHash(rngState);
}

PT_DEV float rnd(uint32_t &rngState) // :149-165
{
    rngState ^= rngState << 13;
    rngState ^= rngState >> 17;
    rngState ^= rngState << 5;
    return __uint_as_float(0x3f800000u | (rngState >> 9)) - 1.0f;
}

PT_DEV f2 sampleUniformDiskConcentric(f2 u) // :168-184
{
    f2 offset;
    offset.x = 2.0f * u.x - 1.0f;
    offset.y = 2.0f * u.y - 1.0f;
    f2 r;
    r.x = 0.0f;
    r.y = 0.0f;
    if (offset.x == 0.0f && offset.y == 0.0f)
        return r;
    float s, c;
    if (abs_(offset.x) > abs_(offset.y))
    {
        const float theta = (div_(PT_PI, 4)) * (div_(offset.y, offset.x));
        sincos_(theta, s, c);
        r.x = offset.x * c;
        r.y = offset.x * s;
    }
    else
    {
        const float theta = div_(PT_PI, 2) - (div_(PT_PI, 4)) * (div_(offset.x, offset.y));
        sincos_(theta, s, c);
        r.x = offset.y * c;
        r.y = offset.y * s;
    }
    return r;
}

PT_DEV f3 sampleCosineHemisphere(f2 u) // :186-191
{
    const f2 d = sampleUniformDiskConcentric(u);
    const float z = sqrt_(1 - d.x * d.x - d.y * d.y);
    return F3(d.x, d.y, z);
}

PT_DEV mat3 computeTangentSpace(f3 normal) // :193-202
{
    const f3 t1 = cross(normal, F3(1.0f, 0.0f, 0.0f));
    const f3 t2 = cross(normal, F3(0.0f, 1.0f, 0.0f));
    const f3 tangent = length(t1) > length(t2) ? t1 : t2;
    const f3 bitangent = cross(normal, tangent);
    mat3 m;
    m.c0 = normalize(tangent);
    m.c1 = normalize(bitangent);
    m.c2 = normal;
    return m;
}

// ---- shading.glsl ---------------------------------------------------------------------

PT_DEV float GGXDistribution(f3 H, float alpha) // :3-14 (D clamped to <= 1: kept quirk)
{
    const float Hx2 = H.x * H.x;
    const float Hy2 = H.y * H.y;
    const float Hz2 = H.z * H.z;
    const float alpha2 = alpha * alpha;
    const float b = div_(Hx2, alpha2) + div_(Hy2, alpha2) + Hz2;
    const float denom = PT_PI * alpha2 * (b * b);
    return div_(1.0f, fmax_(denom, 1.0f));
}

PT_DEV float Lambda(f3 V, float alpha) // :16-27
{
    const float Vx2 = V.x * V.x;
    const float Vy2 = V.y * V.y;
    const float Vz2 = abs_(V.z) * abs_(V.z);
    const float alpha2 = alpha * alpha;
    const float nom = sqrt_(1.0f + div_(alpha2 * Vx2 + alpha2 * Vy2, Vz2)) - 1.0f;
    return div_(nom, 2.0f);
}

PT_DEV float GGXSmith(f3 V, float alpha) { return div_(1.0f, 1.0f + Lambda(V, alpha)); } // :29-32

PT_DEV float DielectricFresnel(float VdotH, float eta) // :34-48
{
    const float cosThetaI = VdotH;
    const float sinThetaT2 = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaT2 > 1.0f)
        return 1.0f;
    const float cosThetaT = sqrt_(fmax_(1.0f - sinThetaT2, 0.0f));
    const float rs = div_(eta * cosThetaT - cosThetaI, eta * cosThetaT + cosThetaI);
    const float rp = div_(eta * cosThetaI - cosThetaT, eta * cosThetaI + cosThetaT);
    return div_(rs * rs + rp * rp, 2.0f);
}

PT_DEV float SchlickFresnel(float VdotH) // :50-53
{
    const float x = clamp_(1.0f - VdotH, 0.0f, 1.0f);
    const float x2 = x * x;
    return x2 * x2 * x;
}

PT_DEV f3 EvaluateReflection(f3 V, f3 L, f3 F, float alpha, float &pdf) // :56-77
{
    if (L.z < 0.00001f)
    {
        pdf = 0.0f;
        return F3s(0.0f);
    }
    const f3 H = normalize(V + L);
    const float VdotH = dot(V, H);
    const float D = GGXDistribution(H, alpha);
    const float Gv = GGXSmith(V, alpha);
    const float Gl = GGXSmith(L, alpha);
    const float G = Gv * Gl;
    const float Dv = div_(Gv * fmax_(VdotH, 0.0f) * D, V.z);
    pdf = div_(Dv, 4.0f * VdotH);
    return (F * (D * G)) / (4.0f * V.z);
}

PT_DEV f3 EvaluateRefraction(f3 V, f3 L, f3 F, float alpha, float eta, float &pdf) // :80-108
{
    if (L.z > -0.00001f)
    {
        pdf = 0.0f;
        return F3s(0.0f);
    }
    f3 H = normalize(V * eta + L);
    if (H.z < 0.0f)
        H = -H;
    const float VdotH = dot(V, H);
    const float LdotH = dot(L, H);
    const float D = GGXDistribution(H, alpha);
    const float Gv = GGXSmith(V, alpha);
    const float Gl = GGXSmith(L, alpha);
    const float G = Gv * Gl;
    const float Dv = div_(Gv * abs_(VdotH) * D, V.z);
    const float denominator = LdotH + eta * VdotH;
    const float jacobian = div_((eta * eta) * abs_(LdotH), denominator * denominator);
    pdf = Dv * jacobian;
    return ((F * (D * G)) * (div_(abs_(VdotH), abs_(V.z)))) * jacobian;
}

PT_DEV f3 SampleGGX(f2 u, f3 V, float alpha) // :111-129
{
    const f3 Vh = normalize(F3(alpha * V.x, alpha * V.y, abs_(V.z)));
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    const f3 T1 = lensq > 0 ? F3(-Vh.y, Vh.x, 0) * rsq_(lensq) : F3(1, 0, 0); // inversesqrt
    const f3 T2 = cross(Vh, T1);
    const float r = sqrt_(u.x);
    const float phi = 2.0f * PT_PI * u.y;
    float sn, cs;
    sincos_(phi, sn, cs);
    const float t1 = r * cs;
    float t2 = r * sn;
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrt_(1.0f - t1 * t1) + s * t2;
    const f3 Nh = (T1 * t1 + T2 * t2) + Vh * sqrt_(fmax_(0.0f, 1.0f - t1 * t1 - t2 * t2));
    return normalize(F3(alpha * Nh.x, alpha * Nh.y, fmax_(0.0f, Nh.z)));
}

// ---- bsdf.glsl --------------------------------------------------------------------------

struct MaterialSample // ShaderRendererTypes.incl:129-140
{
    f3 EmissiveColor;
    f3 Color;
    f3 Normal;
    float Roughness;
    float Metalness;
    float Transmission;
    float Eta;
    f3 AttenuationColor;
    float AttenuationDistance;
};

struct BSDFSample
{
    f3 Direction;
    float Pdf;
    f3 Color;
};

PT_DEV f3 evaluateBSDF(const MaterialSample &m, f3 V, f3 L, float &outPdf) // :72-103
{
    const bool isReflection = L.z > 0.0f;
    const f3 H = isReflection ? normalize(V + L) : normalize(V * m.Eta + L);
    const float FD = DielectricFresnel(abs_(dot(V, H)), m.Eta);
    // sampleLobePdfs, :62-70
    const float pDiffuse = (1.0f - m.Metalness) * (1.0f - FD) * (1.0f - m.Transmission);
    const float pGlossy = (1.0f - m.Metalness) * FD;
    const float pMetallic = m.Metalness;
    const float pTransmissive = (1.0f - m.Metalness) * (1.0f - FD) * m.Transmission;
    const float alpha = m.Roughness * m.Roughness;

    f3 bsdf = F3s(0.0f);
    outPdf = 0.0f;
    float pdf;
    if (isReflection)
    {
        // evaluateDiffuseBRDF, :11-15
        pdf = div_(L.z * 1.0f, PT_PI);
        bsdf = bsdf + ((m.Color * L.z) / PT_PI) * pDiffuse;
        outPdf += pdf * pDiffuse;
        // evaluateGlossyBSDF, :22-25
        bsdf = bsdf + EvaluateReflection(V, L, F3s(1.0f), alpha, pdf) * pGlossy;
        outPdf += pdf * pGlossy;
        // evaluateMetallicBRDF, :32-37
        const f3 Hm = normalize(V + L);
        const f3 F0 = mix(m.Color, F3s(1.0f), SchlickFresnel(dot(V, Hm)));
        bsdf = bsdf + EvaluateReflection(V, L, F0, alpha, pdf) * pMetallic;
        outPdf += pdf * pMetallic;
    }
    else
    {
        // evaluateBTDF, :44-47
        bsdf = bsdf + EvaluateRefraction(V, L, m.Color, alpha, m.Eta, pdf) * pTransmissive;
        outPdf += pdf * pTransmissive;
    }
    return bsdf;
}

PT_DEV BSDFSample sampleBSDF(const MaterialSample &m, f3 V, uint32_t &rngState) // :105-132
{
    const float alpha = m.Roughness * m.Roughness;
    f2 u;
    u.x = rnd(rngState);
    u.y = rnd(rngState);
    const f3 H = SampleGGX(u, V, alpha);
    const float FD = DielectricFresnel(abs_(dot(V, H)), m.Eta);

    f3 L;
    if (rnd(rngState) < m.Metalness)
        L = normalize(reflect(-V, H));
    else
    {
        if (rnd(rngState) < FD)
            L = normalize(reflect(-V, H));
        else
        {
            if (rnd(rngState) < m.Transmission)
                L = normalize(refract(-V, H, m.Eta));
            else
            {
                f2 u2;
                u2.x = rnd(rngState);
                u2.y = rnd(rngState);
                L = sampleCosineHemisphere(u2);
            }
        }
    }
    BSDFSample ret;
    ret.Direction = L;
    ret.Color = evaluateBSDF(m, V, L, ret.Pdf);
    return ret;
}

// ---- ray.glsl ------------------------------------------------------------------------------

PT_DEV f3 pinholeDirection(float pcx, float pcy, uint32_t resX, uint32_t resY, const float *ViewInverse, const float *ProjInverse) // :64-65,72-73
{
    const float inUVx = div_(pcx, (float)resX);
    const float inUVy = div_(pcy, (float)resY);
    const float dx = inUVx * 2.0f - 1.0f;
    const float dy = inUVy * 2.0f - 1.0f;
    const f4 target = mul4(ProjInverse, dx, dy, 1, 1);
    const f3 nt = normalize(F3(target.x, target.y, target.z));
    const f4 d = mul4(ViewInverse, nt.x, nt.y, nt.z, 0);
    return F3(d.x, d.y, d.z);
}

// :58-85; with DIFF also the directions through the pixels one to the right (rxDir) / below (ryDir)
template <bool DIFF>
PT_DEV void constructPrimaryRay(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                const float *ProjInverse, f2 u, f3 &origin, f3 &direction, f3 &rxDir, f3 &ryDir)
{
    const float pcx = (float)px + u.x;
    const float pcy = (float)py + u.y;
    const f4 o = mul4(ViewInverse, 0, 0, 0, 1);
    origin = F3(o.x, o.y, o.z);
    direction = pinholeDirection(pcx, pcy, resX, resY, ViewInverse, ProjInverse);
    if (DIFF)
    {
        rxDir = pinholeDirection(pcx + 1.0f, pcy + 0.0f, resX, resY, ViewInverse, ProjInverse);
        ryDir = pinholeDirection(pcx + 0.0f, pcy + 1.0f, resX, resY, ViewInverse, ProjInverse);
    }
}
PT_DEV void constructPrimaryRay(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                const float *ProjInverse, f2 u, f3 &origin, f3 &direction)
{
    f3 rx, ry;
    constructPrimaryRay<false>(px, py, resX, resY, ViewInverse, ProjInverse, u, origin, direction, rx, ry);
}

PT_DEV f3 lensDirection(float pcx, float pcy, uint32_t resX, uint32_t resY, const float *ViewInverse, const float *ProjInverse,
                        f3 originCameraSpace, float focalDistance) // :24-25,35-38
{
    const float inUVx = div_(pcx, (float)resX);
    const float inUVy = div_(pcy, (float)resY);
    const float dx = inUVx * 2.0f - 1.0f;
    const float dy = inUVy * 2.0f - 1.0f;
    const f4 target = mul4(ProjInverse, dx, dy, 1, 1);
    const float ft = div_(focalDistance, target.z);
    const f3 pFocus = F3(target.x, target.y, target.z) * ft;
    const f3 nd = normalize(pFocus - originCameraSpace);
    const f4 d = mul4(ViewInverse, nd.x, nd.y, nd.z, 0);
    return F3(d.x, d.y, d.z);
}

template <bool DIFF>
PT_DEV void constructPrimaryRayLens(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                    const float *ProjInverse, f2 u, f2 u2, float lensRadius, float focalDistance,
                                    f3 &origin, f3 &direction, f3 &rxDir, f3 &ryDir) // :16-56
{
    const float pcx = (float)px + u.x;
    const float pcy = (float)py + u.y;
    const f2 disk = sampleUniformDiskConcentric(u2);
    const float plx = lensRadius * disk.x, ply = lensRadius * disk.y;
    const f3 originCameraSpace = F3(plx, ply, 0);
    const f4 o = mul4(ViewInverse, originCameraSpace.x, originCameraSpace.y, originCameraSpace.z, 1);
    origin = F3(o.x, o.y, o.z);
    direction = lensDirection(pcx, pcy, resX, resY, ViewInverse, ProjInverse, originCameraSpace, focalDistance);
    if (DIFF)
    {
        rxDir = lensDirection(pcx + 1.0f, pcy + 0.0f, resX, resY, ViewInverse, ProjInverse, originCameraSpace, focalDistance);
        ryDir = lensDirection(pcx + 0.0f, pcy + 1.0f, resX, resY, ViewInverse, ProjInverse, originCameraSpace, focalDistance);
    }
}
PT_DEV void constructPrimaryRayLens(uint32_t px, uint32_t py, uint32_t resX, uint32_t resY, const float *ViewInverse,
                                    const float *ProjInverse, f2 u, f2 u2, float lensRadius, float focalDistance,
                                    f3 &origin, f3 &direction)
{
    f3 rx, ry;
    constructPrimaryRayLens<false>(px, py, resX, resY, ViewInverse, ProjInverse, u, u2, lensRadius, focalDistance, origin, direction, rx, ry);
}

PT_DEV float offsetComponent(float o, float n) // :93-106 (Waechter-Binder)
{
    const float origin_const = 1.0f / 32.0f;
    const float float_scale = 1.0f / 65536.0f;
    const float int_scale = 256.0f;
    const int32_t of_i = (int32_t)(int_scale * n);
    const uint32_t bits = __float_as_uint(o) + (uint32_t)((o < 0) ? -of_i : of_i);
    const float p_i = __uint_as_float(bits);
    return (abs_(o) < origin_const) ? o + float_scale * n : p_i;
}
PT_DEV f3 offsetRayOriginSelfIntersection(f3 origin, f3 normal)
{
    return F3(offsetComponent(origin.x, normal.x), offsetComponent(origin.y, normal.y), offsetComponent(origin.z, normal.z));
}

PT_DEV f3 offsetRayOriginShadowTerminator(f3 P, f3 p0, f3 n0, f3 p1, f3 n1, f3 p2, f3 n2, f3 bary, bool isRefracted) // :109-131
{
    f3 tmpu = P - p0;
    f3 tmpv = P - p1;
    f3 tmpw = P - p2;
    if (isRefracted)
    {
        n0 = -n0;
        n1 = -n1;
        n2 = -n2;
    }
    const float dotu = fmin_(0.0f, dot(tmpu, n0));
    const float dotv = fmin_(0.0f, dot(tmpv, n1));
    const float dotw = fmin_(0.0f, dot(tmpw, n2));
    tmpu = tmpu - n0 * dotu;
    tmpv = tmpv - n1 * dotv;
    tmpw = tmpw - n2 * dotw;
    return ((P + tmpu * bary.x) + tmpv * bary.y) + tmpw * bary.z;
}

// ---- sampling.glsl ---------------------------------------------------------------------------

struct LightSample
{
    f3 Direction;
    float Distance;
    f3 Color;
    float Attenuation;
};

PT_DEV LightSample sampleLight(const PtxLightsUbo *ubo, f3 u, f3 position, float &pdf) // :25-56
{
    const uint32_t lightCount = ubo->LightCount;
    const uint32_t lightIndex = (uint32_t)(u.x * (float)(lightCount + 1));
    pdf = div_(1.0f, (float)(lightCount + 1));
    LightSample ret;
    f2 uyz;
    uyz.x = u.y;
    uyz.y = u.z;
    if (lightIndex >= lightCount)
    {
        const f2 d2 = sampleUniformDiskConcentric(uyz);
        const f3 diskPoint = F3(d2.x, d2.y, 0.0f) * 0.001f;
        const f3 direction = normalize(F3(ubo->Directional.Direction[0], ubo->Directional.Direction[1], ubo->Directional.Direction[2]));
        ret.Direction = normalize(direction + mul(computeTangentSpace(direction), diskPoint));
        ret.Color = F3(ubo->Directional.Color[0], ubo->Directional.Color[1], ubo->Directional.Color[2]);
        ret.Distance = 100000.0f;
        ret.Attenuation = 1.0f;
        return ret;
    }
    const PtxPointLight *light = &ubo->Lights[lightIndex];
    const f3 lpos = F3(light->Position[0], light->Position[1], light->Position[2]);
    const f2 d2 = sampleUniformDiskConcentric(uyz);
    const f3 diskPoint = F3(d2.x, d2.y, 0.0f) * 0.1f;
    const f3 direction = normalize(position - lpos);
    const f3 newPosition = lpos + mul(computeTangentSpace(direction), diskPoint);
    ret.Distance = length(position - newPosition);
    ret.Direction = normalize(position - newPosition);
    ret.Color = F3(light->Color[0], light->Color[1], light->Color[2]);
    const float attenuation = div_(1.0f, light->AttenuationConstant + ret.Distance * light->AttenuationLinear +
                                         ret.Distance * ret.Distance * light->AttenuationQuadratic);
    ret.Attenuation = clamp_(attenuation, 0.0f, 1.0f);
    return ret;
}

// ---- tracing.glsl: ray differentials and texture footprint ------------------------------------------
// These feed only textureGrad: a scene whose textures are all the fixed 1x1 defaults cannot see
// them, and its kernels (TEX = false) do not carry them.  A scene with
// uploaded textures runs the TEX = true variants, which do.

PT_DEV void computeDpnDuv(const f3 *p, const f3 *n, const f2 *uv, f3 vtxTangent, f3 vtxBitangent, f3 &dpdu, f3 &dpdv, f3 &dndu,
                          f3 &dndv) // :2-28
{
    const f3 e1 = p[1] - p[0], e2 = p[2] - p[0];
    const f3 en1 = n[1] - n[0], en2 = n[2] - n[0];
    const float du1 = uv[1].x - uv[0].x, dv1 = uv[1].y - uv[0].y, du2 = uv[2].x - uv[0].x, dv2 = uv[2].y - uv[0].y;
    const float det = du1 * dv2 - du2 * dv1;
    if (abs_(det) < 1e-8f)
    {
        dpdu = vtxTangent;
        dpdv = vtxBitangent;
        dndu = F3s(0.0f);
        dndv = F3s(0.0f);
    }
    else
    {
        const float invDet = div_(1.0f, det);
        dpdu = (e1 * dv2 - e2 * dv1) * invDet;
        dpdv = (e1 * (-du2) + e2 * du1) * invDet;
        dndu = (en1 * dv2 - en2 * dv1) * invDet;
        dndv = (en1 * (-du2) + en2 * du1) * invDet;
    }
}

PT_DEV void computeDpDxy(f3 p, f3 rxOrigin, f3 rxDirection, f3 ryOrigin, f3 ryDirection, f3 n, f3 &dpdx, f3 &dpdy) // :31-41
{
    const float d = -dot(n, p);
    const float tx = div_(-dot(n, rxOrigin) - d, dot(n, rxDirection));
    const f3 px = rxOrigin + rxDirection * tx;
    const float ty = div_(-dot(n, ryOrigin) - d, dot(n, ryDirection));
    const f3 py = ryOrigin + ryDirection * ty;
    dpdx = px - p;
    dpdy = py - p;
}

PT_DEV float differenceOfProducts(float a, float b, float c, float d) // :44-50 (explicit fma in the GLSL)
{
    const float cd = c * d;
    const float dop = __builtin_fmaf(a, b, -cd);
    const float error = __builtin_fmaf(-c, d, cd);
    return dop + error;
}

PT_DEV float clampInf(float x) { return __builtin_isinf(x) ? 0.0f : clamp_(x, -1e8f, 1e8f); }

PT_DEV f4 computeDerivatives(f3 dpdx, f3 dpdy, f3 dpdu, f3 dpdv) // :53-78
{
    const float ata00 = dot(dpdu, dpdu);
    const float ata01 = dot(dpdu, dpdv);
    const float ata11 = dot(dpdv, dpdv);
    float invDet = div_(1, differenceOfProducts(ata00, ata11, ata01, ata01));
    invDet = __builtin_isinf(invDet) ? 0.0f : invDet;
    const float atb0x = dot(dpdu, dpdx);
    const float atb1x = dot(dpdv, dpdx);
    const float atb0y = dot(dpdu, dpdy);
    const float atb1y = dot(dpdv, dpdy);
    f4 r;
    r.x = clampInf(differenceOfProducts(ata11, atb0x, ata01, atb1x) * invDet);
    r.y = clampInf(differenceOfProducts(ata00, atb1x, ata01, atb0x) * invDet);
    r.z = clampInf(differenceOfProducts(ata11, atb0y, ata01, atb1y) * invDet);
    r.w = clampInf(differenceOfProducts(ata00, atb1y, ata01, atb0y) * invDet);
    return r;
}

struct DiffRays
{
    f3 rxOrigin, rxDirection, ryOrigin, ryDirection;
};

PT_DEV void computeReflectedDifferentialRays(f4 derivatives, f3 n, f3 p, f3 viewDir, f3 reflectedDir, f3 dndu, f3 dndv, DiffRays &r) // :81-108
{
    const float dudx = derivatives.x, dvdx = derivatives.y, dudy = derivatives.z, dvdy = derivatives.w;
    const f3 dndx = dndu * dudx + dndv * dvdx;
    const f3 dndy = dndu * dudy + dndv * dvdy;
    const float d = -dot(n, p);
    const float tx = div_(-dot(n, r.rxOrigin) - d, dot(n, r.rxDirection));
    const f3 px = r.rxOrigin + r.rxDirection * tx;
    const float ty = div_(-dot(n, r.ryOrigin) - d, dot(n, r.ryDirection));
    const f3 py = r.ryOrigin + r.ryDirection * ty;
    const f3 dwodx = -r.rxDirection - viewDir;
    const f3 dwody = -r.ryDirection - viewDir;
    r.rxOrigin = px;
    r.ryOrigin = py;
    const float dwoDotn_dx = dot(dwodx, n) + dot(viewDir, dndx);
    const float dwoDotn_dy = dot(dwody, n) + dot(viewDir, dndy);
    const float vn = dot(viewDir, n);
    r.rxDirection = normalize((reflectedDir - dwodx) + (dndx * vn + n * dwoDotn_dx) * 2.0f);
    r.ryDirection = normalize((reflectedDir - dwody) + (dndy * vn + n * dwoDotn_dy) * 2.0f);
}

PT_DEV void computeRefractedDifferentialRays(f4 derivatives, f3 n, f3 p, f3 viewDir, f3 refractedDir, f3 dndu, f3 dndv, float eta,
                                             DiffRays &r) // :111-148
{
    const float dudx = derivatives.x, dvdx = derivatives.y, dudy = derivatives.z, dvdy = derivatives.w;
    f3 dndx = dndu * dudx + dndv * dvdx;
    f3 dndy = dndu * dudy + dndv * dvdy;
    const float d = -dot(n, p);
    const float tx = div_(-dot(n, r.rxOrigin) - d, dot(n, r.rxDirection));
    const f3 px = r.rxOrigin + r.rxDirection * tx;
    const float ty = div_(-dot(n, r.ryOrigin) - d, dot(n, r.ryDirection));
    const f3 py = r.ryOrigin + r.ryDirection * ty;
    const f3 dwodx = -r.rxDirection - viewDir;
    const f3 dwody = -r.ryDirection - viewDir;
    r.rxOrigin = px;
    r.ryOrigin = py;
    if (dot(viewDir, n) < 0.0f)
    {
        n = -n;
        dndx = -dndx;
        dndy = -dndy;
    }
    const float dwoDotn_dx = dot(dwodx, n) + dot(viewDir, dndx);
    const float dwoDotn_dy = dot(dwody, n) + dot(viewDir, dndy);
    const float mu = div_(dot(viewDir, n), eta) - abs_(dot(refractedDir, n));
    const float dmudx = dwoDotn_dx * (div_(1.0f, eta) + div_(div_(1.0f, eta * eta) * dot(viewDir, n), dot(refractedDir, n)));
    const float dmudy = dwoDotn_dy * (div_(1.0f, eta) + div_(div_(1.0f, eta * eta) * dot(viewDir, n), dot(refractedDir, n)));
    r.rxDirection = normalize((refractedDir - dwodx * eta) + (dndx * mu + n * dmudx));
    r.ryDirection = normalize((refractedDir - dwody * eta) + (dndy * mu + n * dmudy));
}

PT_DEV float computeLod(f4 derivatives) // :151-161, log2 through the fixed kernel
{
    const float sx = sqrt_(derivatives.x * derivatives.x + derivatives.y * derivatives.y);
    const float sy = sqrt_(derivatives.z * derivatives.z + derivatives.w * derivatives.w);
    const float smax = fmax_(sx, sy);
    return smax == 0.0f ? 0.0f : (float)log2_((double)smax);
}

// ---- software sampler (row N1) ---------------------------------------------------------------------
// What the Vulkan sampler of Renderer.cpp:103-112 does (linear min/mag/mip, repeat addressing),
// with fixed arithmetic shared with the oracle.  Anisotropic filtering is implementation-defined
// in Vulkan and is NOT modelled: textureGrad is isotropic trilinear.

struct DevTexture
{
    uint32_t width, height, levels, format;
    uint32_t levelOffset[16]; // texels, into the pool of its format
};

struct TextureView
{
    const DevTexture *textures; // render time: levelOffset indexes the decoded pool; upload time (k_blit_level): the pool of the format
    uint32_t textureCount;
    const uint32_t *texels8; // upload time only: RGBA8 pool
    const float4 *texelsF;   // render time: the decoded pool of every texture; upload time: the RGBA32F pool
    const float *srgbLut;    // upload time only: 256 entries, sRGB byte -> linear
};

PT_DEV uint32_t levelDim(uint32_t d, uint32_t level) { const uint32_t v = d >> level; return v ? v : 1u; }

PT_DEV float srgbToLinear(float c) { return c <= 0.04045f ? c / 12.92f : pow_((c + 0.055f) / 1.055f, 2.4f); }
PT_DEV float linearToSrgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * pow_(c, 1.0f / 2.4f) - 0.055f; }
PT_DEV uint32_t quantize8(float x)
{
    if (!(x > 0.0f))
        return 0u;
    if (x > 1.0f)
        x = 1.0f;
    return (uint32_t)__builtin_floorf(x * 255.0f + 0.5f);
}

// Upload-time form of a texel (k_blit_level builds mip chains in the image's own format, Image.cpp:264-300): decode from
// the pool of the texture's format.
PT_DEV f4 fetchTexelEncoded(const TextureView &tv, const DevTexture &t, uint32_t level, uint32_t x, uint32_t y)
{
    const size_t idx = (size_t)t.levelOffset[level] + (size_t)y * levelDim(t.width, level) + x;
    f4 r;
    if (t.format == PTX_TEXTURE_RGBA32F)
    {
        const float4 v = tv.texelsF[idx];
        r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
        return r;
    }
    const uint32_t p = tv.texels8[idx];
    if (t.format == PTX_TEXTURE_RGBA8_SRGB)
    {
        r.x = tv.srgbLut[p & 255u]; r.y = tv.srgbLut[(p >> 8) & 255u]; r.z = tv.srgbLut[(p >> 16) & 255u];
    }
    else
    {
        r.x = (float)(p & 255u) / 255.0f; r.y = (float)((p >> 8) & 255u) / 255.0f; r.z = (float)((p >> 16) & 255u) / 255.0f;
    }
    r.w = (float)(p >> 24) / 255.0f;
    return r;
}

// Render-time form: every level of every texture sits DECODED (four floats per texel, the values fetchTexelEncoded returns)
// in one pool, written once at upload by k_decode_texels -- a texel is ONE dwordx4 load.  In the 8-bit pools a texel cost
// a load, three dependent loads from the sRGB table (or three IEEE divisions by 255) and one more division for alpha, per
// texel of every bilinear footprint of every anisotropic tap; 288 GB of HBM hold the 4x larger pool of any scene the
// reference's 1 GiB texture budget (Config.h:63-64) admits.  The values are the same bits, so nothing downstream changes.
PT_DEV f4 fetchTexel(const TextureView &tv, const DevTexture &t, uint32_t level, uint32_t x, uint32_t y)
{
    const float4 v = tv.texelsF[(size_t)t.levelOffset[level] + (size_t)y * levelDim(t.width, level) + x];
    f4 r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}

PT_DEV f4 lerp4(f4 a, f4 b, float t)
{
    f4 r;
    r.x = a.x * (1.0f - t) + b.x * t; r.y = a.y * (1.0f - t) + b.y * t; r.z = a.z * (1.0f - t) + b.z * t; r.w = a.w * (1.0f - t) + b.w * t;
    return r;
}

// Repeat addressing: floor(x) mod n for an integer-valued x0, in float so that CPU and GPU agree for any finite x.  This is
// sampler addressing, not a shader `/`: the quotient x0 * rcp(n) carries two roundings, so for an n that is not a power of two
// floor() may land one below or above the true k (x0 = n = 41: 41 * RN(1/41) < 1) -- the remainder is therefore CORRECTED by one
// period, not clamped.  For |x0| < 2^22 the quotient's error is below |x0| / n * 2^-23 < 1/2, k is off by at most one, k * n and
// x0 - k * n are exact integers below 2^24, and the result is the mathematical floor(x0) mod n (tests/test_textures.py checks it
// against np.mod for every extent the sampler can meet); beyond that range a float no longer names one texel and the two guards
// only keep the index inside the level.
PT_DEV uint32_t wrapRepeat(float x0, uint32_t n)
{
    const float fn = (float)n;
    float m = x0 - __builtin_floorf(div_(x0, fn)) * fn;
    if (m < 0.0f) m += fn;
    else if (m >= fn) m -= fn;
    if (!(m >= 0.0f)) m = 0.0f;
    const uint32_t i = (uint32_t)m;
    return i >= n ? n - 1 : i;
}

// One level of a texture as the bilinear lookup needs it, computed once per textureGrad call (the anisotropic filter runs up to
// sixteen taps on the same two levels).  iw / ih: the exact reciprocal of an extent that is a power of two, 0 otherwise.
struct LevelView
{
    uint32_t w, h, offset;
    float fw, fh, iw, ih;
};
PT_DEV float exactReciprocalOfPow2(uint32_t n, float fn) { return (n & (n - 1u)) == 0u ? __uint_as_float(0x7f000000u - __float_as_uint(fn)) : 0.0f; }
PT_DEV LevelView levelView(const DevTexture &t, uint32_t level)
{
    LevelView lv;
    lv.w = levelDim(t.width, level);
    lv.h = levelDim(t.height, level);
    lv.offset = t.levelOffset[level];
    lv.fw = (float)lv.w;
    lv.fh = (float)lv.h;
    lv.iw = exactReciprocalOfPow2(lv.w, lv.fw);
    lv.ih = exactReciprocalOfPow2(lv.h, lv.fh);
    return lv;
}

// wrapRepeat(x0, n) for an integer-valued x0, bit for bit, without its reciprocal where that can be shown: for |x0| < 2^22
// wrapRepeat returns the mathematical floor(x0) mod n (above) and its two guards never fire.  With n = 2^k the quotient x0 * 2^-k
// is exact, so floor() is the true k and no correction is needed: one multiply for the reciprocal, the product and the two
// corrections, four times per bilinear lookup, two lookups per trilinear tap, up to sixteen taps per textureGrad.
constexpr float kExactWrapLimit = 4194304.0f; // 2^22
PT_DEV uint32_t wrapRepeatInt(float x0, uint32_t n, float fn, float inv)
{
    if (inv != 0.0f && abs_(x0) < kExactWrapLimit)
        return (uint32_t)(x0 - __builtin_floorf(x0 * inv) * fn);
    return wrapRepeat(x0, n);
}
// wrapRepeat(x0 + 1, n) from i0 = wrapRepeat(x0, n): inside the exact range it is the next index
PT_DEV uint32_t wrapRepeatNext(float x0, uint32_t i0, uint32_t n)
{
    if (abs_(x0) < kExactWrapLimit)
        return i0 + 1u == n ? 0u : i0 + 1u;
    return wrapRepeat(x0 + 1.0f, n);
}

PT_DEV f4 fetchLevelTexel(const TextureView &tv, const LevelView &lv, uint32_t x, uint32_t y)
{
    // the pool holds fewer than 2^32 texels (ptx_scene_upload) and an extent is at most 2^15: 32-bit index arithmetic
    const float4 v = tv.texelsF[lv.offset + y * lv.w + x];
    f4 r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}

PT_DEV f4 sampleLevel(const TextureView &tv, const LevelView &lv, float u, float v)
{
    if (lv.w == 1 && lv.h == 1) // exact for 1x1 (hardware weights are fixed point and sum to 1)
        return fetchLevelTexel(tv, lv, 0, 0);
    if (!(abs_(u) < 1e9f)) u = 0.0f;
    if (!(abs_(v) < 1e9f)) v = 0.0f;
    const float x = u * lv.fw - 0.5f, y = v * lv.fh - 0.5f;
    const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
    const float ax = x - x0, ay = y - y0;
    const uint32_t ix0 = wrapRepeatInt(x0, lv.w, lv.fw, lv.iw), iy0 = wrapRepeatInt(y0, lv.h, lv.fh, lv.ih);
    const uint32_t ix1 = wrapRepeatNext(x0, ix0, lv.w), iy1 = wrapRepeatNext(y0, iy0, lv.h);
    const f4 top = lerp4(fetchLevelTexel(tv, lv, ix0, iy0), fetchLevelTexel(tv, lv, ix1, iy0), ax);
    const f4 bot = lerp4(fetchLevelTexel(tv, lv, ix0, iy1), fetchLevelTexel(tv, lv, ix1, iy1), ax);
    return lerp4(top, bot, ay);
}
PT_DEV f4 sampleLevel(const TextureView &tv, const DevTexture &t, uint32_t level, float u, float v)
{
    return sampleLevel(tv, levelView(t, level), u, v);
}

// textureGrad with the reference's sampler (trilinear, anisotropy at the device maximum, Renderer.cpp:103-110), by the
// scheme of the Vulkan / EXT_texture_filter_anisotropic specifications: eta = min(rho_max / rho_min, 16), N = ceil(eta)
// trilinear taps at LOD log2(rho_max / eta) along the longer gradient at (i / (N + 1) - 1/2), averaged; N = 1 is the
// isotropic lookup.
#ifndef PTX_EXP_MAX_ANISOTROPY
#define PTX_EXP_MAX_ANISOTROPY 16.0f // (experiments only: what the anisotropic taps cost; the oracle's PTO_MAX_ANISOTROPY is 16)
#endif
constexpr float kMaxAnisotropy = PTX_EXP_MAX_ANISOTROPY;

// The two levels of a trilinear lookup and the blend between them: the same for every tap of one textureGrad.
struct TrilinearView
{
    LevelView l0, l1;
    float f;
    bool single; // f == 0 or no second level: the lookup is the first level's
};
PT_DEV TrilinearView trilinearView(const DevTexture &t, float lod)
{
    const float q = (float)(t.levels - 1);
    if (!(lod >= 0.0f)) lod = 0.0f;
    if (lod > q) lod = q;
    const float d0 = __builtin_floorf(lod);
    const uint32_t l0 = (uint32_t)d0, l1 = l0 + 1 < t.levels ? l0 + 1 : t.levels - 1;
    TrilinearView tv3;
    tv3.f = lod - d0;
    tv3.single = tv3.f == 0.0f || l1 == l0;
    tv3.l0 = levelView(t, l0);
    tv3.l1 = levelView(t, l1);
    return tv3;
}
PT_DEV f4 trilinearSample(const TextureView &tv, const TrilinearView &tl, float u, float v)
{
    const f4 c0 = sampleLevel(tv, tl.l0, u, v);
    if (tl.single)
        return c0;
    return lerp4(c0, sampleLevel(tv, tl.l1, u, v), tl.f);
}
PT_DEV f4 trilinearSample(const TextureView &tv, const DevTexture &t, float lod, float u, float v)
{
    return trilinearSample(tv, trilinearView(t, lod), u, v);
}

PT_DEV f4 textureGradSample(const TextureView &tv, const DevTexture &t, float u, float v, float dudx, float dvdx, float dudy, float dvdy)
{
    if (t.levels <= 1)
        return sampleLevel(tv, t, 0, u, v);
    const float mux = dudx * (float)t.width, mvx = dvdx * (float)t.height;
    const float muy = dudy * (float)t.width, mvy = dvdy * (float)t.height;
    const float rx = sqrt_(mux * mux + mvx * mvx), ry = sqrt_(muy * muy + mvy * mvy);
    const float rmax = fmax_(rx, ry), rmin = fmin_(rx, ry);
    float eta = 1.0f;
    if (rmax > 0.0f && rmax < 3.0e38f) // finite footprint: otherwise a single tap
        eta = rmin > 0.0f ? fmin_(div_(rmax, rmin), kMaxAnisotropy) : kMaxAnisotropy;
    if (!(eta >= 1.0f)) eta = 1.0f;
    const float n = __builtin_ceilf(eta);
    const float rho = div_(rmax, eta);
    const float lod = rho > 0.0f ? (float)log2_((double)rho) : 0.0f;
    const TrilinearView tl = trilinearView(t, lod);
    if (n <= 1.0f || lod >= (float)(t.levels - 1)) // every tap would read the 1x1 top level: one tap
        return trilinearSample(tv, tl, u, v);
    const float du = rx >= ry ? dudx : dudy, dv = rx >= ry ? dvdx : dvdy;
    f4 sum;
    sum.x = sum.y = sum.z = sum.w = 0.0f;
    const int taps = (int)n;
    for (int i = 1; i <= taps; i++)
    {
        const float w = div_((float)i, n + 1.0f) - 0.5f;
        const f4 c = trilinearSample(tv, tl, u + du * w, v + dv * w);
        sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
    }
    const float rn = rcp_(n);
    sum.x *= rn; sum.y *= rn; sum.z *= rn; sum.w *= rn;
    return sum;
}

// ---- material.glsl with the fixed 1x1 default textures ------------------------------------------

// Texels of slots 0..8 after format decode (ShaderRendererTypes.incl:49-56; sRGB for
// Color/Specular/Emissive, UNORM otherwise: TextureUploader.cpp:571-594).  Scene
// textures (index >= 9) go through the software sampler (TEX variants); an index past the
// uploaded table samples as the white placeholder (Renderer.cpp:421-429).
PT_DEV f4 sampleTexture(uint32_t idx)
{
    f4 w;
    w.x = w.y = w.z = w.w = 1.0f;
    if (idx == PTX_DEFAULT_NORMAL_TEXTURE_INDEX)
    {
        w.x = 128.0f / 255.0f;
        w.y = 128.0f / 255.0f;
    }
    else if (idx == PTX_DEFAULT_EMISSIVE_TEXTURE_INDEX || idx == PTX_DEFAULT_GLOSSINESS_TEXTURE_INDEX ||
             idx == PTX_DEFAULT_SHININESS_TEXTURE_INDEX)
        w.x = w.y = w.z = w.w = 0.0f;
    return w;
}

PT_DEV f3 ReconstructNormalFromXY(f3 n) // :55-60
{
    n = F3(2.0f * n.x - 1.0f, 2.0f * n.y - 1.0f, 2.0f * n.z - 1.0f);
    return F3(n.x, n.y, sqrt_(fmax_(1 - n.x * n.x - n.y * n.y, 0.0f)));
}

// textureGrad(textures[idx], uv, dv.xy, dv.zw)
template <bool TEX>
PT_DEV f4 sampleTexture(const TextureView &tv, uint32_t idx, f2 uv, f4 dv)
{
    if (TEX && idx >= PTX_SCENE_TEXTURE_OFFSET && idx - PTX_SCENE_TEXTURE_OFFSET < tv.textureCount)
        return textureGradSample(tv, tv.textures[idx - PTX_SCENE_TEXTURE_OFFSET], uv.x, uv.y, dv.x, dv.y, dv.z, dv.w);
    return sampleTexture(idx);
}

PT_DEV f3 rgb(f4 t) { return F3(t.x, t.y, t.z); }
PT_DEV f3 ld3(const float *p) { return F3(p[0], p[1], p[2]); }

// The three vertices of one triangle, deindexed and stored in BVH leaf order beside the traversal's Tri record:
// the closest-hit stage reads ONE contiguous, 16-byte aligned 176-byte record per hit instead of chasing
// hit -> pair -> 3 indices -> 3 scattered 56-byte vertices (three dependent gathers at 2 waves / SIMD).
// Vertex k occupies floats [14 k, 14 k + 14) in PtxVertex order (object space, as common.glsl:27-46 fetches them);
// floats 42..43 are padding.  Then what closestHit.rchit:63-74 derives from the corners alone, computed once per
// (re)build with the very same arithmetic instead of once per hit: world-space positions [44, 53) and normals
// [53, 62) of the three corners (sampling.glsl:5-15) and the geometric normal [62, 65) -- a quarter of the
// normalisations of the closest-hit stage.
struct ShadeTri
{
    float4 v[17];
};
static_assert(sizeof(ShadeTri) == 272, "ShadeTri is 272 B");

struct SceneView // read-only device views of the uploaded scene
{
    const ShadeTri *shadeTris; // [triangle slot in leaf order]
    const PtxVertex *vertices;
    const uint32_t *indices;
    const PtxMetallicRoughnessMaterial *mr;
    const PtxSpecularGlossinessMaterial *sg;
    const PtxPhongMaterial *phong;
    const struct DevPair *pairs;
    const PtxLightsUbo *lights;
    uint32_t dxNormalTextures;
    uint32_t skyKind; // PTX_SKYBOX_*; its 1 / 6 single-level images sit at tex.textures[tex.textureCount ...]
    TextureView tex;
};

// one (instance, mesh): world = A_instance * A_mesh * x (sampling.glsl:5-15); Rinv is
// the inverse of the linear part, for the inverse-transpose normal transform
struct DevPair
{
    float M[12];
    float Rinv[9]; // columns c0, c1, c2
    uint32_t vertexOffset, indexOffset, materialId;
    uint32_t flags; // kPairNonOpaque: geometry without VK_GEOMETRY_OPAQUE_BIT, the any-hit stages run (AccelerationStructure.cpp:94-97);
                    // kPairTextured: its material samples at least one scene texture (k_shade's sort groups those hits)
};
constexpr uint32_t kPairNonOpaque = 1u, kPairTextured = 2u;

// What anyhit.rahit leaves in the payload: the nearest ignored (alpha < 0.5) candidate.  The any-hit
// invocation order is the driver's; nearest-with-id-tie-break is the order-independent reading of
// anyhit.rahit:54-61 that closestHit.rchit:105-106 can observe.
struct Decal
{
    float dist;          // -1 = none (payload.DirectLightPdf)
    uint32_t slot;       // which triangle (leaf order) and where on it: its colour (payload.LightDirection) and alpha
    float u, v;          // (payload.LightDistance) are fetched by the closest-hit stage, and only if the hit lies behind it
    uint32_t pair, prim; // tie-break of equal distances, and the material of the colour fetch
};
PT_DEV Decal noDecal()
{
    Decal d;
    d.dist = -1.0f;
    d.slot = 0u;
    d.u = d.v = 0.0f;
    d.pair = d.prim = 0xffffffffu;
    return d;
}

// anyhit.rahit:38-52 / occlusionAnyhit.rahit:37-50: texture(textures[colorIdx], uv) * colorFactor at a
// candidate hit (getColorTextureIdx / getColorFactor, material.glsl:25-54).  texture() in a ray-tracing
// stage has no implicit derivatives: base level.
PT_DEV f4 hitBaseColor(const SceneView &sv, uint32_t pairIdx, uint32_t slot, float u, float v)
{
    const DevPair *pr = &sv.pairs[pairIdx];
    const f3 bary = F3(1.0f - u - v, u, v);
    // texture coordinates of the three vertices: floats 3..4, 17..18, 31..32 of the record
    const ShadeTri *st = &sv.shadeTris[slot];
    const float4 q0 = st->v[0], q1 = st->v[1], q4 = st->v[4], q7 = st->v[7], q8 = st->v[8];
    const float u0 = q0.w, v0 = q1.x, u1 = q4.y, v1 = q4.z, u2 = q7.w, v2 = q8.x;
    const float tu = (u0 * bary.x + u1 * bary.y) + u2 * bary.z;
    const float tv = (v0 * bary.x + v1 * bary.y) + v2 * bary.z;
    const uint32_t materialType = pr->materialId & 0xffu, materialIndex = pr->materialId >> 8;
    uint32_t idx = 0;
    f4 factor;
    factor.x = 1.0f; factor.y = 0.0f; factor.z = 0.0f; factor.w = 1.0f;
    const float *col = nullptr;
    if (materialType == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS)
    {
        idx = sv.mr[materialIndex].ColorIdx;
        col = sv.mr[materialIndex].Color;
    }
    else if (materialType == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS)
    {
        idx = sv.sg[materialIndex].ColorIdx;
        col = sv.sg[materialIndex].Color;
    }
    else if (materialType == PTX_MATERIAL_TYPE_PHONG)
    {
        idx = sv.phong[materialIndex].ColorIdx;
        col = sv.phong[materialIndex].Color;
    }
    if (col)
    {
        factor.x = col[0]; factor.y = col[1]; factor.z = col[2]; factor.w = col[3];
    }
    f4 t;
    if (idx >= PTX_SCENE_TEXTURE_OFFSET && idx - PTX_SCENE_TEXTURE_OFFSET < sv.tex.textureCount)
        t = sampleLevel(sv.tex, sv.tex.textures[idx - PTX_SCENE_TEXTURE_OFFSET], 0, tu, tv);
    else
        t = sampleTexture(idx);
    t.x *= factor.x; t.y *= factor.y; t.z *= factor.z; t.w *= factor.w;
    return t;
}

// ---- miss.rmiss ------------------------------------------------------------------------------------

PT_DEV f3 hdrToLdr(f3 rgb) { return rgb / (1.0f + maxComponent(rgb)); } // common.glsl:17-20

PT_DEV f2 missSkyboxTexCoords(f3 dir) // miss.rmiss:20-25
{
    const float PI = 3.14159265359f; // common.glsl:3
    const float longitude = atan2_(dir.z, dir.x);
    const float latitude = asin_(-dir.y);
    return F2(div_(div_(longitude, 2.0f), PI) + 0.5f, div_(latitude, PI) + 0.5f);
}

// Cube map face selection of the Vulkan specification (largest magnitude, z before y before x on ties): face and the
// face coordinates (sc, tc) with major axis length ma.
PT_DEV void cubeFace(f3 r, uint32_t &face, float &sc, float &tc, float &ma)
{
    const float ax = abs_(r.x), ay = abs_(r.y), az = abs_(r.z);
    if (az >= ax && az >= ay)
    {
        face = r.z < 0.0f ? 5u : 4u;
        sc = r.z < 0.0f ? -r.x : r.x;
        tc = -r.y;
        ma = az;
    }
    else if (ay >= ax)
    {
        face = r.y < 0.0f ? 3u : 2u;
        sc = r.x;
        tc = r.y < 0.0f ? -r.z : r.z;
        ma = ay;
    }
    else
    {
        face = r.x < 0.0f ? 1u : 0u;
        sc = r.x < 0.0f ? r.z : -r.z;
        tc = -r.y;
        ma = ax;
    }
}

// Texel (ix, iy) of a face, where ONE of the indices may lie one step outside 0 .. n-1: the texel across that edge
// (seamless cube maps, Vulkan "Cube Map Edge Handling": the reference's sampler filters across face borders).  The texel
// centre, folded over the edge onto the cube's surface, goes through the face selection again: the index along the edge is
// kept, the index across it becomes the neighbour's border row.
PT_DEV f4 cubeTexel(const TextureView &tv, const DevTexture *faces, uint32_t face, int ix, int iy)
{
    const int n = (int)faces[face].width;
    if (ix >= 0 && ix < n && iy >= 0 && iy < n)
        return fetchTexel(tv, faces[face], 0, (uint32_t)ix, (uint32_t)iy);
    float sc = 2.0f * (div_((float)ix + 0.5f, (float)n)) - 1.0f, tc = 2.0f * (div_((float)iy + 0.5f, (float)n)) - 1.0f, ma = 1.0f;
    if (ix < 0 || ix >= n)
    {
        ma = 1.0f - (abs_(sc) - 1.0f);
        sc = sc < 0.0f ? -1.0f : 1.0f;
    }
    else
    {
        ma = 1.0f - (abs_(tc) - 1.0f);
        tc = tc < 0.0f ? -1.0f : 1.0f;
    }
    f3 r; // inverse of the selection table
    if (face == 0u) r = F3(ma, -tc, -sc);
    else if (face == 1u) r = F3(-ma, -tc, sc);
    else if (face == 2u) r = F3(sc, ma, tc);
    else if (face == 3u) r = F3(sc, -ma, -tc);
    else if (face == 4u) r = F3(sc, -tc, ma);
    else r = F3(-sc, -tc, -ma);
    uint32_t f2;
    float s2, t2, m2;
    cubeFace(r, f2, s2, t2, m2);
    const float mx = (float)(n - 1);
    const uint32_t jx = (uint32_t)clamp_(__builtin_floorf((0.5f * (div_(s2, m2)) + 0.5f) * (float)n), 0.0f, mx);
    const uint32_t jy = (uint32_t)clamp_(__builtin_floorf((0.5f * (div_(t2, m2)) + 0.5f) * (float)n), 0.0f, mx);
    return fetchTexel(tv, faces[f2], 0, jx, jy);
}

// bilinear lookup at (u, v) of a face with the footprint continuing on the neighbouring faces; at a corner of the cube,
// where three faces meet and the fourth texel does not exist, it is the mean of the other three (Vulkan "Cube Map Corner
// Handling")
PT_DEV f4 sampleFaceSeamless(const TextureView &tv, const DevTexture *faces, uint32_t face, float u, float v)
{
    const int n = (int)faces[face].width;
    if (!(abs_(u) < 1e9f)) u = 0.0f;
    if (!(abs_(v) < 1e9f)) v = 0.0f;
    const float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
    const float ax = x - x0, ay = y - y0;
    const int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
    const bool ox0 = ix0 < 0, ox1 = ix1 >= n, oy0 = iy0 < 0, oy1 = iy1 >= n;
    f4 zero;
    zero.x = zero.y = zero.z = zero.w = 0.0f;
    f4 c00 = (ox0 && oy0) ? zero : cubeTexel(tv, faces, face, ix0, iy0);
    f4 c10 = (ox1 && oy0) ? zero : cubeTexel(tv, faces, face, ix1, iy0);
    f4 c01 = (ox0 && oy1) ? zero : cubeTexel(tv, faces, face, ix0, iy1);
    f4 c11 = (ox1 && oy1) ? zero : cubeTexel(tv, faces, face, ix1, iy1);
    if ((ox0 || ox1) && (oy0 || oy1))
    {
        const float third = 1.0f / 3.0f;
        f4 mean;
        mean.x = ((c00.x + c10.x) + (c01.x + c11.x)) * third;
        mean.y = ((c00.y + c10.y) + (c01.y + c11.y)) * third;
        mean.z = ((c00.z + c10.z) + (c01.z + c11.z)) * third;
        mean.w = ((c00.w + c10.w) + (c01.w + c11.w)) * third;
        if (ox0 && oy0) c00 = mean;
        else if (ox1 && oy0) c10 = mean;
        else if (ox0 && oy1) c01 = mean;
        else c11 = mean;
    }
    const f4 top = lerp4(c00, c10, ax);
    const f4 bot = lerp4(c01, c11, ax);
    return lerp4(top, bot, ay);
}

// texture(samplerCube, dir): face and (s, t) by the Vulkan cube map face selection tables; bilinear filtering with
// seamless edges.
PT_DEV f4 sampleCube(const TextureView &tv, const DevTexture *faces, f3 r)
{
    uint32_t face;
    float sc, tc, ma;
    cubeFace(r, face, sc, tc, ma);
    const float u = 0.5f * (div_(sc, ma)) + 0.5f, v = 0.5f * (div_(tc, ma)) + 0.5f;
    return sampleFaceSeamless(tv, faces, face, u, v);
}

// payload.Emissive of miss.rmiss:16-39 (Pdf = -1 is the caller's path termination)
PT_DEV f3 missEmissive(const SceneView &sv, f3 rayDir)
{
    if (sv.skyKind == PTX_SKYBOX_2D)
    {
        const f2 uv = missSkyboxTexCoords(rayDir);
        return hdrToLdr(rgb(sampleLevel(sv.tex, sv.tex.textures[sv.tex.textureCount], 0, uv.x, uv.y)));
    }
    if (sv.skyKind == PTX_SKYBOX_CUBE)
        return rgb(sampleCube(sv.tex, sv.tex.textures + sv.tex.textureCount, rayDir));
    return F3(0.08f, 0.09f, 0.1f);
}

PT_DEV f3 specGlossMetalness(f3 specular, f3 color) // material.glsl:109-110, :138-139
{
    return F3(div_(fmax_(specular.x - 0.04f, 0.0f), (color.x - 0.04f) + 0.00001f),
              div_(fmax_(specular.y - 0.04f, 0.0f), (color.y - 0.04f) + 0.00001f),
              div_(fmax_(specular.z - 0.04f, 0.0f), (color.z - 0.04f) + 0.00001f));
}

// The five textureGrad results a material branch consumes, in the slot order of its struct: emissive, colour, normal,
// then roughness + metallic (MetallicRoughness), specular + glossiness (SpecularGlossiness) or specular + shininess (Phong).
struct MaterialTexels
{
    f4 emissive, color, normal, a, b;
};

// material.glsl:62-84, :86-113, :115-142: the three overloads, as functions of the texels their textureGrad calls
// return (the fetches have no side effects, so taking them first changes nothing).
PT_DEV MaterialSample sampleMaterial(const PtxMetallicRoughnessMaterial *m, const MaterialTexels &t, bool isHitFromInside)
{
    MaterialSample ret;
    ret.EmissiveColor = (rgb(t.emissive) + ld3(m->EmissiveColor)) * m->EmissiveIntensity;
    ret.Color = rgb(t.color) * ld3(m->Color);
    ret.Normal = ReconstructNormalFromXY(rgb(t.normal));
    ret.Roughness = t.a.y * m->Roughness;
    ret.Metalness = t.b.z * m->Metalness;
    ret.Transmission = m->Transmission;
    ret.AttenuationColor = ld3(m->AttenuationColor);
    ret.AttenuationDistance = m->AttenuationDistance;
    ret.Eta = isHitFromInside ? m->Ior : (div_(1.0f, m->Ior));
    return ret;
}
PT_DEV MaterialSample sampleMaterial(const PtxSpecularGlossinessMaterial *m, const MaterialTexels &t, bool isHitFromInside)
{
    MaterialSample ret;
    ret.EmissiveColor = (rgb(t.emissive) + ld3(m->EmissiveColor)) * m->EmissiveIntensity;
    ret.Color = rgb(t.color) * ld3(m->Color);
    ret.Normal = ReconstructNormalFromXY(rgb(t.normal));
    ret.Transmission = m->Transmission;
    ret.AttenuationColor = ld3(m->AttenuationColor);
    ret.AttenuationDistance = m->AttenuationDistance;
    ret.Eta = isHitFromInside ? m->Ior : (div_(1.0f, m->Ior));
    const f3 specular = rgb(t.a) * ld3(m->Specular);
    const float glossiness = t.b.w * m->Glossiness;
    ret.Roughness = 1.0f - glossiness;
    const f3 diff = specGlossMetalness(specular, ret.Color);
    ret.Metalness = div_(diff.x + diff.y + diff.z, 3.0f);
    return ret;
}
PT_DEV MaterialSample sampleMaterial(const PtxPhongMaterial *m, const MaterialTexels &t, bool isHitFromInside)
{
    MaterialSample ret;
    ret.EmissiveColor = (rgb(t.emissive) + ld3(m->EmissiveColor)) * m->EmissiveIntensity;
    ret.Color = rgb(t.color) * ld3(m->Color);
    ret.Normal = ReconstructNormalFromXY(rgb(t.normal));
    ret.Transmission = m->Transmission;
    ret.AttenuationColor = ld3(m->AttenuationColor);
    ret.AttenuationDistance = m->AttenuationDistance;
    ret.Eta = isHitFromInside ? m->Ior : (div_(1.0f, m->Ior));
    const f3 specular = rgb(t.a) * ld3(m->Specular);
    const float shininess = t.b.w * m->Shininess;
    ret.Roughness = 1.0f - shininess;
    const f3 diff = specGlossMetalness(specular, ret.Color);
    ret.Metalness = div_(diff.x + diff.y + diff.z, 3.0f);
    return ret;
}
// material.glsl:161-171: the unknown-type default (fields the GLSL leaves undefined are zero here) and flipNormalY
PT_DEV MaterialSample unknownMaterial()
{
    MaterialSample ret;
    ret.Normal = ret.AttenuationColor = F3s(0.0f);
    ret.Roughness = ret.Metalness = ret.Transmission = ret.Eta = ret.AttenuationDistance = 0.0f;
    ret.Color = F3(1.0f, 0.0f, 0.0f);
    ret.EmissiveColor = F3(1.0f, 0.0f, 0.0f);
    return ret;
}

template <bool TEX>
PT_DEV MaterialSample sampleMaterial(const SceneView &sv, uint32_t materialId, f2 texCoords, f4 derivatives, bool isHitFromInside) // :144-171
{
    const uint32_t materialType = materialId & 0xffu;
    const uint32_t materialIndex = materialId >> 8;
    MaterialSample ret;
    if (materialType > PTX_MATERIAL_TYPE_PHONG) // :163-166
        ret = unknownMaterial();
    else
    {
        // the five texture slots of the three material structs sit at the same offsets, in the order of MaterialTexels
        const PtxMetallicRoughnessMaterial *mr = &sv.mr[materialIndex];
        const PtxSpecularGlossinessMaterial *sg = &sv.sg[materialIndex];
        const PtxPhongMaterial *ph = &sv.phong[materialIndex];
        uint32_t i0, i1, i2, i3, i4;
        if (materialType == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS) // :62-84
        {
            i0 = mr->EmissiveIdx; i1 = mr->ColorIdx; i2 = mr->NormalIdx; i3 = mr->RoughnessIdx; i4 = mr->MetallicIdx;
        }
        else if (materialType == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS) // :86-113
        {
            i0 = sg->EmissiveIdx; i1 = sg->ColorIdx; i2 = sg->NormalIdx; i3 = sg->SpecularIdx; i4 = sg->GlossinessIdx;
        }
        else // :115-142
        {
            i0 = ph->EmissiveIdx; i1 = ph->ColorIdx; i2 = ph->NormalIdx; i3 = ph->SpecularIdx; i4 = ph->ShininessIdx;
        }
        MaterialTexels t;
        if (TEX)
        {
            // ONE copy of the sampler in the kernel, run five times, instead of five (fifteen over the three branches) inlined
            // copies: the textured shade kernel was 270 KB of code, several times the instruction cache, and its register
            // peak was the sampler's live state times the calls the scheduler overlapped.  The fetches have no side
            // effects, so running them before the branch's arithmetic changes nothing.
#pragma nounroll
            for (int k = 0; k < 5; k++)
            {
                const uint32_t idx = k == 0 ? i0 : k == 1 ? i1 : k == 2 ? i2 : k == 3 ? i3 : i4;
                const f4 c = sampleTexture<true>(sv.tex, idx, texCoords, derivatives);
                if (k == 0) t.emissive = c;
                else if (k == 1) t.color = c;
                else if (k == 2) t.normal = c;
                else if (k == 3) t.a = c;
                else t.b = c;
            }
        }
        else
        {
            t.emissive = sampleTexture(i0); t.color = sampleTexture(i1); t.normal = sampleTexture(i2); t.a = sampleTexture(i3); t.b = sampleTexture(i4);
        }
        if (materialType == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS)
            ret = sampleMaterial(mr, t, isHitFromInside);
        else if (materialType == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS)
            ret = sampleMaterial(sg, t, isHitFromInside);
        else
            ret = sampleMaterial(ph, t, isHitFromInside);
    }
    if (sv.dxNormalTextures)
        ret.Normal.y *= -1;
    return ret;
}

// ---- closestHit.rchit ---------------------------------------------------------------------------

struct Vtx
{
    f3 Position;
    f3 Normal, Tangent, Bitangent;
};

PT_DEV f3 xformPoint(const float *M, f3 p)
{
    return F3(((p.x * M[0] + p.y * M[1]) + p.z * M[2]) + M[3], ((p.x * M[4] + p.y * M[5]) + p.z * M[6]) + M[7],
              ((p.x * M[8] + p.y * M[9]) + p.z * M[10]) + M[11]);
}
PT_DEV f3 xformVector(const float *M, f3 p)
{
    return F3((p.x * M[0] + p.y * M[1]) + p.z * M[2], (p.x * M[4] + p.y * M[5]) + p.z * M[6],
              (p.x * M[8] + p.y * M[9]) + p.z * M[10]);
}

PT_DEV Vtx transformVertex(const DevPair &pr, Vtx v) // sampling.glsl:5-15
{
    v.Position = xformPoint(pr.M, v.Position);
    v.Tangent = normalize(xformVector(pr.M, v.Tangent));
    v.Bitangent = normalize(xformVector(pr.M, v.Bitangent));
    v.Normal = normalize(F3(dot(v.Normal, F3(pr.Rinv[0], pr.Rinv[1], pr.Rinv[2])), dot(v.Normal, F3(pr.Rinv[3], pr.Rinv[4], pr.Rinv[5])),
                            dot(v.Normal, F3(pr.Rinv[6], pr.Rinv[7], pr.Rinv[8]))));
    return v;
}

PT_DEV Vtx loadVertex(const PtxVertex *p) // common.glsl:27-46
{
    Vtx v;
    v.Position = ld3(p->Position);
    v.Normal = ld3(p->Normal);
    v.Tangent = ld3(p->Tangent);
    v.Bitangent = ld3(p->Bitangent);
    return v;
}

struct TriVertices // the vertices of the hit triangle as common.glsl:27-46 would fetch them, and their world-space corners
{
    Vtx o[3];
    f2 uv[3];
    f3 worldPosition[3], worldNormal[3], geometricNormal;
};

// closestHit.rchit:63-74 for one triangle: transformed corners and the unflipped geometric normal (see ShadeTri)
PT_DEV void worldCorners(const DevPair &pr, const Vtx &o0, const Vtx &o1, const Vtx &o2, f3 *worldPosition, f3 *worldNormal, f3 &geometricNormal);

PT_DEV TriVertices loadTriangle(const ShadeTri *st)
{
    float f[68];
    for (int k = 0; k < 17; k++)
    {
        const float4 q = st->v[k];
        f[4 * k] = q.x; f[4 * k + 1] = q.y; f[4 * k + 2] = q.z; f[4 * k + 3] = q.w;
    }
    TriVertices t;
    for (int k = 0; k < 3; k++)
    {
        t.worldPosition[k] = F3(f[44 + 3 * k], f[45 + 3 * k], f[46 + 3 * k]);
        t.worldNormal[k] = F3(f[53 + 3 * k], f[54 + 3 * k], f[55 + 3 * k]);
    }
    t.geometricNormal = F3(f[62], f[63], f[64]);
    for (int k = 0; k < 3; k++)
    {
        const float *p = &f[14 * k];
        t.o[k].Position = F3(p[0], p[1], p[2]);
        t.uv[k] = F2(p[3], p[4]);
        t.o[k].Normal = F3(p[5], p[6], p[7]);
        t.o[k].Tangent = F3(p[8], p[9], p[10]);
        t.o[k].Bitangent = F3(p[11], p[12], p[13]);
    }
    return t;
}

PT_DEV void worldCorners(const DevPair &pr, const Vtx &o0, const Vtx &o1, const Vtx &o2, f3 *worldPosition, f3 *worldNormal, f3 &geometricNormal)
{
    const Vtx v0 = transformVertex(pr, o0), v1 = transformVertex(pr, o1), v2 = transformVertex(pr, o2);
    const f3 edge1 = v1.Position - v0.Position;
    const f3 edge2 = v2.Position - v0.Position;
    geometricNormal = normalize(cross(edge1, edge2));
    worldPosition[0] = v0.Position; worldPosition[1] = v1.Position; worldPosition[2] = v2.Position;
    worldNormal[0] = v0.Normal; worldNormal[1] = v1.Normal; worldNormal[2] = v2.Normal;
}

PT_DEV f3 interp3(f3 a, f3 b, f3 c, f3 bc) { return (a * bc.x + b * bc.y) + c * bc.z; } // common.glsl:107-110

// What closestHit.rchit writes into the payload (ShaderRendererTypes.incl:101-118); the ray
// differentials (RayDifferentials0..2) travel separately as DiffRays, in the TEX variants only.
struct HitOut
{
    f3 Position;
    f3 Direction;
    float MaxRoughness;
    f3 Bsdf;
    float Pdf;
    f3 Emissive;
    f3 DirectLight;
    float DirectLightPdf;
    f3 LightDirection;
    float LightDistance;
};

// closestHit.rchit:52-161.  (u, v) = hitAttributeEXT barycentrics, t = gl_RayTmaxEXT.
template <bool TEX>
PT_DEV void closestHit(const SceneView &sv, f3 rayDirW, float t, float hu, float hv, uint32_t pairIdx, uint32_t slot,
                       float maxRoughnessIn, uint32_t &rngState, HitOut &out, DiffRays &diff, const Decal &decal)
{
    const f3 bary = F3(1.0f - hu - hv, hu, hv);
    const DevPair pr = sv.pairs[pairIdx];
    const TriVertices tv3 = loadTriangle(&sv.shadeTris[slot]);
    const Vtx o0 = tv3.o[0], o1 = tv3.o[1], o2 = tv3.o[2];

    Vtx ov; // getInterpolatedVertex, common.glsl:112-130
    ov.Position = interp3(o0.Position, o1.Position, o2.Position, bary);
    ov.Normal = interp3(o0.Normal, o1.Normal, o2.Normal, bary);
    ov.Tangent = interp3(o0.Tangent, o1.Tangent, o2.Tangent, bary);
    ov.Bitangent = interp3(o0.Bitangent, o1.Bitangent, o2.Bitangent, bary);
    Vtx vertex = transformVertex(pr, ov);

    // :63-74 the transformed corners and normalize(cross(edge1, edge2)), precomputed per triangle (ShadeTri)
    Vtx v0, v1, v2;
    v0.Position = tv3.worldPosition[0]; v1.Position = tv3.worldPosition[1]; v2.Position = tv3.worldPosition[2];
    v0.Normal = tv3.worldNormal[0]; v1.Normal = tv3.worldNormal[1]; v2.Normal = tv3.worldNormal[2];
    f3 geometricNormal = tv3.geometricNormal;

    const bool isHitFromInside = dot(geometricNormal, rayDirW) > 0.0f;
    if (isHitFromInside)
    {
        geometricNormal = -geometricNormal;
        vertex.Normal = -vertex.Normal;
        vertex.Tangent = -vertex.Tangent;
        vertex.Bitangent = -vertex.Bitangent;
    }

    // :88-99 texture footprint from the ray differentials
    f2 texCoords = F2(0.0f, 0.0f);
    f4 derivatives;
    derivatives.x = derivatives.y = derivatives.z = derivatives.w = 0.0f;
    f3 dndu = F3s(0.0f), dndv = F3s(0.0f);
    if (TEX)
    {
        const f2 uv0 = tv3.uv[0], uv1 = tv3.uv[1], uv2 = tv3.uv[2];
        texCoords = F2((uv0.x * bary.x + uv1.x * bary.y) + uv2.x * bary.z, (uv0.y * bary.x + uv1.y * bary.y) + uv2.y * bary.z);
        const f3 P3[3] = { v0.Position, v1.Position, v2.Position }, N3[3] = { v0.Normal, v1.Normal, v2.Normal };
        const f2 UV3[3] = { uv0, uv1, uv2 };
        f3 dpdu, dpdv, dpdx, dpdy;
        computeDpnDuv(P3, N3, UV3, vertex.Tangent, vertex.Bitangent, dpdu, dpdv, dndu, dndv);
        computeDpDxy(vertex.Position, diff.rxOrigin, diff.rxDirection, diff.ryOrigin, diff.ryDirection, vertex.Normal, dpdx, dpdy);
        derivatives = computeDerivatives(dpdx, dpdy, dpdu, dpdv);
    }

    MaterialSample material = sampleMaterial<TEX>(sv, pr.materialId, texCoords, derivatives, isHitFromInside);

    // :105-106 decals: an ignored alpha < 0.5 candidate in front of this hit tints the base colour
    // (the any-hit stage only remembered WHICH candidate it ignored: its colour is fetched here, when it matters; the
    // traversal kernel carries six decal words instead of a colour, and its sampler call keeps the alpha only)
    if (TEX && decal.dist != -1.0f && t > decal.dist) // non-opaque geometry implies the TEX variants (kernelMode)
    {
        const f4 dc = hitBaseColor(sv, decal.pair, decal.slot, decal.u, decal.v);
        material.Color = mix(material.Color, rgb(dc), dc.w);
    }

    out.MaxRoughness = fmax_(material.Roughness, maxRoughnessIn); // :109
    material.Roughness = fmax_(out.MaxRoughness, 0.01f);          // :112

    mat3 geometryTBN;
    geometryTBN.c0 = vertex.Tangent;
    geometryTBN.c1 = vertex.Bitangent;
    geometryTBN.c2 = vertex.Normal;
    const f3 N = normalize(vertex.Normal + mul(geometryTBN, material.Normal));
    const mat3 TBN = computeTangentSpace(N);
    const mat3 invTBN = inverse(TBN);
    const f3 V = normalize(mul(invTBN, normalize(-rayDirW)));

    BSDFSample bsdf = sampleBSDF(material, V, rngState);

    if (isHitFromInside) // :123-128
    {
        const float e = div_(t, material.AttenuationDistance);
        bsdf.Color.x *= pow_(material.AttenuationColor.x, e);
        bsdf.Color.y *= pow_(material.AttenuationColor.y, e);
        bsdf.Color.z *= pow_(material.AttenuationColor.z, e);
    }

    const bool isRefracted = bsdf.Direction.z < 0.0f;
    const f3 rayOrigin = offsetRayOriginShadowTerminator(vertex.Position, v0.Position, v0.Normal, v1.Position, v1.Normal,
                                                         v2.Position, v2.Normal, bary, isRefracted);

    float lightPdf, lightSmplPdf;
    f3 u3;
    u3.x = rnd(rngState);
    u3.y = rnd(rngState);
    u3.z = rnd(rngState);
    const LightSample light = sampleLight(sv.lights, u3, rayOrigin, lightPdf);
    const f3 L = normalize(mul(invTBN, -light.Direction));
    const f3 lightBsdf = evaluateBSDF(material, V, L, lightSmplPdf);

    out.Direction = normalize(mul(TBN, bsdf.Direction));
    if (isRefracted)
        out.Position = offsetRayOriginSelfIntersection(vertex.Position, -geometricNormal);
    else
        out.Position = rayOrigin;
    out.Bsdf = bsdf.Color;
    out.Pdf = bsdf.Pdf;
    out.Emissive = material.EmissiveColor;
    out.DirectLight = (light.Color * light.Attenuation) * lightBsdf;
    out.DirectLightPdf = lightPdf;
    out.LightDirection = light.Direction;
    out.LightDistance = light.Distance;

    if (TEX) // :150-160 differentials of the continuation ray
    {
        if (isRefracted)
            computeRefractedDifferentialRays(derivatives, vertex.Normal, rayOrigin, -rayDirW, out.Direction, dndu, dndv, material.Eta, diff);
        else
            computeReflectedDifferentialRays(derivatives, vertex.Normal, rayOrigin, -rayDirW, out.Direction, dndu, dndv, diff);
    }
}

// ---- ray / triangle --------------------------------------------------------------------------------

// Moeller-Trumbore on (v0, e1, e2): the fixed stand-in for the driver's unspecified
// ray-triangle test.  Hit iff det != 0, 0 <= u <= 1, v >= 0, u+v <= 1, tmin < t < tmax.
//
// Two passes.  Plain Moeller-Trumbore from the ray origin loses (|o - v0| / size)^2 ulps in the barycentrics: 23 units
// from the camera it accepted a point 16 % of a 2 cm triangle's extent outside it -- a "hit" whose ray does not even
// cross the triangle's bounding box, on which brute force and tree walks cannot agree (tools/full_size_sweep.py,
// street_like).  Its t, however, is good.  So pass 1 solves from o with loose bounds on (u, v), and pass 2 solves again
// from o + t1 d, where the offset to v0 is at most the triangle's size and the correction to t is tiny: accurate for far
// small triangles AND for near hits on huge ones (re-originating through v0 instead loses t > tmin for those).
PT_DEV bool intersectTri(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float tmin, float tmax, float &t, float &u, float &v)
{
    const f3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    if (!(det != 0.0f))
        return false;
    const float inv = div_(1.0f, det);
    // pass 1 (plain, from the ray origin): good t, barycentrics off by (|o - v0| / size)^2 ulps -> loose bounds only
    f3 tvec = o - v0;
    float uu = dot(tvec, pvec) * inv;
    if (!(uu >= -0.25f && uu <= 1.25f))
        return false;
    f3 qvec = cross(tvec, e1);
    float vv = dot(d, qvec) * inv;
    if (!(vv >= -0.25f && uu + vv <= 1.25f))
        return false;
    const float t1 = dot(e2, qvec) * inv;
    if (!(t1 > -1e30f && t1 < 1e30f))
        return false;
    // pass 2 from o + t1 d: the offset to v0 is at most the triangle's size, the correction to t is tiny
    tvec = (o + d * t1) - v0;
    uu = dot(tvec, pvec) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f))
        return false;
    qvec = cross(tvec, e1);
    vv = dot(d, qvec) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f))
        return false;
    const float tt = t1 + dot(e2, qvec) * inv;
    if (!(tt > tmin && tt < tmax))
        return false;
    t = tt;
    u = uu;
    v = vv;
    return true;
}

} // namespace ptd
