/*
 * pt_oracle_math.h -- vector helpers and the GLSL builtins the shaders use, with the
 * arithmetic conventions fixed in pt_oracle.h.  ORACLE ONLY (test infrastructure).
 */
#ifndef PT_ORACLE_MATH_H
#define PT_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct v2 { float x, y; } v2;
typedef struct v3 { float x, y, z; } v3;
typedef struct v4 { float x, y, z, w; } v4;
typedef struct m3 { v3 c0, c1, c2; } m3; /* columns, like GLSL mat3 */

#define PTO_PI 3.14159265359f /* common.glsl:3 */

static inline float f_min(float a, float b) { return (b < a) ? b : a; } /* GLSL min: y < x ? y : x */
static inline float f_max(float a, float b) { return (a < b) ? b : a; } /* GLSL max: x < y ? y : x */
static inline float f_clamp(float x, float lo, float hi) { return f_min(f_max(x, lo), hi); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- the specified division (arithmetic conventions, pt_oracle.h) ----
 * GLSL leaves `/` to the implementation within 2.5 ULP for divisors in [2^-126, 2^126] and lets denormals be flushed
 * (Vulkan 1.3 spec, "Precision and Operation of SPIR-V Instructions").  Every division of the shader path is
 *     a / b  :=  a * rcp(b),      rcp(b) = the correctly rounded reciprocal RN(1 / b) for 2^-126 <= |b| <= 2^126,
 *                                          +-inf for |b| < 2^-126 (zero and denormal divisors), +-0 for |b| > 2^126, NaN for NaN
 * -- two roundings, <= 1.5 ULP.  Where the product's value is NOT the IEEE quotient's (all stated, all tested on both sides:
 * tests/test_oracle_golden.py::test_specified_division_corner_cases_are_the_stated_ones): 0 / (zero or denormal) = 0 * inf = NaN
 * (under the denormal flush GLSL grants, 0 / denormal IS 0 / 0), inf / (|b| > 2^126) = NaN, x / x = 0 for |x| > 2^126 and inf
 * for denormal x.  A NaN or infinity that reaches a sample is rejected by the shader itself (raygen.rgen:99-112: the launch's
 * samples restart), so none of these can stay in the accumulation image.  On the device that is v_rcp_f32 + one FMA Newton step (5 VALU instead of the 11 of a
 * correctly rounded quotient; a vector / scalar is ONE reciprocal and three multiplies); the step makes the result independent
 * of the hardware seed: device == this definition on all 2^32 inputs (tools/experiments/rcp_sqrt_exhaustive.hip,
 * profiles/r05_rcp_sqrt_exhaustive.txt; tests/test_gpu_parity.py::test_specified_reciprocal_on_all_inputs). */
static inline float pto_rcp(float x)
{
    uint32_t ab;
    memcpy(&ab, &x, 4);
    ab &= 0x7fffffffu;
    if (ab - 0x00800000u <= 0x7e800000u - 0x00800000u) /* 2^-126 <= |x| <= 2^126: one range test on the bits */
        return 1.0f / x;
    if (x != x)
        return x;
    return ab < 0x00800000u ? copysignf(INFINITY, x) : copysignf(0.0f, x); /* zero / denormal divisor; |x| > 2^126 incl. infinity */
}
static inline float pto_div(float a, float b) { return a * pto_rcp(b); }

/* ---- the specified reciprocal square root (round 6) ----
 * GLSL grants inversesqrt 2 ULP and normalize() the precision of its expansion; both are
 *     normalize(v) := v * rsq(dot(v, v)),   inversesqrt(x) := rsq(x),
 *     rsq(x) = the correctly rounded RN(1 / sqrt(x)) for positive normal x (2^-126 <= x <= FLT_MAX),
 *              +-inf for +-0 and +-denormal inputs (flushed), +0 for +inf, NaN for negative normal numbers, -inf and NaN.
 * On the positive normal range (float)(1.0 / sqrt((double)x)) IS the correctly rounded value: 1 / sqrt(x * 4^k) = 2^-k / sqrt(x), so
 * 2^24 (mantissa, exponent parity) classes decide it for every input, and pto_rsq_selfcheck() compares all of them with the exact
 * answer in 128-bit integer arithmetic (x * m^2 against 1 at both neighbouring midpoints m; tests/test_oracle_golden.py).
 * On the device: v_rsq_f32 (1 ULP) + ONE compensated Newton step with the second-order term, 8 VALU + 2 for the range select instead
 * of the 21 of rcp(sqrt(x)) -- equal to this definition for every seed within 1 ULP (shown on the CPU for all 2^24 classes x 3
 * seeds) and on all 2^32 inputs on the hardware (tools/experiments/rcp_sqrt_exhaustive.hip, profiles/r06_rsq_exhaustive.txt). */
static inline float pto_rsq(float x)
{
    uint32_t b;
    memcpy(&b, &x, 4);
    if (b - 0x00800000u < 0x7f800000u - 0x00800000u) /* positive normal */
        return (float)(1.0 / sqrt((double)x));
    if (x != x)
        return x;
    if ((b & 0x7fffffffu) < 0x00800000u) /* +-0, +-denormal */
        return copysignf(INFINITY, x);
    return b == 0x7f800000u ? 0.0f : NAN; /* +inf; negative normal numbers and -inf */
}

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3s(float s) { return V3(s, s, s); }
static inline v3 v_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v_div(v3 a, float s) { const float r = pto_rcp(s); return V3(a.x * r, a.y * r, a.z * r); }
static inline v3 v_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float v_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v_cross(v3 a, v3 b)
{
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float v_length(v3 a) { return sqrtf(v_dot(a, a)); }
static inline v3 v_normalize(v3 a) { return v_scale(a, pto_rsq(v_dot(a, a))); }
/* reflect(I, N) = I - 2 * dot(N, I) * N */
static inline v3 v_reflect(v3 I, v3 N) { return v_sub(I, v_scale(N, 2.0f * v_dot(N, I))); }
/* refract(I, N, eta), GLSL 4.60 8.5 */
static inline v3 v_refract(v3 I, v3 N, float eta)
{
    const float d = v_dot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f)
        return V3(0.0f, 0.0f, 0.0f);
    return v_sub(v_scale(I, eta), v_scale(N, eta * d + sqrtf(k)));
}
/* mix(x, y, a) = x * (1 - a) + y * a */
static inline v3 v_mix(v3 x, v3 y, float a) { return v_add(v_scale(x, 1.0f - a), v_scale(y, a)); }

static inline v3 m3_mul(m3 m, v3 v)
{
    return V3((m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z, (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z,
              (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z);
}

/* inverse(mat3): cofactors times 1/det, element [col][row] */
static inline m3 m3_inverse(m3 m)
{
    const float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z;
    const float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z;
    const float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z;
    const float det = (m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02)) + m20 * (m01 * m12 - m11 * m02);
    const float id = pto_rcp(det);
    m3 r;
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

/* mat4 (glm column-major, element [c*4+r]) times vec4 */
static inline v4 m4_mul(const float *m, float x, float y, float z, float w)
{
    v4 r;
    r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
    r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
    r.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
    return r;
}

/* ---- fixed transcendental kernels (shared definition with the HIP kernels) ---- */

/* sin and cos for |x| <= ~8 (the shaders call them on [-pi/4, 2pi]): Cody-Waite
 * reduction by pi/2 in three float pieces, then the degree-7/8 minimax polynomials. */
static inline void pto_sincosf(float x, float *s, float *c)
{
    const float fk = floorf(x * 0.636619772f + 0.5f);
    const int k = (int)fk;
    float r = x - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188216e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    switch (k & 3)
    {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
    }
}

/* log2 of a positive, finite, normal double */
static inline double pto_log2(double x)
{
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &bits, 8);
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

/* 2^t for |t| <= 300 */
static inline double pto_exp2(double t)
{
    const double k = floor(t + 0.5);
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
    const uint64_t bits = (uint64_t)((int64_t)k + 1023) << 52;
    double sc;
    memcpy(&sc, &bits, 8);
    return p * sc;
}

/* pow(x, y) for x >= 0 (GLSL leaves x < 0 undefined; NaN here) */
static inline float pto_powf(float x, float y)
{
    if (y == 0.0f || x == 1.0f)
        return 1.0f;
    if (x != x || y != y || x < 0.0f)
        return u2f(0x7fc00000u);
    if (x == 0.0f)
        return y > 0.0f ? 0.0f : u2f(0x7f800000u);
    if (x == u2f(0x7f800000u))
        return y > 0.0f ? u2f(0x7f800000u) : 0.0f;
    double t = (double)y * pto_log2((double)x);
    if (t > 300.0)
        t = 300.0;
    if (t < -300.0)
        t = -300.0;
    return (float)pto_exp2(t);
}

/* atan(y, x) over the full circle and asin(x): fixed double-precision kernels (GLSL leaves the
 * builtins' accuracy to the implementation).  |t| <= tan(pi/8) after the two reductions, where the
 * odd Taylor series to t^23 is below double rounding of a float result. */
static inline float pto_atan2f(float yf, float xf)
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

static inline float pto_asinf(float xf)
{
    double x = (double)xf;
    if (x > 1.0)
        x = 1.0;
    if (x < -1.0)
        x = -1.0;
    const double c = sqrt((1.0 - x) * (1.0 + x));
    const double ax = c, ay = x < 0.0 ? -x : x; /* atan2(x, c) with c >= 0, in double throughout */
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

/* sRGB transfer functions and 8-bit quantisation shared by the texture pipeline and the output stage */
static inline float srgbToLinear(float c) { return c <= 0.04045f ? c / 12.92f : pto_powf((c + 0.055f) / 1.055f, 2.4f); }
static inline float linearToSrgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * pto_powf(c, 1.0f / 2.4f) - 0.055f; }
static inline uint32_t quantize8(float x)
{
    if (!(x > 0.0f))
        return 0u;
    if (x > 1.0f)
        x = 1.0f;
    return (uint32_t)floorf(x * 255.0f + 0.5f);
}

/* exp(x) through the fixed exp2 kernel */
static inline float pto_expf(float x)
{
    if (x != x)
        return x;
    double t = (double)x * 1.4426950408889634;
    if (t > 300.0)
        t = 300.0;
    if (t < -300.0)
        t = -300.0;
    return (float)pto_exp2(t);
}

/* float -> IEEE binary16 (round to nearest even, overflow to infinity, subnormals kept) -> float:
 * the value an rgba16f image holds after imageStore. */
static inline float pto_f16_round(float f)
{
    const uint32_t x = f2u(f), sign = x & 0x80000000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) /* inf / nan */
        return ax > 0x7f800000u ? u2f(sign | 0x7fc00000u) : f;
    if (ax >= 0x477ff000u) /* >= 65520: rounds to infinity */
        return u2f(sign | 0x7f800000u);
    if (ax < 0x38800000u) /* below 2^-14: half subnormal, quantum 2^-24 */
    {
        const float q = u2f(ax) * 16777216.0f; /* exact */
        const float rq = (q + 12582912.0f) - 12582912.0f; /* round to nearest even integer (|q| < 2^10) */
        return u2f(sign | f2u(rq * (1.0f / 16777216.0f)));
    }
    const uint32_t lsb = (ax >> 13) & 1u;
    const uint32_t r = (ax + 0x0fffu + lsb) & 0xffffe000u;
    return u2f(sign | r);
}

#endif
