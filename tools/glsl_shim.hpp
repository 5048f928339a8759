// glsl_shim.hpp -- minimal GLSL-builtin shim so the reference's pure-math shader
// headers (shading.glsl, bsdf.glsl, parts of common/ray/sampling.glsl) compile as C++.
//
// Used ONLY by tools/gen_golden.py in the build container to produce the golden
// vectors under tests/golden/.  It never travels into the product, and the shader text
// it is applied to stays in /tmp (reference sources are not copied into this repo).
//
// Builtins follow the arithmetic conventions written in oracle/pt_oracle.h.  With
// -DSHIM_FIXED the transcendental builtins use the fixed polynomial kernels and `/` is
// the specified division a * rcp(b) (bit-exact structure check); without it they use
// glibc's libm and IEEE division (independent check of those kernels and of the division
// convention, compared within a few ULP).
//
// GLSL's `float` is the class glsl::Float here (every later `float` token -- the shim's own
// types, the reference's shader text, tools/golden_main.inc -- is that class through the macro
// at the end of this prologue), so that a scalar `a / b` of the shader text goes through an
// operator this file defines; +, -, * and the comparisons are plain IEEE binary32.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../oracle/pt_oracle_math.h"

namespace glsl
{
typedef float f32; // the machine type, for the few places that need it
template <class T> concept Arith = std::is_arithmetic_v<T>;

struct Float
{
    f32 v;
    Float() = default;
    constexpr Float(f32 x) : v(x) {}
    constexpr Float(double x) : v((f32)x) {}
    constexpr Float(int x) : v((f32)x) {}
    constexpr Float(unsigned x) : v((f32)x) {}
    constexpr Float(long x) : v((f32)x) {}
    constexpr Float(unsigned long x) : v((f32)x) {}
    constexpr operator f32() const { return v; }
    Float &operator+=(Float o) { v = v + o.v; return *this; }
    Float &operator-=(Float o) { v = v - o.v; return *this; }
    Float &operator*=(Float o) { v = v * o.v; return *this; }
    Float &operator/=(Float o);
};
inline f32 shimDiv(f32 a, f32 b)
{
#ifdef SHIM_FIXED
    return a * pto_rcp(b); // the specified division (oracle/pt_oracle_math.h)
#else
    return a / b; // IEEE: the independent check
#endif
}
inline Float &Float::operator/=(Float o) { v = shimDiv(v, o.v); return *this; }
inline Float operator-(Float a) { return Float(-a.v); }
inline Float operator+(Float a) { return a; }
#define SHIM_BINOP(op, expr)                                                                          \
    inline Float operator op(Float a, Float b) { const f32 x = a.v, y = b.v; return Float(expr); }    \
    template <Arith T> inline Float operator op(Float a, T b) { const f32 x = a.v, y = (f32)b; return Float(expr); } \
    template <Arith T> inline Float operator op(T a, Float b) { const f32 x = (f32)a, y = b.v; return Float(expr); }
SHIM_BINOP(+, x + y)
SHIM_BINOP(-, x - y)
SHIM_BINOP(*, x * y)
SHIM_BINOP(/, shimDiv(x, y))
#undef SHIM_BINOP
} // namespace glsl

#define float ::glsl::Float // from here on (all standard headers are in)

namespace glsl
{
typedef unsigned int uint;

struct uvec2
{
    uint x, y;
    uvec2() {}
    uvec2(uint a, uint b) : x(a), y(b) {}
    uvec2 xy() const { return *this; }
};
struct ivec2
{
    int x, y;
    ivec2() {}
    explicit ivec2(uvec2 u) : x((int)u.x), y((int)u.y) {}
};
struct vec2
{
    union { float x; float r; }; union { float y; float g; };
    vec2() {}
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float a, float b) : x(a), y(b) {}
    explicit vec2(uvec2 u) : x((float)u.x), y((float)u.y) {}
    vec2 xy() const { return *this; }
};
struct vec3
{
    union { float x; float r; }; union { float y; float g; }; union { float z; float b; };
    vec3() {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b_, float c) : x(a), y(b_), z(c) {}
    vec3(vec2 a, float c) : x(a.x), y(a.y), z(c) {}
    vec3(float a, vec2 bc) : x(a), y(bc.x), z(bc.y) {}
    vec3 xyz() const { return *this; }
    vec3 rgb() const { return *this; }
    vec2 xy() const { return vec2(x, y); }
    vec2 yz() const { return vec2(y, z); }
    vec3 &operator+=(vec3 o) { x = x + o.x; y = y + o.y; z = z + o.z; return *this; }
    vec3 &operator-=(vec3 o) { x = x - o.x; y = y - o.y; z = z - o.z; return *this; }
    vec3 &operator*=(float s) { x = x * s; y = y * s; z = z * s; return *this; }
    vec3 &operator*=(vec3 o) { x = x * o.x; y = y * o.y; z = z * o.z; return *this; }
    vec3 &operator/=(float s) { x = x / s; y = y / s; z = z / s; return *this; }
};
struct ivec3
{
    int x, y, z;
    ivec3() {}
    explicit ivec3(vec3 v) : x((int)v.x), y((int)v.y), z((int)v.z) {}
};
struct vec4
{
    union { float x; float r; }; union { float y; float g; }; union { float z; float b; }; union { float w; float a; };
    vec4() {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float a_, float b_, float c, float d) : x(a_), y(b_), z(c), w(d) {}
    vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    vec4(vec2 a_, vec2 b_) : x(a_.x), y(a_.y), z(b_.x), w(b_.y) {}
    vec4(float a_, vec3 v) : x(a_), y(v.x), z(v.y), w(v.z) {}
    vec3 yzw() const { return vec3(y, z, w); }
    vec3 xyz() const { return vec3(x, y, z); }
    vec3 rgb() const { return vec3(x, y, z); }
    vec2 xy() const { return vec2(x, y); }
    vec2 zw() const { return vec2(z, w); }
};
struct mat3
{
    vec3 c[3];
    mat3() {}
    mat3(vec3 a, vec3 b, vec3 d) { c[0] = a; c[1] = b; c[2] = d; }
};
struct mat3x4 // 3 columns of 4 rows (the transposed affine transform the reference stores per mesh)
{
    vec4 c[3];
    mat3x4() {}
};
struct mat4
{
    float m[16]; // [col*4+row]
    mat4() {}
    explicit mat4(const mat3x4 &a) // GLSL: the missing column comes from the identity
    {
        for (int k = 0; k < 3; k++) { m[k * 4] = a.c[k].x; m[k * 4 + 1] = a.c[k].y; m[k * 4 + 2] = a.c[k].z; m[k * 4 + 3] = a.c[k].w; }
        m[12] = m[13] = m[14] = 0.0f; m[15] = 1.0f;
    }
};

// ---- operators -------------------------------------------------------------
inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator+(uvec2 a, vec2 b) { return vec2((float)a.x + b.x, (float)a.y + b.y); }
inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator-(vec2 a, float s) { return vec2(a.x - s, a.y - s); }
inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
inline vec2 operator/(vec2 a, float s) { return vec2(a.x / s, a.y / s); }
inline vec2 operator+(vec2 a, float s) { return vec2(a.x + s, a.y + s); }
inline vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
inline vec2 operator/(vec2 a, vec2 b) { return vec2(a.x / b.x, a.y / b.y); }
inline vec2 operator/(vec2 a, uvec2 b) { return vec2(a.x / (float)b.x, a.y / (float)b.y); }
inline bool operator==(vec2 a, vec2 b) { return a.x == b.x && a.y == b.y; }

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }
inline vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator/(vec3 a, uint s) { return a / (float)s; } // implicit uint -> float conversion
inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 &operator/=(vec3 &a, vec3 b) { a = a / b; return a; }

inline vec3 operator*(const mat3 &m, vec3 v)
{
    return vec3((m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z, (m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z,
                (m.c[0].z * v.x + m.c[1].z * v.y) + m.c[2].z * v.z);
}
inline vec4 operator*(const mat4 &m, vec4 v)
{
    const v4 r = m4_mul(reinterpret_cast<const f32 *>(m.m), v.x, v.y, v.z, v.w);
    return vec4(r.x, r.y, r.z, r.w);
}

inline vec4 operator*(vec4 a, vec4 b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline float dot(vec4 a, vec4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
inline vec3 operator*(vec4 v, const mat3x4 &m) { return vec3(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])); } // row vector x matrix
inline vec4 operator*(vec4 v, const mat4 &m)
{
    vec4 r;
    float *o = &r.x;
    for (int c = 0; c < 4; c++)
        o[c] = dot(v, vec4(m.m[c * 4], m.m[c * 4 + 1], m.m[c * 4 + 2], m.m[c * 4 + 3]));
    return r;
}
inline mat3x4 operator*(const mat4 &a, const mat3x4 &b) // (4 x 4) . (4 rows x 3 columns)
{
    mat3x4 r;
    for (int c = 0; c < 3; c++)
    {
        const vec4 col = a * b.c[c];
        r.c[c] = col;
    }
    return r;
}
inline mat4 transpose(const mat4 &a)
{
    mat4 r;
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < 4; k++)
            r.m[c * 4 + k] = a.m[k * 4 + c];
    return r;
}
inline mat4 inverse(const mat4 &a) // cofactor expansion
{
    const float *m = a.m;
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    mat4 r;
    for (int k = 0; k < 16; k++)
        r.m[k] = inv[k] / det;
    return r;
}

// ---- builtins --------------------------------------------------------------
inline float abs(float x) { return fabsf(x); }
inline float sqrt(float x) { return sqrtf(x); }
inline glsl::f32 shimRsq(glsl::f32 x)
{
#ifdef SHIM_FIXED
    return pto_rsq(x); // the specified reciprocal square root (oracle/pt_oracle_math.h)
#else
    return 1.0f / sqrtf(x); // IEEE: the independent check
#endif
}
inline float inversesqrt(float x) { return float(shimRsq(x.v)); }
inline float min(float a, float b) { return f_min(a, b); }
inline float max(float a, float b) { return f_max(a, b); }
inline vec3 max(vec3 a, float b) { return vec3(f_max(a.x, b), f_max(a.y, b), f_max(a.z, b)); }
inline float clamp(float x, float lo, float hi) { return f_clamp(x, lo, hi); }
inline bool isinf(float x) { return std::isinf(x.v); }
inline bool isnan(float x) { return std::isnan(x.v); }
inline float fma(float a, float b, float c) { return fmaf(a, b, c); }
inline int floatBitsToInt(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float intBitsToFloat(int i) { float f; memcpy(&f, &i, 4); return f; }
inline float uintBitsToFloat(uint u) { float f; memcpy(&f, &u, 4); return f; }

#ifdef SHIM_FIXED
inline float cos(float x) { f32 s, c; pto_sincosf(x, &s, &c); return c; }
inline float sin(float x) { f32 s, c; pto_sincosf(x, &s, &c); return s; }
inline float pow(float x, float y)
{
    if (y == 2.0f) return x * x;
    if (y == 5.0f) { const float x2 = x * x; return x2 * x2 * x; }
    return pto_powf(x, y);
}
#else
inline float cos(float x) { return cosf(x); }
inline float sin(float x) { return sinf(x); }
inline float pow(float x, float y) { return powf(x, y); }
#endif
template <class T> requires std::is_integral_v<T> inline float pow(float x, T y) { return pow(x, (float)y); }
#ifdef SHIM_FIXED
inline float exp(float x) { return pto_expf(x); }
#else
inline float exp(float x) { return expf(x); }
#endif
#ifdef SHIM_FIXED
inline float atan(float y, float x) { return pto_atan2f(y, x); }
inline float asin(float x) { return pto_asinf(x); }
#else
inline float atan(float y, float x) { return atan2f(y, x); }
inline float asin(float x) { return asinf(x); }
#endif
#ifdef SHIM_FIXED
inline float log2(float x) { return (float)pto_log2((double)x); }
#else
inline float log2(float x) { return log2f(x); }
#endif

inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(uvec2 a, uvec2 b) { return (float)a.x * (float)b.x + (float)a.y * (float)b.y; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline float distance(vec3 a, vec3 b) { return length(a - b); }
inline vec3 normalize(vec3 a) { return a * float(shimRsq(dot(a, a).v)); }
inline vec3 reflect(vec3 I, vec3 N) { return I - N * (2.0f * dot(N, I)); }
inline vec3 refract(vec3 I, vec3 N, float eta)
{
    const float d = dot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f)
        return vec3(0.0f);
    return I * eta - N * (eta * d + sqrtf(k));
}
inline vec3 exp(vec3 v) { return vec3(exp(v.x), exp(v.y), exp(v.z)); }
inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
inline mat3 inverse(const mat3 &m)
{
    m3 a;
    a.c0 = V3(m.c[0].x, m.c[0].y, m.c[0].z);
    a.c1 = V3(m.c[1].x, m.c[1].y, m.c[1].z);
    a.c2 = V3(m.c[2].x, m.c[2].y, m.c[2].z);
    const m3 r = m3_inverse(a);
    return mat3(vec3(r.c0.x, r.c0.y, r.c0.z), vec3(r.c1.x, r.c1.y, r.c1.z), vec3(r.c2.x, r.c2.y, r.c2.z));
}
} // namespace glsl
