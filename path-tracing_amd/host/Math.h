// Math.h -- the small part of glm the reference's host side relies on, restated
// without glm (the glm submodule is absent: /root/reference/.gitmodules:4-6).
//
// Mat4 holds the MATH matrix row-major: m[r][c].  The reference stores
// glm::transpose(M) in column-major glm::mat4 (SceneGraph.cpp:30-32,
// ExampleScenes.cpp:497-499), which is the same memory image, so the first 12 floats
// of a Mat4 are exactly a VkTransformMatrixKHR / glm::mat3x4 (AccelerationStructure.cpp:271).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace PathTracing
{

struct Vec2
{
    float x = 0, y = 0;
};

struct Vec3
{
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit Vec3(float s) : x(s), y(s), z(s) {}
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 operator*(Vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline Vec3 operator*(float s, Vec3 a) { return { a.x * s, a.y * s, a.z * s }; }
inline Vec3 operator-(Vec3 a) { return { -a.x, -a.y, -a.z }; }
inline float Dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 Cross(Vec3 a, Vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline float Length(Vec3 a) { return std::sqrt(Dot(a, a)); }
inline Vec3 Normalize(Vec3 a) { return a * (1.0f / std::sqrt(Dot(a, a))); }
inline float Radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

struct Mat4
{
    float m[4][4];

    static Mat4 Identity()
    {
        Mat4 r;
        std::memset(r.m, 0, sizeof(r.m));
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
        return r;
    }
};

inline Mat4 operator*(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}

// glm::translate(M, v) = M * T(v)
inline Mat4 Translate(const Mat4 &M, Vec3 v)
{
    Mat4 t = Mat4::Identity();
    t.m[0][3] = v.x;
    t.m[1][3] = v.y;
    t.m[2][3] = v.z;
    return M * t;
}

// glm::scale(M, v) = M * S(v)
inline Mat4 Scale(const Mat4 &M, Vec3 v)
{
    Mat4 s = Mat4::Identity();
    s.m[0][0] = v.x;
    s.m[1][1] = v.y;
    s.m[2][2] = v.z;
    return M * s;
}

// glm::rotate(M, angle, axis) = M * R(angle, axis)
inline Mat4 Rotate(const Mat4 &M, float angle, Vec3 v)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const Vec3 axis = Normalize(v);
    const Vec3 temp = axis * (1.0f - c);
    Mat4 r = Mat4::Identity();
    r.m[0][0] = c + temp.x * axis.x;
    r.m[1][0] = temp.x * axis.y + s * axis.z;
    r.m[2][0] = temp.x * axis.z - s * axis.y;
    r.m[0][1] = temp.y * axis.x - s * axis.z;
    r.m[1][1] = c + temp.y * axis.y;
    r.m[2][1] = temp.y * axis.z + s * axis.x;
    r.m[0][2] = temp.z * axis.x + s * axis.y;
    r.m[1][2] = temp.z * axis.y - s * axis.x;
    r.m[2][2] = c + temp.z * axis.z;
    return M * r;
}

// glm::quat (w, x, y, z) with the three operations the animation system uses
struct Quat
{
    float w = 1.0f, x = 0.0f, y = 0.0f, z = 0.0f;
};

// glm::mix for vec3
inline Vec3 Mix(Vec3 a, Vec3 b, float t)
{
    return a * (1.0f - t) + b * t;
}

// glm::slerp: shortest arc, linear blend when the quaternions are nearly parallel
inline Quat Slerp(const Quat &x, const Quat &y, float a)
{
    Quat z = y;
    float cosTheta = x.w * y.w + x.x * y.x + x.y * y.y + x.z * y.z;
    if (cosTheta < 0.0f)
    {
        z = { -y.w, -y.x, -y.y, -y.z };
        cosTheta = -cosTheta;
    }
    if (cosTheta > 1.0f - 1.1920929e-7f)
        return { x.w * (1.0f - a) + z.w * a, x.x * (1.0f - a) + z.x * a, x.y * (1.0f - a) + z.y * a, x.z * (1.0f - a) + z.z * a };
    const float angle = std::acos(cosTheta);
    const float s0 = std::sin((1.0f - a) * angle), s1 = std::sin(a * angle), d = std::sin(angle);
    return { (s0 * x.w + s1 * z.w) / d, (s0 * x.x + s1 * z.x) / d, (s0 * x.y + s1 * z.y) / d, (s0 * x.z + s1 * z.z) / d };
}

// glm::mat4_cast
inline Mat4 ToMat4(const Quat &q)
{
    const float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z, qwx = q.w * q.x,
                qwy = q.w * q.y, qwz = q.w * q.z;
    Mat4 r = Mat4::Identity();
    r.m[0][0] = 1.0f - 2.0f * (qyy + qzz); r.m[1][0] = 2.0f * (qxy + qwz); r.m[2][0] = 2.0f * (qxz - qwy);
    r.m[0][1] = 2.0f * (qxy - qwz); r.m[1][1] = 1.0f - 2.0f * (qxx + qzz); r.m[2][1] = 2.0f * (qyz + qwx);
    r.m[0][2] = 2.0f * (qxz + qwy); r.m[1][2] = 2.0f * (qyz - qwx); r.m[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return r;
}

// glm::angleAxis
inline Quat AngleAxis(float angle, Vec3 axis)
{
    const float s = std::sin(angle * 0.5f);
    const Vec3 a = Normalize(axis);
    return { std::cos(angle * 0.5f), a.x * s, a.y * s, a.z * s };
}

// general 4x4 inverse by cofactors (glm::inverse)
inline Mat4 Inverse(const Mat4 &a)
{
    const float *s = &a.m[0][0];
    float inv[16];
    inv[0] = s[5] * s[10] * s[15] - s[5] * s[11] * s[14] - s[9] * s[6] * s[15] + s[9] * s[7] * s[14] + s[13] * s[6] * s[11] - s[13] * s[7] * s[10];
    inv[4] = -s[4] * s[10] * s[15] + s[4] * s[11] * s[14] + s[8] * s[6] * s[15] - s[8] * s[7] * s[14] - s[12] * s[6] * s[11] + s[12] * s[7] * s[10];
    inv[8] = s[4] * s[9] * s[15] - s[4] * s[11] * s[13] - s[8] * s[5] * s[15] + s[8] * s[7] * s[13] + s[12] * s[5] * s[11] - s[12] * s[7] * s[9];
    inv[12] = -s[4] * s[9] * s[14] + s[4] * s[10] * s[13] + s[8] * s[5] * s[14] - s[8] * s[6] * s[13] - s[12] * s[5] * s[10] + s[12] * s[6] * s[9];
    inv[1] = -s[1] * s[10] * s[15] + s[1] * s[11] * s[14] + s[9] * s[2] * s[15] - s[9] * s[3] * s[14] - s[13] * s[2] * s[11] + s[13] * s[3] * s[10];
    inv[5] = s[0] * s[10] * s[15] - s[0] * s[11] * s[14] - s[8] * s[2] * s[15] + s[8] * s[3] * s[14] + s[12] * s[2] * s[11] - s[12] * s[3] * s[10];
    inv[9] = -s[0] * s[9] * s[15] + s[0] * s[11] * s[13] + s[8] * s[1] * s[15] - s[8] * s[3] * s[13] - s[12] * s[1] * s[11] + s[12] * s[3] * s[9];
    inv[13] = s[0] * s[9] * s[14] - s[0] * s[10] * s[13] - s[8] * s[1] * s[14] + s[8] * s[2] * s[13] + s[12] * s[1] * s[10] - s[12] * s[2] * s[9];
    inv[2] = s[1] * s[6] * s[15] - s[1] * s[7] * s[14] - s[5] * s[2] * s[15] + s[5] * s[3] * s[14] + s[13] * s[2] * s[7] - s[13] * s[3] * s[6];
    inv[6] = -s[0] * s[6] * s[15] + s[0] * s[7] * s[14] + s[4] * s[2] * s[15] - s[4] * s[3] * s[14] - s[12] * s[2] * s[7] + s[12] * s[3] * s[6];
    inv[10] = s[0] * s[5] * s[15] - s[0] * s[7] * s[13] - s[4] * s[1] * s[15] + s[4] * s[3] * s[13] + s[12] * s[1] * s[7] - s[12] * s[3] * s[5];
    inv[14] = -s[0] * s[5] * s[14] + s[0] * s[6] * s[13] + s[4] * s[1] * s[14] - s[4] * s[2] * s[13] - s[12] * s[1] * s[6] + s[12] * s[2] * s[5];
    inv[3] = -s[1] * s[6] * s[11] + s[1] * s[7] * s[10] + s[5] * s[2] * s[11] - s[5] * s[3] * s[10] - s[9] * s[2] * s[7] + s[9] * s[3] * s[6];
    inv[7] = s[0] * s[6] * s[11] - s[0] * s[7] * s[10] - s[4] * s[2] * s[11] + s[4] * s[3] * s[10] + s[8] * s[2] * s[7] - s[8] * s[3] * s[6];
    inv[11] = -s[0] * s[5] * s[11] + s[0] * s[7] * s[9] + s[4] * s[1] * s[11] - s[4] * s[3] * s[9] - s[8] * s[1] * s[7] + s[8] * s[3] * s[5];
    inv[15] = s[0] * s[5] * s[10] - s[0] * s[6] * s[9] - s[4] * s[1] * s[10] + s[4] * s[2] * s[9] + s[8] * s[1] * s[6] - s[8] * s[2] * s[5];
    const float det = s[0] * inv[0] + s[1] * inv[4] + s[2] * inv[8] + s[3] * inv[12];
    const float id = 1.0f / det;
    Mat4 r;
    for (int i = 0; i < 16; i++)
        (&r.m[0][0])[i] = inv[i] * id;
    return r;
}

// glm::lookAtLH (GLM_FORCE_LEFT_HANDED, Camera.cpp:1-2,62)
inline Mat4 LookAtLH(Vec3 eye, Vec3 center, Vec3 up)
{
    const Vec3 f = Normalize(center - eye);
    const Vec3 s = Normalize(Cross(up, f));
    const Vec3 u = Cross(f, s);
    Mat4 r = Mat4::Identity();
    r.m[0][0] = s.x; r.m[0][1] = s.y; r.m[0][2] = s.z; r.m[0][3] = -Dot(s, eye);
    r.m[1][0] = u.x; r.m[1][1] = u.y; r.m[1][2] = u.z; r.m[1][3] = -Dot(u, eye);
    r.m[2][0] = f.x; r.m[2][1] = f.y; r.m[2][2] = f.z; r.m[2][3] = -Dot(f, eye);
    return r;
}

// glm::perspectiveFovLH_ZO (GLM_FORCE_DEPTH_ZERO_TO_ONE, Camera.cpp:67-70)
inline Mat4 PerspectiveFovLH_ZO(float fov, float width, float height, float zNear, float zFar)
{
    const float h = std::cos(0.5f * fov) / std::sin(0.5f * fov);
    const float w = h * height / width;
    Mat4 r;
    std::memset(r.m, 0, sizeof(r.m));
    r.m[0][0] = w;
    r.m[1][1] = h;
    r.m[2][2] = zFar / (zFar - zNear);
    r.m[3][2] = 1.0f;
    r.m[2][3] = -(zFar * zNear) / (zFar - zNear);
    return r;
}

// math matrix -> glm column-major float[16] (element [col*4+row]), the layout of
// RaygenUniformData's mat4 members
inline void ToColumnMajor(const Mat4 &a, float *out)
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            out[c * 4 + r] = a.m[r][c];
}

}
