// Camera.h -- mirror of Path-Tracing/Core/Camera.{h,cpp}: produces the two inverse
// matrices RaygenUniformData carries (Renderer.cpp:1686-1694).  Interactive input
// (InputCamera::OnUpdate's key/mouse handling, Camera.cpp:82-144) is out of scope.
#pragma once

#include <cstdint>
#include <utility>

#include "Math.h"

namespace PathTracing
{

class Camera
{
public:
    Camera(float verticalFOV, float nearClip, float farClip, Vec3 position, Vec3 direction, Vec3 up);
    virtual ~Camera() = default;

    virtual bool OnUpdate(float timeStep) = 0;
    void OnResize(uint32_t width, uint32_t height);

    [[nodiscard]] std::pair<uint32_t, uint32_t> GetExtent() const { return { m_Width, m_Height }; }
    [[nodiscard]] Mat4 GetInvViewMatrix() const { return m_InvView; }
    [[nodiscard]] Mat4 GetInvProjectionMatrix() const { return m_InvProjection; }

    void SetPose(Vec3 position, Vec3 direction);

protected:
    void UpdateInvView();
    void UpdateInvProjection();

    Vec3 m_UpDirection;
    Vec3 m_Position;
    Vec3 m_Direction;

private:
    float m_VerticalFOV;
    float m_NearClip;
    float m_FarClip;
    uint32_t m_Width = 0, m_Height = 0;
    Mat4 m_InvView;
    Mat4 m_InvProjection;
};

class InputCamera : public Camera
{
public:
    // default up = (0, -1, 0) (Camera.cpp:76)
    InputCamera(float verticalFOV, float nearClip, float farClip, Vec3 position, Vec3 direction);
    bool OnUpdate(float) override { return false; }
};

class AnimatedCamera : public Camera
{
public:
    AnimatedCamera(float verticalFOV, float nearClip, float farClip, Vec3 position, Vec3 direction, Vec3 up, const Mat4 &transform);
    bool OnUpdate(float timeStep) override;

private:
    Vec3 m_RelativePosition, m_RelativeDirection, m_RelativeUpDirection;
    Mat4 m_Transform; // scene node's CurrentTransform (math matrix)
};

}
