#include "Camera.h"

namespace PathTracing
{

Camera::Camera(float verticalFOV, float nearClip, float farClip, Vec3 position, Vec3 direction, Vec3 up)
    : m_UpDirection(up), m_Position(position), m_Direction(direction), m_VerticalFOV(verticalFOV), m_NearClip(nearClip),
      m_FarClip(farClip), m_InvView(Mat4::Identity()), m_InvProjection(Mat4::Identity())
{
    UpdateInvView();
}

// Camera.cpp:25-34
void Camera::OnResize(uint32_t width, uint32_t height)
{
    if (m_Width == width && m_Height == height)
        return;
    m_Width = width;
    m_Height = height;
    UpdateInvProjection();
}

void Camera::SetPose(Vec3 position, Vec3 direction)
{
    m_Position = position;
    m_Direction = direction;
    UpdateInvView();
}

// Camera.cpp:52-63
void Camera::UpdateInvView()
{
    m_InvView = Inverse(LookAtLH(m_Position, m_Position + m_Direction, m_UpDirection));
}

// Camera.cpp:65-71
void Camera::UpdateInvProjection()
{
    m_InvProjection = Inverse(PerspectiveFovLH_ZO(Radians(m_VerticalFOV), static_cast<float>(m_Width),
                                                  static_cast<float>(m_Height), m_NearClip, m_FarClip));
}

InputCamera::InputCamera(float verticalFOV, float nearClip, float farClip, Vec3 position, Vec3 direction)
    : Camera(verticalFOV, nearClip, farClip, position, direction, Vec3(0.0f, -1.0f, 0.0f))
{
}

AnimatedCamera::AnimatedCamera(float verticalFOV, float nearClip, float farClip, Vec3 position, Vec3 direction, Vec3 up,
                               const Mat4 &transform)
    : Camera(verticalFOV, nearClip, farClip, position, direction, up), m_RelativePosition(position),
      m_RelativeDirection(direction), m_RelativeUpDirection(up), m_Transform(transform)
{
}

// Camera.cpp:165-180: position/direction/up follow the scene node
bool AnimatedCamera::OnUpdate(float)
{
    const Mat4 &t = m_Transform;
    auto point = [&](Vec3 p) {
        return Vec3(t.m[0][0] * p.x + t.m[0][1] * p.y + t.m[0][2] * p.z + t.m[0][3],
                    t.m[1][0] * p.x + t.m[1][1] * p.y + t.m[1][2] * p.z + t.m[1][3],
                    t.m[2][0] * p.x + t.m[2][1] * p.y + t.m[2][2] * p.z + t.m[2][3]);
    };
    auto vector = [&](Vec3 p) {
        return Vec3(t.m[0][0] * p.x + t.m[0][1] * p.y + t.m[0][2] * p.z, t.m[1][0] * p.x + t.m[1][1] * p.y + t.m[1][2] * p.z,
                    t.m[2][0] * p.x + t.m[2][1] * p.y + t.m[2][2] * p.z);
    };
    m_Position = point(m_RelativePosition);
    m_Direction = vector(m_RelativeDirection);
    m_UpDirection = vector(m_RelativeUpDirection);
    UpdateInvView();
    return true;
}

}
