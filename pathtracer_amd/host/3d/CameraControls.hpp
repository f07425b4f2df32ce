// Headless CameraControls: just the getters the path tracer calls (reference src/PathTraceRenderer.cpp:244-245,271,
// 286-287,291,398).  FW's GUI camera is not in the reference's tree; the matrix conventions here (camera looks down -z,
// orientation columns = right, up, -forward; GL clip space) are this build's definition.
#pragma once
#include "../base/Math.hpp"

namespace FW {

class CameraControls {
public:
    CameraControls() {}
    CameraControls(const Vec3f& position, const Vec3f& forward, const Vec3f& up, F32 fovDegrees, F32 nearDist, F32 farDist)
        : m_position(position), m_forward(forward), m_up(up), m_fov(fovDegrees), m_near(nearDist), m_far(farDist) {}

    const Vec3f& getPosition() const { return m_position; }
    Vec3f getForward() const { return m_forward.normalized(); }
    const Vec3f& getUp() const { return m_up; }
    F32 getFOV() const { return m_fov; }
    F32 getNear() const { return m_near; }
    F32 getFar() const { return m_far; }
    void setPosition(const Vec3f& v) { m_position = v; }
    void setForward(const Vec3f& v) { m_forward = v; }
    void setUp(const Vec3f& v) { m_up = v; }
    void setFOV(F32 v) { m_fov = v; }
    void setNear(F32 v) { m_near = v; }
    void setFar(F32 v) { m_far = v; }

    Mat3f getOrientation() const {
        Mat3f r;
        Vec3f z = -getForward();
        Vec3f x = m_up.cross(z).normalized();
        Vec3f y = z.cross(x).normalized();
        r.setCol(0, x); r.setCol(1, y); r.setCol(2, z);
        return r;
    }
    Mat4f getWorldToCamera() const {
        Mat3f o = getOrientation();
        Vec3f pos = o.transposed() * m_position;
        Mat4f r;
        r.setRow(0, Vec4f(o.getCol(0), -pos.x));
        r.setRow(1, Vec4f(o.getCol(1), -pos.y));
        r.setRow(2, Vec4f(o.getCol(2), -pos.z));
        return r;
    }
    Mat4f getCameraToClip() const { return Mat4f::perspective(m_fov, m_near, m_far); }

private:
    Vec3f m_position = Vec3f(0.0f, 0.0f, 0.48f);
    Vec3f m_forward = Vec3f(0.0f, -0.1f, -1.0f);
    Vec3f m_up = Vec3f(0.0f, 1.0f, 0.0f);
    F32 m_fov = 70.0f, m_near = 0.001f, m_far = 3.0f;
};

}  // namespace FW
