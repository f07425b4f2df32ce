// Rectangular area light with the reference's interface (reference src/AreaLight.hpp:14-80); the GL preview (draw) and the GUI
// state dump are out of scope.  sample()/sampleDisk() are the host-side forms; the kernels restate them on the device and
// receive the light as the POD returned by toPod().
#pragma once
#include "../../include/ptgpu.h"
#include "base/Math.hpp"
#include "base/Random.hpp"
#include "sobol.h"

namespace FW {

class AreaLight {
public:
    AreaLight() : m_intensity(1.0f), m_size(0.25f, 0.25f), m_E(100.0f, 100.0f, 100.0f) {
        setPosition(Vec3f(0.0f));
        setOrientation(Mat3f());
        updatePower();
    }
    AreaLight(Vec3f E, Vec2f size, float intensity) : m_intensity(intensity), m_size(size), m_E(E) {
        setPosition(Vec3f(0.0f));
        setOrientation(Mat3f());
        updatePower();
    }
    Mat3f getOrientation() const { return m_xform.getXYZ(); }
    Vec2f getSize() const { return m_size; }
    Vec3f getEmission() const { return m_E; }
    Vec3f getNormal() const { return -m_xform.getCol(2).getXYZ(); }
    Vec3f getPosition() const { return m_xform.getCol(3).getXYZ(); }
    float getArea() const { return 4.0f * m_size.x * m_size.y; }
    float getIntensity() const { return m_intensity; }
    float getPower() const { return m_power; }

    void sample(float& pdf, Vec3f& p, int base, int dimension, Random& rnd);
    void sampleDisk(float& pdf, Vec3f& p, int base, int dimension, Random& rnd);

    void setEmission(const Vec3f& E) { m_E = E; }
    void setIntensity(const float i) { m_intensity = i; }
    void setPosition(const Vec3f& p) { m_xform.setCol(3, Vec4f(p, 1.0f)); }
    void setSize(const Vec2f& s) { m_size = s; }
    void setOrientation(const Mat3f& R) {
        for (int c = 0; c < 3; ++c) m_xform.setCol(c, Vec4f(R.getCol(c), 0.0f));
    }
    void updatePower() { m_power = 4.0f * m_size.x * m_size.y * m_E.length() * m_intensity; }

    ptgpu_light toPod() const {
        ptgpu_light l;
        for (int i = 0; i < 16; ++i) l.xform[i] = m_xform.m[i];
        l.size[0] = m_size.x; l.size[1] = m_size.y;
        l.E[0] = m_E.x; l.E[1] = m_E.y; l.E[2] = m_E.z;
        l.intensity = m_intensity;
        l.power = m_power;
        return l;
    }

protected:
    float m_intensity;
    float m_power;
    Mat4f m_xform;
    Vec2f m_size;  // half extents
    Vec3f m_E;
};

}  // namespace FW
