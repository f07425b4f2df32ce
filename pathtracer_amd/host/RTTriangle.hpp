// RTTriangle / tri_data / RaycastResult with the reference's member names (reference src/RTTriangle.hpp:8-61,
// src/RaycastResult.hpp:11-30).  The Woop transform is built once per triangle on the host and uploaded as-is;
// the device never recomputes it, so GPU and CPU share bit-identical M and N.
#pragma once
#include <limits>

#include "3d/Mesh.hpp"
#include "base/Math.hpp"

namespace FW {

// Woop et al. unit-triangle transform: M maps (v1-v0, v2-v0, n) to the unit axes, N = -M v0.
struct tri_data {
    Vec3i vertex_indices;
    Mat3f M;
    Vec3f N;
    tri_data() {}
    tri_data(const Vec3f& v0, const Vec3f& v1, const Vec3f& v2, const Vec3f& unitNormal) {
        M.setCol(0, v1 - v0);
        M.setCol(1, v2 - v0);
        M.setCol(2, unitNormal);
        M.invert();
        N = -M * v0;
    }
    // rows (M(r,0), M(r,1), M(r,2), N[r]) r = 0..2: the 48-byte record ptgpu_scene_desc.woop expects
    void packRows(float out12[12]) const {
        for (int r = 0; r < 3; ++r) {
            out12[r * 4 + 0] = M.get(r, 0); out12[r * 4 + 1] = M.get(r, 1); out12[r * 4 + 2] = M.get(r, 2);
            out12[r * 4 + 3] = N[r];
        }
    }
};

struct RTTriangle {
    VertexPNTC m_vertices[3];
    MeshBase::Material* m_material = nullptr;  // non-owning, points into the mesh
    tri_data m_data;

    RTTriangle(const VertexPNTC& v0, const VertexPNTC& v1, const VertexPNTC& v2) {
        m_vertices[0] = v0; m_vertices[1] = v1; m_vertices[2] = v2;
        m_data = tri_data(v0.p, v1.p, v2.p, normal());
    }
    Vec3f min() const { return FW::min(m_vertices[0].p, m_vertices[1].p, m_vertices[2].p); }
    Vec3f max() const { return FW::max(m_vertices[0].p, m_vertices[1].p, m_vertices[2].p); }
    Vec3f centroid() const { return (m_vertices[0].p + m_vertices[1].p + m_vertices[2].p) * (1.0f / 3.0f); }
    float area() const { return cross(m_vertices[1].p - m_vertices[0].p, m_vertices[2].p - m_vertices[0].p).length() * .5f; }
    Vec3f normal() const { return cross(m_vertices[1].p - m_vertices[0].p, m_vertices[2].p - m_vertices[0].p).normalized(); }
};

// Result of one raycast; tri == nullptr means miss (t = FLT_MAX then).
struct RaycastResult {
    const RTTriangle* tri = nullptr;
    float t = std::numeric_limits<float>::max();
    float u = 0.0f, v = 0.0f;
    Vec3f point, orig, dir;
    RaycastResult() {}
    RaycastResult(const RTTriangle* tri_, float t_, float u_, float v_, const Vec3f& point_, const Vec3f& orig_, const Vec3f& dir_)
        : tri(tri_), t(t_), u(u_), v(v_), point(point_), orig(orig_), dir(dir_) {}
    operator bool() const { return tri != nullptr; }
};

}  // namespace FW
