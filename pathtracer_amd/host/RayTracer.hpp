// RayTracer with the reference's interface (reference src/RayTracer.hpp:20-44): constructHierarchy / loadHierarchy /
// saveHierarchy / raycast / ray counter, plus the additive raycastBatch.  raycast() runs on the MI355X through the C ABI
// (ptgpu_raycast_batch); nothing here traverses the BVH on the CPU.
#pragma once
#include <atomic>
#include <string>
#include <vector>

#include "Bvh.hpp"
#include "RTTriangle.hpp"
#include "gpu/Device.hpp"

namespace FW {

Mat3f formBasis(const Vec3f& n);                  // orthonormal basis with n as the last column
Vec2f getTexelCoords(Vec2f uv, const Vec2i size);  // frac(uv) * size

class RayTracer {
public:
    RayTracer();
    ~RayTracer();
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    // MD5 of the vertex positions, used to name the hierarchy cache file
    static std::string computeMD5(const std::vector<Vec3f>& vertices);

    RaycastResult raycast(const Vec3f& orig, const Vec3f& dir, bool occlusion) const;
    // n rays in one device launch; out must hold n results
    void raycastBatch(const Vec3f* orig, const Vec3f* dir, size_t n, bool occlusion, RaycastResult* out) const;

    void constructHierarchy(std::vector<RTTriangle>& triangles, bool sahEnabled);
    // Not in the reference: the linear BVH built on the device (ptgpu_bvh_build_lbvh), installed as bvh.nodes / bvh.list like a tree
    // from the CPU builders.  Hit records can differ from the SAH tree's at exact ties (see include/ptgpu.h).
    // keepTree = false: the tree is built together with the device scene (ptgpu_scene_create_lbvh) and never copied to the host -- bvh stays
    // empty and saveHierarchy throws; for scenes rebuilt every frame.
    void constructHierarchyOnDevice(std::vector<RTTriangle>& triangles, int maxTrisInLeaf = 3, bool keepTree = true);
    void loadHierarchy(const char* filename, std::vector<RTTriangle>& triangles);
    void saveHierarchy(const char* filename, const std::vector<RTTriangle>& triangles);

    int getRayCount() { return (int)m_rayCount.load(); }
    long long getRayCountLong() const { return m_rayCount.load(); }
    void resetRayCounter() { m_rayCount = 0; }
    void addRays(long long n) const { m_rayCount += n; }

    // device scene handle (built lazily from m_triangles + bvh; rebuilt after construct/loadHierarchy)
    ptgpu_scene* deviceScene() const;
    void invalidateDeviceScene();

    std::vector<RTTriangle>* m_triangles = nullptr;
    BVH bvh;
    bool useAlpha = true;

private:
    mutable std::atomic<long long> m_rayCount{0};
    mutable ptgpu_scene* m_scene = nullptr;
    int m_buildWithScene = 0;  // > 0: maxTrisInLeaf of a hierarchy that deviceScene() builds on the device
};

}  // namespace FW
