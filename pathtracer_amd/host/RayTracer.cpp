#include "RayTracer.hpp"

#include <cstring>
#include <fstream>
#include <map>

#include "Md5.hpp"

namespace FW {

Vec2f getTexelCoords(Vec2f uv, const Vec2i size) {
    return Vec2f((uv.x - floor(uv.x)) * size.x, (uv.y - floor(uv.y)) * size.y);
}

Mat3f formBasis(const Vec3f& n) {
    // swap the smallest-magnitude component for 1 to get a vector that is certainly not parallel to n
    Vec3f q = n;
    int k = 0;
    if (abs(q.y) < abs(q[k])) k = 1;
    if (abs(q.z) < abs(q[k])) k = 2;
    q[k] = 1.0f;
    Vec3f t = normalize(q.cross(n));
    Vec3f b = normalize(n.cross(t));
    Mat3f R;
    R.setCol(0, t); R.setCol(1, b); R.setCol(2, n);
    return R;
}

std::string RayTracer::computeMD5(const std::vector<Vec3f>& vertices) {
    return md5Hex(vertices.empty() ? nullptr : &vertices[0], sizeof(Vec3f) * vertices.size());
}

RayTracer::RayTracer() {}
RayTracer::~RayTracer() { invalidateDeviceScene(); }

void RayTracer::invalidateDeviceScene() {
    if (m_scene) { ptgpu_scene_destroy(m_scene); m_scene = nullptr; }
}

void RayTracer::constructHierarchy(std::vector<RTTriangle>& triangles, bool sahEnabled) {
    m_triangles = &triangles;
    m_buildWithScene = 0;
    bvh = BVH(m_triangles, sahEnabled);
    invalidateDeviceScene();
}

void RayTracer::constructHierarchyOnDevice(std::vector<RTTriangle>& triangles, int maxTrisInLeaf, bool keepTree) {
    m_triangles = &triangles;
    m_buildWithScene = 0;
    if (!keepTree) {  // the tree is built when the device scene is (ptgpu_scene_create_lbvh) and never copied to the host
        if (maxTrisInLeaf < 1) throw GpuError("constructHierarchyOnDevice: maxTrisInLeaf < 1");
        m_buildWithScene = maxTrisInLeaf;
        bvh = BVH();
        invalidateDeviceScene();
        return;
    }
    const size_t n = triangles.size();
    std::vector<float> pos(n * 9);
    for (size_t i = 0; i < n; ++i)
        for (int v = 0; v < 3; ++v) {
            const Vec3f& p = triangles[i].m_vertices[v].p;
            pos[i * 9 + v * 3 + 0] = p.x; pos[i * 9 + v * 3 + 1] = p.y; pos[i * 9 + v * 3 + 2] = p.z;
        }
    static_assert(sizeof(FlatNode) == sizeof(ptgpu_flatnode), "FlatNode is uploaded and downloaded as ptgpu_flatnode");
    std::vector<FlatNode> nodes(std::max<size_t>(2 * n, 1));
    std::vector<int32_t> list(std::max<size_t>(n, 1));
    int32_t numNodes = 0, height = 0;
    PT_GPU(ptgpu_bvh_build_lbvh(GpuDevice::instance().ctx(), pos.data(), (int32_t)n, maxTrisInLeaf,
                                reinterpret_cast<ptgpu_flatnode*>(nodes.data()), &numNodes, list.data(), &height));
    nodes.resize((size_t)numNodes);
    list.resize(n);
    int32_t splits = 0;  // BVH::depth counts splits, like the reference's field
    for (const FlatNode& f : nodes) splits += f.leaf ? 0 : 1;
    bvh = BVH(m_triangles, std::move(list), std::move(nodes), splits);
    invalidateDeviceScene();
}

// Cache file layout (reference src/RayTracer.cpp:139-165): int32 listSize, int32 list[], int32 nNodes, then per node
// min(3f) max(3f) startIdx endIdx rightChild, leaf as ONE byte, axis; finally int32 depth.
namespace {
template <class T> void put(std::ostream& s, const T& v) { s.write(reinterpret_cast<const char*>(&v), sizeof v); }
template <class T> bool get(std::istream& s, T& v) { return (bool)s.read(reinterpret_cast<char*>(&v), sizeof v); }
}  // namespace

void RayTracer::saveHierarchy(const char* filename, const std::vector<RTTriangle>&) {
    if (m_buildWithScene) throw GpuError("saveHierarchy: the hierarchy was built on the device without a host copy (constructHierarchyOnDevice keepTree = false)");
    std::ofstream f(filename, std::ios::binary);
    put(f, (int32_t)bvh.list.size());
    for (int32_t v : bvh.list) put(f, v);
    put(f, (int32_t)bvh.nodes.size());
    for (const FlatNode& n : bvh.nodes) {
        put(f, n.box.min); put(f, n.box.max);
        put(f, n.startIdx); put(f, n.endIdx); put(f, n.rightChild);
        put(f, (uint8_t)(n.leaf ? 1 : 0));
        put(f, n.axis);
    }
    put(f, bvh.depth);
}

void RayTracer::loadHierarchy(const char* filename, std::vector<RTTriangle>& triangles) {
    m_triangles = &triangles;
    m_buildWithScene = 0;
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw GpuError(std::string("loadHierarchy: cannot open ") + filename);
    f.seekg(0, std::ios::end);
    const long long fileBytes = (long long)f.tellg();
    f.seekg(0, std::ios::beg);
    auto bad = [&](const char* what) { return GpuError(std::string("loadHierarchy: ") + filename + ": " + what); };
    int32_t listSize = 0, nNodes = 0, depth = 0;
    // counts are checked against the file's length before anything is sized from them
    if (!get(f, listSize) || listSize < 0 || 4LL * listSize > fileBytes) throw bad("bad list size");
    std::vector<int32_t> list((size_t)listSize);
    if (listSize && !f.read(reinterpret_cast<char*>(list.data()), (std::streamsize)listSize * 4)) throw bad("truncated index list");
    if (!get(f, nNodes) || nNodes < 0 || 41LL * nNodes > fileBytes) throw bad("bad node count");
    std::vector<FlatNode> nodes;
    nodes.reserve((size_t)nNodes);
    for (int32_t i = 0; i < nNodes; ++i) {
        Vec3f mn, mx;
        int32_t s = 0, e = 0, r = 0, ax = 0;
        uint8_t leaf = 0;  // the reference writes its bool as one byte (:156); any non-zero value reads as a leaf
        if (!(get(f, mn) && get(f, mx) && get(f, s) && get(f, e) && get(f, r) && get(f, leaf) && get(f, ax))) throw bad("truncated node array");
        nodes.emplace_back(AABB(mn, mx), s, e, r, leaf != 0, ax);
        if (!leaf) nodes.back().leftChild = i + 1;
    }
    if (!get(f, depth)) throw bad("missing depth field");
    bvh = BVH(m_triangles, list, nodes, depth);
    invalidateDeviceScene();
}

ptgpu_scene* RayTracer::deviceScene() const {
    if (m_scene) return m_scene;
    if (!m_triangles) throw GpuError("RayTracer: no hierarchy (call constructHierarchy or loadHierarchy first)");
    const std::vector<RTTriangle>& T = *m_triangles;
    const size_t n = T.size();
    std::vector<float> pos(n * 9), nrm(n * 9), uv(n * 6), woop(n * 12);
    std::vector<int32_t> matId(n);
    std::map<const MeshBase::Material*, int32_t> matIndex;
    std::vector<const MeshBase::Material*> mats;
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            const VertexPNTC& v = T[i].m_vertices[k];
            std::memcpy(&pos[i * 9 + k * 3], &v.p, 12);
            std::memcpy(&nrm[i * 9 + k * 3], &v.n, 12);
            std::memcpy(&uv[i * 6 + k * 2], &v.t, 8);
        }
        T[i].m_data.packRows(&woop[i * 12]);
        auto it = matIndex.find(T[i].m_material);
        if (it == matIndex.end()) { it = matIndex.emplace(T[i].m_material, (int32_t)mats.size()).first; mats.push_back(T[i].m_material); }
        matId[i] = it->second;
    }
    static const MeshBase::Material defaultMaterial;
    std::map<const Image*, int32_t> texIndex;
    std::vector<ptgpu_texture> texs;
    auto texOf = [&](const Texture& t) -> int32_t {
        if (!t.exists() || t.getImage()->getFormat() != ImageFormat::R8_G8_B8_A8) return -1;
        auto it = texIndex.find(t.getImage());
        if (it != texIndex.end()) return it->second;
        ptgpu_texture d;
        d.width = t.getImage()->getSize().x; d.height = t.getImage()->getSize().y;
        d.rgba = (const uint8_t*)t.getImage()->getPtr();
        texs.push_back(d);
        return texIndex[t.getImage()] = (int32_t)texs.size() - 1;
    };
    std::vector<ptgpu_material> pm(mats.size());
    for (size_t i = 0; i < mats.size(); ++i) {
        const MeshBase::Material& m = mats[i] ? *mats[i] : defaultMaterial;
        std::memcpy(pm[i].diffuse, &m.diffuse, 16);
        std::memcpy(pm[i].specular, &m.specular, 12);
        pm[i].glossiness = m.glossiness;
        pm[i].tex_diffuse = texOf(m.textures[MeshBase::TextureType_Diffuse]);
        pm[i].tex_alpha = texOf(m.textures[MeshBase::TextureType_Alpha]);
        pm[i].tex_normal = texOf(m.textures[MeshBase::TextureType_Normal]);
        pm[i].tex_specular = texOf(m.textures[MeshBase::TextureType_Specular]);
    }
    ptgpu_scene_desc d{};
    d.num_tris = (int32_t)n;
    d.pos = pos.data(); d.nrm = nrm.data(); d.uv = uv.data(); d.mat_id = matId.data(); d.woop = woop.data();
    d.num_materials = (int32_t)pm.size(); d.materials = pm.data();
    d.num_textures = (int32_t)texs.size(); d.textures = texs.data();
    if (m_buildWithScene) {
        PT_GPU(ptgpu_scene_create_lbvh(GpuDevice::instance().ctx(), &d, m_buildWithScene, &m_scene, nullptr));
        return m_scene;
    }
    d.num_nodes = (int32_t)bvh.nodes.size();
    d.nodes = reinterpret_cast<const ptgpu_flatnode*>(bvh.nodes.data());  // same 64-byte layout, uploaded as-is
    d.list_size = (int32_t)bvh.list.size(); d.list = bvh.list.data();
    PT_GPU(ptgpu_scene_create(GpuDevice::instance().ctx(), &d, &m_scene));
    return m_scene;
}

void RayTracer::raycastBatch(const Vec3f* orig, const Vec3f* dir, size_t n, bool occlusion, RaycastResult* out) const {
    if (n == 0) return;
    std::vector<int32_t> id(n);
    std::vector<float> t(n), u(n), v(n);
    PT_GPU(ptgpu_raycast_batch(deviceScene(), useAlpha ? 1 : 0, &orig[0].x, &dir[0].x, (int64_t)n, occlusion ? 1 : 0, id.data(),
                               t.data(), u.data(), v.data()));
    m_rayCount += (long long)n;
    for (size_t i = 0; i < n; ++i) {
        if (id[i] >= 0) out[i] = RaycastResult(&(*m_triangles)[id[i]], t[i], u[i], v[i], orig[i] + t[i] * dir[i], orig[i], dir[i]);
        else out[i] = RaycastResult();
    }
}

RaycastResult RayTracer::raycast(const Vec3f& orig, const Vec3f& dir, bool occlusion) const {
    RaycastResult r;
    raycastBatch(&orig, &dir, 1, occlusion, &r);
    return r;
}

}  // namespace FW
