// Minimal mesh / material / image containers with the member names the reference's hot-path code reads from FW's
// 3d/Mesh.hpp and gui/Image.hpp, plus headless loaders (.ptmesh written by tools/convert_head_obj.py,
// and Wavefront OBJ/MTL for already-triangulated or polygonal faces).  FW's importer is not in the reference's tree, so the
// import conventions are this build's: one vertex per distinct v/vt/vn triple in order of first use, one submesh per
// material in order of first use, vt.y not flipped, defaults Kd=(.75,.75,.75,1) Ks=0 Ns=0.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../base/Math.hpp"

namespace FW {

struct VertexPNTC {
    Vec3f p, n;
    Vec2f t;
    Vec3f c;
};

namespace ImageFormat {
enum ID { RGBA_Vec4f, R8_G8_B8_A8 };
}

// Image: either float RGBA (accumulation / display) or 8-bit RGBA (textures).
class Image {
public:
    Image(const Vec2i& size, ImageFormat::ID fmt) : m_size(size), m_format(fmt) {
        if (fmt == ImageFormat::RGBA_Vec4f) m_f.assign((size_t)size.x * size.y, Vec4f(0.0f));
        else m_b.assign((size_t)size.x * size.y * 4, 0);
    }
    const Vec2i& getSize() const { return m_size; }
    ImageFormat::ID getFormat() const { return m_format; }
    Vec4f getVec4f(const Vec2i& pos) const {
        size_t i = (size_t)pos.y * m_size.x + pos.x;
        if (m_format == ImageFormat::RGBA_Vec4f) return m_f[i];
        const uint8_t* p = &m_b[i * 4];
        return Vec4f(p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f, p[3] / 255.0f);
    }
    void setVec4f(const Vec2i& pos, const Vec4f& v) {
        size_t i = (size_t)pos.y * m_size.x + pos.x;
        if (m_format == ImageFormat::RGBA_Vec4f) { m_f[i] = v; return; }
        auto q = [](F32 a) { a = a < 0 ? 0 : (a > 1 ? 1 : a); return (uint8_t)(a * 255.0f + 0.5f); };
        uint8_t* p = &m_b[i * 4];
        p[0] = q(v.x); p[1] = q(v.y); p[2] = q(v.z); p[3] = q(v.w);
    }
    void clear() { for (auto& v : m_f) v = Vec4f(0.0f); for (auto& b : m_b) b = 0; }
    void* getMutablePtr() { return m_format == ImageFormat::RGBA_Vec4f ? (void*)m_f.data() : (void*)m_b.data(); }
    const void* getPtr() const { return m_format == ImageFormat::RGBA_Vec4f ? (const void*)m_f.data() : (const void*)m_b.data(); }

private:
    Vec2i m_size;
    ImageFormat::ID m_format;
    std::vector<Vec4f> m_f;
    std::vector<uint8_t> m_b;
};

class Texture {
public:
    Texture() {}
    explicit Texture(std::shared_ptr<Image> img, const std::string& id = "") : m_image(std::move(img)), m_id(id) {}
    bool exists() const { return (bool)m_image; }
    const Image* getImage() const { return m_image.get(); }
    const std::string& getID() const { return m_id; }

private:
    std::shared_ptr<Image> m_image;
    std::string m_id;
};

class MeshBase {
public:
    enum TextureType { TextureType_Diffuse = 0, TextureType_Alpha, TextureType_Displacement, TextureType_Normal, TextureType_Environment,
                       TextureType_Specular, TextureType_Max };
    struct Material {
        Vec4f diffuse = Vec4f(0.75f, 0.75f, 0.75f, 1.0f);
        Vec3f specular = Vec3f(0.0f);
        F32 glossiness = 0.0f;
        Texture textures[TextureType_Max];
        std::string name;
    };
    virtual ~MeshBase() {}
};

template <class V> class Mesh : public MeshBase {
public:
    int numSubmeshes() const { return (int)m_submeshes.size(); }
    int numVertices() const { return (int)m_vertices.size(); }
    int numTriangles() const { size_t n = 0; for (auto& s : m_submeshes) n += s.indices.size(); return (int)n; }
    const std::vector<Vec3i>& indices(int submesh) const { return m_submeshes[submesh].indices; }
    std::vector<Vec3i>& mutableIndices(int submesh) { return m_submeshes[submesh].indices; }
    const V& vertex(int i) const { return m_vertices[i]; }
    V& mutableVertex(int i) { return m_vertices[i]; }
    const Material& material(int submesh) const { return m_submeshes[submesh].material; }
    Material& material(int submesh) { return m_submeshes[submesh].material; }
    int addSubmesh() { m_submeshes.emplace_back(); return (int)m_submeshes.size() - 1; }
    int addVertex(const V& v) { m_vertices.push_back(v); return (int)m_vertices.size() - 1; }

private:
    struct Submesh { std::vector<Vec3i> indices; Material material; };
    std::vector<V> m_vertices;
    std::vector<Submesh> m_submeshes;
};

typedef Mesh<VertexPNTC> MeshWithColors;

// Loaders (MeshIO.cpp). Return nullptr and fill err on failure.
std::unique_ptr<MeshWithColors> importPtMesh(const std::string& path, std::string* err);
std::unique_ptr<MeshWithColors> importWavefrontMesh(const std::string& path, std::string* err);
std::unique_ptr<MeshWithColors> importMesh(const std::string& path, std::string* err);  // by extension
// Image files (ImageIO.cpp, MeshIO.cpp) -> 8-bit RGBA texture image: PNG (the format of the reference's head/*.png maps; own decoder,
// 8-bit non-interlaced) and binary PPM (P6).  importImage picks by extension.  exportPNG writes the 8-bit RGB picture the reference
// writes per pass with exportLodePngImage (src/PathTraceRenderer.cpp:900-902).
std::shared_ptr<Image> importPNG(const std::string& path, std::string* err);
std::shared_ptr<Image> importPPM(const std::string& path, std::string* err);
std::shared_ptr<Image> importImage(const std::string& path, std::string* err);
bool exportPNG(const std::string& path, const Image& img, std::string* err);
bool exportPPM(const std::string& path, const Image& img, std::string* err);

}  // namespace FW
