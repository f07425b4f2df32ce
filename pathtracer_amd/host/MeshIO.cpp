// Headless scene ingestion for the host classes: .ptmesh, Wavefront OBJ/MTL, textures as PNG or binary PPM (ImageIO.cpp).
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

#include "3d/Mesh.hpp"

namespace FW {

namespace {
std::string dirOf(const std::string& path) {
    size_t k = path.find_last_of("/\\");
    return k == std::string::npos ? std::string() : path.substr(0, k + 1);
}
bool fail(std::string* err, const std::string& msg) { if (err) *err = msg; return false; }

struct MtlEntry { MeshBase::Material mat; std::string mapKd, mapBump; };

// A texture a material names is attached when the file exists next to the mesh -- under the name the .mtl gives (PNG or PPM), else under
// the same stem with .ppm.  A missing file leaves the slot empty, like an importer that cannot find a map (three of head.mtl's six
// maps are absent from the reference's own checkout).
void attachTextures(MeshBase::Material& m, const std::string& dir, const std::string& mapKd, const std::string& mapBump,
                    std::map<std::string, std::shared_ptr<Image>>& cache) {
    auto load = [&](const std::string& name) -> std::shared_ptr<Image> {
        if (name.empty()) return nullptr;
        std::string stem = name.substr(0, name.find_last_of('.'));
        auto it = cache.find(stem);
        if (it != cache.end()) return it->second;
        std::string e;
        std::shared_ptr<Image> img;
        if (std::ifstream(dir + name).good()) img = importImage(dir + name, &e);
        if (!img) img = importPPM(dir + stem + ".ppm", &e);
        return cache[stem] = img;
    };
    if (auto img = load(mapKd)) m.textures[MeshBase::TextureType_Diffuse] = Texture(img, mapKd);
    if (auto img = load(mapBump)) m.textures[MeshBase::TextureType_Normal] = Texture(img, mapBump);
}
}  // namespace

std::shared_ptr<Image> importPPM(const std::string& path, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fail(err, "cannot open " + path); return nullptr; }
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    f >> magic;
    auto skip = [&] { while (f >> std::ws && f.peek() == '#') { std::string l; std::getline(f, l); } };
    skip(); f >> w; skip(); f >> h; skip(); f >> maxv;
    f.get();
    if (magic != "P6" || w <= 0 || h <= 0 || maxv != 255) { fail(err, path + ": not an 8-bit binary PPM"); return nullptr; }
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    if (!f.read((char*)rgb.data(), rgb.size())) { fail(err, path + ": truncated"); return nullptr; }
    auto img = std::make_shared<Image>(Vec2i(w, h), ImageFormat::R8_G8_B8_A8);
    uint8_t* dst = (uint8_t*)img->getMutablePtr();
    for (size_t i = 0; i < (size_t)w * h; ++i) { dst[i * 4] = rgb[i * 3]; dst[i * 4 + 1] = rgb[i * 3 + 1]; dst[i * 4 + 2] = rgb[i * 3 + 2]; dst[i * 4 + 3] = 255; }
    return img;
}

bool exportPPM(const std::string& path, const Image& img, std::string* err) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return fail(err, "cannot create " + path);
    const Vec2i s = img.getSize();
    f << "P6\n" << s.x << " " << s.y << "\n255\n";
    for (int y = 0; y < s.y; ++y)
        for (int x = 0; x < s.x; ++x) {
            Vec4f c = clamp(img.getVec4f(Vec2i(x, y)), 0.0f, 1.0f);
            uint8_t px[3] = {(uint8_t)(c.x * 255.0f + 0.5f), (uint8_t)(c.y * 255.0f + 0.5f), (uint8_t)(c.z * 255.0f + 0.5f)};
            f.write((const char*)px, 3);
        }
    return (bool)f;
}

std::unique_ptr<MeshWithColors> importPtMesh(const std::string& path, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fail(err, "cannot open " + path); return nullptr; }
    char magic[8];
    int32_t hdr[4];
    f.read(magic, 8);
    f.read((char*)hdr, 16);
    if (!f || std::memcmp(magic, "PTMESH1\0", 8) != 0 || hdr[0] < 0 || hdr[1] < 0 || hdr[2] <= 0) { fail(err, path + ": bad .ptmesh header"); return nullptr; }
    const int nv = hdr[0], nt = hdr[1], nm = hdr[2];
    {   // the header's counts must fit the file before anything is sized from them
        f.seekg(0, std::ios::end);
        const long long fileBytes = (long long)f.tellg();
        f.seekg(24, std::ios::beg);
        const long long need = 24 + 32LL * nv + 16LL * nt + 192LL * nm;  // 8 floats per vertex, 3 + 1 ints per triangle, 192-byte material records
        if (need > fileBytes) { fail(err, path + ": header counts exceed the file's size"); return nullptr; }
    }
    std::vector<float> pos((size_t)nv * 3), nrm((size_t)nv * 3), uv((size_t)nv * 2);
    std::vector<int32_t> idx((size_t)nt * 3), tmat(nt);
    f.read((char*)pos.data(), pos.size() * 4); f.read((char*)nrm.data(), nrm.size() * 4); f.read((char*)uv.data(), uv.size() * 4);
    f.read((char*)idx.data(), idx.size() * 4); f.read((char*)tmat.data(), tmat.size() * 4);
    auto mesh = std::make_unique<MeshWithColors>();
    std::map<std::string, std::shared_ptr<Image>> texCache;
    for (int i = 0; i < nm; ++i) {
        struct { char name[32]; float diffuse[4]; float specular[3]; float gloss; char kd[64]; char bump[64]; } rec;
        f.read((char*)&rec, sizeof rec);
        int s = mesh->addSubmesh();
        MeshBase::Material& m = mesh->material(s);
        m.name = std::string(rec.name, strnlen(rec.name, 32));
        m.diffuse = Vec4f(rec.diffuse[0], rec.diffuse[1], rec.diffuse[2], rec.diffuse[3]);
        m.specular = Vec3f(rec.specular[0], rec.specular[1], rec.specular[2]);
        m.glossiness = rec.gloss;
        attachTextures(m, dirOf(path), std::string(rec.kd, strnlen(rec.kd, 64)), std::string(rec.bump, strnlen(rec.bump, 64)), texCache);
    }
    if (!f) { fail(err, path + ": truncated"); return nullptr; }
    for (int i = 0; i < nv; ++i) {
        VertexPNTC v;
        v.p = Vec3f(pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]);
        v.n = Vec3f(nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2]);
        v.t = Vec2f(uv[i * 2], uv[i * 2 + 1]);
        v.c = Vec3f(1.0f);
        mesh->addVertex(v);
    }
    for (int t = 0; t < nt; ++t) {
        if (tmat[t] < 0 || tmat[t] >= nm) { fail(err, path + ": material index out of range"); return nullptr; }
        for (int k = 0; k < 3; ++k)
            if (idx[t * 3 + k] < 0 || idx[t * 3 + k] >= nv) { fail(err, path + ": vertex index out of range"); return nullptr; }
        mesh->mutableIndices(tmat[t]).push_back(Vec3i(idx[t * 3], idx[t * 3 + 1], idx[t * 3 + 2]));
    }
    return mesh;
}

std::unique_ptr<MeshWithColors> importWavefrontMesh(const std::string& path, std::string* err) {
    std::ifstream f(path);
    if (!f) { fail(err, "cannot open " + path); return nullptr; }
    std::vector<Vec3f> P, N;
    std::vector<Vec2f> T;
    std::map<std::string, MtlEntry> mtl;
    std::map<std::string, int> submeshOf;
    std::map<std::tuple<int, int, int>, int> vertexOf;
    std::map<std::string, std::shared_ptr<Image>> texCache;
    auto mesh = std::make_unique<MeshWithColors>();
    int cur = -1;
    auto useMaterial = [&](const std::string& name) {
        auto it = submeshOf.find(name);
        if (it != submeshOf.end()) { cur = it->second; return; }
        cur = mesh->addSubmesh();
        submeshOf[name] = cur;
        auto m = mtl.find(name);
        if (m != mtl.end()) {
            mesh->material(cur) = m->second.mat;
            attachTextures(mesh->material(cur), dirOf(path), m->second.mapKd, m->second.mapBump, texCache);
        }
        mesh->material(cur).name = name;
    };
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag) || tag[0] == '#') continue;
        if (tag == "v") { Vec3f v; ss >> v.x >> v.y >> v.z; P.push_back(v); }
        else if (tag == "vn") { Vec3f v; ss >> v.x >> v.y >> v.z; N.push_back(v); }
        else if (tag == "vt") { Vec2f v; ss >> v.x >> v.y; T.push_back(v); }
        else if (tag == "mtllib") {
            std::string name; ss >> name;
            std::ifstream mf(dirOf(path) + name);
            std::string ml, curName;
            while (std::getline(mf, ml)) {
                std::istringstream ms(ml);
                std::string k;
                if (!(ms >> k)) continue;
                if (k == "newmtl") { ms >> curName; mtl[curName] = MtlEntry(); }
                else if (curName.empty()) continue;
                else if (k == "Kd") { Vec4f& d = mtl[curName].mat.diffuse; ms >> d.x >> d.y >> d.z; }
                else if (k == "Ks") { Vec3f& s = mtl[curName].mat.specular; ms >> s.x >> s.y >> s.z; }
                else if (k == "Ns") ms >> mtl[curName].mat.glossiness;
                else if (k == "map_Kd") ms >> mtl[curName].mapKd;
                else if (k == "map_bump" || k == "bump") ms >> mtl[curName].mapBump;
            }
        } else if (tag == "usemtl") { std::string name; ss >> name; useMaterial(name); }
        else if (tag == "f") {
            if (cur < 0) useMaterial("default");
            std::vector<int> corner;
            std::string c;
            while (ss >> c) {
                int vi = 0, ti = 0, ni = 0;
                if (std::sscanf(c.c_str(), "%d/%d/%d", &vi, &ti, &ni) == 3) {}
                else if (std::sscanf(c.c_str(), "%d//%d", &vi, &ni) == 2) { ti = 0; }
                else if (std::sscanf(c.c_str(), "%d/%d", &vi, &ti) == 2) { ni = 0; }
                else if (std::sscanf(c.c_str(), "%d", &vi) == 1) { ti = ni = 0; }
                else { fail(err, path + ": bad face corner '" + c + "'"); return nullptr; }
                if (vi < 0) vi += (int)P.size() + 1;
                if (ti < 0) ti += (int)T.size() + 1;
                if (ni < 0) ni += (int)N.size() + 1;
                if (vi < 1 || vi > (int)P.size() || ti > (int)T.size() || ni > (int)N.size()) { fail(err, path + ": face index out of range"); return nullptr; }
                auto key = std::make_tuple(vi, ti, ni);
                auto it = vertexOf.find(key);
                if (it == vertexOf.end()) {
                    VertexPNTC v;
                    v.p = P[vi - 1];
                    if (ti) v.t = T[ti - 1];
                    if (ni) v.n = N[ni - 1];
                    v.c = Vec3f(1.0f);
                    it = vertexOf.emplace(key, mesh->addVertex(v)).first;
                }
                corner.push_back(it->second);
            }
            for (size_t k = 2; k < corner.size(); ++k)  // fan triangulation
                mesh->mutableIndices(cur).push_back(Vec3i(corner[0], corner[k - 1], corner[k]));
        }
    }
    // faces without normals get the geometric normal
    for (int s = 0; s < mesh->numSubmeshes(); ++s)
        for (const Vec3i& t : mesh->indices(s)) {
            Vec3f gn = cross(mesh->vertex(t.y).p - mesh->vertex(t.x).p, mesh->vertex(t.z).p - mesh->vertex(t.x).p).normalized();
            for (int k = 0; k < 3; ++k) {
                VertexPNTC& v = mesh->mutableVertex(t[k]);
                if (v.n.x == 0.0f && v.n.y == 0.0f && v.n.z == 0.0f) v.n = gn;
            }
        }
    return mesh;
}

std::unique_ptr<MeshWithColors> importMesh(const std::string& path, std::string* err) {
    size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext == ".ptmesh") return importPtMesh(path, err);
    if (ext == ".obj") return importWavefrontMesh(path, err);
    fail(err, "unsupported mesh format '" + ext + "'");
    return nullptr;
}

}  // namespace FW
