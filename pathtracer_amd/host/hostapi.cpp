// C shim over the C++ host classes so that pytest (ctypes) and bench.py can drive them.  Not part of the drop-in
// boundary (that is include/ptgpu.h); this only exposes FW::BVH / RayTracer / AreaLight / CameraControls to Python.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "3d/CameraControls.hpp"
#include "AreaLight.hpp"
#include "Bvh.hpp"
#include "Md5.hpp"
#include "PathTraceRenderer.hpp"
#include "RayTracer.hpp"
#include "util.hpp"

using namespace FW;

namespace {
thread_local std::string g_err;
template <class F> int guarded(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
std::vector<RTTriangle> trianglesFromPositions(int n, const float* pos) {
    std::vector<RTTriangle> tris;
    tris.reserve(n);
    for (int i = 0; i < n; ++i) {
        VertexPNTC v[3];
        for (int k = 0; k < 3; ++k) v[k].p = Vec3f(pos[i * 9 + k * 3], pos[i * 9 + k * 3 + 1], pos[i * 9 + k * 3 + 2]);
        tris.emplace_back(v[0], v[1], v[2]);
    }
    return tris;
}
// App::constructTracer's triangle array (reference src/App.cpp:826-838)
void trianglesFromMesh(MeshWithColors& mesh, std::vector<RTTriangle>& out) {
    out.clear();
    out.reserve(mesh.numTriangles());
    for (int i = 0; i < mesh.numSubmeshes(); ++i)
        for (const Vec3i& idx : mesh.indices(i)) {
            RTTriangle t(mesh.vertex(idx[0]), mesh.vertex(idx[1]), mesh.vertex(idx[2]));
            t.m_data.vertex_indices = idx;
            t.m_material = &mesh.material(i);
            out.push_back(t);
        }
}
struct TracerBox {
    std::unique_ptr<MeshWithColors> mesh;
    std::vector<RTTriangle> tris;
    RayTracer rt;
};
}  // namespace

extern "C" {

const char* pth_last_error(void) { return g_err.c_str(); }

void pth_woop_from_positions(int n, const float* pos, float* out12) {
    std::vector<RTTriangle> tris = trianglesFromPositions(n, pos);
    for (int i = 0; i < n; ++i) tris[i].m_data.packRows(out12 + (size_t)i * 12);
}

// stats: totalSplits, badSplits, depthField, trueDepth, buildMicros, flattenMicros
int pth_bvh_build(int n, const float* pos, int sah, int threads, int32_t* out_list, void* out_nodes, int64_t stats[6]) {
    std::vector<RTTriangle> tris = trianglesFromPositions(n, pos);
    BVH bvh(&tris, sah != 0, threads);
    std::memcpy(out_list, bvh.list.data(), bvh.list.size() * 4);
    std::memcpy(out_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(FlatNode));
    if (stats) {
        stats[0] = bvh.numTotalSplits(); stats[1] = bvh.numBadSplits(); stats[2] = bvh.depth; stats[3] = bvh.trueDepth();
        stats[4] = (int64_t)(bvh.buildMillis * 1000.0); stats[5] = (int64_t)(bvh.flattenMillis * 1000.0);
    }
    return (int)bvh.nodes.size();
}

void pth_camera_make(const float eye[3], const float fwd[3], const float up[3], float fov, float nearD, float farD, int width, int height,
                     ptgpu_camera* out) {
    CameraControls cam(Vec3f(eye[0], eye[1], eye[2]), Vec3f(fwd[0], fwd[1], fwd[2]), Vec3f(up[0], up[1], up[2]), fov, nearD, farD);
    *out = makeCameraPod(cam, width, height);
}

void pth_light_make(const float pos[3], const float orient_cols[9], const float size[2], const float E[3], float intensity, ptgpu_light* out) {
    AreaLight l(Vec3f(E[0], E[1], E[2]), Vec2f(size[0], size[1]), intensity);
    Mat3f R;
    std::memcpy(R.m, orient_cols, 36);
    l.setOrientation(R);
    l.setPosition(Vec3f(pos[0], pos[1], pos[2]));
    l.updatePower();
    *out = l.toPod();
}

// AreaLight::sample through the host class, RNG seeded with (k0..k3)
void pth_light_sample(const ptgpu_light* pod, int base, int bounce, const uint32_t k[4], int disk, float* pdf, float p[3]) {
    AreaLight l(Vec3f(pod->E[0], pod->E[1], pod->E[2]), Vec2f(pod->size[0], pod->size[1]), pod->intensity);
    Mat4f X;
    std::memcpy(X.m, pod->xform, 64);
    l.setOrientation(X.getXYZ());
    l.setPosition(X.getCol(3).getXYZ());
    Random R(k[0], k[1], k[2], k[3]);
    Vec3f P;
    if (disk) l.sampleDisk(*pdf, P, base, bounce, R); else l.sample(*pdf, P, base, bounce, R);
    std::memcpy(p, &P, 12);
}

void pth_map_to_disk(float a, float b, float* x, float* y) { mapToDisk(a, b, *x, *y); }
void pth_form_basis(const float n[3], float out9[9]) { Mat3f R = formBasis(Vec3f(n[0], n[1], n[2])); std::memcpy(out9, R.m, 36); }
float pth_sobol_sample(uint64_t index, uint32_t dim, uint32_t scramble) { return sobol::sample(index, dim, scramble); }
uint32_t pth_random_u32(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t draw) {
    Random R(k0, k1, k2, k3);
    uint32_t v = 0;
    for (uint32_t i = 0; i <= draw; ++i) v = R.getU32();
    return v;
}
void pth_md5(const void* data, uint64_t len, char out33[33]) { std::memcpy(out33, md5Hex(data, len).c_str(), 33); }

// ---- image files (ImageIO.cpp): decoded 8-bit RGBA texture, picture out ----
int pth_image_load(const char* path, void** out_handle, int32_t wh[2]) {
    return guarded([&] {
        std::string err;
        std::shared_ptr<Image> img = importImage(path, &err);
        if (!img) throw std::runtime_error(err);
        wh[0] = img->getSize().x; wh[1] = img->getSize().y;
        *out_handle = new std::shared_ptr<Image>(img);
    });
}
void pth_image_copy(void* h, uint8_t* out_rgba) {
    const Image& img = **static_cast<std::shared_ptr<Image>*>(h);
    std::memcpy(out_rgba, img.getPtr(), (size_t)img.getSize().x * img.getSize().y * 4);
}
void pth_image_free(void* h) { delete static_cast<std::shared_ptr<Image>*>(h); }
// rgba: [h][w][4] floats in [0,1] (a resolved picture); written as 8-bit RGB PNG (.png) or PPM (.ppm)
int pth_image_save(const char* path, const float* rgba, int32_t w, int32_t h) {
    return guarded([&] {
        Image img(Vec2i(w, h), ImageFormat::RGBA_Vec4f);
        std::memcpy(img.getMutablePtr(), rgba, (size_t)w * h * 16);
        std::string err, p = path;
        const bool ok = (p.size() > 4 && p.substr(p.size() - 4) == ".ppm") ? exportPPM(p, img, &err) : exportPNG(p, img, &err);
        if (!ok) throw std::runtime_error(err);
    });
}

// ---- RayTracer over a mesh file ----
int pth_rt_open(const char* mesh_path, int sah, void** out) {
    return guarded([&] {
        auto box = std::make_unique<TracerBox>();
        std::string err;
        box->mesh = importMesh(mesh_path, &err);
        if (!box->mesh) throw std::runtime_error(err);
        trianglesFromMesh(*box->mesh, box->tris);
        box->rt.constructHierarchy(box->tris, sah != 0);
        *out = box.release();
    });
}
void pth_rt_close(void* h) { delete static_cast<TracerBox*>(h); }
int pth_rt_counts(void* h, int32_t out[4]) {
    TracerBox* b = static_cast<TracerBox*>(h);
    out[0] = (int32_t)b->tris.size(); out[1] = (int32_t)b->rt.bvh.nodes.size(); out[2] = (int32_t)b->rt.bvh.list.size(); out[3] = b->rt.bvh.depth;
    return 0;
}
int pth_rt_get_bvh(void* h, int32_t* out_list, void* out_nodes) {
    TracerBox* b = static_cast<TracerBox*>(h);
    std::memcpy(out_list, b->rt.bvh.list.data(), b->rt.bvh.list.size() * 4);
    std::memcpy(out_nodes, b->rt.bvh.nodes.data(), b->rt.bvh.nodes.size() * sizeof(FlatNode));
    return 0;
}
// the triangle array App::constructTracer builds (reference src/App.cpp:826-838): pos/nrm [n][3][3], uv [n][3][2], submesh (material) index [n]
int pth_rt_get_triangles(void* h, float* pos, float* nrm, float* uv, int32_t* mat_id) {
    TracerBox* b = static_cast<TracerBox*>(h);
    for (size_t i = 0; i < b->tris.size(); ++i) {
        for (int k = 0; k < 3; ++k) {
            const VertexPNTC& v = b->tris[i].m_vertices[k];
            std::memcpy(pos + i * 9 + k * 3, &v.p, 12);
            std::memcpy(nrm + i * 9 + k * 3, &v.n, 12);
            std::memcpy(uv + i * 6 + k * 2, &v.t, 8);
        }
        int m = -1;
        for (int sub = 0; sub < b->mesh->numSubmeshes(); ++sub)
            if (b->tris[i].m_material == &b->mesh->material(sub)) m = sub;
        mat_id[i] = m;
    }
    return 0;
}
int pth_rt_num_materials(void* h) { return static_cast<TracerBox*>(h)->mesh->numSubmeshes(); }
// material i: out_f = diffuse rgba, specular rgb, glossiness; out_tex = {width, height} of the diffuse and of the normal map (0 0 = none)
int pth_rt_material(void* h, int i, char name64[64], float out_f[8], int32_t out_tex[4]) {
    TracerBox* b = static_cast<TracerBox*>(h);
    if (i < 0 || i >= b->mesh->numSubmeshes()) return -1;
    const MeshBase::Material& m = b->mesh->material(i);
    std::snprintf(name64, 64, "%s", m.name.c_str());
    std::memcpy(out_f, &m.diffuse, 16); std::memcpy(out_f + 4, &m.specular, 12); out_f[7] = m.glossiness;
    const Texture* t[2] = {&m.textures[MeshBase::TextureType_Diffuse], &m.textures[MeshBase::TextureType_Normal]};
    for (int k = 0; k < 2; ++k) {
        out_tex[2 * k] = t[k]->exists() ? t[k]->getImage()->getSize().x : 0;
        out_tex[2 * k + 1] = t[k]->exists() ? t[k]->getImage()->getSize().y : 0;
    }
    return 0;
}
int pth_rt_save_hierarchy(void* h, const char* path) { return guarded([&] { auto* b = static_cast<TracerBox*>(h); b->rt.saveHierarchy(path, b->tris); }); }
int pth_rt_load_hierarchy(void* h, const char* path) { return guarded([&] { auto* b = static_cast<TracerBox*>(h); b->rt.loadHierarchy(path, b->tris); }); }
// RayTracer::constructHierarchyOnDevice (not in the reference): replace the hierarchy by the device-built linear BVH
int pth_rt_construct_on_device(void* h, int max_tris_in_leaf, int keep_tree) {
    return guarded([&] { auto* b = static_cast<TracerBox*>(h); b->rt.constructHierarchyOnDevice(b->tris, max_tris_in_leaf, keep_tree != 0); });
}
int pth_rt_mesh_md5(void* h, char out33[33]) {
    TracerBox* b = static_cast<TracerBox*>(h);
    std::vector<Vec3f> p;
    for (int i = 0; i < b->mesh->numVertices(); ++i) p.push_back(b->mesh->vertex(i).p);
    std::memcpy(out33, RayTracer::computeMD5(p).c_str(), 33);
    return 0;
}
// RayTracer::raycast for n rays; single != 0 calls raycast() once per ray (the reference's signature), else raycastBatch()
int pth_rt_raycast(void* h, const float* orig, const float* dir, int64_t n, int occlusion, int single, int32_t* id, float* t, float* u,
                   float* v, float* point) {
    return guarded([&] {
        TracerBox* b = static_cast<TracerBox*>(h);
        std::vector<RaycastResult> res(n);
        const Vec3f* O = reinterpret_cast<const Vec3f*>(orig);
        const Vec3f* D = reinterpret_cast<const Vec3f*>(dir);
        if (single) for (int64_t i = 0; i < n; ++i) res[i] = b->rt.raycast(O[i], D[i], occlusion != 0);
        else b->rt.raycastBatch(O, D, (size_t)n, occlusion != 0, res.data());
        for (int64_t i = 0; i < n; ++i) {
            id[i] = res[i].tri ? (int32_t)(res[i].tri - b->tris.data()) : -1;
            t[i] = res[i].t; u[i] = res[i].u; v[i] = res[i].v;
            if (point) std::memcpy(point + i * 3, &res[i].point, 12);
        }
    });
}
// PathTraceRenderer over the tracer: settings through the reference's setters, `passes` passes, then updatePicture.
// out_accum / out_resolved: [h][w][4] floats (either may be NULL).  stats: [0] total rays, [1] passes done.
int pth_pt_render(void* h, const float eye[3], const float fwd[3], const float up[3], float fov, float nearD, float farD, int width, int height,
                  const ptgpu_light* lights, int num_lights, const ptgpu_params* prm, int passes, int block_begin, int block_stride,
                  float* out_accum, float* out_resolved, uint64_t stats[2]) {
    return guarded([&] {
        TracerBox* b = static_cast<TracerBox*>(h);
        CameraControls cam(Vec3f(eye[0], eye[1], eye[2]), Vec3f(fwd[0], fwd[1], fwd[2]), Vec3f(up[0], up[1], up[2]), fov, nearD, farD);
        std::vector<AreaLight> ls;
        for (int i = 0; i < num_lights; ++i) {
            const ptgpu_light& p = lights[i];
            AreaLight l(Vec3f(p.E[0], p.E[1], p.E[2]), Vec2f(p.size[0], p.size[1]), p.intensity);
            Mat4f X;
            std::memcpy(X.m, p.xform, 64);
            l.setOrientation(X.getXYZ());
            l.setPosition(X.getCol(3).getXYZ());
            l.updatePower();
            ls.push_back(l);
        }
        PathTraceRenderer r;
        r.setAAmode(prm->sobol_block != 0); r.setAArays(prm->num_aa_rays); r.setAperture(prm->aperture); r.setAttenuation(prm->attenuation);
        r.setBilinearFiltering(prm->bilinear != 0); r.setDOF(prm->depth_of_field != 0); r.setFilterWidth(prm->filter_width);
        r.setFocalDistance(prm->focal_distance); r.setIntensity(prm->intensity); r.setLightRays(prm->num_light_rays);
        r.setNormalMapped(prm->normal_mapped != 0); r.setTermination(prm->termination_prob); r.setWhitted(prm->whitted != 0);
        r.setBlockSubset(block_begin, block_stride);
        r.setPrintStats(false);
        r.setMaxPasses(passes);
        b->rt.useAlpha = prm->use_alpha != 0;
        Image dest(Vec2i(width, height), ImageFormat::RGBA_Vec4f);
        r.startPathTracingProcess(b->mesh.get(), &ls, &b->rt, &dest, prm->bounces, cam);
        while (r.isRunning()) r.waitPass();
        r.updatePicture(&dest);
        size_t n = (size_t)width * height * 16;
        if (out_accum) std::memcpy(out_accum, r.m_context.m_image->getPtr(), n);
        if (out_resolved) std::memcpy(out_resolved, dest.getPtr(), n);
        if (stats) { stats[0] = r.getTotalRays(); stats[1] = (uint64_t)r.m_context.m_pass; }
    });
}

// startPathTracingProcess, then PathTraceRenderer::stop() after stop_after_ms while the first pass is still in flight (reference
// :918-930).  out_ms[0] = how long stop() took, out_ms[1] = 1 if the pass was still running when stop() was called.  out_accum: the
// accumulation image as stop() left it.
int pth_pt_render_stop(void* h, const float eye[3], const float fwd[3], const float up[3], float fov, float nearD, float farD, int width, int height,
                       const ptgpu_light* lights, int num_lights, const ptgpu_params* prm, int stop_after_ms, float* out_accum, double out_ms[2]) {
    return guarded([&] {
        TracerBox* b = static_cast<TracerBox*>(h);
        CameraControls cam(Vec3f(eye[0], eye[1], eye[2]), Vec3f(fwd[0], fwd[1], fwd[2]), Vec3f(up[0], up[1], up[2]), fov, nearD, farD);
        std::vector<AreaLight> ls;
        for (int i = 0; i < num_lights; ++i) {
            const ptgpu_light& p = lights[i];
            AreaLight l(Vec3f(p.E[0], p.E[1], p.E[2]), Vec2f(p.size[0], p.size[1]), p.intensity);
            Mat4f X;
            std::memcpy(X.m, p.xform, 64);
            l.setOrientation(X.getXYZ());
            l.setPosition(X.getCol(3).getXYZ());
            l.updatePower();
            ls.push_back(l);
        }
        PathTraceRenderer r;
        r.setAAmode(prm->sobol_block != 0); r.setAArays(prm->num_aa_rays); r.setAperture(prm->aperture); r.setAttenuation(prm->attenuation);
        r.setBilinearFiltering(prm->bilinear != 0); r.setDOF(prm->depth_of_field != 0); r.setFilterWidth(prm->filter_width);
        r.setFocalDistance(prm->focal_distance); r.setIntensity(prm->intensity); r.setLightRays(prm->num_light_rays);
        r.setNormalMapped(prm->normal_mapped != 0); r.setTermination(prm->termination_prob); r.setWhitted(prm->whitted != 0);
        r.setPrintStats(false);
        b->rt.useAlpha = prm->use_alpha != 0;
        Image dest(Vec2i(width, height), ImageFormat::RGBA_Vec4f);
        r.startPathTracingProcess(b->mesh.get(), &ls, &b->rt, &dest, prm->bounces, cam);
        std::this_thread::sleep_for(std::chrono::milliseconds(stop_after_ms));
        r.checkFinish();  // as the reference's UI loop does; starts no new pass unless this one is over
        out_ms[1] = (r.isRunning() && r.m_context.m_pass == 0) ? 1.0 : 0.0;
        const auto t0 = std::chrono::steady_clock::now();
        r.stop();
        out_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (r.isRunning()) throw GpuError("stop() left the renderer running");
        r.updatePicture(&dest);
        if (out_accum) std::memcpy(out_accum, r.m_context.m_image->getPtr(), (size_t)width * height * 16);
    });
}

// The static per-sample functions of PathTraceRenderer, called the way the reference calls them (one sample / hit / position / block per
// call).  seeds [n][4]: Random(k0,k1,k2,k3) per sample; advance [n]: draws made on the stream before the call.
// out_rgb [n][3], out_ctr [n]: tracePath / traceWhitted results and the stream counters afterwards.  light_out [n][2]: getLightToSample at
// the sample's (x, y, 0.3) and the counter after it.  tex_out [n][9]: getTextureParameters of the primary hit of ray n (zero if it misses).
// block_accum: pathTraceBlock(block_idx) into a zeroed context image [h][w][4] (may be NULL).
int pth_pt_statics(void* h, const float eye[3], const float fwd[3], const float up[3], float fov, float nearD, float farD, int width, int height,
                   const ptgpu_light* lights, int num_lights, const ptgpu_params* prm, int n, const float* xy, const int32_t* base,
                   const uint32_t* seeds, const uint32_t* advance, float* out_rgb, uint32_t* out_ctr, int32_t* light_out, float* tex_out,
                   int block_idx, float* block_accum) {
    return guarded([&] {
        TracerBox* b = static_cast<TracerBox*>(h);
        CameraControls cam(Vec3f(eye[0], eye[1], eye[2]), Vec3f(fwd[0], fwd[1], fwd[2]), Vec3f(up[0], up[1], up[2]), fov, nearD, farD);
        std::vector<AreaLight> ls;
        for (int i = 0; i < num_lights; ++i) {
            const ptgpu_light& p = lights[i];
            AreaLight l(Vec3f(p.E[0], p.E[1], p.E[2]), Vec2f(p.size[0], p.size[1]), p.intensity);
            Mat4f X;
            std::memcpy(X.m, p.xform, 64);
            l.setOrientation(X.getXYZ());
            l.setPosition(X.getCol(3).getXYZ());
            l.updatePower();
            ls.push_back(l);
        }
        PathTraceRenderer r;
        r.setAAmode(prm->sobol_block != 0); r.setAArays(prm->num_aa_rays); r.setAperture(prm->aperture); r.setAttenuation(prm->attenuation);
        r.setBilinearFiltering(prm->bilinear != 0); r.setDOF(prm->depth_of_field != 0); r.setFilterWidth(prm->filter_width);
        r.setFocalDistance(prm->focal_distance); r.setIntensity(prm->intensity); r.setLightRays(prm->num_light_rays);
        r.setNormalMapped(prm->normal_mapped != 0); r.setTermination(prm->termination_prob); r.setWhitted(prm->whitted != 0);
        b->rt.useAlpha = prm->use_alpha != 0;
        Image dest(Vec2i(width, height), ImageFormat::RGBA_Vec4f);
        r.prepareContext(&ls, &b->rt, &dest, prm->bounces, cam);
        r.m_context.m_passSeed = prm->pass_seed;
        std::vector<PathVisualizationNode> vis;
        for (int i = 0; i < n; ++i) {
            Random R(seeds[4 * i], seeds[4 * i + 1], seeds[4 * i + 2], seeds[4 * i + 3]);
            R.setCounter(advance[i]);
            Vec3f c = prm->whitted ? PathTraceRenderer::traceWhitted(xy[2 * i], xy[2 * i + 1], r.m_context, base[i], R, vis)
                                   : PathTraceRenderer::tracePath(xy[2 * i], xy[2 * i + 1], r.m_context, base[i], R, vis);
            std::memcpy(out_rgb + 3 * i, &c, 12);
            out_ctr[i] = R.counter();
            if (light_out) {
                Random R2(seeds[4 * i], seeds[4 * i + 1], seeds[4 * i + 2], seeds[4 * i + 3]);
                R2.setCounter(advance[i]);
                light_out[2 * i] = PathTraceRenderer::getLightToSample(Vec3f(xy[2 * i], xy[2 * i + 1], 0.3f), &ls, R2);
                light_out[2 * i + 1] = (int32_t)R2.counter();
            }
        }
        if (block_accum) {
            MulticoreLauncher::Task t;
            t.data = &r.m_context;
            t.idx = block_idx;
            r.m_context.m_image->clear();
            PathTraceRenderer::pathTraceBlock(t);
            std::memcpy(block_accum, r.m_context.m_image->getPtr(), (size_t)width * height * 16);
        }
    });
}
// getTextureParameters(hit, ...) for the hits of n rays traced with RayTracer::raycastBatch; out [n][9], zero rows for misses
int pth_pt_texture_parameters(void* h, int bilinear, int normal_mapped, const float* orig, const float* dir, int64_t n, float* out) {
    return guarded([&] {
        TracerBox* b = static_cast<TracerBox*>(h);
        PathTraceRenderer r;
        r.setBilinearFiltering(bilinear != 0);
        r.setNormalMapped(normal_mapped != 0);
        PathTraceRenderer::setActiveTracer(&b->rt);
        std::vector<RaycastResult> res(n);
        b->rt.raycastBatch(reinterpret_cast<const Vec3f*>(orig), reinterpret_cast<const Vec3f*>(dir), (size_t)n, false, res.data());
        for (int64_t i = 0; i < n; ++i) {
            float* o = out + i * 9;
            std::memset(o, 0, 36);
            if (!res[i].tri) continue;
            Vec3f d, nn, sp;
            PathTraceRenderer::getTextureParameters(res[i], d, nn, sp);
            std::memcpy(o, &d, 12); std::memcpy(o + 3, &nn, 12); std::memcpy(o + 6, &sp, 12);
        }
    });
}

int pth_rt_ray_count(void* h) { return static_cast<TracerBox*>(h)->rt.getRayCount(); }
void pth_rt_reset_ray_counter(void* h) { static_cast<TracerBox*>(h)->rt.resetRayCounter(); }

}  // extern "C"
