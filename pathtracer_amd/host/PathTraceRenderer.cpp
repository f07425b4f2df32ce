#include "PathTraceRenderer.hpp"

#include <cstdio>
#include <cstring>
#include <thread>

namespace FW {

// defaults of the tunables (reference src/PathTraceRenderer.cpp:15-28)
bool PathTraceRenderer::debugVis = false;
bool PathTraceRenderer::m_bilinearFiltering = true;
bool PathTraceRenderer::m_depthOfField = false;
bool PathTraceRenderer::m_normalMapped = false;
bool PathTraceRenderer::m_sobolBlock = false;
bool PathTraceRenderer::m_whitted = false;
float PathTraceRenderer::m_aperture = 0.0f;
float PathTraceRenderer::m_attenuation = 1.0f;
float PathTraceRenderer::m_filterWidth = 1.0f;
float PathTraceRenderer::m_focalDistance = 0.5f;
float PathTraceRenderer::m_intensity = 1.0f;
float PathTraceRenderer::m_terminationProbability = 0.2f;
int PathTraceRenderer::m_NumAaRays = 4;
int PathTraceRenderer::m_numLightRays = 32;

PathTraceRenderer::PathTraceRenderer() {}
PathTraceRenderer::~PathTraceRenderer() {
    try { stop(); } catch (...) {}
}

// What tracePath derives from the camera for every sample (reference src/PathTraceRenderer.cpp:244-248, 271, 286-291, 398),
// computed once per render.
ptgpu_camera makeCameraPod(const CameraControls& cam, int width, int height) {
    ptgpu_camera out;
    Mat4f projection = Mat4f::fitToView(Vec2f(-1, -1), Vec2f(2, 2), Vec2f(Vec2i(width, height))) * cam.getCameraToClip();
    Mat4f invP = (projection * cam.getWorldToCamera()).inverted();
    std::memcpy(out.invP, invP.m, 64);
    Vec3f f = cam.getForward(), c0 = cam.getOrientation().getCol(0), c1 = cam.getOrientation().getCol(1);
    std::memcpy(out.forward, &f, 12);
    std::memcpy(out.orient_col0, &c0, 12);
    std::memcpy(out.orient_col1, &c1, 12);
    out.far_ = cam.getFar();
    return out;
}

RayTracer* PathTraceRenderer::s_rt = nullptr;

ptgpu_params PathTraceRenderer::paramsOf(const PathTracerContext& ctx) {
    ptgpu_params p{};
    Vec2i size = ctx.m_destImage ? ctx.m_destImage->getSize() : Vec2i(0, 0);
    p.width = size.x; p.height = size.y;
    p.bounces = ctx.m_bounces;
    p.num_aa_rays = m_NumAaRays;
    p.sobol_block = m_sobolBlock; p.bilinear = m_bilinearFiltering; p.normal_mapped = m_normalMapped;
    p.depth_of_field = m_depthOfField; p.whitted = m_whitted; p.num_light_rays = m_numLightRays;
    p.use_alpha = (ctx.m_rt && ctx.m_rt->useAlpha) ? 1 : 0;
    p.pass_seed = ctx.m_passSeed;
    p.aperture = m_aperture; p.attenuation = m_attenuation; p.filter_width = m_filterWidth;
    p.focal_distance = m_focalDistance; p.intensity = m_intensity; p.termination_prob = m_terminationProbability;
    return p;
}
ptgpu_params PathTraceRenderer::currentParams() const { return paramsOf(m_context); }

namespace {
std::vector<ptgpu_light> lightPods(const std::vector<AreaLight>* lights) {
    std::vector<ptgpu_light> pods;
    if (lights) for (const AreaLight& l : *lights) pods.push_back(l.toPod());
    return pods;
}
void requireContext(const PathTracerContext& ctx, const char* who) {
    if (!ctx.m_rt || !ctx.m_camera || !ctx.m_destImage || !ctx.m_lights || ctx.m_lights->empty())
        throw GpuError(std::string(who) + ": the context needs a tracer, camera, destination image and at least one light");
}
}  // namespace

void PathTraceRenderer::prepareContext(std::vector<AreaLight>* lights, RayTracer* rt, Image* dest, int bounces, const CameraControls& camera) {
    m_context.m_camera = &camera;
    m_context.m_rt = rt;
    m_context.m_lights = lights;
    m_context.m_light = (lights && !lights->empty()) ? &(*lights)[0] : nullptr;
    m_context.m_bounces = bounces;
    m_context.m_destImage = dest;
    if (dest && (!m_context.m_image || !(m_context.m_image->getSize() == dest->getSize())))
        m_context.m_image.reset(new Image(dest->getSize(), ImageFormat::RGBA_Vec4f));
    s_rt = rt;
}

// tracePath / traceWhitted: one sample through ptgpu_trace_samples (reference :236-551 / :553-660)
Vec3f PathTraceRenderer::traceOne(float x, float y, PathTracerContext& ctx, int samplerBase, Random& rnd, bool whitted) {
    requireContext(ctx, whitted ? "traceWhitted" : "tracePath");
    ptgpu_params p = paramsOf(ctx);
    p.whitted = whitted ? 1 : 0;
    const Vec2i size = ctx.m_destImage->getSize();
    const ptgpu_camera cam = makeCameraPod(*ctx.m_camera, size.x, size.y);
    const std::vector<ptgpu_light> pods = lightPods(ctx.m_lights);
    const float xy[2] = {x, y};
    const int32_t base = samplerBase;
    const uint32_t state[2] = {rnd.key(), rnd.counter()};
    float out[4];
    PT_GPU(ptgpu_trace_samples(ctx.m_rt->deviceScene(), &p, &cam, pods.data(), (int32_t)pods.size(), 1, xy, &base, state, out));
    uint32_t ctr;
    std::memcpy(&ctr, &out[3], 4);
    rnd.setCounter(ctr);
    return Vec3f(out[0], out[1], out[2]);
}
Vec3f PathTraceRenderer::tracePath(float x, float y, PathTracerContext& ctx, int samplerBase, Random& rnd, std::vector<PathVisualizationNode>&) {
    return traceOne(x, y, ctx, samplerBase, rnd, false);
}
Vec3f PathTraceRenderer::traceWhitted(float x, float y, PathTracerContext& ctx, int samplerBase, Random& rnd, std::vector<PathVisualizationNode>&) {
    return traceOne(x, y, ctx, samplerBase, rnd, true);
}

// reference :183-210
int PathTraceRenderer::getLightToSample(const Vec3f pos, const std::vector<AreaLight>* lights, Random& R) {
    const std::vector<ptgpu_light> pods = lightPods(lights);
    if (pods.empty()) throw GpuError("getLightToSample: no lights");
    const uint32_t state[2] = {R.key(), R.counter()};
    int32_t out[2];
    PT_GPU(ptgpu_light_to_sample_batch(GpuDevice::instance().ctx(), pods.data(), (int32_t)pods.size(), 1, &pos.x, state, out));
    R.setCounter((uint32_t)out[1]);
    return out[0];
}

// reference :75-181
void PathTraceRenderer::getTextureParameters(const RaycastResult& hit, Vec3f& diffuse, Vec3f& n, Vec3f& specular) {
    if (!s_rt || !s_rt->m_triangles || !hit.tri) throw GpuError("getTextureParameters: no active tracer or no hit");
    const std::vector<RTTriangle>& T = *s_rt->m_triangles;
    const ptrdiff_t idx = hit.tri - T.data();
    if (idx < 0 || (size_t)idx >= T.size()) throw GpuError("getTextureParameters: the hit's triangle does not belong to the active tracer");
    ptgpu_params p{};
    p.width = p.height = 1; p.num_aa_rays = 1; p.num_light_rays = 1; p.filter_width = 1.0f; p.termination_prob = 1.0f;
    p.bilinear = m_bilinearFiltering; p.normal_mapped = m_normalMapped;
    const int32_t tri = (int32_t)idx;
    float out[9];
    PT_GPU(ptgpu_texture_parameters_batch(s_rt->deviceScene(), &p, 1, &tri, &hit.u, &hit.v, out));
    diffuse = Vec3f(out[0], out[1], out[2]); n = Vec3f(out[3], out[4], out[5]); specular = Vec3f(out[6], out[7], out[8]);
}

// reference :663-790: one block, accumulated into the context's (host) accumulation image
void PathTraceRenderer::pathTraceBlock(MulticoreLauncher::Task& t) {
    PathTracerContext& ctx = *static_cast<PathTracerContext*>(t.data);
    requireContext(ctx, "pathTraceBlock");
    if (!ctx.m_image) throw GpuError("pathTraceBlock: the context has no accumulation image");
    const ptgpu_params p = paramsOf(ctx);
    const ptgpu_camera cam = makeCameraPod(*ctx.m_camera, p.width, p.height);
    const std::vector<ptgpu_light> pods = lightPods(ctx.m_lights);
    PT_GPU(ptgpu_render_pass(ctx.m_rt->deviceScene(), &p, &cam, pods.data(), (int32_t)pods.size(), t.idx, t.idx + 1, 1,
                             (float*)ctx.m_image->getMutablePtr()));
}

void PathTraceRenderer::launchPass() {
    ptgpu_params p = currentParams();
    PT_GPU(ptgpu_render_pass_dev(m_context.m_rt->deviceScene(), &p, &m_cameraPod, m_lightPods.data(), (int32_t)m_lightPods.size(), m_blockBegin, -1,
                                 m_blockStride, m_accum.ptr()));
    if (m_gatherTiles) PT_GPU(ptgpu_gather_tiles_dev(GpuDevice::instance().ctx(), p.width, p.height, m_accum.ptr()));
    ++m_context.m_passSeed;  // stands in for the reference's process-global seed counter (:689-691): every pass draws new samples
    m_passInFlight = true;
}

void PathTraceRenderer::setTileRanks(int rank, int world, const void* commId) {
    if (world <= 1) { m_gatherTiles = false; setBlockSubset(0, 1); return; }
    PT_GPU(ptgpu_comm_init(GpuDevice::instance().ctx(), world, rank, commId));
    setBlockSubset(rank, world);
    m_gatherTiles = true;
}

void PathTraceRenderer::startPathTracingProcess(const MeshWithColors* scene, std::vector<AreaLight>* lights, RayTracer* rt, Image* dest,
                                                int bounces, const CameraControls& camera) {
    FW_ASSERT(!m_context.m_bForceExit);
    if (!rt || !dest || !lights || lights->empty()) throw GpuError("startPathTracingProcess: tracer, destination image and at least one light are required");
    if (m_running) stop();
    m_context.m_bForceExit = false;
    m_context.m_bResidual = false;
    m_context.m_camera = &camera;
    m_context.m_rt = rt;
    m_context.m_scene = scene;
    m_context.m_lights = lights;
    m_context.m_light = &(*lights)[0];
    m_context.m_pass = 0;
    s_rt = rt;
    m_context.m_bounces = bounces;
    m_context.m_image.reset(new Image(dest->getSize(), ImageFormat::RGBA_Vec4f));
    m_context.m_destImage = dest;

    // the reference's block list (:813-841); the device numbers blocks the same way (row-major)
    m_context.m_blocks.clear();
    const int bs = 32, w = dest->getSize().x, h = dest->getSize().y;
    for (int y = 0; y < (h + bs - 1) / bs; ++y)
        for (int x = 0; x < (w + bs - 1) / bs; ++x)
            m_context.m_blocks.push_back(PathTracerBlock{bs * x, bs * y, min(bs * x + bs, w) - bs * x, min(bs * y + bs, h) - bs * y});
    dest->clear();

    m_cameraPod = makeCameraPod(camera, w, h);
    m_lightPods.clear();
    for (const AreaLight& l : *lights) m_lightPods.push_back(l.toPod());
    m_accum.resize((size_t)w * h * 4);
    m_accum.zero();
    m_resolved.resize((size_t)w * h * 4);

    ptgpu_context* ctx = GpuDevice::instance().ctx();
    PT_GPU(ptgpu_reset_counters(ctx));
    m_context.start = std::chrono::steady_clock::now();
    m_running = true;
    launchPass();
}

void PathTraceRenderer::updatePicture(Image* dest) {
    FW_ASSERT(m_context.m_image != 0);
    FW_ASSERT(m_context.m_image->getSize() == dest->getSize());
    FW_ASSERT(dest->getFormat() == ImageFormat::RGBA_Vec4f);
    ptgpu_context* ctx = GpuDevice::instance().ctx();
    const Vec2i size = dest->getSize();
    // resolve + gamma on the device (reference :859-869), then one copy into the display image; stream order puts it after
    // whatever part of the current pass has been enqueued, like the reference's UI thread reading a half-finished pass
    PT_GPU(ptgpu_resolve_dev(ctx, m_accum.ptr(), size.x, size.y, m_resolved.ptr()));
    m_resolved.download((float*)dest->getMutablePtr(), (size_t)size.x * size.y * 4);
    m_accum.download((float*)m_context.m_image->getMutablePtr(), (size_t)size.x * size.y * 4);
}

void PathTraceRenderer::checkFinish() {
    if (!m_running || !m_passInFlight) return;
    ptgpu_context* ctx = GpuDevice::instance().ctx();
    int done = 0;
    PT_GPU(ptgpu_query(ctx, &done));
    if (!done) return;
    m_passInFlight = false;
    ++m_context.m_pass;
    float time = static_cast<float>(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - m_context.start).count()) * 0.001f;

    // 1 ray = 1 raycast() call, counted on the device (reference src/RayTracer.cpp:192); fold it into the tracer's counter
    uint64_t spill[3] = {0, 0, 0};
    PT_GPU(ptgpu_spill_status(ctx, spill));
    if (spill[1]) std::fprintf(stderr, "PathTraceRenderer: %llu out-of-block samples of this pass were dropped (spill list full)\n", (unsigned long long)spill[1]);
    ptgpu_counters c{};
    PT_GPU(ptgpu_get_counters(ctx, &c));
    PT_GPU(ptgpu_reset_counters(ctx));
    m_context.m_rt->addRays((long long)c.rays);
    m_TotalRays += (unsigned long long)m_context.m_rt->getRayCountLong();
    m_raysPerSecond = time > 0.0f ? m_TotalRays / time : 0.0f;
    m_context.m_rt->resetRayCounter();
    if (m_printStats) {
        if (time < 60.0f) std::printf("  Done! time: %5.2f s   ", time);
        else std::printf("  Done! time: %5.2f min   ", time / 60.0f);
        std::printf("Total Mrays: %6.1f (%5.2f Mrays/sec)\n", m_TotalRays / 1000000.0f, m_raysPerSecond / 1000000.0f);
    }
    if (!m_outputDir.empty() && m_context.m_destImage) {
        // after the completion of each full round through the image (reference :900-902)
        updatePicture(m_context.m_destImage);
        char fn[64];
        std::snprintf(fn, sizeof fn, "pt-%03dppp.png", m_context.m_pass);
        std::string err;
        if (!exportPNG(m_outputDir + "/" + fn, *m_context.m_destImage, &err)) std::fprintf(stderr, "PathTraceRenderer: %s\n", err.c_str());
    }
    if (!m_context.m_bForceExit && (m_maxPasses <= 0 || m_context.m_pass < m_maxPasses)) {
        launchPass();
        if (m_printStats) std::printf("Pass %d\n", m_context.m_pass);
    } else {
        m_running = false;
        if (m_printStats && m_context.m_bForceExit) std::printf("\nStopped!\n");
    }
}

void PathTraceRenderer::waitPass() {
    if (!m_running || !m_passInFlight) return;
    PT_GPU(ptgpu_sync(GpuDevice::instance().ctx()));
    checkFinish();
}

void PathTraceRenderer::stop() {
    m_context.m_bForceExit = true;
    if (m_running) {
        // reference :918-930: the workers see m_bForceExit before their next sample (:709,736) and stop() only waits for them to return;
        // no pass is counted and no picture written.  Here the device's persistent waves see ptgpu_cancel's word at their next chunk of
        // samples, batches that are not complete are left out of the accumulation image, and the stream drains in milliseconds.
        if (m_passInFlight) {
            ptgpu_context* ctx = GpuDevice::instance().ctx();
            PT_GPU(ptgpu_cancel(ctx));
            PT_GPU(ptgpu_sync(ctx));
            m_passInFlight = false;
        }
        m_running = false;
        if (m_printStats) std::printf("\nStopped!\n");
    }
    m_context.m_bForceExit = false;
}

}  // namespace FW
