// PathTraceRenderer with the reference's interface (reference src/PathTraceRenderer.hpp:26-139).  The pass over the 32x32
// blocks that the reference hands to a CPU thread pool (m_launcher.push(pathTraceBlock, ...), src/PathTraceRenderer.cpp:
// 847-852) is ONE asynchronous device pass here (ptgpu_render_pass_dev); tracePath / pathTraceBlock / getTextureParameters /
// getLightToSample run inside the gfx950 kernels, so they are not host functions any more.  Start / poll / stop keep the
// reference's protocol: startPathTracingProcess returns at once, checkFinish() notices a finished pass, does the pass
// book-keeping (pass count, total rays, Mrays/s) and starts the next one, stop() ends the loop.
#pragma once
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "3d/CameraControls.hpp"
#include "3d/Mesh.hpp"
#include "AreaLight.hpp"
#include "RayTracer.hpp"
#include "gpu/Device.hpp"

namespace FW {

// one 32x32 (edge-clipped) block of the image (reference src/PathTraceRenderer.hpp:26-31)
struct PathTracerBlock {
    int m_x, m_y, m_width, m_height;
};

// Task handed to a worker by FW::MulticoreLauncher (absent from the reference's tree; the renderer's static pathTraceBlock keeps
// its signature, reference src/PathTraceRenderer.hpp:102): data = PathTracerContext*, idx = block index.
struct MulticoreLauncher {
    struct Task {
        void* data = nullptr;
        int idx = 0;
    };
};

// debug path visualisation records (reference src/PathTraceRenderer.hpp:56-79): GUI feature, kept so that tracePath's signature holds
struct PathVisualizationLine { Vec3f start, stop, color; };
struct PathVisualizationLabel { std::string text; Vec3f position; };
struct PathVisualizationNode {
    std::vector<PathVisualizationLabel> labels;
    std::vector<PathVisualizationLine> lines;
};

struct PathTracerContext {
    AreaLight* m_light = nullptr;
    const CameraControls* m_camera = nullptr;
    const MeshWithColors* m_scene = nullptr;
    Image* m_destImage = nullptr;
    RayTracer* m_rt = nullptr;
    std::unique_ptr<Image> m_image;  // accumulation image (host copy, refreshed by updatePicture)
    std::vector<AreaLight>* m_lights = nullptr;
    std::vector<PathTracerBlock> m_blocks;
    bool m_bForceExit = false;
    bool m_bResidual = false;
    int m_bounces = 0;
    int m_pass = 0;
    uint32_t m_passSeed = 0;  // stands in for the reference's process-global seed counter (:689-691): advanced once per pass
    std::chrono::steady_clock::time_point start;
};

// what tracePath reads from the camera per sample, computed once per render
ptgpu_camera makeCameraPod(const CameraControls& cam, int width, int height);

class PathTraceRenderer {
public:
    PathTraceRenderer();
    ~PathTraceRenderer();

    static bool debugVis;  // path visualisation is a GUI feature: accepted, ignored
    PathTracerContext m_context;

    bool isRunning(void) const { return m_running; }

    // negative #bounces = -N: Russian roulette after N bounces; positive N: exactly N bounces
    void startPathTracingProcess(const MeshWithColors* scene, std::vector<AreaLight>* lights, RayTracer* rt, Image* dest, int bounces,
                                 const CameraControls& camera);
    // The reference's static per-sample functions (src/PathTraceRenderer.hpp:97-102), same signatures.  Each is ONE call of the device
    // code the render pass runs (ptgpu_trace_samples / ptgpu_light_to_sample_batch / ptgpu_texture_parameters_batch /
    // ptgpu_render_pass over one block): a launch per call, meant for probing and tests, not for rendering.  R is advanced by the
    // draws the sample consumed, as when the reference passes its Random by reference.  visualization is accepted and left empty.
    static Vec3f tracePath(float x, float y, PathTracerContext& ctx, int samplerBase, Random& rnd, std::vector<PathVisualizationNode>& visualization);
    static Vec3f traceWhitted(float x, float y, PathTracerContext& ctx, int samplerBase, Random& rnd, std::vector<PathVisualizationNode>& visualization);
    static int getLightToSample(const Vec3f pos, const std::vector<AreaLight>* lights, Random& R);
    // the hit's triangle must belong to the tracer of the last startPathTracingProcess / setActiveTracer call
    static void getTextureParameters(const RaycastResult& hit, Vec3f& diffuse, Vec3f& n, Vec3f& specular);
    static void pathTraceBlock(MulticoreLauncher::Task& t);  // accumulates block t.idx into ctx.m_image (host image)
    static void setActiveTracer(RayTracer* rt) { s_rt = rt; }
    // fills a context for the static functions without starting a render
    void prepareContext(std::vector<AreaLight>* lights, RayTracer* rt, Image* dest, int bounces, const CameraControls& camera);

    void checkFinish(void);
    void stop(void);
    void updatePicture(Image* display);

    void resetRays() { m_TotalRays = 0; }
    void setAAmode(bool b) { m_sobolBlock = b; }
    void setAArays(int a) { m_NumAaRays = a; }
    void setAperture(float a) { m_aperture = a; }
    void setAttenuation(float a) { m_attenuation = a; }
    void setBilinearFiltering(bool b) { m_bilinearFiltering = b; }
    void setDOF(bool b) { m_depthOfField = b; }
    void setFilterWidth(float a) { m_filterWidth = a; }
    void setFocalDistance(float a) { m_focalDistance = a; }
    void setIntensity(float a) { m_intensity = a; }
    void setLightRays(int a) { m_numLightRays = a; }
    void setNormalMapped(bool b) { m_normalMapped = b; }
    void setTermination(float a) { m_terminationProbability = a; }
    void setWhitted(bool b) { m_whitted = b; }

    // ---- additions of this build (none changes the reference's behaviour when left alone) ----
    // Multi-GPU tiles: this process renders blocks begin, begin+stride, ... (rank r of N: begin = r, stride = N)
    void setBlockSubset(int begin, int stride) { m_blockBegin = begin; m_blockStride = stride; }
    // One process per GPU: this process is rank `rank` of `world`; every pass renders its round-robin share of the blocks and ends with
    // ptgpu_gather_tiles_dev (RCCL all-gather), so updatePicture shows the full image on every rank.  commId: the
    // PTGPU_COMM_ID_BYTES bytes rank 0 got from ptgpu_comm_create_id, the same on all ranks (collective call).
    void setTileRanks(int rank, int world, const void* commId);
    void setPrintStats(bool b) { m_printStats = b; }
    // The reference writes "pt-%03dppp.png" after every pass (src/PathTraceRenderer.cpp:900-902); here that is opt-in (a library should
    // not litter the working directory): directory to write into, empty = off.
    void setOutputImages(const std::string& dir) { m_outputDir = dir; }
    void setMaxPasses(int n) { m_maxPasses = n; }  // 0 = run until stop() like the reference
    void waitPass();                               // block until the pass in flight has finished, then checkFinish()
    unsigned long long getTotalRays() const { return m_TotalRays; }
    float getRaysPerSecond() const { return m_raysPerSecond; }
    const float* deviceAccumulation() const { return m_accum.ptr(); }
    ptgpu_params currentParams() const;

protected:
    void launchPass();
    static ptgpu_params paramsOf(const PathTracerContext& ctx);
    static Vec3f traceOne(float x, float y, PathTracerContext& ctx, int samplerBase, Random& rnd, bool whitted);
    static RayTracer* s_rt;

    float m_raysPerSecond = 0.0f;
    static bool m_bilinearFiltering;
    static bool m_depthOfField;
    static bool m_normalMapped;
    static bool m_sobolBlock;
    static bool m_whitted;
    static float m_aperture;
    static float m_attenuation;
    static float m_filterWidth;
    static float m_focalDistance;
    static float m_intensity;
    static float m_terminationProbability;
    static int m_NumAaRays;
    static int m_numLightRays;
    unsigned long long m_TotalRays = 0;

    bool m_running = false, m_passInFlight = false, m_printStats = true;
    int m_blockBegin = 0, m_blockStride = 1, m_maxPasses = 0;
    bool m_gatherTiles = false;
    std::string m_outputDir;
    GpuBuffer<float> m_accum, m_resolved;
    ptgpu_camera m_cameraPod{};
    std::vector<ptgpu_light> m_lightPods;
};

}  // namespace FW
