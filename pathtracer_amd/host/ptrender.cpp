// ptrender -- headless driver of the host classes: what App::loadMesh / App::constructTracer / the path-trace mode of
// App::handleEvent do (reference src/App.cpp:146-239, 457-489, 754-908) without the window: load a mesh, build (or load the
// cached) hierarchy, render N passes with PathTraceRenderer on the MI355X, write the picture, print the reference's statistics.
//
//   ptrender <mesh.ptmesh|mesh.obj> [-builder sah|object_median|spatial_median|lbvh] [-spp N] [-passes N] [-size WxH]
//            [-bounces N] [-rr] [-aa sobol|gauss] [-filter W] [-normal_maps] [-dof APERTURE FOCAL] [-whitted] [-light_size S] [-emission E]
//            [-bat_render] [-output_images] [-o out.png|out.ppm] [-cache DIR] [-ranks N -rank R [-comm_id FILE]]
//
// -bat_render: always build the hierarchy (never the cache) and ignore textures, like the reference's batch mode.
// -ranks/-rank: one process per GPU (device = PTGPU_DEVICE or LOCAL_RANK); each renders its round-robin share of the 32x32 blocks.
// With -comm_id FILE the ranks form an RCCL communicator (rank 0 writes the id to FILE, the others wait for it) and every pass ends
// with ptgpu_gather_tiles_dev, so each rank holds -- and rank 0 writes -- the full picture; without it only the own tiles are rendered.
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "PathTraceRenderer.hpp"

using namespace FW;

static bool fileExists(const std::string& f) { return std::ifstream(f).good(); }

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: ptrender <mesh> [-builder sah|object_median|spatial_median|lbvh] [-spp N] [-passes N] [-size WxH] [-bounces N] [-rr]\n"
                             "       [-aa sobol|gauss] [-filter W] [-normal_maps] [-dof APERTURE FOCAL] [-light_size S] [-bat_render] [-output_images]\n"
                             "       [-o out.png|out.ppm] [-cache DIR] [-ranks N -rank R]\n");
        return 2;
    }
    std::string meshPath = argv[1], outPath, cacheDir = ".", commIdFile;
    bool sah = true, lbvh = false, batch = false, outputImages = false, rr = false, sobolAA = true, normalMaps = false, dof = false, whitted = false;
    int spp = 4, passes = 1, width = 512, height = 512, bounces = 1, ranks = 1, rank = 0;
    float filterWidth = 1.0f, aperture = 0.0001f, focal = 0.1f, lightSize = 0.25f, emission = 100.0f;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "-builder") { std::string b = next(); sah = (b == "sah"); lbvh = (b == "lbvh"); if (b != "sah" && b != "lbvh" && b != "object_median" && b != "spatial_median" && b != "none" && b != "linear") std::printf("BVH builder not recognized, using Surface Area Heuristic\n"), sah = true; }
        else if (a == "-spp") spp = std::atoi(next().c_str());
        else if (a == "-passes") passes = std::atoi(next().c_str());
        else if (a == "-size") { std::string s = next(); if (std::sscanf(s.c_str(), "%dx%d", &width, &height) != 2) { std::fprintf(stderr, "-size WxH\n"); return 2; } }
        else if (a == "-bounces") bounces = std::atoi(next().c_str());
        else if (a == "-rr") rr = true;
        else if (a == "-aa") sobolAA = (next() != "gauss");
        else if (a == "-filter") filterWidth = (float)std::atof(next().c_str());
        else if (a == "-normal_maps") normalMaps = true;
        else if (a == "-dof") { dof = true; aperture = (float)std::atof(next().c_str()); focal = (float)std::atof(next().c_str()); }
        else if (a == "-whitted") whitted = true;
        else if (a == "-emission") emission = (float)std::atof(next().c_str());
        else if (a == "-light_size") lightSize = (float)std::atof(next().c_str());
        else if (a == "-bat_render") batch = true;
        else if (a == "-output_images") outputImages = true;
        else if (a == "-use_textures") {}
        else if (a == "-o") outPath = next();
        else if (a == "-cache") cacheDir = next();
        else if (a == "-ranks") ranks = std::atoi(next().c_str());
        else if (a == "-rank") rank = std::atoi(next().c_str());
        else if (a == "-comm_id") commIdFile = next();
        else std::printf("argument \"%s\" not found!\n", a.c_str());
    }
    try {
        std::string err;
        std::unique_ptr<MeshWithColors> mesh = importMesh(meshPath, &err);
        if (!mesh) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (batch)  // the reference's batch mode renders untextured
            for (int i = 0; i < mesh->numSubmeshes(); ++i)
                for (auto& t : mesh->material(i).textures) t = Texture();
        // App::constructTracer: triangle array (reference src/App.cpp:826-838)
        std::vector<RTTriangle> tris;
        tris.reserve(mesh->numTriangles());
        for (int i = 0; i < mesh->numSubmeshes(); ++i)
            for (const Vec3i& idx : mesh->indices(i)) {
                RTTriangle t(mesh->vertex(idx[0]), mesh->vertex(idx[1]), mesh->vertex(idx[2]));
                t.m_data.vertex_indices = idx;
                t.m_material = &mesh->material(i);
                tris.push_back(t);
            }
        std::vector<Vec3f> positions;
        for (int i = 0; i < mesh->numVertices(); ++i) positions.push_back(mesh->vertex(i).p);
        std::string md5 = RayTracer::computeMD5(positions);
        std::printf("Mesh MD5: %s\n", md5.c_str());
        RayTracer rt;
        std::string cacheFile = cacheDir + "/Hierarchy-" + md5 + (sah ? "-sah" : (lbvh ? "-lbvh" : "")) + "-x64.bin";
        if (!batch && fileExists(cacheFile)) {
            rt.loadHierarchy(cacheFile.c_str(), tris);
            std::printf("Loaded hierarchy from %s\n\n", cacheFile.c_str());
        } else {
            auto t0 = std::chrono::steady_clock::now();
            if (lbvh) rt.constructHierarchyOnDevice(tris);  // device-side linear BVH (not in the reference)
            else rt.constructHierarchy(tris, sah);
            int ms = (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            std::printf("Build time: %d ms\n", ms);
            if (!batch) { rt.saveHierarchy(cacheFile.c_str(), tris); std::printf("Saved hierarchy to %s\n", cacheFile.c_str()); }
        }
        std::printf("Triangles: %d  nodes: %d\n", (int)tris.size(), (int)rt.bvh.nodes.size());

        CameraControls camera;  // the SURVEY.md 8d probe view of head.obj
        std::vector<AreaLight> lights(1, AreaLight(Vec3f(emission), Vec2f(lightSize, lightSize), 1.0f));
        Mat3f R;
        R.setCol(0, Vec3f(1, 0, 0)); R.setCol(1, Vec3f(0, 0, -1)); R.setCol(2, Vec3f(0, 1, 0));  // normal = -col2 = straight down
        lights[0].setOrientation(R);
        lights[0].setPosition(Vec3f(0.0f, 0.9f, 0.2f));
        lights[0].updatePower();

        PathTraceRenderer renderer;
        renderer.setAAmode(sobolAA); renderer.setAArays(spp); renderer.setFilterWidth(filterWidth);
        renderer.setNormalMapped(normalMaps); renderer.setBilinearFiltering(true);
        renderer.setDOF(dof); renderer.setAperture(aperture); renderer.setFocalDistance(focal);
        renderer.setWhitted(whitted); renderer.setTermination(0.2f);
        renderer.setBlockSubset(rank, ranks);
        if (ranks > 1 && !commIdFile.empty()) {
            unsigned char id[PTGPU_COMM_ID_BYTES];
            if (rank == 0) {
                PT_GPU(ptgpu_comm_create_id(id));
                std::ofstream f(commIdFile + ".tmp", std::ios::binary);
                f.write((const char*)id, sizeof id);
                f.close();
                std::rename((commIdFile + ".tmp").c_str(), commIdFile.c_str());
            } else {
                for (int tries = 0; !fileExists(commIdFile); ++tries) {
                    if (tries > 1200) { std::fprintf(stderr, "no communicator id in %s\n", commIdFile.c_str()); return 1; }
                    std::this_thread::sleep_for(std::chrono::milliseconds(50));
                }
                std::ifstream f(commIdFile, std::ios::binary);
                if (!f.read((char*)id, sizeof id)) { std::fprintf(stderr, "%s: short communicator id\n", commIdFile.c_str()); return 1; }
            }
            renderer.setTileRanks(rank, ranks, id);
        }
        renderer.setMaxPasses(passes);
        if (outputImages) renderer.setOutputImages(".");
        Image dest(Vec2i(width, height), ImageFormat::RGBA_Vec4f);
        renderer.startPathTracingProcess(mesh.get(), &lights, &rt, &dest, rr ? -bounces : bounces, camera);
        while (renderer.isRunning()) renderer.waitPass();
        renderer.updatePicture(&dest);
        double samples = (double)width * height * spp * passes / ranks;
        std::printf("%d pass(es), %.1f Msamples, %.1f Mrays, %.2f Mrays/sec\n", renderer.m_context.m_pass, samples / 1e6,
                    renderer.getTotalRays() / 1e6, renderer.getRaysPerSecond() / 1e6);
        if (!outPath.empty() && (rank == 0 || commIdFile.empty())) {
            const bool ppm = outPath.size() > 4 && outPath.substr(outPath.size() - 4) == ".ppm";
            if (!(ppm ? exportPPM(outPath, dest, &err) : exportPNG(outPath, dest, &err))) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            std::printf("Wrote %s\n", outPath.c_str());
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ptrender: %s\n", e.what());
        return 1;
    }
    return 0;
}
