// ORACLE -- TEST INFRASTRUCTURE ONLY (see pt_math.h header note).
//
// C ABI of the CPU restatement of Esgrove/pathtracer's per-sample hot path.
// Every function cites the reference file:line it restates.  The POD structs
// are deliberately layout-identical to include/ptgpu.h so tests can hand the
// very same bytes to the oracle and to the HIP library.
//
// Parity status (DESIGN.md "Oracle"): Sobol values and the rtIntersect.inl
// tests are pinned against the reference's own sources compiled into
// oracle/_ref; the BVH builder is pinned by the node/leaf/bad-split counts the
// reference itself printed for head.obj; everything that depends
// on the absent FW framework (Random stream, matrix inverse rounding, camera
// matrices, OBJ import conventions) is "parity unpinned".
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// reference src/BvhNode.hpp:67-113 (sizeof == 64)
typedef struct pto_flatnode {
    float bmin[3], bmax[3];
    int32_t startIdx, endIdx;  // inclusive range into the triangle index list
    uint8_t leaf;
    uint8_t _padleaf[3];
    int32_t rightChild;
    int32_t axis;
    int32_t leftChild;
    int32_t triangles;
    int32_t pad0, pad1, pad2;
} pto_flatnode;

// FW MeshBase::Material subset used by the path; tex_* index the texture table, -1 = none
typedef struct pto_material {
    float diffuse[4];
    float specular[3];
    float glossiness;
    int32_t tex_diffuse, tex_alpha, tex_normal, tex_specular;
} pto_material;

typedef struct pto_texture {
    int32_t width, height;
    const uint8_t* rgba;  // width*height*4, row-major, texel (x,y) at (y*width+x)*4
} pto_texture;

// reference src/AreaLight.hpp:74-79
typedef struct pto_light {
    float xform[16];  // column-major: cols 0,1 tangent/bitangent, col 2 = -normal, col 3 = position
    float size[2];    // half extents
    float E[3];
    float intensity;
    float power;      // updatePower(): 4*sx*sy*|E|*intensity
} pto_light;

// what tracePath reads from CameraControls (reference src/PathTraceRenderer.cpp:244-291)
typedef struct pto_camera {
    float invP[16];       // (fitToView * cameraToClip * worldToCamera)^-1, column-major
    float forward[3];     // getForward()
    float orient_col0[3]; // getOrientation().getCol(0)  ("up" in the reference's DoF code)
    float orient_col1[3]; // getOrientation().getCol(1)  ("right")
    float far_;           // getFar()
} pto_camera;

// the 14 static tunables (reference src/PathTraceRenderer.cpp:15-28) + per-render values
typedef struct pto_params {
    int32_t width, height;
    int32_t bounces;         // <0: Russian roulette after |bounces|
    int32_t num_aa_rays;
    int32_t sobol_block;     // m_sobolBlock
    int32_t bilinear;        // m_bilinearFiltering
    int32_t normal_mapped;   // m_normalMapped
    int32_t depth_of_field;  // m_depthOfField
    int32_t whitted;         // m_whitted
    int32_t num_light_rays;  // m_numLightRays
    int32_t use_alpha;       // RayTracer::useAlpha
    uint32_t pass_seed;      // replaces the reference's global atomic seed counter (PathTraceRenderer.cpp:689-691)
    float aperture, attenuation, filter_width, focal_distance, intensity, termination_prob;
} pto_params;

typedef struct pto_scene {
    int32_t num_tris;
    const float* pos;       // [num_tris][3 verts][3]
    const float* nrm;       // [num_tris][3][3]
    const float* uv;        // [num_tris][3][2]
    const int32_t* mat_id;  // [num_tris]
    const float* woop;      // [num_tris][3 rows][4] = (M row r | N[r])
    int32_t num_materials;
    const pto_material* materials;
    int32_t num_textures;
    const pto_texture* textures;
    int32_t num_nodes;
    const pto_flatnode* nodes;
    int32_t list_size;
    const int32_t* list;
} pto_scene;

typedef struct pto_counters {
    uint64_t rays, node_visits, tri_tests;
} pto_counters;

// --- Sobol (reference src/sobol.h:39-49) ---
float pto_sobol_sample(uint64_t index, uint32_t dimension, uint32_t scramble);
const uint32_t* pto_sobol_matrices(void);  // 1024*52 entries

// --- small pure functions, exported for known-answer tests ---
void pto_map_to_disk(float a, float b, float* x, float* y, int libm);   // src/util.cpp:22-47
void pto_form_basis(const float n[3], float out_cols[9]);               // src/RayTracer.cpp:33-55
void pto_woop_from_vertices(const float v0[3], const float v1[3], const float v2[3], float out12[12]);  // src/RTTriangle.hpp:14-20,45-47
int pto_intersect_woop(const float woop12[12], const float orig[3], const float dir[3], float* t, float* u, float* v);  // src/RTTriangle.hpp:50-59
int pto_box_intersect(const pto_flatnode* n, const float orig[3], const float inv_dir[3], const int sign[3], float t1);  // src/BvhNode.cpp:9-57
int pto_intersect_triangle1(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3],
                            float* t, float* u, float* v);               // src/rtIntersect.inl:215-259
float pto_gauss(float x, float y, int libm);                             // src/PathTraceRenderer.cpp:52-55
void pto_saturate(const float in[4], float amount, float out[4]);        // src/PathTraceRenderer.cpp:36-49
void pto_math_probe(int fn, const float* x, const float* y, float* out, int n, int libm);  // fn 0 sin 1 cos 2 exp 3 pow
uint32_t pto_rng_u32(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t draw);    // this build's Random stand-in
void pto_light_sample(const pto_light* l, int base, int bounce, uint32_t rng_key[4], float* pdf, float p[3]);  // src/AreaLight.cpp:69-85

// --- BVH build + flatten (reference src/Bvh.cpp:11-274) ---
// out_nodes must hold 2*num_tris entries; returns node count. stats: [totalSplits, badSplits, leaves, trueDepth, maxLeafTris, depthField]
int32_t pto_bvh_build(int32_t num_tris, const float* pos, int sah, int32_t* out_list, pto_flatnode* out_nodes, int32_t stats[6]);

// --- RayTracer::raycast (reference src/RayTracer.cpp:174-290), batched over n rays, nthreads host threads ---
void pto_raycast_batch(const pto_scene* s, int use_alpha, const float* orig, const float* dir, int64_t n, int occlusion,
                       int32_t* out_id, float* out_t, float* out_u, float* out_v, pto_counters* ctr, int nthreads);

// primary + shadow rays of BASELINE config C2: pixel-centre primary rays from invP, then one
// shadow ray per primary hit towards the Sobol light point (index = pixel index, dims 4,5, no RNG offset).
void pto_gen_primary(const pto_camera* cam, int width, int height, float* orig, float* dir);
void pto_gen_shadow(const pto_scene* s, const pto_light* light, int64_t n, const float* orig, const float* dir,
                    const int32_t* hit_id, const float* hit_t, const float* hit_u, const float* hit_v,
                    float* sh_orig, float* sh_dir, int32_t* sh_valid);

// --- PathTraceRenderer (reference src/PathTraceRenderer.cpp:236-871) ---
// One full pass over all 32x32 blocks (pathTraceBlock, :663-790), accumulating into accum[height][width][4].
// If sample_out != NULL (sobol_block mode only) also writes per-sample radiance [block][i][3] packed in block order.
void pto_render_pass(const pto_scene* s, const pto_params* p, const pto_camera* cam, const pto_light* lights, int num_lights,
                     float* accum, pto_counters* ctr, int nthreads, int libm, int block_begin, int block_end);
// updatePicture (:855-871)
void pto_resolve(const float* accum, int width, int height, float* out_rgba, int libm);
// single tracePath call (:236-551) for unit tests
void pto_trace_path(const pto_scene* s, const pto_params* p, const pto_camera* cam, const pto_light* lights, int num_lights,
                    float x, float y, int sampler_base, const uint32_t rng_key[4], int libm, float out_rgb[3], pto_counters* ctr);

#ifdef __cplusplus
}
#endif
