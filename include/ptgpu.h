/*
 * ptgpu.h -- C ABI of the MI355X (gfx950) implementation of Esgrove/pathtracer's
 * per-sample path-tracing hot path.
 *
 * The reference has no FFI/plugin layer: its boundary is the C++ class interface in
 * namespace FW.  This header is the device boundary inserted *under*
 * those classes: the host-side C++ mirror in pathtracer_amd/host/ (same class and
 * method names as the reference) calls only these functions.  Each entry point cites
 * the reference interface it replaces.
 *
 * Conventions: every function returns 0 on success or a negative PTGPU_E_* code and
 * never throws; ptgpu_last_error() gives a thread-local message.  No callbacks into
 * host code.  One context per device; a context is used from one host thread at a
 * time.  Inputs are copied at scene creation: the caller keeps ownership of what it
 * passes in.  Functions whose name ends in _dev take DEVICE pointers and only enqueue
 * work on the context's stream (ptgpu_set_stream); the others take HOST pointers,
 * copy, run and synchronise.
 *
 * All arithmetic is fp32 with no FMA contraction, in the reference's operation order,
 * so closest-hit / any-hit results (id, t, u, v) are bit-identical to the CPU code.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTGPU_OK 0
#define PTGPU_E_INVALID (-1)   /* bad argument / malformed scene (message says what) */
#define PTGPU_E_HIP (-2)       /* a HIP runtime call failed */
#define PTGPU_E_NODEVICE (-3)  /* no usable gfx950 device */
#define PTGPU_E_UNSUPPORTED (-4)
#define PTGPU_E_INTERNAL (-5)  /* a device-side safety limit ended a launch early (message says which); the call's output is incomplete */

typedef struct ptgpu_context ptgpu_context;
typedef struct ptgpu_scene ptgpu_scene;

/* FlatNode exactly as the reference lays it out (src/BvhNode.hpp:67-113, sizeof == 64):
 * depth-first pre-order array, left child = index + 1 (src/Bvh.cpp:70-96). Uploaded as-is. */
typedef struct ptgpu_flatnode {
    float bmin[3], bmax[3];
    int32_t startIdx, endIdx; /* inclusive range into the triangle index list */
    uint8_t leaf;
    uint8_t _padleaf[3];
    int32_t rightChild;
    int32_t axis;
    int32_t leftChild;
    int32_t triangles;
    int32_t pad0, pad1, pad2;
} ptgpu_flatnode;

/* FW MeshBase::Material fields the path reads (src/PathTraceRenderer.cpp:76-83,134,174-175,311;
 * src/RayTracer.cpp:206). tex_* index ptgpu_scene_desc.textures, -1 = texture absent. */
typedef struct ptgpu_material {
    float diffuse[4];
    float specular[3];
    float glossiness;
    int32_t tex_diffuse, tex_alpha, tex_normal, tex_specular;
} ptgpu_material;

/* 8-bit RGBA image; Image::getVec4f(x,y) == texel / 255 */
typedef struct ptgpu_texture {
    int32_t width, height;
    const uint8_t* rgba; /* HOST pointer, width*height*4 bytes, row-major */
} ptgpu_texture;

/* AreaLight state (src/AreaLight.hpp:74-79) */
typedef struct ptgpu_light {
    float xform[16]; /* column-major Mat4f m_xform */
    float size[2];   /* m_size (half extents) */
    float E[3];      /* m_E */
    float intensity; /* m_intensity */
    float power;     /* m_power, as updatePower() leaves it (src/AreaLight.hpp:54-57) */
} ptgpu_light;

/* What tracePath reads from CameraControls per sample (src/PathTraceRenderer.cpp:244-291), computed once on the host */
typedef struct ptgpu_camera {
    float invP[16];       /* (fitToView * cameraToClip * worldToCamera).inverted(), column-major */
    float forward[3];     /* getForward() */
    float orient_col0[3]; /* getOrientation().getCol(0) */
    float orient_col1[3]; /* getOrientation().getCol(1) */
    float far_;           /* getFar() */
} ptgpu_camera;

/* The static tunables of PathTraceRenderer (src/PathTraceRenderer.cpp:15-28) and the per-render values of
 * PathTracerContext (src/PathTraceRenderer.hpp:33-54), passed by value to the kernels. */
typedef struct ptgpu_params {
    int32_t width, height;
    int32_t bounces;        /* m_bounces: <0 = Russian roulette after |bounces| */
    int32_t num_aa_rays;    /* m_NumAaRays */
    int32_t sobol_block;    /* m_sobolBlock */
    int32_t bilinear;       /* m_bilinearFiltering */
    int32_t normal_mapped;  /* m_normalMapped */
    int32_t depth_of_field; /* m_depthOfField */
    int32_t whitted;        /* m_whitted */
    int32_t num_light_rays; /* m_numLightRays */
    int32_t use_alpha;      /* RayTracer::useAlpha (src/RayTracer.hpp:39) */
    uint32_t pass_seed;     /* stands in for the process-global seed counter at src/PathTraceRenderer.cpp:689-691 */
    float aperture, attenuation, filter_width, focal_distance, intensity, termination_prob;
} ptgpu_params;

/* Scene as the host classes hold it: the caller's triangle vector flattened to arrays (one entry per
 * RTTriangle, src/RTTriangle.hpp:23-33) and the BVH's public members (src/Bvh.hpp:36-43). */
typedef struct ptgpu_scene_desc {
    int32_t num_tris;
    const float* pos;      /* [num_tris][3][3]  m_vertices[k].p */
    const float* nrm;      /* [num_tris][3][3]  m_vertices[k].n */
    const float* uv;       /* [num_tris][3][2]  m_vertices[k].t */
    const int32_t* mat_id; /* [num_tris]        index of m_material in materials[] */
    const float* woop;     /* [num_tris][3][4]  row r = (M(r,0), M(r,1), M(r,2), N[r]) of tri_data (src/RTTriangle.hpp:8-21) */
    int32_t num_materials;
    const ptgpu_material* materials;
    int32_t num_textures;
    const ptgpu_texture* textures;
    int32_t num_nodes;
    const ptgpu_flatnode* nodes; /* BVH::nodes */
    int32_t list_size;
    const int32_t* list;         /* BVH::list */
} ptgpu_scene_desc;

/* Same definition as the reference's counter: 1 ray = 1 raycast() call (src/RayTracer.cpp:192). node_visits and
 * tri_tests count FlatNode::intersect and intersect_woop calls of the reference's traversal (src/RayTracer.cpp:197,203)
 * and are only collected while ptgpu_enable_counters(ctx, 1) is in force (they cost time). */
typedef struct ptgpu_counters {
    uint64_t rays, node_visits, tri_tests;
} ptgpu_counters;

/* ---- context -------------------------------------------------------------------------------------------------- */
int ptgpu_init(int device, ptgpu_context** out_ctx);
int ptgpu_shutdown(ptgpu_context* ctx);
const char* ptgpu_last_error(void);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the context's own stream. */
int ptgpu_set_stream(ptgpu_context* ctx, void* hip_stream);
int ptgpu_sync(ptgpu_context* ctx);
/* *out_done = 1 when everything enqueued on the context's stream has finished, else 0.  Replaces the completion poll of
 * PathTraceRenderer::checkFinish (m_launcher.getNumFinished() == getNumTasks(), src/PathTraceRenderer.cpp:878). */
int ptgpu_query(ptgpu_context* ctx, int* out_done);
/* device memory for callers that have no allocator of their own (the C++ host classes) */
int ptgpu_malloc(ptgpu_context* ctx, uint64_t bytes, void** out_dev);
int ptgpu_free(ptgpu_context* ctx, void* dev);
/* free / total device memory of the context's GPU (hipMemGetInfo): lets a caller of ptgpu_raycast_batch_dev_noalloc verify that it allocates nothing */
int ptgpu_mem_info(ptgpu_context* ctx, uint64_t* out_free_bytes, uint64_t* out_total_bytes);
int ptgpu_memcpy_h2d(ptgpu_context* ctx, void* dev, const void* host, uint64_t bytes);
int ptgpu_memcpy_d2h(ptgpu_context* ctx, void* host, const void* dev, uint64_t bytes);
int ptgpu_memset(ptgpu_context* ctx, void* dev, int value, uint64_t bytes);

/* replaces RayTracer::getRayCount / resetRayCounter (src/RayTracer.hpp:34-35) */
int ptgpu_enable_counters(ptgpu_context* ctx, int on);
int ptgpu_get_counters(ptgpu_context* ctx, ptgpu_counters* out);
int ptgpu_reset_counters(ptgpu_context* ctx);

/* ---- scene: replaces RayTracer::constructHierarchy's hand-over of triangles + BVH (src/RayTracer.cpp:167-172) and
 * App::constructTracer's triangle array (src/App.cpp:826-838).  Validates the tree (ranges, child indices, depth <= 64 =
 * the reference's traversal stack, src/RayTracer.cpp:190) and builds the device traversal layout. */
int ptgpu_scene_create(ptgpu_context* ctx, const ptgpu_scene_desc* desc, ptgpu_scene** out_scene);
int ptgpu_scene_destroy(ptgpu_scene* scene);
/* bytes of device memory held by the scene: [0] interior nodes, [1] triangle records, [2] shading attributes, [3] textures */
int ptgpu_scene_stats(const ptgpu_scene* scene, uint64_t out_bytes[4], int32_t* out_depth);
/* How long this scene's rays are, and what the path-trace passes derive from it.  No counterpart in the reference (its worker threads
 * have no such scheduling): before a scene's first pass the library traces 8,192 random rays inside the scene's box (fixed seed, counter
 * words of their own: ptgpu_get_counters does not see them) and picks from their mean node visits + triangle tests per ray the number
 * of lanes of a wave that must wait for shading before the wave stops traversing to serve them (32 for short rays, down to 12 for
 * long ones).  Scheduling only: results do not depend on it.  This call runs the probe if no pass has yet
 * (synchronises the context's stream then) and returns both numbers; either pointer may be NULL. */
int ptgpu_scene_ray_length(ptgpu_scene* scene, float* out_steps_per_ray, int32_t* out_service_threshold);

/* ---- RayTracer::raycast(orig, dir, occlusion) (src/RayTracer.cpp:174-290), n rays per call.
 * orig/dir: [n][3] floats (Vec3f arrays); the traced segment is [orig, orig+dir].  use_alpha = RayTracer::useAlpha.
 * out_id: index into the caller's triangle vector or -1 (RaycastResult::tri == nullptr); out_t = FLT_MAX, u = v = 0 on a
 * miss (src/RaycastResult.hpp:27).  Any out_* may be NULL.  occlusion != 0: any-hit early exit (src/RayTracer.cpp:235-238). */
int ptgpu_raycast_batch(ptgpu_scene* scene, int use_alpha, const float* orig, const float* dir, int64_t n, int occlusion,
                        int32_t* out_id, float* out_t, float* out_u, float* out_v);
int ptgpu_raycast_batch_dev(ptgpu_scene* scene, int use_alpha, const float* d_orig, const float* d_dir, int64_t n, int occlusion,
                            int32_t* d_id, float* d_t, float* d_u, float* d_v);

/* The one entry point that ALLOCATES NOTHING.  It has no counterpart in the reference (whose raycast() never allocates because it runs
 * on the CPU); it is for callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation.  Same arguments, kernels and results (bit for bit) as ptgpu_raycast_batch_dev, but the kernels' scratch memory -- counter
 * words, the deferred-ray list of deep trees, the stack overflow of a bounded persistent grid -- lives in a fixed 22 MiB device array
 * inside the library (one copy per GPU, held until the process ends) instead of the context's grow-only buffers, and no hipMalloc /
 * hipFree / allocating routine is called.  Limits: n <= PTGPU_NOALLOC_MAX_RAYS per call (PTGPU_E_UNSUPPORTED beyond; split the batch);
 * node-visit / triangle-test counting is not offered and the rays it traces do not show in ptgpu_get_counters.  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (two contexts or streams) race -- serialise them. */
#define PTGPU_NOALLOC_MAX_RAYS 4194304
int ptgpu_raycast_batch_dev_noalloc(ptgpu_scene* scene, int use_alpha, const float* d_orig, const float* d_dir, int64_t n, int occlusion,
                                    int32_t* d_id, float* d_t, float* d_u, float* d_v);

/* Ray generators of BASELINE config 2: pixel-centre primary rays unprojected through invP exactly as
 * src/PathTraceRenderer.cpp:244-266 with the pixel->clip mapping of :716-717, and one shadow ray per primary hit built as
 * :338-344 builds it (origin O + 1e-4 N with N the interpolated vertex normal, direction Pl - O un-normalised) towards the
 * Sobol point of the light with index = ray index, dims 4 and 5 (src/AreaLight.cpp:75-81 without the RNG offset). */
int ptgpu_gen_primary_dev(ptgpu_context* ctx, const ptgpu_camera* cam, int32_t width, int32_t height, float* d_orig, float* d_dir);
int ptgpu_gen_shadow_dev(ptgpu_scene* scene, const ptgpu_light* light, int64_t n, const float* d_orig, const float* d_dir,
                         const int32_t* d_id, const float* d_t, const float* d_u, const float* d_v,
                         float* d_sh_orig, float* d_sh_dir, int32_t* d_sh_valid);

/* The BASELINE config 2 frame in ONE launch: for every pixel the primary ray of ptgpu_gen_primary_dev traced closest-hit
 * (RayTracer::raycast(orig, dir, false), src/RayTracer.cpp:174-290) and, where it hits, the shadow ray of ptgpu_gen_shadow_dev traced
 * any-hit (occlusion = true) -- the rays stay in registers instead of travelling through HBM between four launches.  DEVICE arrays
 * [height*width] indexed by pixel: the primary hit record (d_id, d_t, d_u, d_v as ptgpu_raycast_batch_dev writes them; required) and
 * the occluder the shadow ray found (d_shadow_id: -1 = none or no primary hit; NULL = primary rays only); bit-identical to the
 * four-call sequence.  Alpha cutout is not applied.  Any tree depth the scene accepts (<= 64): stack entries beyond the 16 kept in LDS go
 * to a grow-only device buffer of the context ((tree depth - 17) x 512 bytes per 8x8 pixel tile, allocated by the first such call; beyond
 * 4 GiB of it a kernel with all 24 entries in LDS runs instead, or, on trees deeper than 25 levels, PTGPU_E_UNSUPPORTED is returned). */
int ptgpu_trace_frame_dev(ptgpu_scene* scene, const ptgpu_camera* cam, const ptgpu_light* light, int32_t width, int32_t height, int32_t* d_id,
                          float* d_t, float* d_u, float* d_v, int32_t* d_shadow_id);

/* ---- PathTraceRenderer: replaces the worker-thread launch in startPathTracingProcess / checkFinish
 * (m_launcher.push(pathTraceBlock, ...), src/PathTraceRenderer.cpp:847-852, :906-910) and pathTraceBlock + tracePath
 * themselves (:663-790, :236-551).  One call = one pass over the 32x32 blocks (row-major block index as :816-841 numbers
 * them) b = block_begin, block_begin + block_stride, ... < block_end (block_end < 0: all blocks), every block with all its
 * samples (block area x num_aa_rays), accumulated into accum[height][width][4] exactly as :727-732 (Sobol-block mode) or
 * :743-787 (Gaussian mode) do: += on the existing contents, samples of a pixel added in the reference's order.  The block
 * subset is how image tiles are sharded over GPUs: rank r of N passes block_begin = r, block_stride = N.
 * Per-sample RNG streams are keyed by (block, pass_seed, i, k) (documented deviation from the reference's one sequential
 * Random per block, which cannot be parallelised); params->pass_seed plays the role of the reference's seed counter.
 * lights: HOST array.  _dev: d_accum is a DEVICE pointer and the call only enqueues work. */
int ptgpu_render_pass_dev(ptgpu_scene* scene, const ptgpu_params* params, const ptgpu_camera* cam, const ptgpu_light* lights,
                          int32_t num_lights, int32_t block_begin, int32_t block_end, int32_t block_stride, float* d_accum);
int ptgpu_render_pass(ptgpu_scene* scene, const ptgpu_params* params, const ptgpu_camera* cam, const ptgpu_light* lights,
                      int32_t num_lights, int32_t block_begin, int32_t block_end, int32_t block_stride, float* accum);
/* PathTraceRenderer::stop() while a pass is running: the reference's workers poll m_bForceExit before every sample and return
 * (src/PathTraceRenderer.cpp:709,736), and stop() waits only for that (:918-930).  ptgpu_cancel cancels every render pass enqueued on the
 * context so far: the integrator's persistent waves read a cancel word in device memory when they draw a new chunk of samples and draw no more, the samples
 * in flight run to their end, and a batch of blocks whose samples are not all there is NOT accumulated -- so d_accum is left holding whole
 * batches only (finite, a valid partial image) and the stream drains within milliseconds; ptgpu_sync / ptgpu_query then report
 * completion as usual.  Passes enqueued after the call are not affected.  The call itself only enqueues a one-wave kernel on a
 * high-priority stream of the context and does not wait; it may be made from any thread, in particular while another thread is
 * blocked in ptgpu_sync or in the host-pointer ptgpu_render_pass. */
int ptgpu_cancel(ptgpu_context* ctx);

/* ---- the reference's static per-sample functions (src/PathTraceRenderer.hpp:97-102), evaluated by the same device code the render pass runs.
 * ptgpu_trace_samples[_dev]: Vec3f tracePath(x, y, ctx, samplerBase, R, ...) (src/PathTraceRenderer.cpp:236-551), or traceWhitted
 * (:553-660) when params->whitted, for n arbitrary samples: xy [n][2] clip-space position, sampler_base [n] Sobol index base,
 * rng_state [n][2] = (key, counter) of the sample's random stream (this build's counter-based stand-in for FW::Random: key =
 * the hashed seed words, counter = draws consumed so far).  out [n][4] floats: radiance rgb, and in the 4th word the BITS of the
 * stream's counter after the call (the reference advances the caller's Random by reference).  params->width/height are not used
 * (the camera's invP carries the image size).  lights: HOST array.  _dev: device arrays, work only enqueued. */
int ptgpu_trace_samples_dev(ptgpu_scene* scene, const ptgpu_params* params, const ptgpu_camera* cam, const ptgpu_light* lights,
                            int32_t num_lights, int64_t n, const float* d_xy, const int32_t* d_sampler_base, const uint32_t* d_rng_state,
                            float* d_out);
int ptgpu_trace_samples(ptgpu_scene* scene, const ptgpu_params* params, const ptgpu_camera* cam, const ptgpu_light* lights, int32_t num_lights,
                        int64_t n, const float* xy, const int32_t* sampler_base, const uint32_t* rng_state, float* out);
/* void getTextureParameters(hit, diffuse, n, specular) (src/PathTraceRenderer.cpp:75-181) for n hits given as (triangle index, u, v);
 * uses params->bilinear and params->normal_mapped.  HOST arrays; out [n][9] = diffuse rgb, shading normal, specular rgb. */
int ptgpu_texture_parameters_batch(ptgpu_scene* scene, const ptgpu_params* params, int64_t n, const int32_t* tri, const float* u,
                                   const float* v, float* out_diffuse_normal_specular);
/* int getLightToSample(ctx, pos, R) (src/PathTraceRenderer.cpp:183-210) for n positions [n][3], each with its own stream state
 * rng_state [n][2] (key, counter).  HOST arrays; out [n][2] = chosen light, counter after the call (one draw when num_lights > 1). */
int ptgpu_light_to_sample_batch(ptgpu_context* ctx, const ptgpu_light* lights, int32_t num_lights, int64_t n, const float* pos,
                                const uint32_t* rng_state, int32_t* out_index_and_counter);

/* float sobol::sample(index, dimension) (src/sobol.h:39-49; scramble 0, as every call site passes) for n (index, dimension < 1024) pairs,
 * evaluated by the device functions the kernels draw their Sobol points with (byte-indexed pair table for the low 24 index bits, the
 * bit loop above them).  HOST arrays.  Exists so that the device's Sobol values can be compared word for word with the reference's own
 * sobol.cpp. */
int ptgpu_sobol_sample_batch(ptgpu_context* ctx, int64_t n, const uint64_t* index, const uint32_t* dimension, float* out);

/* PathTraceRenderer::updatePicture (src/PathTraceRenderer.cpp:855-871): out = (pow(rgb / w, 1/2.2), w / w) per pixel, w == 0
 * pixels pass through undivided.  Both pointers are DEVICE pointers, [height][width][4] floats. */
int ptgpu_resolve_dev(ptgpu_context* ctx, const float* d_accum, int32_t width, int32_t height, float* d_out_rgba);

/* Out-of-block samples of the Sobol-block passes on this context (a sample whose position bx + bw * sobol() rounds up to bx + bw lands
 * one pixel outside its block, src/PathTraceRenderer.cpp:714-728: some 30,000 per 1024-spp pass of a 4K frame): out[0] = how many the
 * LAST pass had, out[1] = how many were DROPPED because a list was full SINCE THE LAST CALL of this function (0 in any sane render: the
 * pass's own list is sized from its sample count, the list waiting for a tile exchange holds 65,536; the count is cleared once
 * reported; ptgpu_render_pass reports it as an error, callers of the _dev variant ask here), out[2] = entries waiting for the next
 * tile exchange (with tiles dealt over ranks every out-of-block sample travels, ptgpu_gather_tiles_dev / ptgpu_tiles_pack_dev).
 * Synchronises the context's stream.  Returns PTGPU_E_INTERNAL if the last pass ran the
 * experimental pooled integrator (environment PTGPU_PT_POOL=1) and its watchdog ended a launch early. */
int ptgpu_spill_status(ptgpu_context* ctx, uint64_t out[3]);

/* ---- image tiles across the GPUs of one node (one process and one context per GPU).  No counterpart in the reference, whose worker
 * threads write disjoint 32x32 blocks of ONE shared image (src/PathTraceRenderer.cpp:727-732, :785-787, blocks built at :816-841); here
 * rank r of N renders blocks r, r+N, ... (ptgpu_render_pass_dev with block_begin = r, block_stride = N) into its own full-size
 * accumulation image and, after a pass, ptgpu_gather_tiles_dev fills in every other rank's blocks with ONE all-gather over RCCL (xGMI).
 * Nothing is summed: the N-GPU image equals the 1-GPU image bit for bit, including samples that land one pixel outside their block in
 * another rank's tile (they travel in a small fixed section of each rank's slice and are added, in rank order, on every rank).
 *
 * ptgpu_comm_create_id: rank 0 makes a PTGPU_COMM_ID_BYTES id (an ncclUniqueId) and hands it to the other ranks by any means (file,
 * environment, MPI, a torch.distributed broadcast).  ptgpu_comm_init: collective over all ranks (ncclCommInitRank on the context's
 * device); the communicator belongs to the context and is destroyed with it or by ptgpu_comm_destroy.  RCCL is loaded on first use.
 * ptgpu_gather_tiles_dev: enqueued on the context's stream after the pass; all ranks end with the full image in d_accum. */
#define PTGPU_COMM_ID_BYTES 128
int ptgpu_comm_create_id(void* out_id);
int ptgpu_comm_init(ptgpu_context* ctx, int32_t world, int32_t rank, const void* id);
int ptgpu_comm_destroy(ptgpu_context* ctx);
int ptgpu_gather_tiles_dev(ptgpu_context* ctx, int32_t width, int32_t height, float* d_accum);
/* The two halves of the gather for callers that bring their own transport (MPI, torch.distributed, a host relay): pack this rank's
 * blocks and pending foreign samples into one slice of ptgpu_tiles_slice_floats() floats; exchange so that d_slices holds the world
 * slices in rank order; unpack writes the other ranks' blocks into d_accum and adds every rank's foreign samples. */
int ptgpu_tiles_slice_floats(int32_t width, int32_t height, int32_t world, int64_t* out_floats);
int ptgpu_tiles_pack_dev(ptgpu_context* ctx, int32_t width, int32_t height, int32_t world, int32_t rank, const float* d_accum, float* d_slice);
int ptgpu_tiles_unpack_dev(ptgpu_context* ctx, int32_t width, int32_t height, int32_t world, int32_t rank, const float* d_slices, float* d_accum);

/* Optional device-side BVH builder (no counterpart in the reference, whose builders are the CPU SAH sweep and
 * median splits of src/Bvh.cpp:98-234, constructed by BVH::BVH(triangles, sahEnabled), src/Bvh.cpp:11-42): a linear BVH over 63-bit
 * Morton codes, emitted in the reference's FlatNode layout (src/BvhNode.hpp:67-113; pre-order, left child = index + 1, src/Bvh.cpp:70-96)
 * with subtrees of <= max_tris_in_leaf triangles collapsed to leaves (the reference's maxTrisInLeaf is 3), so ptgpu_scene_create and
 * every kernel take it like BVH::nodes / BVH::list.  It is NOT the reference's tree: ties between triangles at exactly equal t and the
 * occluder an any-hit ray reports follow this tree's order, and a ray that grazes a box face within rounding can be culled by one
 * tree and not by another (the reference's slab test has no padding; measured: 1 of 8.3 M primary rays on a 260 k-triangle scene
 * with axis-aligned walls, 0 on the other test scenes).  All other closest-hit records are the same floats.
 * pos: HOST [num_tris][3][3].  out_nodes: HOST, room for 2*num_tris nodes.  out_list: HOST [num_tris].  out_depth: tree height
 * (ptgpu_scene_create refuses trees deeper than 64; Morton ties are split by position, which can get deep for many coincident
 * centroids).  Synchronous. */
int ptgpu_bvh_build_lbvh(ptgpu_context* ctx, const float* pos, int32_t num_tris, int32_t max_tris_in_leaf, ptgpu_flatnode* out_nodes,
                         int32_t* out_num_nodes, int32_t* out_list, int32_t* out_depth);

/* ptgpu_bvh_build_lbvh and ptgpu_scene_create in one step without the host round trip: the triangle arrays of `desc` are uploaded, the
 * hierarchy is built on the device from them (desc->nodes / list must be NULL, num_nodes / list_size 0) and the traversal records are
 * written by device kernels straight from the device-built tree -- for scenes that change per frame, where the reference would call
 * RayTracer::constructHierarchy again (src/RayTracer.cpp:167-172).  The scene is identical, record for record, to the one
 * ptgpu_scene_create makes from ptgpu_bvh_build_lbvh's output (10 M triangles: 0.12 s instead of 0.42 s + 0.66 s).  out_num_nodes
 * (may be NULL) receives the node count of the built tree.  The tree itself is not returned: callers that want to cache it
 * (RayTracer::saveHierarchy) or to run the CPU oracle on it use ptgpu_bvh_build_lbvh. */
int ptgpu_scene_create_lbvh(ptgpu_context* ctx, const ptgpu_scene_desc* desc, int32_t max_tris_in_leaf, ptgpu_scene** out_scene, int32_t* out_num_nodes);

/* rtIntersect.inl intersect_triangle1 (src/rtIntersect.inl:215-259; two-sided Moller-Trumbore, EPSILON 1e-9) for n independent
 * (ray, triangle) pairs, HOST pointers: orig/dir/v0/v1/v2 [n][3].  out_hit = 1/0; t,u,v are written as the reference leaves them on a
 * hit and as 0 on a miss.  The reference includes this file but never calls it (its live test is intersect_woop); kept as a separate
 * entry point, not as a traversal variant, so that traversal parity is defined by what the reference executes. */
int ptgpu_intersect_triangle1_batch(ptgpu_context* ctx, const float* orig, const float* dir, const float* v0, const float* v1,
                                    const float* v2, int64_t n, int32_t* out_hit, float* out_t, float* out_u, float* out_v);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
