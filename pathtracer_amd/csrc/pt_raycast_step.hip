// One-ray-per-lane raycast kernels (reference src/RayTracer.cpp:174-290 RayTracer::raycast, one call per lane), the one-launch frame
// kernel built from the same traverser, and their launchers.
// Built with -mllvm -amdgpu-sched-strategy=max-ilp (see pt_raycast_step.h and __graft_entry__.build()).
#include "pt_raycast_step.h"

#include "pt_context.h"
#include "pt_frame_rays.hip.h"

constexpr int RC_STEP_BLOCK = ptd::RC_STEP_BLOCK;

// One ray per lane for the lane's whole life (no refill), step-wise traverser.  Never built with global-memory stack overflow:
// a grid of n/64 workgroups has no bounded set of waves to give overflow regions to.
// DEFER: the tree may need more than NLDS entries; a ray that does is given up here (nothing written out of bounds) and its index
// appended to defer_list (count in counters[5]) for k_raycast_persist's overflow variant to trace.
// OVF: the entries beyond NLDS go to a global-memory region of the workgroup's own instead (ovf_entries x 64 entries per workgroup,
// ovf_entries = tree depth - 1 - NLDS: all a lane can ever need), when the launcher could afford one region per workgroup of the grid.
template <bool ANYHIT, bool ALPHA, bool COUNT, int NLDS, bool DEFER, bool OVF = false>
__global__ __launch_bounds__(RC_STEP_BLOCK) void k_raycast_step(ptd::TravScene S, const float* __restrict__ orig, const float* __restrict__ dir,
                                                           long long n, int32_t* __restrict__ out_id, float* __restrict__ out_t,
                                                           float* __restrict__ out_u, float* __restrict__ out_v, unsigned long long* counters,
                                                           int32_t* __restrict__ defer_list, uint2* stack_ovf = nullptr, int ovf_entries = 0) {
    static_assert(!(DEFER && OVF), "one way of outgrowing the LDS stack per kernel");
    static_assert(RC_STEP_BLOCK == 64, "one wave per workgroup: the overflow region is indexed by workgroup");
    __shared__ uint2 lds_stack[RC_STEP_BLOCK / 64][NLDS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long i = (long long)blockIdx.x * RC_STEP_BLOCK + threadIdx.x;
    if (i == 0) atomicAdd(&counters[0], (unsigned long long)n);  // 1 ray = 1 raycast() call (reference src/RayTracer.cpp:192)
    ptd::Trav<ALPHA, COUNT, NLDS, OVF, DEFER> tr;
    tr.reset_counts();
    tr.attach(&lds_stack[wave][0][lane], OVF ? stack_ovf + (size_t)blockIdx.x * (size_t)ovf_entries * 64 + lane : nullptr);
    tr.alive = false;
    if (i < n) {
        tr.begin(S, ptd::ld3(orig + i * 3), ptd::ld3(dir + i * 3), ANYHIT);
        while (tr.alive) tr.step(S);
        if (tr.deferred()) {
            defer_list[atomicAdd(&counters[5], 1ull)] = (int32_t)i;  // at most n entries: the list holds n
        } else {
            ptd::HitRec h = tr.result(S);
            if (out_id) out_id[i] = h.id;
            if (out_t) out_t[i] = h.t;
            if (out_u) out_u[i] = h.u;
            if (out_v) out_v[i] = h.v;
        }
    }
    if (COUNT) {
        atomicAdd(&counters[1], (unsigned long long)tr.n_nodes);
        atomicAdd(&counters[2], (unsigned long long)tr.n_tris);
    }
}

namespace {
template <bool ANYHIT, int NLDS>
void launch(int beyond, dim3 grid, hipStream_t stream, const ptd::TravScene& trav, const float* o, const float* d, long long n, int32_t* id,
            float* t, float* u, float* v, unsigned long long* counters, int32_t* defer_list, uint2* stack_ovf, int ovf_entries) {
    if (beyond == ptd::RC_BEYOND_DEFER)
        hipLaunchKernelGGL((k_raycast_step<ANYHIT, false, false, NLDS, true>), grid, dim3(RC_STEP_BLOCK), 0, stream, trav, o, d, n, id, t, u, v, counters, defer_list, nullptr, 0);
    else if (beyond == ptd::RC_BEYOND_OVERFLOW)
        hipLaunchKernelGGL((k_raycast_step<ANYHIT, false, false, NLDS, false, true>), grid, dim3(RC_STEP_BLOCK), 0, stream, trav, o, d, n, id, t, u, v, counters, defer_list, stack_ovf, ovf_entries);
    else
        hipLaunchKernelGGL((k_raycast_step<ANYHIT, false, false, NLDS, false>), grid, dim3(RC_STEP_BLOCK), 0, stream, trav, o, d, n, id, t, u, v, counters, defer_list, nullptr, 0);
}
template <bool ANYHIT>
void launch_nlds(int nlds, int beyond, dim3 grid, hipStream_t stream, const ptd::TravScene& trav, const float* o, const float* d, long long n,
                 int32_t* id, float* t, float* u, float* v, unsigned long long* counters, int32_t* defer_list, uint2* stack_ovf, int ovf_entries) {
    if (nlds == 8) launch<ANYHIT, 8>(beyond, grid, stream, trav, o, d, n, id, t, u, v, counters, defer_list, stack_ovf, ovf_entries);
    else if (nlds == 12) launch<ANYHIT, 12>(beyond, grid, stream, trav, o, d, n, id, t, u, v, counters, defer_list, stack_ovf, ovf_entries);
    else if (nlds == 16) launch<ANYHIT, 16>(beyond, grid, stream, trav, o, d, n, id, t, u, v, counters, defer_list, stack_ovf, ovf_entries);
    else launch<ANYHIT, 24>(beyond, grid, stream, trav, o, d, n, id, t, u, v, counters, defer_list, stack_ovf, ovf_entries);
}
}  // namespace

void ptd::launch_raycast_step(bool anyhit, int nlds, int beyond, hipStream_t stream, const TravScene& trav, const float* o, const float* d,
                              long long n, int32_t* id, float* t, float* u, float* v, unsigned long long* counters, int32_t* defer_list,
                              uint2* stack_ovf, int ovf_entries) {
    const dim3 grid((unsigned)((n + RC_STEP_BLOCK - 1) / RC_STEP_BLOCK));
    if (anyhit) launch_nlds<true>(nlds, beyond, grid, stream, trav, o, d, n, id, t, u, v, counters, defer_list, stack_ovf, ovf_entries);
    else launch_nlds<false>(nlds, beyond, grid, stream, trav, o, d, n, id, t, u, v, counters, defer_list, stack_ovf, ovf_entries);
}

// ------------------------------------------------------------------------------------------------
// BASELINE configs[1] in ONE launch: per pixel the primary ray of k_gen_primary traced closest-hit and, where it hits, the shadow ray of
// k_gen_shadow traced any-hit -- the rays live in registers, never in HBM (the four-launch form writes and reads 2 x 24 B of ray per
// pixel and phase and pays three kernel boundaries).  Same device functions, same arithmetic: every output equals the
// gen_primary -> raycast -> gen_shadow -> raycast sequence bit for bit.  Measured on the 1080p head frame: 0.262 ms against 0.332 ms
// for the four launches (and 0.321 ms for two launches, one per phase, that re-read the hit record: most of the gain is a wave going
// straight from its primary rays to its shadow rays).  Per-wave ray counts go to spread counter words (RAY_SPREAD).
// ------------------------------------------------------------------------------------------------
// OVF: the tree may need more than NLDS stack entries; each workgroup (= wave) then owns ovf_entries x 64 entries of stack_ovf, where
// ovf_entries = tree depth - 1 - NLDS is all a lane can ever push beyond the LDS part.
template <int NLDS, int TILE_W, bool OVF = false>
__global__ __launch_bounds__(64) void k_trace_frame(ptd::TravScene S, const float* __restrict__ nrm, const uint2* __restrict__ sobol_pairs,
                                                          const uint32_t* __restrict__ sobol_matrices, ptgpu_camera cam,
                                                          ptgpu_light L, int width, int height, int tiles_x, int32_t* __restrict__ out_id,
                                                          float* __restrict__ out_t, float* __restrict__ out_u, float* __restrict__ out_v,
                                                          int32_t* __restrict__ out_shadow_id, unsigned long long* counters,
                                                          uint2* stack_ovf = nullptr, int ovf_entries = 0) {
    __shared__ uint2 lds_stack[NLDS][64];
    constexpr int TILE_H = 64 / TILE_W;
    const int lane = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int px = tx * TILE_W + (lane % TILE_W), py = ty * TILE_H + (lane / TILE_W);
    const bool have = px < width && py < height;
    const long long i = (long long)py * width + px;
    ptd::Trav<false, false, NLDS, OVF, false> tr;
    tr.reset_counts();
    tr.attach(&lds_stack[0][lane], OVF ? stack_ovf + (size_t)blockIdx.x * (size_t)ovf_entries * 64 + lane : nullptr);
    tr.alive = false;
    ptd::F3 Ro = ptd::f3(0, 0, 0), Rd = Ro;
    if (have) {
        const float div_x = 1.0f / (float)width, div_y = 1.0f / (float)height;
        const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
        camera_ray_nodof(cam, fx * div_x * 2.0f - 1.0f, fy * div_y * -2.0f + 1.0f, Ro, Rd);
        tr.begin(S, Ro, Rd, false);
    }
    while (tr.alive) tr.step(S);
    ptd::HitRec h;
    h.id = -1; h.t = 3.402823466e+38f; h.u = 0.0f; h.v = 0.0f;
    if (have) {
        h = tr.result(S);
        out_id[i] = h.id; out_t[i] = h.t; out_u[i] = h.u; out_v[i] = h.v;
    }
    const bool shadow = have && h.id >= 0 && out_shadow_id;
    if (shadow) {
        ptd::F3 so, sd;
        shadow_ray(nrm, sobol_pairs, sobol_matrices, L, (unsigned long long)i, Ro, Rd, h.id, h.t, h.u, h.v, so, sd);
        tr.begin(S, so, sd, true);
    }
    while (tr.alive) tr.step(S);
    if (have && out_shadow_id) out_shadow_id[i] = shadow ? tr.result(S).id : -1;
    const unsigned long long rays = (unsigned long long)__popcll(__ballot(have)) + (unsigned long long)__popcll(__ballot(shadow));
    if (lane == 0 && rays) atomicAdd(&counters[CTR_WORDS + (blockIdx.x & (RAY_SPREAD - 1))], rays);
}


void ptd::launch_trace_frame(hipStream_t stream, dim3 grid, bool with_regions, int tile_w, const TravScene& trav, const float* nrm, const uint2* sobol_pairs,
                             const uint32_t* sobol_matrices, const ptgpu_camera& cam, const ptgpu_light& light, int width, int height, int tiles_x,
                             int32_t* id, float* t, float* u, float* v, int32_t* shadow_id, unsigned long long* counters, uint2* stack_ovf,
                             int ovf_entries) {
    if (with_regions) {  // 16 LDS entries + the workgroup's overflow region, 8x8 tiles
        hipLaunchKernelGGL((k_trace_frame<16, 8, true>), grid, dim3(64), 0, stream, trav, nrm, sobol_pairs, sobol_matrices, cam, light, width, height, tiles_x,
                           id, t, u, v, shadow_id, counters, stack_ovf, ovf_entries);
        return;
    }
#define PT_FRAME(TW) hipLaunchKernelGGL((k_trace_frame<24, TW>), grid, dim3(64), 0, stream, trav, nrm, sobol_pairs, sobol_matrices, cam, light, width, height, \
                                        tiles_x, id, t, u, v, shadow_id, counters, nullptr, 0)
    if (tile_w == 4) PT_FRAME(4); else if (tile_w == 16) PT_FRAME(16); else if (tile_w == 32) PT_FRAME(32); else if (tile_w == 64) PT_FRAME(64); else PT_FRAME(8);
#undef PT_FRAME
}
