// Ray generators of BASELINE configs[1], shared by the four-launch kernels (ptgpu.hip: k_gen_primary, k_gen_shadow) and the one-launch
// frame kernel (pt_raycast_step.hip: k_trace_frame) so that both forms compute the same rays bit for bit.
#pragma once
#include "../../include/ptgpu.h"
#include "pt_device.hip.h"

// camera ray of a pixel centre, no depth of field: tracePath's set-up, reference src/PathTraceRenderer.cpp:244-292
__device__ __forceinline__ void camera_ray_nodof(const ptgpu_camera& cam, float x, float y, ptd::F3& Ro, ptd::F3& Rd) {
    ptd::F4 a = ptd::mul44(cam.invP, x, y, 0.0f, 1.0f);
    float ia = 1.0f / a.w;
    Ro = ptd::f3(a.x * ia, a.y * ia, a.z * ia);
    ptd::F4 b = ptd::mul44(cam.invP, x, y, 1.0f, 1.0f);
    float ib = 1.0f / b.w;
    Rd = ptd::f3(b.x * ib, b.y * ib, b.z * ib) - Ro;
}


// shadow ray of BASELINE configs[1] for the hit (tri, t, u, v) of ray (ro, rd) number i: reference src/PathTraceRenderer.cpp:338-344 towards
// the Sobol light point with index i, dimensions 4 and 5
// (the two Sobol values come from the byte-indexed pair table: three independent loads instead of up to 2 x 31 dependent trips of
// sobol::sample's bit loop -- the same words, XOR being associative; i < 2^31)
__device__ __forceinline__ void shadow_ray(const float* __restrict__ nrm, const uint2* __restrict__ sobol_pairs, const uint32_t* __restrict__ sobol_matrices,
                                           const ptgpu_light& L, unsigned long long i,
                                           ptd::F3 ro, ptd::F3 rd, int tri, float t, float uu, float vv, ptd::F3& o, ptd::F3& d) {
    ptd::F3 O = ro + t * rd;
    const float* nn = nrm + (size_t)tri * 9;
    ptd::F3 N = ptd::normalize((1 - uu - vv) * ptd::ld3(nn) + uu * ptd::ld3(nn + 3) + vv * ptd::ld3(nn + 6));
    float sa, sb;
    ptd::sobol32x2m(sobol_pairs, sobol_matrices, (uint32_t)i, 4u, sa, sb);
    float x = (2.0f * sa - 1.0f) * L.size[0];
    float y = (2.0f * sb - 1.0f) * L.size[1];
    ptd::F3 Pl = ptd::f3(L.xform[12], L.xform[13], L.xform[14]) + ptd::f3(L.xform[0], L.xform[1], L.xform[2]) * x +
                 ptd::f3(L.xform[4], L.xform[5], L.xform[6]) * y;
    o = O + 0.0001f * N;
    d = Pl - O;
}

