// The one-ray-per-lane raycast kernels live in a translation unit of their own (pt_raycast_step.hip), compiled with LLVM's max-ILP
// scheduling strategy: these short kernels run 1.5 % faster under it (closest 0.1350 vs 0.1368 ms, any-hit 0.1722 vs 0.1752 ms on
// the configs[1] frame), the path tracer's long kernel 1.7 % slower, so the flag cannot be given to the whole library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptgpu.h"
#include "pt_traverse.hip.h"

namespace ptd {

// One-ray-per-lane kernel: ONE wave per workgroup.  A workgroup keeps its LDS and wave slots until its slowest wave ends, so
// with 4 waves the three faster ones idle; measured on the configs[1] frame 64 vs 256 threads: closest 0.153 vs 0.160 ms,
// any-hit 0.188 vs 0.236 ms, incoherent 0.296 vs 0.375 ms.
constexpr int RC_STEP_BLOCK = 64;

// What a lane does when its stack outgrows the nlds LDS entries:
//   RC_BEYOND_NEVER     the tree is shallow enough that it cannot happen
//   RC_BEYOND_DEFER     the ray is appended to defer_list (count in counters[5]) instead of being traced (k_raycast_persist's overflow variant takes it)
//   RC_BEYOND_OVERFLOW  deeper entries live in stack_ovf: ovf_entries x 64 entries of 8 bytes per workgroup of the grid (ceil(n / 64) of them)
constexpr int RC_BEYOND_NEVER = 0, RC_BEYOND_DEFER = 1, RC_BEYOND_OVERFLOW = 2;
// Launches k_raycast_step<anyhit, no alpha, no counting, nlds, ...> over n rays on `stream`; nlds is 8, 12, 16 or (anything else) 24.
__attribute__((visibility("hidden"))) void launch_raycast_step(bool anyhit, int nlds, int beyond, hipStream_t stream, const TravScene& trav,
                                                               const float* o, const float* d, long long n, int32_t* id, float* t, float* u,
                                                               float* v, unsigned long long* counters, int32_t* defer_list, uint2* stack_ovf,
                                                               int ovf_entries);

// The configs[1] frame in one launch (k_trace_frame): with_regions = 16 LDS stack entries + ovf_entries x 64 entries of stack_ovf per
// workgroup (8x8 tiles); otherwise all 24 entries in LDS (trees up to depth 25) and tiles tile_w pixels wide.  grid = one workgroup per tile.
__attribute__((visibility("hidden"))) void launch_trace_frame(hipStream_t stream, dim3 grid, bool with_regions, int tile_w, const TravScene& trav,
                                                              const float* nrm, const uint2* sobol_pairs, const uint32_t* sobol_matrices,
                                                              const ptgpu_camera& cam, const ptgpu_light& light, int width, int height, int tiles_x,
                                                              int32_t* id, float* t, float* u, float* v, int32_t* shadow_id,
                                                              unsigned long long* counters, uint2* stack_ovf, int ovf_entries);

}  // namespace ptd
