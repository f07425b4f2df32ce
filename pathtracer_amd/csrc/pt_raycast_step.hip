// One-ray-per-lane raycast kernels (reference src/RayTracer.cpp:174-290 RayTracer::raycast, one call per lane) and their launcher.
// Built with -mllvm -amdgpu-sched-strategy=max-ilp (see pt_raycast_step.h and __graft_entry__.build()).
#include "pt_raycast_step.h"

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
