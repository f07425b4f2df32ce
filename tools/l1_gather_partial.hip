// Companion of l1_gather_peak.hip: what does a record-gather step cost when only k of a wave's 64 lanes are active (each on a record of
// its own, 64 bytes as 4 x dwordx4, next index taken from the data)?  Decides whether packing a path tracer's traversing lanes more
// densely into waves would relieve the vector memory path: if the cost were per wave-instruction, fewer, fuller steps would; if it is
// per active lane (per distinct line), they would not.  Prints one JSON line per k: wave-steps/s and lane-steps/s chip-wide.
//   hipcc -O3 --offload-arch=gfx950 tools/l1_gather_partial.hip -o build/l1_gather_partial && build/l1_gather_partial
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64, 4) void k_gather(const uint4* __restrict__ rec, uint32_t mask, int iters, uint32_t* __restrict__ out, int k, int spread)
{
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64u + lane;
    // spread = 0: lanes 0..k-1 active; spread = 1: k lanes spread evenly over the wave (every quad partly active)
    const bool active = spread ? ((lane * (uint32_t)k) / 64u != ((lane + 1u) * (uint32_t)k) / 64u) : (lane < (uint32_t)k);
    uint32_t state = gid * 2654435761u, idx = state >> 7 & mask, acc = 0;
    if (active)
        for (int i = 0; i < iters; i++) {
            const uint4* p = rec + (size_t)idx * 4;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            const uint32_t s = (a.x ^ a.y ^ a.z ^ a.w) + 3u * (b.x ^ b.y ^ b.z ^ b.w) + 5u * (c.x ^ c.y ^ c.z ^ c.w) + 7u * (d.x ^ d.y ^ d.z ^ d.w);
            acc += s;
            state += 0x9E3779B9u;
            idx = (s ^ state) >> 3 & mask;
        }
    out[gid] = acc;
}

int main()
{
    const uint32_t records = 32768u;  // 2 MiB: head.obj's tree is 1.6 MB
    std::vector<uint32_t> h((size_t)records * 16);
    uint32_t x = 12345u;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = x; }
    uint4* d_rec; uint32_t* d_out;
    const int iters = 3000;
    CK(hipMalloc(&d_rec, h.size() * 4));
    CK(hipMalloc(&d_out, (size_t)256 * 32 * 4 * 64 * 4));
    CK(hipMemcpy(d_rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // resident waves per CU: one round of 256 x W single-wave workgroups (all resident at once), except the first line set: 16 per CU, 4 rounds
    for (int wpc : {0, 4, 8, 16, 32})
    for (int spread = 0; spread < (wpc ? 1 : 2); ++spread)
        for (int k : {8, 16, 24, 32, 40, 48, 56, 64}) {
            const int groups = wpc ? 256 * wpc : 256 * 16 * 4;
            k_gather<<<groups, 64>>>(d_rec, records - 1, iters / 10, d_out, k, spread);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0));
                k_gather<<<groups, 64>>>(d_rec, records - 1, iters, d_out, k, spread);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double wave_steps = (double)groups * iters, lane_steps = wave_steps * k;
            printf("{\"waves_per_cu\": %d, \"active_lanes\": %d, \"spread\": %d, \"ms\": %.3f, \"Gwave_steps_per_s\": %.2f, \"Glane_steps_per_s\": %.1f, \"GB_per_s\": %.0f, "
                   "\"cycles_per_wave_step_per_CU\": %.1f}\n", wpc ? wpc : 16, k, spread, best, wave_steps / best * 1e-6, lane_steps / best * 1e-6,
                   lane_steps * 64 / best * 1e-6, best * 1e-3 * 2.4e9 * 256 / wave_steps);
            fflush(stdout);
        }
    return 0;
}
