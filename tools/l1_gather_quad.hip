// Companion of l1_gather_partial.hip: the same dependent gather of 64-byte records, fetched by FOUR lanes per record (lane & 3 takes one
// dwordx4 quarter: ONE load instruction per step, 16 records per wave64) instead of one lane per record (four load instructions, k records
// per wave).  The vector memory path charges a load instruction 16 cycles whatever its exec mask, so a wave of one-ray lanes at 30-50 %
// utilisation pays 64 cycles for 20-30 records; does a quad-per-record wave really get 16 records for 16?  The next index is reduced over
// the quad with DPP row shuffles (what a traverser would do with its partial slab results).
//   hipcc -O3 --offload-arch=gfx950 tools/l1_gather_quad.hip -o build/l1_gather_quad && build/l1_gather_quad
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t quad_xor(uint32_t v, int m) {
    // quad_perm DPP: lane ^ 1 = [1,0,3,2] (0xB1), lane ^ 2 = [2,3,0,1] (0x4E)
    return m == 1 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true) : (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
}

// MODE 0: one lane per record, k lanes active (lanes 0..k-1).  MODE 1: one quad per record, kq of 16 quads active.
// MODE 2: one lane per record as in mode 0 (k lanes, spread over the wave), but the records are FETCHED by quads: the active lanes post
// their indices to LDS by rank, quad q loads the record of rank 16 j + q (one dwordx4 per round j), parks it in LDS, and the owners
// read their 64 bytes back -- ceil(k / 16) load instructions instead of 4, at the price of LDS round trips.
template <int MODE>
__global__ __launch_bounds__(64, 4) void k_gather(const uint4* __restrict__ rec, uint32_t mask, int iters, uint32_t* __restrict__ out, int k)
{
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64u + lane;
    uint32_t acc = 0;
    if (MODE == 0) {
        uint32_t state = gid * 2654435761u, idx = state >> 7 & mask;
        if (lane < (uint32_t)k)
            for (int i = 0; i < iters; i++) {
                const uint4* p = rec + (size_t)idx * 4;
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                const uint32_t s = (a.x ^ a.y ^ a.z ^ a.w) + 3u * (b.x ^ b.y ^ b.z ^ b.w) + 5u * (c.x ^ c.y ^ c.z ^ c.w) + 7u * (d.x ^ d.y ^ d.z ^ d.w);
                acc += s;
                state += 0x9E3779B9u;
                idx = (s ^ state) >> 3 & mask;
            }
    } else if (MODE == 2) {
        __shared__ uint32_t refs[64];
        __shared__ uint4 stage[4][17];  // [quarter][rank in round], rows 16 bytes apart in bank space
        const bool active = (lane * (uint32_t)k) / 64u != ((lane + 1u) * (uint32_t)k) / 64u;
        uint32_t state = gid * 2654435761u, idx = state >> 7 & mask;
        const unsigned long long act = __ballot(active);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        const uint32_t sub = lane & 3u, q = lane >> 2;
        const int rounds = (k + 15) / 16;
        for (int i = 0; i < iters; i++) {
            if (active) refs[rank] = idx;
            uint4 a, b, c, d;
            for (int j = 0; j < rounds; ++j) {
                if (16u * j + q < (uint32_t)k) stage[sub][q] = rec[(size_t)refs[16 * j + q] * 4 + sub];
                if (active && (rank >> 4) == (uint32_t)j) { a = stage[0][rank & 15]; b = stage[1][rank & 15]; c = stage[2][rank & 15]; d = stage[3][rank & 15]; }
            }
            if (active) {
                const uint32_t s = (a.x ^ a.y ^ a.z ^ a.w) + 3u * (b.x ^ b.y ^ b.z ^ b.w) + 5u * (c.x ^ c.y ^ c.z ^ c.w) + 7u * (d.x ^ d.y ^ d.z ^ d.w);
                acc += s;
                state += 0x9E3779B9u;
                idx = (s ^ state) >> 3 & mask;
            }
        }
    } else {
        const uint32_t sub = lane & 3u, quad = gid >> 2;
        const uint32_t mul = 1u + 2u * sub;
        uint32_t state = quad * 2654435761u, idx = state >> 7 & mask;
        if ((lane >> 2) < (uint32_t)k)
            for (int i = 0; i < iters; i++) {
                const uint4 a = rec[(size_t)idx * 4 + sub];
                uint32_t s = mul * (a.x ^ a.y ^ a.z ^ a.w);
                s += quad_xor(s, 1);
                s += quad_xor(s, 2);
                acc += s;
                state += 0x9E3779B9u;
                idx = (s ^ state) >> 3 & mask;
            }
    }
    out[gid] = acc;
}

int main(int argc, char** argv)
{
    const uint32_t records = argc > 1 ? (uint32_t)atoi(argv[1]) : 32768u;  // 2 MiB: head.obj's tree is 1.6 MB
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
    for (int mode = 0; mode < 3; ++mode)
        for (int wpc : {8, 16, 32})
            for (int k : (mode == 1 ? std::vector<int>{4, 8, 12, 16} : mode == 2 ? std::vector<int>{8, 12, 16, 24, 32} : std::vector<int>{8, 16, 24, 32, 48, 64})) {
                const int groups = 256 * wpc;
                auto launch = [&](int it) {
                    if (mode == 2) k_gather<2><<<groups, 64>>>(d_rec, records - 1, it, d_out, k);
                    else if (mode) k_gather<1><<<groups, 64>>>(d_rec, records - 1, it, d_out, k);
                    else k_gather<0><<<groups, 64>>>(d_rec, records - 1, it, d_out, k);
                };
                launch(iters / 10);
                CK(hipDeviceSynchronize());
                float best = 1e30f;
                for (int r = 0; r < 3; r++) {
                    CK(hipEventRecord(e0));
                    launch(iters);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double wave_steps = (double)groups * iters, rec_steps = wave_steps * k;
                printf("{\"mode\": \"%s\", \"waves_per_cu\": %d, \"records_per_wave_step\": %d, \"ms\": %.3f, \"Grecords_per_s\": %.1f, \"GB_per_s\": %.0f, "
                       "\"cycles_per_wave_step_per_CU\": %.1f, \"cycles_per_record_per_CU\": %.2f}\n", mode == 2 ? "lane_via_quads" : mode ? "quad" : "lane", wpc, k, best, rec_steps / best * 1e-6,
                       rec_steps * 64 / best * 1e-6, best * 1e-3 * 2.4e9 * 256 / wave_steps, best * 1e-3 * 2.4e9 * 256 / rec_steps);
                fflush(stdout);
            }
    return 0;
}
