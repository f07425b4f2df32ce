// Measures the ceiling the traversal kernels run under: the rate at which a CU's vector memory path delivers 64-byte records gathered
// per lane (every lane of a wave reads its own record, as a BVH step does) from a table that stays in cache. Not part of the product:
// a micro-benchmark whose result DESIGN.md and bench.py's roofline note quote.
//   hipcc -O3 --offload-arch=gfx950 tools/l1_gather_peak.hip -o build/l1_gather_peak && build/l1_gather_peak
// Prints one JSON line per case: table size, dependent (next index from the loaded data, like a tree walk) or independent indices,
// waves per SIMD allowed by the launch, GB/s.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CTRL> __device__ inline uint32_t quad(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ inline uint4 quad4(uint4 v) { return make_uint4(quad<CTRL>(v.x), quad<CTRL>(v.y), quad<CTRL>(v.z), quad<CTRL>(v.w)); }
__device__ inline uint4 sel(bool c, uint4 a, uint4 b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

// The four lanes of a quad fetch each of their four records together: load k reads record k's 64 bytes as 4 x 16 contiguous bytes
// (one lane each), then a 4x4 transpose inside the quad (DPP quad_perm, no LDS) hands every lane its own record.
__device__ inline void load_record_quad(const uint4* __restrict__ rec, uint32_t idx, uint4& a, uint4& b, uint4& c, uint4& d)
{
    const uint32_t j = threadIdx.x & 3u;
    uint4 m0 = rec[(size_t)quad<0x00>(idx) * 4 + j], m1 = rec[(size_t)quad<0x55>(idx) * 4 + j];
    uint4 m2 = rec[(size_t)quad<0xAA>(idx) * 4 + j], m3 = rec[(size_t)quad<0xFF>(idx) * 4 + j];
    const bool odd = j & 1u, hi = j & 2u;
    uint4 t;
    t = sel(odd, quad4<0xB1>(m1), m0); m1 = sel(!odd, quad4<0xB1>(m0), m1); m0 = t;
    t = sel(odd, quad4<0xB1>(m3), m2); m3 = sel(!odd, quad4<0xB1>(m2), m3); m2 = t;
    t = sel(hi, quad4<0x4E>(m2), m0); m2 = sel(!hi, quad4<0x4E>(m0), m2); m0 = t;
    t = sel(hi, quad4<0x4E>(m3), m1); m3 = sel(!hi, quad4<0x4E>(m1), m3); m1 = t;
    a = m0; b = m1; c = m2; d = m3;
}

template <int DEP, int WAVES, int COOP = 0>
__global__ __launch_bounds__(64, WAVES) void k_gather(const uint4* __restrict__ rec, uint32_t mask, int iters, uint32_t* __restrict__ out, uint32_t share)
{
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    uint32_t state = (gid / share) * 2654435761u,  // `share` neighbouring lanes walk the same records, as coherent rays do
             idx = state >> 7 & mask, acc = 0;
    for (int i = 0; i < iters; i++) {
        uint4 a, b, c, d;
        if (COOP) load_record_quad(rec, idx, a, b, c, d);
        else { const uint4* p = rec + (size_t)idx * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
        const uint32_t s = (a.x ^ a.y ^ a.z ^ a.w) + 3u * (b.x ^ b.y ^ b.z ^ b.w) + 5u * (c.x ^ c.y ^ c.z ^ c.w) + 7u * (d.x ^ d.y ^ d.z ^ d.w);
        acc += s;
        if (DEP) { state += 0x9E3779B9u; idx = (s ^ state) >> 3 & mask; }  // depends on the data, never settles into a cycle
        else { state = state * 1664525u + 1013904223u; idx = state >> 10 & mask; }
    }
    out[gid] = acc;
}

template <int DEP, int WAVES, int COOP = 0>
static void run(const uint4* d_rec, uint32_t records, uint32_t* d_out, int groups, int iters, uint32_t share = 1)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_gather<DEP, WAVES, COOP><<<groups, 64>>>(d_rec, records - 1, iters / 10, d_out, share);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        k_gather<DEP, WAVES, COOP><<<groups, 64>>>(d_rec, records - 1, iters, d_out, share);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)groups * 64 * iters * 64;
    printf("{\"table_KiB\": %u, \"dependent\": %d, \"waves_per_simd\": %d, \"lanes_per_record\": %u, \"quad_loads\": %d, \"ms\": %.4f, \"GB_per_s\": %.1f}\n", records / 16, DEP, WAVES, share, COOP, best, bytes / best * 1e-6);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main()
{
    const uint32_t max_records = 1u << 16;  // 4 MiB
    std::vector<uint32_t> h((size_t)max_records * 16);
    uint32_t x = 12345u;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = x; }
    uint4* d_rec; uint32_t* d_out;
    const int groups = 256 * 4 * 8 * 2, iters = 4000;
    CK(hipMalloc(&d_rec, h.size() * 4));
    CK(hipMalloc(&d_out, (size_t)groups * 64 * 4));
    CK(hipMemcpy(d_rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (uint32_t records : {256u, 4096u, 32768u, 65536u}) {  // 16 KiB (L1), 256 KiB, 2 MiB (head.obj's tree: 1.6 MB), 4 MiB (one XCD's L2)
        run<1, 8>(d_rec, records, d_out, groups, iters);
        run<1, 4>(d_rec, records, d_out, groups, iters);
        run<0, 8>(d_rec, records, d_out, groups, iters);
    }
    for (uint32_t share : {2u, 4u, 8u, 16u, 64u}) run<1, 8>(d_rec, 32768u, d_out, groups, iters, share);
    // quad-cooperative loads: same answers, and what they cost
    std::vector<uint32_t> r0((size_t)groups * 64), r1(r0.size());
    run<1, 8>(d_rec, 32768u, d_out, groups, iters);
    CK(hipMemcpy(r0.data(), d_out, r0.size() * 4, hipMemcpyDeviceToHost));
    run<1, 8, 1>(d_rec, 32768u, d_out, groups, iters);
    CK(hipMemcpy(r1.data(), d_out, r1.size() * 4, hipMemcpyDeviceToHost));
    if (r0 != r1) { fprintf(stderr, "quad loads: results differ\n"); return 1; }
    run<1, 4, 1>(d_rec, 32768u, d_out, groups, iters);
    run<1, 8, 1>(d_rec, 256u, d_out, groups, iters);
    run<1, 8, 1>(d_rec, 32768u, d_out, groups, iters, 4);
    CK(hipFree(d_rec)); CK(hipFree(d_out));
    return 0;
}
