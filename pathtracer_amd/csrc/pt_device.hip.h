// Device-side arithmetic for the gfx950 path-tracing kernels.
//
// Everything here is fp32 evaluated left-to-right with NO FMA contraction (the translation
// unit is compiled with -ffp-contract=off), matching the operation order of the reference's
// CPU code so that hit records are bit-identical (SURVEY.md 7 "Bit-exactness").
// min/max are compare-selects, not v_min/v_max: the reference's FW::min/max propagate NaN
// from the second operand and the slab test relies on plain comparisons being false on NaN.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct F2 { float x, y; };
struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };

__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// sqrtf and '/' are IEEE-rounded here (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); the
// __fsqrt_rn intrinsic is NOT (it maps to the native approximate sqrt in this ROCm).
__device__ __forceinline__ float length(F3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ F3 normalize(F3 a) { return a * (1.0f / length(a)); }
__device__ __forceinline__ float sel_min(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float sel_max(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { F2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
__device__ __forceinline__ F2 operator-(F2 a, F2 b) { F2 r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
__device__ __forceinline__ F2 operator*(float s, F2 a) { F2 r; r.x = a.x * s; r.y = a.y * s; return r; }

__device__ __forceinline__ F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ F2 ld2(const float* p) { F2 r; r.x = p[0]; r.y = p[1]; return r; }

// column-major 4x4 times (x,y,z,w)
__device__ __forceinline__ F4 mul44(const float* m, float x, float y, float z, float w) {
    F4 r;
    r.x = m[0] * x + m[4] * y + m[8] * z + m[12] * w;
    r.y = m[1] * x + m[5] * y + m[9] * z + m[13] * w;
    r.z = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
    r.w = m[3] * x + m[7] * y + m[11] * z + m[15] * w;
    return r;
}

// ---- Sobol' sampler: sobol::sample (reference src/sobol.h:39-49); table built by tools/gen_sobol_table.py ----
extern __device__ const uint32_t g_sobol_matrices[1024 * 52];

__device__ __forceinline__ float sobol_sample(unsigned long long index, unsigned dimension) {
    unsigned result = 0u;
    const uint32_t* row = g_sobol_matrices + dimension * 52u;
    for (unsigned i = 0; index; index >>= 1, ++i)
        if (index & 1ull) result ^= row[i];
    return (float)result * (1.f / 4294967296.f);
}

// two consecutive dimensions (dimension even) of one index -- the reference always draws them in pairs.  The low 24 index bits go
// through the byte table (ptgpu.hip k_build_sobol_pairs), any higher bits through sobol::sample's bit loop over `matrices`
// (= g_sobol_matrices; a parameter so that a translation unit other than the one defining the table can pass its address in).
__device__ __forceinline__ void sobol32x2m(const uint2* __restrict__ tab, const uint32_t* __restrict__ matrices, uint32_t index, unsigned dimension,
                                           float& a, float& b) {
    const uint2* t = tab + (size_t)(dimension >> 1) * 768;
    const uint2 e0 = t[index & 255u], e1 = t[256u + ((index >> 8) & 255u)], e2 = t[512u + ((index >> 16) & 255u)];
    unsigned ra = e0.x ^ e1.x ^ e2.x, rb = e0.y ^ e1.y ^ e2.y;
    index >>= 24;
    if (index) {
        const uint32_t* row = matrices + dimension * 52u + 24u;
        for (unsigned i = 0; index; index >>= 1, ++i) {
            uint32_t m = 0u - (index & 1u);
            ra ^= row[i] & m;
            rb ^= row[i + 52] & m;
        }
    }
    a = (float)ra * (1.f / 4294967296.f);
    b = (float)rb * (1.f / 4294967296.f);
}

// ---- stand-in for FW::Random (absent from the reference's tree): counter-based, one stream per sample ----
struct Rng {
    uint32_t key, ctr;
    __device__ __forceinline__ static uint32_t mix(uint32_t x) {
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return x;
    }
    __device__ __forceinline__ void seed(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
        key = mix(k3 + mix(k2 + mix(k1 + mix(k0 + 0x9E3779B9u))));
        ctr = 0;
    }
    __device__ __forceinline__ uint32_t at(uint32_t n) const { return mix(key ^ mix(n * 0x9E3779B9u + 0x7F4A7C15u)); }
    __device__ __forceinline__ uint32_t u32() { return at(ctr++); }
    __device__ __forceinline__ static int to_s32(uint32_t u, int lo, int hi) {
        return (hi <= lo) ? lo : lo + (int)(u % (uint32_t)(hi - lo));
    }
    __device__ __forceinline__ int getS32(int lo, int hi) { return to_s32(u32(), lo, hi); }
    __device__ __forceinline__ int peekS32(int lo, int hi) const { return to_s32(at(ctr), lo, hi); }
    __device__ __forceinline__ float getF32(float lo, float hi) {
        return ((float)u32() * (1.0f / 4294967296.0f)) * (hi - lo) + lo;
    }
};

// ---- portable transcendentals: fixed polynomial kernels, + - * only, bit-identical to oracle PtmMath ----
__device__ __forceinline__ float pow2i(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }

__device__ __forceinline__ void ptm_sincos(float x, float& s, float& c) {
    float q = floorf(x * 0.636619772f + 0.5f);
    int k = (int)q;
    float r = ((x - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188216e-8f;
    float r2 = r * r;
    float ps = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    float pc = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2
               - 0.5f * r2 + 1.0f;
    int m = k & 3;
    float ss = (m & 1) ? pc : ps;
    float cc = (m & 1) ? ps : pc;
    s = (m & 2) ? -ss : ss;
    c = (m == 1 || m == 2) ? -cc : cc;
}
__device__ __forceinline__ float ptm_exp(float x) {
    if (!(x > -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.0f) return __uint_as_float(0x7f800000u);
    float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r
               + 1.6666665459e-1f) * r + 5.0000001201e-1f;
    float e = p * (r * r) + r + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    return e * pow2i(n1) * pow2i(n2);
}
__device__ __forceinline__ float ptm_log(float x) {
    uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 126;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                  + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y = y + fe * -2.12194440e-4f;
    y = y - 0.5f * z;
    return (m + y) + fe * 0.693359375f;
}
__device__ __forceinline__ float ptm_pow(float x, float y) {
    const float INF = __uint_as_float(0x7f800000u);
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (!(x > 0.0f)) return (y > 0.0f) ? 0.0f : INF;
    if (x == INF) return (y > 0.0f) ? INF : 0.0f;
    if (x < 1.17549435e-38f) return (y > 0.0f) ? 0.0f : INF;
    return ptm_exp(y * ptm_log(x));
}

}  // namespace ptd
