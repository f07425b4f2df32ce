// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked or imported by the product
// (pathtracer_amd/); only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// Minimal vector/matrix math for the CPU restatement of the reference's hot
// path.  The reference uses NVIDIA's "FW" framework (base/Math.hpp), which is
// NOT in its tree, so the exact operation order of FW's
// operators is "parity unpinned"; this file fixes one definition (left-to-right
// sums, normalize = v * (1/len)) that the HIP kernels restate bit-for-bit.
// Build with -ffp-contract=off: no FMA contraction anywhere.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace pto {

struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct V4 { float x, y, z, w; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V3& operator*=(V3& a, V3 b) { a = a * b; return a; }
inline V3& operator*=(V3& a, float s) { a = a * s; return a; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }

inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
inline V2 operator*(float s, V2 a) { return V2{a.x * s, a.y * s}; }

// FW::min / FW::max semantics: plain compare-select (NaN in b propagates).
inline float fmin2(float a, float b) { return (a < b) ? a : b; }
inline float fmax2(float a, float b) { return (a > b) ? a : b; }
inline float fmin3(float a, float b, float c) { return fmin2(fmin2(a, b), c); }
inline float fmax3(float a, float b, float c) { return fmax2(fmax2(a, b), c); }

// Column-major 3x3 (FW Mat3f: setCol/getCol, M*v).
struct M3 {
    float m[9];  // m[col*3 + row]
    float at(int r, int c) const { return m[c * 3 + r]; }
    void setCol(int c, V3 v) { m[c * 3] = v.x; m[c * 3 + 1] = v.y; m[c * 3 + 2] = v.z; }
    V3 col(int c) const { return V3{m[c * 3], m[c * 3 + 1], m[c * 3 + 2]}; }
};
inline V3 operator*(const M3& M, V3 v) {
    return V3{M.at(0, 0) * v.x + M.at(0, 1) * v.y + M.at(0, 2) * v.z,
              M.at(1, 0) * v.x + M.at(1, 1) * v.y + M.at(1, 2) * v.z,
              M.at(2, 0) * v.x + M.at(2, 1) * v.y + M.at(2, 2) * v.z};
}
// Inverse by adjugate / determinant (FW's Mat3f::invert rounding is unpinned).
inline M3 inverse(const M3& A) {
    float a = A.at(0, 0), b = A.at(0, 1), c = A.at(0, 2);
    float d = A.at(1, 0), e = A.at(1, 1), f = A.at(1, 2);
    float g = A.at(2, 0), h = A.at(2, 1), i = A.at(2, 2);
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float det = a * c00 + b * c01 + c * c02;
    float r = 1.0f / det;
    M3 R;
    R.m[0] = c00 * r;             R.m[3] = (c * h - b * i) * r; R.m[6] = (b * f - c * e) * r;
    R.m[1] = c01 * r;             R.m[4] = (a * i - c * g) * r; R.m[7] = (c * d - a * f) * r;
    R.m[2] = c02 * r;             R.m[5] = (b * g - a * h) * r; R.m[8] = (a * e - b * d) * r;
    return R;
}

// Column-major 4x4 times (x,y,z,w), left-to-right sums.
inline V4 mul44(const float* m, V4 v) {
    return V4{m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w,
              m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
              m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w,
              m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w};
}

// ---------------------------------------------------------------------------
// Transcendentals.  LibmMath = what the reference calls (glibc here, MSVC CRT on
// its native platform; ulp-level results differ between C libraries).  PtmMath =
// "portable math": fixed polynomial kernels (Cephes single-precision
// coefficients) written with + - * only, so the CPU oracle and the HIP kernels
// produce bit-identical values.  tests compare PtmMath images with LibmMath
// images by RMSE, and GPU output with PtmMath output exactly.
// ---------------------------------------------------------------------------
struct LibmMath {
    static float sin(float x) { return std::sin(x); }
    static float cos(float x) { return std::cos(x); }
    static float exp(float x) { return std::exp(x); }
    static float pow(float x, float y) { return std::pow(x, y); }
};

inline float ptm_pow2i(int n) {  // 2^n for n in [-126, 127]
    uint32_t b = (uint32_t)(n + 127) << 23;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
inline void ptm_sincos(float x, float& s, float& c) {
    float q = std::floor(x * 0.636619772f + 0.5f);
    int k = (int)q;
    float r = ((x - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188216e-8f;
    float r2 = r * r;
    float ps = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    float pc = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2
               - 0.5f * r2 + 1.0f;
    switch (k & 3) {
        case 0: s = ps; c = pc; break;
        case 1: s = pc; c = -ps; break;
        case 2: s = -ps; c = -pc; break;
        default: s = -pc; c = ps; break;
    }
}
inline float ptm_exp(float x) {
    if (!(x > -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.0f) return INFINITY;
    float n = std::floor(x * 1.44269504088896341f + 0.5f);
    float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r
               + 1.6666665459e-1f) * r + 5.0000001201e-1f;
    float e = p * (r * r) + r + 1.0f;
    int ni = (int)n;
    // split the scale so 2^n stays a normal float for the whole accepted range
    int n1 = ni / 2, n2 = ni - n1;
    return e * ptm_pow2i(n1) * ptm_pow2i(n2);
}
inline float ptm_log(float x) {  // x > 0, normal
    uint32_t b;
    std::memcpy(&b, &x, 4);
    int e = (int)(b >> 23) - 126;
    b = (b & 0x007fffffu) | 0x3f000000u;  // mantissa in [0.5, 1)
    float m;
    std::memcpy(&m, &b, 4);
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
inline float ptm_pow(float x, float y) {  // x >= 0 only (all call sites clamp or are colours/distances)
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (!(x > 0.0f)) return (y > 0.0f) ? 0.0f : INFINITY;
    if (x == INFINITY) return (y > 0.0f) ? INFINITY : 0.0f;
    if (x < 1.17549435e-38f) return (y > 0.0f) ? 0.0f : INFINITY;  // subnormal base: flush (not reached in practice)
    return ptm_exp(y * ptm_log(x));
}
struct PtmMath {
    static float sin(float x) { float s, c; ptm_sincos(x, s, c); return s; }
    static float cos(float x) { float s, c; ptm_sincos(x, s, c); return c; }
    static float exp(float x) { return ptm_exp(x); }
    static float pow(float x, float y) { return ptm_pow(x, y); }
};

}  // namespace pto
