// Host-side vector/matrix types with the names and call syntax the reference's hot-path code uses from
// FW's base/Math.hpp (Vec2i, Vec2f, Vec3i, Vec3f, Vec4f, Mat3f, Mat4f; column-major matrices).  FW itself is
// not in the reference's tree, so operation order is this build's definition: sums left-to-right,
// normalized() = v * (1 / length).  Compiled with -ffp-contract=off so results match the device and the
// CPU oracle bit for bit.
#pragma once
#include <cmath>

#include "Defs.hpp"

namespace FW {

template <class T> inline T min(T a, T b) { return (a < b) ? a : b; }
template <class T> inline T max(T a, T b) { return (a > b) ? a : b; }
template <class T> inline T min(T a, T b, T c) { return min(min(a, b), c); }
template <class T> inline T max(T a, T b, T c) { return max(max(a, b), c); }
inline F32 abs(F32 a) { return std::fabs(a); }
inline F32 sqrt(F32 a) { return std::sqrt(a); }
inline F32 floor(F32 a) { return std::floor(a); }
inline F32 pow(F32 a, F32 b) { return std::pow(a, b); }
inline F32 exp(F32 a) { return std::exp(a); }
inline F32 sin(F32 a) { return std::sin(a); }
inline F32 cos(F32 a) { return std::cos(a); }
inline F32 rcp(F32 a) { return 1.0f / a; }

struct Vec2i {
    S32 x = 0, y = 0;
    Vec2i() {}
    Vec2i(S32 x_, S32 y_) : x(x_), y(y_) {}
    bool operator==(const Vec2i& o) const { return x == o.x && y == o.y; }
    bool operator!=(const Vec2i& o) const { return !(*this == o); }
};
struct Vec3i {
    S32 x = 0, y = 0, z = 0;
    Vec3i() {}
    Vec3i(S32 x_, S32 y_, S32 z_) : x(x_), y(y_), z(z_) {}
    S32& operator[](int i) { return (&x)[i]; }
    S32 operator[](int i) const { return (&x)[i]; }
};
struct Vec2f {
    F32 x = 0, y = 0;
    Vec2f() {}
    explicit Vec2f(F32 s) : x(s), y(s) {}
    Vec2f(F32 x_, F32 y_) : x(x_), y(y_) {}
    Vec2f(const Vec2i& v) : x((F32)v.x), y((F32)v.y) {}
    operator Vec2i() const { return Vec2i((S32)x, (S32)y); }  // truncation, as the reference's texel look-ups rely on
    Vec2f operator+(const Vec2f& o) const { return Vec2f(x + o.x, y + o.y); }
    Vec2f operator-(const Vec2f& o) const { return Vec2f(x - o.x, y - o.y); }
    Vec2f operator*(F32 s) const { return Vec2f(x * s, y * s); }
    F32& operator[](int i) { return (&x)[i]; }
    F32 operator[](int i) const { return (&x)[i]; }
};
inline Vec2f operator*(F32 s, const Vec2f& v) { return Vec2f(v.x * s, v.y * s); }

struct Vec3f {
    F32 x = 0, y = 0, z = 0;
    Vec3f() {}
    explicit Vec3f(F32 s) : x(s), y(s), z(s) {}
    Vec3f(F32 x_, F32 y_, F32 z_) : x(x_), y(y_), z(z_) {}
    Vec3f(const Vec2f& v, F32 z_) : x(v.x), y(v.y), z(z_) {}
    F32& operator[](int i) { return (&x)[i]; }
    F32 operator[](int i) const { return (&x)[i]; }
    Vec3f operator+(const Vec3f& o) const { return Vec3f(x + o.x, y + o.y, z + o.z); }
    Vec3f operator-(const Vec3f& o) const { return Vec3f(x - o.x, y - o.y, z - o.z); }
    Vec3f operator-() const { return Vec3f(-x, -y, -z); }
    Vec3f operator*(const Vec3f& o) const { return Vec3f(x * o.x, y * o.y, z * o.z); }
    Vec3f operator*(F32 s) const { return Vec3f(x * s, y * s, z * s); }
    Vec3f& operator+=(const Vec3f& o) { return *this = *this + o; }
    Vec3f& operator*=(const Vec3f& o) { return *this = *this * o; }
    Vec3f& operator*=(F32 s) { return *this = *this * s; }
    F32 dot(const Vec3f& o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3f cross(const Vec3f& o) const { return Vec3f(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
    F32 length() const { return std::sqrt(dot(*this)); }
    Vec3f normalized() const { return *this * (1.0f / length()); }
    Vec2f getXY() const { return Vec2f(x, y); }
    bool operator==(const Vec3f& o) const { return x == o.x && y == o.y && z == o.z; }
};
inline Vec3f operator*(F32 s, const Vec3f& v) { return v * s; }
inline F32 dot(const Vec3f& a, const Vec3f& b) { return a.dot(b); }
inline Vec3f cross(const Vec3f& a, const Vec3f& b) { return a.cross(b); }
inline Vec3f normalize(const Vec3f& v) { return v.normalized(); }
inline Vec3f min(const Vec3f& a, const Vec3f& b) { return Vec3f(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
inline Vec3f max(const Vec3f& a, const Vec3f& b) { return Vec3f(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline Vec3f min(const Vec3f& a, const Vec3f& b, const Vec3f& c) {
    return Vec3f(min(a.x, b.x, c.x), min(a.y, b.y, c.y), min(a.z, b.z, c.z));
}
inline Vec3f max(const Vec3f& a, const Vec3f& b, const Vec3f& c) {
    return Vec3f(max(a.x, b.x, c.x), max(a.y, b.y, c.y), max(a.z, b.z, c.z));
}

struct Vec4f {
    F32 x = 0, y = 0, z = 0, w = 0;
    Vec4f() {}
    explicit Vec4f(F32 s) : x(s), y(s), z(s), w(s) {}
    Vec4f(F32 x_, F32 y_, F32 z_, F32 w_) : x(x_), y(y_), z(z_), w(w_) {}
    Vec4f(const Vec3f& v, F32 w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    Vec3f getXYZ() const { return Vec3f(x, y, z); }
    Vec4f operator+(const Vec4f& o) const { return Vec4f(x + o.x, y + o.y, z + o.z, w + o.w); }
    Vec4f operator*(F32 s) const { return Vec4f(x * s, y * s, z * s, w * s); }
    Vec4f& operator+=(const Vec4f& o) { return *this = *this + o; }
    Vec4f& operator/=(F32 s) { x /= s; y /= s; z /= s; w /= s; return *this; }
    F32& operator[](int i) { return (&x)[i]; }
    F32 operator[](int i) const { return (&x)[i]; }
};
inline Vec4f operator*(F32 s, const Vec4f& v) { return v * s; }
inline Vec4f clamp(const Vec4f& v, F32 lo, F32 hi) {
    auto c = [lo, hi](F32 a) { return a < lo ? lo : (a > hi ? hi : a); };
    return Vec4f(c(v.x), c(v.y), c(v.z), c(v.w));
}

// Column-major 3x3.
class Mat3f {
public:
    F32 m[9];  // m[col * 3 + row]
    Mat3f() { setIdentity(); }
    void setIdentity() { for (int i = 0; i < 9; ++i) m[i] = (i % 4 == 0) ? 1.0f : 0.0f; }
    F32 get(int r, int c) const { return m[c * 3 + r]; }
    F32& at(int r, int c) { return m[c * 3 + r]; }
    void setCol(int c, const Vec3f& v) { m[c * 3] = v.x; m[c * 3 + 1] = v.y; m[c * 3 + 2] = v.z; }
    Vec3f getCol(int c) const { return Vec3f(m[c * 3], m[c * 3 + 1], m[c * 3 + 2]); }
    Vec3f operator*(const Vec3f& v) const {
        return Vec3f(get(0, 0) * v.x + get(0, 1) * v.y + get(0, 2) * v.z, get(1, 0) * v.x + get(1, 1) * v.y + get(1, 2) * v.z,
                     get(2, 0) * v.x + get(2, 1) * v.y + get(2, 2) * v.z);
    }
    Mat3f operator-() const { Mat3f r; for (int i = 0; i < 9; ++i) r.m[i] = -m[i]; return r; }
    Mat3f transposed() const { Mat3f r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.at(i, j) = get(j, i); return r; }
    // in-place inverse: adjugate over determinant (cofactor expansion along the first row)
    void invert() {
        F32 a = get(0, 0), b = get(0, 1), c = get(0, 2), d = get(1, 0), e = get(1, 1), f = get(1, 2), g = get(2, 0), h = get(2, 1), i = get(2, 2);
        F32 c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
        F32 r = 1.0f / (a * c00 + b * c01 + c * c02);
        at(0, 0) = c00 * r; at(0, 1) = (c * h - b * i) * r; at(0, 2) = (b * f - c * e) * r;
        at(1, 0) = c01 * r; at(1, 1) = (a * i - c * g) * r; at(1, 2) = (c * d - a * f) * r;
        at(2, 0) = c02 * r; at(2, 1) = (b * g - a * h) * r; at(2, 2) = (a * e - b * d) * r;
    }
};

// Column-major 4x4.
class Mat4f {
public:
    F32 m[16];  // m[col * 4 + row]
    Mat4f() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    F32 get(int r, int c) const { return m[c * 4 + r]; }
    F32& at(int r, int c) { return m[c * 4 + r]; }
    void setCol(int c, const Vec4f& v) { m[c * 4] = v.x; m[c * 4 + 1] = v.y; m[c * 4 + 2] = v.z; m[c * 4 + 3] = v.w; }
    Vec4f getCol(int c) const { return Vec4f(m[c * 4], m[c * 4 + 1], m[c * 4 + 2], m[c * 4 + 3]); }
    void setRow(int r, const Vec4f& v) { at(r, 0) = v.x; at(r, 1) = v.y; at(r, 2) = v.z; at(r, 3) = v.w; }
    Mat3f getXYZ() const { Mat3f r; for (int c = 0; c < 3; ++c) r.setCol(c, getCol(c).getXYZ()); return r; }
    Vec4f operator*(const Vec4f& v) const {
        return Vec4f(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
                     m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w, m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w);
    }
    Mat4f operator*(const Mat4f& o) const {
        Mat4f r;
        for (int c = 0; c < 4; ++c)
            for (int rr = 0; rr < 4; ++rr)
                r.at(rr, c) = get(rr, 0) * o.get(0, c) + get(rr, 1) * o.get(1, c) + get(rr, 2) * o.get(2, c) + get(rr, 3) * o.get(3, c);
        return r;
    }
    bool operator!=(const Mat4f& o) const { for (int i = 0; i < 16; ++i) if (m[i] != o.m[i]) return true; return false; }
    // Gauss-Jordan with partial pivoting, carried out in double and rounded once (the camera matrix is inverted once per render)
    Mat4f inverted() const {
        double a[4][8];
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = get(r, c); a[r][c + 4] = (r == c) ? 1.0 : 0.0; }
        for (int col = 0; col < 4; ++col) {
            int piv = col;
            for (int r = col + 1; r < 4; ++r) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
            if (piv != col) for (int c = 0; c < 8; ++c) { double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
            double inv = 1.0 / a[col][col];
            for (int c = 0; c < 8; ++c) a[col][c] *= inv;
            for (int r = 0; r < 4; ++r) if (r != col) { double f = a[r][col]; for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c]; }
        }
        Mat4f out;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out.at(r, c) = (F32)a[r][c + 4];
        return out;
    }
    static Mat4f scale(const Vec3f& s) { Mat4f r; r.at(0, 0) = s.x; r.at(1, 1) = s.y; r.at(2, 2) = s.z; return r; }
    static Mat4f translate(const Vec3f& t) { Mat4f r; r.at(0, 3) = t.x; r.at(1, 3) = t.y; r.at(2, 3) = t.z; return r; }
    // GL-style perspective: fov in degrees (vertical), clip z in [-1, 1]
    static Mat4f perspective(F32 fov, F32 nearDist, F32 farDist) {
        F32 f = 1.0f / std::tan(fov * FW_PI / 360.0f);
        F32 d = 1.0f / (nearDist - farDist);
        Mat4f r;
        r.setRow(0, Vec4f(f, 0, 0, 0));
        r.setRow(1, Vec4f(0, f, 0, 0));
        r.setRow(2, Vec4f(0, 0, (nearDist + farDist) * d, 2.0f * nearDist * farDist * d));
        r.setRow(3, Vec4f(0, 0, -1.0f, 0));
        return r;
    }
    // uniform fit of the rectangle (pos, size) into a viewport of viewSize pixels, centred
    static Mat4f fitToView(const Vec2f& pos, const Vec2f& size, const Vec2f& viewSize) {
        F32 s = min(viewSize.x / size.x, viewSize.y / size.y);
        Mat4f sc = scale(Vec3f(s / viewSize.x * 2.0f, s / viewSize.y * 2.0f, 1.0f));
        Mat4f tr = translate(Vec3f((pos.x + size.x * 0.5f) * -1.0f, (pos.y + size.y * 0.5f) * -1.0f, 0.0f));
        return sc * tr;
    }
};

}  // namespace FW
