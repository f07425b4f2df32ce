// ORACLE -- TEST INFRASTRUCTURE ONLY (see pt_math.h header note).
//
// CPU restatement of the reference's per-sample functions.  Each function names
// the reference lines it follows.  Templated on the math policy (LibmMath = what
// the reference calls; PtmMath = portable kernels the GPU reproduces bit-exactly).
#pragma once
#include <cfloat>
#include <vector>

#include "pt_math.h"
#include "pt_oracle.h"

namespace pto {

extern const uint32_t g_sobol_matrices[1024 * 52];

// reference src/sobol.h:39-49
inline float sobol_sample(unsigned long long index, unsigned dimension, unsigned scramble = 0u) {
    unsigned result = scramble;
    for (unsigned i = dimension * 52u; index; index >>= 1, ++i)
        if (index & 1) result ^= g_sobol_matrices[i];
    return result * (1.f / (1ULL << 32));
}

// ---------------------------------------------------------------------------
// Stand-in for FW::Random (base/Random.hpp is not in the reference's tree; the
// generator is unknown => "parity unpinned").  Counter-based so that every
// sample owns an independent stream: key = hash of 4 words, draw n = hash(key, n).
// getS32(lo,hi) / getF32(lo,hi) follow the FW call-site semantics (hi exclusive).
// ---------------------------------------------------------------------------
struct Rng {
    uint32_t key, ctr;
    static uint32_t mix(uint32_t x) {
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return x;
    }
    Rng(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) : ctr(0) {
        key = mix(k3 + mix(k2 + mix(k1 + mix(k0 + 0x9E3779B9u))));
    }
    static Rng from_state(uint32_t key_, uint32_t ctr_) { Rng r(0, 0, 0, 0); r.key = key_; r.ctr = ctr_; return r; }
    uint32_t at(uint32_t n) const { return mix(key ^ mix(n * 0x9E3779B9u + 0x7F4A7C15u)); }
    uint32_t u32() { return at(ctr++); }
    uint32_t peek_u32() const { return at(ctr); }
    static int to_s32(uint32_t u, int lo, int hi) { return (hi <= lo) ? lo : lo + (int)(u % (uint32_t)(hi - lo)); }
    int getS32(int lo, int hi) { return to_s32(u32(), lo, hi); }
    int peekS32(int lo, int hi) const { return to_s32(peek_u32(), lo, hi); }  // draw on a by-value copy (PathTraceRenderer.cpp:57,62)
    float getF32(float lo, float hi) { return ((float)u32() * (1.0f / 4294967296.0f)) * (hi - lo) + lo; }
};

struct Counters {
    uint64_t rays = 0, node_visits = 0, tri_tests = 0;
};

struct Hit {
    int tri;  // index into the caller's triangle arrays, -1 = miss (RaycastResult::tri == nullptr)
    float t, u, v;
    V3 point, orig, dir;
    explicit operator bool() const { return tri >= 0; }
};

inline V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V2 ld2(const float* p) { return V2{p[0], p[1]}; }

// Image::getVec4f on an 8-bit RGBA image: channel / 255 (FW's scale is unpinned).
inline V4 texel(const pto_texture& t, int x, int y) {
    // reference indexes out of bounds when frac() rounds to 1.0; clamp
    if (x < 0) x = 0;
    if (x > t.width - 1) x = t.width - 1;
    if (y < 0) y = 0;
    if (y > t.height - 1) y = t.height - 1;
    const uint8_t* p = t.rgba + ((size_t)y * t.width + x) * 4;
    return V4{p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f, p[3] / 255.0f};
}
// reference src/RayTracer.cpp:28-30, followed by the Vec2f->Vec2i truncation at the call sites
inline void texel_coords(V2 uv, const pto_texture& t, int& x, int& y) {
    x = (int)((uv.x - std::floor(uv.x)) * t.width);
    y = (int)((uv.y - std::floor(uv.y)) * t.height);
}

// reference src/BvhNode.cpp:9-57 (Williams et al. slab test)
inline bool box_intersect(const pto_flatnode& n, V3 orig, V3 inv, const int sign[3], float t1) {
    const float* b[2] = {n.bmin, n.bmax};
    float tmin = (b[sign[0]][0] - orig.x) * inv.x;
    float tmax = (b[1 - sign[0]][0] - orig.x) * inv.x;
    float tymin = (b[sign[1]][1] - orig.y) * inv.y;
    float tymax = (b[1 - sign[1]][1] - orig.y) * inv.y;
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (b[sign[2]][2] - orig.z) * inv.z;
    float tzmax = (b[1 - sign[2]][2] - orig.z) * inv.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < t1) && (tmax > 0.0f);
}

// reference src/RTTriangle.hpp:50-59; w = 3 rows of (M row | N)
inline bool intersect_woop(const float* w, V3 o, V3 d, float& t, float& u, float& v) {
    float ox = (w[0] * o.x + w[1] * o.y + w[2] * o.z) + w[3];
    float oy = (w[4] * o.x + w[5] * o.y + w[6] * o.z) + w[7];
    float oz = (w[8] * o.x + w[9] * o.y + w[10] * o.z) + w[11];
    float dx = w[0] * d.x + w[1] * d.y + w[2] * d.z;
    float dy = w[4] * d.x + w[5] * d.y + w[6] * d.z;
    float dz = w[8] * d.x + w[9] * d.y + w[10] * d.z;
    t = -oz / dz;
    u = ox + dx * t;
    v = oy + dy * t;
    return u > 0.0f && v > 0.0f && u + v < 1.0f;
}

// reference src/RayTracer.cpp:174-290
inline Hit raycast(const pto_scene& s, bool use_alpha, V3 orig, V3 dir, bool occlusion, Counters& ctr) {
    float tmin = 1.0f, umin = 0.0f, vmin = 0.0f;
    int imin = -1;
    bool hit = false;
    V3 inv = V3{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    int sign[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    int top = 0, cur = 0;
    int stack[64];
    ++ctr.rays;
    if (s.num_nodes <= 0) return Hit{-1, FLT_MAX, 0, 0, V3{}, V3{}, V3{}};
    while (true) {
        const pto_flatnode& node = s.nodes[cur];
        ++ctr.node_visits;
        if (box_intersect(node, orig, inv, sign, tmin)) {
            if (node.leaf) {
                for (int i = node.startIdx; i <= node.endIdx; ++i) {
                    float t, u, v;
                    int ti = s.list[i];
                    ++ctr.tri_tests;
                    if (intersect_woop(s.woop + (size_t)ti * 12, orig, dir, t, u, v)) {
                        if (use_alpha) {  // :204-225 alpha-texture cutout
                            const pto_material& m = s.materials[s.mat_id[ti]];
                            if (m.tex_alpha >= 0) {
                                const float* tc = s.uv + (size_t)ti * 6;
                                V2 uv = (1 - u - v) * ld2(tc) + u * ld2(tc + 2) + v * ld2(tc + 4);
                                const pto_texture& tex = s.textures[m.tex_alpha];
                                int x, y;
                                texel_coords(uv, tex, x, y);
                                if (texel(tex, x, y).y < 0.5f) continue;
                            }
                        }
                        if (t > 0.0f && t < tmin) { imin = ti; tmin = t; umin = u; vmin = v; hit = true; }
                        if (occlusion && hit)
                            return Hit{imin, tmin, umin, vmin, orig + tmin * dir, orig, dir};
                    }
                }
                if (top == 0) break;
                cur = stack[--top];
            } else {
                if (sign[node.axis]) { stack[top++] = cur + 1; cur = node.rightChild; }
                else { stack[top++] = node.rightChild; cur = cur + 1; }
            }
        } else {
            if (top == 0) break;
            cur = stack[--top];
        }
    }
    if (hit) return Hit{imin, tmin, umin, vmin, orig + tmin * dir, orig, dir};
    return Hit{-1, FLT_MAX, 0, 0, V3{}, V3{}, V3{}};
}

// reference src/util.cpp:22-47 (Shirley-Chiu)
template <class Mth> inline void map_to_disk(float a, float b, float& x, float& y) {
    const float PI = 3.14159265358979323846f;
    float phi, r;
    if (a > -b) {
        if (a > b) { r = a; phi = (PI / 4.0f) * (b / a); }
        else { r = b; phi = (PI / 4.0f) * (2 - (a / b)); }
    } else {
        if (a < b) { r = -a; phi = (PI / 4.0f) * (4.0f + (b / a)); }
        else {
            r = -b;
            if (b != 0) phi = (PI / 4.0f) * (6.0f - (a / b));
            else phi = 0;
        }
    }
    x = r * Mth::cos(phi);
    y = r * Mth::sin(phi);
}

// reference src/RayTracer.cpp:33-55
inline M3 form_basis(V3 n) {
    V3 q = n;
    int minV = 0;
    if (std::fabs(q.y) < std::fabs(q[minV])) minV = 1;
    if (std::fabs(q.z) < std::fabs(q[minV])) minV = 2;
    q[minV] = 1.0f;
    V3 t = normalize(cross(q, n));
    V3 b = normalize(cross(n, t));
    M3 R;
    R.setCol(0, t); R.setCol(1, b); R.setCol(2, n);
    return R;
}

inline int sobol_bounce_dim(int bounce) { return bounce > 254 ? 254 : bounce; }  // sobol.h:40 asserts dim < 1024

// reference src/PathTraceRenderer.cpp:57-73 (R is taken BY VALUE there: the caller's stream does not advance)
template <class Mth> inline V3 draw_cosine_weighted_direction(int base, int bounce, const Rng& R) {
    float x, y;
    int dim = 2 + 4 * sobol_bounce_dim(bounce);
    int rnd = R.peekS32(1, 1234);
    float a = 2.0f * sobol_sample((unsigned long long)(long long)(base + rnd), dim) - 1.0f;
    float b = 2.0f * sobol_sample((unsigned long long)(long long)(base + rnd), dim + 1) - 1.0f;
    map_to_disk<Mth>(a, b, x, y);
    float z = std::sqrt(1 - x * x - y * y);
    return V3{x, y, z};
}

inline V3 light_col(const pto_light& l, int c) { return V3{l.xform[c * 4], l.xform[c * 4 + 1], l.xform[c * 4 + 2]}; }
inline V3 light_position(const pto_light& l) { return light_col(l, 3); }  // src/AreaLight.hpp:31
inline V3 light_normal(const pto_light& l) { return -light_col(l, 2); }   // src/AreaLight.hpp:30
inline V3 light_emission(const pto_light& l) { return ld3(l.E); }

// reference src/AreaLight.cpp:69-85
inline void light_sample(const pto_light& l, float& pdf, V3& p, int base, int dimension, Rng& rnd) {
    V3 tangent = light_col(l, 0), bitangent = light_col(l, 1);
    int dim = 4 + 4 * sobol_bounce_dim(dimension);
    int r = rnd.getS32(1, 10000);
    float x = (2.0f * sobol_sample((unsigned long long)(long long)(base + r), dim) - 1.0f) * l.size[0];
    float y = (2.0f * sobol_sample((unsigned long long)(long long)(base + r), dim + 1) - 1.0f) * l.size[1];
    p = light_position(l) + tangent * x + bitangent * y;
    pdf = 1.0f / (4.0f * l.size[0] * l.size[1]);
}

// reference src/AreaLight.cpp:87-108
template <class Mth> inline void light_sample_disk(const pto_light& l, float& pdf, V3& p, int base, int dimension, Rng& rnd) {
    const float PI = 3.14159265358979323846f;
    V3 tangent = light_col(l, 0), bitangent = light_col(l, 1);
    int dim = 4 + 4 * sobol_bounce_dim(dimension);
    int r = rnd.getS32(1, 1000);
    float x, y;
    float a = (2.0f * sobol_sample((unsigned long long)(long long)(base + r), dim) - 1.0f) * l.size[0];
    float b = (2.0f * sobol_sample((unsigned long long)(long long)(base + r), dim + 1) - 1.0f) * l.size[1];
    map_to_disk<Mth>(a, b, x, y);
    p = light_position(l) + tangent * x + bitangent * y;
    pdf = 1.0f / (PI * l.size[0] * l.size[1]);
}

// reference src/PathTraceRenderer.cpp:183-210
inline int get_light_to_sample(V3 pos, const pto_light* lights, int n, Rng& R) {
    if (n <= 1) return 0;
    std::vector<float> values(n);
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        values[i] = lights[i].power / length(pos - light_position(lights[i]));
        sum += values[i];
    }
    for (int i = 0; i < n; i++) values[i] /= sum;
    float r = R.getF32(0.0f, 1.0f);
    float value = 0.0f;
    for (int i = 0; i < n; i++) {
        value += values[i];
        if (r <= value) return i;
    }
    return n - 1;
}

// reference src/PathTraceRenderer.cpp:75-181
template <class Mth>
inline void get_texture_parameters(const pto_scene& s, const pto_params& prm, const Hit& hit, V3& diffuse, V3& n, V3& specular) {
    const pto_material& mat = s.materials[s.mat_id[hit.tri]];
    const float* tc = s.uv + (size_t)hit.tri * 6;
    V2 uv = (1 - hit.u - hit.v) * ld2(tc) + hit.u * ld2(tc + 2) + hit.v * ld2(tc + 4);

    diffuse = V3{mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]};
    if (mat.tex_diffuse >= 0) {
        const pto_texture& tex = s.textures[mat.tex_diffuse];
        if (prm.bilinear) {
            float fx = uv.x - std::floor(uv.x);
            float fy = uv.y - std::floor(uv.y);
            float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
            float w1 = fx1 * fy1, w2 = fx * fy1, w3 = fx1 * fy, w4 = fx * fy;
            int x1 = (int)fmin2(fmax2(fx * tex.width - 0.5f, 0.0f), tex.width - 2.0f);
            int y1 = (int)fmin2(fmax2(fy * tex.height - 0.5f, 0.0f), tex.height - 2.0f);
            int x2 = x1 + 1, y2 = y1 + 1;
            V4 d1 = texel(tex, x1, y1), d2 = texel(tex, x2, y1), d3 = texel(tex, x1, y2), d4 = texel(tex, x2, y2);
            diffuse = V3{d1.x * w1 + d2.x * w2 + d3.x * w3 + d4.x * w4,
                         d1.y * w1 + d2.y * w2 + d3.y * w3 + d4.y * w4,
                         d1.z * w1 + d2.z * w2 + d3.z * w3 + d4.z * w4};
        } else {
            int x, y;
            texel_coords(uv, tex, x, y);
            V4 d = texel(tex, x, y);
            diffuse = V3{d.x, d.y, d.z};
        }
        diffuse = V3{Mth::pow(diffuse.x, 2.2f), Mth::pow(diffuse.y, 2.2f), Mth::pow(diffuse.z, 2.2f)};
    }
    const float* nn = s.nrm + (size_t)hit.tri * 9;
    n = normalize((1 - hit.u - hit.v) * ld3(nn) + hit.u * ld3(nn + 3) + hit.v * ld3(nn + 6));

    if (mat.tex_normal >= 0 && prm.normal_mapped) {
        const pto_texture& tex = s.textures[mat.tex_normal];
        int x, y;
        texel_coords(uv, tex, x, y);
        V4 tv = texel(tex, x, y);
        V3 texture_normal = V3{2.0f * tv.x - 1.0f, 2.0f * tv.y - 1.0f, 2.0f * tv.z - 1.0f};
        const float* pp = s.pos + (size_t)hit.tri * 9;
        V3 v0 = ld3(pp), v1 = ld3(pp + 3), v2 = ld3(pp + 6);
        V2 uv0 = ld2(tc), uv1 = ld2(tc + 2), uv2 = ld2(tc + 4);
        V3 deltaPos1 = v1 - v0, deltaPos2 = v2 - v0;
        V2 deltaUV1 = uv1 - uv0, deltaUV2 = uv2 - uv0;
        float r = 1.0f / (deltaUV1.x * deltaUV2.y - deltaUV1.y * deltaUV2.x);
        V3 tangent = normalize((deltaPos1 * deltaUV2.y - deltaPos2 * deltaUV1.y) * r);
        V3 bitangent = normalize((deltaPos2 * deltaUV1.x - deltaPos1 * deltaUV2.x) * r);
        M3 TBN;
        TBN.setCol(0, tangent); TBN.setCol(1, bitangent); TBN.setCol(2, n);
        n = normalize(TBN * texture_normal);
    }
    specular = V3{mat.specular[0], mat.specular[1], mat.specular[2]};
    if (mat.tex_specular >= 0) {
        const pto_texture& tex = s.textures[mat.tex_specular];
        int x, y;
        texel_coords(uv, tex, x, y);
        V4 d = texel(tex, x, y);
        specular = V3{d.x, d.y, d.z};
    }
}

struct PrimaryRay { V3 Ro, Rd; };

// reference src/PathTraceRenderer.cpp:244-292 (also :561-606 for the Whitted variant: no max(0.01, .))
template <class Mth>
inline PrimaryRay camera_ray(const pto_params& prm, const pto_camera& cam, float x, float y, Rng& R, bool whitted_variant) {
    V4 Roh = mul44(cam.invP, V4{x, y, 0.0f, 1.0f});
    float iw0 = 1.0f / Roh.w;
    V3 Ro = V3{Roh.x * iw0, Roh.y * iw0, Roh.z * iw0};
    V4 Rdh = mul44(cam.invP, V4{x, y, 1.0f, 1.0f});
    float iw1 = 1.0f / Rdh.w;
    V3 Rd = V3{Rdh.x * iw1, Rdh.y * iw1, Rdh.z * iw1};
    Rd = Rd - Ro;
    if (prm.depth_of_field) {
        V3 Rn = normalize(Rd);
        V3 Pn = ld3(cam.forward);
        float Pd = whitted_variant ? prm.focal_distance / dot(Rn, Pn) : prm.focal_distance / fmax2(0.01f, dot(Rn, Pn));
        V3 Rf = Ro + Pd * Rn;
        float xr, yr;
        float a = R.getF32(-1.0f, 1.0f);
        float b = R.getF32(-1.0f, 1.0f);
        map_to_disk<Mth>(a, b, xr, yr);
        V3 up = ld3(cam.orient_col0), right = ld3(cam.orient_col1);
        Ro += prm.aperture * (yr * up + xr * right);
        Rd = cam.far_ * normalize(Rf - Ro);
    }
    return PrimaryRay{Ro, Rd};
}

// The direct-light block that the reference writes out three times
// (src/PathTraceRenderer.cpp:346-376, :416-450, :500-534).  mode 0 = first hit
// (no throughput), 1 = fixed bounce (x throughput), 2 = roulette (x throughput x inv_prob).
template <class Mth>
inline void shade_unoccluded(int mode, V3& N, V3 V, V3 D, V3 L, V3 Nl, float pdf, V3 diffuse, V3 specular, float gloss,
                             float normalization, V3 throughput, V3 emission, float inv_prob, V3& Ei) {
    const float invPi = 0.31830988618379f;
    if (dot(N, V) > 0.0f) N *= -1.0f;
    float cos_theta = fmax2(0.0f, dot(N, D));
    float cos_theta_l = fmax2(0.0f, dot(Nl, -D));
    float d = length(L);
    float inv_d = fmin2(1.0f / (d * d), 100000.0f);
    float inv_pdf = fmin2(1.0f / pdf, 100000.0f);
    V3 H = normalize(D - V);
    V3 color = diffuse;
    color += specular * Mth::pow(fmax2(0.0f, dot(N, H)), gloss) * normalization;
    if (mode == 0)
        Ei += color * invPi * emission * cos_theta * cos_theta_l * inv_d * inv_pdf;
    else if (mode == 1)
        Ei += color * invPi * throughput * emission * cos_theta * cos_theta_l * inv_d * inv_pdf;
    else
        Ei += color * invPi * throughput * emission * cos_theta * cos_theta_l * inv_d * inv_pdf * inv_prob;
}

// reference src/PathTraceRenderer.cpp:236-551
template <class Mth>
inline V3 trace_path(const pto_scene& s, const pto_params& prm, const pto_camera& cam, const pto_light* lights, int nl,
                     float x, float y, int samplerBase, Rng& R, Counters& ctr) {
    const float PI = 3.14159265358979323846f;
    PrimaryRay pr = camera_ray<Mth>(prm, cam, x, y, R, false);
    Hit result = raycast(s, prm.use_alpha != 0, pr.Ro, pr.Rd, false, ctr);
    V3 Ei = v3(0.0f);
    if (result.tri >= 0) {
        float gloss = s.materials[s.mat_id[result.tri]].glossiness;
        float normalization = (gloss + 2.0f) / (4.0f * PI * (2.0f - Mth::pow(2.0f, -gloss / 2.0f)));
        int bounce = 0;
        int bounces = prm.bounces < 0 ? -prm.bounces : prm.bounces;
        V3 throughput = v3(1.0f);
        V3 O = result.point;
        V3 V = normalize(result.dir);
        V3 diffuse, N, specular;
        get_texture_parameters<Mth>(s, prm, result, diffuse, N, specular);
        int li = get_light_to_sample(O, lights, nl, R);
        float pdf;
        V3 Pl;
        light_sample(lights[li], pdf, Pl, samplerBase, bounce, R);
        V3 L = Pl - O;
        V3 D = normalize(L);
        V3 emission = light_emission(lights[li]) * prm.intensity;  // computed once, reused for every bounce (:341)
        if (!raycast(s, prm.use_alpha != 0, O + 0.0001f * N, L, true, ctr))
            shade_unoccluded<Mth>(0, N, V, D, L, light_normal(lights[li]), pdf, diffuse, specular, gloss, normalization,
                                  throughput, emission, 1.0f, Ei);

        auto bounce_body = [&](int mode, float inv_prob) -> bool {  // :391-451 and :473-534
            V = normalize(form_basis(N) * draw_cosine_weighted_direction<Mth>(samplerBase, bounce, R));
            throughput *= diffuse;
            result = raycast(s, prm.use_alpha != 0, O + 0.0001f * N, cam.far_ * V, false, ctr);
            if (!result) return false;
            O = result.point;
            get_texture_parameters<Mth>(s, prm, result, diffuse, N, specular);
            li = get_light_to_sample(O, lights, nl, R);
            light_sample(lights[li], pdf, Pl, samplerBase, bounce, R);
            L = Pl - O;
            D = normalize(L);
            if (!raycast(s, prm.use_alpha != 0, O + 0.0001f * N, L, true, ctr))
                shade_unoccluded<Mth>(mode, N, V, D, L, light_normal(lights[li]), pdf, diffuse, specular, gloss,
                                      normalization, throughput, emission, inv_prob, Ei);
            return true;
        };
        while (bounce < bounces) {
            if (!bounce_body(1, 1.0f)) break;
            bounce += 1;
        }
        if (prm.bounces < 0) {  // Russian roulette (:466-548)
            float terminate = R.getF32(0.0f, 1.0f);
            float inv_prob = 1.0f / prm.termination_prob;
            while (terminate > prm.termination_prob) {
                if (!bounce_body(2, inv_prob)) break;
                terminate = R.getF32(0.0f, 1.0f);
                bounce += 1;
            }
        }
    }
    return Ei;
}

// reference src/PathTraceRenderer.cpp:553-660
template <class Mth>
inline V3 trace_whitted(const pto_scene& s, const pto_params& prm, const pto_camera& cam, const pto_light* lights, int nl,
                        float x, float y, int samplerBase, Rng& R, Counters& ctr) {
    const float PI = 3.14159265358979323846f;
    PrimaryRay pr = camera_ray<Mth>(prm, cam, x, y, R, true);
    Hit result = raycast(s, prm.use_alpha != 0, pr.Ro, pr.Rd, false, ctr);
    V3 color = v3(0.0f);
    if (result.tri >= 0) {
        V3 O = result.point;
        V3 V = normalize(result.dir);
        V3 diffuse, N, specular;
        get_texture_parameters<Mth>(s, prm, result, diffuse, N, specular);
        float gloss = s.materials[s.mat_id[result.tri]].glossiness;
        float normalization = (gloss + 2.0f) / (4.0f * PI * (2.0f - Mth::pow(2.0f, -gloss / 2.0f)));
        int li = get_light_to_sample(O, lights, nl, R);
        float pdf, shadow = 0.0f, d;
        V3 Pl, L, D;
        for (int i = 0; i < prm.num_light_rays; i++) {
            light_sample_disk<Mth>(lights[li], pdf, Pl, samplerBase, 0, R);
            L = Pl - O;
            D = normalize(L);
            if (!raycast(s, prm.use_alpha != 0, O + 0.0001f * D, L, true, ctr)) shadow += 1.0f;
        }
        L = light_position(lights[li]) - O;
        D = normalize(L);
        d = length(L);
        float attenuation = prm.intensity / (prm.attenuation * Mth::pow(d, 2.0f) + prm.attenuation * d + 0.5f);
        V3 H = normalize(D - V);
        color = diffuse * fmax2(0.0f, dot(N, D));
        color += specular * Mth::pow(fmax2(0.0f, dot(N, H)), gloss) * normalization;
        color *= attenuation;
        color *= shadow / prm.num_light_rays;
    }
    return color;
}

// reference src/PathTraceRenderer.cpp:52-55
template <class Mth> inline float gauss(float x, float y) {
    const float PI = 3.14159265358979323846f;
    float sigma = 1.0f;
    return 1.0f / (std::sqrt(2.0f * PI) * sigma) * Mth::exp(-1.0f * (x * x + y * y) / (2.0f * sigma * sigma));
}

// reference src/PathTraceRenderer.cpp:36-49
inline V4 saturate(V4 c, float amount) {
    const float Pr = 0.299f, Pg = 0.587f, Pb = 0.114f;
    float P = std::sqrt(c.x * c.x * Pr + c.y * c.y * Pg + c.z * c.z * Pb);
    c.x = P + (c.x - P) * amount;
    c.y = P + (c.y - P) * amount;
    c.z = P + (c.z - P) * amount;
    auto cl = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
    return V4{cl(c.x), cl(c.y), cl(c.z), cl(c.w)};
}

struct Block { int x, y, w, h; };

// reference src/PathTraceRenderer.cpp:813-841
inline std::vector<Block> make_blocks(int width, int height) {
    std::vector<Block> blocks;
    const int bs = 32;
    int bcx = (width + bs - 1) / bs, bcy = (height + bs - 1) / bs;
    for (int y = 0; y < bcy; ++y) {
        int ey = (y * bs + bs < height) ? y * bs + bs : height;
        for (int x = 0; x < bcx; ++x) {
            int ex = (x * bs + bs < width) ? x * bs + bs : width;
            blocks.push_back(Block{bs * x, bs * y, ex - bs * x, ey - bs * y});
        }
    }
    return blocks;
}

// Sobol sequence base of a block (reference :689-694).  The reference seeds Random with
// t.idx + a process-global atomic counter; this build uses (block, pass_seed) and fixes the
// (unspecified in C++) evaluation order of the two draws: multiplier first, then offset.
inline int block_sobol_base(int idx, uint32_t pass_seed) {
    Rng Rb((uint32_t)idx, pass_seed, 0xB10Cu, 0u);
    int mul = Rb.getS32(1, 9);
    int add = Rb.getS32(1, 12345);
    return idx * mul + add;
}

// reference src/PathTraceRenderer.cpp:663-790, one block.  The reference threads ONE Random
// through all samples of a block; this build gives every sample its own stream keyed by
// (block, pass_seed, i, k) (documented deviation, SURVEY.md 7 "Sequential RNG").
// A Sobol-block sample whose pixel lies outside its own block (the position block.x + block.w * sobol() rounds up to the block's far
// edge: sobol() == 1.0f exactly, SURVEY.md R8, or merely close to 1 when block.x is in the thousands -- some 30,000 samples of a 4K frame at
// 1024 spp): the reference's worker thread adds it to the neighbouring block's pixel while another thread may be working on that
// block (a race).  This build defers such samples to the end of the pass and adds them per pixel in (source block, sample) order
// (device: ptgpu_apply_spills_sorted); with image tiles sharded over ranks they all travel with the tiles and every rank adds every
// rank's after the exchange, in that same order (device: ptgpu_tiles_pack_dev / _unpack_dev).  `spills` collects them in sample order.
struct Spill { float r, g, b, w; int pixel, src_block, sample; };

template <class Mth>
inline void path_trace_block(const pto_scene& s, const pto_params& prm, const pto_camera& cam, const pto_light* lights, int nl,
                             int idx, const Block& block, float* accum, Counters& ctr, std::vector<Spill>* spills = nullptr) {
    int base = block_sobol_base(idx, prm.pass_seed);
    int height = prm.height, width = prm.width;
    float div_x = 1.0f / (float)width, div_y = 1.0f / (float)height;
    float filt_scale = 2.0f / prm.filter_width;
    if (prm.sobol_block) {
        float scaleToAaRays = 1.0f / (float)prm.num_aa_rays;
        for (int i = 0; i < block.w * block.h * prm.num_aa_rays; ++i) {
            float px = block.x + block.w * sobol_sample((unsigned long long)(long long)(base + i), 0);
            float py = block.y + block.h * sobol_sample((unsigned long long)(long long)(base + i), 1);
            float x = px * div_x * 2.0f - 1.0f;
            float y = py * div_y * -2.0f + 1.0f;
            Rng R((uint32_t)idx, prm.pass_seed, (uint32_t)i, 0u);
            V3 color = prm.whitted ? trace_whitted<Mth>(s, prm, cam, lights, nl, x, y, base + i, R, ctr)
                                   : trace_path<Mth>(s, prm, cam, lights, nl, x, y, base + i, R, ctr);
            int pixel_x = (int)std::floor(px); if (pixel_x > width - 1) pixel_x = width - 1;
            int pixel_y = (int)std::floor(py); if (pixel_y > height - 1) pixel_y = height - 1;
            if (spills && (pixel_x < block.x || pixel_x >= block.x + block.w || pixel_y < block.y || pixel_y >= block.y + block.h)) {
                spills->push_back(Spill{color.x * scaleToAaRays, color.y * scaleToAaRays, color.z * scaleToAaRays, 1.0f * scaleToAaRays,
                                        pixel_y * width + pixel_x, idx, i});
                continue;
            }
            float* prev = accum + ((size_t)pixel_y * width + pixel_x) * 4;
            prev[0] += color.x * scaleToAaRays;
            prev[1] += color.y * scaleToAaRays;
            prev[2] += color.z * scaleToAaRays;
            prev[3] += 1.0f * scaleToAaRays;
        }
    } else {
        for (int i = 0; i < block.w * block.h; ++i) {
            int pixel_x = block.x + (i % block.w);
            int pixel_y = block.y + (i / block.w);
            V4 color_AA = V4{0, 0, 0, 0};
            for (int k = 0; k < prm.num_aa_rays; k++) {
                float a = prm.filter_width * (sobol_sample((unsigned long long)(long long)(base + i + k), 0) - 0.5f) + 0.5f;
                float b = prm.filter_width * (sobol_sample((unsigned long long)(long long)(base + i + k), 1) - 0.5f) + 0.5f;
                float px = (float)pixel_x + a;
                float py = (float)pixel_y + b;
                if (px < 0.0f) px *= -1.0f;
                if (py < 0.0f) py *= -1.0f;
                if (px > width) px -= a;
                if (py > height) py -= b;
                float x = px * div_x * 2.0f - 1.0f;
                float y = py * div_y * -2.0f + 1.0f;
                Rng R((uint32_t)idx, prm.pass_seed, (uint32_t)i, (uint32_t)k);
                V3 color = prm.whitted ? trace_whitted<Mth>(s, prm, cam, lights, nl, x, y, base + i, R, ctr)
                                       : trace_path<Mth>(s, prm, cam, lights, nl, x, y, base + i, R, ctr);
                float g = gauss<Mth>(filt_scale * (a - 0.5f), filt_scale * (b - 0.5f));
                color_AA.x += g * color.x;
                color_AA.y += g * color.y;
                color_AA.z += g * color.z;
                color_AA.w += g * 1.0f;
            }
            color_AA = V4{color_AA.x / color_AA.w, color_AA.y / color_AA.w, color_AA.z / color_AA.w, color_AA.w / color_AA.w};
            color_AA = saturate(color_AA, prm.attenuation);
            float* prev = accum + ((size_t)pixel_y * width + pixel_x) * 4;
            prev[0] += color_AA.x; prev[1] += color_AA.y; prev[2] += color_AA.z; prev[3] += color_AA.w;
        }
    }
}

}  // namespace pto
