// ORACLE -- TEST INFRASTRUCTURE ONLY (see pt_math.h header note).
// C ABI over pt_core.h / pt_bvh.cpp; multi-threaded over rays / 32x32 blocks for the
// cpu_baseline leg of bench.py (same block tiling as reference src/PathTraceRenderer.cpp:816-852).
#include <atomic>
#include <cstring>
#include <thread>

#include "pt_core.h"

namespace pto {
const uint32_t g_sobol_matrices[1024 * 52] = {
#include "_gen/sobol_matrices.inc"
};

template <class F> static void parallel_for(int64_t n, int64_t chunk, int nthreads, F&& fn) {
    if (nthreads <= 1 || n <= chunk) {
        fn(0, n, 0);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            for (;;) {
                int64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                fn(b, (b + chunk < n) ? b + chunk : n, t);
            }
        });
    for (auto& x : th) x.join();
}
static void add_counters(pto_counters* out, const std::vector<Counters>& c) {
    if (!out) return;
    for (const auto& k : c) { out->rays += k.rays; out->node_visits += k.node_visits; out->tri_tests += k.tri_tests; }
}
}  // namespace pto

using namespace pto;

extern "C" {

float pto_sobol_sample(uint64_t index, uint32_t dimension, uint32_t scramble) { return sobol_sample(index, dimension, scramble); }
const uint32_t* pto_sobol_matrices(void) { return g_sobol_matrices; }

void pto_map_to_disk(float a, float b, float* x, float* y, int libm) {
    if (libm) map_to_disk<LibmMath>(a, b, *x, *y); else map_to_disk<PtmMath>(a, b, *x, *y);
}
void pto_form_basis(const float n[3], float out_cols[9]) {
    M3 R = form_basis(ld3(n));
    for (int i = 0; i < 9; ++i) out_cols[i] = R.m[i];
}
// tri_data ctor, reference src/RTTriangle.hpp:14-20 with normal() from :45-47
void pto_woop_from_vertices(const float v0[3], const float v1[3], const float v2[3], float out12[12]) {
    V3 a = ld3(v0), b = ld3(v1), c = ld3(v2);
    V3 nrm = normalize(cross(b - a, c - a));
    M3 M;
    M.setCol(0, b - a); M.setCol(1, c - a); M.setCol(2, nrm);
    M = inverse(M);
    M3 negM;
    for (int i = 0; i < 9; ++i) negM.m[i] = -M.m[i];
    V3 N = negM * a;
    for (int r = 0; r < 3; ++r) {
        out12[r * 4 + 0] = M.at(r, 0); out12[r * 4 + 1] = M.at(r, 1); out12[r * 4 + 2] = M.at(r, 2);
        out12[r * 4 + 3] = N[r];
    }
}
int pto_intersect_woop(const float w[12], const float o[3], const float d[3], float* t, float* u, float* v) {
    return intersect_woop(w, ld3(o), ld3(d), *t, *u, *v) ? 1 : 0;
}
int pto_box_intersect(const pto_flatnode* n, const float orig[3], const float inv_dir[3], const int sign[3], float t1) {
    return box_intersect(*n, ld3(orig), ld3(inv_dir), sign, t1) ? 1 : 0;
}
// reference src/rtIntersect.inl:215-259 (two-sided Moller-Trumbore; dead code in the reference)
int pto_intersect_triangle1(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3],
                            float* t, float* u, float* v) {
    const float EPSILON = 0.000000001f;
    V3 orig = ld3(o), dir = ld3(d), a = ld3(v0);
    V3 edge1 = ld3(v1) - a, edge2 = ld3(v2) - a;
    V3 pvec = cross(dir, edge2);
    float det = dot(edge1, pvec);
    V3 tvec, qvec;
    if (det > EPSILON) {
        tvec = orig - a;
        *u = dot(tvec, pvec);
        if (*u < 0.0 || *u > det) return 0;
        qvec = cross(tvec, edge1);
        *v = dot(dir, qvec);
        if (*v < 0.0 || *u + *v > det) return 0;
    } else if (det < -EPSILON) {
        tvec = orig - a;
        *u = dot(tvec, pvec);
        if (*u > 0.0 || *u < det) return 0;
        qvec = cross(tvec, edge1);
        *v = dot(dir, qvec);
        if (*v > 0.0 || *u + *v < det) return 0;
    } else
        return 0;
    float inv_det = 1.f / det;
    *t = dot(edge2, qvec) * inv_det;
    *u *= inv_det;
    *v *= inv_det;
    return 1;
}
float pto_gauss(float x, float y, int libm) { return libm ? gauss<LibmMath>(x, y) : gauss<PtmMath>(x, y); }
void pto_saturate(const float in[4], float amount, float out[4]) {
    V4 r = saturate(V4{in[0], in[1], in[2], in[3]}, amount);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void pto_math_probe(int fn, const float* x, const float* y, float* out, int n, int libm) {
    for (int i = 0; i < n; ++i) {
        float a = x[i], b = y ? y[i] : 0.0f;
        switch (fn) {
            case 0: out[i] = libm ? LibmMath::sin(a) : PtmMath::sin(a); break;
            case 1: out[i] = libm ? LibmMath::cos(a) : PtmMath::cos(a); break;
            case 2: out[i] = libm ? LibmMath::exp(a) : PtmMath::exp(a); break;
            default: out[i] = libm ? LibmMath::pow(a, b) : PtmMath::pow(a, b); break;
        }
    }
}
uint32_t pto_rng_u32(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t draw) { return Rng(k0, k1, k2, k3).at(draw); }
void pto_light_sample(const pto_light* l, int base, int bounce, uint32_t k[4], float* pdf, float p[3]) {
    Rng R(k[0], k[1], k[2], k[3]);
    V3 P;
    light_sample(*l, *pdf, P, base, bounce, R);
    p[0] = P.x; p[1] = P.y; p[2] = P.z;
}

void pto_raycast_batch(const pto_scene* s, int use_alpha, const float* orig, const float* dir, int64_t n, int occlusion,
                       int32_t* out_id, float* out_t, float* out_u, float* out_v, pto_counters* ctr, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> c(nthreads);
    parallel_for(n, 4096, nthreads, [&](int64_t b, int64_t e, int t) {
        for (int64_t i = b; i < e; ++i) {
            Hit h = raycast(*s, use_alpha != 0, ld3(orig + i * 3), ld3(dir + i * 3), occlusion != 0, c[t]);
            out_id[i] = h.tri;
            if (out_t) out_t[i] = h.t;
            if (out_u) out_u[i] = h.u;
            if (out_v) out_v[i] = h.v;
        }
    });
    add_counters(ctr, c);
}

// Pixel-centre primary rays, reference src/PathTraceRenderer.cpp:244-266 with (px,py) = pixel + 0.5 mapped
// to clip space as :716-717 does.
void pto_gen_primary(const pto_camera* cam, int width, int height, float* orig, float* dir) {
    pto_params prm{};
    prm.depth_of_field = 0;
    float div_x = 1.0f / (float)width, div_y = 1.0f / (float)height;
    Rng R(0, 0, 0, 0);
    for (int py = 0; py < height; ++py)
        for (int px = 0; px < width; ++px) {
            float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
            float x = fx * div_x * 2.0f - 1.0f;
            float y = fy * div_y * -2.0f + 1.0f;
            PrimaryRay r = camera_ray<PtmMath>(prm, *cam, x, y, R, false);
            size_t i = (size_t)py * width + px;
            orig[i * 3] = r.Ro.x; orig[i * 3 + 1] = r.Ro.y; orig[i * 3 + 2] = r.Ro.z;
            dir[i * 3] = r.Rd.x; dir[i * 3 + 1] = r.Rd.y; dir[i * 3 + 2] = r.Rd.z;
        }
}

// Shadow ray of SURVEY.md 8d config C2: from the hit point offset along the shading normal towards the
// Sobol light point with index = ray index, dims 4,5 (AreaLight.cpp:75-81 without the RNG offset),
// origin/direction exactly as reference src/PathTraceRenderer.cpp:338-344.
void pto_gen_shadow(const pto_scene* s, const pto_light* l, int64_t n, const float* orig, const float* dir,
                    const int32_t* hit_id, const float* hit_t, const float* hit_u, const float* hit_v,
                    float* sh_orig, float* sh_dir, int32_t* sh_valid) {
    for (int64_t i = 0; i < n; ++i) {
        int ok = hit_id[i] >= 0;
        sh_valid[i] = ok;
        V3 so = v3(0.0f), sd = v3(0.0f);
        if (ok) {
            V3 O = ld3(orig + i * 3) + hit_t[i] * ld3(dir + i * 3);
            const float* nn = s->nrm + (size_t)hit_id[i] * 9;
            float u = hit_u[i], v = hit_v[i];
            V3 N = normalize((1 - u - v) * ld3(nn) + u * ld3(nn + 3) + v * ld3(nn + 6));
            float x = (2.0f * sobol_sample((unsigned long long)i, 4) - 1.0f) * l->size[0];
            float y = (2.0f * sobol_sample((unsigned long long)i, 5) - 1.0f) * l->size[1];
            V3 Pl = light_position(*l) + light_col(*l, 0) * x + light_col(*l, 1) * y;
            so = O + 0.0001f * N;
            sd = Pl - O;
        }
        sh_orig[i * 3] = so.x; sh_orig[i * 3 + 1] = so.y; sh_orig[i * 3 + 2] = so.z;
        sh_dir[i * 3] = sd.x; sh_dir[i * 3 + 1] = sd.y; sh_dir[i * 3 + 2] = sd.z;
    }
}

// One pass over blocks block_begin, block_begin + block_stride, ... < block_end.  Out-of-block samples are applied after every block is
// done, in (block, sample) order.  foreign_out == NULL: all of them to accum.  foreign_out given (8 floats per entry: r g b w, then pixel,
// source block, sample as int bits, 1 pad -- the device's ForeignSpill): with block_stride == 1 those whose pixel lies in one of this
// call's blocks go to accum and the others to foreign_out; with block_stride > 1 (tiles sharded over ranks) ALL go to foreign_out, for
// every rank to add every rank's after the exchange, per pixel in (source block, sample) order (pathtracer_amd/multigpu.TileGather).
void pto_render_pass2(const pto_scene* s, const pto_params* p, const pto_camera* cam, const pto_light* lights, int nl,
                      float* accum, pto_counters* ctr, int nthreads, int libm, int block_begin, int block_end, int block_stride,
                      float* foreign_out, int foreign_cap, int* foreign_count) {
    std::vector<Block> blocks = make_blocks(p->width, p->height);
    if (block_end < 0 || block_end > (int)blocks.size()) block_end = (int)blocks.size();
    if (block_begin < 0) block_begin = 0;
    if (block_stride < 1) block_stride = 1;
    if (nthreads < 1) nthreads = 1;
    const int count = block_end > block_begin ? (block_end - block_begin + block_stride - 1) / block_stride : 0;
    std::vector<Counters> c(nthreads);
    std::vector<std::vector<Spill>> spills(count);
    parallel_for(count, 1, nthreads, [&](int64_t b, int64_t e, int t) {
        for (int64_t i = b; i < e; ++i) {
            int idx = block_begin + (int)i * block_stride;
            if (libm) path_trace_block<LibmMath>(*s, *p, *cam, lights, nl, idx, blocks[idx], accum, c[t], &spills[i]);
            else path_trace_block<PtmMath>(*s, *p, *cam, lights, nl, idx, blocks[idx], accum, c[t], &spills[i]);
        }
    });
    const int blocks_x = (p->width + 31) / 32;
    int nf = 0;
    for (const auto& list : spills)
        for (const Spill& sp : list) {
            const int idx = ((sp.pixel / p->width) >> 5) * blocks_x + ((sp.pixel % p->width) >> 5);
            const bool own = idx >= block_begin && idx < block_end && (idx - block_begin) % block_stride == 0;
            if (!foreign_out || (own && block_stride == 1)) {
                float* px = accum + (size_t)sp.pixel * 4;
                px[0] += sp.r; px[1] += sp.g; px[2] += sp.b; px[3] += sp.w;
            } else if (nf < foreign_cap) {
                float* o = foreign_out + (size_t)nf * 8;
                o[0] = sp.r; o[1] = sp.g; o[2] = sp.b; o[3] = sp.w;
                std::memcpy(o + 4, &sp.pixel, 4);
                std::memcpy(o + 5, &sp.src_block, 4);
                std::memcpy(o + 6, &sp.sample, 4);
                o[7] = 0.0f;
                ++nf;
            }
        }
    if (foreign_count) *foreign_count = nf;
    add_counters(ctr, c);
}

void pto_render_pass(const pto_scene* s, const pto_params* p, const pto_camera* cam, const pto_light* lights, int nl,
                     float* accum, pto_counters* ctr, int nthreads, int libm, int block_begin, int block_end) {
    pto_render_pass2(s, p, cam, lights, nl, accum, ctr, nthreads, libm, block_begin, block_end, 1, nullptr, 0, nullptr);
}

// reference src/PathTraceRenderer.cpp:855-871
void pto_resolve(const float* accum, int width, int height, float* out, int libm) {
    for (int64_t i = 0; i < (int64_t)width * height; ++i) {
        V4 D = V4{accum[i * 4], accum[i * 4 + 1], accum[i * 4 + 2], accum[i * 4 + 3]};
        if (D.w != 0.0f) {
            float r = 1.0f / D.w;
            D = V4{D.x * r, D.y * r, D.z * r, D.w * r};
        }
        const float g = 1.0f / 2.2f;
        out[i * 4 + 0] = libm ? LibmMath::pow(D.x, g) : PtmMath::pow(D.x, g);
        out[i * 4 + 1] = libm ? LibmMath::pow(D.y, g) : PtmMath::pow(D.y, g);
        out[i * 4 + 2] = libm ? LibmMath::pow(D.z, g) : PtmMath::pow(D.z, g);
        out[i * 4 + 3] = D.w;
    }
}

void pto_trace_path(const pto_scene* s, const pto_params* p, const pto_camera* cam, const pto_light* lights, int nl,
                    float x, float y, int sampler_base, const uint32_t k[4], int libm, float out_rgb[3], pto_counters* ctr) {
    Rng R(k[0], k[1], k[2], k[3]);
    Counters c;
    V3 r;
    if (p->whitted)
        r = libm ? trace_whitted<LibmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base, R, c)
                 : trace_whitted<PtmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base, R, c);
    else
        r = libm ? trace_path<LibmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base, R, c)
                 : trace_path<PtmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base, R, c);
    out_rgb[0] = r.x; out_rgb[1] = r.y; out_rgb[2] = r.z;
    if (ctr) { ctr->rays += c.rays; ctr->node_visits += c.node_visits; ctr->tri_tests += c.tri_tests; }
}

// tracePath / traceWhitted for n samples given as tracePath's own arguments: clip-space (x, y), Sobol base, the random stream's
// (key, counter) state.  out [n][4]: radiance rgb + the counter after the call as bits (the device's ptgpu_trace_samples layout).
void pto_trace_samples(const pto_scene* s, const pto_params* p, const pto_camera* cam, const pto_light* lights, int nl, int64_t n,
                       const float* xy, const int32_t* sampler_base, const uint32_t* rng_state, int libm, float* out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    parallel_for(n, 256, nthreads, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            Rng R = Rng::from_state(rng_state[2 * i], rng_state[2 * i + 1]);
            Counters c;
            V3 r;
            const float x = xy[2 * i], y = xy[2 * i + 1];
            if (p->whitted)
                r = libm ? trace_whitted<LibmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base[i], R, c)
                         : trace_whitted<PtmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base[i], R, c);
            else
                r = libm ? trace_path<LibmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base[i], R, c)
                         : trace_path<PtmMath>(*s, *p, *cam, lights, nl, x, y, sampler_base[i], R, c);
            out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z;
            std::memcpy(out + 4 * i + 3, &R.ctr, 4);
        }
    });
}
uint32_t pto_rng_key(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) { return Rng(k0, k1, k2, k3).key; }
// getTextureParameters for n hits (triangle, u, v); out [n][9] = diffuse, normal, specular
void pto_texture_parameters(const pto_scene* s, const pto_params* p, int64_t n, const int32_t* tri, const float* u, const float* v, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        Hit h{tri[i], 0.0f, u[i], v[i], V3{}, V3{}, V3{}};
        V3 d, nn, sp;
        get_texture_parameters<PtmMath>(*s, *p, h, d, nn, sp);
        float* o = out + i * 9;
        o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = nn.x; o[4] = nn.y; o[5] = nn.z; o[6] = sp.x; o[7] = sp.y; o[8] = sp.z;
    }
}
// getLightToSample for n positions; out [n][2] = light index, stream counter after the call
void pto_light_to_sample(const pto_light* lights, int nl, int64_t n, const float* pos, const uint32_t* rng_state, int32_t* out) {
    for (int64_t i = 0; i < n; ++i) {
        Rng R = Rng::from_state(rng_state[2 * i], rng_state[2 * i + 1]);
        out[2 * i] = get_light_to_sample(ld3(pos + 3 * i), lights, nl, R);
        out[2 * i + 1] = (int32_t)R.ctr;
    }
}

}  // extern "C"
