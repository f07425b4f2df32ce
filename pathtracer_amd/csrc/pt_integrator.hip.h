// Persistent-wave path-trace integrator for gfx950.
//
// Restates PathTraceRenderer::tracePath (reference src/PathTraceRenderer.cpp:236-551) with the functions it calls:
// getTextureParameters (:75-181), getLightToSample (:183-210), drawCosineWeightedDirection (:57-73),
// AreaLight::sample (src/AreaLight.cpp:69-85), sobol::sample (src/sobol.h:39-49), mapToDisk (src/util.cpp:22-47),
// formBasis (src/RayTracer.cpp:33-55), and the per-sample part of pathTraceBlock (:706-727, :743-779).
//
// Execution model (the reference runs one thread per 32x32 block and one sample after another):
//   * every lane of a wave64 owns ONE path sample at a time and keeps its whole state in registers;
//   * all lanes that hold a live ray advance the SAME step-wise BVH traverser (pt_traverse.hip.h, one 64-byte record
//     fetch per step whatever the lane is doing);
//   * lanes whose ray has finished wait until enough of them have (ballot + popcount), then the wave runs the
//     "service" section once for all of them: shade / sample the light / start the shadow ray / start the bounce ray /
//     retire the sample and pull the next one from the wave's sample range (ballot + prefix rank);
//   * a wave exits when its sample range is exhausted and no lane holds a sample.
// Which lane or wave computes a sample never affects its value: the sample's Sobol indices and RNG stream are
// functions of (block, pass_seed, i, k) only, and every arithmetic step is the oracle's, in the oracle's order
// (no FMA contraction, IEEE divide/sqrt, portable transcendentals) => per-sample radiance is bit-identical to the CPU.
// Radiance is written per sample; pixels are accumulated afterwards in the reference's order (pt_accumulate kernels).
#pragma once
#include "pt_traverse.hip.h"
#include "../../include/ptgpu.h"

namespace ptd {

struct ShadeScene {
    const float* pos;  // [num_tris][3][3]
    const float* nrm;  // [num_tris][3][3]
    const float* uv;   // [num_tris][3][2]
    const ptgpu_material* materials;
    const DevTexture* textures;
    const uint2* sobol_pairs;  // [SOBOL_PAIRS][3][256], see sobol32x2
};

// work description of one integrator launch: blocks b = block_begin + (j0 + jj) * block_stride, jj in [0, num_blocks)
struct PassDesc {
    int32_t block_begin, block_stride, j0, num_blocks;
    int32_t blocks_x;          // blocks per image row (reference :816-818)
    int32_t samples_per_block; // padded: 1024 * num_aa_rays
    int32_t num_lights;
    float inv_prob;            // 1 / m_terminationProbability (:468)
    // EXT instantiation only (ptgpu_trace_samples_dev): the launch traces num_ext caller-given samples instead of blocks -- sample s
    // has clip-space position ext_xy[2s..], Sobol base ext_base[s] and random stream state (key, counter) ext_rng[2s..]; exactly the
    // arguments of the reference's tracePath(x, y, ctx, samplerBase, R, ...) (src/PathTraceRenderer.hpp:97-98)
    const float* ext_xy;
    const int32_t* ext_base;
    const uint32_t* ext_rng;
    uint32_t num_ext;
    // Cancellation (ptgpu_cancel; the reference's workers poll m_bForceExit before every sample, src/PathTraceRenderer.cpp:709,736): a word
    // in device memory that holds the number of the last cancelled pass, stored by ptgpu_cancel's one-wave kernel on a stream of its own
    // while this launch runs; this launch belongs to pass `epoch`.  NULL: never cancelled.
    const uint32_t* cancel;
    uint32_t epoch;
};

// Has the pass this launch belongs to been cancelled?  Wave-uniform; one agent-scope load (coherent with the agent-scope store of
// k_cancel_set whichever XCD either runs on).
__device__ __forceinline__ bool pass_cancelled(const PassDesc& pd) {
#ifdef PT_NO_CANCEL  // A/B builds only (tools/build_variant.sh): what the poll costs
    return false;
#endif
    if (!pd.cancel) return false;
    const uint32_t w = __hip_atomic_load(pd.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int32_t)(__builtin_amdgcn_readfirstlane(w) - pd.epoch) >= 0;
}

// x / 255.0f for x = 0, 1, ..., 255: one multiply and two fused steps give the correctly rounded quotient for every one of the 256
// inputs (checked exhaustively in tests/test_oracle_cpu.py), i.e. the same float as the reference's IEEE division, in 3 instructions
// instead of the ~10 of the general division sequence (18 of them per textured hit).
__device__ __forceinline__ float div255(float x) {
    const float y = 1.0f / 255.0f;
    const float q = x * y;
    const float r = __builtin_fmaf(-255.0f, q, x);
    return __builtin_fmaf(r, y, q);
}

// a texel in two halves, so that a caller can issue several fetches before it consumes the first
__device__ __forceinline__ uint32_t texel_fetch(const DevTexture& t, int x, int y) {
    x = x < 0 ? 0 : (x > t.width - 1 ? t.width - 1 : x);
    y = y < 0 ? 0 : (y > t.height - 1 ? t.height - 1 : y);
    return *reinterpret_cast<const uint32_t*>(t.rgba + ((size_t)y * t.width + x) * 4);
}
__device__ __forceinline__ F4 texel_unpack(uint32_t p) {
    F4 r;
    r.x = div255((float)(p & 0xffu)); r.y = div255((float)((p >> 8) & 0xffu));
    r.z = div255((float)((p >> 16) & 0xffu)); r.w = div255((float)(p >> 24));
    return r;
}
__device__ __forceinline__ F4 texel(const DevTexture& t, int x, int y) { return texel_unpack(texel_fetch(t, x, y)); }
__device__ __forceinline__ void texel_coords(F2 uv, const DevTexture& t, int& x, int& y) {
    x = (int)((uv.x - floorf(uv.x)) * (float)t.width);
    y = (int)((uv.y - floorf(uv.y)) * (float)t.height);
}

// src/util.cpp:22-47
__device__ __forceinline__ void map_to_disk(float a, float b, float& x, float& y) {
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
    float s, c;
    ptm_sincos(phi, s, c);
    x = r * c;
    y = r * s;
}

// src/RayTracer.cpp:33-55, applied to a vector: returns formBasis(n) * v
__device__ __forceinline__ F3 basis_times(F3 n, F3 v) {
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    int k = 0;
    float am = ax;
    if (ay < am) { k = 1; am = ay; }
    if (az < am) { k = 2; }
    F3 q = n;
    if (k == 0) q.x = 1.0f; else if (k == 1) q.y = 1.0f; else q.z = 1.0f;
    F3 t = normalize(cross(q, n));
    F3 b = normalize(cross(n, t));
    return f3(t.x * v.x + b.x * v.y + n.x * v.z, t.y * v.x + b.y * v.y + n.y * v.z, t.z * v.x + b.z * v.y + n.z * v.z);
}

__device__ __forceinline__ int sobol_bounce_dim(int bounce) { return bounce > 254 ? 254 : bounce; }

// 32-bit index form of sobol::sample (every index this build forms is a non-negative int)
__device__ __forceinline__ float sobol32(uint32_t index, unsigned dimension) {
    unsigned result = 0u;
    const uint32_t* row = g_sobol_matrices + dimension * 52u;
    for (unsigned i = 0; index; index >>= 1, ++i) result ^= row[i] & (0u - (index & 1u));
    return (float)result * (1.f / 4294967296.f);
}
// two consecutive dimensions (dimension even) of one index -- the reference always draws them in pairs.  The low 24 index bits go
// through the byte table (ptgpu.hip k_build_sobol_pairs), any higher bits through sobol::sample's bit loop.
constexpr int SOBOL_PAIRS = 512;
__device__ __forceinline__ void sobol32x2(const uint2* __restrict__ tab, uint32_t index, unsigned dimension, float& a, float& b) {
    sobol32x2m(tab, g_sobol_matrices, index, dimension, a, b);
}

__device__ __forceinline__ F3 light_col(const ptgpu_light& l, int c) { return f3(l.xform[c * 4], l.xform[c * 4 + 1], l.xform[c * 4 + 2]); }

// src/PathTraceRenderer.cpp:183-210
__device__ __forceinline__ int get_light_to_sample(F3 pos, const ptgpu_light* lights, int n, Rng& R) {
    if (n <= 1) return 0;
    float sum = 0.0f;
    for (int i = 0; i < n; i++) sum += lights[i].power / length(pos - light_col(lights[i], 3));
    float r = R.getF32(0.0f, 1.0f);
    float value = 0.0f;
    for (int i = 0; i < n; i++) {
        value += (lights[i].power / length(pos - light_col(lights[i], 3))) / sum;
        if (r <= value) return i;
    }
    return n - 1;
}

// src/PathTraceRenderer.cpp:75-181
__device__ __forceinline__ void get_texture_parameters(const ShadeScene& S, const ptgpu_params& prm, int tri, int mat_id, float hu, float hv,
                                                       F3& diffuse, F3& n, F3& specular) {
    const ptgpu_material& mat = S.materials[mat_id];
    const float* tc = S.uv + (size_t)tri * 6;
    F2 uv0 = ld2(tc), uv1 = ld2(tc + 2), uv2 = ld2(tc + 4);
    F2 uv = (1 - hu - hv) * uv0 + hu * uv1 + hv * uv2;
    const float* nn = S.nrm + (size_t)tri * 9;
    const F3 nrm0 = ld3(nn), nrm1 = ld3(nn + 3), nrm2 = ld3(nn + 6);  // issued here, used after the texels

    // Every texel this hit needs is fetched before the first is used: the maps live beyond L2 (Infinity Cache or HBM), and three
    // dependent miss latencies in a row (diffuse, normal, specular) become one.
    const bool has_d = mat.tex_diffuse >= 0, has_n = mat.tex_normal >= 0 && prm.normal_mapped, has_s = mat.tex_specular >= 0;
    uint32_t rd1 = 0, rd2 = 0, rd3 = 0, rd4 = 0, rn = 0, rs = 0;
    float w1 = 0.0f, w2 = 0.0f, w3 = 0.0f, w4 = 0.0f;
    F3 v0 = f3(0.0f, 0.0f, 0.0f), v1 = v0, v2 = v0;  // the triangle's corners, for the tangent frame of a normal map
    if (has_d) {
        DevTexture tex = S.textures[mat.tex_diffuse];
        if (prm.bilinear) {
            float fx = uv.x - floorf(uv.x);
            float fy = uv.y - floorf(uv.y);
            float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
            w1 = fx1 * fy1; w2 = fx * fy1; w3 = fx1 * fy; w4 = fx * fy;
            int x1 = (int)sel_min(sel_max(fx * (float)tex.width - 0.5f, 0.0f), (float)tex.width - 2.0f);
            int y1 = (int)sel_min(sel_max(fy * (float)tex.height - 0.5f, 0.0f), (float)tex.height - 2.0f);
            int x2 = x1 + 1, y2 = y1 + 1;
            rd1 = texel_fetch(tex, x1, y1); rd2 = texel_fetch(tex, x2, y1); rd3 = texel_fetch(tex, x1, y2); rd4 = texel_fetch(tex, x2, y2);
        } else {
            int x, y;
            texel_coords(uv, tex, x, y);
            rd1 = texel_fetch(tex, x, y);
        }
    }
    if (has_n) {
        DevTexture tex = S.textures[mat.tex_normal];
        int x, y;
        texel_coords(uv, tex, x, y);
        rn = texel_fetch(tex, x, y);
        const float* pp = S.pos + (size_t)tri * 9;
        v0 = ld3(pp); v1 = ld3(pp + 3); v2 = ld3(pp + 6);
    }
    if (has_s) {
        DevTexture tex = S.textures[mat.tex_specular];
        int x, y;
        texel_coords(uv, tex, x, y);
        rs = texel_fetch(tex, x, y);
    }

    diffuse = f3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]);
    if (has_d) {
        if (prm.bilinear) {
            F4 d1 = texel_unpack(rd1), d2 = texel_unpack(rd2), d3 = texel_unpack(rd3), d4 = texel_unpack(rd4);
            diffuse = f3(d1.x * w1 + d2.x * w2 + d3.x * w3 + d4.x * w4,
                         d1.y * w1 + d2.y * w2 + d3.y * w3 + d4.y * w4,
                         d1.z * w1 + d2.z * w2 + d3.z * w3 + d4.z * w4);
        } else {
            F4 d = texel_unpack(rd1);
            diffuse = f3(d.x, d.y, d.z);
        }
        diffuse = f3(ptm_pow(diffuse.x, 2.2f), ptm_pow(diffuse.y, 2.2f), ptm_pow(diffuse.z, 2.2f));
    }
    n = normalize((1 - hu - hv) * nrm0 + hu * nrm1 + hv * nrm2);

    if (has_n) {
        F4 tv = texel_unpack(rn);
        F3 texture_normal = f3(2.0f * tv.x - 1.0f, 2.0f * tv.y - 1.0f, 2.0f * tv.z - 1.0f);
        F3 deltaPos1 = v1 - v0, deltaPos2 = v2 - v0;
        F2 deltaUV1 = uv1 - uv0, deltaUV2 = uv2 - uv0;
        float r = 1.0f / (deltaUV1.x * deltaUV2.y - deltaUV1.y * deltaUV2.x);
        F3 tangent = normalize((deltaPos1 * deltaUV2.y - deltaPos2 * deltaUV1.y) * r);
        F3 bitangent = normalize((deltaPos2 * deltaUV1.x - deltaPos1 * deltaUV2.x) * r);
        F3 tn = texture_normal;
        n = normalize(f3(tangent.x * tn.x + bitangent.x * tn.y + n.x * tn.z, tangent.y * tn.x + bitangent.y * tn.y + n.y * tn.z,
                         tangent.z * tn.x + bitangent.z * tn.y + n.z * tn.z));
    }
    specular = f3(mat.specular[0], mat.specular[1], mat.specular[2]);
    if (has_s) {
        F4 d = texel_unpack(rs);
        specular = f3(d.x, d.y, d.z);
    }
}

// src/PathTraceRenderer.cpp:52-55
__device__ __forceinline__ float gauss(float x, float y) {
    const float PI = 3.14159265358979323846f;
    float sigma = 1.0f;
    return 1.0f / (sqrtf(2.0f * PI) * sigma) * ptm_exp(-1.0f * (x * x + y * y) / (2.0f * sigma * sigma));
}

// Sobol sequence base of a block (reference :689-694); see oracle block_sobol_base for the deviation note
__device__ __forceinline__ int block_sobol_base(int idx, uint32_t pass_seed) {
    Rng Rb;
    Rb.seed((uint32_t)idx, pass_seed, 0xB10Cu, 0u);
    int mul = Rb.getS32(1, 9);
    int add = Rb.getS32(1, 12345);
    return idx * mul + add;
}

// Diagnostic build only (-DPT_STAMPS): per-wave cycle and lane-count sums per loop section, added into counters[8..].
// Sections: 0 control+classification, 1 shade, 2 hit set-up, 3 whitted shadow, 4 bounce, 5 new sample, 6 traversal step.
// counters[8+3k] = cycles, [9+3k] = invocations, [10+3k] = lanes served.  Never quote this build's run time.
#ifdef PT_STAMPS
#define PT_STAMP_BEGIN() const long long st_t0_ = (long long)__builtin_amdgcn_s_memtime()
#define PT_STAMP_END(k, lanes) do { st_cyc[k] += (long long)__builtin_amdgcn_s_memtime() - st_t0_; st_inv[k] += 1; st_lanes[k] += (lanes); } while (0)
#else
#define PT_STAMP_BEGIN() do {} while (0)
#define PT_STAMP_END(k, lanes) do {} while (0)
#endif

// Wave priorities (s_setprio) per section of the loop.  A SIMD's four waves are in different sections at any time; the traversal steps
// are the ones that issue memory instructions and then wait, so they go first and the arithmetic of the other sections fills the time
// their loads are in flight.  Measured, untextured 16 spp pass: all 0: 21.96 ms; steps 1, 2 or 3: 19.9; steps 3 + hit block 1: 19.4
// (hit 2: the same); a priority for the new-sample block on top changes nothing.
#ifndef PT_PRIO_STEP
#define PT_PRIO_STEP 3
#endif
#ifndef PT_PRIO_HIT
#define PT_PRIO_HIT 1
#endif
#ifndef PT_PRIO_NEW
#define PT_PRIO_NEW 0
#endif
#ifndef PT_PHASE_NODE_REPS
#define PT_PHASE_NODE_REPS 3
#endif
#ifndef PT_PHASE_TRI_REPS
#define PT_PHASE_TRI_REPS 3
#endif
#ifndef PT_PHASE_BIAS_N
#define PT_PHASE_BIAS_N 1
#endif
#ifndef PT_PHASE_BIAS_T
#define PT_PHASE_BIAS_T 1  // a node phase when node lanes are at least as many as triangle lanes (with wave priorities: 1:1 19.2 ms, 2:1 19.45, 3:1 19.8; without them 2:1 was best)
#endif
#ifndef PT_STEPS_PER_ITERATION
#define PT_STEPS_PER_ITERATION 3
#endif

// per-lane path state between two service rounds
//   ST_EXT      closest-hit ray in flight, or finished and waiting for the hit block
//   ST_SH       shadow ray in flight; for path tracing everything that follows it is already decided (see "pending bounce")
//   ST_BOUNCE   waits for a bounce ray from the hit point it already has (miss inside the fixed-bounce loop, then roulette)
//   ST_REBOUNCE the same, but throughput is already updated (pending bounce discarded because the normal was flipped)
//   ST_WSHADOW  Whitted: waits for its next light ray;  ST_FIN: transient result of advance()
enum : int { ST_IDLE = 0, ST_EXT = 1, ST_SH = 2, ST_BOUNCE = 3, ST_WSHADOW = 4, ST_REBOUNCE = 5, ST_FIN = 6 };
enum : uint32_t { PB_PENDING = 64u, PB_BACKFACING = 128u };  // above Trav::Prepared's bits

struct PathState {
    F3 O, N, V, diffuse, throughput, Ei;
    F3 c;      // path tracing: radiance this hit adds if its shadow ray is unoccluded; Whitted: the specular colour
    F3 inv_n;  // pending bounce ray: reciprocal direction
    uint32_t bits;  // pending bounce ray: Trav::Prepared::bits | PB_*
    float gloss, normalization, aux;  // aux: 4th output word (local pixel bits or Gaussian weight)
    int li0;  // path tracing: the light chosen at the path's FIRST hit, whose emission every later hit uses (:341); Whitted: this hit's light
    int bounce, base, mode, stage;
    int sid;  // padded sample id inside this launch
    Rng R;
};

// One integrator launch.  rad[sid] = (Ei.rgb, aux) for every valid padded sample id of the launch's blocks.
// spill: samples of Sobol-block mode whose pixel falls outside their own block (sobol() == 1.0f exactly, p = 2^-24).
template <bool COUNT, int NLDS, bool OVF, int BLOCK, bool ALPHA, bool EXT = false>
__device__ __forceinline__ void integrate(const TravScene& TS, const ShadeScene& SS, const ptgpu_params& prm, const ptgpu_camera& cam,
                                          const ptgpu_light* __restrict__ lights, const PassDesc& pd, float4* __restrict__ rad,
                                          unsigned long long* counters, uint2* lds_lane, uint2* gl_lane, int service_threshold_packed) {
    const int service_threshold = service_threshold_packed & 0xff, control_min = (service_threshold_packed >> 8) & 0xff;
    const int lane = threadIdx.x & 63;
    Trav<ALPHA, COUNT, NLDS, OVF> tr;
    tr.reset_counts();
    tr.attach(lds_lane, gl_lane);
    tr.alive = false;
    PathState ps;
    ps.stage = ST_IDLE;
    ps.sid = -1;
    uint32_t n_rays = 0;

    // Sample ids of one launch fit 31 bits (the host caps a batch at 2^30).  The chunk cursor is wave-uniform: readfirstlane keeps it in
    // scalar registers (as 64-bit per-lane values it cost 8 VGPRs for the whole life of the kernel).
    const uint32_t total = EXT ? pd.num_ext : (uint32_t)pd.num_blocks * (uint32_t)pd.samples_per_block;
    const uint32_t total_waves = gridDim.x * (BLOCK / 64);
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
    uint32_t next = 0, end = 0, round = 0xffffffffu;
    bool exhausted = false;
    const int bounces = prm.bounces < 0 ? -prm.bounces : prm.bounces;
    const float PI = 3.14159265358979323846f;

    // finish a sample: its radiance leaves through rad[], the lane becomes idle
    auto finish = [&]() {
        float4 o;
        o.x = ps.Ei.x; o.y = ps.Ei.y; o.z = ps.Ei.z; o.w = EXT ? __uint_as_float(ps.R.ctr) : ps.aux;  // EXT: where the sample left its random stream
        rad[ps.sid] = o;
        ps.stage = ST_IDLE;
        ps.sid = -1;
    };
    // loop control of tracePath between two rays: 1 = fixed-bounce loop test (:389), 2 = roulette entry (:466-468),
    // 3 = roulette loop test (:471).  Leaves the lane asking for a bounce ray (ST_BOUNCE) or at the end of its path (ST_FIN).
    auto advance = [&](int next_ctl, float terminate) {
        if (next_ctl == 1) {
            if (ps.bounce < bounces) { ps.mode = 1; ps.stage = ST_BOUNCE; return; }
            next_ctl = 2;
        }
        if (next_ctl == 2) {
            if (prm.bounces < 0) { terminate = ps.R.getF32(0.0f, 1.0f); next_ctl = 3; }
            else { ps.stage = ST_FIN; return; }
        }
        if (terminate > prm.termination_prob) { ps.mode = 2; ps.stage = ST_BOUNCE; }
        else ps.stage = ST_FIN;
    };

#ifdef PT_STAMPS
    long long st_cyc[7] = {0, 0, 0, 0, 0, 0, 0}, st_inv[7] = {0, 0, 0, 0, 0, 0, 0}, st_lanes[7] = {0, 0, 0, 0, 0, 0, 0}, st_phase[3] = {0, 0, 0};
#endif
    // Whitted: next disk sample of the chosen light and its shadow ray (shared tail of the hit and shade blocks)
    auto whitted_shadow = [&]() {
        // AreaLight::sampleDisk, src/AreaLight.cpp:87-108, with dimension 0
        const ptgpu_light& Lt = lights[ps.li0];
        int r = ps.R.getS32(1, 1000);
        float sa, sb;
        sobol32x2(SS.sobol_pairs, (uint32_t)(ps.base + r), 4u, sa, sb);
        float a = (2.0f * sa - 1.0f) * Lt.size[0];
        float b = (2.0f * sb - 1.0f) * Lt.size[1];
        float x, y;
        map_to_disk(a, b, x, y);
        F3 Pl = light_col(Lt, 3) + light_col(Lt, 0) * x + light_col(Lt, 1) * y;
        F3 L = Pl - ps.O;
        F3 D = normalize(L);
        tr.begin(TS, ps.O + 0.0001f * D, L, true);  // origin offset along D here, not N (:634)
        n_rays += 1;
        ps.stage = ST_SH;
    };

    // Wave-uniform scheduling.  A lane is traversing (alive) or waits for ONE of three service blocks:
    //   H  after a closest-hit ray that hit: texture parameters, light choice, light sample, shadow ray, AND everything the reference
    //      does after the shadow ray except adding the light's contribution: the contribution itself (kept in ps.c), the loop control
    //      with its random draws, and the bounce ray (kept pending: direction in ps.V, reciprocal in ps.inv_n).  None of that depends
    //      on the shadow ray's outcome, with one exception (below), and the sample's random stream is consumed in the same order.
    //      Lanes that only need a bounce ray from the hit point they already have (ST_BOUNCE / ST_REBOUNCE) are served here too.
    //   S  Whitted only: count the light ray, next light ray or final colour
    //   N  retire / new sample
    // When a path-tracing shadow ray ends, the cheap control section adds ps.c if it was unoccluded and either finishes the sample or
    // launches the pending bounce ray -- no second wait for a service block.  The exception: the reference flips the shading normal of a
    // back-facing hit only inside its "unoccluded" branch (:346-353), and the bounce leaves along the normal; for a back-facing,
    // unoccluded hit the pending bounce (built from the unflipped normal) is discarded and the lane asks for a new one (ST_REBOUNCE).
    // The wave runs the block most lanes wait for once that count reaches service_threshold (or exceeds the traversing lanes), so each
    // block runs with many of its lanes; otherwise all traversing lanes take one step.  The control section (settling finished rays,
    // classification) is entered only after a block has run or once control_min more rays have finished (or none is left running):
    // entering it for every single finished ray cost more instructions than the idle lanes it saved (16: -5.5 % per pass).
    bool dirty = true;
    unsigned long long prev_alive = 0ull;
    for (;;) {
        unsigned long long alive_mask = __ballot(tr.alive);
        const int newly_done = __popcll(prev_alive & ~alive_mask);
        if (dirty || newly_done >= control_min || (newly_done > 0 && alive_mask == 0ull)) {
            PT_STAMP_BEGIN();
            dirty = false;
            if (!tr.alive && ps.stage == ST_SH && ps.mode != 3) {
                // shadow ray of a path-tracing hit: reference :346-376 / :416-450 / :500-534 reduced to their outcome
                const bool unoccluded = tr.pmin < 0;
                if (unoccluded) ps.Ei = ps.Ei + ps.c;
                if (!(ps.bits & PB_PENDING)) finish();
                else if (unoccluded && (ps.bits & PB_BACKFACING)) {
                    ps.N = ps.N * -1.0f;
                    ps.stage = ST_REBOUNCE;
                } else {
                    typename Trav<ALPHA, COUNT, NLDS, OVF>::Prepared pr;
                    pr.inv = ps.inv_n;
                    pr.bits = ps.bits;
                    tr.launch(TS, ps.O + 0.0001f * ps.N, cam.far_ * ps.V, pr, false);
                    n_rays += 1;
                    ps.stage = ST_EXT;
                }
            }
            // closest-hit ray that missed (after the section above: a bounce ray launched there may miss the scene's box at once)
            if (!tr.alive && ps.stage == ST_EXT && tr.pmin < 0) {
                if (ps.mode == 1) {  // inside the fixed-bounce loop: `break`, then the roulette section
                    advance(2, 0.0f);
                    if (ps.stage == ST_FIN) finish();
                } else finish();     // primary miss (Ei = 0), or miss inside the roulette loop
            }
            alive_mask = __ballot(tr.alive);
            const int nA = __popcll(alive_mask);
            const int nH = __popcll(__ballot((!tr.alive && ps.stage == ST_EXT) || ps.stage == ST_BOUNCE || ps.stage == ST_REBOUNCE));
            const int nS = __popcll(__ballot(!tr.alive && ps.stage == ST_SH));  // Whitted only: other shadow rays were settled above
            const int nN = exhausted ? 0 : __popcll(__ballot(ps.stage == ST_IDLE));
            if (nA == 0 && nH + nS + nN == 0) break;  // sample range exhausted and every lane idle
            int best = nH;
            if (nS > best) best = nS;
            if (nN > best) best = nN;
            const bool run = best > 0 && (best >= service_threshold || best >= nA);
            PT_STAMP_END(0, nH + nS + nN);
            if (run && nH == best) {
                PT_STAMP_BEGIN();
#if PT_PRIO_HIT
                __builtin_amdgcn_s_setprio(PT_PRIO_HIT);
#endif
                const bool hit_lane = !tr.alive && ps.stage == ST_EXT;
                bool want_bounce = ps.stage == ST_BOUNCE || ps.stage == ST_REBOUNCE;
                if (hit_lane) {
                    // hit set-up: reference :300-344 (first hit), :403-414 / :486-498 (bounce hits), :608-613 (Whitted)
                    const float4 rec = tri_record(TS, tr.pmin)[3];
                    const int hit_tri = __float_as_int(rec.x), hit_mat = __float_as_int(rec.w);
                    const bool first = ps.mode == 0;
                    if (first || prm.whitted) {
                        ps.gloss = SS.materials[hit_mat].glossiness;
                        ps.normalization = (ps.gloss + 2.0f) / (4.0f * PI * (2.0f - ptm_pow(2.0f, -ps.gloss / 2.0f)));
                        ps.bounce = 0;
                        ps.throughput = f3(1.0f, 1.0f, 1.0f);
                        ps.V = normalize(tr.d);
                    }
                    ps.O = tr.o + tr.tmin * tr.d;
                    F3 specular;
                    get_texture_parameters(SS, prm, hit_tri, hit_mat, tr.umin, tr.vmin, ps.diffuse, ps.N, specular);
                    const int li = get_light_to_sample(ps.O, lights, pd.num_lights, ps.R);
                    if (first || prm.whitted) ps.li0 = li;
                    if (prm.whitted) {
                        ps.c = specular;
                        ps.inv_n.x = 0.0f;  // "shadow": number of unoccluded light rays (inv_n is otherwise unused here); ps.bounce = light-ray index
                        ps.mode = 3;
                        ps.stage = ST_WSHADOW;
                    } else {
                        const ptgpu_light& Lt = lights[li];
                        // AreaLight::sample, src/AreaLight.cpp:69-85
                        int dim = 4 + 4 * sobol_bounce_dim(ps.bounce);
                        int r = ps.R.getS32(1, 10000);
                        float sa, sb;
                        sobol32x2(SS.sobol_pairs, (uint32_t)(ps.base + r), (unsigned)dim, sa, sb);
                        float lx = (2.0f * sa - 1.0f) * Lt.size[0];
                        float ly = (2.0f * sb - 1.0f) * Lt.size[1];
                        F3 Pl = light_col(Lt, 3) + light_col(Lt, 0) * lx + light_col(Lt, 1) * ly;
                        const float pdf = 1.0f / (4.0f * Lt.size[0] * Lt.size[1]);
                        const F3 L = Pl - ps.O;
                        const ptgpu_light& L0 = lights[ps.li0];
                        const F3 emission = f3(L0.E[0], L0.E[1], L0.E[2]) * prm.intensity;  // set once per path in the reference (:341)
                        tr.begin(TS, ps.O + 0.0001f * ps.N, L, true);
                        n_rays += 1;
                        // what the light adds if the shadow ray finds nothing: reference :346-376 / :416-450 / :500-534
                        {
                            F3 D = normalize(L);
                            F3 Nl = -light_col(Lt, 2);
                            const bool back = dot(ps.N, ps.V) > 0.0f;
                            const F3 Nf = back ? ps.N * -1.0f : ps.N;
                            float cos_theta = sel_max(0.0f, dot(Nf, D));
                            float cos_theta_l = sel_max(0.0f, dot(Nl, -D));
                            float d = length(L);
                            float inv_d = sel_min(1.0f / (d * d), 100000.0f);
                            float inv_pdf = sel_min(1.0f / pdf, 100000.0f);
                            F3 H = normalize(D - ps.V);
                            F3 color = ps.diffuse;
                            color = color + specular * ptm_pow(sel_max(0.0f, dot(Nf, H)), ps.gloss) * ps.normalization;
                            const float invPi = 0.31830988618379f;
                            if (ps.mode == 0) ps.c = color * invPi * emission * cos_theta * cos_theta_l * inv_d * inv_pdf;
                            else if (ps.mode == 1) ps.c = color * invPi * ps.throughput * emission * cos_theta * cos_theta_l * inv_d * inv_pdf;
                            else ps.c = color * invPi * ps.throughput * emission * cos_theta * cos_theta_l * inv_d * inv_pdf * pd.inv_prob;
                            ps.bits = back ? PB_BACKFACING : 0u;
                        }
                        // loop control after the shade (random draws in the reference's order)
                        if (ps.mode == 2) { const float terminate = ps.R.getF32(0.0f, 1.0f); ps.bounce += 1; advance(3, terminate); }
                        else { if (ps.mode == 1) ps.bounce += 1; advance(1, 0.0f); }
                        want_bounce = ps.stage == ST_BOUNCE;
                    }
                }
                if (ps.stage == ST_WSHADOW) whitted_shadow();
                if (want_bounce) {
                    // reference :391-401 / :473-484; drawCosineWeightedDirection :57-73 (RNG peeked, not advanced)
                    int dim = 2 + 4 * sobol_bounce_dim(ps.bounce);
                    int rnd = ps.R.peekS32(1, 1234);
                    float sa, sb;
                    sobol32x2(SS.sobol_pairs, (uint32_t)(ps.base + rnd), (unsigned)dim, sa, sb);
                    float a = 2.0f * sa - 1.0f, b = 2.0f * sb - 1.0f;
                    float x, y;
                    map_to_disk(a, b, x, y);
                    float z = sqrtf(1 - x * x - y * y);
                    ps.V = normalize(basis_times(ps.N, f3(x, y, z)));
                    if (ps.stage != ST_REBOUNCE) ps.throughput = ps.throughput * ps.diffuse;
                    const F3 bo = ps.O + 0.0001f * ps.N, bd = cam.far_ * ps.V;
                    const auto pr = Trav<ALPHA, COUNT, NLDS, OVF>::prepare(TS, bo, bd);
                    if (hit_lane) {  // the shadow ray goes first
                        ps.inv_n = pr.inv;
                        ps.bits |= pr.bits | PB_PENDING;
                    } else {
                        tr.launch(TS, bo, bd, pr, false);
                        n_rays += 1;
                        ps.stage = ST_EXT;
                    }
                }
                if (hit_lane && !prm.whitted) ps.stage = ST_SH;
                PT_STAMP_END(2, nH);
#if PT_PRIO_HIT
                __builtin_amdgcn_s_setprio(0);
#endif
                dirty = true;
            } else if (run && nS == best) {
                PT_STAMP_BEGIN();
                if (!tr.alive && ps.stage == ST_SH) {
                    // traceWhitted's light-ray loop (reference :614-637): count unoccluded disk samples
                    if (tr.pmin < 0) ps.inv_n.x += 1.0f;
                    ps.bounce += 1;
                    if (ps.bounce < prm.num_light_rays) ps.stage = ST_WSHADOW;
                    else {
                        // reference :639-656
                        const ptgpu_light& Lt = lights[ps.li0];
                        F3 L = light_col(Lt, 3) - ps.O;
                        F3 D = normalize(L);
                        float d = length(L);
                        float attenuation = prm.intensity / (prm.attenuation * ptm_pow(d, 2.0f) + prm.attenuation * d + 0.5f);
                        F3 H = normalize(D - ps.V);
                        F3 color = ps.diffuse * sel_max(0.0f, dot(ps.N, D));
                        color = color + ps.c * ptm_pow(sel_max(0.0f, dot(ps.N, H)), ps.gloss) * ps.normalization;
                        color = color * attenuation;
                        color = color * (ps.inv_n.x / (float)prm.num_light_rays);
                        ps.Ei = color;
                        finish();
                    }
                }
                if (ps.stage == ST_WSHADOW) whitted_shadow();
                PT_STAMP_END(1, nS);
                dirty = true;
            } else if (run) {
                PT_STAMP_BEGIN();
#if PT_PRIO_NEW
                __builtin_amdgcn_s_setprio(PT_PRIO_NEW);
#endif
                // ---- pull new samples for idle lanes (ballot + prefix rank over the wave's current chunk)
                unsigned long long need = __ballot(ps.stage == ST_IDLE);
                while (need != 0ull && !exhausted) {
                    if (next >= end) {
                        // round r hands this wave chunk r*W + (w + 997 r) mod W: rotating the assignment every round keeps one wave
                        // from drawing the same image region (all cheap or all expensive) in every round
                        round = __builtin_amdgcn_readfirstlane(round + 1u);
                        // a cancelled pass hands out no more samples (the ones in flight run to their end)
                        if (!EXT && pass_cancelled(pd)) { exhausted = true; break; }
                        const unsigned long long cbase = ((unsigned long long)round * total_waves + (wave_id + (round % total_waves) * 997u) % total_waves) * 64ull;
                        if (cbase >= (unsigned long long)total) { exhausted = true; break; }
                        next = __builtin_amdgcn_readfirstlane((uint32_t)cbase);
                        end = next + 64u < total ? next + 64u : total;
                    }
                    const int avail = (int)(end - next);
                    const int rank = __popcll(need & ((1ull << lane) - 1ull));
                    if (ps.stage == ST_IDLE && rank < avail) {
                        // The uniform divisors and sizes of this block pass through an empty asm: everything derived from them
                        // (reciprocals, division magic, int->float) is then computed here, per batch of new samples, instead of
                        // being hoisted out of the kernel's loop into VGPRs that stay occupied (and spilled) for its whole life.
                        int n_width = prm.width, n_height = prm.height, n_spb = pd.samples_per_block, n_blocks_x = pd.blocks_x, n_aa = prm.num_aa_rays;
                        float n_filter = prm.filter_width;
                        asm volatile("" : "+s"(n_width), "+s"(n_height), "+s"(n_spb), "+s"(n_blocks_x), "+s"(n_aa), "+s"(n_filter));
                        const float div_x = 1.0f / (float)n_width, div_y = 1.0f / (float)n_height;
                        const int sid = (int)(next + (uint32_t)rank);
                        bool valid;
                        float px = 0.0f, py = 0.0f, ext_x = 0.0f, ext_y = 0.0f;
                        int sbase;
                        if (EXT) {  // caller-given sample: tracePath's own arguments
                            valid = true;
                            ext_x = pd.ext_xy[2 * (size_t)sid]; ext_y = pd.ext_xy[2 * (size_t)sid + 1];
                            sbase = pd.ext_base[sid];
                            ps.R.key = pd.ext_rng[2 * (size_t)sid]; ps.R.ctr = pd.ext_rng[2 * (size_t)sid + 1];
                            ps.aux = 0.0f;
                        } else {
                        const int jj = (int)((uint32_t)sid / (uint32_t)n_spb);
                        const int s = sid - jj * n_spb;
                        const int idx = pd.block_begin + (pd.j0 + jj) * pd.block_stride;
                        const int bx = (idx % n_blocks_x) * 32, by = (idx / n_blocks_x) * 32;
                        const int bw = (bx + 32 < n_width) ? 32 : n_width - bx;
                        const int bh = (by + 32 < n_height) ? 32 : n_height - by;
                        const int base = block_sobol_base(idx, prm.pass_seed);
                        if (prm.sobol_block) {  // reference :706-727
                            const int i = s;
                            valid = i < bw * bh * n_aa;
                            float sa, sb;
                            sobol32x2(SS.sobol_pairs, (uint32_t)(base + i), 0u, sa, sb);
                            px = (float)bx + (float)bw * sa;
                            py = (float)by + (float)bh * sb;
                            sbase = base + i;
                            ps.R.seed((uint32_t)idx, prm.pass_seed, (uint32_t)i, 0u);
                            int pixel_x = (int)floorf(px); if (pixel_x > n_width - 1) pixel_x = n_width - 1;
                            int pixel_y = (int)floorf(py); if (pixel_y > n_height - 1) pixel_y = n_height - 1;
                            ps.aux = __int_as_float(pixel_y * n_width + pixel_x);  // global pixel index; accumulate checks the block
                        } else {  // reference :743-779
                            const int i = s / n_aa, k = s - i * n_aa;
                            valid = i < bw * bh;
                            const int pixel_x = bx + (i % bw), pixel_y = by + (i / bw);
                            float sa, sb;
                            sobol32x2(SS.sobol_pairs, (uint32_t)(base + i + k), 0u, sa, sb);
                            float a = n_filter * (sa - 0.5f) + 0.5f;
                            float b = n_filter * (sb - 0.5f) + 0.5f;
                            px = (float)pixel_x + a;
                            py = (float)pixel_y + b;
                            if (px < 0.0f) px *= -1.0f;
                            if (py < 0.0f) py *= -1.0f;
                            if (px > (float)n_width) px -= a;
                            if (py > (float)n_height) py -= b;
                            sbase = base + i;
                            ps.R.seed((uint32_t)idx, prm.pass_seed, (uint32_t)i, (uint32_t)k);
                            const float fs = 2.0f / n_filter;
                            ps.aux = gauss(fs * (a - 0.5f), fs * (b - 0.5f));
                        }
                        }
                        if (valid) {
                            float x = EXT ? ext_x : px * div_x * 2.0f - 1.0f;
                            float y = EXT ? ext_y : py * div_y * -2.0f + 1.0f;
                            // camera ray, reference :244-292
                            F4 Roh = mul44(cam.invP, x, y, 0.0f, 1.0f);
                            float iw0 = 1.0f / Roh.w;
                            F3 Ro = f3(Roh.x * iw0, Roh.y * iw0, Roh.z * iw0);
                            F4 Rdh = mul44(cam.invP, x, y, 1.0f, 1.0f);
                            float iw1 = 1.0f / Rdh.w;
                            F3 Rd = f3(Rdh.x * iw1, Rdh.y * iw1, Rdh.z * iw1) - Ro;
                            if (prm.depth_of_field) {
                                F3 Rn = normalize(Rd);
                                F3 Pn = f3(cam.forward[0], cam.forward[1], cam.forward[2]);
                                float Pd = prm.whitted ? prm.focal_distance / dot(Rn, Pn) : prm.focal_distance / sel_max(0.01f, dot(Rn, Pn));  // :589 vs :272
                                F3 Rf = Ro + Pd * Rn;
                                float xr, yr;
                                float a = ps.R.getF32(-1.0f, 1.0f);
                                float b = ps.R.getF32(-1.0f, 1.0f);
                                map_to_disk(a, b, xr, yr);
                                F3 up = f3(cam.orient_col0[0], cam.orient_col0[1], cam.orient_col0[2]);
                                F3 right = f3(cam.orient_col1[0], cam.orient_col1[1], cam.orient_col1[2]);
                                Ro = Ro + prm.aperture * (yr * up + xr * right);
                                Rd = cam.far_ * normalize(Rf - Ro);
                            }
                            ps.sid = sid;
                            ps.base = sbase;
                            ps.mode = 0;
                            ps.bounce = 0;
                            ps.Ei = f3(0.0f, 0.0f, 0.0f);
                            tr.begin(TS, Ro, Rd, false);
                            n_rays += 1;
                            ps.stage = ST_EXT;
                        } else {
                            ps.stage = ST_EXT;  // placeholder so that this lane is not handed a second id in this while-loop pass
                            ps.sid = -2;
                        }
                    }
                    const int n_need = __popcll(need);
                    next = __builtin_amdgcn_readfirstlane(next + (uint32_t)(n_need < avail ? n_need : avail));
                    need = __ballot(ps.stage == ST_IDLE);
                }
                if (ps.sid == -2) { ps.stage = ST_IDLE; ps.sid = -1; }  // padded sample id: nothing to trace
#if PT_PRIO_NEW
                __builtin_amdgcn_s_setprio(0);
#endif
                PT_STAMP_END(5, nN);
                dirty = true;
            }
            prev_alive = __ballot(tr.alive);
        }
        {
            PT_STAMP_BEGIN();
#if PT_PRIO_STEP
            __builtin_amdgcn_s_setprio(PT_PRIO_STEP);
#endif
            // Several steps per scheduler iteration (the ballots and tests of the loop head are paid once for all of them), in a phase
            // of one kind: either the lanes that are at interior nodes step, or the lanes that are at triangles do (see PT_PHASE_BIAS_*).
            // A wave whose lanes are a mix of both would execute both halves of the step's code for every step; lanes of the other
            // kind wait a phase (idle lanes cost nothing, instructions do).
#ifndef PT_UNPHASED_STEPS
            {
                const bool node_lane = tr.alive && tr.at_node(), tri_lane = tr.alive && !tr.at_node();
#ifdef PT_STAMPS
                {   // lanes of the chosen kind at the phase's first step (word 29), node phases (30), node lanes + triangle lanes (31)
                    const int nn_ = __popcll(__ballot(node_lane)), nt_ = __popcll(__ballot(tri_lane));
                    const bool np_ = nn_ * PT_PHASE_BIAS_N >= nt_ * PT_PHASE_BIAS_T;
                    st_phase[0] += np_ ? nn_ : nt_; st_phase[1] += np_ ? 1 : 0; st_phase[2] += nn_ + nt_;
                }
#endif
                if (__popcll(__ballot(node_lane)) * PT_PHASE_BIAS_N >= __popcll(__ballot(tri_lane)) * PT_PHASE_BIAS_T) {
                    if (tr.alive && tr.at_node()) tr.step_node(TS);
#if PT_PHASE_NODE_REPS >= 2
                    if (tr.alive && tr.at_node()) tr.step_node(TS);
#endif
#if PT_PHASE_NODE_REPS >= 3
                    if (tr.alive && tr.at_node()) tr.step_node(TS);
#endif
#if PT_PHASE_NODE_REPS >= 4
                    if (tr.alive && tr.at_node()) tr.step_node(TS);
#endif
                } else {
                    if (tr.alive && !tr.at_node()) tr.step_triangle(TS);
#if PT_PHASE_TRI_REPS >= 2
                    if (tr.alive && !tr.at_node()) tr.step_triangle(TS);
#endif
#if PT_PHASE_TRI_REPS >= 3
                    if (tr.alive && !tr.at_node()) tr.step_triangle(TS);
#endif
                }
            }
#else
            // (written out: as an unrolled loop the same code compiled to a 25 % slower kernel)
            if (tr.alive) tr.step(TS);
#if PT_STEPS_PER_ITERATION >= 2
            if (tr.alive) tr.step(TS);
#endif
#if PT_STEPS_PER_ITERATION >= 3
            if (tr.alive) tr.step(TS);
#endif
#if PT_STEPS_PER_ITERATION >= 4
            if (tr.alive) tr.step(TS);
#endif
#endif
            PT_STAMP_END(6, __popcll(alive_mask));
#if PT_PRIO_STEP
            __builtin_amdgcn_s_setprio(0);
#endif
        }
    }
#ifdef PT_STAMPS
    if (lane == 0)
        for (int k = 0; k < 7; ++k) {
            atomicAdd(&counters[8 + 3 * k], (unsigned long long)st_cyc[k]);
            atomicAdd(&counters[9 + 3 * k], (unsigned long long)st_inv[k]);
            atomicAdd(&counters[10 + 3 * k], (unsigned long long)st_lanes[k]);
        }
    if (lane == 0)
        for (int k = 0; k < 3; ++k) atomicAdd(&counters[29 + k], (unsigned long long)st_phase[k]);
#endif
    // 1 ray = 1 raycast() call (reference src/RayTracer.cpp:192)
    unsigned long long wr = n_rays;
    for (int off = 32; off > 0; off >>= 1) wr += __shfl_down(wr, off);
    if (lane == 0 && wr) atomicAdd(&counters[0], wr);
    if (COUNT) {
        atomicAdd(&counters[1], (unsigned long long)tr.n_nodes);
        atomicAdd(&counters[2], (unsigned long long)tr.n_tris);
    }
}

}  // namespace ptd
