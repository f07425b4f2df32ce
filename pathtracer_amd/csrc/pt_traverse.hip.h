// BVH traversal + Woop ray/triangle test for gfx950, one ray per lane (wave64).
//
// Restates RayTracer::raycast (reference src/RayTracer.cpp:174-290) with FlatNode::intersect
// (src/BvhNode.cpp:9-57) and RTTriangle::intersect_woop (src/RTTriangle.hpp:50-59) so that the
// hit record (triangle, t, u, v) is bit-identical to the CPU code:
//
//  * same arithmetic, same operation order, no FMA contraction, IEEE divide;
//  * same node order: the child on the near side of the split axis first (sign[node.axis],
//    RayTracer.cpp:263-271), leaves scanned startIdx..endIdx inclusive, a hit replaces the
//    current one only if t > 0 && t < tmin (ties keep the first, :227), any-hit returns on the
//    first accepted triangle (:235-238).
//
// Device layout (built by ptgpu_scene_create from the reference-layout 64-byte node array):
//  * interior node = 64 B holding BOTH children's boxes and references, so one fetch decides
//    two slab tests and a failing child never costs a dependent memory round trip.  The slab
//    test of the far child is evaluated when its parent is visited; its entry distance is kept
//    on the stack and re-compared with the ray's current tmin when popped.  That is exactly
//    the test the reference performs at pop time, because only the final "tmin_box < t1"
//    clause of FlatNode::intersect depends on t1 (BvhNode.cpp:56) and t1 only shrinks.
//  * triangle record = 64 B in LEAF ORDER (no bvh.list indirection while scanning a leaf):
//    three rows (M row | N) of the Woop matrix, then {triangle id, triangles remaining in this
//    leaf, alpha-texture id or -1, material id}.
//  * per-lane stack of {node ref, entry distance} pairs: the first NLDS entries live in LDS laid
//    out [entry][lane] (conflict-free ds_read_b64/ds_write_b64), deeper entries spill to a
//    per-wave region of global memory laid out the same way (rarely touched).  Depth is bounded by the validated tree depth (<= 64 like the reference's
//    int nodesToVisit[64]).
#pragma once
#include "pt_device.hip.h"

namespace ptd {

struct DevTexture {
    const uint8_t* rgba;
    int32_t width, height;
};

struct TravScene {
    // ONE array of 64-byte records addressed by the signed reference itself: interior node i at records[i], triangle record p (position
    // in bvh.list) at records[~p] = records[-(p + 1)], i.e. the triangle records lie in reverse order below the interior ones.  A step's
    // fetch address is records + ref * 64 whatever the reference is (no select between two bases), and "next triangle of this leaf"
    // stays ref - 1.
    const float4* records;
    float root_bmin[3], root_bmax[3];
    int32_t root_ref;     // >= 0: interior node index; < 0: leaf starting at triangle record ~root_ref
    int32_t list_size;    // 0 => empty scene, every ray misses
    const float* uv;      // [num_tris][3][2], for the alpha cutout only
    const DevTexture* textures;
    int32_t alpha_on;     // RayTracer::useAlpha && the scene has an alpha map (only read by ALPHA instantiations)
};

struct HitRec {
    int32_t id;  // caller's triangle index, -1 = miss
    float t, u, v;
};

__device__ __forceinline__ float4 ldg4(const float4* p) { return *p; }
// the 4 x float4 record of the triangle at position p of bvh.list
__device__ __forceinline__ const float4* tri_record(const TravScene& S, int p) { return S.records + (ptrdiff_t)(~p) * 4; }

// One child's slab test.  near*/far* are the box planes already selected by the ray's sign bits
// (bounds[sign] / bounds[1-sign] in the reference).  Returns the t1-independent part of the result
// in `ok` (both overlap checks and tmax > 0) and the entry distance in `tentry`.
__device__ __forceinline__ void slab(float nx, float ny, float nz, float fx, float fy, float fz, F3 o, F3 inv,
                                     bool& ok, float& tentry) {
    float tmin = (nx - o.x) * inv.x;
    float tmax = (fx - o.x) * inv.x;
    float tymin = (ny - o.y) * inv.y;
    float tymax = (fy - o.y) * inv.y;
    bool miss = (tmin > tymax) || (tymin > tmax);
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (nz - o.z) * inv.z;
    float tzmax = (fz - o.z) * inv.z;
    miss = miss || (tmin > tzmax) || (tzmin > tmax);
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    ok = !miss && (tmax > 0.0f);
    tentry = tmin;
}

// slab_fast2: the same test for rays whose arithmetic cannot produce a NaN (every 1/dir component finite and non-zero, origin finite; box
// planes are validated NaN-free at scene creation): with near = min and far = max of the two plane distances per axis, the
// reference's compare chain reduces to  max3(near) <= min3(far)  (its misses are exactly "some near_i > some far_j", i != j, and
// near_i <= far_i always), its tmax is min3(far) and its tmin is max3(near) -- the same floats, because min/max return one of
// their operands.  Rays with a zero, infinite or NaN direction component keep slab(): there 0 * inf = NaN decides the outcome
// through comparisons that are false, which min/max would not reproduce.
// Both children at once, on packed fp32 pairs (v_pk_add_f32 / v_pk_mul_f32: the same IEEE subtract and multiply, two per instruction).
// Interior record: a = (L.lo.x, L.lo.y, L.hi.x, L.hi.y), b = the same for R, c = (L.lo.z, L.hi.z, R.lo.z, R.hi.z).
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void slab_fast2(float4 a, float4 b, float4 c, float2v oxy, float2v ixy, float oz, float iz,
                                           bool& okL, float& teL, bool& okR, float& teR) {
    const float2v ozz = {oz, oz}, izz = {iz, iz};
    const float2v l0 = (float2v{a.x, a.y} - oxy) * ixy, l1 = (float2v{a.z, a.w} - oxy) * ixy;
    const float2v r0 = (float2v{b.x, b.y} - oxy) * ixy, r1 = (float2v{b.z, b.w} - oxy) * ixy;
    const float2v lz = (float2v{c.x, c.y} - ozz) * izz, rz = (float2v{c.z, c.w} - ozz) * izz;
    teL = fmaxf(fmaxf(fminf(l0.x, l1.x), fminf(l0.y, l1.y)), fminf(lz.x, lz.y));
    const float txL = fminf(fminf(fmaxf(l0.x, l1.x), fmaxf(l0.y, l1.y)), fmaxf(lz.x, lz.y));
    okL = (teL <= txL) && (txL > 0.0f);
    teR = fmaxf(fmaxf(fminf(r0.x, r1.x), fminf(r0.y, r1.y)), fminf(rz.x, rz.y));
    const float txR = fminf(fminf(fmaxf(r0.x, r1.x), fmaxf(r0.y, r1.y)), fmaxf(rz.x, rz.y));
    okR = (teR <= txR) && (txR > 0.0f);
}

// What happens when a lane needs more than NLDS entries:
//   OVF    entries NLDS.. live in a per-wave region of global memory (persistent kernels: a bounded set of waves);
//   DEFER  nothing is written; `full` is raised and the traverser gives the ray up, for the caller to re-trace it with an
//          OVF kernel (one-ray-per-lane grids, which have no bounded set of waves to own overflow regions);
//   neither: the host has proved from the tree depth that NLDS entries suffice.
// The two regions are addressed through address-space-qualified pointers so that every access is a plain ds_* or global_*
// instruction: with generic pointers the compiler merges the two cases of push/pop into one flat access on a selected pointer (slower,
// and it split the 8-byte pop into two dependent round trips).  An entry travels as one 64-bit word: reference low, entry distance high.
typedef __attribute__((address_space(3))) unsigned long long LdsWord;
typedef __attribute__((address_space(1))) unsigned long long GlobalWord;
template <int NLDS, bool OVF, bool DEFER = false>
struct LaneStack {
    LdsWord* lds;    // this lane's slot of entry 0; entry e is lds[e * 64]
    GlobalWord* gl;  // OVF only: this lane's slot of overflow entry 0 in global memory; entry e is gl[e * 64]
    int sp;
    bool full;   // DEFER only
    __device__ __forceinline__ void attach(uint2* lds_lane, uint2* gl_lane) {
        // a generic pointer into LDS is the aperture base plus the LDS offset: its low 32 bits are the address_space(3) pointer
        lds = (LdsWord*)(uintptr_t)(uint32_t)(uintptr_t)lds_lane;
        gl = (GlobalWord*)(uintptr_t)gl_lane;
        sp = 0;
        full = false;
    }
    __device__ __forceinline__ void push(int ref, float tentry) {
        const unsigned long long e = ((unsigned long long)__float_as_uint(tentry) << 32) | (unsigned long long)(uint32_t)ref;
        if (DEFER) {
            if (sp < NLDS) lds[sp * 64] = e; else full = true;
        } else if (!OVF || __builtin_expect(sp < NLDS, 1)) lds[sp * 64] = e;
        else gl[(sp - NLDS) * 64] = e;
        ++sp;
    }
    __device__ __forceinline__ uint2 pop() {
        --sp;
        unsigned long long e;
        if (!OVF || __builtin_expect(sp < NLDS, 1)) e = lds[sp * 64];
        else e = gl[(sp - NLDS) * 64];
        return make_uint2((uint32_t)e, (uint32_t)(e >> 32));
    }
};
// bytes of global overflow space one wave needs (entries NLDS..63 for 64 lanes)
template <int NLDS> constexpr size_t stack_overflow_bytes_per_wave() { return (size_t)(64 - NLDS) * 64 * sizeof(uint2); }

// ALPHA: alpha-texture cutout compiled in (and enabled by TravScene::alpha_on).
// COUNT: also count the reference's node visits / triangle tests (far children are then always pushed so
// that the counts equal the CPU traversal's exactly; results are unchanged).

// ------------------------------------------------------------------------------------------------------------------
// Step-wise traverser: the reference's traversal cut into single steps so that a wave never waits on
// its slowest lane inside a phase.  Node records and triangle records are both 64 B, so ONE 4 x dwordx4 fetch per step
// serves every lane whatever it is doing (visiting an interior node or testing one triangle); the two arithmetic
// bodies then run under their own exec masks.  A lane that finishes can be handed a new ray by its kernel between two
// steps (persistent waves with refill).
// ------------------------------------------------------------------------------------------------------------------
template <bool ALPHA, bool COUNT, int NLDS, bool OVF, bool DEFER = false>
struct Trav {
    F3 o, d, inv;
    uint32_t signbits;
    float tmin, umin, vmin;
    int pmin;
    int cur;
    bool alive, anyhit;
    bool fast;  // no NaN can arise in this ray's slab tests: slab_fast applies
    LaneStack<NLDS, OVF, DEFER> st;
    uint32_t n_nodes, n_tris;

    __device__ __forceinline__ void reset_counts() { n_nodes = 0; n_tris = 0; }

    __device__ __forceinline__ void attach(uint2* lds_lane, uint2* gl_lane) { st.attach(lds_lane, gl_lane); }
    // DEFER: the ray ran out of LDS stack and was given up (alive is false, no result): re-trace it with an OVF kernel
    __device__ __forceinline__ bool deferred() const { return DEFER && st.full; }

    // begin() in two halves, for a caller that prepares a ray while another one is still in flight: prepare() does everything that
    // costs arithmetic (reciprocal direction, sign bits, the NaN-freedom flag, the root-box test), launch() installs the result.
    struct Prepared {
        F3 inv;
        uint32_t bits;  // 0-2 sign of 1/dir, 3 fast, 4 root box entered, 5 scene not empty
    };
    __device__ __forceinline__ static Prepared prepare(const TravScene& S, F3 o_, F3 d_) {
        Prepared p;
        p.inv = f3(1.0f / d_.x, 1.0f / d_.y, 1.0f / d_.z);
        const bool sx = p.inv.x < 0.0f, sy = p.inv.y < 0.0f, sz = p.inv.z < 0.0f;
        p.bits = (sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u);
        {   // x * 0 == 0 exactly for finite x only; != 0 excludes 1/inf
            const float z = p.inv.x * 0.0f + p.inv.y * 0.0f + p.inv.z * 0.0f + o_.x * 0.0f + o_.y * 0.0f + o_.z * 0.0f;
            if ((z == 0.0f) && p.inv.x != 0.0f && p.inv.y != 0.0f && p.inv.z != 0.0f) p.bits |= 8u;
        }
        if (S.list_size > 0) {
            bool ok; float te;
            slab(sx ? S.root_bmax[0] : S.root_bmin[0], sy ? S.root_bmax[1] : S.root_bmin[1], sz ? S.root_bmax[2] : S.root_bmin[2],
                 sx ? S.root_bmin[0] : S.root_bmax[0], sy ? S.root_bmin[1] : S.root_bmax[1], sz ? S.root_bmin[2] : S.root_bmax[2],
                 o_, p.inv, ok, te);
            if (ok && (te < 1.0f)) p.bits |= 16u;  // 1.0f: the ray's initial tmin
            p.bits |= 32u;
        }
        return p;
    }
    __device__ __forceinline__ void launch(const TravScene& S, F3 o_, F3 d_, const Prepared& p, bool anyhit_) {
        o = o_; d = d_; anyhit = anyhit_;
        tmin = 1.0f; umin = 0.0f; vmin = 0.0f; pmin = -1;
        inv = p.inv;
        signbits = p.bits & 7u;
        fast = (p.bits & 8u) != 0u;
        st.sp = 0;
        st.full = false;
        cur = S.root_ref;
        alive = (p.bits & 16u) != 0u;
        if (COUNT && (p.bits & 32u)) n_nodes += 1;
    }
    __device__ __forceinline__ void begin(const TravScene& S, F3 o_, F3 d_, bool anyhit_) { launch(S, o_, d_, prepare(S, o_, d_), anyhit_); }

    __device__ __forceinline__ void pop_next() {
        alive = false;
        while (st.sp > 0) {
            uint2 e = st.pop();
            if (COUNT) n_nodes += 1;
            if (__uint_as_float(e.y) < tmin) { cur = (int)e.x; alive = true; break; }
        }
    }

    // interior-node half of a step, given the node's record
    __device__ __forceinline__ void visit_node(float4 a, float4 b, float4 c, float4 m) {
        const bool sx = (signbits & 1u) != 0u, sy = (signbits & 2u) != 0u, sz = (signbits & 4u) != 0u;
        bool okL, okR; float teL, teR;
        if (fast) {
            slab_fast2(a, b, c, float2v{o.x, o.y}, float2v{inv.x, inv.y}, o.z, inv.z, okL, teL, okR, teR);
        } else {
            slab(sx ? a.z : a.x, sy ? a.w : a.y, sz ? c.y : c.x, sx ? a.x : a.z, sy ? a.y : a.w, sz ? c.x : c.y, o, inv, okL, teL);
            slab(sx ? b.z : b.x, sy ? b.w : b.y, sz ? c.w : c.z, sx ? b.x : b.z, sy ? b.y : b.w, sz ? c.z : c.w, o, inv, okR, teR);
        }
        int refL = __float_as_int(m.x), refR = __float_as_int(m.y);
        bool rightFirst = ((signbits >> __float_as_int(m.z)) & 1u) != 0u;
        int refN = rightFirst ? refR : refL, refF = rightFirst ? refL : refR;
        bool okN = rightFirst ? okR : okL, okF = rightFirst ? okL : okR;
        float teN = rightFirst ? teR : teL, teF = rightFirst ? teL : teR;
        bool passN = okN && (teN < tmin);
        bool passF = okF && (teF < tmin);
        if (COUNT) {
            n_nodes += 1;
            st.push(refF, passF ? teF : __uint_as_float(0x7f800000u));
            passF = false;
        }
        if (passN) {
            if (passF) st.push(refF, teF);
            cur = refN;
        } else if (passF) {
            cur = refF;
        } else {
            pop_next();
        }
        if (DEFER && st.full) alive = false;
    }
    // triangle half of a step, given the triangle's record
    __device__ __forceinline__ void test_triangle(const TravScene& S, float4 a, float4 b, float4 c, float4 m) {
        if (COUNT) n_tris += 1;
        float ox = (a.x * o.x + a.y * o.y + a.z * o.z) + a.w;
        float oy = (b.x * o.x + b.y * o.y + b.z * o.z) + b.w;
        float oz = (c.x * o.x + c.y * o.y + c.z * o.z) + c.w;
        float dx = a.x * d.x + a.y * d.y + a.z * d.z;
        float dy = b.x * d.x + b.y * d.y + b.z * d.z;
        float dz = c.x * d.x + c.y * d.y + c.z * d.z;
        float t = -oz / dz;
        float u = ox + dx * t;
        float v = oy + dy * t;
        bool done = false;
        if (u > 0.0f && v > 0.0f && u + v < 1.0f) {
            bool keep = true;
            if (ALPHA) {
                int atex = __float_as_int(m.z);
                if (S.alpha_on && atex >= 0) {
                    const float* tc = S.uv + (size_t)__float_as_int(m.x) * 6;
                    F2 uv = (1 - u - v) * ld2(tc) + u * ld2(tc + 2) + v * ld2(tc + 4);
                    DevTexture tx = S.textures[atex];
                    int x = (int)((uv.x - floorf(uv.x)) * tx.width);
                    int y = (int)((uv.y - floorf(uv.y)) * tx.height);
                    x = x < 0 ? 0 : (x > tx.width - 1 ? tx.width - 1 : x);
                    y = y < 0 ? 0 : (y > tx.height - 1 ? tx.height - 1 : y);
                    float alpha = (float)tx.rgba[((size_t)y * tx.width + x) * 4 + 1] / 255.0f;
                    keep = !(alpha < 0.5f);
                }
            }
            if (keep && t > 0.0f && t < tmin) {
                pmin = ~cur; tmin = t; umin = u; vmin = v;
                done = anyhit;
            }
        }
        if (done) alive = false;
        else if (__float_as_int(m.y) != 0) cur -= 1;  // next record of this leaf: ~(pos + 1) == cur - 1
        else pop_next();
    }

    // one step for a lane with alive == true
    __device__ __forceinline__ void step(const TravScene& S) {
        const float4* rp = S.records + (ptrdiff_t)cur * 4;
        float4 a = ldg4(rp), b = ldg4(rp + 1), c = ldg4(rp + 2), m = ldg4(rp + 3);
        const bool sx = (signbits & 1u) != 0u, sy = (signbits & 2u) != 0u, sz = (signbits & 4u) != 0u;
        if (cur >= 0) {
            bool okL, okR; float teL, teR;
            if (fast) {
                slab_fast2(a, b, c, float2v{o.x, o.y}, float2v{inv.x, inv.y}, o.z, inv.z, okL, teL, okR, teR);
            } else {
                slab(sx ? a.z : a.x, sy ? a.w : a.y, sz ? c.y : c.x, sx ? a.x : a.z, sy ? a.y : a.w, sz ? c.x : c.y, o, inv, okL, teL);
                slab(sx ? b.z : b.x, sy ? b.w : b.y, sz ? c.w : c.z, sx ? b.x : b.z, sy ? b.y : b.w, sz ? c.z : c.w, o, inv, okR, teR);
            }
            int refL = __float_as_int(m.x), refR = __float_as_int(m.y);
            bool rightFirst = ((signbits >> __float_as_int(m.z)) & 1u) != 0u;
            int refN = rightFirst ? refR : refL, refF = rightFirst ? refL : refR;
            bool okN = rightFirst ? okR : okL, okF = rightFirst ? okL : okR;
            float teN = rightFirst ? teR : teL, teF = rightFirst ? teL : teR;
            bool passN = okN && (teN < tmin);
            bool passF = okF && (teF < tmin);
            if (COUNT) {
                n_nodes += 1;
                st.push(refF, passF ? teF : __uint_as_float(0x7f800000u));
                passF = false;
            }
            if (passN) {
                if (passF) st.push(refF, teF);
                cur = refN;
            } else if (passF) {
                cur = refF;
            } else {
                pop_next();
            }
            if (DEFER && st.full) alive = false;
        } else {
            if (COUNT) n_tris += 1;
            float ox = (a.x * o.x + a.y * o.y + a.z * o.z) + a.w;
            float oy = (b.x * o.x + b.y * o.y + b.z * o.z) + b.w;
            float oz = (c.x * o.x + c.y * o.y + c.z * o.z) + c.w;
            float dx = a.x * d.x + a.y * d.y + a.z * d.z;
            float dy = b.x * d.x + b.y * d.y + b.z * d.z;
            float dz = c.x * d.x + c.y * d.y + c.z * d.z;
            float t = -oz / dz;
            float u = ox + dx * t;
            float v = oy + dy * t;
            bool done = false;
            if (u > 0.0f && v > 0.0f && u + v < 1.0f) {
                bool keep = true;
                if (ALPHA) {
                    int atex = __float_as_int(m.z);
                    if (S.alpha_on && atex >= 0) {
                        const float* tc = S.uv + (size_t)__float_as_int(m.x) * 6;
                        F2 uv = (1 - u - v) * ld2(tc) + u * ld2(tc + 2) + v * ld2(tc + 4);
                        DevTexture tx = S.textures[atex];
                        int x = (int)((uv.x - floorf(uv.x)) * tx.width);
                        int y = (int)((uv.y - floorf(uv.y)) * tx.height);
                        x = x < 0 ? 0 : (x > tx.width - 1 ? tx.width - 1 : x);
                        y = y < 0 ? 0 : (y > tx.height - 1 ? tx.height - 1 : y);
                        float alpha = (float)tx.rgba[((size_t)y * tx.width + x) * 4 + 1] / 255.0f;
                        keep = !(alpha < 0.5f);
                    }
                }
                if (keep && t > 0.0f && t < tmin) {
                    pmin = ~cur; tmin = t; umin = u; vmin = v;
                    done = anyhit;
                }
            }
            if (done) alive = false;
            else if (__float_as_int(m.y) != 0) cur -= 1;  // next record of this leaf: ~(pos + 1) == cur - 1
            else pop_next();
        }
    }
    // (step() above stays one body of its own: routed through visit_node / test_triangle the one-ray-per-lane kernels compiled 4 % slower)
    // the same for callers that run the two kinds of step in separate, wave-uniform phases (pt_integrator.hip.h): a wave whose lanes
    // are all at interior nodes, or all at triangles, executes one half of step()'s code instead of both
    __device__ __forceinline__ bool at_node() const { return cur >= 0; }
    __device__ __forceinline__ void step_node(const TravScene& S) {
        const float4* rp = S.records + (ptrdiff_t)cur * 4;
        visit_node(ldg4(rp), ldg4(rp + 1), ldg4(rp + 2), ldg4(rp + 3));
    }
    __device__ __forceinline__ void step_triangle(const TravScene& S) {
        const float4* rp = S.records + (ptrdiff_t)cur * 4;
        test_triangle(S, ldg4(rp), ldg4(rp + 1), ldg4(rp + 2), ldg4(rp + 3));
    }

    __device__ __forceinline__ HitRec result(const TravScene& S) const {
        HitRec h;
        h.id = -1; h.t = 3.402823466e+38f; h.u = 0.0f; h.v = 0.0f;
        if (pmin >= 0) {
            h.id = __float_as_int(tri_record(S, pmin)[3].x);
            h.t = tmin; h.u = umin; h.v = vmin;
        }
        return h;
    }
};

}  // namespace ptd
