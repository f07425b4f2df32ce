// Linear BVH built on the device ("the step before traversal ... must produce a tree the same traversal kernel
// consumes ... as an option, not a replacement for parity runs").
//
// The reference only has CPU builders (src/Bvh.cpp:98-234: full-sweep SAH, spatial / object median), 13 s for 10 M triangles on 16
// host threads.  This one is Karras' construction ("Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees",
// HPG 2012) over 63-bit Morton codes of the triangles' box centres:
//   k_lbvh_prims      per-triangle boxes + bounds of the box centres (wave reduction, then ordered-integer atomics)
//   k_lbvh_morton     63-bit Morton key per triangle
//   rocprim::radix_sort_pairs   (key, triangle index)
//   k_lbvh_hierarchy  one thread per internal node: its key range and its two children
//   k_lbvh_fit        bottom-up: boxes and the size of each subtree after collapsing subtrees of <= max_tris_in_leaf triangles to leaves
//   k_lbvh_emit       every surviving node finds its depth-first pre-order index by walking to the root and writes a FlatNode
// The output is the reference's layout (src/BvhNode.hpp:67-113, src/Bvh.cpp:70-96): pre-order array, left child = index + 1, leaves hold
// inclusive ranges of the index list, so ptgpu_scene_create, every kernel and the CPU oracle take it like a tree from Bvh.cpp.
// The tree is NOT the reference's: which of two triangles with exactly equal t is reported, and which occluder an any-hit ray reports,
// follow this tree's order; and because the reference's slab test has no padding, a ray grazing a (zero-thickness) box face within
// rounding can be culled by one tree and not by another (1 of 8.3 M primary rays on the atrium scene).  Otherwise the closest hit does
// not depend on the tree and its record is the same floats.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

#include "../../include/ptgpu.h"

namespace ptl {

// float <-> unsigned with the same ordering (for atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float unord(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}

// bounds[0..2] = min of box centres, [3..5] = max (ordered-integer encoding), [6] = 1 if a vertex is not finite
__global__ __launch_bounds__(256) void k_lbvh_prims(const float* __restrict__ pos, int n, float4* __restrict__ lo, float4* __restrict__ hi,
                                                    uint32_t* bounds) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float c[3] = {0.0f, 0.0f, 0.0f};
    bool have = false, bad = false;
    if (i < n) {
        const float* p = pos + (size_t)i * 9;
        float mn[3], mx[3];
        for (int k = 0; k < 3; ++k) {
            const float a = p[k], b = p[3 + k], d = p[6 + k];
            mn[k] = fminf(a, fminf(b, d));
            mx[k] = fmaxf(a, fmaxf(b, d));
            if (!(fabsf(a) <= 3.402823466e+38f) || !(fabsf(b) <= 3.402823466e+38f) || !(fabsf(d) <= 3.402823466e+38f)) bad = true;
            c[k] = 0.5f * mn[k] + 0.5f * mx[k];
        }
        lo[i] = make_float4(mn[0], mn[1], mn[2], 0.0f);
        hi[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
        have = !bad;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&bounds[6], 1u);
    for (int k = 0; k < 3; ++k) {
        uint32_t vmin = have ? ord(c[k]) : 0xffffffffu, vmax = have ? ord(c[k]) : 0u;
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t a = __shfl_xor(vmin, off), b = __shfl_xor(vmax, off);
            vmin = a < vmin ? a : vmin;
            vmax = b > vmax ? b : vmax;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bounds[k], vmin);
            atomicMax(&bounds[3 + k], vmax);
        }
    }
}

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x001f00000000ffffull;
    x = (x | x << 16) & 0x001f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void k_lbvh_morton(const float4* __restrict__ lo, const float4* __restrict__ hi, int n, const uint32_t* __restrict__ bounds,
                                                     unsigned long long* __restrict__ keys, int32_t* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = lo[i], b = hi[i];
    const float c[3] = {0.5f * a.x + 0.5f * b.x, 0.5f * a.y + 0.5f * b.y, 0.5f * a.z + 0.5f * b.z};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float mn = unord(bounds[k]), mx = unord(bounds[3 + k]);
        const float ext = mx - mn;
        float t = ext > 0.0f ? (c[k] - mn) / ext : 0.0f;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        const double s = (double)t * 2097151.0;  // 2^21 - 1
        q[k] = (uint32_t)s;
    }
    keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    vals[i] = i;
}

// tree under construction: internal node k in [0, n-1); a child reference c >= 0 is an internal node, c < 0 is sorted primitive ~c
struct Tree {
    int n;
    const unsigned long long* keys;  // sorted
    int2* child;      // [n-1] {left, right}
    int2* range;      // [n-1] {first, last} sorted positions covered
    int32_t* parent;  // [n-1], -1 for the root (internal 0)
    int32_t* leaf_parent;  // [n]
    float4* box_lo;   // [n-1]
    float4* box_hi;
    int32_t* size;    // [n-1] nodes of the emitted subtree (1 when the node is collapsed to a leaf)
    uint32_t* arrived;  // [n-1]
};

// common-prefix length of the keys at sorted positions i and j; equal keys are told apart by their positions (Karras, section 4)
__device__ __forceinline__ int delta(const Tree& T, int i, int j) {
    if (j < 0 || j >= T.n) return -1;
    const unsigned long long x = T.keys[i] ^ T.keys[j];
    if (x == 0ull) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)x);
}

__global__ __launch_bounds__(256) void k_lbvh_hierarchy(Tree T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T.n - 1) return;
    const int d = (delta(T, i, i + 1) - delta(T, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(T, i, i - d);
    int lmax = 2;
    while (delta(T, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(T, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(T, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(T, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const int left = (first == gamma) ? ~gamma : gamma;
    const int right = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    T.child[i] = make_int2(left, right);
    T.range[i] = make_int2(first, last);
    if (left >= 0) T.parent[left] = i; else T.leaf_parent[~left] = i;
    if (right >= 0) T.parent[right] = i; else T.leaf_parent[~right] = i;
    if (i == 0) T.parent[0] = -1;
}

// bottom-up from every primitive; the second thread to arrive at a node owns it (acquire/release on the arrival counter orders the
// first thread's plain stores before the second thread's plain loads).  Node i's key range begins or ends at i (Karras), so a node whose
// range lies inside the 256 primitives of one workgroup is only ever reached by threads of that workgroup: its arrival counter lives in
// LDS (workgroup scope).  Only the nodes that span workgroups -- a few per workgroup -- take the agent-scope atomic in memory, which on
// this GPU (eight L2s, so agent-scope release / acquire reach past them) is the slow part: with every counter in memory the pass over
// 10 M triangles took 42 ms, with LDS counters for the local nodes 27.
__global__ __launch_bounds__(256) void k_lbvh_fit(Tree T, const int32_t* __restrict__ vals, const float4* __restrict__ prim_lo,
                                                  const float4* __restrict__ prim_hi, int max_leaf) {
    __shared__ uint32_t local_arrived[256];
    local_arrived[threadIdx.x] = 0u;
    __syncthreads();
    const int lo_p = blockIdx.x * 256, hi_p = lo_p + 255;
    const int p = lo_p + threadIdx.x;
    if (p >= T.n) return;
    int cur = T.leaf_parent[p];
    while (cur >= 0) {
        const int2 r = T.range[cur];
        uint32_t before;
        if (r.x >= lo_p && r.y <= hi_p)
            before = __hip_atomic_fetch_add(&local_arrived[cur - lo_p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            before = __hip_atomic_fetch_add(&T.arrived[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == 0u) return;
        const int2 ch = T.child[cur];
        float4 alo, ahi, blo, bhi;
        int sa, sb;
        if (ch.x >= 0) { alo = T.box_lo[ch.x]; ahi = T.box_hi[ch.x]; sa = T.size[ch.x]; }
        else { const int t = vals[~ch.x]; alo = prim_lo[t]; ahi = prim_hi[t]; sa = 1; }
        if (ch.y >= 0) { blo = T.box_lo[ch.y]; bhi = T.box_hi[ch.y]; sb = T.size[ch.y]; }
        else { const int t = vals[~ch.y]; blo = prim_lo[t]; bhi = prim_hi[t]; sb = 1; }
        T.box_lo[cur] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.0f);
        T.box_hi[cur] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.0f);
        T.size[cur] = (r.y - r.x + 1 <= max_leaf) ? 1 : 1 + sa + sb;
        cur = T.parent[cur];
    }
}

// out[0] = tree height (root = 1)
__global__ __launch_bounds__(256) void k_lbvh_emit(Tree T, const int32_t* __restrict__ vals, const float4* __restrict__ prim_lo,
                                                   const float4* __restrict__ prim_hi, int max_leaf, ptgpu_flatnode* __restrict__ nodes,
                                                   int32_t* out) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n_int = T.n - 1;
    if (g >= n_int + T.n) return;
    int ref, par, first, last;
    float4 lo, hi;
    if (g < n_int) {
        ref = g; par = T.parent[g];
        const int2 r = T.range[g];
        first = r.x; last = r.y;
        lo = T.box_lo[g]; hi = T.box_hi[g];
    } else {
        const int p = g - n_int;
        ref = ~p; par = T.leaf_parent[p];
        first = last = p;
        const int t = vals[p];
        lo = prim_lo[t]; hi = prim_hi[t];
    }
    // survives iff it is the root or its parent is not collapsed
    if (par >= 0) {
        const int2 pr = T.range[par];
        if (pr.y - pr.x + 1 <= max_leaf) return;
    }
    const bool leaf = last - first + 1 <= max_leaf;
    // pre-order index: one step per ancestor; being a right child adds the left sibling's whole subtree
    int idx = 0, level = 1, cur = ref, p = par;
    while (p >= 0) {
        const int2 ch = T.child[p];
        if (ch.y == cur) {
            int sl = 1;
            if (ch.x >= 0) sl = T.size[ch.x];
            idx += 1 + sl;
        } else idx += 1;
        cur = p;
        p = T.parent[p];
        ++level;
    }
    ptgpu_flatnode f;
    f.bmin[0] = lo.x; f.bmin[1] = lo.y; f.bmin[2] = lo.z;
    f.bmax[0] = hi.x; f.bmax[1] = hi.y; f.bmax[2] = hi.z;
    f.startIdx = first; f.endIdx = last;
    f.leaf = leaf ? 1 : 0;
    f._padleaf[0] = f._padleaf[1] = f._padleaf[2] = 0;
    f.rightChild = 0; f.axis = 0;
    f.leftChild = leaf ? 0 : idx + 1;
    f.triangles = last - first + 1;
    f.pad0 = f.pad1 = f.pad2 = 0;
    if (!leaf) {
        const int2 ch = T.child[ref];
        float4 alo, ahi, blo, bhi;
        int sl = 1;
        if (ch.x >= 0) { alo = T.box_lo[ch.x]; ahi = T.box_hi[ch.x]; sl = T.size[ch.x]; }
        else { const int t = vals[~ch.x]; alo = prim_lo[t]; ahi = prim_hi[t]; }
        if (ch.y >= 0) { blo = T.box_lo[ch.y]; bhi = T.box_hi[ch.y]; }
        else { const int t = vals[~ch.y]; blo = prim_lo[t]; bhi = prim_hi[t]; }
        f.rightChild = idx + 1 + sl;
        // traversal enters the child on the near side of `axis` first (RayTracer.cpp:263-271): the axis along which the right child's
        // box centre lies furthest beyond the left child's
        const float dx = (blo.x + bhi.x) - (alo.x + ahi.x), dy = (blo.y + bhi.y) - (alo.y + ahi.y), dz = (blo.z + bhi.z) - (alo.z + ahi.z);
        f.axis = (dx >= dy && dx >= dz) ? 0 : (dy >= dz ? 1 : 2);
    } else {
        atomicMax(&out[0], level);
    }
    nodes[idx] = f;
}

}  // namespace ptl
