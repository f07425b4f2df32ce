// BVH with the reference's public surface (reference src/Bvh.hpp:11-54, src/BvhNode.hpp:16-113): a pointer tree built by
// SAH full sweep or spatial/object median, flattened depth-first into 64-byte FlatNodes (left child = index + 1).
// The flattened array and the index list are what ptgpu_scene_create receives, as-is.
//
// Unlike the reference, independent subtrees are built on several host threads (SURVEY.md 7 "Builder cost"): every
// construct() call only permutes its own [startIdx, endIdx] slice of `list`, so the resulting tree is identical to the
// serial build node for node (tests compare it with the oracle's serial restatement).
#pragma once
#include <atomic>
#include <memory>
#include <vector>

#include "RTTriangle.hpp"

namespace FW {

struct AABB {
    Vec3f min, max;
    AABB() {}
    AABB(const Vec3f& mn, const Vec3f& mx) : min(mn), max(mx) {}
    F32 area() const { Vec3f d(max - min); return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }
    int MaximumExtent() const {
        Vec3f d(max - min);
        if (d.x > d.y && d.x > d.z) return 0;
        return (d.y > d.z) ? 1 : 2;
    }
};

struct Node {
    AABB box;
    int32_t startIdx = 0, endIdx = 0;
    bool leaf = false;
    int32_t axis = 0;
    std::unique_ptr<Node> leftChild, rightChild;
};

// Byte layout fixed by the reference (sizeof == 64): box 24, startIdx, endIdx, leaf (bool padded to 4), rightChild, axis,
// leftChild, triangles, 3 pad ints.
struct FlatNode {
    AABB box;
    int32_t startIdx = 0, endIdx = 0;
    bool leaf = false;
    int32_t rightChild = 0;
    int32_t axis = 0;
    int32_t leftChild = 0;
    int32_t triangles = 0;
    int32_t pad0 = 0, pad1 = 0, pad2 = 0;
    FlatNode() {}
    FlatNode(const AABB& b, int32_t start, int32_t end, bool isLeaf, int32_t ax)
        : box(b), startIdx(start), endIdx(end), leaf(isLeaf), axis(ax), triangles(end - start + 1) {}
    FlatNode(const AABB& b, int32_t start, int32_t end, int32_t right, bool isLeaf, int32_t ax)
        : box(b), startIdx(start), endIdx(end), leaf(isLeaf), rightChild(right), axis(ax), triangles(end - start + 1) {}
    // Williams et al. slab test, as the device kernels evaluate it (kept on the host for tests and tools)
    bool intersect(const Vec3f& orig, const Vec3f& invDir, const int sign[3], const float& t1) const;
};
static_assert(sizeof(FlatNode) == 64, "FlatNode must stay 64 bytes: it is uploaded as-is");

class BVH {
public:
    BVH() {}
    BVH(std::vector<RTTriangle>* triangles, bool sah, int numThreads = 0);  // numThreads 0 = hardware concurrency
    BVH(std::vector<RTTriangle>* triangles, std::vector<int32_t> list_, std::vector<FlatNode> nodes_, int depth_)
        : list(std::move(list_)), nodes(std::move(nodes_)), depth(depth_), m_triangles(triangles) {}

    void construct(std::unique_ptr<Node>& node, int startIdx, int endIdx);
    void flatten(std::unique_ptr<Node>& node);
    AABB computeBB(int startIdx, int endIdx);
    AABB Union(AABB a, AABB b);
    AABB Union(AABB a, Vec3f b);

    std::unique_ptr<Node> root;
    std::vector<int32_t> list;    // triangle indices, permuted by the build
    std::vector<FlatNode> nodes;  // depth-first pre-order
    int32_t depth = 0;            // counts splits, like the reference's field (not the tree height)

    int32_t numBadSplits() const { return badSplits; }
    int32_t numTotalSplits() const { return totalSplits; }
    int32_t trueDepth() const;  // height of the flattened tree (root = 1)
    double buildMillis = 0.0, flattenMillis = 0.0;

private:
    void constructTask(std::unique_ptr<Node>& node, int startIdx, int endIdx, int spawnBudget);
    bool m_sahEnabled = true;
    int32_t badSplits = 0, totalSplits = 0;
    int32_t maxTrisInLeaf = 3;
    std::vector<RTTriangle>* m_triangles = nullptr;
    // per-triangle values the sweep needs over and over (same floats the reference recomputes on every comparison)
    std::vector<Vec3f> m_centroid, m_min, m_max;
    // split counters bumped from several builder threads; movable so that BVH stays assignable like the reference's
    struct Counter {
        std::atomic<int32_t> v{0};
        Counter() {}
        Counter(Counter&& o) noexcept : v(o.v.load()) {}
        Counter& operator=(Counter&& o) noexcept { v = o.v.load(); return *this; }
        void operator++(int) { v.fetch_add(1, std::memory_order_relaxed); }
        operator int32_t() const { return v.load(); }
    };
    Counter m_badSplitsA, m_totalSplitsA;
    int m_numThreads = 1;
};

}  // namespace FW
