// ORACLE -- TEST INFRASTRUCTURE ONLY (see pt_math.h header note).
//
// Serial restatement of the reference's BVH builder and flattener
// (reference src/Bvh.cpp:11-274, src/BvhNode.hpp:48-113, src/RTTriangle.hpp:35-47).
// Uses libstdc++'s std::sort / std::nth_element / std::partition exactly where the
// reference calls parallel_sort / nth_element / partition, with the reference's
// end-exclusive iterator quirk (the last element of a range is never sorted,
// Bvh.cpp:125,174,206,215) preserved.  MSVC's algorithms order ties differently, so
// a Windows build's tree is not reproducible; the counts the reference printed for
// head.obj when compiled with libstdc++ are what pins this file.
#include <algorithm>
#include <cfloat>
#include <memory>
#include <vector>

#include "pt_core.h"

namespace pto {

struct AABB {
    V3 mn, mx;
    float area() const {  // BvhNode.hpp:20-23
        V3 d = mx - mn;
        return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    int maximumExtent() const {  // BvhNode.hpp:24-32
        V3 d = mx - mn;
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        else return 2;
    }
};

struct Node {  // BvhNode.hpp:48-64
    AABB box;
    int startIdx = 0, endIdx = 0;
    bool leaf = false;
    int axis = 0;
    std::unique_ptr<Node> left, right;
};

struct Builder {
    const float* pos;  // [n][3][3]
    bool sah;
    std::vector<int> list;
    std::vector<pto_flatnode> nodes;
    int depthField = 0, badSplits = 0, totalSplits = 0;
    const int maxTrisInLeaf = 3;

    V3 vtx(int tri, int k) const { return ld3(pos + (size_t)tri * 9 + k * 3); }
    V3 tmin(int t) const {  // RTTriangle::min (FW::min of three Vec3f, componentwise)
        V3 a = vtx(t, 0), b = vtx(t, 1), c = vtx(t, 2);
        return V3{fmin3(a.x, b.x, c.x), fmin3(a.y, b.y, c.y), fmin3(a.z, b.z, c.z)};
    }
    V3 tmax(int t) const {
        V3 a = vtx(t, 0), b = vtx(t, 1), c = vtx(t, 2);
        return V3{fmax3(a.x, b.x, c.x), fmax3(a.y, b.y, c.y), fmax3(a.z, b.z, c.z)};
    }
    V3 centroid(int t) const { return (vtx(t, 0) + vtx(t, 1) + vtx(t, 2)) * (1.0f / 3.0f); }  // RTTriangle.hpp:39

    static AABB unite(AABB a, AABB b) {  // Bvh.cpp:264-268
        return AABB{V3{fmin2(a.mn.x, b.mn.x), fmin2(a.mn.y, b.mn.y), fmin2(a.mn.z, b.mn.z)},
                    V3{fmax2(a.mx.x, b.mx.x), fmax2(a.mx.y, b.mx.y), fmax2(a.mx.z, b.mx.z)}};
    }
    static AABB unite(AABB b, V3 p) {  // Bvh.cpp:270-274
        return AABB{V3{fmin2(b.mn.x, p.x), fmin2(b.mn.y, p.y), fmin2(b.mn.z, p.z)},
                    V3{fmax2(b.mx.x, p.x), fmax2(b.mx.y, p.y), fmax2(b.mx.z, p.z)}};
    }
    AABB computeBB(int s, int e) const {  // Bvh.cpp:237-261
        V3 minV = tmin(list[s]), maxV = tmax(list[s]);
        for (int i = s + 1; i <= e; ++i) {
            V3 a = tmin(list[i]), b = tmax(list[i]);
            minV = V3{fmin2(minV.x, a.x), fmin2(minV.y, a.y), fmin2(minV.z, a.z)};
            maxV = V3{fmax2(maxV.x, b.x), fmax2(maxV.y, b.y), fmax2(maxV.z, b.z)};
        }
        return AABB{minV, maxV};
    }

    void construct(std::unique_ptr<Node>& node, int startIdx, int endIdx) {  // Bvh.cpp:98-234
        node->box = computeBB(startIdx, endIdx);
        node->startIdx = startIdx;
        node->endIdx = endIdx;
        int tris = endIdx - startIdx + 1;
        if (tris > maxTrisInLeaf) {
            int dim = 0;
            int mid = (startIdx + endIdx) / 2;
            if (sah) {
                float area = 1.0f / node->box.area();
                float minCost = FLT_MAX;
                std::vector<float> la(tris - 1), ra(tris - 1);
                for (int d = 0; d <= 2; d++) {
                    std::sort(list.begin() + startIdx, list.begin() + endIdx,
                              [this, d](int x, int y) { return centroid(x)[d] < centroid(y)[d]; });
                    AABB bl{tmin(list[startIdx]), tmax(list[startIdx])};
                    la[0] = bl.area();
                    AABB br{tmin(list[endIdx]), tmax(list[endIdx])};
                    ra[0] = br.area();
                    for (int i = 1; i < tris - 1; ++i) {
                        bl = unite(bl, AABB{tmin(list[startIdx + i]), tmax(list[startIdx + i])});
                        la[i] = bl.area();
                        br = unite(br, AABB{tmin(list[endIdx - i]), tmax(list[endIdx - i])});
                        ra[i] = br.area();
                    }
                    for (int i = 0; i < tris - 1; ++i) {
                        int n_l = i + 1, n_r = tris - 1 - i;
                        float cost = 1.0f / maxTrisInLeaf + (n_l * la[i] + n_r * ra[n_r - 1]) * area;
                        if (cost < minCost) { minCost = cost; dim = d; mid = startIdx + i + 1; }
                    }
                }
                if (minCost < tris) {
                    node->axis = dim;
                    totalSplits += 1;
                    depthField += 1;
                    if (mid == startIdx || mid == endIdx) { badSplits += 1; mid = (startIdx + endIdx) / 2; }
                    std::nth_element(list.begin() + startIdx, list.begin() + mid, list.begin() + endIdx,
                                     [this, dim](int x, int y) { return centroid(x)[dim] < centroid(y)[dim]; });
                    node->left = std::make_unique<Node>();
                    construct(node->left, startIdx, mid - 1);
                    node->right = std::make_unique<Node>();
                    construct(node->right, mid, endIdx);
                } else {
                    node->leaf = true;
                    return;
                }
            } else {
                AABB cb{centroid(list[startIdx]), centroid(list[startIdx])};
                for (int i = startIdx + 1; i <= endIdx; ++i) cb = unite(cb, centroid(list[i]));
                dim = cb.maximumExtent();
                node->axis = dim;
                float split = 0.5f * (cb.mn[dim] + cb.mx[dim]);
                auto it = std::partition(list.begin() + startIdx, list.begin() + endIdx,
                                         [this, dim, split](int i) { return centroid(i)[dim] < split; });
                mid = (int)(it - list.begin());
                if (mid == startIdx || mid == endIdx) {
                    badSplits += 1;
                    mid = (startIdx + endIdx) / 2;
                    std::nth_element(list.begin() + startIdx, list.begin() + mid, list.begin() + endIdx,
                                     [this, dim](int x, int y) { return centroid(x)[dim] < centroid(y)[dim]; });
                }
                totalSplits += 1;
                depthField += 1;
                node->left = std::make_unique<Node>();
                construct(node->left, startIdx, mid - 1);
                node->right = std::make_unique<Node>();
                construct(node->right, mid, endIdx);
            }
        } else {
            node->leaf = true;
        }
    }

    void flatten(const std::unique_ptr<Node>& node) {  // Bvh.cpp:70-96
        pto_flatnode f;
        std::memset(&f, 0, sizeof f);
        f.bmin[0] = node->box.mn.x; f.bmin[1] = node->box.mn.y; f.bmin[2] = node->box.mn.z;
        f.bmax[0] = node->box.mx.x; f.bmax[1] = node->box.mx.y; f.bmax[2] = node->box.mx.z;
        f.startIdx = node->startIdx;
        f.endIdx = node->endIdx;
        f.leaf = node->leaf ? 1 : 0;
        f.axis = node->axis;
        f.triangles = node->endIdx - node->startIdx + 1;
        int index = (int)nodes.size();
        nodes.push_back(f);
        if (!node->leaf) {
            nodes[index].leftChild = index + 1;
            flatten(node->left);
            nodes[index].rightChild = (int)nodes.size();
            flatten(node->right);
        }
    }
};

static void tree_stats(const std::vector<pto_flatnode>& n, int i, int d, int& leaves, int& maxd, int& maxleaf) {
    if (n[i].leaf) {
        leaves++;
        if (d > maxd) maxd = d;
        if (n[i].triangles > maxleaf) maxleaf = n[i].triangles;
        return;
    }
    tree_stats(n, i + 1, d + 1, leaves, maxd, maxleaf);
    tree_stats(n, n[i].rightChild, d + 1, leaves, maxd, maxleaf);
}

}  // namespace pto

extern "C" int32_t pto_bvh_build(int32_t num_tris, const float* pos, int sah, int32_t* out_list, pto_flatnode* out_nodes,
                                 int32_t stats[6]) {
    using namespace pto;
    if (num_tris <= 0) return 0;
    Builder b{pos, sah != 0};
    b.list.resize(num_tris);
    for (int i = 0; i < num_tris; ++i) b.list[i] = i;
    auto root = std::make_unique<Node>();
    b.construct(root, 0, num_tris - 1);
    b.flatten(root);
    std::copy(b.list.begin(), b.list.end(), out_list);
    std::copy(b.nodes.begin(), b.nodes.end(), out_nodes);
    if (stats) {
        int leaves = 0, maxd = 0, maxleaf = 0;
        tree_stats(b.nodes, 0, 1, leaves, maxd, maxleaf);
        stats[0] = b.totalSplits; stats[1] = b.badSplits; stats[2] = leaves;
        stats[3] = maxd; stats[4] = maxleaf; stats[5] = b.depthField;
    }
    return (int32_t)b.nodes.size();
}
