// SAH / median BVH builder and flattener (host side, one-off per scene).  Behaviour follows reference src/Bvh.cpp:11-274
// including its quirks: ranges are sorted / partitioned with an END-EXCLUSIVE iterator at the INCLUSIVE endIdx, so the last
// triangle of a range never takes part (Bvh.cpp:125,174,206,215); a split at either end of the range falls back to the
// object median ("bad split"); `depth` counts splits.
#include "Bvh.hpp"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstring>
#include <future>
#include <cstdlib>
#include <thread>

namespace FW {

bool FlatNode::intersect(const Vec3f& orig, const Vec3f& invDir, const int sign[3], const float& t1) const {
    const Vec3f b[2] = {box.min, box.max};
    float lo = (b[sign[0]].x - orig.x) * invDir.x, hi = (b[1 - sign[0]].x - orig.x) * invDir.x;
    float ylo = (b[sign[1]].y - orig.y) * invDir.y, yhi = (b[1 - sign[1]].y - orig.y) * invDir.y;
    if ((lo > yhi) || (ylo > hi)) return false;
    if (ylo > lo) lo = ylo;
    if (yhi < hi) hi = yhi;
    float zlo = (b[sign[2]].z - orig.z) * invDir.z, zhi = (b[1 - sign[2]].z - orig.z) * invDir.z;
    if ((lo > zhi) || (zlo > hi)) return false;
    if (zlo > lo) lo = zlo;
    if (zhi < hi) hi = zhi;
    return (lo < t1) && (hi > 0.0f);
}

BVH::BVH(std::vector<RTTriangle>* triangles, bool sah, int numThreads) : m_sahEnabled(sah), m_triangles(triangles) {
    const size_t n = triangles->size();
    // default: the machine's cores, but never more than PT_HOST_THREADS / OMP_NUM_THREADS (or 16): hardware_concurrency() reports the
    // whole host on shared GPU boxes, many times this process's share
    int cap = 16;
    for (const char* var : {"PT_HOST_THREADS", "OMP_NUM_THREADS"})
        if (const char* e = std::getenv(var)) { int v = std::atoi(e); if (v > 0) { cap = v; break; } }
    m_numThreads = numThreads > 0 ? numThreads : std::min(cap, (int)std::max(1u, std::thread::hardware_concurrency()));
    list.resize(n);
    m_centroid.resize(n); m_min.resize(n); m_max.resize(n);
    for (size_t i = 0; i < n; ++i) {
        list[i] = (int32_t)i;
        m_centroid[i] = (*triangles)[i].centroid();
        m_min[i] = (*triangles)[i].min();
        m_max[i] = (*triangles)[i].max();
    }
    if (n == 0) return;
    root = std::make_unique<Node>();
    auto t0 = std::chrono::steady_clock::now();
    // spawnBudget = how many more times this branch may fork; 2^budget >= threads keeps every core busy on the top levels
    int budget = 0;
    while ((1 << budget) < m_numThreads) ++budget;
    constructTask(root, 0, (int)n - 1, m_numThreads > 1 ? budget + 2 : 0);
    auto t1 = std::chrono::steady_clock::now();
    badSplits = m_badSplitsA; totalSplits = m_totalSplitsA; depth = totalSplits;
    nodes.reserve(2 * n);
    flatten(root);
    auto t2 = std::chrono::steady_clock::now();
    buildMillis = std::chrono::duration<double, std::milli>(t1 - t0).count();
    flattenMillis = std::chrono::duration<double, std::milli>(t2 - t1).count();
    m_centroid.clear(); m_min.clear(); m_max.clear();
    m_centroid.shrink_to_fit(); m_min.shrink_to_fit(); m_max.shrink_to_fit();
}

void BVH::construct(std::unique_ptr<Node>& node, int startIdx, int endIdx) { constructTask(node, startIdx, endIdx, 0); }

void BVH::constructTask(std::unique_ptr<Node>& node, int startIdx, int endIdx, int spawnBudget) {
    node->box = computeBB(startIdx, endIdx);
    node->startIdx = startIdx;
    node->endIdx = endIdx;
    const int tris = endIdx - startIdx + 1;
    if (tris <= maxTrisInLeaf) { node->leaf = true; return; }

    int dim = 0;
    int mid = (startIdx + endIdx) / 2;
    auto first = list.begin() + startIdx, last = list.begin() + endIdx;  // NB: excludes list[endIdx]
    if (m_sahEnabled) {
        const float invArea = 1.0f / node->box.area();
        float minCost = FLT_MAX;
        std::vector<float> leftArea(tris - 1), rightArea(tris - 1);
        for (int d = 0; d <= 2; d++) {
            std::sort(first, last, [this, d](int a, int b) { return m_centroid[a][d] < m_centroid[b][d]; });
            AABB L(m_min[list[startIdx]], m_max[list[startIdx]]), R(m_min[list[endIdx]], m_max[list[endIdx]]);
            leftArea[0] = L.area();
            rightArea[0] = R.area();
            for (int i = 1; i < tris - 1; ++i) {
                L = Union(L, AABB(m_min[list[startIdx + i]], m_max[list[startIdx + i]]));
                leftArea[i] = L.area();
                R = Union(R, AABB(m_min[list[endIdx - i]], m_max[list[endIdx - i]]));
                rightArea[i] = R.area();
            }
            for (int i = 0; i < tris - 1; ++i) {
                const int nL = i + 1, nR = tris - 1 - i;
                float cost = 1.0f / maxTrisInLeaf + (nL * leftArea[i] + nR * rightArea[nR - 1]) * invArea;
                if (cost < minCost) { minCost = cost; dim = d; mid = startIdx + i + 1; }
            }
        }
        if (!(minCost < tris)) { node->leaf = true; return; }
        node->axis = dim;
        m_totalSplitsA++;
        if (mid == startIdx || mid == endIdx) { m_badSplitsA++; mid = (startIdx + endIdx) / 2; }
        std::nth_element(first, list.begin() + mid, last, [this, dim](int a, int b) { return m_centroid[a][dim] < m_centroid[b][dim]; });
    } else {
        AABB cb(m_centroid[list[startIdx]], m_centroid[list[startIdx]]);
        for (int i = startIdx + 1; i <= endIdx; ++i) cb = Union(cb, m_centroid[list[i]]);
        dim = cb.MaximumExtent();
        node->axis = dim;
        const float split = 0.5f * (cb.min[dim] + cb.max[dim]);
        auto it = std::partition(first, last, [this, dim, split](int i) { return m_centroid[i][dim] < split; });
        mid = (int)(it - list.begin());
        if (mid == startIdx || mid == endIdx) {
            m_badSplitsA++;
            mid = (startIdx + endIdx) / 2;
            std::nth_element(first, list.begin() + mid, last, [this, dim](int a, int b) { return m_centroid[a][dim] < m_centroid[b][dim]; });
        }
        m_totalSplitsA++;
    }
    node->leftChild = std::make_unique<Node>();
    node->rightChild = std::make_unique<Node>();
    if (spawnBudget > 0 && tris > 4096) {
        auto fut = std::async(std::launch::async, [this, &node, startIdx, mid, spawnBudget] {
            constructTask(node->leftChild, startIdx, mid - 1, spawnBudget - 1);
        });
        constructTask(node->rightChild, mid, endIdx, spawnBudget - 1);
        fut.get();
    } else {
        constructTask(node->leftChild, startIdx, mid - 1, 0);
        constructTask(node->rightChild, mid, endIdx, 0);
    }
}

void BVH::flatten(std::unique_ptr<Node>& node) {
    const int32_t index = (int32_t)nodes.size();
    nodes.emplace_back();
    {
        // the array is uploaded and cached byte for byte: zero the record first so the 3 padding bytes after `leaf` are defined
        FlatNode& f = nodes[index];
        std::memset(static_cast<void*>(&f), 0, sizeof(FlatNode));
        f.box = node->box;
        f.startIdx = node->startIdx;
        f.endIdx = node->endIdx;
        f.leaf = node->leaf;
        f.axis = node->axis;
        f.triangles = node->endIdx - node->startIdx + 1;
    }
    if (node->leaf) return;
    nodes[index].leftChild = index + 1;
    flatten(node->leftChild);
    nodes[index].rightChild = (int32_t)nodes.size();
    flatten(node->rightChild);
}

AABB BVH::computeBB(int startIdx, int endIdx) {
    const bool cached = !m_min.empty();
    auto lo = [&](int t) { return cached ? m_min[t] : (*m_triangles)[t].min(); };
    auto hi = [&](int t) { return cached ? m_max[t] : (*m_triangles)[t].max(); };
    Vec3f mn = lo(list[startIdx]), mx = hi(list[startIdx]);
    for (int i = startIdx + 1; i <= endIdx; ++i) {
        mn = FW::min(mn, lo(list[i]));
        mx = FW::max(mx, hi(list[i]));
    }
    return AABB(mn, mx);
}

AABB BVH::Union(AABB a, AABB b) { return AABB(FW::min(a.min, b.min), FW::max(a.max, b.max)); }
AABB BVH::Union(AABB a, Vec3f p) { return AABB(FW::min(a.min, p), FW::max(a.max, p)); }

int32_t BVH::trueDepth() const {
    if (nodes.empty()) return 0;
    int32_t best = 0;
    std::vector<std::pair<int32_t, int32_t>> st{{0, 1}};
    while (!st.empty()) {
        auto [i, d] = st.back();
        st.pop_back();
        if (d > best) best = d;
        if (!nodes[i].leaf) { st.push_back({nodes[i].rightChild, d + 1}); st.push_back({i + 1, d + 1}); }
    }
    return best;
}

}  // namespace FW
