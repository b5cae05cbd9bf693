// jtx_bvh_build.cpp -- host-side BVH construction for the MI355X core.
//
// Produces exactly the tree Scene::buildBVH builds in the reference (scene.cpp:96-135): recursive
// binned SAH with 12 buckets (bvh.cpp:9-133), cost 0.5 + sum(count*SA)/SA(parent), mid split by
// nth_element for two primitives, degenerate centroid bounds -> leaf, and the depth-first
// LinearBVHNode order of flattenBVH (bvh.cpp:135-149) with the first child implicit at i+1.
// Unlike the reference it never materialises a pointer tree: buildTree's recursion order IS the
// flattened (pre-order) order and an interior box equals the bounds of its primitives (min/max are
// exact), so nodes are emitted straight into the linear array.
#include "jtx_host.hpp"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <future>
#include <limits>
#include <thread>

namespace jtxh {

namespace {

struct Bounds {
    float lo[3], hi[3];
    Bounds() {                                                   // AABB() aabb.hpp:11-16
        for (int i = 0; i < 3; ++i) { lo[i] = std::numeric_limits<float>::max(); hi[i] = std::numeric_limits<float>::lowest(); }
    }
    void grow(const float p[3]) { for (int i = 0; i < 3; ++i) { lo[i] = p[i] < lo[i] ? p[i] : lo[i]; hi[i] = p[i] > hi[i] ? p[i] : hi[i]; } }
    void grow(const Bounds &b) { for (int i = 0; i < 3; ++i) { lo[i] = b.lo[i] < lo[i] ? b.lo[i] : lo[i]; hi[i] = b.hi[i] > hi[i] ? b.hi[i] : hi[i]; } }
    float area() const {                                         // aabb.hpp:87-90
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    int longestAxis() const {                                    // aabb.hpp:51-56
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx > dy && dx > dz) return 0;
        if (dy > dz) return 1;
        return 2;
    }
    float offset(const float p[3], int a) const {                // aabb.hpp:58-64
        float o = p[a] - lo[a];
        if (hi[a] > lo[a]) o /= hi[a] - lo[a];
        return o;
    }
};

struct Prim {
    int index, mesh, orig;
    Bounds b;
    float centroid(int a) const { return 0.5f * b.lo[a] + 0.5f * b.hi[a]; }   // mesh.hpp:207-209
};

constexpr int kBuckets = 12;

// A subtree is built into arrays of its OWN (node indices and primitive offsets relative to the subtree), so that the two
// children of a big node can be built on different threads and stitched behind their parent: buildTree's recursion order IS
// the flattened pre-order, the left subtree's nodes follow the parent and the right subtree's follow those.  The tree is
// the sequential one node for node -- the threads only decide who computes which part.
struct Subtree {
    std::vector<jtx_mi_bvh_node> nodes;
    std::vector<Prim> ordered;
    int maxDepth = 0;
};

struct Builder {
    int maxPrims;
    int threadsLeft;                                  // spawn budget (decremented on the spawning thread only: deterministic shape is irrelevant, results are)

    static void emitLeaf(Subtree &t, int slot, Prim *p, size_t n) {
        jtx_mi_bvh_node &ln = t.nodes[slot];
        ln.offset = (int) t.ordered.size();
        if (n > 65535) throw std::runtime_error("BVH leaf with more than 65535 primitives (uint16 numPrimitives, bvh.hpp:13)");
        ln.num_prims = (uint16_t) n;
        t.ordered.insert(t.ordered.end(), p, p + n);
    }

    static int bucketOf(const Bounds &cb, const Prim &q, int dim) {
        float c[3] = {q.centroid(0), q.centroid(1), q.centroid(2)};
        int b = (int) (kBuckets * cb.offset(c, dim));
        if (b == kBuckets) b = kBuckets - 1;
        return b;
    }

    // appends `sub` (built with indices relative to itself) behind the nodes already in `t`
    static void append(Subtree &t, Subtree &&sub) {
        const int nodeBase = (int) t.nodes.size(), primBase = (int) t.ordered.size();
        for (jtx_mi_bvh_node n : sub.nodes) {
            n.offset += n.num_prims ? primBase : nodeBase;
            t.nodes.push_back(n);
        }
        t.ordered.insert(t.ordered.end(), sub.ordered.begin(), sub.ordered.end());
        if (sub.maxDepth > t.maxDepth) t.maxDepth = sub.maxDepth;
    }

    void build(Subtree &t, Prim *p, size_t n, int depth) {
        const int slot = (int) t.nodes.size();
        t.nodes.push_back(jtx_mi_bvh_node{});
        if (depth > t.maxDepth) t.maxDepth = depth;

        Bounds bounds;
        for (size_t i = 0; i < n; ++i) bounds.grow(p[i].b);
        for (int a = 0; a < 3; ++a) { t.nodes[slot].pmin[a] = bounds.lo[a]; t.nodes[slot].pmax[a] = bounds.hi[a]; }

        if (bounds.area() == 0 || n == 1) return emitLeaf(t, slot, p, n);                // bvh.cpp:18-28

        Bounds cb;
        for (size_t i = 0; i < n; ++i) { float c[3] = {p[i].centroid(0), p[i].centroid(1), p[i].centroid(2)}; cb.grow(c); }
        const int dim = cb.longestAxis();
        if (cb.lo[dim] == cb.hi[dim]) return emitLeaf(t, slot, p, n);                    // bvh.cpp:36-46

        size_t mid = n / 2;
        if (n == 2) {                                                                   // bvh.cpp:50-57
            std::nth_element(p, p + mid, p + n, [dim](const Prim &a, const Prim &b) { return a.centroid(dim) < b.centroid(dim); });
        } else {
            int count[kBuckets] = {};
            Bounds bb[kBuckets];
            for (size_t i = 0; i < n; ++i) { const int b = bucketOf(cb, p[i], dim); count[b]++; bb[b].grow(p[i].b); }
            float cost[kBuckets - 1] = {};
            { int below = 0; Bounds acc;                                                // bvh.cpp:74-81
              for (int i = 0; i < kBuckets - 1; ++i) { below += count[i]; acc.grow(bb[i]); cost[i] += below * acc.area(); } }
            { int above = 0; Bounds acc;                                                // bvh.cpp:84-90
              for (int i = kBuckets - 1; i > 0; --i) { above += count[i]; acc.grow(bb[i]); cost[i - 1] += above * acc.area(); } }
            int best = -1; float bestCost = std::numeric_limits<float>::infinity();
            for (int i = 0; i < kBuckets - 1; ++i) if (cost[i] < bestCost) { bestCost = cost[i]; best = i; }
            const float leafCost = (float) n;
            bestCost = 0.5f + bestCost / bounds.area();
            if ((int) n > maxPrims || bestCost < leafCost) {                            // bvh.cpp:105-112
                Prim *m = std::partition(p, p + n, [&](const Prim &q) { return bucketOf(cb, q, dim) <= best; });
                mid = (size_t) (m - p);
                // Where every split cost is inf or NaN (boxes whose surface area overflows fp32) no cost is < INF, minBucket stays -1
                // (bvh.cpp:96-101), the partition puts everything on one side and the reference's buildTree recurses on the same span for
                // ever (bvh.cpp:126-127).  There is no tree to reproduce: say so instead of running out of memory.
                if (mid == 0 || mid == n)
                    throw std::runtime_error("buildTree does not terminate on this geometry: the SAH costs of a node are all inf / NaN (box areas "
                                             "overflow fp32), std::partition leaves one side empty (bvh.cpp:96-111, 126-127)");
            } else {
                return emitLeaf(t, slot, p, n);
            }
        }
        t.nodes[slot].axis = (uint8_t) dim;
        t.nodes[slot].num_prims = 0;
        if (n >= 16384 && threadsLeft > 0) {
            // big node: the right subtree on another thread, into arrays of its own, stitched behind the left one
            --threadsLeft;
            Builder other{maxPrims, threadsLeft / 2};
            threadsLeft -= other.threadsLeft;
            auto rightJob = [&other, p, mid, n, depth] { Subtree r; other.build(r, p + mid, n - mid, depth + 1); return r; };
            std::future<Subtree> fut;
            try { fut = std::async(std::launch::async, rightJob); }
            catch (const std::system_error &) {}                     // no thread to be had (a process / thread limit): this one builds it itself
            build(t, p, mid, depth + 1);                             // first child lands at slot + 1
            Subtree right = fut.valid() ? fut.get() : rightJob();
            t.nodes[slot].offset = (int) t.nodes.size();            // secondChildOffset (bvh.cpp:146)
            append(t, std::move(right));
        } else {
            build(t, p, mid, depth + 1);
            t.nodes[slot].offset = (int) t.nodes.size();
            build(t, p + mid, n - mid, depth + 1);
        }
    }
};

inline void xformPoint(const float m[16], const float *v, float out[3]) {   // Transform::applyToPoint (DESIGN.md math spec)
    for (int r = 0; r < 3; ++r) out[r] = m[4 * r + 0] * v[0] + m[4 * r + 1] * v[1] + m[4 * r + 2] * v[2] + m[4 * r + 3];
}

} // namespace

void meshVertices(const jtx_mi_mesh &m, int tri, float v0[3], float v1[3], float v2[3]) {   // Mesh::getVertices mesh.hpp:71-77
    const int32_t *i = m.indices + 3 * (size_t) tri;
    xformPoint(m.transform, m.vertices + 3 * (size_t) i[0], v0);
    xformPoint(m.transform, m.vertices + 3 * (size_t) i[1], v1);
    xformPoint(m.transform, m.vertices + 3 * (size_t) i[2], v2);
}

void buildBVH(const jtx_mi_scene_desc &d, BvhResult &out) {
    const size_t n = (size_t) d.num_tri_refs;
    std::vector<Prim> work(n);
    for (size_t i = 0; i < n; ++i) {
        const jtx_mi_tri_ref &r = d.tri_refs[i];
        if (r.mesh_index < 0 || r.mesh_index >= d.num_meshes) throw std::runtime_error("tri_ref.mesh_index out of range");
        const jtx_mi_mesh &m = d.meshes[r.mesh_index];
        if (r.index < 0 || r.index >= m.num_triangles) throw std::runtime_error("tri_ref.index out of range");
        for (int k = 0; k < 3; ++k) {
            const int32_t vi = m.indices[3 * (size_t) r.index + k];
            if (vi < 0 || vi >= m.num_vertices) throw std::runtime_error("mesh index out of range");
        }
        float v0[3], v1[3], v2[3];
        meshVertices(m, r.index, v0, v1, v2);
        work[i].index = r.index; work[i].mesh = r.mesh_index; work[i].orig = (int) i;
        work[i].b = Bounds(); work[i].b.grow(v0); work[i].b.grow(v1); work[i].b.grow(v2);   // tBounds mesh.hpp:79-84
    }
    Subtree tree;
    tree.nodes.reserve(2 * n + 1); tree.ordered.reserve(n);
    int hw = (int) std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    Builder b{d.max_prims_in_node > 0 ? d.max_prims_in_node : 1, (getenv("JTX_BVH_THREADS") ? atoi(getenv("JTX_BVH_THREADS")) : (hw > 32 ? 32 : hw)) - 1};
    if (n) b.build(tree, work.data(), n, 0);
    out.nodes = std::move(tree.nodes);
    out.max_depth = tree.maxDepth;
    out.refs.resize(n); out.orig.resize(n);
    for (size_t i = 0; i < n; ++i) { out.refs[i] = jtx_mi_tri_ref{tree.ordered[i].index, tree.ordered[i].mesh}; out.orig[i] = tree.ordered[i].orig; }
    out.scene_radius = 0;
    if (!out.nodes.empty()) {                                        // getSceneRadius scene.hpp:81-84
        const jtx_mi_bvh_node &r = out.nodes[0];
        const float dx = r.pmax[0] - r.pmin[0], dy = r.pmax[1] - r.pmin[1], dz = r.pmax[2] - r.pmin[2];
        out.scene_radius = std::sqrt(dx * dx + dy * dy + dz * dz) / 2;
    }
}

} // namespace jtxh
