// jtx_build_dev.hip -- Scene::rebuildBVH ON THE DEVICE (buildTree bvh.cpp:9-133, flattenBVH bvh.cpp:135-149,
// Scene::buildBVH scene.cpp:96-135; the edit loop that needs it: display.cpp:545-588 sets rebuildBVH_, :902-905 rebuilds).
//
// The reference's builder is a top-down recursion: bounds -> centroid bounds -> longest axis -> 12 centroid buckets ->
// 11 SAH split costs -> std::partition -> recurse.  Which primitives go left and right at a node depends only on the node's
// primitive SET (bucket(p) <= best bucket), never on their order, so the recursion can run breadth first: one level of the
// tree per round, every open node of the level at once --
//   k_bin_large / k_decide_large : nodes of more than 32 primitives: the bucket tables (count, box, centroid box per bucket) are
//                      reduced per workgroup in LDS (a workgroup's 256 consecutive positions touch at most 9 such nodes) and
//                      merged with a few global atomics per workgroup; one thread then runs the reference's forward /
//                      backward cost passes (bvh.cpp:74-90) on the 12 buckets
//   k_split_small    : nodes of 2..32 primitives: one thread per node, the 11 candidate splits evaluated directly
//   k_flags / scan / k_partition_slots / k_scatter : std::partition AS libstdc++ EXECUTES IT, in parallel: which element it swaps
//                      with which is a function of the predicate flags alone (see k_partition_slots), so one exclusive scan over
//                      all positions per level reproduces the reference's primitive order inside every node -- and with it
//                      the order inside leaves of several primitives, which decides hits at exactly equal distances
// -- every float the decision depends on is computed with the reference's own fp32 operations (centroid = 0.5 lo + 0.5 hi,
// offset = (c - lo) / (hi - lo), bucket = int(12 offset), cost = count * area, 0.5 + cost / area(parent)); min / max are
// exact and order-free.  The tree therefore equals the reference's NODE FOR NODE (boxes, split axes, child order, leaves,
// depth-first numbering), and Scene::triangles_ comes out in the reference's order (for the standard library the oracle and
// the host builder are compiled with: libstdc++'s std::partition / std::nth_element); scene_info.device_built only records
// where the tree was built.
// After the levels: a radix sort by (segment start, depth) IS the depth-first order of flattenBVH; subtree sizes by binary
// search in it; positions in the 8 direction-sign orderings by walking up the parents; the surface-area cut of the 8-ary
// nodes (WideBuilder::prepareCuts / fill of jtx_capi.hip, same arithmetic) level by level with a work queue.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"
#include "jtx_wide_quant.hpp"
#include "../../include/jtx_mi.h"
#include <hipcub/hipcub.hpp>
#include <cfloat>
#include <vector>

namespace jtx {

namespace {

constexpr int NB = 12;            // BVH_NUM_BUCKETS (bvh.cpp:60)
constexpr int SMALL = 32;         // nodes of at most this many primitives are split by one thread
constexpr int TAB = 13;           // ints per bucket: count, lo[3], hi[3], centroid lo[3], centroid hi[3] (ordered keys)
constexpr int MAXSLOT = 12;       // large nodes a workgroup of 256 positions can touch (256 / 33 + 2)

struct BNode {                    // a node of the tree under construction (ids in creation order: a level is a contiguous range)
    int start, count;             // its primitives: positions [start, start + count) of the order array
    int child;                    // first child (the second is child + 1); -1: leaf; -2: open = to be split in this round
    int depth, dim, best;         // split axis; last bucket of the left side (two primitives: -1 keep their order, -2 swap)
    int leftCount, aux, parent;   // aux: index of the node's bucket table (large nodes)
    float lo[3], hi[3], clo[3], chi[3];
};

// order-preserving map float -> int, so that integer atomicMin / atomicMax are float min / max (exact, order-free)
JD int fkey(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
JD float fval(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

JD float areaOf(const float lo[3], const float hi[3]) {                   // AABB::surfaceArea aabb.hpp:87-90
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
JD int longestAxis(const float lo[3], const float hi[3]) {               // aabb.hpp:51-56
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx > dy && dx > dz) return 0;
    if (dy > dz) return 1;
    return 2;
}
JD float centroidOf(float lo, float hi) { return 0.5f * lo + 0.5f * hi; }   // Triangle::centroid mesh.hpp:207-209
JD int bucketOf(float clo, float chi, float c) {                         // bvh.cpp:65-66 with AABB::offset aabb.hpp:58-64
    float o = c - clo;
    if (chi > clo) o /= chi - clo;
    int b = (int) (NB * o);
    if (b == NB) b = NB - 1;
    return b;
}
struct Box { float lo[3], hi[3];
    JD void clear() { for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }           // AABB() aabb.hpp:11-16
    JD void grow(const float l[3], const float h[3]) { for (int k = 0; k < 3; ++k) { lo[k] = l[k] < lo[k] ? l[k] : lo[k]; hi[k] = h[k] > hi[k] ? h[k] : hi[k]; } }
    JD void growPoint(const float p[3]) { grow(p, p); } };

struct Counters { int nodes, large; };       // nodes created so far; bucket tables handed out for the NEXT round
// The level a round works on lives ON THE DEVICE (round 4): the host enqueues rounds in batches with grids sized for the widest level
// the round could have (2^round nodes, at most one per primitive) and reads the state back once per batch -- three or four
// synchronisations per build instead of one per level (31 on the atrium).
struct LevelState { int ls, le, nlarge, pad; };   // the open level = nodes [ls, le); its nodes of more than SMALL primitives
__global__ void k_advance_level(Counters *cnt, LevelState *lv) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    lv->ls = lv->le; lv->le = cnt->nodes; lv->nlarge = cnt->large;
    cnt->large = 0;                               // the tables of the next level are handed out while its nodes are created
}

// a freshly created node decides at once whether it is a leaf (bvh.cpp:18-28, 30-46) or stays open for the next round
JD void classify(BNode &n, Counters *cnt) {
    n.child = -1; n.dim = 0; n.best = 0; n.leftCount = 0; n.aux = -1;
    if (areaOf(n.lo, n.hi) == 0 || n.count == 1) return;
    const int dim = longestAxis(n.clo, n.chi);
    if (n.clo[dim] == n.chi[dim]) return;
    n.dim = dim; n.child = -2;
    if (n.count > SMALL) n.aux = atomicAdd(&cnt->large, 1);
}

// cnt[1] (never a level's counters): .nodes = capacity of the node array, .large = why the build must be declined (1: a split left one side
// empty -- every SAH cost inf / NaN, where the reference's buildTree recurses for ever, bvh.cpp:96-111, 126-127; 2: node array full)
JD void makeChildren(BNode *nodes, int id, int best, int leftCount, const Box &lb, const Box &lc, const Box &rb, const Box &rc, Counters *cnt) {
    BNode &n = nodes[id];
    if (leftCount <= 0 || leftCount >= n.count) { atomicMax(&cnt[1].large, 1); n.child = -1; return; }
    const int c = atomicAdd(&cnt->nodes, 2);
    if (c + 2 > cnt[1].nodes) { atomicMax(&cnt[1].large, 2); atomicSub(&cnt->nodes, 2); n.child = -1; return; }
    n.child = c; n.best = best; n.leftCount = leftCount;
    for (int s = 0; s < 2; ++s) {
        BNode &ch = nodes[c + s];
        ch.start = s ? n.start + leftCount : n.start; ch.count = s ? n.count - leftCount : leftCount;
        ch.depth = n.depth + 1; ch.parent = id;
        const Box &b = s ? rb : lb, &cb = s ? rc : lc;
        for (int k = 0; k < 3; ++k) { ch.lo[k] = b.lo[k]; ch.hi[k] = b.hi[k]; ch.clo[k] = cb.lo[k]; ch.chi[k] = cb.hi[k]; }
        classify(ch, cnt);
    }
}

// ---- primitive boxes (Mesh::tBounds mesh.hpp:79-84 on Mesh::getVertices' transformed corners) ----
JD void xformPointB(const float *m, float x, float y, float z, float o[3]) {      // Transform::applyToPoint, row by row
    o[0] = m[0] * x + m[1] * y + m[2] * z + m[3];
    o[1] = m[4] * x + m[5] * y + m[6] * z + m[7];
    o[2] = m[8] * x + m[9] * y + m[10] * z + m[11];
}
__global__ void __launch_bounds__(256) k_prim_boxes(const float4 *prim_src, const float *mesh_xf, int np, float4 *blo, float4 *bhi, int *rootKeys) {
    __shared__ int red[12];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 12) red[threadIdx.x] = (threadIdx.x % 6) < 3 ? fkey(FLT_MAX) : fkey(-FLT_MAX);
    __syncthreads();
    if (i < np) {
        const float4 s0 = prim_src[5 * (size_t) i], s1 = prim_src[5 * (size_t) i + 1], s2 = prim_src[5 * (size_t) i + 2], s4 = prim_src[5 * (size_t) i + 4];
        const float *m = mesh_xf + 16 * (size_t) __float_as_int(s4.z);
        float v0[3], v1[3], v2[3];
        xformPointB(m, s0.x, s0.y, s0.z, v0); xformPointB(m, s0.w, s1.x, s1.y, v1); xformPointB(m, s1.z, s1.w, s2.x, v2);
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(fminf(v0[k], v1[k]), v2[k]); hi[k] = fmaxf(fmaxf(v0[k], v1[k]), v2[k]); }
        blo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f); bhi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        for (int k = 0; k < 3; ++k) {
            const float c = centroidOf(lo[k], hi[k]);
            atomicMin(&red[k], fkey(lo[k])); atomicMax(&red[3 + k], fkey(hi[k]));
            atomicMin(&red[6 + k], fkey(c)); atomicMax(&red[9 + k], fkey(c));
        }
    }
    __syncthreads();
    if (threadIdx.x < 12) { if ((threadIdx.x % 6) < 3) atomicMin(&rootKeys[threadIdx.x], red[threadIdx.x]); else atomicMax(&rootKeys[threadIdx.x], red[threadIdx.x]); }
}
__global__ void k_root_keys_init(int *rootKeys, Counters *cnt, int maxNodes) {
    if (threadIdx.x < 12) rootKeys[threadIdx.x] = (threadIdx.x % 6) < 3 ? fkey(FLT_MAX) : fkey(-FLT_MAX);
    if (threadIdx.x == 0) { cnt[0].nodes = 1; cnt[0].large = 0; cnt[1].nodes = maxNodes; cnt[1].large = 0; }
}
__global__ void k_root_init(BNode *nodes, const int *rootKeys, int np, Counters *cnt, const int *orig, int *order, int *prim_node) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < np) { order[orig[i]] = i; prim_node[i] = 0; }         // the builder's input order: the scene's own Scene::triangles (scene.cpp:97-100)
    if (i == 0) {
        BNode &n = nodes[0];
        n.start = 0; n.count = np; n.depth = 0; n.parent = -1;
        for (int k = 0; k < 3; ++k) { n.lo[k] = fval(rootKeys[k]); n.hi[k] = fval(rootKeys[3 + k]); n.clo[k] = fval(rootKeys[6 + k]); n.chi[k] = fval(rootKeys[9 + k]); }
        classify(n, cnt);
    }
}

// ---- one round ----
__global__ void __launch_bounds__(256) k_table_init(int *table, const LevelState *lv) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= lv->nlarge * NB) return;
    int *t = table + (size_t) e * TAB;
    t[0] = 0;
    for (int k = 0; k < 3; ++k) { t[1 + k] = fkey(FLT_MAX); t[4 + k] = fkey(-FLT_MAX); t[7 + k] = fkey(FLT_MAX); t[10 + k] = fkey(-FLT_MAX); }
}

__global__ void __launch_bounds__(256) k_bin_large(const BNode *nodes, const int *order, const int *prim_node, const float4 *blo, const float4 *bhi,
                                                   int np, int *table, const LevelState *lv) {
    if (lv->nlarge == 0) return;                  // (uniform: the whole grid leaves)
    __shared__ int ltab[MAXSLOT * NB * TAB];
    __shared__ int slotAux[MAXSLOT];
    __shared__ int waveTot[4];
    for (int e = threadIdx.x; e < MAXSLOT * NB; e += 256) {
        int *t = ltab + e * TAB;
        t[0] = 0;
        for (int k = 0; k < 3; ++k) { t[1 + k] = fkey(FLT_MAX); t[4 + k] = fkey(-FLT_MAX); t[7 + k] = fkey(FLT_MAX); t[10 + k] = fkey(-FLT_MAX); }
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    int node = -1; bool open = false; int nstart = 0, ndim = 0, naux = 0; float nclo = 0.0f, nchi = 0.0f;
    if (i < np) {
        node = prim_node[i];
        if (node >= 0) {
            const BNode &nd = nodes[node];
            open = nd.child == -2 && nd.count > SMALL;
            nstart = nd.start; ndim = nd.dim; naux = nd.aux; nclo = nd.clo[ndim]; nchi = nd.chi[ndim];
        }
    }
    // slot of the node inside this workgroup: segments are contiguous, so a node's first position here carries the flag
    const bool isStart = open && (i == nstart || threadIdx.x == 0);
    const unsigned long long bal = __ballot(isStart);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pre = __popcll(bal & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull)));     // inclusive count within the wave
    if (lane == 0) waveTot[w] = __popcll(bal);
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += waveTot[k];
    const int slot = base + pre - 1;
    const int nslots = waveTot[0] + waveTot[1] + waveTot[2] + waveTot[3];
    if (isStart && slot < MAXSLOT) slotAux[slot] = naux;
    __syncthreads();
    if (open && slot >= 0 && slot < MAXSLOT) {
        const int prim = order[i];
        const float4 l = blo[prim], h = bhi[prim];
        const float lo[3] = {l.x, l.y, l.z}, hi[3] = {h.x, h.y, h.z};
        const float c[3] = {centroidOf(lo[0], hi[0]), centroidOf(lo[1], hi[1]), centroidOf(lo[2], hi[2])};
        const int b = bucketOf(nclo, nchi, c[ndim]);
        int *t = ltab + (slot * NB + b) * TAB;
        atomicAdd(&t[0], 1);
        for (int k = 0; k < 3; ++k) {
            atomicMin(&t[1 + k], fkey(lo[k])); atomicMax(&t[4 + k], fkey(hi[k]));
            atomicMin(&t[7 + k], fkey(c[k])); atomicMax(&t[10 + k], fkey(c[k]));
        }
    }
    __syncthreads();
    const int used = nslots < MAXSLOT ? nslots : MAXSLOT;
    for (int e = threadIdx.x; e < used * NB; e += 256) {
        const int *t = ltab + e * TAB;
        if (t[0] == 0) continue;
        int *g = table + ((size_t) slotAux[e / NB] * NB + (e % NB)) * TAB;
        atomicAdd(&g[0], t[0]);
        for (int k = 0; k < 3; ++k) {
            atomicMin(&g[1 + k], t[1 + k]); atomicMax(&g[4 + k], t[4 + k]);
            atomicMin(&g[7 + k], t[7 + k]); atomicMax(&g[10 + k], t[10 + k]);
        }
    }
}

// bvh.cpp:69-127 on the 12 buckets of one node
__global__ void __launch_bounds__(64) k_decide_large(BNode *nodes, const LevelState *lv, const int *table, int maxPrims, Counters *cnt) {
    const int id = lv->ls + blockIdx.x * 64 + threadIdx.x;
    if (id >= lv->le || lv->nlarge == 0) return;
    BNode &n = nodes[id];
    if (n.child != -2 || n.count <= SMALL) return;
    const int *t = table + (size_t) n.aux * NB * TAB;
    int count[NB]; Box bb[NB], cc[NB];
    for (int b = 0; b < NB; ++b) {
        const int *e = t + b * TAB;
        count[b] = e[0];
        for (int k = 0; k < 3; ++k) { bb[b].lo[k] = fval(e[1 + k]); bb[b].hi[k] = fval(e[4 + k]); cc[b].lo[k] = fval(e[7 + k]); cc[b].hi[k] = fval(e[10 + k]); }
    }
    float cost[NB - 1];
    for (int i = 0; i < NB - 1; ++i) cost[i] = 0.0f;
    { int below = 0; Box acc; acc.clear();                                                  // bvh.cpp:74-81
      for (int i = 0; i < NB - 1; ++i) { below += count[i]; acc.grow(bb[i].lo, bb[i].hi); cost[i] += below * areaOf(acc.lo, acc.hi); } }
    { int above = 0; Box acc; acc.clear();                                                  // bvh.cpp:84-90
      for (int i = NB - 1; i > 0; --i) { above += count[i]; acc.grow(bb[i].lo, bb[i].hi); cost[i - 1] += above * areaOf(acc.lo, acc.hi); } }
    int best = -1; float bestCost = __builtin_inff();
    for (int i = 0; i < NB - 1; ++i) if (cost[i] < bestCost) { bestCost = cost[i]; best = i; }
    const float leafCost = (float) n.count;
    bestCost = 0.5f + bestCost / areaOf(n.lo, n.hi);
    if (!(n.count > maxPrims || bestCost < leafCost)) { n.child = -1; return; }              // bvh.cpp:105, 114-123: a leaf
    Box lb, lc, rb, rc; lb.clear(); lc.clear(); rb.clear(); rc.clear();
    int left = 0;
    for (int b = 0; b < NB; ++b) {
        if (count[b] == 0) continue;
        if (b <= best) { left += count[b]; lb.grow(bb[b].lo, bb[b].hi); lc.grow(cc[b].lo, cc[b].hi); }
        else { rb.grow(bb[b].lo, bb[b].hi); rc.grow(cc[b].lo, cc[b].hi); }
    }
    makeChildren(nodes, id, best, left, lb, lc, rb, rc, cnt);
}

// nodes of 2..SMALL primitives: one thread; the 11 candidate splits evaluated directly from the primitives (same sums:
// counts are integers, boxes min / max)
__global__ void __launch_bounds__(64) k_split_small(BNode *nodes, const LevelState *lv, const int *order, const float4 *blo, const float4 *bhi,
                                                    int maxPrims, Counters *cnt) {
    const int id = lv->ls + blockIdx.x * 64 + threadIdx.x;
    if (id >= lv->le) return;
    BNode &n = nodes[id];
    if (n.child != -2 || n.count > SMALL) return;
    const int dim = n.dim, cnt_n = n.count, start = n.start;
    auto loadPrim = [&](int j, float lo[3], float hi[3], float c[3]) {
        const int prim = order[start + j];
        const float4 l = blo[prim], h = bhi[prim];
        lo[0] = l.x; lo[1] = l.y; lo[2] = l.z; hi[0] = h.x; hi[1] = h.y; hi[2] = h.z;
        for (int k = 0; k < 3; ++k) c[k] = centroidOf(lo[k], hi[k]);
    };
    Box lb, lc, rb, rc; lb.clear(); lc.clear(); rb.clear(); rc.clear();
    if (cnt_n == 2) {                                                                       // bvh.cpp:50-57: nth_element around the middle
        float l0[3], h0[3], c0[3], l1[3], h1[3], c1[3];
        loadPrim(0, l0, h0, c0); loadPrim(1, l1, h1, c1);
        const bool swap = c1[dim] < c0[dim];
        if (swap) { lb.grow(l1, h1); lc.growPoint(c1); rb.grow(l0, h0); rc.growPoint(c0); }
        else { lb.grow(l0, h0); lc.growPoint(c0); rb.grow(l1, h1); rc.growPoint(c1); }
        makeChildren(nodes, id, swap ? -2 : -1, 1, lb, lc, rb, rc, cnt);
        return;
    }
    unsigned long long bk0 = 0ull, bk1 = 0ull;                                               // 4-bit bucket of primitive j
    for (int j = 0; j < cnt_n; ++j) {
        float lo[3], hi[3], c[3]; loadPrim(j, lo, hi, c);
        const unsigned long long b = (unsigned long long) bucketOf(n.clo[dim], n.chi[dim], c[dim]);
        if (j < 16) bk0 |= b << (4 * j); else bk1 |= b << (4 * (j - 16));
    }
    auto bucket = [&](int j) { return (int) ((j < 16 ? bk0 >> (4 * j) : bk1 >> (4 * (j - 16))) & 15ull); };
    int best = -1; float bestCost = __builtin_inff();
    for (int i = 0; i < NB - 1; ++i) {
        Box below, above; below.clear(); above.clear(); int nb = 0, na = 0;
        for (int j = 0; j < cnt_n; ++j) {
            float lo[3], hi[3], c[3]; loadPrim(j, lo, hi, c);
            if (bucket(j) <= i) { ++nb; below.grow(lo, hi); } else { ++na; above.grow(lo, hi); }
        }
        float cost = 0.0f;
        cost += nb * areaOf(below.lo, below.hi);
        cost += na * areaOf(above.lo, above.hi);
        if (cost < bestCost) { bestCost = cost; best = i; }
    }
    const float leafCost = (float) cnt_n;
    bestCost = 0.5f + bestCost / areaOf(n.lo, n.hi);
    if (!(cnt_n > maxPrims || bestCost < leafCost)) { n.child = -1; return; }
    int left = 0;
    for (int j = 0; j < cnt_n; ++j) {
        float lo[3], hi[3], c[3]; loadPrim(j, lo, hi, c);
        if (bucket(j) <= best) { ++left; lb.grow(lo, hi); lc.growPoint(c); } else { rb.grow(lo, hi); rc.growPoint(c); }
    }
    makeChildren(nodes, id, best, left, lb, lc, rb, rc, cnt);
}

JD bool goesLeft(const BNode &nd, int i, int prim, const float4 *blo, const float4 *bhi) {
    if (nd.count == 2) return (i == nd.start) ? nd.best == -1 : nd.best == -2;
    const float4 l = blo[prim], h = bhi[prim];
    const float lo = nd.dim == 0 ? l.x : (nd.dim == 1 ? l.y : l.z), hi = nd.dim == 0 ? h.x : (nd.dim == 1 ? h.y : h.z);
    return bucketOf(nd.clo[nd.dim], nd.chi[nd.dim], centroidOf(lo, hi)) <= nd.best;             // the partition predicate, bvh.cpp:107-111
}

__global__ void __launch_bounds__(256) k_flags(const BNode *nodes, const int *order, int *prim_node, const float4 *blo, const float4 *bhi, int np, int *flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    int f = 0;
    const int node = prim_node[i];
    if (node >= 0) {
        const BNode &nd = nodes[node];
        if (nd.child < 0) prim_node[i] = -1;                                                 // its node became a leaf: the primitive stays where it is
        else f = goesLeft(nd, i, order[i], blo, bhi) ? 1 : 0;
    }
    flags[i] = f;
}

// std::partition as libstdc++ runs it on random-access iterators (bvh.cpp:107-111 -> stl_algo.h __partition): walk inwards from
// both ends, swap the k-th "false" from the left with the k-th "true" from the right while they have not met.  Which elements
// meet is a function of the flags alone: with L = number of trues, the falses standing in [0, L) and the trues standing in
// [L, n) are equally many, the k-th of the one (from the left) swaps with the k-th of the other (from the right), everything
// else stays where it is.  Two passes: every mover writes its position into the slot of its rank, then takes its partner's.
__global__ void __launch_bounds__(256) k_partition_slots(const BNode *nodes, const int *prim_node, const int *flags, const int *scan, int np,
                                                         int *slotF, int *slotT) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const int node = prim_node[i];
    if (node < 0) return;
    const BNode &nd = nodes[node];
    if (nd.count == 2) return;                                            // nth_element of two: handled in k_scatter
    const int before = scan[i] - scan[nd.start];                          // trues in [start, i)
    const bool t = flags[i] != 0, inLeft = i < nd.start + nd.leftCount;
    if (!t && inLeft) slotF[nd.start + ((i - nd.start) - before)] = i;    // the k-th false from the left
    if (t && !inLeft) slotT[nd.start + (nd.leftCount - before - 1)] = i;  // the k-th true from the right
}

__global__ void __launch_bounds__(256) k_scatter(const BNode *nodes, const int *order, const int *prim_node, const int *flags, const int *scan,
                                                 const int *slotF, const int *slotT, int np, int *order2, int *prim_node2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const int node = prim_node[i], prim = order[i];
    if (node < 0) { order2[i] = prim; prim_node2[i] = -1; return; }
    const BNode &nd = nodes[node];
    const bool left = flags[i] != 0;
    int at = i;
    if (nd.count == 2) at = left ? nd.start : nd.start + 1;               // std::nth_element on two elements: insertion sort (stl_algo.h __introselect)
    else {
        const int before = scan[i] - scan[nd.start];
        const bool inLeft = i < nd.start + nd.leftCount;
        if (!left && inLeft) at = slotT[nd.start + ((i - nd.start) - before)];
        else if (left && !inLeft) at = slotF[nd.start + (nd.leftCount - before - 1)];
    }
    order2[at] = prim; prim_node2[at] = left ? nd.child : nd.child + 1;
}

// ---- depth-first numbering (flattenBVH bvh.cpp:135-149) ----
__global__ void __launch_bounds__(256) k_sort_keys(const BNode *nodes, int nn, unsigned long long *keys, int *vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    keys[i] = ((unsigned long long) nodes[i].start << 8) | (unsigned) nodes[i].depth;
    vals[i] = i;
}
__global__ void __launch_bounds__(256) k_dfs_of(const int *vals, int nn, int *dfsOf) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < nn) dfsOf[vals[r]] = r;
}
struct EmitOut { float4 *nbox; jtx_mi_bvh_node *hnodes; /* counters3: [0] leaves [1] max depth [2] a leaf of more than 65535 primitives */ int *parentDfs, *axisOf, *size, *leaf_nodes, *depthCount, *counters3; };   // counters3: [0] leaves [1] max depth
__global__ void __launch_bounds__(256) k_emit(const BNode *nodes, const int *vals, const int *dfsOf, const unsigned long long *keys, int nn, EmitOut o) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nn) return;
    const BNode &n = nodes[vals[r]];
    const bool leaf = n.child < 0;
    const int offset = leaf ? n.start : dfsOf[n.child + 1], nprims = leaf ? n.count : 0;
    o.nbox[2 * (size_t) r] = make_float4(n.lo[0], n.hi[0], n.lo[1], n.hi[1]);
    o.nbox[2 * (size_t) r + 1] = make_float4(n.lo[2], n.hi[2], __int_as_float(offset), __int_as_float(nprims));
    jtx_mi_bvh_node h;
    for (int k = 0; k < 3; ++k) { h.pmin[k] = n.lo[k]; h.pmax[k] = n.hi[k]; }
    if (nprims > 65535) atomicExch(&o.counters3[2], 1);                                     // uint16 numPrimitives, bvh.hpp:13
    h.offset = offset; h.num_prims = (uint16_t) nprims; h.axis = (uint8_t) (leaf ? 0 : n.dim); h.pad = 0;
    o.hnodes[r] = h;
    o.parentDfs[r] = n.parent >= 0 ? dfsOf[n.parent] : -1;
    o.axisOf[r] = leaf ? 0 : n.dim;
    // subtree size: the nodes of the subtree are exactly those behind r whose segment starts before the end of this one
    const unsigned long long endKey = (unsigned long long) (n.start + n.count) << 8;
    int lo = r + 1, hi = nn;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < endKey) lo = mid + 1; else hi = mid; }
    o.size[r] = lo - r;
    if (leaf) o.leaf_nodes[atomicAdd(&o.counters3[0], 1)] = r;
    else atomicAdd(&o.depthCount[n.depth], 1);
    atomicMax(&o.counters3[1], n.depth);
}
// wide levels: counters [0] granules used [1] wide nodes [2] failed [3] items of the next level [4] items of this level [5] levels filled
__global__ void k_wide_advance(int *c) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (c[4] > 0) c[5]++;
    c[4] = c[3]; c[3] = 0;
}
__global__ void __launch_bounds__(256) k_level_nodes(const BNode *nodes, const int *vals, int nn, const int *levelBegin, int *cursor, int *level_nodes) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nn) return;
    const BNode &n = nodes[vals[r]];
    if (n.child < 0) return;
    level_nodes[levelBegin[n.depth] + atomicAdd(&cursor[n.depth], 1)] = r;
}
// position of every node in the near-first depth-first order of each direction-sign octant (scene.cpp:40-46): walking up, a
// node stands behind each ancestor, and behind the ancestor's near subtree when it lies in the far one
__global__ void __launch_bounds__(256) k_positions(const float4 *nbox, const int *parentDfs, const int *axisOf, const int *size, int nn, int *pos) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= nn) return;
    int p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n = g;
    for (int par = parentDfs[n]; par >= 0; n = par, par = parentDfs[n]) {
        const int first = par + 1, second = __float_as_int(nbox[2 * (size_t) par + 1].z), axis = axisOf[par];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int near = ((k >> axis) & 1) ? second : first;
            p[k] += 1 + (n != near ? size[near] : 0);
        }
    }
    for (int k = 0; k < 8; ++k) pos[(size_t) k * nn + g] = p[k];
}

__global__ void __launch_bounds__(256) k_gather_prims(const int *order, int np, const float4 *src, const float4 *tris, const float4 *shade, const int *orig,
                                                      float4 *src2, float4 *tris2, float4 *shade2, int *orig2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const size_t j = (size_t) order[i];
    for (int k = 0; k < 5; ++k) src2[5 * (size_t) i + k] = src[5 * j + k];
    for (int k = 0; k < 3; ++k) tris2[3 * (size_t) i + k] = tris[3 * j + k];
    for (int k = 0; k < 4; ++k) shade2[4 * (size_t) i + k] = shade[4 * j + k];
    orig2[i] = orig[j];
}

// ---- 8-ary nodes: WideBuilder::prepareCuts / fill (jtx_capi.hip) ----
JD bool leafNode(const float4 *nbox, int i) { return __float_as_int(nbox[2 * (size_t) i + 1].w) != 0; }
JD int secondChild(const float4 *nbox, int i) { return __float_as_int(nbox[2 * (size_t) i + 1].z); }
JD void cornersOf(const float4 *nbox, int i, float lo[3], float hi[3]) {
    const float4 b0 = nbox[2 * (size_t) i], b1 = nbox[2 * (size_t) i + 1];
    lo[0] = b0.x; hi[0] = b0.y; lo[1] = b0.z; hi[1] = b0.w; lo[2] = b1.x; hi[2] = b1.y;
}
JD double areaD(const float4 *nbox, int i) {
    float lo[3], hi[3]; cornersOf(nbox, i, lo, hi);
    const double dx = (double) hi[0] - lo[0], dy = (double) hi[1] - lo[1], dz = (double) hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__global__ void __launch_bounds__(256) k_wide_cuts(const int *level_nodes, int begin, int count, const float4 *nbox, float *F) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= count) return;
    const int i = level_nodes[begin + j];                     // an interior node of this depth; its children are done
    const int L = i + 1, R = secondChild(nbox, i);
    float best[9];
    for (int k = 2; k <= 8; ++k) {
        float c = 3.0e38f;
        for (int a = 1; a < k; ++a) { const float v = F[8 * (size_t) L + a - 1] + F[8 * (size_t) R + (k - a) - 1]; if (v < c) c = v; }
        best[k] = c;
    }
    const float W = (float) areaD(nbox, i) + best[8];
    F[8 * (size_t) i] = W;
    for (int k = 2; k <= 8; ++k) F[8 * (size_t) i + k - 1] = best[k] < W ? best[k] : W;
}
struct WItem { int b, at; };           // binary node; granule of its wide node
struct WideOut { uint4 *wide; int *wide_map; int *counters; long long cap; };    // counters: see k_wide_advance
__global__ void __launch_bounds__(64) k_wide_fill(const WItem *in, WItem *out, const float4 *nbox, const int *axisOf, const float *F, WideOut o) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= o.counters[4] || o.counters[2]) return;
    const int b = in[j].b, at = in[j].at;
    struct TNode { int node, left, right; };
    TNode t[15]; int nt = 1;
    t[0].node = b; t[0].left = t[0].right = -1;
    {   // the cut of at most 8 subtrees that minimises the summed box area of the wide nodes below (k_wide_cuts)
        struct Job { int ti, k; } jobs[16]; int nj = 0;
        jobs[nj].ti = 0; jobs[nj].k = 8; ++nj;
        bool root = true;
        while (nj) {
            const Job jb = jobs[--nj];
            const int node = t[jb.ti].node;
            if (leafNode(nbox, node)) continue;
            const int L = node + 1, R = secondChild(nbox, node);
            int bi = 0; float bc = root ? 3.0e38f : F[8 * (size_t) node];
            for (int i = 1; i < jb.k; ++i) { const float c = F[8 * (size_t) L + i - 1] + F[8 * (size_t) R + (jb.k - i) - 1]; if (c < bc) { bc = c; bi = i; } }
            root = false;
            if (bi == 0) continue;
            const int l = nt, r = nt + 1;
            t[l].node = L; t[l].left = t[l].right = -1; t[r].node = R; t[r].left = t[r].right = -1; nt += 2;
            t[jb.ti].left = l; t[jb.ti].right = r;
            jobs[nj].ti = r; jobs[nj].k = jb.k - bi; ++nj;
            jobs[nj].ti = l; jobs[nj].k = bi; ++nj;
        }
    }
    int order[8], n = 0;
    { int st[16], sp = 0; st[sp++] = 0;
      while (sp) { const int i = st[--sp]; if (t[i].left < 0) order[n++] = i; else { st[sp++] = t[i].right; st[sp++] = t[i].left; } } }
    int slotOf[15]; for (int i = 0; i < 15; ++i) slotOf[i] = -1;
    int child[8], ni = 0, nl = 0;
    for (int k = 0; k < n; ++k) if (!leafNode(nbox, t[order[k]].node)) { slotOf[order[k]] = ni; child[ni++] = t[order[k]].node; }
    for (int k = 0; k < n; ++k) if (leafNode(nbox, t[order[k]].node)) { slotOf[order[k]] = ni + nl; child[ni + nl] = t[order[k]].node; ++nl; }
    float pmin[3], pmax[3]; cornersOf(nbox, b, pmin, pmax);
    jtxq::NodeGrid grid;
    if (!jtxq::nodeGrid(pmin, pmax, grid)) { atomicExch(&o.counters[2], 1); return; }
    uint8_t qlo[3][8] = {}, qhi[3][8] = {};
    for (int s = 0; s < ni + nl; ++s) {
        float cmin[3], cmax[3]; cornersOf(nbox, child[s], cmin, cmax);
        uint8_t lo3[3], hi3[3];
        if (!jtxq::quantiseChild(grid, pmin, pmax, cmin, cmax, lo3, hi3)) { atomicExch(&o.counters[2], 1); return; }
        for (int k = 0; k < 3; ++k) { qlo[k][s] = lo3[k]; qhi[k][s] = hi3[k]; }
    }
    uint32_t perm[8];
    for (int oc = 0; oc < 8; ++oc) {                          // visiting order per octant: near first inside the treelet (scene.cpp:40-46)
        uint32_t pm = 0; int cnt = 0;
        int st[16], sp = 0; st[sp++] = 0;
        while (sp) {
            const int i = st[--sp];
            if (t[i].left < 0) { pm |= (uint32_t) slotOf[i] << (3 * cnt++); continue; }
            const bool neg = (oc >> axisOf[t[i].node]) & 1;
            st[sp++] = neg ? t[i].left : t[i].right;
            st[sp++] = neg ? t[i].right : t[i].left;
        }
        perm[oc] = pm;
    }
    // children block [ni nodes][nl leaf records] (jtx_wide_quant.hpp)
    const int need = (int) jtxq::blockGranules(ni, nl);
    const long long base = (long long) atomicAdd(&o.counters[0], need);
    if (base + need > o.cap || base + need >= (long long) jtxq::kMaxGranules) { atomicExch(&o.counters[2], 1); return; }
    uint32_t ndw[16], tw[4 * jtxq::kTails];
    jtxq::encodeGridAndPlanes(ndw, grid, ni, ni + nl, qlo, qhi);
    if (!jtxq::encodeTail(tw, (uint32_t) base, perm, ni + nl)) { atomicExch(&o.counters[2], 1); return; }
    uint4 *nd = o.wide + at;
    for (int g = 0; g < 4; ++g) nd[g] = make_uint4(ndw[4 * g], ndw[4 * g + 1], ndw[4 * g + 2], ndw[4 * g + 3]);
    for (uint32_t t = 0; t < jtxq::kTails; ++t) nd[4 + t] = make_uint4(tw[4 * t], tw[4 * t + 1], tw[4 * t + 2], tw[4 * t + 3]);
    if (b == 0) {                                             // the root-peel record: group word, orders, the children's exact boxes
        uint32_t rec[4 * 14];
        for (int i = 0; i < 4 * 14; ++i) rec[i] = 0u;
        jtxq::encodePeelHeader(rec, (uint32_t) base, ni, ni + nl, perm);
        for (int s = 0; s < ni + nl; ++s) { float cmin[3], cmax[3]; cornersOf(nbox, child[s], cmin, cmax); jtxq::encodePeelBox(rec, s, cmin, cmax); }
        for (int g = 0; g < 14; ++g) o.wide[jtxq::kPeelRec + g] = make_uint4(rec[4 * g], rec[4 * g + 1], rec[4 * g + 2], rec[4 * g + 3]);
        for (int g = 14; g < (int) jtxq::kRootNode; ++g) o.wide[g] = make_uint4(0u, 0u, 0u, 0u);
        for (int g = (int) (jtxq::kRootNode + jtxq::kNodeG); g < (int) jtxq::kFirstBlock; ++g) o.wide[g] = make_uint4(0u, 0u, 0u, 0u);
    }
    int *rec = o.wide_map + 16 * (size_t) atomicAdd(&o.counters[1], 1);
    rec[0] = at; rec[1] = b; rec[2] = ni; rec[3] = nl; rec[4] = (int) jtxq::leafAt((uint32_t) base, ni, 0); rec[5] = rec[6] = rec[7] = 0;
    for (int s = 0; s < 8; ++s) rec[8 + s] = s < ni + nl ? child[s] : -1;
    for (int s = ni; s < ni + nl; ++s) {                      // leaf records: the exact box + primitivesOffset + numPrimitives
        float cmin[3], cmax[3]; cornersOf(nbox, child[s], cmin, cmax);
        const float4 b1 = nbox[2 * (size_t) child[s] + 1];
        uint4 *lr = o.wide + jtxq::leafAt((uint32_t) base, ni, s - ni);
        lr[0] = make_uint4(__float_as_uint(cmin[0]), __float_as_uint(cmax[0]), __float_as_uint(cmin[1]), __float_as_uint(cmax[1]));
        lr[1] = make_uint4(__float_as_uint(cmin[2]), __float_as_uint(cmax[2]), __float_as_uint(b1.z), __float_as_uint(b1.w));
    }
    if (ni) {
        const int q = atomicAdd(&o.counters[3], ni);
        for (int s = 0; s < ni; ++s) { out[q + s].b = child[s]; out[q + s].at = (int) jtxq::nodeAt((uint32_t) base, s); }
    }
}

inline unsigned blocks(size_t n, int b) { return (unsigned) ((n + b - 1) / b); }

struct Arena {            // one allocation for all temporaries of a build
    char *base = nullptr; size_t used = 0, cap = 0;
    template <class T> T *take(size_t n) { used = (used + 255) & ~(size_t) 255; T *p = (T *) (base + used); used += n * sizeof(T); return p; }
};

} // namespace

} // namespace jtx

using namespace jtx;

#define BCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)
#define LAUNCH(kernel, grid, block, ...) do { hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__); BCHK(hipGetLastError()); } while (0)

// `arena`: the builder's temporaries, owned by the caller (the scene handle) and only ever grown -- an edit loop rebuilds without a
// single allocation (round 3 took and gave back ~100 MB per call).  R.declined != nullptr: the build was not done for a stated reason
// (nothing written that the caller may use); the caller falls back to the host build.
static size_t arenaBytes(int np, size_t &scanBytes, size_t &sortBytes, hipStream_t st) {
    const size_t maxNodes = 2 * (size_t) np;
    scanBytes = sortBytes = 0;
    (void) hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, (int *) nullptr, (int *) nullptr, np, st);
    (void) hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, (unsigned long long *) nullptr, (unsigned long long *) nullptr, (int *) nullptr, (int *) nullptr,
                                              (int) maxNodes, 0, 64, st);
    const size_t maxLarge = (size_t) np / (SMALL + 1) + 2;
    return maxNodes * sizeof(BNode) + 2 * (size_t) np * sizeof(float4) + 8 * (size_t) np * sizeof(int) + 2 * maxLarge * NB * TAB * sizeof(int)
           + scanBytes + sortBytes + maxNodes * (2 * sizeof(unsigned long long) + 6 * sizeof(int)) + 8 * maxNodes * sizeof(float)
           + 2 * maxNodes * sizeof(WItem) + (64 << 10);
}

hipError_t jtx_device_build_reserve(int np, DevBuildArena &arenaMem, hipStream_t st) {
    if (np <= 0) return hipSuccess;
    size_t a, b;
    const size_t need = arenaBytes(np, a, b, st);
    if (arenaMem.cap < need) {
        if (arenaMem.base) { BCHK(hipStreamSynchronize(st)); (void) hipFree(arenaMem.base); arenaMem.base = nullptr; arenaMem.cap = 0; }
        BCHK(hipMalloc(&arenaMem.base, need));
        arenaMem.cap = need;
    }
    return hipSuccess;
}

hipError_t jtx_device_build(const DevBuildBuffers &B, DevBuildArena &arenaMem, DevBuildResult &R, hipStream_t st) {
    const int np = B.np;
    R = DevBuildResult{};
    if (np <= 0) return hipSuccess;
    const size_t maxNodes = 2 * (size_t) np;
    Arena arena;
    // ---- temporaries ----
    size_t scanBytes = 0, sortBytes = 0;
    arena.cap = arenaBytes(np, scanBytes, sortBytes, st);
    const size_t maxLarge = (size_t) np / (SMALL + 1) + 2;
    BCHK(jtx_device_build_reserve(np, arenaMem, st));
    arena.base = (char *) arenaMem.base;
    BNode *nodes = arena.take<BNode>(maxNodes);
    float4 *blo = arena.take<float4>(np), *bhi = arena.take<float4>(np);
    int *order[2] = {arena.take<int>(np), arena.take<int>(np)};
    int *pnode[2] = {arena.take<int>(np), arena.take<int>(np)};
    int *flags = arena.take<int>(np), *scan = arena.take<int>(np), *slotF = arena.take<int>(np), *slotT = arena.take<int>(np);
    int *table = arena.take<int>(maxLarge * NB * TAB);
    void *scanTmp = arena.take<char>(scanBytes), *sortTmp = arena.take<char>(sortBytes);
    unsigned long long *keys = arena.take<unsigned long long>(maxNodes), *keys2 = arena.take<unsigned long long>(maxNodes);
    int *vals = arena.take<int>(maxNodes), *vals2 = arena.take<int>(maxNodes), *dfsOf = arena.take<int>(maxNodes);
    int *parentDfs = arena.take<int>(maxNodes), *axisOf = arena.take<int>(maxNodes);
    float *F = arena.take<float>(8 * maxNodes);
    WItem *items[2] = {arena.take<WItem>(maxNodes), arena.take<WItem>(maxNodes)};
    int *small = arena.take<int>(512);            // rootKeys[12] | Counters[2] @16 | LevelState @24 | counters3 @32 | depthCount[128] @64 | cursor[128] @192 | levelBegin[129] @320 | wide counters[6] @480
    int *rootKeys = small; Counters *cnt = (Counters *) (small + 16); LevelState *lv = (LevelState *) (small + 24);
    int *counters3 = small + 32, *depthCount = small + 64, *cursor = small + 192, *levelBeginDev = small + 320, *wideCnt = small + 480;
    if (arena.used > arena.cap) return hipErrorOutOfMemory;

    BCHK(hipMemsetAsync(small, 0, 512 * sizeof(int), st));
    LAUNCH(k_root_keys_init, dim3(1), dim3(64), rootKeys, cnt, (int) maxNodes);
    LAUNCH(k_prim_boxes, dim3(blocks(np, 256)), dim3(256), B.prim_src, B.mesh_xf, np, blo, bhi, rootKeys);
    LAUNCH(k_root_init, dim3(blocks(np, 256)), dim3(256), nodes, rootKeys, np, cnt, B.orig, order[0], pnode[0]);
    LAUNCH(k_advance_level, dim3(1), dim3(1), cnt, lv);                                   // level 0 = the root

    // ---- the rounds: one level of the tree each, enqueued in batches ----
    int cur = 0, rounds = 0;
    LevelState h{};
    while (true) {
        const int batch = rounds == 0 ? 12 : 6;
        for (int b = 0; b < batch; ++b, ++rounds) {
            const size_t wl = rounds < 30 ? ((size_t) 1 << rounds < (size_t) np ? (size_t) 1 << rounds : (size_t) np) : (size_t) np;   // nodes a level can have
            const size_t wlarge = wl < maxLarge ? wl : maxLarge;
            LAUNCH(k_table_init, dim3(blocks(wlarge * NB, 256)), dim3(256), table, lv);
            LAUNCH(k_bin_large, dim3(blocks(np, 256)), dim3(256), nodes, order[cur], pnode[cur], blo, bhi, np, table, lv);
            LAUNCH(k_decide_large, dim3(blocks(wl, 64)), dim3(64), nodes, lv, table, B.max_prims, cnt);
            LAUNCH(k_split_small, dim3(blocks(wl, 64)), dim3(64), nodes, lv, order[cur], blo, bhi, B.max_prims, cnt);
            LAUNCH(k_flags, dim3(blocks(np, 256)), dim3(256), nodes, order[cur], pnode[cur], blo, bhi, np, flags);
            BCHK(hipcub::DeviceScan::ExclusiveSum(scanTmp, scanBytes, flags, scan, np, st));
            LAUNCH(k_partition_slots, dim3(blocks(np, 256)), dim3(256), nodes, pnode[cur], flags, scan, np, slotF, slotT);
            LAUNCH(k_scatter, dim3(blocks(np, 256)), dim3(256), nodes, order[cur], pnode[cur], flags, scan, slotF, slotT, np, order[cur ^ 1], pnode[cur ^ 1]);
            cur ^= 1;
            LAUNCH(k_advance_level, dim3(1), dim3(1), cnt, lv);
        }
        Counters guard{};
        BCHK(hipMemcpyAsync(&h, lv, sizeof h, hipMemcpyDeviceToHost, st));
        BCHK(hipMemcpyAsync(&guard, cnt + 1, sizeof guard, hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
        if (guard.large == 1) { R.declined = "buildTree does not terminate on this geometry: the SAH costs of a node are all inf / NaN (box areas overflow fp32), "
                                             "the partition leaves one side empty (bvh.cpp:96-111, 126-127)"; return hipSuccess; }
        if (guard.large == 2 || (size_t) h.le > maxNodes) { R.declined = "more nodes than a binary tree over these primitives can have"; return hipSuccess; }
        if (h.le == h.ls) break;                                                           // the last round opened nothing: the tree stands
        if (rounds > 120) { R.declined = "the tree is deeper than 120 levels (the reference's traversal stack holds 64, scene.cpp:13)"; return hipSuccess; }
    }
    const int nn = h.le;
    R.nn = nn;

    // ---- depth-first numbering, host-format nodes, sizes, lists ----
    int endBit = 8; while (endBit < 64 && ((unsigned long long) np << 8) >> endBit) ++endBit;
    LAUNCH(k_sort_keys, dim3(blocks(nn, 256)), dim3(256), nodes, nn, keys, vals);
    BCHK(hipcub::DeviceRadixSort::SortPairs(sortTmp, sortBytes, keys, keys2, vals, vals2, nn, 0, endBit, st));
    LAUNCH(k_dfs_of, dim3(blocks(nn, 256)), dim3(256), vals2, nn, dfsOf);
    EmitOut eo{B.nbox, (jtx_mi_bvh_node *) B.hnodes, parentDfs, axisOf, B.size, B.leaf_nodes, depthCount, counters3};
    LAUNCH(k_emit, dim3(blocks(nn, 256)), dim3(256), nodes, vals2, dfsOf, keys2, nn, eo);
    int hs[3 + 128];
    BCHK(hipMemcpyAsync(hs, counters3, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
    BCHK(hipMemcpyAsync(hs + 3, depthCount, 128 * sizeof(int), hipMemcpyDeviceToHost, st));
    BCHK(hipStreamSynchronize(st));
    R.nleaves = hs[0]; R.max_depth = hs[1];
    if (R.max_depth >= 127) { R.declined = "the tree is deeper than 126 levels"; return hipSuccess; }
    if (hs[2]) { R.declined = "a leaf holds more than 65535 primitives (LinearBVHNode::numPrimitives is 16 bits, bvh.hpp:13)"; return hipSuccess; }
    int maxInterior = -1;
    for (int d = 0; d < 128; ++d) if (hs[3 + d] > 0) maxInterior = d;
    R.level_begin.assign((maxInterior > 0 ? maxInterior : 0) + 2, 0);                      // as jtx_mi_scene_create lays it out
    for (int d = 0; d <= maxInterior; ++d) R.level_begin[d + 1] = R.level_begin[d] + hs[3 + d];
    {
        // (a SYNCHRONOUS copy: the runtime may read a pageable source of hipMemcpyAsync when the copy EXECUTES, not when it is
        //  enqueued -- a block-scoped array was gone by then, and k_level_nodes scattered through garbage offsets: an intermittent
        //  illegal access, caught by the round-3 test runs.  The stream is idle here: the sync above.)
        int lb[129] = {0};
        for (size_t d = 0; d < R.level_begin.size() && d < 129; ++d) lb[d] = R.level_begin[d];
        BCHK(hipMemcpy(levelBeginDev, lb, 129 * sizeof(int), hipMemcpyHostToDevice));
    }
    LAUNCH(k_level_nodes, dim3(blocks(nn, 256)), dim3(256), nodes, vals2, nn, levelBeginDev, cursor, B.level_nodes);
    LAUNCH(k_positions, dim3(blocks(nn, 256)), dim3(256), B.nbox, parentDfs, axisOf, B.size, nn, B.pos);
    LAUNCH(k_gather_prims, dim3(blocks(np, 256)), dim3(256), order[cur], np, B.prim_src, B.tris, B.shade, B.orig, B.prim_src_out, B.tris_out, B.shade_out, B.orig_out);
    BCHK(hipMemcpyAsync(B.order, order[cur], (size_t) np * sizeof(int), hipMemcpyDeviceToDevice, st));

    // ---- the 8-ary nodes: levels enqueued in batches of 8, the work queue and its counters stay on the device ----
    R.wide_ok = false; R.num_wide = 0; R.wide_depth = 0; R.wide_granules = 0;
    if (B.wide && nn >= 2 && R.level_begin.size() >= 2) {
        BCHK(hipMemsetAsync(F, 0, 8 * (size_t) nn * sizeof(float), st));
        for (int d = (int) R.level_begin.size() - 2; d >= 0; --d) {
            const int count = R.level_begin[d + 1] - R.level_begin[d];
            if (count > 0) LAUNCH(k_wide_cuts, dim3(blocks(count, 256)), dim3(256), B.level_nodes, R.level_begin[d], count, B.nbox, F);
        }
        int wc[6] = {(int) jtxq::kFirstBlock, 0, 0, 0, 1, 0};
        WItem rootItem{0, (int) jtxq::kRootNode};
        BCHK(hipStreamSynchronize(st));
        BCHK(hipMemcpy(wideCnt, wc, sizeof wc, hipMemcpyHostToDevice));
        BCHK(hipMemcpy(items[0], &rootItem, sizeof rootItem, hipMemcpyHostToDevice));
        WideOut wo{B.wide, B.wide_map, wideCnt, (long long) B.wide_cap};
        const size_t maxWide = (size_t) nn / 2 + 1;
        int wcur = 0, level = 0;
        bool ok = true;
        while (true) {
            for (int b = 0; b < 8; ++b, ++level) {
                size_t wl = 1; for (int i = 0; i < level && wl < maxWide; ++i) wl *= 8;
                if (wl > maxWide) wl = maxWide;
                LAUNCH(k_wide_fill, dim3(blocks(wl, 64)), dim3(64), items[wcur], items[wcur ^ 1], B.nbox, axisOf, F, wo);
                LAUNCH(k_wide_advance, dim3(1), dim3(1), wideCnt);
                wcur ^= 1;
            }
            BCHK(hipMemcpyAsync(wc, wideCnt, sizeof wc, hipMemcpyDeviceToHost, st));
            BCHK(hipStreamSynchronize(st));
            if (wc[2]) { ok = false; break; }
            if (wc[4] == 0) break;
            if (level >= 64) { ok = false; break; }
        }
        R.wide_ok = ok; R.wide_depth = wc[5]; R.num_wide = wc[1]; R.wide_granules = (size_t) wc[0];
    }
    BCHK(hipStreamSynchronize(st));
    return hipGetLastError();
}
